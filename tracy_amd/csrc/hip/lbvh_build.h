// lbvh_build.h -- device-side build of the module's BVH.
//
// The reference builds its kd-tree on the host at start-up (kdtree.h:160-354, seconds on the dragon)
// and its CUDA module has no acceleration structure at all (cuda_trace.cu:22-46). The default builder
// of this module is the host-side binned-SAH build (bvh_builder.h); this is the alternative for
// scenes whose start-up time matters (1.6 M triangles in 22 ms instead of 350 ms, the dragon in 10
// instead of 25; the render is 2-5 % slower), built entirely on the GPU: 63-bit Morton codes of the
// triangle centroids and one radix sort group the triangles into Morton clusters of about two; the
// UPPER levels are binned SAH over the cluster boxes (a launch per level, a workgroup per node); below
// a cluster the ranges are split by binned SAH as well (a launch per generation, a wave per range),
// with the host builder's depth guard; ranges of <= 2 triangles are leaves, made by one launch at the
// end (a thread per leaf: box, link, bottom-up box fit);
// records compacted and emitted directly in the two formats the kernels traverse (bvh_builder.h).
// History: round 1 was a plain linear BVH (splits at the highest differing Morton bit: 28-33 % slower
// to render than the SAH tree), round 2 put an SAH top tree over <= 32768 clusters, built by one host
// thread (15-18 %), round 3 moved it to the device, shrank the clusters and replaced the Morton splits.
// The arrays stay in HBM; only the leaf-order -> input-triangle permutation goes back to the host (it
// orders the shading records).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../host/bvh_builder.h"

namespace lbvh {

struct DeviceBvh {
    float* nodes = nullptr;         // node_count x 16 floats   (64-byte fp32 records)
    uint32_t* nodes16 = nullptr;    // node_count x 8 words     (32-byte fp16 records)
    float* tris = nullptr;          // tri_count x 12 floats    (48-byte Moller-Trumbore records, leaf order)
    uint32_t* prim_index = nullptr; // tri_count: leaf-order position -> input triangle
    uint32_t node_count = 0, tri_count = 0, max_depth = 0, max_leaf = 0;
    float frame_center[3] = {0.f, 0.f, 0.f};
    float frame_scale = 1.f;
    double fp16_inflation = 1.0;
    double sah_cost = 0.0;
    double device_ms = 0.0;         // HIP-event time of the build kernels + sort (without the triangle upload)
};

constexpr size_t kMinTriangles = 9;   // smaller scenes fit one leaf: the caller uses the host builder

// Builds on the current device, on `stream`. Returns 0 on success; on failure `err` names the HIP call
// or the limit (2 = a tree deeper than bvh::kMaxDepth, which the depth guard rules out: the caller would
// fall back to the host builder).
int build(const std::vector<bvh::InputTriangle>& tris, hipStream_t stream, DeviceBvh& out, std::string& err);

// copies the device arrays into a bvh::Result (for bvh::validate and the tests)
int download(const DeviceBvh& d, hipStream_t stream, bvh::Result& out, std::string& err);

void release(DeviceBvh& d);

}  // namespace lbvh
