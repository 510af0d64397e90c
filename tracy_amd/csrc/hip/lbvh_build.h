// lbvh_build.h -- device-side build of the module's BVH.
//
// The reference builds its kd-tree on the host at start-up (kdtree.h:160-354, seconds on the dragon)
// and its CUDA module has no acceleration structure at all (cuda_trace.cu:22-46). The default builder
// of this module is the host-side binned-SAH build (bvh_builder.h); this is the alternative for
// scenes whose start-up time matters more than traversal speed (1.6 M triangles in 30 ms instead of
// 320 ms; the render is 5-17 % slower): a linear BVH built on the GPU -- 63-bit Morton codes of the
// triangle centroids, one radix sort, a top-down pass per tree level that splits every range where
// its highest differing code bit flips (with the host builder's depth guard), a bottom-up box fit,
// ranges of <= 2 triangles as leaves, records compacted and emitted directly in the two formats the
// kernels traverse (bvh_builder.h). Round 2: the UPPER levels, where the Morton order clusters badly,
// are built with binned SAH over the boxes of <= 32768 Morton clusters (a few hundred KB to the host
// and back, a few ms on one host thread): 45 -> 34 box tests per ray on the dragon. The arrays stay in
// HBM; only the leaf-order -> input-triangle permutation goes back to the host (it orders the
// shading records).
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
