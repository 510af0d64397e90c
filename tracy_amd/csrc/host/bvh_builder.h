// bvh_builder.h -- host-side build of the hip module's own acceleration structure.
//
// The reference's CPU module builds a median-split kd-tree with duplicated triangles
// (src/kdtree.h:160-282) and its CUDA module uses no acceleration structure at all
// (cuda_trace.cu:22-46). This module replaces both with one flat, world-space BVH2 built with
// binned SAH and laid out for CDNA4: one 64-byte record per inner node that carries BOTH
// child boxes (a lane fetches one cache line per traversal step and tests two boxes), leaves
// encoded in the child links, triangles re-ordered into leaf order as 48-byte Moller-Trumbore
// records (v0, e1, e2 + packed reference id).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bvh {

constexpr int kMaxDepth = 32;     // == traversal stack entries per lane (LDS)
constexpr int kMaxLeafSize = 8;   // leaf count is packed in 4 bits
constexpr int kSahBins = 16;
constexpr double kMaxFp16Inflation = 1.25;

struct InputTriangle {
    float v0[3], v1[3], v2[3];
    uint32_t packed_id;  // (mesh << 24) | triangle, as cpu_details.h:34
};

// 64-byte inner-node record (16 floats):
//   [0..3]   c0.min.x c0.max.x c0.min.y c0.max.y
//   [4..7]   c1.min.x c1.max.x c1.min.y c1.max.y
//   [8..11]  c0.min.z c0.max.z c1.min.z c1.max.z
//   [12..15] child0 child1 (int bits) 0 0
// child >= 0: index of an inner-node record; child < -1: leaf, ~child = (first_prim << 4) | count
// (count may be 0 for the padding leaf of tiny scenes; the value -1 is never produced)
//
// Compact form of the same record, 32 bytes (8 x u32), the one the kernels traverse: bounds as IEEE
// half floats in the FRAME x' = (x - frame_center) * frame_scale (frame_scale a power of two, the
// centre = per-axis median of the triangle centroids so that fp16 is finest where the geometry is
// dense), minima rounded toward -inf and maxima toward +inf AFTER the padding above, so a quantised
// box always contains the fp32 box:
//   [0] c0.min.x | c0.max.x << 16   [1] c0.min.y | c0.max.y << 16
//   [2] c1.min.x | c1.max.x << 16   [3] c1.min.y | c1.max.y << 16
//   [4] c0.min.z | c0.max.z << 16   [5] c1.min.z | c1.max.z << 16
//   [6] child0                      [7] child1
struct Result {
    std::vector<float> nodes;          // 16 floats per inner node, node 0 is the root
    std::vector<uint32_t> nodes16;     // 8 words per inner node, same order
    float frame_center[3] = {0.f, 0.f, 0.f};
    float frame_scale = 1.f;           // power of two
    // sum of the child-box half areas of all records, quantised / exact (>= 1): how much more often a
    // random ray enters a box because of the fp16 rounding. Near 1 when the geometry is dense around the
    // frame centre; large when small triangles sit far from it (a half float has 11 significant bits).
    // The module falls back to the fp32 records above kMaxFp16Inflation.
    double fp16_inflation = 1.0;
    std::vector<float> tris;           // 12 floats per primitive in leaf order: v0.xyz id | e1.xyz 0 | e2.xyz 0
    std::vector<uint32_t> prim_index;  // leaf-order position -> index into the input triangle array
    uint32_t max_depth = 0;
    uint32_t max_leaf = 0;
    double sah_cost = 0.0;
};

void build(const std::vector<InputTriangle>& tris, Result& out);

// C_traverse of the SAH (cost of one inner-record step in units of one triangle test; TRH_SAH_CT overrides it): shared with
// the device-side builder so that Result::sah_cost / trh_bvh_info.sah_cost of the two builders are on one scale
float cost_traverse();

// On-disk cache of build results. build_cached() looks in `dir` for a file named after the
// FNV-1a hash of the input triangles (and of the builder's constants and file-format version), checks its header,
// sizes and a checksum of the arrays, and returns its contents; on a miss or any mismatch it builds, then writes the
// file (to a temporary name, then rename: readers never see a partial file). from_cache_out tells which happened.
// An empty dir means build(). The file is this builder's own format ("TRHBVH1\0", little endian, arrays as they are
// in Result); a cache written by another version of the builder is ignored, not migrated.
void build_cached(const std::vector<InputTriangle>& tris, Result& out, const std::string& dir, bool* from_cache_out);

// Structural self-check of a build result (host only, used by trh_bvh_build_host and the CPU tests):
// every triangle in exactly one leaf, every leaf box contains its triangles' vertices, every fp16 box
// contains the fp32 box it was rounded from, depth / leaf size / link encoding within the documented
// limits, every record reachable exactly once. Returns the number of violations (0 = sound) and a text
// for the first one. hash_out: FNV-1a over the three arrays (layout determinism across thread counts).
uint64_t validate(const std::vector<InputTriangle>& tris, const Result& r, std::string& first_violation, uint64_t* hash_out);

}  // namespace bvh
