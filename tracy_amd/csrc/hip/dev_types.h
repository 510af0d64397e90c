// dev_types.h -- POD types shared by the host side of the C ABI and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TRH_STACK_SIZE 32      // LDS traversal-stack entries per lane == bvh::kMaxDepth (the kernels add one sentinel row)
#define TRH_BLOCK 256          // threads per workgroup (4 wave64)
#define TRH_WAVES_PER_BLOCK (TRH_BLOCK / 64)
#define TRH_MISS_ID 0xffffffffu

// 64-byte material record (src/material.h:103-117 restated as POD)
struct DevMaterial {
    float albedo[3];
    float roughness;
    float metalness;
    float ior;
    float emissive[3];
    float translucent;
    uint32_t tex[5];   // basecolor, normal, roughness, metalness, emissive; 0xffffffff = none
    uint32_t pad;
};
static_assert(sizeof(DevMaterial) == 64, "DevMaterial must be one 64-byte record");

struct DevTexture {
    const float4* texels;   // RGBA32F, row 0 first (src/texture.h:90)
    int32_t width;
    int32_t height;
};

// Everything a kernel needs, passed by value as a kernel argument.
struct DevScene {
    // exactly one of the two is non-NULL (chosen at upload from the builder's inflation measure):
    const float4* nodes;       // 4 float4 per fp32 inner-node record (see bvh_builder.h)
    const uint4* nodes16;      // 2 uint4 per compact record: fp16 bounds in the BVH frame + 2 child links
    const float4* tris;        // 3 float4 per primitive, leaf order: (v0, id) (e1, 0) (e2, 0)
    const float4* shade;       // 4 float4 per primitive: n0 n1 n2 (9) uv0 uv1 uv2 (6) material (1)
    const float4* tangents;    // 3 float4 per primitive (t0 t1 t2), NULL if no normal map is used
    const float4* mesh_aabb;   // 2 float4 per mesh: the reference's Mesh::GetAABB (min, max), unpadded
    const DevMaterial* materials;
    const DevTexture* textures;
    int32_t width, height;
    float width_f, height_f;   // (float)width, (float)height: as arguments they are scalar operands, computed in the kernel they are two more long-lived VGPRs
    uint32_t max_bounces;
    uint32_t flags;
    float eye[3];
    float ivp[16];             // column-major inverse(projection * view)
    float cam_k[4];            // ivp column 2 * 1.f + ivp column 3 * 1.f: the ray-independent half of Camera::GetRayFrom's mat4 * vec4 (set with the camera)
    float frame_c[3];          // BVH frame: x' = (x - frame_c) * frame_s, frame_s a power of two
    float frame_sinv;          // 1 / frame_s
    float frame_inv_lim;       // largest |1/d| whose product with frame_sinv stays finite
    uint32_t debug_view;       // trh_debug_view (0 = render radiance)
    float view3[9];            // mat3(Camera::GetView()), column-major: the NORMALS debug view
};

struct RenderParams {
    uint32_t spp_begin;
    uint32_t spp_count;
    uint32_t spp_magic;       // floor(2^32 / spp_count) (2^32 - 1 for one sample): item / spp_count as a multiply-high + one correction
    uint32_t seed32;          // pcg_hash of the folded 64-bit seed
    uint32_t tiles_x, tiles_y;
    uint32_t tile_w_log2, tile_h_log2;   // wave tile = (1 << tile_w_log2) x (1 << tile_h_log2) pixels
    uint32_t stack_entries;   // LDS stack rows per lane (BVH depth + the sentinel row)
    uint32_t refill_thresh;   // lanes waiting for shade/refill that end a traversal phase
    uint32_t sched_beta4;     // inner step is chosen when 4*n_inner >= sched_beta4*n_leaf (0 = while-while)
    uint32_t acc_copies;          // LDS accumulator copies per wave (1, 2, 4, 8)
    const uint32_t* tile_order;   // NULL: raster order; else the order the tile counter indexes (tile_order.h)
    uint32_t* tile_cost;          // NULL, or per-tile shader-clock ticks of this launch (input of the next launch's order)
    // k_megakernel_ab only: the wave's ray ring. Every resident wave owns 4 arrays of pq_cap float4 (wave slot = blockIdx.x *
    // TRH_WAVES_PER_BLOCK + wave in block): three hold a path (origin, direction, throughput, RNG, bounce | pixel), the
    // fourth the hit of its current ray.
    float4* pq;
    uint32_t pq_cap;              // entries per wave: a power of two, >= 128
    uint32_t pq_target;           // the camera phase stops once this many rays are ready to be traversed (<= pq_cap - 64)
};

// device-side counters, one cache line
struct DevCounters {
    unsigned long long rays, samples, box_tests, tri_tests, hits, tex_fetches;
    // wave-level iteration counts (instrumented kernel only): lane utilisation per phase =
    // lane-level count / (64 * wave-level count)
    unsigned long long wave_inner_iters, wave_leaf_iters, wave_shade_calls, wave_gen_calls, lanes_gen;
    // shader-clock cycles the waves spent per phase (instrumented kernel only): inner steps, triangle steps, shade, refill/generate, whole kernel
    unsigned long long phase_cycles[5];
    unsigned long long max_wave_cycles;   // instrumented kernel only: the longest-lived wave (load balance = mean / max)
    unsigned long long waves;             // instrumented kernel only: waves that reported
    unsigned long long idle_leaf_in_inner, idle_wait_in_inner, idle_none_in_inner;   // instrumented kernel only: lanes standing on a leaf / waiting for shade / without a path, summed over inner steps
    unsigned long long min_start, max_end, sum_start, sum_end;   // instrumented kernel only: wave start / end times (device-wide 100 MHz clock)
};
