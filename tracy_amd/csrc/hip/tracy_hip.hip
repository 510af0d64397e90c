// tracy_hip.hip -- gfx950 kernels and the C ABI of include/tracy_hip.h.
//
// Kernels (all hand-written HIP for CDNA4, wave64; STATS = instrumented build, N16 = 32-byte fp16
// node records instead of the 64-byte fp32 fallback):
//   k_megakernel<STATS, N16>         default pipeline: persistent waves pull tiles (8x8 pixels x S samples);
//                                    lanes are not bound to pixels, a wave-local scheduler hands the tile's
//                                    camera paths to lanes as they fall idle and votes between inner-node
//                                    steps and triangle steps (see the comment at the kernel).
//   k_megakernel_simple<STATS, N16>  first-generation kernel (lane == pixel), kept for A/B measurements.
//   wavefront.h                      generate / extend / shade / accumulate pipeline with queues in HBM.
//   k_primary_hits<N16>              parity probe (a): nearest hit of un-jittered pixel-centre rays.
//   k_sample_radiance<N16>           parity probe (b): radiance of one sample index per pixel.
//   k_resolve                        mean = sum / count (CPUDetails::UpdateOutput, cpu_details.cpp:187-199).
//   k_present                        tonemap + BGRA8 picture (CPUDetails::Tonemap/UpdateBitmap, :201-243).
// Traversal stacks live in LDS, one column per lane (entry i at stack[i * 256]) so that
// ds_read/ds_write_b32 are bank-conflict free at any depth; the megakernel sizes them to the tree's depth.
#include "../../../include/tracy_hip.h"
#include "../host/bvh_builder.h"
#include "lbvh_build.h"
#include "tile_order.h"
#include "dev_trace.h"
#include "ctx_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace dev;

// =====================================================================================
// kernels
// =====================================================================================

DEVFN uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool STATS>
DEVFN void flush_counters(const Counters& c, DevCounters* g) {
    const uint32_t rays = wave_sum(c.rays);
    uint32_t bt = 0, tt = 0, hh = 0, tf = 0;
    if (STATS) { bt = wave_sum(c.box_tests); tt = wave_sum(c.tri_tests); hh = wave_sum(c.hits); tf = wave_sum(c.tex_fetches); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g->rays, (unsigned long long)rays);
        if (STATS) {
            atomicAdd(&g->box_tests, (unsigned long long)bt);
            atomicAdd(&g->tri_tests, (unsigned long long)tt);
            atomicAdd(&g->hits, (unsigned long long)hh);
            atomicAdd(&g->tex_fetches, (unsigned long long)tf);
        }
    }
    if (STATS) {
        const uint32_t a = wave_sum(c.w_inner), b = wave_sum(c.w_leaf), d = wave_sum(c.w_shade), e = wave_sum(c.w_gen), f = wave_sum(c.l_gen);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&g->wave_inner_iters, (unsigned long long)a);
            atomicAdd(&g->wave_leaf_iters, (unsigned long long)b);
            atomicAdd(&g->wave_shade_calls, (unsigned long long)d);
            atomicAdd(&g->wave_gen_calls, (unsigned long long)e);
            atomicAdd(&g->lanes_gen, (unsigned long long)f);
            for (int k = 0; k < 5; ++k) atomicAdd(&g->phase_cycles[k], c.cyc[k]);
            atomicMax(&g->max_wave_cycles, c.cyc[4]);
            atomicAdd(&g->idle_leaf_in_inner, (unsigned long long)c.idle_leaf);
            atomicAdd(&g->idle_wait_in_inner, (unsigned long long)c.idle_wait);
            atomicAdd(&g->idle_none_in_inner, (unsigned long long)c.idle_none);
            atomicAdd(&g->waves, 1ull);
            // device-wide constant-rate clock (the shader clock of c.cyc[] is per XCD and not comparable across waves)
            const unsigned long long t_end = wall_clock64(), t_start = c.t_start;
            atomicMax(&g->min_start, ~t_start);   // the counters start at zero: keep the maximum of the complement
            atomicMax(&g->max_end, t_end);
            atomicAdd(&g->sum_start, t_start);
            atomicAdd(&g->sum_end, t_end);
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_megakernel_simple: first-generation kernel, kept for A/B measurements (TRH_PIPELINE_MEGAKERNEL_SIMPLE).
// One lane owns one pixel and loops over its samples; a wave waits for its slowest lane every bounce.
// ---------------------------------------------------------------------------------------
template <bool STATS, bool N16>
__global__ __launch_bounds__(TRH_BLOCK) void k_megakernel_simple(DevScene sc, RenderParams rp, float4* __restrict__ accum,
                                                                 DevCounters* __restrict__ counters, unsigned int* __restrict__ tile_counter) {
    __shared__ int s_stack[(TRH_STACK_SIZE + 1) * TRH_BLOCK];
    int* stack = &s_stack[threadIdx.x];
    const int lane = threadIdx.x & 63;
    const uint32_t ntiles = rp.tiles_x * rp.tiles_y;
    const uint32_t seed_hash = rp.seed32;
    Counters cnt = {};

    for (;;) {
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(tile_counter, 1u);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= ntiles) break;  // wave-uniform exit: every wave reaches it
        const int px = (int)(tile % rp.tiles_x) * 8 + (lane & 7);
        const int py = (int)(tile / rp.tiles_x) * 8 + (lane >> 3);
        const bool valid = px < sc.width && py < sc.height;
        uint32_t s = rp.spp_begin;
        const uint32_t end = valid ? rp.spp_begin + rp.spp_count : rp.spp_begin;
        V3 acc = mk(0.f, 0.f, 0.f);
        PathState ps;
        ps.o = ps.d = ps.throughput = ps.radiance = mk(0.f, 0.f, 0.f);
        ps.rng = 0;
        ps.bounce = 0;
        bool alive = false;
        for (;;) {
            if (!alive && s < end) {
                generate(sc, ps, px, py, s, seed_hash);
                alive = true;
            }
            if (__ballot(alive) == 0ull) break;
            if (alive) {
                Hit hit;
                intersect<STATS, N16>(sc, ps.o, ps.d, hit, stack, cnt);
                alive = shade<STATS>(sc, ps, hit, cnt);
                if (!alive) {
                    acc = acc + ps.radiance;  // samples are summed in increasing sample index
                    ++s;
                }
            }
        }
        if (valid) {
            const size_t p = (size_t)py * sc.width + px;
            float4 a = accum[p];
            a.x += acc.x; a.y += acc.y; a.z += acc.z; a.w += (float)rp.spp_count;
            accum[p] = a;
        }
    }
    flush_counters<STATS>(cnt, counters);
}

// ---------------------------------------------------------------------------------------
// k_megakernel: persistent waves with a wave-local path scheduler.
//
// A wave pulls a tile (P pixels x S samples = P*S camera paths, "items") from a global counter.
// Lanes are NOT bound to pixels: whenever a lane's path ends it takes the next item of the tile
// (ballot + mbcnt prefix, no memory traffic), so a tile that mixes sky pixels (1 ray per sample)
// and dragon pixels (3-5 rays) keeps all 64 lanes busy until the tile's items run out.
// Inside an item the lane alternates between a resumable BVH traversal (trav_inner/trav_leaf)
// and shading. The wave stays in the traversal loop until `refill_thresh` lanes have finished
// their ray (or nobody is traversing); only then it pays for the divergent shade + refill code,
// with at least that many lanes active. Radiance is accumulated per pixel in LDS (ds_add_f32)
// and flushed to the float4 accumulation buffer once per tile by plain stores: a tile is owned
// by exactly one wave per launch, so no global atomics are needed.
// LDS per workgroup: stack_entries x 256 ints (traversal stacks, one bank-conflict-free column
// per lane) + 4 waves x P x 3 floats (tile accumulators).
// ---------------------------------------------------------------------------------------
#ifndef TRH_INNER_REPEAT
#define TRH_INNER_REPEAT 3
#endif
// Waves per SIMD the megakernel is compiled for. 5 (<= 96 VGPRs; 6 dwords spilled, all tile-level or in the rare LDS add)
// against 4 (119 VGPRs, no spill): dragon -10 %, Cornell -8.5 % at 64 spp per launch once the hot-path spills were gone
// (camera constants precomputed on the host, dev_trace.h camera_ray); with them the fifth wave was worth 2.7 %. Launches
// of a few samples are slower (dragon 8 spp +7 %: more waves, fewer tiles each). 6 waves (80 VGPRs, 23 dwords spilled in
// shade / generate) lose 15 % even when the tree is shallow enough for six workgroups' stacks. profiles/r02_occupancy.txt
#ifndef TRH_MEGA_WAVES_PER_EU
#define TRH_MEGA_WAVES_PER_EU 5
#endif
template <bool STATS, bool N16>
__global__ __launch_bounds__(TRH_BLOCK) __attribute__((amdgpu_waves_per_eu(TRH_MEGA_WAVES_PER_EU, TRH_MEGA_WAVES_PER_EU)))
void k_megakernel(DevScene sc, RenderParams rp, float4* __restrict__ accum,
                                                          DevCounters* __restrict__ counters, unsigned int* __restrict__ tile_counter) {
    extern __shared__ int s_dyn[];
    s_dyn[threadIdx.x] = -1;   // sentinel row of this lane's stack column (dev_trace.h)
    // Register diet (profiles/r02_register_diet.md): nothing derived from the thread index stays in a VGPR across the
    // persistent loop. The wave's index is a scalar, the lane index is recomputed where it is needed (two v_mbcnt),
    // and a lane finds its stack column again from its stack pointer (column = address mod one row).
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t P_log2 = rp.tile_w_log2 + rp.tile_h_log2;
    const uint32_t P = 1u << P_log2;
    // Tile accumulators: rp.acc_copies (1, 2, 4 or 8) copies per wave, a lane adds into copy `lane mod copies`. The lanes of
    // a wave hold samples of the same one or two pixels, so with one copy their ds_add_f32 would all hit the same three
    // addresses and serialise (measured: 7 % of the launch). The copy stride is odd, so the copies of a pixel lie in
    // different LDS banks.
    const uint32_t acc_stride = P * 3u + 1u;
    float* const wacc = reinterpret_cast<float*>(s_dyn + rp.stack_entries * TRH_BLOCK) + wave_in_block * (acc_stride * rp.acc_copies);   // wave-uniform
    const uint32_t ntiles = rp.tiles_x * rp.tiles_y;
    const uint32_t total = P * rp.spp_count;   // items per tile
    Counters cnt = {};
    const unsigned long long t_kernel = STATS ? __builtin_readcyclecounter() : 0ull;
    if (STATS) cnt.t_start = wall_clock64();

    for (;;) {
        const uint32_t lane = fresh_lane_id();
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(tile_counter, 1u);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= ntiles) break;  // wave-uniform exit: every wave reaches it
        if (rp.tile_order) tile = __builtin_amdgcn_readfirstlane(rp.tile_order[tile]);   // expensive tiles first (tile_order.h)
        const unsigned long long t_tile = rp.tile_cost ? __builtin_readcyclecounter() : 0ull;
        const int tile_x0 = (int)((tile % rp.tiles_x) << rp.tile_w_log2);
        const int tile_y0 = (int)((tile / rp.tiles_x) << rp.tile_h_log2);
        for (uint32_t i = lane; i < acc_stride * rp.acc_copies; i += 64) wacc[i] = 0.f;
        __builtin_amdgcn_wave_barrier();

        uint32_t next = 0;          // wave-uniform: next unassigned item of this tile
        PathState ps;
        ps.o = ps.d = ps.throughput = ps.radiance = mk(0.f, 0.f, 0.f);
        ps.rng = 0;
        ps.bounce = 0;
        Trav tv;
        tv.inv = tv.oi = mk(0.f, 0.f, 0.f);
        tv.selx = tv.sely = tv.selz = 0x03020100u;
        tv.node = -1;
        tv.top = s_dyn + (wave_in_block * 64u + lane);
        tv.hit.t = 0.f; tv.hit.u = tv.hit.v = 0.f; tv.hit.prim = -1; tv.hit.id = TRH_MISS_ID;
        uint32_t pix = 0;           // pixel-in-tile of the item this lane works on
        bool has_path = false;      // a traversal result is waiting to be shaded
        bool exhausted = false;     // no items left for this lane

        for (;;) {
            // ---- shade the lanes whose ray is finished
            const unsigned long long t_shade = STATS ? __builtin_readcyclecounter() : 0ull;
            if (has_path && !trav_active(tv)) {
                const bool cont = shade<STATS>(sc, ps, tv.hit, cnt);
                // Radiance goes to the tile accumulator as soon as it is collected (an emitter hit, or the sky at the end of
                // the path) and is not carried through the next traversal: three registers less in the hot loop. A path adds
                // the same terms to its pixel as before; only a path that meets light twice adds them one by one instead of
                // their sum. (Most paths of an emitter-lit scene under a black sky carry no light at all: no add.)
                if (ps.radiance.x != 0.f || ps.radiance.y != 0.f || ps.radiance.z != 0.f) {
                    float* const lane_acc = wacc + ((fresh_lane_id() & (rp.acc_copies - 1u)) * acc_stride + pix * 3u);
                    __hip_atomic_fetch_add(&lane_acc[0], ps.radiance.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&lane_acc[1], ps.radiance.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&lane_acc[2], ps.radiance.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ps.radiance = mk(0.f, 0.f, 0.f);
                if (cont) trav_begin(sc, tv, ps.o, ps.d, stack_column(s_dyn, tv.top));
                else has_path = false;
            }
            const unsigned long long t_refill = STATS ? __builtin_readcyclecounter() : 0ull;
            if (STATS) cnt.cyc[2] += t_refill - t_shade;
            // ---- refill: idle lanes take the next items of the tile, in lane order
            const bool want = !has_path && !exhausted;
            const unsigned long long wmask = __ballot(want);
            if (wmask != 0ull) {
                if (want) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
                    const uint32_t item = next + rank;
                    if (item < total) {
                        // items run sample-major: the 64 lanes of a wave hold samples of one or two pixels, so their camera
                        // rays walk the same nodes (-5 % dragon, -6 % spheres against pixel-major order)
                        // item / spp_count without the compiler's per-launch reciprocal (a VGPR for the whole kernel):
                        // floor(item * floor(2^32 / spp) / 2^32) is the quotient or one less (item < 2^24, spp <= 2^16)
                        pix = __umulhi(item, rp.spp_magic);
                        uint32_t sample = item - pix * rp.spp_count;
                        if (sample >= rp.spp_count) { pix += 1u; sample -= rp.spp_count; }
                        sample += rp.spp_begin;
                        const int px = tile_x0 + (int)(pix & ((1u << rp.tile_w_log2) - 1u));
                        const int py = tile_y0 + (int)(pix >> rp.tile_w_log2);
                        if (px < sc.width && py < sc.height) {
                            if (STATS) { cnt.l_gen += 1; if (first_active_lane()) cnt.w_gen += 1; }
                            generate(sc, ps, px, py, sample, rp.seed32);
                            trav_begin(sc, tv, ps.o, ps.d, stack_column(s_dyn, tv.top));
                            has_path = true;
                        }   // else: pixel outside the frame (partial edge tile): the item is dropped
                    } else {
                        exhausted = true;
                    }
                }
                next += (uint32_t)__popcll(wmask);
            }
            if (STATS) cnt.cyc[3] += __builtin_readcyclecounter() - t_refill;
            // ---- traverse until enough lanes are waiting again
            unsigned long long tmask = __ballot(trav_active(tv));
            if (tmask == 0ull) {
                if (__ballot(!exhausted) == 0ull) break;   // tile done
                continue;                                  // dropped items / paths that ended at once
            }
            // Majority scheduling: every iteration the wave runs EITHER one inner-node step (for the lanes
            // standing on an inner record) OR one leaf (for the lanes standing on a leaf), whichever has
            // more takers (weighted by sched_beta4 / 4). Classic while-while (descend until every lane has
            // a leaf) makes a 64-wide wave wait for the slowest of 64 descents; beta4 == 0 reproduces it.
            // Lane state is readable from tv.node alone: >= 0 inner record, < -1 leaf, == -1 nothing to
            // traverse (finished, or no path), so each census is one v_cmp + s_bcnt1.
            const uint32_t nP = (uint32_t)__popcll(__ballot(has_path));   // constant during the traversal phase
            // The phase ends when nobody traverses any more or when refill_thresh lanes wait for shading. Both are one
            // test on the number of waiting lanes: nobody traversing <=> all nP lanes wait, so the limit is min(thresh, nP).
            const uint32_t wait_limit = nP < rp.refill_thresh ? nP : rp.refill_thresh;
            // issue priority to the waves that are traversing (memory-bound steps: get their loads out first), over
            // the waves in the compute-heavy shade / generate phase: -1 % (the other way round: +2 %)
            __builtin_amdgcn_s_setprio(3);
            for (;;) {
                // (every instruction of this census is paid once per iteration by the whole wave: 13 instructions as
                // written here, 41 in round 1's form with its separate tests and branches -- profiles/r02_isa_breakdown.md)
#if defined(TRH_PERTURB_SALU)   // development: what does a scalar instruction cost here? (profiles/r02_perturbation.md)
                {
                    uint32_t dummy = nP;
#pragma unroll
                    for (int q = 0; q < TRH_PERTURB_SALU; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy) : : "scc");
                }
#endif
#if defined(TRH_PERTURB_VALU)   // development: ... and a vector instruction?
                {
                    float dummy = ps.o.x;
#pragma unroll
                    for (int q = 0; q < TRH_PERTURB_VALU; ++q) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dummy));
                }
#endif
#if defined(TRH_PERTURB_NOP)    // development: ... and plain issue slots?
#pragma unroll
                for (int q = 0; q < TRH_PERTURB_NOP; ++q) asm volatile("s_nop 0");
#endif
                const uint32_t nI = (uint32_t)__popcll(__ballot(tv.node >= 0));
                const uint32_t nL = (uint32_t)__popcll(__ballot(tv.node < -1));
                if (nP - (nI + nL) >= wait_limit) break;
                const unsigned long long t_step = STATS ? __builtin_readcyclecounter() : 0ull;
                // nI == 0 needs no test of its own: then nL > 0 here, and 0 >= nL * beta4 is false for beta4 >= 1
                if (nI * 4u >= nL * rp.sched_beta4) {
                    if (STATS && fresh_lane_id() == 0u) { cnt.idle_leaf += nL; cnt.idle_wait += nP - nI - nL; cnt.idle_none += 64u - nP; }
#pragma unroll
                    for (int r = 0; r < TRH_INNER_REPEAT; ++r)
                        if (tv.node >= 0) trav_inner<STATS, N16>(sc, ps.o, tv, nullptr, cnt);
                    if (STATS) cnt.cyc[0] += __builtin_readcyclecounter() - t_step;
                } else {
                    if (tv.node < -1) trav_leaf<STATS>(sc, ps.o, ps.d, tv, nullptr, cnt);
                    if (STATS) cnt.cyc[1] += __builtin_readcyclecounter() - t_step;
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        // ---- flush the tile: plain read-modify-write, the tile belongs to this wave
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = fresh_lane_id(); i < P; i += 64) {
            const int px = tile_x0 + (int)(i & ((1u << rp.tile_w_log2) - 1u));
            const int py = tile_y0 + (int)(i >> rp.tile_w_log2);
            if (px < sc.width && py < sc.height) {
                const size_t p = (size_t)py * sc.width + px;
                float4 a = accum[p];
                float r = 0.f, g = 0.f, b = 0.f;
                for (uint32_t k = 0; k < rp.acc_copies; ++k) {   // fixed order: run-to-run identical
                    r += wacc[k * acc_stride + i * 3 + 0];
                    g += wacc[k * acc_stride + i * 3 + 1];
                    b += wacc[k * acc_stride + i * 3 + 2];
                }
                a.x += r; a.y += g; a.z += b; a.w += (float)rp.spp_count;
                accum[p] = a;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (rp.tile_cost && fresh_lane_id() == 0u) {
            const unsigned long long dt = __builtin_readcyclecounter() - t_tile;
            rp.tile_cost[tile] = dt > 0xffffffffull ? 0xffffffffu : (uint32_t)dt;
        }
    }
    if (STATS) cnt.cyc[4] += __builtin_readcyclecounter() - t_kernel;
    flush_counters<STATS>(cnt, counters);
}

// ---------------------------------------------------------------------------------------
// k_megakernel_ab: a wavefront path tracer inside every wave (round 3; default pipeline).
//
// k_megakernel lets every lane run its own path from camera ray to last bounce; lanes that finished a ray wait for the
// divergent shade + refill code until enough of them have gathered (lanes per VALU instruction on the dragon: 0.57,
// 0.46 on the bounce rays alone). Here a wave splits the work of its tile into three phases over a ring of paths it
// keeps in global memory (L2-resident: written and read by this wave only, no atomics, no inter-wave traffic):
//   * camera phase: 64 items at a time (sample-major: camera rays through ONE pixel or two neighbours) -- generate, ONE
//     packet traversal for the whole wave (trav_packet: wave-uniform node index and stack, records and triangles by
//     scalar loads, every lane busy in every instruction), shade, roulette; the paths that go on are appended to the
//     ring (ballot + mbcnt compaction, coalesced float4 stores: origin, direction, throughput, RNG state, bounce | pixel);
//   * traverse phase: lanes take RAYS from the ring (origin + direction: 32 bytes) and run k_megakernel's vote loop
//     (inner steps against triangle steps, LDS stack columns). A lane that finishes stores the hit (16 bytes) next to
//     its path and takes the next ray as soon as refill_thresh lanes are ready to do the same -- no shading in this
//     loop, no path state in its registers. The phase ends when the ready rays are used up and every lane is done;
//   * shade phase: the traversed paths, 64 at a time, every lane busy: attribute interpolation, textures, Scatter, sky,
//     roulette, LDS tile accumulators; survivors go back to the ring as new rays.
// Queue discipline: ring indices shade_head <= head <= tail (mod pq_cap): [shade_head, head) traversed, waiting to be
// shaded; [head, tail) ready to be traversed. The camera phase runs while the tile has items, the ring has 64 free
// entries and fewer than pq_target rays are ready, so that a traverse phase is long against its drain.
// Same sample set, same arithmetic per path as k_megakernel; only the order of a pixel's float additions differs.
// ---------------------------------------------------------------------------------------
template <bool STATS, bool N16>
__global__ __launch_bounds__(TRH_BLOCK) __attribute__((amdgpu_waves_per_eu(TRH_MEGA_WAVES_PER_EU, TRH_MEGA_WAVES_PER_EU)))
void k_megakernel_ab(DevScene sc, RenderParams rp, float4* __restrict__ accum, DevCounters* __restrict__ counters, unsigned int* __restrict__ tile_counter) {
    extern __shared__ int s_dyn[];
    s_dyn[threadIdx.x] = -1;   // sentinel row of this lane's stack column (dev_trace.h)
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t P_log2 = rp.tile_w_log2 + rp.tile_h_log2;
    const uint32_t P = 1u << P_log2;
    const uint32_t acc_stride = P * 3u + 1u;
    float* const wacc = reinterpret_cast<float*>(s_dyn + rp.stack_entries * TRH_BLOCK) + wave_in_block * (acc_stride * rp.acc_copies);   // wave-uniform
    float4* const qa = rp.pq + (size_t)(blockIdx.x * (uint32_t)TRH_WAVES_PER_BLOCK + wave_in_block) * 4u * rp.pq_cap;                   // wave-uniform
    float4* const qb = qa + rp.pq_cap;
    float4* const qc = qb + rp.pq_cap;
    float4* const qh = qc + rp.pq_cap;
    const uint32_t qmask = rp.pq_cap - 1u;
    const uint32_t ntiles = rp.tiles_x * rp.tiles_y;
    const uint32_t total = P * rp.spp_count;   // items per tile
    Counters cnt = {};
    const unsigned long long t_kernel = STATS ? __builtin_readcyclecounter() : 0ull;
    if (STATS) cnt.t_start = wall_clock64();

    for (;;) {
        const uint32_t lane = fresh_lane_id();
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(tile_counter, 1u);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= ntiles) break;  // wave-uniform exit: every wave reaches it
        if (rp.tile_order) tile = __builtin_amdgcn_readfirstlane(rp.tile_order[tile]);   // expensive tiles first (tile_order.h)
        const unsigned long long t_tile = rp.tile_cost ? __builtin_readcyclecounter() : 0ull;
        const int tile_x0 = (int)((tile % rp.tiles_x) << rp.tile_w_log2);
        const int tile_y0 = (int)((tile / rp.tiles_x) << rp.tile_h_log2);
        for (uint32_t i = lane; i < acc_stride * rp.acc_copies; i += 64) wacc[i] = 0.f;
        __builtin_amdgcn_wave_barrier();

        uint32_t next = 0;                                // wave-uniform: next item of this tile without a camera ray
        uint32_t shade_head = 0, head = 0, tail = 0;      // wave-uniform ring cursors (free-running, masked on use)
        PathState ps;
        ps.o = ps.d = ps.throughput = ps.radiance = mk(0.f, 0.f, 0.f);
        ps.rng = 0;
        ps.bounce = 0;
        Hit hit;
        hit.t = 0.f; hit.u = hit.v = 0.f; hit.prim = -1; hit.id = TRH_MISS_ID;
        uint32_t pix = 0;           // pixel-in-tile of the path this lane shades
        bool has_path = false;      // the lane holds a path whose ray is traced: shade it

        for (;;) {
            // ---- shade (camera phase and shade phase end here), then queue the paths that go on
            const unsigned long long t_phase0 = STATS ? __builtin_readcyclecounter() : 0ull;
            if (has_path) {
                const bool cont = shade<STATS>(sc, ps, hit, cnt);
                if ((ps.radiance.x != 0.f) | (ps.radiance.y != 0.f) | (ps.radiance.z != 0.f)) {
                    float* const lane_acc = wacc + ((fresh_lane_id() & (rp.acc_copies - 1u)) * acc_stride + pix * 3u);
                    __hip_atomic_fetch_add(&lane_acc[0], ps.radiance.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&lane_acc[1], ps.radiance.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&lane_acc[2], ps.radiance.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ps.radiance = mk(0.f, 0.f, 0.f);
                has_path = cont;
            }
            {
                const unsigned long long pmask = __builtin_amdgcn_ballot_w64(has_path);
                if (pmask != 0ull) {
                    if (has_path) {
                        const uint32_t idx = (tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(pmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmask, 0u))) & qmask;
                        qa[idx] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
                        qb[idx] = make_float4(ps.d.y, ps.d.z, ps.throughput.x, ps.throughput.y);
                        qc[idx] = make_float4(ps.throughput.z, __uint_as_float(ps.rng), __uint_as_float((ps.bounce << 8) | pix), 0.f);
                        has_path = false;
                    }
                    tail += (uint32_t)__popcll(pmask);
                }
            }
            const unsigned long long t_phase = STATS ? __builtin_readcyclecounter() : 0ull;
            if (STATS) cnt.cyc[2] += t_phase - t_phase0;
            // ---- shade phase: the next 64 traversed paths
            if (shade_head != head) {
                const uint32_t i = shade_head + fresh_lane_id();
                const uint32_t left = head - shade_head;
                if (i - shade_head < left) {
                    const uint32_t idx = i & qmask;
                    const float4 a = qa[idx], b = qb[idx], c = qc[idx], h = qh[idx];
#if defined(TRH_MEASURE_RING_STALL)   // development: how long does a shade batch wait for its ring entries?
                    if (STATS) { const unsigned long long t0 = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (fresh_lane_id() == 0u) cnt.idle_none += (uint32_t)((__builtin_readcyclecounter() - t0) >> 4); }
#endif
                    ps.o = mk(a.x, a.y, a.z);
                    ps.d = mk(a.w, b.x, b.y);
                    ps.throughput = mk(b.z, b.w, c.x);
                    ps.rng = __float_as_uint(c.y);
                    const uint32_t bp = __float_as_uint(c.z);
                    ps.bounce = bp >> 8;
                    pix = bp & 255u;
                    hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.prim = __float_as_int(h.w);
                    has_path = true;
                }
                shade_head += left < 64u ? left : 64u;
                continue;
            }
            // ---- camera phase: 64 camera rays, one packet traversal; the loop's head shades them
            if (next < total && (tail - shade_head) + 64u <= rp.pq_cap && (tail - head) < rp.pq_target) {
                const uint32_t item = next + fresh_lane_id();
                next += 64u;
                bool valid = item < total;
                if (valid) {
                    pix = __umulhi(item, rp.spp_magic);
                    uint32_t sample = item - pix * rp.spp_count;
                    if (sample >= rp.spp_count) { pix += 1u; sample -= rp.spp_count; }
                    sample += rp.spp_begin;
                    const int px = tile_x0 + (int)(pix & ((1u << rp.tile_w_log2) - 1u));
                    const int py = tile_y0 + (int)(pix >> rp.tile_w_log2);
                    valid = px < sc.width && py < sc.height;   // pixel outside the frame (partial edge tile): the item is dropped
                    if (valid) generate(sc, ps, px, py, sample, rp.seed32);
                }
                trav_packet<N16>(sc, ps.o, ps.d, valid, hit);
                has_path = valid;
                if (STATS) { cnt.cyc[3] += __builtin_readcyclecounter() - t_phase; if (fresh_lane_id() == 0u) cnt.w_gen += 1; if (valid) cnt.l_gen += 1; }
                continue;
            }
            if (head == tail) {
                if (next >= total) break;   // nothing ready, nothing to shade, no items: tile done
                continue;                   // (not reached: an empty ring always has room for the camera phase)
            }
            // ---- traverse phase: every ready ray, lanes refilled as they finish
            // the ring stores above are complete before the loads below (same CU, same L1: workgroup scope)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            {
                Trav tv;
                tv.inv = tv.oi = mk(0.f, 0.f, 0.f);
                tv.selx = tv.sely = tv.selz = 0x03020100u;
                tv.node = -1;
                tv.top = s_dyn + (wave_in_block * 64u + fresh_lane_id());
                tv.hit.t = 0.f; tv.hit.u = tv.hit.v = 0.f; tv.hit.prim = -1; tv.hit.id = TRH_MISS_ID;
                V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f);
                uint32_t slot = 0;
                bool has_ray = false;
                __builtin_amdgcn_s_setprio(3);
                for (;;) {
                    // lanes that finished their ray store the hit beside the path and take the next ready ray
                    const bool want = !trav_active(tv);
                    if (want && has_ray) {
                        qh[slot] = make_float4(tv.hit.t, tv.hit.u, tv.hit.v, __int_as_float(tv.hit.prim));
                        has_ray = false;
                    }
                    if (head != tail) {
                        const unsigned long long wmask = __builtin_amdgcn_ballot_w64(want);
                        if (want) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
                            if (rank < tail - head) {
                                slot = (head + rank) & qmask;
                                const float4 a = qa[slot];
                                const float2 b = *reinterpret_cast<const float2*>(&qb[slot]);
#if defined(TRH_MEASURE_RING_STALL)   // development: ... and a refill of the traverse phase for its rays?
                                if (STATS) { const unsigned long long t0 = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (first_active_lane()) cnt.idle_wait += (uint32_t)((__builtin_readcyclecounter() - t0) >> 4); }
#endif
                                ro = mk(a.x, a.y, a.z);
                                rd = mk(a.w, b.x, b.y);
                                trav_begin(sc, tv, ro, rd, stack_column(s_dyn, tv.top));
                                has_ray = true;
                            }
                        }
                        const uint32_t taken = (uint32_t)__popcll(wmask);
                        // (readfirstlane: the cursor is wave-uniform by construction, but updated under a condition on itself the
                        // compiler cannot see through -- left alone it lives in a VGPR and every test of it becomes an exec-mask loop)
                        head = (uint32_t)__builtin_amdgcn_readfirstlane((int)(head + (taken < tail - head ? taken : tail - head)));
                    }
                    const uint32_t nR = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(has_ray));
                    if (nR == 0u) break;   // ready rays used up (a lane without a ray found none), every hit stored
                    // leave the vote loop when refill_thresh lanes have finished -- or all of them, once nothing is left to take
                    const uint32_t wait_limit = (uint32_t)__builtin_amdgcn_readfirstlane((int)((head != tail && rp.refill_thresh < nR) ? rp.refill_thresh : nR));
                    for (;;) {
                        const uint32_t nI = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tv.node >= 0));
                        const uint32_t nL = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tv.node < -1));
                        if (nR - (nI + nL) >= wait_limit) break;
                        const unsigned long long t_step = STATS ? __builtin_readcyclecounter() : 0ull;
                        if (nI * 4u >= nL * rp.sched_beta4) {
                            #if !defined(TRH_MEASURE_RING_STALL)
                            if (STATS && fresh_lane_id() == 0u) { cnt.idle_leaf += nL; cnt.idle_wait += nR - nI - nL; cnt.idle_none += 64u - nR; }
#endif
#pragma unroll
                            for (int r = 0; r < TRH_INNER_REPEAT; ++r)
                                if (tv.node >= 0) trav_inner<STATS, N16>(sc, ro, tv, nullptr, cnt);
                            if (STATS) cnt.cyc[0] += __builtin_readcyclecounter() - t_step;
                        } else {
                            if (tv.node < -1) trav_leaf<STATS>(sc, ro, rd, tv, nullptr, cnt);
                            if (STATS) cnt.cyc[1] += __builtin_readcyclecounter() - t_step;
                        }
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the hits are stored before the shade phase loads them
        }
        // ---- flush the tile: plain read-modify-write, the tile belongs to this wave
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = fresh_lane_id(); i < P; i += 64) {
            const int px = tile_x0 + (int)(i & ((1u << rp.tile_w_log2) - 1u));
            const int py = tile_y0 + (int)(i >> rp.tile_w_log2);
            if (px < sc.width && py < sc.height) {
                const size_t p = (size_t)py * sc.width + px;
                float4 a = accum[p];
                float r = 0.f, g = 0.f, b = 0.f;
                for (uint32_t k = 0; k < rp.acc_copies; ++k) {   // fixed order: run-to-run identical
                    r += wacc[k * acc_stride + i * 3 + 0];
                    g += wacc[k * acc_stride + i * 3 + 1];
                    b += wacc[k * acc_stride + i * 3 + 2];
                }
                a.x += r; a.y += g; a.z += b; a.w += (float)rp.spp_count;
                accum[p] = a;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (rp.tile_cost && fresh_lane_id() == 0u) {
            const unsigned long long dt = __builtin_readcyclecounter() - t_tile;
            rp.tile_cost[tile] = dt > 0xffffffffull ? 0xffffffffu : (uint32_t)dt;
        }
    }
    if (STATS) cnt.cyc[4] += __builtin_readcyclecounter() - t_kernel;
    flush_counters<STATS>(cnt, counters);
}

template <bool N16>
__global__ __launch_bounds__(TRH_BLOCK) void k_primary_hits(DevScene sc, uint32_t* __restrict__ ids, float* __restrict__ t_out, float2* __restrict__ uv_out) {
    __shared__ int s_stack[(TRH_STACK_SIZE + 1) * TRH_BLOCK];
    int* stack = &s_stack[threadIdx.x];
    const uint32_t p = blockIdx.x * TRH_BLOCK + threadIdx.x;
    if (p >= (uint32_t)(sc.width * sc.height)) return;
    const int px = (int)(p % (uint32_t)sc.width), py = (int)(p / (uint32_t)sc.width);
    const float u = ((float)px + 0.5f) / (float)sc.width;
    const float v = ((float)py + 0.5f) / (float)sc.height;
    V3 o, d;
    camera_ray(sc, u, v, o, d);
    Hit hit;
    Counters cnt = {};
    intersect<false, N16>(sc, o, d, hit, stack, cnt);
    ids[p] = hit.id;
    t_out[p] = hit.t;
    uv_out[p] = make_float2(hit.u, hit.v);
}

template <bool N16>
__global__ __launch_bounds__(TRH_BLOCK) void k_sample_radiance(DevScene sc, uint32_t sample, uint32_t seed_hash, float4* __restrict__ out) {
    __shared__ int s_stack[(TRH_STACK_SIZE + 1) * TRH_BLOCK];
    int* stack = &s_stack[threadIdx.x];
    const uint32_t p = blockIdx.x * TRH_BLOCK + threadIdx.x;
    if (p >= (uint32_t)(sc.width * sc.height)) return;
    const int px = (int)(p % (uint32_t)sc.width), py = (int)(p / (uint32_t)sc.width);
    PathState ps;
    generate(sc, ps, px, py, sample, seed_hash);
    Counters cnt = {};
    bool alive = true;
    while (alive) {
        Hit hit;
        intersect<false, N16>(sc, ps.o, ps.d, hit, stack, cnt);
        alive = shade<false>(sc, ps, hit, cnt);
    }
    out[p] = make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z, (float)ps.bounce);
}

__global__ void k_resolve(const float4* __restrict__ accum, float4* __restrict__ out, uint32_t n) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 a = accum[p];
    const float inv = a.w > 0.f ? 1.f / a.w : 0.f;
    out[p] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w);
}

// CPUDetails::Tonemap + Bitmap::SetPixel for one pixel (cpu_details.cpp:218-243, bitmap.cpp:58-69)
DEVFN float srgb_curve(float c) {   // glm convertLinearToSRGB (common.h:152): clamp, then the exact piece-wise curve
    c = fminf(fmaxf(c, 0.f), 1.f);
    return c < 0.0031308f ? c * 12.92f : powf(c, 0.41666f) * 1.055f - 0.055f;
}
DEVFN float tone_curve(float x, int mode) {
    if (mode == TRH_TONEMAP_REINHARD) return fminf(fmaxf(x / (1.f + x), 0.f), 1.f);
    if (mode == TRH_TONEMAP_ACES) return fminf(fmaxf((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.f), 1.f);
    return x;
}
__global__ void k_present(const float4* __restrict__ accum, uint32_t* __restrict__ out, int width, int height, int mode, float exposure) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (uint32_t)(width * height)) return;
    const int x = (int)(p % (uint32_t)width), y = (int)(p / (uint32_t)width);
    const float4 a = accum[p];
    const float inv = a.w > 0.f ? 1.f / a.w : 0.f;
    // the reference's no-tonemap build returns the colour as it is, without kExposure (cpu_details.cpp:236-238)
    const float ex = mode != TRH_TONEMAP_NONE ? exposure : 1.f;
    float c[3] = {a.x * inv * ex, a.y * inv * ex, a.z * inv * ex};
    uint32_t b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = c[k];
        if (mode != TRH_TONEMAP_NONE) v = srgb_curve(tone_curve(v, mode));
        v = fminf(fmaxf(255.99f * v, 0.f), 255.f);
        b[k] = (uint32_t)v;   // (u8) truncation
    }
    out[(size_t)(height - 1 - y) * width + x] = b[2] | (b[1] << 8) | (b[0] << 16);
}

#include "wavefront.h"

// =====================================================================================
// host side of the C ABI
// =====================================================================================

// ---- scratch of trh_render_noalloc (include/tracy_hip.h): the tile counter and the tile-order arrays of ONE launch
// sequence, in a file-scope device array instead of hipMalloc'ed memory. Every GPU the library runs on holds its own
// copy (9.25 MiB) at its own address until the process ends; host code never uses the array's name as an address but
// the per-device pointer trh_init obtains with hipGetSymbolAddress (g_scratch_ptr).
constexpr uint32_t kNoAllocMaxTiles = 1u << 18;                  // 262 144 tiles: 1080p and 4K at >= 8 spp per launch fit
constexpr size_t kNoAllocSortTempBytes = 4u << 20;               // hipcub radix-sort temporary storage for that many keys (separate in/out arrays: a second key and value array + histograms)
constexpr size_t kNoAllocArrayBytes = (size_t)kNoAllocMaxTiles * sizeof(uint32_t);
constexpr size_t kNoAllocScratchBytes = 256 + 5 * kNoAllocArrayBytes + kNoAllocSortTempBytes;
static __device__ __align__(256) unsigned char g_scratch[kNoAllocScratchBytes];

namespace {

constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};   // per device: the address of that GPU's copy of g_scratch
std::mutex g_scratch_mutex;

thread_local std::string g_err;
std::string g_bvh_cache_dir = [] { const char* e = getenv("TRH_BVH_CACHE"); return std::string(e ? e : ""); }();   // trh_set_bvh_cache_dir

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                           \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return fail(TRH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

uint32_t host_pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
uint32_t seed_hash_of(uint64_t seed) {
    const uint32_t seed32 = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u);
    return host_pcg_hash(seed32);
}

}  // namespace

// Scratch of one megakernel launch sequence: the tile counter the persistent waves pull from and the arrays of the
// cost-ordered tile hand-out (tile_order.h). trh_render keeps one per context in hipMalloc'ed memory (the order is
// reused by the next launch of the same tiling); trh_render_noalloc lays one out in the file-scope g_scratch array.
struct TileState {
    unsigned int* tile_counter = nullptr;
    uint32_t *cost = nullptr, *order = nullptr, *keys = nullptr, *keys_sorted = nullptr, *iota = nullptr;
    void* sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    uint32_t count = 0, shape = 0;   // the tiling the arrays belong to
    bool order_valid = false;
};

struct trh_context {
    int builder = TRH_BUILDER_HOST_SAH;
    uint32_t debug_view = 0;
    float view3[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    int pipeline = TRH_PIPELINE_MEGAKERNEL;
    bool has_scene = false;
    bool has_camera = false;
    DevScene sc{};
    std::vector<void*> allocs;  // everything freed at shutdown / re-upload
    float4* accum = nullptr;          // buffer in use (internal or caller-bound)
    float4* accum_internal = nullptr;
    float4* resolved = nullptr;
    uint32_t* bitmap = nullptr;       // BGRA8 picture of trh_present_bgra8
    DevCounters* counters = nullptr;
    // cost-ordered tile scheduling (tile_order.h); the arrays live until the next scene upload
    bool tile_feedback = true;
    TileState ts;
    uint64_t samples = 0;        // since the last stats reset
    uint64_t accum_spp = 0;      // samples per pixel in the accumulation buffer
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // render launches not yet timed
    double kernel_ms = 0.0;
    uint64_t launches = 0;
    trh_bvh_info bvh{};
    int simple_grid[2][2] = {{0, 0}, {0, 0}};
    uint32_t refill_thresh = 40;   // swept 16..48 with the cost-ordered 2048-item tiles
    uint32_t sched_beta4 = 6;
    size_t lds_pad = 0;
    // k_megakernel_ab: the bounce queues of the resident waves (allocated at the first launch, freed with the scene)
    bool stats_ab = false;         // TRH_STATS_AB=1 (development): the instrumented launch runs k_megakernel_ab too (phase times, lane utilisation)
    bool camera_packets = true;    // TRH_CAMERA_PACKETS=0: the round-2 kernel (k_megakernel) for every launch
    float4* pq = nullptr;
    size_t pq_waves = 0;           // wave slots the allocation holds
    uint32_t pq_cap = 512;         // ring entries per wave (a power of two): 32 KB per wave; swept 256..4096, profiles/r03_ring_sweeps.txt
    uint32_t pq_target = 448;      // ready rays that end the camera phase
    uint32_t ab_refill_thresh = 24;   // finished lanes that end a vote loop of the traverse phase (their refill is a 32-byte load)
    // device buffer of the parity probes (trh_primary_hits / trh_sample_radiance), grown on demand, freed at shutdown
    void* probe_buf = nullptr;
    size_t probe_bytes = 0;
    // wavefront pipeline state (allocated on first use)
    WfBuffers wf{};
    std::vector<void*> wf_allocs;
    size_t wf_item_capacity = 0;
    uint32_t wf_pool_limit = 1u << 22;
};

namespace {

template <class T>
int dev_alloc_copy(trh_context* c, const T* host, size_t count, T** out) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    c->allocs.push_back(p);
    if (count) HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (T*)p;
    return TRH_OK;
}

int free_scene(trh_context* c) {
    for (void* p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    for (void* p : c->wf_allocs) (void)hipFree(p);
    c->wf_allocs.clear();
    c->wf = WfBuffers{};
    c->wf_item_capacity = 0;
    c->pq = nullptr;       // (in c->allocs)
    c->pq_waves = 0;
    c->accum = nullptr;
    c->accum_internal = nullptr;
    c->resolved = nullptr;
    c->bitmap = nullptr;
    {
        unsigned int* counter = c->ts.tile_counter;   // allocated by trh_init, lives as long as the context
        c->ts = TileState{};
        c->ts.tile_counter = counter;
    }
    c->has_scene = false;
    return TRH_OK;
}

int ensure_probe_buffer(trh_context* c, size_t bytes) {
    if (c->probe_bytes >= bytes) return TRH_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->probe_buf);
    c->probe_buf = nullptr;
    c->probe_bytes = 0;
    HIP_TRY(hipMalloc(&c->probe_buf, bytes));
    c->probe_bytes = bytes;
    return TRH_OK;
}

int collect_timings(trh_context* c) {
    for (auto& pr : c->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(pr.second));
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        c->kernel_ms += ms;
        c->launches += 1;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    c->pending.clear();
    return TRH_OK;
}

template <bool STATS, bool N16>
int launch_megakernel_simple(trh_context* c, RenderParams rp) {
    int& grid = c->simple_grid[STATS][N16];
    if (grid == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_megakernel_simple<STATS, N16>, TRH_BLOCK, 0));
        if (per_cu < 1) per_cu = 1;
        grid = per_cu * c->num_cus;
    }
    rp.tile_w_log2 = rp.tile_h_log2 = 3;
    rp.tiles_x = (uint32_t)(c->sc.width + 7) / 8;
    rp.tiles_y = (uint32_t)(c->sc.height + 7) / 8;
    HIP_TRY(hipMemsetAsync(c->ts.tile_counter, 0, sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL((k_megakernel_simple<STATS, N16>), dim3(grid), dim3(TRH_BLOCK), 0, c->stream, c->sc, rp, c->accum, c->counters, c->ts.tile_counter);
    HIP_TRY(hipGetLastError());
    return TRH_OK;
}

// (re)allocates the context's tile-order arrays for a tiling of ntiles tiles; a new tiling starts in raster order
int ensure_tile_order(trh_context* c, uint32_t ntiles, uint32_t shape) {
    TileState& ts = c->ts;
    if (ts.count == ntiles && ts.shape == shape && ts.cost) return TRH_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    void** arrays[6] = {(void**)&ts.cost, (void**)&ts.order, (void**)&ts.keys, (void**)&ts.keys_sorted, (void**)&ts.iota, &ts.sort_temp};
    for (void** a : arrays) {
        if (*a) {
            c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), *a), c->allocs.end());
            (void)hipFree(*a);
            *a = nullptr;
        }
    }
    ts.order_valid = false;
    ts.count = 0;
    ts.sort_temp_bytes = tile_order::scratch_bytes(ntiles);
    for (int k = 0; k < 6; ++k) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, k < 5 ? (size_t)ntiles * sizeof(uint32_t) : std::max<size_t>(ts.sort_temp_bytes, 16)));
        c->allocs.push_back(p);
        *arrays[k] = p;
    }
    HIP_TRY(hipMemsetAsync(ts.cost, 0, (size_t)ntiles * sizeof(uint32_t), c->stream));
    ts.count = ntiles;
    ts.shape = shape;
    return TRH_OK;
}

// The tiling of a launch: tile = P pixels x spp items, P = 2x2 ... 16x16 pixels. Two wishes pull on P: a tile should hold
// enough items for the wave-local scheduler to keep its lanes busy (best about 8192 = 128 per lane, no use below 512), and
// the launch should have about 16 tiles per resident wave to balance. Measured on the dragon at 1080p (32400 tiles of 64
// pixels = 7.9 per wave): 64 spp: 8x8 pixels 15.9 ms, 8x4 13.0, 4x4 13.6; 32 spp: 8x8 8.3, 8x4 7.1; 16 spp: 4.3 / 4.0;
// 8 spp: 2.44 / 2.56; 1 spp: 16x16 pixels 1.01 ms, 16x8 0.84, 8x8 0.77.
// Round 2 (5 waves per SIMD): with the balance wish in force (launches of <= 64 spp) nothing changes between 2048 and 16384
// items; big launches like more items per tile, i.e. fewer tile drains: 512 spp per launch 77.8 ms at 2048 items (2x2
// pixels), 76.0 at 4096, 75.6 at 8192 (4x4 pixels), 76.2 at 16384; Cornell 70.0 / 68.9 / 68.3 / 69.1. Hence 8192.
void choose_tiling(const trh_context* c, uint32_t spp_count, uint32_t& wl, uint32_t& hl, uint32_t& ntiles) {
    const double pixels = (double)c->sc.width * (double)c->sc.height;
    const double resident_waves = (double)c->num_cus * 4.0 * TRH_MEGA_WAVES_PER_EU;
    static const double items_wish = [] { const char* e = getenv("TRH_TILE_ITEMS"); return e && atof(e) >= 64.0 ? atof(e) : 8192.0; }();   // development knob
    // more than 32 pixels per tile only for launches of few samples: rays of far-apart pixels share less (helmet 4K, 64 spp:
    // 128-pixel tiles 20.0 ms, 32-pixel tiles 18.1)
    const double p_wish = std::min(std::min(items_wish / (double)spp_count, pixels / (resident_waves * 16.0)), spp_count >= 64u ? 32.0 : 256.0);
    uint32_t p_log2 = 2;
    static const int forced = [] { const char* e = getenv("TRH_TILE_PLOG2"); return e ? atoi(e) : -1; }();   // development knob
    while (p_log2 < 8u && (double)(2u << p_log2) <= p_wish * 1.4142135) ++p_log2;   // nearest power of two
    // few samples: rather 512 items per tile than 16 tiles per wave, down to 4 tiles per wave
    while (p_log2 < 8u && (uint64_t)(1u << p_log2) * spp_count < 512u && pixels / (double)(2u << p_log2) >= resident_waves * 4.0) ++p_log2;
    if (forced >= 2 && forced <= 8) p_log2 = (uint32_t)forced;
    wl = (p_log2 + 1u) / 2u;
    hl = p_log2 / 2u;
    ntiles = (((uint32_t)c->sc.width + (1u << wl) - 1u) >> wl) * (((uint32_t)c->sc.height + (1u << hl) - 1u) >> hl);
}

// one launch of the megakernel with the tiling (wl, hl); `ts` holds the tile counter and, for an ordered launch, the
// arrays of the cost order (already sized for this tiling)
template <bool STATS, bool N16>
int launch_megakernel_tiled(trh_context* c, TileState& ts, RenderParams rp, uint32_t wl, uint32_t hl, bool ordered) {
    rp.tile_w_log2 = wl;
    rp.tile_h_log2 = hl;
    rp.tiles_x = ((uint32_t)c->sc.width + (1u << wl) - 1u) >> wl;
    rp.tiles_y = ((uint32_t)c->sc.height + (1u << hl) - 1u) >> hl;
    rp.stack_entries = std::max<uint32_t>(c->bvh.max_depth, 1u) + 1u;   // + the sentinel row
    rp.refill_thresh = c->refill_thresh;
    rp.sched_beta4 = c->sched_beta4;
    rp.spp_magic = rp.spp_count <= 1u ? 0xffffffffu : (uint32_t)((1ull << 32) / rp.spp_count);
    if (ordered) {
        rp.tile_order = ts.order_valid ? ts.order : nullptr;
        rp.tile_cost = ts.cost;
    }
    // accumulator copies per wave (see the kernel): as many (<= 8, <= 3 KB per wave) as leave four workgroups per CU their LDS
    const size_t stack_bytes = (size_t)rp.stack_entries * TRH_BLOCK * 4;
    auto acc_bytes = [&](uint32_t copies) { return (size_t)TRH_WAVES_PER_BLOCK * ((3u << (wl + hl)) + 1u) * copies * 4; };
    rp.acc_copies = 8;
    while (rp.acc_copies > 1u && (acc_bytes(rp.acc_copies) > 4u * 3200u || (stack_bytes + acc_bytes(rp.acc_copies)) * (unsigned)TRH_MEGA_WAVES_PER_EU > 160u * 1024u)) rp.acc_copies >>= 1;
    size_t lds = stack_bytes + acc_bytes(rp.acc_copies);
    lds += c->lds_pad;   // development knob: lowers occupancy to measure latency sensitivity
    // Camera rays as packets + bounce queue (k_megakernel_ab) for every launch of the context's own tile state; the
    // instrumented kernel counts the per-ray work of the lane scheduler (the algorithmic figure of the roofline), and
    // trh_render_noalloc (a tile state in the static scratch) has no queue memory: both run k_megakernel.
    const bool ab = (!STATS || c->stats_ab) && c->camera_packets && c->pipeline == TRH_PIPELINE_MEGAKERNEL && &ts == &c->ts;
    int per_cu = 0;
    if (ab) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_megakernel_ab<STATS, N16>, TRH_BLOCK, lds));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_megakernel<STATS, N16>, TRH_BLOCK, lds));
    if (per_cu < 1) per_cu = 1;
    const int grid = per_cu * c->num_cus;
    if (ab) {
        const size_t waves = (size_t)grid * TRH_WAVES_PER_BLOCK;
        if (c->pq_waves < waves) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->pq) {
                c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), (void*)c->pq), c->allocs.end());
                (void)hipFree(c->pq);
                c->pq = nullptr;
                c->pq_waves = 0;
            }
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, waves * 4u * c->pq_cap * sizeof(float4)));
            c->allocs.push_back(p);
            c->pq = (float4*)p;
            c->pq_waves = waves;
        }
        rp.pq = c->pq;
        rp.pq_cap = c->pq_cap;
        rp.pq_target = std::min(c->pq_target, c->pq_cap - 64u);
        rp.refill_thresh = c->ab_refill_thresh;
    }
    HIP_TRY(hipMemsetAsync(ts.tile_counter, 0, sizeof(unsigned int), c->stream));
    if (ab) hipLaunchKernelGGL((k_megakernel_ab<STATS, N16>), dim3(grid), dim3(TRH_BLOCK), lds, c->stream, c->sc, rp, c->accum, c->counters, ts.tile_counter);
    else hipLaunchKernelGGL((k_megakernel<STATS, N16>), dim3(grid), dim3(TRH_BLOCK), lds, c->stream, c->sc, rp, c->accum, c->counters, ts.tile_counter);
    HIP_TRY(hipGetLastError());
    if (ordered) {
        const int e = tile_order::sort_by_cost(ts.cost, ts.order, ts.count, ts.keys, ts.keys_sorted, ts.iota, ts.sort_temp, ts.sort_temp_bytes, c->stream);
        if (e != 0) return fail(TRH_ERR_HIP, std::string("tile_order::sort_by_cost: ") + hipGetErrorString((hipError_t)e));
        ts.order_valid = true;
    }
    return TRH_OK;
}

// The launch sequence of one trh_render call on the tile state `ts` (whose arrays fit the tiling (wl, hl) if `ordered`).
// Cost-ordered tiles (tile_order.h): a launch records what every tile cost, the next launch of the same tiling hands the
// tiles out most expensive first. The first launch of a tiling has no costs yet: its first sample is traced as a launch of
// its own, in raster order, and its tile costs order the rest (dragon, 64 spp on a fresh context: 14.5 ms instead of 21).
template <bool STATS, bool N16>
int launch_megakernel_on(trh_context* c, TileState& ts, RenderParams rp, uint32_t wl, uint32_t hl, bool ordered) {
    if (ordered && !ts.order_valid && rp.spp_count >= 16) {
        RenderParams probe = rp;
        probe.spp_count = 1;
        if (int prc = launch_megakernel_tiled<STATS, N16>(c, ts, probe, wl, hl, true)) return prc;
        rp.spp_begin += 1;
        rp.spp_count -= 1;
    }
    return launch_megakernel_tiled<STATS, N16>(c, ts, rp, wl, hl, ordered);
}

template <bool STATS, bool N16>
int launch_megakernel(trh_context* c, RenderParams rp) {
    uint32_t wl, hl, ntiles;
    choose_tiling(c, rp.spp_count, wl, hl, ntiles);
    // Launches of a few samples keep raster order: the re-sort would cost a noticeable part of a 1 ms launch.
    const bool ordered = c->tile_feedback && rp.spp_count >= 8;
    if (ordered)
        if (int orc = ensure_tile_order(c, ntiles, (wl << 8) | hl)) return orc;   // a new tiling starts without an order
    return launch_megakernel_on<STATS, N16>(c, c->ts, rp, wl, hl, ordered);
}

// ---------------------------------------------------------------------------------------
// wavefront pipeline: host orchestration (kernels in wavefront.h)
// ---------------------------------------------------------------------------------------
int wf_ensure(trh_context* c, size_t items) {
    const uint32_t npix = (uint32_t)(c->sc.width * c->sc.height);
    if (c->wf.pool == 0) {
        uint32_t pool = c->wf_pool_limit;
        const uint64_t want = ((uint64_t)npix * 4 + 255) / 256 * 256;   // small frames: a few samples per pixel in flight
        if (want < pool) pool = (uint32_t)want;
        if (pool < 1024) pool = 1024;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, (size_t)pool * 5 * sizeof(float4)));
        c->wf_allocs.push_back(p);
        c->wf.st = (float4*)p;
        uint32_t** qs[3] = {&c->wf.q_a, &c->wf.q_b, &c->wf.q_free};
        for (auto q : qs) {
            HIP_TRY(hipMalloc(&p, (size_t)pool * sizeof(uint32_t)));
            c->wf_allocs.push_back(p);
            *q = (uint32_t*)p;
        }
        HIP_TRY(hipMalloc(&p, 256));
        c->wf_allocs.push_back(p);
        c->wf.ctl = (WfCtl*)p;
        c->wf.pool = pool;
    }
    if (items > c->wf_item_capacity) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->wf.item_rad) {
            (void)hipFree(c->wf.item_rad);
            c->wf_allocs.erase(std::remove(c->wf_allocs.begin(), c->wf_allocs.end(), (void*)c->wf.item_rad), c->wf_allocs.end());
            c->wf.item_rad = nullptr;
        }
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, items * sizeof(float4)));
        c->wf_allocs.push_back(p);
        c->wf.item_rad = (float4*)p;
        c->wf_item_capacity = items;
    }
    return TRH_OK;
}

template <bool N16>
int launch_wavefront(trh_context* c, const RenderParams& rp) {
    const uint32_t npix = (uint32_t)(c->sc.width * c->sc.height);
    const uint64_t kItemBudget = 1ull << 28;   // 4 GiB of float4 per batch at most
    uint32_t batch_spp = (uint32_t)std::min<uint64_t>(rp.spp_count, std::max<uint64_t>(1, kItemBudget / npix));
    int rc = wf_ensure(c, (size_t)npix * batch_spp);
    if (rc) return rc;
    const uint32_t pool = c->wf.pool;
    const uint32_t stack_entries = std::max<uint32_t>(c->bvh.max_depth, 1u) + 1u;   // + the sentinel row
    const size_t lds = (size_t)stack_entries * TRH_BLOCK * 4;
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_extend<N16>, TRH_BLOCK, lds));
    if (per_cu < 1) per_cu = 1;
    const int extend_grid = per_cu * c->num_cus;
    const dim3 pool_grid((pool + 255) / 256);
    for (uint32_t done = 0; done < rp.spp_count; done += batch_spp) {
        const uint32_t spp = std::min(batch_spp, rp.spp_count - done);
        const uint32_t items = npix * spp;
        hipLaunchKernelGGL(k_wf_init_free, pool_grid, dim3(256), 0, c->stream, c->wf.q_free, pool, c->wf.ctl, items);
        uint32_t* q_cur = c->wf.q_a;
        uint32_t* q_next = c->wf.q_b;
        const uint32_t min_iters = items / pool;   // the item cursor cannot reach the end earlier
        // The host only enqueues: the loop's end is a device-side condition (nothing queued, every item handed out) that the
        // host looks at every kWfPoll iterations -- one small copy + one synchronisation per poll instead of one per
        // iteration (round 2: the per-iteration round trip was part of the pipeline's time). Iterations past the end are
        // launches of kernels that find their queues empty (a few microseconds each).
        constexpr uint32_t kWfPoll = 8;
        for (uint32_t iter = 0;; ++iter) {
            hipLaunchKernelGGL(k_wf_generate, pool_grid, dim3(256), 0, c->stream, c->sc, c->wf, q_next, rp.spp_begin + done, rp.seed32);
            hipLaunchKernelGGL(k_wf_advance, dim3(1), dim3(1), 0, c->stream, c->wf);
            std::swap(q_cur, q_next);
            hipLaunchKernelGGL(k_wf_extend<N16>, dim3(extend_grid), dim3(TRH_BLOCK), lds, c->stream, c->sc, c->wf, q_cur, stack_entries,
                               c->refill_thresh, c->sched_beta4);
            hipLaunchKernelGGL(k_wf_shade, dim3((pool + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK), dim3(WF_SHADE_BLOCK), 0, c->stream, c->sc, c->wf, q_cur, q_next, c->counters);
            HIP_TRY(hipGetLastError());
            if (iter >= min_iters && (iter - min_iters) % kWfPoll == kWfPoll - 1u) {
                WfCtl h{};
                HIP_TRY(hipMemcpyAsync(&h, c->wf.ctl, sizeof(h), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (h.n_next == 0 && h.item_cursor >= h.items_end) break;
            }
            if (iter > 1000000u) return fail(TRH_ERR_STATE, "wavefront pipeline did not terminate");
        }
        hipLaunchKernelGGL(k_wf_accumulate, dim3((npix + 255) / 256), dim3(256), 0, c->stream, c->wf.item_rad, c->accum, npix, spp);
        HIP_TRY(hipGetLastError());
    }
    return TRH_OK;
}

// validate a scene description and flatten its triangles (world space; the reference bakes transforms at load)
int flatten_triangles(const trh_scene_desc* d, std::vector<bvh::InputTriangle>& tris, bool& any_normal_map) {
    if (d->mesh_count > 256) return fail(TRH_ERR_LIMIT, "trh_scene_upload: more than 256 meshes (8-bit mesh id)");
    if (d->material_count == 0) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: material 0 (sky) is required");
    tris.reserve(d->index_count / 3);
    any_normal_map = false;
    for (uint32_t m = 0; m < d->material_count; ++m) {
        for (int k = 0; k < TRH_TEX_COUNT; ++k) {
            const uint32_t t = d->materials[m].tex[k];
            if (t != TRH_NO_TEXTURE && t >= d->texture_count) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: material references a missing texture");
        }
        if (d->materials[m].tex[TRH_TEX_NORMAL] != TRH_NO_TEXTURE) any_normal_map = true;
    }
    for (uint32_t m = 0; m < d->mesh_count; ++m) {
        const trh_mesh& me = d->meshes[m];
        if ((uint64_t)me.first_vertex + me.vertex_count > d->vertex_count || (uint64_t)me.first_index + me.index_count > d->index_count)
            return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: mesh range out of bounds");
        if (me.material >= d->material_count) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: mesh material out of range");
        if (me.index_count / 3 > 0xffffffu) return fail(TRH_ERR_LIMIT, "trh_scene_upload: more than 2^24 triangles in one mesh");
        for (uint32_t t = 0; t < me.index_count / 3; ++t) {
            bvh::InputTriangle it;
            for (int k = 0; k < 3; ++k) {
                const uint32_t li = d->indices[me.first_index + t * 3 + k];
                if (li >= me.vertex_count) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: vertex index out of range");
                const float* v = d->vertices + (size_t)(me.first_vertex + li) * TRH_VERTEX_FLOATS;
                float* dst = k == 0 ? it.v0 : (k == 1 ? it.v1 : it.v2);
                // NaN / inf positions would make the builders' binning and partition comparators undefined
                if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2]))
                    return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: vertex position is not finite");
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
            }
            it.packed_id = (m << 24) | t;
            tris.push_back(it);
        }
    }
    if (tris.size() >= (1u << 27)) return fail(TRH_ERR_LIMIT, "trh_scene_upload: more than 2^27 triangles");

    return TRH_OK;
}

int render_impl(trh_context* c, uint32_t spp_begin, uint32_t spp_count, uint64_t seed, bool stats) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_render: ctx is NULL");
    if (!c->has_scene || !c->has_camera) return fail(TRH_ERR_STATE, "trh_render: scene and camera must be uploaded first");
    if (spp_count == 0) return TRH_OK;
    if ((uint64_t)spp_begin + spp_count > 0xffffffffull) return fail(TRH_ERR_LIMIT, "trh_render: sample index exceeds 32 bits");
    HIP_TRY(hipSetDevice(c->device));
    RenderParams rp{};
    rp.spp_begin = spp_begin;
    rp.spp_count = spp_count;
    rp.seed32 = seed_hash_of(seed);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct EventGuard {   // the two events are destroyed on every early return; `pending` takes them over at the end
        hipEvent_t &a, &b;
        ~EventGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } event_guard{e0, e1};
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, c->stream));
    int rc = TRH_OK;
    const bool n16 = c->sc.nodes16 != nullptr;   // node record format of this scene (chosen at upload)
    // one launch per 65536 samples per pixel at most: a tile's item count (pixels x samples) stays far inside 32 bits
    constexpr uint32_t kMaxSppPerLaunch = 1u << 16;
    for (uint32_t done = 0; done < spp_count && rc == TRH_OK; done += kMaxSppPerLaunch) {
        rp.spp_begin = spp_begin + done;
        rp.spp_count = std::min(kMaxSppPerLaunch, spp_count - done);
        if (c->pipeline == TRH_PIPELINE_WAVEFRONT && !stats) rc = n16 ? launch_wavefront<true>(c, rp) : launch_wavefront<false>(c, rp);
        else if (c->pipeline == TRH_PIPELINE_MEGAKERNEL_SIMPLE)
            rc = stats ? (n16 ? launch_megakernel_simple<true, true>(c, rp) : launch_megakernel_simple<true, false>(c, rp))
                       : (n16 ? launch_megakernel_simple<false, true>(c, rp) : launch_megakernel_simple<false, false>(c, rp));
        else
            rc = stats ? (n16 ? launch_megakernel<true, true>(c, rp) : launch_megakernel<true, false>(c, rp))
                       : (n16 ? launch_megakernel<false, true>(c, rp) : launch_megakernel<false, false>(c, rp));
    }
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e1, c->stream));
    c->pending.emplace_back(e0, e1);
    e0 = e1 = nullptr;
    c->samples += (uint64_t)c->sc.width * c->sc.height * spp_count;
    c->accum_spp += spp_count;
    if (c->pending.size() > 4096) return collect_timings(c);
    return TRH_OK;
}

// releases everything a context owns (also the partly built context of a failing trh_init)
void destroy_context(trh_context* c) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)collect_timings(c);
    free_scene(c);
    (void)hipFree(c->probe_buf);
    (void)hipFree(c->counters);
    (void)hipFree(c->ts.tile_counter);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // namespace

extern "C" {

const char* trh_last_error(void) { return g_err.c_str(); }
int trh_abi_version(void) { return TRH_ABI_VERSION; }

int trh_device_count(int* count_out) {
    if (!count_out) return fail(TRH_ERR_INVALID_ARG, "trh_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count_out = 0; return fail(TRH_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count_out = n;
    return TRH_OK;
}

int trh_init(int device, trh_context** ctx_out) {
    if (!ctx_out) return fail(TRH_ERR_INVALID_ARG, "trh_init: ctx_out is NULL");
    *ctx_out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(TRH_ERR_NO_DEVICE, "trh_init: no HIP device (this module has no CPU fallback)");
    if (device < 0 || device >= n) return fail(TRH_ERR_INVALID_ARG, "trh_init: device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    auto* c = new trh_context();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { destroy_context(c); return fail(TRH_ERR_HIP, "hipStreamCreate failed"); }
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(DevCounters)) != hipSuccess) { destroy_context(c); return fail(TRH_ERR_HIP, "hipMalloc(counters) failed"); }
    c->counters = (DevCounters*)p;
    (void)hipMemset(p, 0, sizeof(DevCounters));
    if (hipMalloc(&p, 256) != hipSuccess) { destroy_context(c); return fail(TRH_ERR_HIP, "hipMalloc(tile counter) failed"); }
    c->ts.tile_counter = (unsigned int*)p;
    (void)hipMemset(p, 0, 256);
    // trh_render_noalloc: the address of THIS device's copy of the file-scope scratch array, looked up once per device
    // with the device current (every GPU holds its own copy at its own address)
    if (device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            void* sp = nullptr;
            const hipError_t se = hipGetSymbolAddress(&sp, HIP_SYMBOL(g_scratch));
            if (se != hipSuccess || !sp) { destroy_context(c); return fail(TRH_ERR_HIP, std::string("hipGetSymbolAddress(g_scratch): ") + hipGetErrorString(se)); }
            g_scratch_ptr[device] = (unsigned char*)sp;
        }
    }
    if (const char* e = getenv("TRH_REFILL_THRESH")) {   // development knob
        const int v = atoi(e);
        if (v >= 1 && v <= 64) c->refill_thresh = (uint32_t)v;
    }
    if (const char* e = getenv("TRH_WF_POOL_LOG2")) {      // development knob
        const int v = atoi(e);
        if (v >= 8 && v <= 26) c->wf_pool_limit = 1u << v;
    }
    if (const char* e = getenv("TRH_BUILDER")) { if (std::string(e) == "lbvh") c->builder = TRH_BUILDER_DEVICE_LBVH; }   // default of new contexts
    if (const char* e = getenv("TRH_LDS_PAD")) c->lds_pad = (size_t)atoi(e);
    if (const char* e = getenv("TRH_STATS_AB")) c->stats_ab = atoi(e) != 0;
    if (const char* e = getenv("TRH_CAMERA_PACKETS")) c->camera_packets = atoi(e) != 0;   // development knob: 0 = k_megakernel for every launch
    if (const char* e = getenv("TRH_PQ_CAP")) {             // development knob: ring entries per wave (rounded up to a power of two)
        const int v = atoi(e);
        if (v >= 128 && v <= (1 << 20)) { uint32_t p2 = 128; while (p2 < (uint32_t)v) p2 <<= 1; c->pq_cap = p2; }
    }
    if (const char* e = getenv("TRH_PQ_TARGET")) {          // development knob
        const int v = atoi(e);
        if (v >= 1) c->pq_target = (uint32_t)v;
    }
    if (const char* e = getenv("TRH_AB_REFILL")) {          // development knob
        const int v = atoi(e);
        if (v >= 1 && v <= 64) c->ab_refill_thresh = (uint32_t)v;
    }
    if (const char* e = getenv("TRH_SCHED_BETA4")) {     // development knob
        const int v = atoi(e);
        if (v >= 1 && v <= 256) c->sched_beta4 = (uint32_t)v;   // >= 1: the vote relies on it (0 was round 1's while-while mode)
    }
    *ctx_out = c;
    return TRH_OK;
}

int trh_shutdown(trh_context* c) {
    if (!c) return TRH_OK;
    destroy_context(c);
    return TRH_OK;
}

}  // extern "C"

namespace {

// the device arrays of a linear-BVH build that no context has adopted (yet) are released on every return path
struct DeviceBvhGuard {
    lbvh::DeviceBvh& d;
    ~DeviceBvhGuard() { lbvh::release(d); }
};

// trh_scene_upload proper. `shared` (may be NULL): a host-built acceleration structure of this very scene that the
// caller built once for several devices (trh_group_scene_upload); used instead of building when the context's builder
// is the host builder.
int scene_upload_impl(trh_context* c, const trh_scene_desc* d, const bvh::Result* shared, bool shared_from_cache, double shared_build_ms) {
    if (!c || !d) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: NULL argument");
    if (d->width == 0 || d->height == 0) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: empty frame");
    if (d->mesh_count > 256) return fail(TRH_ERR_LIMIT, "trh_scene_upload: more than 256 meshes (8-bit mesh id)");
    if (d->material_count == 0) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: material 0 (sky) is required");
    if ((uint64_t)d->width * d->height > (1ull << 31)) return fail(TRH_ERR_LIMIT, "trh_scene_upload: frame too large");
    if (d->max_bounces >= (1u << 24)) return fail(TRH_ERR_LIMIT, "trh_scene_upload: max_bounces must be below 2^24 (a path's bounce count shares a word with its pixel in the ring)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_scene(c);

    std::vector<bvh::InputTriangle> tris;
    bool any_normal_map = false;
    if (int frc = flatten_triangles(d, tris, any_normal_map)) return frc;

    // ---- build the BVH: on the host (binned SAH, default) or on the device (linear BVH)
    const auto t0 = std::chrono::steady_clock::now();
    bvh::Result local;
    lbvh::DeviceBvh dev_bvh;
    DeviceBvhGuard dev_bvh_guard{dev_bvh};
    float *dev_nodes = nullptr, *dev_tris = nullptr;   // record arrays of a device-side build, adopted by the context
    uint32_t* dev_nodes16 = nullptr;
    bool built_on_device = false;
    if (c->builder == TRH_BUILDER_DEVICE_LBVH && tris.size() >= lbvh::kMinTriangles) {
        std::string why;
        const int brc = lbvh::build(tris, c->stream, dev_bvh, why);
        if (brc == 0) {
            built_on_device = true;
            // the arrays stay in HBM; the host needs the leaf order (shading records) and the scalars
            bvh::Result& r = local;
            r.prim_index.resize(tris.size());
            HIP_TRY(hipMemcpyAsync(r.prim_index.data(), dev_bvh.prim_index, tris.size() * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            (void)hipFree(dev_bvh.prim_index);
            dev_bvh.prim_index = nullptr;
            // from here on the context owns the three record arrays (free_scene releases them)
            c->allocs.push_back(dev_bvh.nodes);
            c->allocs.push_back(dev_bvh.nodes16);
            c->allocs.push_back(dev_bvh.tris);
            float* const adopted_nodes = dev_bvh.nodes;
            uint32_t* const adopted_nodes16 = dev_bvh.nodes16;
            float* const adopted_tris = dev_bvh.tris;
            memcpy(r.frame_center, dev_bvh.frame_center, sizeof(r.frame_center));
            r.frame_scale = dev_bvh.frame_scale;
            r.fp16_inflation = dev_bvh.fp16_inflation;
            r.max_depth = dev_bvh.max_depth;
            r.max_leaf = dev_bvh.max_leaf;
            r.sah_cost = dev_bvh.sah_cost;
            dev_bvh.nodes = nullptr; dev_bvh.nodes16 = nullptr; dev_bvh.tris = nullptr;   // nothing left for the guard
            dev_nodes = adopted_nodes; dev_nodes16 = adopted_nodes16; dev_tris = adopted_tris;
        } else if (brc != 2) {   // 2 = deeper than the traversal stack: the host builder bounds its depth
            return fail(TRH_ERR_HIP, why.c_str());
        } else if (getenv("TRH_VERBOSE")) {
            fprintf(stderr, "trh_scene_upload: %s; using the host builder\n", why.c_str());
        }
    }
    bool from_cache = false;
    const bool use_shared = !built_on_device && shared != nullptr && shared->prim_index.size() == tris.size();
    if (!built_on_device && !use_shared) bvh::build_cached(tris, local, g_bvh_cache_dir, &from_cache);
    if (use_shared) from_cache = shared_from_cache;
    const bvh::Result& r = use_shared ? *shared : local;
    const auto t1 = std::chrono::steady_clock::now();
    if (r.max_depth > (uint32_t)TRH_STACK_SIZE) return fail(TRH_ERR_LIMIT, "trh_scene_upload: BVH deeper than the traversal stack");

    // ---- per-primitive shading records in leaf order
    const size_t nprim = r.prim_index.size();
    std::vector<float> shade(nprim * 16), tang(any_normal_map ? nprim * 12 : 0);
    for (size_t p = 0; p < nprim; ++p) {
        const uint32_t pid = tris[r.prim_index[p]].packed_id;
        const uint32_t m = pid >> 24, t = pid & 0xffffffu;
        const trh_mesh& me = d->meshes[m];
        const float* vv[3];
        for (int k = 0; k < 3; ++k) vv[k] = d->vertices + (size_t)(me.first_vertex + d->indices[me.first_index + t * 3 + k]) * TRH_VERTEX_FLOATS;
        float* s = &shade[p * 16];
        for (int k = 0; k < 3; ++k) { s[k * 3 + 0] = vv[k][3]; s[k * 3 + 1] = vv[k][4]; s[k * 3 + 2] = vv[k][5]; }
        for (int k = 0; k < 3; ++k) { s[9 + k * 2] = vv[k][6]; s[9 + k * 2 + 1] = vv[k][7]; }
        memcpy(&s[15], &me.material, 4);
        if (any_normal_map) {
            float* tg = &tang[p * 12];
            for (int k = 0; k < 3; ++k) { tg[k * 3 + 0] = vv[k][8]; tg[k * 3 + 1] = vv[k][9]; tg[k * 3 + 2] = vv[k][10]; }
            tg[9] = tg[10] = tg[11] = 0.f;
        }
    }

    // ---- upload
    DevScene sc{};
    float* dp = nullptr;
    int rc;
    // node record format: 32-byte fp16 records unless the quantisation would inflate the boxes too much
    bool use_fp16 = r.fp16_inflation <= bvh::kMaxFp16Inflation;
    if (const char* e = getenv("TRH_NODE_FORMAT")) {   // development / test knob: 16 or 32
        if (atoi(e) == 16) use_fp16 = true;
        if (atoi(e) == 32) use_fp16 = false;
    }
    auto drop = [c](void* p) {   // the record format that is not used
        c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), p), c->allocs.end());
        (void)hipFree(p);
    };
    if (use_fp16) {
        if (built_on_device) {
            sc.nodes16 = (const uint4*)dev_nodes16;
            drop(dev_nodes);
        } else {
            uint32_t* dq = nullptr;
            if ((rc = dev_alloc_copy(c, r.nodes16.data(), r.nodes16.size(), &dq))) return rc;
            sc.nodes16 = (const uint4*)dq;
        }
        memcpy(sc.frame_c, r.frame_center, 12);
        sc.frame_sinv = 1.f / r.frame_scale;
        sc.frame_inv_lim = r.frame_scale < 1.f ? 3.0e38f * r.frame_scale : 3.402823466e+38f;
    } else {
        if (built_on_device) {
            sc.nodes = (const float4*)dev_nodes;
            drop(dev_nodes16);
        } else {
            if ((rc = dev_alloc_copy(c, r.nodes.data(), r.nodes.size(), &dp))) return rc;
            sc.nodes = (const float4*)dp;
        }
        sc.frame_c[0] = sc.frame_c[1] = sc.frame_c[2] = 0.f;
        sc.frame_sinv = 1.f;
        sc.frame_inv_lim = 3.402823466e+38f;
    }
    if (built_on_device) {
        sc.tris = (const float4*)dev_tris;
    } else {
        if ((rc = dev_alloc_copy(c, r.tris.data(), r.tris.size(), &dp))) return rc;
        sc.tris = (const float4*)dp;
    }
    if ((rc = dev_alloc_copy(c, shade.data(), shade.size(), &dp))) return rc;
    sc.shade = (const float4*)dp;
    if (any_normal_map) {
        if ((rc = dev_alloc_copy(c, tang.data(), tang.size(), &dp))) return rc;
        sc.tangents = (const float4*)dp;
    }
    std::vector<float> maabb((size_t)d->mesh_count * 8, 0.f);
    for (uint32_t m = 0; m < d->mesh_count; ++m) {
        memcpy(&maabb[m * 8 + 0], d->meshes[m].aabb_min, 12);
        memcpy(&maabb[m * 8 + 4], d->meshes[m].aabb_max, 12);
    }
    if ((rc = dev_alloc_copy(c, maabb.data(), maabb.size(), &dp))) return rc;
    sc.mesh_aabb = (const float4*)dp;
    std::vector<DevMaterial> mats(d->material_count);
    for (uint32_t m = 0; m < d->material_count; ++m) {
        const trh_material& s = d->materials[m];
        DevMaterial& o = mats[m];
        memcpy(o.albedo, s.albedo, 12);
        o.roughness = s.roughness; o.metalness = s.metalness; o.ior = s.ior;
        memcpy(o.emissive, s.emissive, 12);
        o.translucent = s.translucent;
        memcpy(o.tex, s.tex, sizeof(o.tex));
        o.pad = 0;
    }
    DevMaterial* dm = nullptr;
    if ((rc = dev_alloc_copy(c, mats.data(), mats.size(), &dm))) return rc;
    sc.materials = dm;
    std::vector<DevTexture> texs(d->texture_count);
    for (uint32_t t = 0; t < d->texture_count; ++t) {
        const trh_texture& s = d->textures[t];
        if (!s.rgba || !s.width || !s.height) return fail(TRH_ERR_INVALID_ARG, "trh_scene_upload: empty texture");
        float* px = nullptr;
        if ((rc = dev_alloc_copy(c, s.rgba, (size_t)s.width * s.height * 4, &px))) return rc;
        texs[t].texels = (const float4*)px;
        texs[t].width = (int32_t)s.width;
        texs[t].height = (int32_t)s.height;
    }
    DevTexture* dt = nullptr;
    if ((rc = dev_alloc_copy(c, texs.data(), texs.size(), &dt))) return rc;
    sc.textures = dt;
    sc.width = (int32_t)d->width;
    sc.height = (int32_t)d->height;
    sc.width_f = (float)sc.width;
    sc.height_f = (float)sc.height;
    sc.max_bounces = d->max_bounces ? d->max_bounces : 1;
    sc.flags = d->flags;
    // keep the camera if one was set before (re-upload of a scene)
    memcpy(sc.eye, c->sc.eye, sizeof(sc.eye));
    memcpy(sc.ivp, c->sc.ivp, sizeof(sc.ivp));
    memcpy(sc.cam_k, c->sc.cam_k, sizeof(sc.cam_k));
    sc.debug_view = c->debug_view;
    memcpy(sc.view3, c->view3, sizeof(sc.view3));

    const size_t npix = (size_t)d->width * d->height;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, npix * sizeof(float4)));
    c->allocs.push_back(p);
    c->accum = c->accum_internal = (float4*)p;
    HIP_TRY(hipMemset(p, 0, npix * sizeof(float4)));
    HIP_TRY(hipMalloc(&p, npix * sizeof(float4)));
    c->allocs.push_back(p);
    c->resolved = (float4*)p;
    HIP_TRY(hipMemset(p, 0, npix * sizeof(float4)));
    HIP_TRY(hipMalloc(&p, npix * sizeof(uint32_t)));
    c->allocs.push_back(p);
    c->bitmap = (uint32_t*)p;
    HIP_TRY(hipMemset(p, 0, npix * sizeof(uint32_t)));

    c->sc = sc;
    c->has_scene = true;
    c->samples = 0;
    c->accum_spp = 0;
    c->bvh.node_count = built_on_device ? dev_bvh.node_count : (uint32_t)(r.nodes.size() / 16);
    c->bvh.tri_count = (uint32_t)nprim;
    c->bvh.max_depth = r.max_depth;
    c->bvh.max_leaf_size = r.max_leaf;
    c->bvh.node_bytes = (uint64_t)c->bvh.node_count * (use_fp16 ? 32u : 64u);   // the records the kernels traverse
    c->bvh.fp16_inflation = r.fp16_inflation;
    c->bvh.tri_bytes = (uint64_t)nprim * 48u;
    c->bvh.builder = built_on_device ? TRH_BUILDER_DEVICE_LBVH : TRH_BUILDER_HOST_SAH;
    c->bvh.from_cache = from_cache ? 1u : 0u;
    c->bvh.device_build_ms = built_on_device ? dev_bvh.device_ms : 0.0;
    c->bvh.sah_cost = r.sah_cost;
    c->bvh.build_ms = use_shared ? shared_build_ms : std::chrono::duration<double, std::milli>(t1 - t0).count();
    return TRH_OK;
}

}  // namespace

extern "C" {

int trh_scene_upload(trh_context* c, const trh_scene_desc* d) { return scene_upload_impl(c, d, nullptr, false, 0.0); }

int trh_bvh_build_host(const trh_scene_desc* d, trh_bvh_info* info_out, uint64_t* hash_out, uint64_t* violations_out) {
    if (!d || !violations_out) return fail(TRH_ERR_INVALID_ARG, "trh_bvh_build_host: NULL argument");
    std::vector<bvh::InputTriangle> tris;
    bool any_normal_map = false;
    if (int frc = flatten_triangles(d, tris, any_normal_map)) return frc;
    const auto t0 = std::chrono::steady_clock::now();
    bvh::Result r;
    bool from_cache = false;
    bvh::build_cached(tris, r, g_bvh_cache_dir, &from_cache);
    const auto t1 = std::chrono::steady_clock::now();
    std::string why;
    *violations_out = bvh::validate(tris, r, why, hash_out);
    if (*violations_out) g_err = "trh_bvh_build_host: " + why;
    if (info_out) {
        const bool use_fp16 = r.fp16_inflation <= bvh::kMaxFp16Inflation;
        info_out->node_count = (uint32_t)(r.nodes.size() / 16);
        info_out->tri_count = (uint32_t)r.prim_index.size();
        info_out->max_depth = r.max_depth;
        info_out->max_leaf_size = r.max_leaf;
        info_out->node_bytes = use_fp16 ? r.nodes16.size() * 4 : r.nodes.size() * 4;
        info_out->tri_bytes = r.tris.size() * 4;
        info_out->sah_cost = r.sah_cost;
        info_out->build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        info_out->fp16_inflation = r.fp16_inflation;
        info_out->builder = TRH_BUILDER_HOST_SAH;
        info_out->from_cache = from_cache ? 1u : 0u;
        info_out->device_build_ms = 0.0;
    }
    return TRH_OK;
}

int trh_bvh_build_device(trh_context* c, const trh_scene_desc* d, trh_bvh_info* info_out, uint64_t* hash_out, uint64_t* violations_out) {
    if (!c || !d || !violations_out) return fail(TRH_ERR_INVALID_ARG, "trh_bvh_build_device: NULL argument");
    std::vector<bvh::InputTriangle> tris;
    bool any_normal_map = false;
    if (int frc = flatten_triangles(d, tris, any_normal_map)) return frc;
    if (tris.size() < lbvh::kMinTriangles) return fail(TRH_ERR_LIMIT, "trh_bvh_build_device: fewer than 9 triangles (the host builder handles these)");
    HIP_TRY(hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    lbvh::DeviceBvh dev_bvh;
    std::string why;
    const int brc = lbvh::build(tris, c->stream, dev_bvh, why);
    const auto t1 = std::chrono::steady_clock::now();
    if (brc != 0) return fail(brc == 2 ? TRH_ERR_LIMIT : TRH_ERR_HIP, why.c_str());
    bvh::Result r;
    const int drc = lbvh::download(dev_bvh, c->stream, r, why);
    const double device_ms = dev_bvh.device_ms;
    lbvh::release(dev_bvh);
    if (drc != 0) return fail(TRH_ERR_HIP, why.c_str());
    *violations_out = bvh::validate(tris, r, why, hash_out);
    if (*violations_out) g_err = "trh_bvh_build_device: " + why;
    if (info_out) {
        const bool use_fp16 = r.fp16_inflation <= bvh::kMaxFp16Inflation;
        info_out->node_count = (uint32_t)(r.nodes.size() / 16);
        info_out->tri_count = (uint32_t)r.prim_index.size();
        info_out->max_depth = r.max_depth;
        info_out->max_leaf_size = r.max_leaf;
        info_out->node_bytes = use_fp16 ? r.nodes16.size() * 4 : r.nodes.size() * 4;
        info_out->tri_bytes = r.tris.size() * 4;
        info_out->sah_cost = r.sah_cost;
        info_out->build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        info_out->fp16_inflation = r.fp16_inflation;
        info_out->builder = TRH_BUILDER_DEVICE_LBVH;
        info_out->from_cache = 0;
        info_out->device_build_ms = device_ms;
    }
    return TRH_OK;
}

int trh_set_bvh_cache_dir(const char* directory) {
    g_bvh_cache_dir = directory ? directory : "";
    return TRH_OK;
}

int trh_set_builder(trh_context* c, int builder) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_set_builder: ctx is NULL");
    if (builder != TRH_BUILDER_HOST_SAH && builder != TRH_BUILDER_DEVICE_LBVH) return fail(TRH_ERR_INVALID_ARG, "trh_set_builder: unknown builder");
    c->builder = builder;
    return TRH_OK;
}

int trh_set_camera(trh_context* c, const float eye[3], const float inv_view_proj[16]) {
    if (!c || !eye || !inv_view_proj) return fail(TRH_ERR_INVALID_ARG, "trh_set_camera: NULL argument");
    memcpy(c->sc.eye, eye, 12);
    memcpy(c->sc.ivp, inv_view_proj, 64);
    for (int i = 0; i < 4; ++i) c->sc.cam_k[i] = inv_view_proj[2 * 4 + i] * 1.f + inv_view_proj[3 * 4 + i] * 1.f;   // dev_trace.h camera_ray
    c->has_camera = true;
    return TRH_OK;
}

int trh_reset_accumulation(trh_context* c) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_reset_accumulation: ctx is NULL");
    if (!c->has_scene) return fail(TRH_ERR_STATE, "trh_reset_accumulation: no scene");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->accum, 0, (size_t)c->sc.width * c->sc.height * sizeof(float4), c->stream));
    c->accum_spp = 0;
    return TRH_OK;
}

int trh_set_debug_view(trh_context* c, int view, const float view3x3[9]) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_set_debug_view: ctx is NULL");
    if (view < TRH_VIEW_NONE || view > TRH_VIEW_EMISSIVE) return fail(TRH_ERR_INVALID_ARG, "trh_set_debug_view: unknown view");
    c->debug_view = (uint32_t)view;
    if (view3x3) memcpy(c->view3, view3x3, sizeof(c->view3));
    c->sc.debug_view = c->debug_view;
    memcpy(c->sc.view3, c->view3, sizeof(c->view3));
    return TRH_OK;
}

int trh_set_pipeline(trh_context* c, int pipeline) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_set_pipeline: ctx is NULL");
    if (pipeline != TRH_PIPELINE_MEGAKERNEL && pipeline != TRH_PIPELINE_MEGAKERNEL_SIMPLE && pipeline != TRH_PIPELINE_WAVEFRONT &&
        pipeline != TRH_PIPELINE_MEGAKERNEL_LANES)
        return fail(TRH_ERR_INVALID_ARG, "trh_set_pipeline: unknown pipeline");
    c->pipeline = pipeline;
    return TRH_OK;
}

int trh_render(trh_context* c, uint32_t spp_begin, uint32_t spp_count, uint64_t seed) { return render_impl(c, spp_begin, spp_count, seed, false); }
int trh_render_stats(trh_context* c, uint32_t spp_begin, uint32_t spp_count, uint64_t seed) { return render_impl(c, spp_begin, spp_count, seed, true); }

// trh_render for callers that must not allocate (HIP graph capture, tight serving loops): the same megakernel launch
// sequence, with the tile counter and the tile-order arrays laid out in this device's copy of g_scratch instead of
// hipMalloc'ed memory. Calls no hipMalloc / hipFree / hipEventCreate and keeps no state between calls: the tile order of
// the previous call is never trusted (another context on this GPU may have used the array since), so every call of
// >= 16 spp starts with the one-sample probe launch, exactly like trh_render on a context that has no order yet.
int trh_render_noalloc(trh_context* c, uint32_t spp_begin, uint32_t spp_count, uint64_t seed) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_render_noalloc: ctx is NULL");
    if (!c->has_scene || !c->has_camera) return fail(TRH_ERR_STATE, "trh_render_noalloc: scene and camera must be uploaded first");
    if (c->pipeline != TRH_PIPELINE_MEGAKERNEL && c->pipeline != TRH_PIPELINE_MEGAKERNEL_LANES)
        return fail(TRH_ERR_STATE, "trh_render_noalloc: only the megakernel pipeline runs without allocations");
    if (spp_count == 0) return TRH_OK;
    if (spp_count > (1u << 16)) return fail(TRH_ERR_LIMIT, "trh_render_noalloc: more than 65536 samples per pixel in one call");
    if ((uint64_t)spp_begin + spp_count > 0xffffffffull) return fail(TRH_ERR_LIMIT, "trh_render_noalloc: sample index exceeds 32 bits");
    if (c->device >= kMaxDevices || !g_scratch_ptr[c->device]) return fail(TRH_ERR_STATE, "trh_render_noalloc: no scratch address for this device");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t wl, hl, ntiles;
    choose_tiling(c, spp_count, wl, hl, ntiles);
    const bool ordered = c->tile_feedback && spp_count >= 8;
    unsigned char* const base = g_scratch_ptr[c->device];   // the device address, never the array's name
    TileState ts;
    ts.tile_counter = (unsigned int*)base;
    if (ordered) {
        if (ntiles > kNoAllocMaxTiles)
            return fail(TRH_ERR_LIMIT, "trh_render_noalloc: this frame size and sample count need " + std::to_string(ntiles) +
                                           " tiles, the static scratch holds " + std::to_string(kNoAllocMaxTiles) + " (use trh_render, or more samples per call)");
        const size_t temp = tile_order::scratch_bytes(ntiles);   // a size query: allocates nothing
        if (temp > kNoAllocSortTempBytes) return fail(TRH_ERR_LIMIT, "trh_render_noalloc: the tile sort needs more temporary storage than the static scratch reserves");
        uint32_t* arrays = (uint32_t*)(base + 256);
        ts.cost = arrays;
        ts.order = arrays + (size_t)kNoAllocMaxTiles;
        ts.keys = arrays + 2 * (size_t)kNoAllocMaxTiles;
        ts.keys_sorted = arrays + 3 * (size_t)kNoAllocMaxTiles;
        ts.iota = arrays + 4 * (size_t)kNoAllocMaxTiles;
        ts.sort_temp = base + 256 + 5 * kNoAllocArrayBytes;
        ts.sort_temp_bytes = kNoAllocSortTempBytes;
        ts.count = ntiles;
        ts.shape = (wl << 8) | hl;
        ts.order_valid = false;
        // the array keeps the last call's contents: the costs start at zero as in a fresh hipMalloc'ed + cleared array
        HIP_TRY(hipMemsetAsync(ts.cost, 0, (size_t)ntiles * sizeof(uint32_t), c->stream));
    }
    RenderParams rp{};
    rp.spp_begin = spp_begin;
    rp.spp_count = spp_count;
    rp.seed32 = seed_hash_of(seed);
    const int rc = c->sc.nodes16 ? launch_megakernel_on<false, true>(c, ts, rp, wl, hl, ordered) : launch_megakernel_on<false, false>(c, ts, rp, wl, hl, ordered);
    if (rc) return rc;
    c->samples += (uint64_t)c->sc.width * c->sc.height * spp_count;
    c->accum_spp += spp_count;
    return TRH_OK;
}

// development / test aid (not declared in the header): temporary storage the tile sort asks for at `ntiles` tiles
size_t trh_debug_sort_temp_bytes(uint32_t ntiles) { return tile_order::scratch_bytes(ntiles); }

int trh_synchronize(trh_context* c) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_synchronize: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TRH_OK;
}

int trh_accum_device_ptr(trh_context* c, void** dptr_out, size_t* nbytes_out) {
    if (!c || !dptr_out) return fail(TRH_ERR_INVALID_ARG, "trh_accum_device_ptr: NULL argument");
    if (!c->has_scene) return fail(TRH_ERR_STATE, "trh_accum_device_ptr: no scene");
    *dptr_out = c->accum;
    if (nbytes_out) *nbytes_out = (size_t)c->sc.width * c->sc.height * sizeof(float4);
    return TRH_OK;
}

int trh_bind_accum_buffer(trh_context* c, void* dptr, size_t nbytes) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_bind_accum_buffer: ctx is NULL");
    if (!c->has_scene) return fail(TRH_ERR_STATE, "trh_bind_accum_buffer: no scene");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!dptr) { c->accum = c->accum_internal; return TRH_OK; }
    const size_t need = (size_t)c->sc.width * c->sc.height * sizeof(float4);
    if (nbytes < need) return fail(TRH_ERR_INVALID_ARG, "trh_bind_accum_buffer: buffer smaller than width*height*16 bytes");
    if (((uintptr_t)dptr & 15u) != 0) return fail(TRH_ERR_INVALID_ARG, "trh_bind_accum_buffer: buffer must be 16-byte aligned");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dptr) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != c->device) {
        (void)hipGetLastError();
        return fail(TRH_ERR_INVALID_ARG, "trh_bind_accum_buffer: not device memory of this context's device");
    }
    c->accum = (float4*)dptr;
    return TRH_OK;
}

int trh_resolve(trh_context* c, float* rgba_host, uint64_t* samples_out) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_resolve: ctx is NULL");
    if (!c->has_scene) return fail(TRH_ERR_STATE, "trh_resolve: no scene");
    if (int rc = trh_internal::resolve_from(c, c->accum, rgba_host)) return rc;
    if (samples_out) *samples_out = c->accum_spp;
    return TRH_OK;
}

int trh_present_bgra8(trh_context* c, int tonemap, float exposure, uint32_t* bgra_host) {
    if (!c) return fail(TRH_ERR_INVALID_ARG, "trh_present_bgra8: ctx is NULL");
    if (!c->has_scene) return fail(TRH_ERR_STATE, "trh_present_bgra8: no scene");
    return trh_internal::present_from(c, c->accum, tonemap, exposure, bgra_host);
}

int trh_primary_hits(trh_context* c, uint32_t* ids_host, float* t_host, float* uv_host) {
    if (!c || !ids_host) return fail(TRH_ERR_INVALID_ARG, "trh_primary_hits: NULL argument");
    if (!c->has_scene || !c->has_camera) return fail(TRH_ERR_STATE, "trh_primary_hits: scene and camera must be uploaded first");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = (uint32_t)(c->sc.width * c->sc.height);
    if (int prc = ensure_probe_buffer(c, (size_t)n * 16)) return prc;   // ids | t | uv, freed with the context
    uint32_t* d_ids = (uint32_t*)c->probe_buf;
    float* d_t = (float*)c->probe_buf + n;
    float2* d_uv = (float2*)((float*)c->probe_buf + 2 * (size_t)n);
    hipLaunchKernelGGL(c->sc.nodes16 ? k_primary_hits<true> : k_primary_hits<false>, dim3((n + TRH_BLOCK - 1) / TRH_BLOCK), dim3(TRH_BLOCK), 0, c->stream, c->sc, d_ids, d_t, d_uv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ids_host, d_ids, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (t_host) HIP_TRY(hipMemcpyAsync(t_host, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (uv_host) HIP_TRY(hipMemcpyAsync(uv_host, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TRH_OK;
}

int trh_sample_radiance(trh_context* c, uint32_t sample_index, uint64_t seed, float* rgba_host) {
    if (!c || !rgba_host) return fail(TRH_ERR_INVALID_ARG, "trh_sample_radiance: NULL argument");
    if (!c->has_scene || !c->has_camera) return fail(TRH_ERR_STATE, "trh_sample_radiance: scene and camera must be uploaded first");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = (uint32_t)(c->sc.width * c->sc.height);
    if (int prc = ensure_probe_buffer(c, (size_t)n * sizeof(float4))) return prc;   // freed with the context
    float4* d_out = (float4*)c->probe_buf;
    hipLaunchKernelGGL(c->sc.nodes16 ? k_sample_radiance<true> : k_sample_radiance<false>, dim3((n + TRH_BLOCK - 1) / TRH_BLOCK), dim3(TRH_BLOCK), 0, c->stream, c->sc, sample_index, seed_hash_of(seed), d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgba_host, d_out, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TRH_OK;
}

int trh_get_stats(trh_context* c, trh_stats* out, int reset) {
    if (!c || !out) return fail(TRH_ERR_INVALID_ARG, "trh_get_stats: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = collect_timings(c);
    if (rc) return rc;
    DevCounters h{};
    HIP_TRY(hipMemcpy(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
    out->rays = h.rays;
    out->samples = c->samples;
    out->box_tests = h.box_tests;
    out->tri_tests = h.tri_tests;
    out->hits = h.hits;
    out->tex_fetches = h.tex_fetches;
    out->kernel_ms = c->kernel_ms;
    out->launches = c->launches;
    if (reset) {
        HIP_TRY(hipMemset(c->counters, 0, sizeof(DevCounters)));
        c->samples = 0;
        c->kernel_ms = 0.0;
        c->launches = 0;
    }
    return TRH_OK;
}

int trh_get_debug_counters(trh_context* c, uint64_t* out, uint32_t n) {
    if (!c || !out) return fail(TRH_ERR_INVALID_ARG, "trh_get_debug_counters: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    DevCounters h{};
    HIP_TRY(hipMemcpy(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
    const uint64_t v[19] = {h.wave_inner_iters, h.wave_leaf_iters, h.wave_shade_calls, h.wave_gen_calls, h.lanes_gen,
                            h.phase_cycles[0], h.phase_cycles[1], h.phase_cycles[2], h.phase_cycles[3], h.phase_cycles[4], h.max_wave_cycles, h.waves,
                            ~h.min_start, h.max_end, h.sum_start, h.sum_end, h.idle_leaf_in_inner, h.idle_wait_in_inner, h.idle_none_in_inner};
    for (uint32_t i = 0; i < n; ++i) out[i] = i < 19 ? v[i] : 0;
    return TRH_OK;
}

int trh_get_bvh_info(trh_context* c, trh_bvh_info* out) {
    if (!c || !out) return fail(TRH_ERR_INVALID_ARG, "trh_get_bvh_info: NULL argument");
    if (!c->has_scene) return fail(TRH_ERR_STATE, "trh_get_bvh_info: no scene");
    *out = c->bvh;
    return TRH_OK;
}

}  // extern "C"

// ---- internal interface of the multi-device group (ctx_internal.h, tracy_group.hip)
namespace trh_internal {

int fail(int code, const std::string& msg) { return ::fail(code, msg); }
int device_of(const trh_context* c) { return c->device; }
hipStream_t stream_of(const trh_context* c) { return c->stream; }
int builder_of(const trh_context* c) { return c->builder; }
bool can_render(const trh_context* c) { return c->has_scene && c->has_camera; }
float* accum_of(const trh_context* c, size_t* float_count) {
    if (float_count) *float_count = (size_t)c->sc.width * c->sc.height * 4;
    return (float*)c->accum;
}

int build_host_shared(const trh_scene_desc* d, bvh::Result& out, bool& from_cache, double& build_ms) {
    if (!d) return fail(TRH_ERR_INVALID_ARG, "trh_group_scene_upload: NULL argument");
    std::vector<bvh::InputTriangle> tris;
    bool any_normal_map = false;
    if (int frc = flatten_triangles(d, tris, any_normal_map)) return frc;
    const auto t0 = std::chrono::steady_clock::now();
    bvh::build_cached(tris, out, g_bvh_cache_dir, &from_cache);
    build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return TRH_OK;
}

int scene_upload_shared(trh_context* c, const trh_scene_desc* d, const bvh::Result* shared, bool from_cache, double build_ms) {
    return scene_upload_impl(c, d, shared, from_cache, build_ms);
}

int resolve_from(trh_context* c, const void* sums, float* rgba_host) {
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = (uint32_t)(c->sc.width * c->sc.height);
    hipLaunchKernelGGL(k_resolve, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float4*)sums, c->resolved, n);
    HIP_TRY(hipGetLastError());
    if (rgba_host) HIP_TRY(hipMemcpyAsync(rgba_host, c->resolved, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TRH_OK;
}

int present_from(trh_context* c, const void* sums, int tonemap, float exposure, uint32_t* bgra_host) {
    if (tonemap < TRH_TONEMAP_NONE || tonemap > TRH_TONEMAP_ACES) return fail(TRH_ERR_INVALID_ARG, "trh_present_bgra8: unknown tonemap");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = (uint32_t)(c->sc.width * c->sc.height);
    hipLaunchKernelGGL(k_present, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float4*)sums, c->bitmap, c->sc.width, c->sc.height, tonemap, exposure);
    HIP_TRY(hipGetLastError());
    if (bgra_host) HIP_TRY(hipMemcpyAsync(bgra_host, c->bitmap, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return TRH_OK;
}

}  // namespace trh_internal
