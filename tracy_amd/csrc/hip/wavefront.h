// wavefront.h -- the queue-based ("wavefront") pipeline of the hip module (TRH_PIPELINE_WAVEFRONT).
//
// Same per-sample paths as the megakernel (same RNG keys, same device functions of dev_trace.h), but
// organised as the classic streaming path tracer: path state lives in an 80-byte record per slot in HBM, index queues
// connect the stages, every stage is its own kernel:
//
//   k_wf_generate : free slots -> new camera paths (next items of the batch) -> extend queue
//   k_wf_extend   : persistent waves pull rays from the extend queue (one atomic per refill), run the
//                   majority-vote BVH traversal with LDS stacks, refill idle lanes from the queue when
//                   `refill_thresh` lanes are idle (dynamic fetch), write 16-byte hit records
//   k_wf_shade    : one lane per queue entry: Material::Scatter / sky / Russian roulette; survivors are
//                   compacted into the next extend queue, finished paths write their radiance to the
//                   per-item buffer and their slot to the free queue (wave64 ballot + mbcnt prefix, LDS
//                   prefix over the block's waves, one 64-bit atomicAdd per 1024-thread block)
//   k_wf_advance  : 1 thread: swaps queue counters, advances the item cursor
//   k_wf_accumulate: per pixel, sums the batch's samples in increasing sample index into `accum`
//                   (deterministic, no float atomics)
//
// The pool holds WF_POOL path slots (80 B of state each); finished paths are replaced at once
// ("regeneration"), so every extend launch sees a full queue until the batch's items run out.
#pragma once
#include "dev_trace.h"

#define WF_CHUNK 256u

namespace dev {

struct WfCtl {
    uint32_t n_cur;        // entries in the current extend queue
    uint32_t head;         // extend queue read cursor (dynamic fetch)
    uint32_t n_next;       // entries appended to the next extend queue so far   } one 8-byte word:
    uint32_t n_free;       // entries in the free-slot queue                     } bumped by ONE atomic per block
    uint32_t item_cursor;  // next unassigned item of the batch
    uint32_t items_end;    // items in the batch (= pixels * samples of the batch)
    uint32_t pad[2];
};
#define WF_SHADE_BLOCK 1024

struct WfBuffers {
    // path state, 80 contiguous bytes per slot (5 float4): a queue entry is a random slot, so the five
    // 16-byte fields of one path must share cache lines (as five separate arrays every field cost a
    // 64-byte line of its own: measured 0.81 ms per 4M-entry shade launch)
    //   [0] o.xyz, rng   [1] d.xyz, item   [2] throughput.xyz, bounce   [3] radiance.xyz, -   [4] hit: t, u, v, prim
    float4* st;
    uint32_t* q_a;      // extend queue (ping)
    uint32_t* q_b;      // extend queue (pong)
    uint32_t* q_free;   // free slots
    float4* item_rad;   // per item of the batch: radiance.xyz, bounce iterations
    WfCtl* ctl;
    uint32_t pool;
};

}  // namespace dev

using namespace dev;

__global__ __launch_bounds__(256) void k_wf_generate(DevScene sc, WfBuffers wf, uint32_t* __restrict__ q_next, uint32_t spp_begin, uint32_t seed_hash) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const WfCtl ctl = *wf.ctl;
    const bool in_range = i < ctl.n_free;
    const uint32_t item = ctl.item_cursor + i;
    const bool make = in_range && item < ctl.items_end;
    uint32_t slot = 0;
    if (make) {
        slot = wf.q_free[i];
        const uint32_t npix = (uint32_t)(sc.width * sc.height);
        const uint32_t p = item % npix, s = item / npix;
        PathState ps;
        generate(sc, ps, (int)(p % (uint32_t)sc.width), (int)(p / (uint32_t)sc.width), spp_begin + s, seed_hash);
        wf.st[slot * 5u + 0u] = make_float4(ps.o.x, ps.o.y, ps.o.z, __uint_as_float(ps.rng));
        wf.st[slot * 5u + 1u] = make_float4(ps.d.x, ps.d.y, ps.d.z, __uint_as_float(item));
        wf.st[slot * 5u + 2u] = make_float4(1.f, 1.f, 1.f, __uint_as_float(0u));
        wf.st[slot * 5u + 3u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the new paths are a prefix of the free queue: they go behind the survivors shade appended,
    // no atomics needed (k_wf_advance adds the count)
    if (make) q_next[ctl.n_next + i] = slot;
}

// 1 thread. After generate: the next queue becomes the current one.
__global__ void k_wf_advance(WfBuffers wf) {
    WfCtl c = *wf.ctl;
    const uint32_t taken = min(c.n_free, c.items_end - c.item_cursor);
    c.item_cursor += taken;
    c.n_cur = c.n_next + taken;
    c.n_next = 0;
    c.n_free = 0;
    c.head = 0;
    *wf.ctl = c;
}

template <bool N16>
__global__ __launch_bounds__(TRH_BLOCK) void k_wf_extend(DevScene sc, WfBuffers wf, const uint32_t* __restrict__ q_cur, uint32_t stack_entries,
                                                         uint32_t refill_thresh, uint32_t sched_beta4) {
    extern __shared__ int s_dyn[];
    (void)stack_entries;
    int* stack = s_dyn + threadIdx.x;
    stack[0] = -1;   // sentinel row (dev_trace.h)
    const uint32_t n = wf.ctl->n_cur;
    Counters cnt = {};
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 0.f);
    Trav tv;
    tv.inv = tv.oi = mk(0.f, 0.f, 0.f);
    tv.selx = tv.sely = tv.selz = 0x03020100u;
    tv.node = -1;
    tv.top = stack;
    tv.hit.t = 0.f; tv.hit.u = tv.hit.v = 0.f; tv.hit.prim = -1; tv.hit.id = TRH_MISS_ID;
    uint32_t chunk_pos = 0, chunk_end = 0;   // wave-uniform: the wave's current chunk of the queue
    bool queue_empty = false;                // wave-uniform: the last chunk fetch found the queue exhausted
    bool has_ray = false;     // this lane owns a ray whose traversal is not written back yet
    bool drained = false;     // the queue had nothing left for this lane
    uint32_t slot = 0;
    for (;;) {
        // ---- write back finished rays
        if (has_ray && !trav_active(tv)) {
            wf.st[slot * 5u + 4u] = make_float4(tv.hit.t, tv.hit.u, tv.hit.v, __int_as_float(tv.hit.prim));
            has_ray = false;
        }
        // ---- dynamic fetch: idle lanes take the next rays of the wave's current chunk; a new chunk of
        // WF_CHUNK queue entries costs one atomicAdd (a single word saturates at ~88 dequeues/us on this
        // chip, so per-refill atomics from ~6000 waves would serialise the whole kernel)
        const bool want = !has_ray && !drained;
        const unsigned long long wmask = __ballot(want);
        if (wmask != 0ull) {
            if (chunk_pos >= chunk_end) {
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == 0u) base = atomicAdd(&wf.ctl->head, WF_CHUNK);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                chunk_pos = base < n ? base : n;
                chunk_end = base + WF_CHUNK < n ? base + WF_CHUNK : n;
                if (base >= n) queue_empty = true;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
            if (want) {
                const uint32_t idx = chunk_pos + rank;
                if (idx < chunk_end) {
                    slot = q_cur[idx];
                    const float4 a = wf.st[slot * 5u + 0u], b = wf.st[slot * 5u + 1u];
                    o = mk(a.x, a.y, a.z);
                    d = mk(b.x, b.y, b.z);
                    trav_begin(sc, tv, o, d, stack);
                    has_ray = true;
                } else if (queue_empty) {
                    drained = true;
                }
            }
            const uint32_t take = (uint32_t)__popcll(wmask);
            chunk_pos = chunk_pos + take < chunk_end ? chunk_pos + take : chunk_end;
        }
        if (__ballot(has_ray) == 0ull) break;   // every lane drained: all waves reach this exit
        // ---- majority-vote traversal until enough lanes are idle again
        const uint32_t nP = (uint32_t)__popcll(__ballot(has_ray));
        for (;;) {
            const uint32_t nI = (uint32_t)__popcll(__ballot(tv.node >= 0));
            const uint32_t nL = (uint32_t)__popcll(__ballot(tv.node < -1));
            if ((nI | nL) == 0u || nP - nI - nL >= refill_thresh) break;
            if (nI != 0u && nI * 4u >= nL * sched_beta4) {
                if (tv.node >= 0) trav_inner<false, N16>(sc, o, tv, stack, cnt);
            } else {
                if (tv.node < -1) trav_leaf<false>(sc, o, d, tv, stack, cnt);
            }
        }
    }
}

__global__ __launch_bounds__(WF_SHADE_BLOCK) void k_wf_shade(DevScene sc, WfBuffers wf, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                                  DevCounters* __restrict__ counters) {
    __shared__ uint32_t s_cnt[2][WF_SHADE_BLOCK / 64];
    __shared__ uint32_t s_base[2];
    __shared__ uint32_t s_rays[WF_SHADE_BLOCK / 64];
    const uint32_t i = blockIdx.x * WF_SHADE_BLOCK + threadIdx.x;
    const uint32_t n = wf.ctl->n_cur;
    const bool valid = i < n;
    Counters cnt = {};
    uint32_t slot = 0;
    bool cont = false;
    if (valid) {
        slot = q_cur[i];
        const float4 a = wf.st[slot * 5u + 0u], b = wf.st[slot * 5u + 1u], c = wf.st[slot * 5u + 2u], r = wf.st[slot * 5u + 3u], h = wf.st[slot * 5u + 4u];
        PathState ps;
        ps.o = mk(a.x, a.y, a.z);
        ps.rng = __float_as_uint(a.w);
        ps.d = mk(b.x, b.y, b.z);
        const uint32_t item = __float_as_uint(b.w);
        ps.throughput = mk(c.x, c.y, c.z);
        ps.bounce = __float_as_uint(c.w);
        ps.radiance = mk(r.x, r.y, r.z);
        Hit hit;
        hit.t = h.x; hit.u = h.y; hit.v = h.z; hit.prim = __float_as_int(h.w); hit.id = 0;
        cont = shade<false>(sc, ps, hit, cnt);
        if (cont) {
            wf.st[slot * 5u + 0u] = make_float4(ps.o.x, ps.o.y, ps.o.z, __uint_as_float(ps.rng));
            wf.st[slot * 5u + 1u] = make_float4(ps.d.x, ps.d.y, ps.d.z, __uint_as_float(item));
            wf.st[slot * 5u + 2u] = make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, __uint_as_float(ps.bounce));
            wf.st[slot * 5u + 3u] = make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z, 0.f);
        } else {
            wf.item_rad[item] = make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z, (float)ps.bounce);
        }
    }
    // ---- queue compaction: wave64 ballot + mbcnt prefix inside a wave, LDS prefix over the 16 waves of
    // the block, ONE 64-bit atomicAdd per block that reserves room in both queues at once
    const bool to_next = valid && cont, to_free = valid && !cont;
    const unsigned long long m_next = __ballot(to_next), m_free = __ballot(to_free);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave_rays = wave_sum(cnt.rays);
    if (lane == 0u) { s_cnt[0][wave] = (uint32_t)__popcll(m_next); s_cnt[1][wave] = (uint32_t)__popcll(m_free); s_rays[wave] = wave_rays; }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t tn = 0, tf = 0, rays = 0;
        for (uint32_t w = 0; w < WF_SHADE_BLOCK / 64; ++w) {
            const uint32_t a = s_cnt[0][w], b = s_cnt[1][w];
            s_cnt[0][w] = tn; s_cnt[1][w] = tf;
            tn += a; tf += b;
            rays += s_rays[w];
        }
        // ++raycount_ (cpu_trace.cpp:115-116): one atomic per block, not per wave -- 65k same-word atomics
        // per launch cost 0.7 ms on their own
        if (rays) atomicAdd(&counters->rays, (unsigned long long)rays);
        unsigned long long old = 0ull;
        if ((tn | tf) != 0u) old = atomicAdd(reinterpret_cast<unsigned long long*>(&wf.ctl->n_next), ((unsigned long long)tf << 32) | tn);
        s_base[0] = (uint32_t)old;
        s_base[1] = (uint32_t)(old >> 32);
    }
    __syncthreads();
    if (to_next) q_next[s_base[0] + s_cnt[0][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_next >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_next, 0u))] = slot;
    if (to_free) wf.q_free[s_base[1] + s_cnt[1][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_free >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_free, 0u))] = slot;
}

// per pixel: accum += sum over the batch's samples, in increasing sample index
__global__ void k_wf_accumulate(const float4* __restrict__ item_rad, float4* __restrict__ accum, uint32_t npix, uint32_t spp) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float x = 0.f, y = 0.f, z = 0.f;
    for (uint32_t s = 0; s < spp; ++s) {
        const float4 r = item_rad[(size_t)s * npix + p];
        x += r.x; y += r.y; z += r.z;
    }
    float4 a = accum[p];
    a.x += x; a.y += y; a.z += z; a.w += (float)spp;
    accum[p] = a;
}

// start of a batch: every slot free, the item cursor at 0 (thread 0 also resets the control block: no host-side
// staging buffer, no synchronisation before the first iteration)
__global__ void k_wf_init_free(uint32_t* __restrict__ q_free, uint32_t n, WfCtl* __restrict__ ctl, uint32_t items) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q_free[i] = i;
    if (i == 0u) {
        WfCtl c{};
        c.n_free = n;
        c.items_end = items;
        *ctl = c;
    }
}
