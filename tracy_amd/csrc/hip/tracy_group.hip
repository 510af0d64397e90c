// tracy_group.hip -- the multi-GPU form of the C ABI (trh_group_*, include/tracy_hip.h).
//
// Tracy's host is one process with one thread (src/win_raytracer.cpp:431-589, "// TODO: threads" :493), so a
// `-kernel hip` module that is to use the 8 GPUs of a node drives them itself: one trh_context (own stream, own
// replica of the scene) per device, the samples of every render call split by sample index, and ONE exchange step
// when a picture is wanted: ncclReduce(sum, float, W*H*4, root = device 0) over xGMI inside ncclGroupStart/End,
// each rank's reduce enqueued on that device's render stream (so it is ordered after the device's render launches
// without any host synchronisation), then mean / tonemap on device 0. Nothing else crosses devices: paths are
// independent, the accumulation is a commutative sum.
//
// TRH_GROUP_ALIAS_DEVICE=1 (a test knob for boxes with fewer GPUs than members): the members of a group may share
// devices -- member i runs on device i mod (devices present), every member with its own context, stream, scene replica
// and accumulation buffer -- and the exchange step is a device-side sum (k_group_add on the root's stream, ordered after
// the members' streams by events) where RCCL would refuse a communicator with duplicate devices. Everything else of
// the N > 1 path (shared host-built BVH uploaded N times, the sample split, N launches in flight from one thread, the
// separate `reduced` buffer, resolve / present from it) is the code a real N-GPU group runs.
//
// RCCL is bound at run time (dlopen of librccl.so.1) the first time a group of more than one device is made: hosts
// with one GPU never load it, and a process that already holds a copy (PyTorch ships one) shares that copy instead
// of mapping a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ctx_internal.h"

using trh_internal::fail;

namespace {

#define HIP_TRY(expr)                                                                                           \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return fail(TRH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

bool load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return true;
    const char* names[] = {getenv("TRH_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        g_rccl.error = dlerror();
    }
    if (!h) return false;
    Rccl r;
    r.handle = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))dlsym(h, "ncclReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        g_rccl.error = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclReduce / ncclGroupStart / ncclGroupEnd / ncclGetErrorString";
        dlclose(h);
        return false;
    }
    g_rccl = r;
    return true;
}

#define NCCL_TRY(expr)                                                                                                         \
    do {                                                                                                                       \
        ncclResult_t r_ = (expr);                                                                                              \
        if (r_ != ncclSuccess) return fail(TRH_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));               \
    } while (0)

}  // namespace

// reduced += member (float frames on one device): the alias-device form of the exchange step
__global__ void k_group_add(float4* __restrict__ reduced, const float4* __restrict__ member, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = reduced[i];
    const float4 b = member[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reduced[i] = a;
}

struct trh_group {
    std::vector<trh_context*> ctx;     // ctx[g] renders on device g; ctx[0] is the root (reduce target, resolve, present)
    std::vector<ncclComm_t> comms;     // one communicator per device, made by ncclCommInitAll; empty without RCCL
    bool use_rccl = false;
    bool alias = false;                // TRH_GROUP_ALIAS_DEVICE: members share devices, the exchange step is k_group_add
    bool exchange = false;             // the group has an exchange step (RCCL or alias): resolve / present read `reduced`
    bool torn = false;                 // a render call failed after some members had been launched: the sums are not those of
                                       // accum_spp samples any more; every call but reset / upload / shutdown refuses
    std::vector<hipEvent_t> done;      // alias: one event per member, recorded on its stream before the sum
    float* reduced = nullptr;          // device 0: W*H float4, the sum over devices (the per-device sums stay untouched)
    size_t reduced_floats = 0;
    bool has_scene = false;
    uint64_t accum_spp = 0;            // samples per pixel over all devices since the last reset
};

extern "C" {

void trh_split_samples(uint32_t spp_begin, uint32_t spp_count, int ngpus, int g, uint32_t* begin_out, uint32_t* count_out) {
    // device g of G owns [g*S/G, (g+1)*S/G) of the call's S samples (BASELINE.json configs[3])
    const uint64_t lo = (uint64_t)spp_count * (uint64_t)g / (uint64_t)ngpus;
    const uint64_t hi = (uint64_t)spp_count * (uint64_t)(g + 1) / (uint64_t)ngpus;
    if (begin_out) *begin_out = spp_begin + (uint32_t)lo;
    if (count_out) *count_out = (uint32_t)(hi - lo);
}

int trh_group_shutdown(trh_group* g) {
    if (!g) return TRH_OK;
    for (size_t i = 0; i < g->ctx.size(); ++i)
        if (g->ctx[i]) (void)trh_synchronize(g->ctx[i]);
    if (g->use_rccl)
        for (ncclComm_t c : g->comms)
            if (c) (void)g_rccl.CommDestroy(c);
    for (hipEvent_t e : g->done)
        if (e) (void)hipEventDestroy(e);
    if (g->reduced && !g->ctx.empty() && g->ctx[0]) {
        (void)hipSetDevice(trh_internal::device_of(g->ctx[0]));
        (void)hipFree(g->reduced);
    }
    for (trh_context* c : g->ctx) (void)trh_shutdown(c);
    delete g;
    return TRH_OK;
}

int trh_group_init(int ngpus_requested, int* ngpus_out, trh_group** group_out) {
    if (!group_out) return fail(TRH_ERR_INVALID_ARG, "trh_group_init: group_out is NULL");
    *group_out = nullptr;
    if (ngpus_out) *ngpus_out = 0;
    int have = 0;
    if (int rc = trh_device_count(&have)) return rc;
    if (have == 0) return fail(TRH_ERR_NO_DEVICE, "trh_group_init: no HIP device (this module has no CPU fallback)");
    const char* alias_env = getenv("TRH_GROUP_ALIAS_DEVICE");
    const bool alias = alias_env && atoi(alias_env) != 0 && ngpus_requested > have;
    if (alias && ngpus_requested > 64) return fail(TRH_ERR_LIMIT, "trh_group_init: more than 64 members");
    const int n = alias ? ngpus_requested : (ngpus_requested <= 0 ? have : std::min(ngpus_requested, have));   // <= 0: every device of the node
    auto* g = new trh_group();
    g->alias = alias;
    g->ctx.assign((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) {
        if (int rc = trh_init(alias ? 0 : i, &g->ctx[(size_t)i])) {
            const std::string why = trh_last_error();
            trh_group_shutdown(g);
            return fail(rc, why);
        }
    }
    // one device needs no exchange step; TRH_GROUP_FORCE_RCCL=1 runs it all the same (a 1-rank reduce is a copy), so
    // that the RCCL path can be exercised on a one-GPU box
    const char* force = getenv("TRH_GROUP_FORCE_RCCL");
    g->use_rccl = !alias && (n > 1 || (force && atoi(force) != 0));
    g->exchange = g->use_rccl || alias;
    if (alias) {
        g->done.assign((size_t)n, nullptr);
        for (int i = 0; i < n; ++i)
            if (hipEventCreateWithFlags(&g->done[(size_t)i], hipEventDisableTiming) != hipSuccess) {
                trh_group_shutdown(g);
                return fail(TRH_ERR_HIP, "trh_group_init: hipEventCreate failed");
            }
    }
    if (g->use_rccl) {
        if (!load_rccl()) {
            const std::string why = "trh_group_init: cannot load RCCL (" + g_rccl.error + ")";
            g->use_rccl = false;
            trh_group_shutdown(g);
            return fail(TRH_ERR_HIP, why);
        }
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; ++i) devs[(size_t)i] = i;
        g->comms.assign((size_t)n, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n, devs.data());
        if (r != ncclSuccess) {
            const std::string why = std::string("trh_group_init: ncclCommInitAll: ") + g_rccl.GetErrorString(r);
            g->comms.clear();
            trh_group_shutdown(g);
            return fail(TRH_ERR_HIP, why);
        }
    }
    if (ngpus_out) *ngpus_out = n;
    *group_out = g;
    return TRH_OK;
}

int trh_group_size(const trh_group* g) { return g ? (int)g->ctx.size() : 0; }

int trh_group_context(trh_group* g, int index, trh_context** ctx_out) {
    if (!g || !ctx_out) return fail(TRH_ERR_INVALID_ARG, "trh_group_context: NULL argument");
    if (index < 0 || index >= (int)g->ctx.size()) return fail(TRH_ERR_INVALID_ARG, "trh_group_context: index out of range");
    *ctx_out = g->ctx[(size_t)index];
    return TRH_OK;
}

int trh_group_scene_upload(trh_group* g, const trh_scene_desc* scene) {
    if (!g || !scene) return fail(TRH_ERR_INVALID_ARG, "trh_group_scene_upload: NULL argument");
    // a failure below leaves the members with different scenes: has_scene stays false and every render / resolve call
    // refuses until an upload has gone through on all of them
    g->has_scene = false;
    g->torn = false;
    // the host builder runs ONCE; every device gets its own replica of the arrays (dragon: 15 MB). The device-side
    // builder runs on every device (milliseconds, and the arrays are born in that device's HBM).
    bvh::Result shared;
    bool from_cache = false, have_shared = false;
    double build_ms = 0.0;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const bool host_builder = trh_internal::builder_of(g->ctx[i]) == TRH_BUILDER_HOST_SAH;
        if (host_builder && !have_shared && g->ctx.size() > 1) {
            if (int rc = trh_internal::build_host_shared(scene, shared, from_cache, build_ms)) return rc;
            have_shared = true;
        }
        if (int rc = trh_internal::scene_upload_shared(g->ctx[i], scene, have_shared ? &shared : nullptr, from_cache, build_ms)) return rc;
    }
    if (g->exchange) {
        size_t floats = 0;
        (void)trh_internal::accum_of(g->ctx[0], &floats);
        HIP_TRY(hipSetDevice(trh_internal::device_of(g->ctx[0])));
        if (floats != g->reduced_floats) {
            HIP_TRY(hipStreamSynchronize(trh_internal::stream_of(g->ctx[0])));
            (void)hipFree(g->reduced);
            g->reduced = nullptr;
            g->reduced_floats = 0;
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, floats * sizeof(float)));
            g->reduced = (float*)p;
            g->reduced_floats = floats;
        }
    }
    g->accum_spp = 0;
    g->has_scene = true;
    return TRH_OK;
}

int trh_group_set_builder(trh_group* g, int builder) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_set_builder: group is NULL");
    for (trh_context* c : g->ctx)
        if (int rc = trh_set_builder(c, builder)) return rc;
    return TRH_OK;
}

int trh_group_set_camera(trh_group* g, const float eye[3], const float inv_view_proj[16]) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_set_camera: group is NULL");
    for (trh_context* c : g->ctx)
        if (int rc = trh_set_camera(c, eye, inv_view_proj)) return rc;
    return TRH_OK;
}

int trh_group_set_debug_view(trh_group* g, int view, const float view3x3[9]) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_set_debug_view: group is NULL");
    for (trh_context* c : g->ctx)
        if (int rc = trh_set_debug_view(c, view, view3x3)) return rc;
    return TRH_OK;
}

int trh_group_reset_accumulation(trh_group* g) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_reset_accumulation: group is NULL");
    for (trh_context* c : g->ctx)
        if (int rc = trh_reset_accumulation(c)) return rc;
    g->accum_spp = 0;
    g->torn = false;
    return TRH_OK;
}

int trh_group_render(trh_group* g, uint32_t spp_begin, uint32_t spp_count, uint64_t seed) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_render: group is NULL");
    if (!g->has_scene) return fail(TRH_ERR_STATE, "trh_group_render: no scene");
    if (g->torn) return fail(TRH_ERR_STATE, "trh_group_render: an earlier render call failed part-way; call trh_group_reset_accumulation first");
    if ((uint64_t)spp_begin + spp_count > 0xffffffffull) return fail(TRH_ERR_LIMIT, "trh_group_render: sample index exceeds 32 bits");
    const int n = (int)g->ctx.size();
    // everything trh_render can refuse is checked for ALL members before the first launch, so that a refusal leaves
    // the per-member sums and accum_spp in step
    for (int i = 0; i < n; ++i)
        if (!trh_internal::can_render(g->ctx[(size_t)i])) return fail(TRH_ERR_STATE, "trh_group_render: scene and camera must be set on every member first");
    // asynchronous on every device's stream: the host thread only enqueues, the N launches run side by side
    int launched = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t b = 0, cnt = 0;
        trh_split_samples(spp_begin, spp_count, n, i, &b, &cnt);
        if (cnt == 0) continue;
        const char* inject = getenv("TRH_GROUP_FAIL_MEMBER");   // test knob: member k fails its launch (tests/test_group.py)
        int rc = (inject && *inject && atoi(inject) == i) ? fail(TRH_ERR_HIP, "trh_group_render: injected failure (TRH_GROUP_FAIL_MEMBER)")
                                                           : trh_render(g->ctx[(size_t)i], b, cnt, seed);
        if (rc) {
            // a device error after earlier members were launched: their buffers hold samples accum_spp does not count
            if (launched > 0) g->torn = true;
            return rc;
        }
        ++launched;
    }
    g->accum_spp += spp_count;
    return TRH_OK;
}

int trh_group_synchronize(trh_group* g) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_synchronize: group is NULL");
    for (trh_context* c : g->ctx)
        if (int rc = trh_synchronize(c)) return rc;
    return TRH_OK;
}

}  // extern "C"

namespace {

// the alias-device exchange step: reduced = member 0, then += every other member, on the root's stream after each
// member's stream has reached this point (same order of additions as a chain reduce over ranks 0, 1, 2, ...)
int sum_to_root_alias(trh_group* g) {
    const int n = (int)g->ctx.size();
    hipStream_t root = trh_internal::stream_of(g->ctx[0]);
    HIP_TRY(hipSetDevice(trh_internal::device_of(g->ctx[0])));
    for (int i = 0; i < n; ++i) {
        size_t floats = 0;
        float* acc = trh_internal::accum_of(g->ctx[(size_t)i], &floats);
        if (floats != g->reduced_floats) return fail(TRH_ERR_STATE, "trh_group: members hold frames of different sizes");
        if (i > 0) {
            HIP_TRY(hipEventRecord(g->done[(size_t)i], trh_internal::stream_of(g->ctx[(size_t)i])));
            HIP_TRY(hipStreamWaitEvent(root, g->done[(size_t)i], 0));
        }
        if (i == 0) {
            HIP_TRY(hipMemcpyAsync(g->reduced, acc, floats * sizeof(float), hipMemcpyDeviceToDevice, root));
        } else {
            const uint32_t px = (uint32_t)(floats / 4);
            hipLaunchKernelGGL(k_group_add, dim3((px + 255) / 256), dim3(256), 0, root, (float4*)g->reduced, (const float4*)acc, px);
            HIP_TRY(hipGetLastError());
        }
    }
    // the members must not render into their buffers before the sum has read them
    HIP_TRY(hipEventRecord(g->done[0], root));
    for (int i = 1; i < n; ++i) HIP_TRY(hipStreamWaitEvent(trh_internal::stream_of(g->ctx[(size_t)i]), g->done[0], 0));
    return TRH_OK;
}

// the exchange step: sum of the per-device accumulation buffers into g->reduced on device 0
int reduce_to_root(trh_group* g) {
    if (g->alias) return sum_to_root_alias(g);
    const int n = (int)g->ctx.size();
    NCCL_TRY(g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
        size_t floats = 0;
        float* acc = trh_internal::accum_of(g->ctx[(size_t)i], &floats);
        if (floats != g->reduced_floats) {
            (void)g_rccl.GroupEnd();
            return fail(TRH_ERR_STATE, "trh_group: devices hold frames of different sizes");
        }
        // (the examples of the library make the communicator's device current before every call of a single-threaded group;
        // later versions no longer need it, and it costs nothing)
        if (hipSetDevice(trh_internal::device_of(g->ctx[(size_t)i])) != hipSuccess) {
            (void)g_rccl.GroupEnd();
            return fail(TRH_ERR_HIP, "trh_group: hipSetDevice failed before ncclReduce");
        }
        const ncclResult_t r = g_rccl.Reduce(acc, i == 0 ? g->reduced : acc, floats, ncclFloat32, ncclSum, 0, g->comms[(size_t)i], trh_internal::stream_of(g->ctx[(size_t)i]));
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail(TRH_ERR_HIP, std::string("ncclReduce: ") + g_rccl.GetErrorString(r));
        }
    }
    NCCL_TRY(g_rccl.GroupEnd());
    HIP_TRY(hipSetDevice(trh_internal::device_of(g->ctx[0])));
    return TRH_OK;
}

}  // namespace

extern "C" {

int trh_group_resolve(trh_group* g, float* rgba_host, uint64_t* samples_out) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_resolve: group is NULL");
    if (!g->has_scene) return fail(TRH_ERR_STATE, "trh_group_resolve: no scene");
    if (g->torn) return fail(TRH_ERR_STATE, "trh_group_resolve: an earlier render call failed part-way; call trh_group_reset_accumulation first");
    if (!g->exchange) {
        if (int rc = trh_resolve(g->ctx[0], rgba_host, nullptr)) return rc;
    } else {
        if (int rc = reduce_to_root(g)) return rc;
        if (int rc = trh_internal::resolve_from(g->ctx[0], g->reduced, rgba_host)) return rc;   // synchronises device 0's stream
        for (size_t i = 1; i < g->ctx.size(); ++i)
            if (int rc = trh_synchronize(g->ctx[i])) return rc;
    }
    if (samples_out) *samples_out = g->accum_spp;
    return TRH_OK;
}

int trh_group_present_bgra8(trh_group* g, int tonemap, float exposure, uint32_t* bgra_host) {
    if (!g) return fail(TRH_ERR_INVALID_ARG, "trh_group_present_bgra8: group is NULL");
    if (!g->has_scene) return fail(TRH_ERR_STATE, "trh_group_present_bgra8: no scene");
    if (g->torn) return fail(TRH_ERR_STATE, "trh_group_present_bgra8: an earlier render call failed part-way; call trh_group_reset_accumulation first");
    if (!g->exchange) return trh_present_bgra8(g->ctx[0], tonemap, exposure, bgra_host);
    if (int rc = reduce_to_root(g)) return rc;
    if (int rc = trh_internal::present_from(g->ctx[0], g->reduced, tonemap, exposure, bgra_host)) return rc;
    for (size_t i = 1; i < g->ctx.size(); ++i)
        if (int rc = trh_synchronize(g->ctx[i])) return rc;
    return TRH_OK;
}

int trh_group_get_stats(trh_group* g, trh_stats* out, int reset) {
    if (!g || !out) return fail(TRH_ERR_INVALID_ARG, "trh_group_get_stats: NULL argument");
    memset(out, 0, sizeof(*out));
    for (trh_context* c : g->ctx) {
        trh_stats s{};
        if (int rc = trh_get_stats(c, &s, reset)) return rc;
        out->rays += s.rays;
        out->samples += s.samples;
        out->box_tests += s.box_tests;
        out->tri_tests += s.tri_tests;
        out->hits += s.hits;
        out->tex_fetches += s.tex_fetches;
        out->kernel_ms = std::max(out->kernel_ms, s.kernel_ms);   // the devices run side by side: the slowest one counts
        out->launches = std::max(out->launches, s.launches);
    }
    return TRH_OK;
}

}  // extern "C"
