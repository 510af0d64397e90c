/*
 * tracy_hip.h -- C ABI of the MI355X (gfx950) render module for Tracy ("-kernel hip").
 *
 * This is the drop-in boundary of the per-sample path-tracing hot path. A host-side
 * `HipTrace : TracyModule` (reference interface: src/module.h:17-54) flattens the
 * reference `Scene` into the POD arrays below and drives the GPU through these entry
 * points; see INTEGRATION.md for the binding a Tracy maintainer would add.
 *
 * Conventions: plain C, plain pointers and sizes, no C++/torch types. Every function
 * returns 0 on success and a non-zero trh_status on error; trh_last_error() returns a
 * thread-local message. Input buffers are copied during the call (caller keeps ownership).
 * There is NO CPU fallback: without a HIP device every compute entry point fails with
 * TRH_ERR_NO_DEVICE.
 *
 * Citations are into the reference tree (carcass82/tracy).
 */
#ifndef TRACY_HIP_H
#define TRACY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRH_ABI_VERSION 2

typedef enum trh_status {
    TRH_OK = 0,
    TRH_ERR_INVALID_ARG = 1,
    TRH_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime error at init */
    TRH_ERR_HIP = 3,         /* a HIP runtime call failed */
    TRH_ERR_STATE = 4,       /* call order violated (e.g. render before scene upload) */
    TRH_ERR_LIMIT = 5        /* size beyond a documented limit */
} trh_status;

#define TRH_NO_TEXTURE 0xffffffffu

/* Texture slots, same order as Material::TextureID (src/material.h:17). */
enum { TRH_TEX_BASECOLOR = 0, TRH_TEX_NORMAL = 1, TRH_TEX_ROUGHNESS = 2, TRH_TEX_METALNESS = 3, TRH_TEX_EMISSIVE = 4, TRH_TEX_COUNT = 5 };

/* POD mirror of Material's data members (src/material.h:103-117).
 * tex[i] == TRH_NO_TEXTURE  <=>  texture_flag_[i] == false. */
typedef struct trh_material {
    float albedo[3];
    float roughness;
    float metalness;
    float ior;
    float emissive[3];     /* already emissive_scale * albedo (src/material.h:24) */
    float translucent;
    uint32_t tex[TRH_TEX_COUNT];
} trh_material;

/* RGBA32F texel array exactly as Texture::pixels_ (src/texture.h:90,93-108):
 * row 0 first, already divided by 255 and sRGB->linear converted when requested. */
typedef struct trh_texture {
    uint32_t width;
    uint32_t height;
    const float* rgba;     /* width*height*4 floats */
} trh_texture;

/* One Mesh (src/mesh.h:106-112). Indices are mesh-local (0..vertex_count-1). */
typedef struct trh_mesh {
    uint32_t first_vertex;
    uint32_t vertex_count;
    uint32_t first_index;
    uint32_t index_count;  /* 3 * triangle count */
    uint32_t material;
    float aabb_min[3];
    float aabb_max[3];
} trh_mesh;

/* Vertex AoS of the reference (src/vertex.h:36-40): 14 floats =
 * pos[3] normal[3] uv0[2] tangent[3] bitangent[3]. */
#define TRH_VERTEX_FLOATS 14

#define TRH_FLAG_RUSSIAN_ROULETTE 1u   /* USE_RUSSIAN_ROULETTE (CMakeLists.txt:113)  */
#define TRH_FLAG_DEFAULT (TRH_FLAG_RUSSIAN_ROULETTE)

typedef struct trh_scene_desc {
    const float* vertices;       /* vertex_count * TRH_VERTEX_FLOATS */
    uint32_t vertex_count;
    const uint32_t* indices;     /* index_count, mesh-local */
    uint32_t index_count;
    const trh_mesh* meshes;
    uint32_t mesh_count;         /* <= 256: mesh id is 8 bit in the packed hit id (cpu_details.h:34) */
    const trh_material* materials;  /* materials[0] is the sky (src/scene.h:21) */
    uint32_t material_count;
    const trh_texture* textures;
    uint32_t texture_count;
    uint32_t width;
    uint32_t height;
    uint32_t max_bounces;        /* TRACY_MAX_BOUNCES, default 5 (CMakeLists.txt:110) */
    uint32_t flags;              /* TRH_FLAG_* */
} trh_scene_desc;

/* Which kernel pipeline trh_render uses. All produce the same per-sample paths. */
typedef enum trh_pipeline {
    TRH_PIPELINE_MEGAKERNEL = 0, /* default: persistent waves; every wave runs camera / traverse / shade phases over its own
                                    ring of paths (camera rays as wave packets, bounce rays by the lane scheduler) */
    TRH_PIPELINE_WAVEFRONT = 1,  /* generate / extend / shade queues in HBM, one kernel per stage, wave64 ballot compaction */
    TRH_PIPELINE_MEGAKERNEL_SIMPLE = 2, /* first-generation kernel (lane == pixel), kept for A/B */
    TRH_PIPELINE_MEGAKERNEL_LANES = 3   /* round 2's default: every lane runs its own path from camera ray to last bounce
                                           (needs no ring memory: the kernel of trh_render_noalloc and of trh_render_stats) */
} trh_pipeline;

typedef struct trh_context trh_context;   /* opaque; one per HIP device */

typedef struct trh_stats {
    uint64_t rays;          /* bounce-loop iterations, the reference's ++raycount_ (cpu_trace.cpp:115-116) */
    uint64_t samples;       /* camera paths */
    uint64_t box_tests;     /* child-AABB slab tests (only counted by trh_render_stats) */
    uint64_t tri_tests;     /* Moller-Trumbore tests  (only counted by trh_render_stats) */
    uint64_t hits;          /* bounce iterations that hit geometry (trh_render_stats) */
    uint64_t tex_fetches;   /* Texture::GetPixel calls (trh_render_stats) */
    double   kernel_ms;     /* sum of HIP-event durations of render launches since last reset */
    uint64_t launches;      /* number of trh_render calls timed in kernel_ms */
} trh_stats;

const char* trh_last_error(void);
int trh_abi_version(void);
int trh_device_count(int* count_out);

/* Startup, device half. Replaces the device/context part of TracyModule::Startup
 * (src/module.h:35; CUDA analogue CUDATraceKernel::Setup, cuda_trace.cu:227-260). */
int trh_init(int device, trh_context** ctx_out);

/* Shutdown (src/module.h:38; cuda_trace.cu:311-326). Frees all device memory. */
int trh_shutdown(trh_context* ctx);

/* Scene preprocessing + upload, once. Replaces CPUDetails::ProcessScene
 * (cpu_details.cpp:26-86: acceleration-structure build) and CUDATraceKernel::SetupScene
 * (cuda_trace.cu:262-309: deep copy to the device). Builds the module's own BVH on the
 * host, lays nodes/triangles out in 64 B-aligned arrays and copies them to HBM. */
int trh_scene_upload(trh_context* ctx, const trh_scene_desc* scene);

/* Which builder trh_scene_upload uses for the acceleration structure (call before the upload):
 * the host-side binned-SAH build (default: best traversal speed, tens of ms per 100k triangles on the
 * host cores) or the device-side linear BVH (Morton sort + Karras hierarchy on the GPU: milliseconds
 * for millions of triangles, but 10-35 % slower to traverse). Both emit the same record formats and leaf
 * links, so every kernel runs unchanged. Scenes of < 9 triangles, and trees the linear build makes
 * deeper than the traversal stack (32), are built by the host builder whatever the setting;
 * trh_bvh_info.builder tells which one built the resident structure. No counterpart in the
 * reference (its kd-tree is always built on the host, kdtree.h:160-354). */
typedef enum trh_builder { TRH_BUILDER_HOST_SAH = 0, TRH_BUILDER_DEVICE_LBVH = 1 } trh_builder;
int trh_set_builder(trh_context* ctx, int builder /* trh_builder */);

/* Camera (re)upload. Replaces CUDATraceKernel::UpdateCamera (cuda_trace.cu:221-225).
 * inv_view_proj is column-major, = inverse(projection * view) (src/camera.h:45,54).
 * Does NOT reset the accumulation (call trh_reset_accumulation for a camera cut). */
int trh_set_camera(trh_context* ctx, const float eye[3], const float inv_view_proj[16]);

/* TracyEvent::eCameraCut handler (cpu_trace.cpp:72-82; cpu_details.h:90). */
int trh_reset_accumulation(trh_context* ctx);

int trh_set_pipeline(trh_context* ctx, int pipeline /* trh_pipeline */);

/* Debug views: the run-time form of the reference's compile-time DEBUG_SHOW_* switches
 * (CMakeLists.txt:23-35; cpu_trace.cpp:127-137, cuda_trace.cu:94-103). With a view selected every
 * camera path ends at its first hit and returns one material input instead of radiance: base colour,
 * .5 * normalize(1 + mat3(view) * normal), metalness, roughness or emissive (textures included); a
 * path that misses returns the sky as usual. view3x3 = the upper-left 3x3 of Camera::GetView
 * (src/camera.h:57), column-major, 9 floats, used by the NORMALS view only; NULL keeps the last one
 * (identity at start). Takes effect for the next trh_render / trh_sample_radiance; does not reset the
 * accumulation. */
typedef enum trh_debug_view {
    TRH_VIEW_NONE = 0,
    TRH_VIEW_BASECOLOR = 1,
    TRH_VIEW_NORMALS = 2,
    TRH_VIEW_METALNESS = 3,
    TRH_VIEW_ROUGHNESS = 4,
    TRH_VIEW_EMISSIVE = 5
} trh_debug_view;
int trh_set_debug_view(trh_context* ctx, int view /* trh_debug_view */, const float view3x3[9]);

/* THE HOT CALL. Replaces CpuTrace::OnUpdate + Trace (cpu_trace.cpp:33-170): traces
 * samples [spp_begin, spp_begin+spp_count) of every pixel (one camera path each) and adds
 * their radiance into the context's float4 accumulation buffer (rgb sum, .w = count).
 * RNG state of sample s of pixel p is derived from (p, s, seed), then follows the
 * reference's PCG fastrand sequence (src/random.h:66-77). Asynchronous on the context's
 * stream; trh_synchronize / trh_resolve / trh_get_stats wait for it.
 * Scheduling only (never the samples traced): a call of more than 65536 spp is split into several
 * launches; every launch records what each screen tile cost and the next launch of the same tiling
 * starts with the expensive tiles; the first launch of >= 16 spp after a scene upload or a change of
 * the sample count traces its first sample as a launch of its own to get those costs. The order of
 * the float additions into a pixel therefore depends on the history of calls -- the same sequence of
 * calls gives the same bits, different sequences agree to rounding. */
int trh_render(trh_context* ctx, uint32_t spp_begin, uint32_t spp_count, uint64_t seed);

/* Same as trh_render but runs the instrumented kernel build that also counts box tests,
 * triangle tests, hits and texture fetches (the roofline numerator). Slower. */
int trh_render_stats(trh_context* ctx, uint32_t spp_begin, uint32_t spp_count, uint64_t seed);

/* trh_render for callers that cannot allocate. NO COUNTERPART IN THE REFERENCE: it exists for hosts that
 * may not call hipMalloc: tight serving loops, harnesses that forbid it, code that will run under HIP graph capture
 * (capture itself is NOT exercised by the tests: the call also queries occupancy and sets the device, which a capturing
 * host should check against its ROCm version). Same samples, same
 * kernels and the same launch sequence as trh_render with TRH_PIPELINE_MEGAKERNEL_LANES on a context that has no tile
 * order yet (every call of >= 16 spp traces its first sample as a launch of its own): bit-equal to that; equal to the
 * default pipeline up to the order of a pixel's float additions (the default pipeline's ring of paths is hipMalloc'ed
 * memory that grows with the launch, so it is not used here). The launch scratch -- the tile counter the
 * persistent waves pull from, and the cost / order / key arrays and the sort's temporary storage of the
 * cost-ordered tile hand-out -- lives in a file-scope __device__ array of 9.25 MiB that every GPU the library
 * runs on holds until the process ends, instead of in hipMalloc'ed memory. The call performs no hipMalloc,
 * hipFree, event creation or other allocating routine and records no launch timing (trh_stats.kernel_ms /
 * launches do not count it; rays and samples do). Limits: TRH_PIPELINE_MEGAKERNEL or _LANES set (TRH_ERR_STATE otherwise),
 * <= 65536 spp per call, and at most 262144 tiles (1080p and 4K frames at >= 8 spp per call fit; TRH_ERR_LIMIT
 * names the tile count otherwise). All contexts on one GPU share that GPU's array: calls that overlap on one
 * GPU (two contexts, two threads) race; calls on different GPUs do not. */
int trh_render_noalloc(trh_context* ctx, uint32_t spp_begin, uint32_t spp_count, uint64_t seed);

int trh_synchronize(trh_context* ctx);

/* Device pointer of the accumulation buffer (width*height float4, rgb sums + count) so
 * that a multi-GPU host can reduce it in place (RCCL / torch.distributed) before resolve. */
int trh_accum_device_ptr(trh_context* ctx, void** dptr_out, size_t* nbytes_out);

/* Makes the context accumulate into caller-owned device memory (width*height float4, 16-byte
 * aligned, on the context's device) instead of its internal buffer, e.g. a torch tensor that
 * torch.distributed / RCCL then reduces in place. The buffer is NOT cleared by this call.
 * dptr == NULL re-binds the internal buffer. No counterpart in the reference (single device). */
int trh_bind_accum_buffer(trh_context* ctx, void* dptr, size_t nbytes);

/* Replaces CPUDetails::UpdateOutput's running mean (cpu_details.cpp:187-199) for the whole
 * frame: mean = sum / count, on the device, into an internal float4 image (row 0 = bottom
 * row, like the reference's output[]). If rgba_host != NULL the image is copied to the host
 * (width*height*4 floats, .w = sample count). samples_out receives the samples per pixel
 * accumulated since the last reset (as counted by this context). */
int trh_resolve(trh_context* ctx, float* rgba_host, uint64_t* samples_out);

/* Display-side post, replaces CPUDetails::Tonemap + UpdateBitmap + Bitmap::SetPixel
 * (cpu_details.cpp:201-243, bitmap.cpp:58-69) for the whole frame on the device:
 * mean = sum / count, colour * exposure, tone curve, srgb(), clamp(255.99 * c, 0, 255), u8
 * truncation, packed as 0x00RRGGBB (bytes B,G,R,0 -- the XImage/DIB layout of bitmap.cpp:28-39).
 * The no-tonemap variant leaves the exposure out, like the reference's #else branch (cpu_details.cpp:236-238).
 * Output row 0 is the TOP row of the picture (the reference flips in SetPixel; it writes row
 * `height - y`, one past the end for y = 0 -- this call uses the evident intent height-1-y).
 * tonemap selects the reference's compile-time variants (CMakeLists.txt:92-103):
 * sRGB (default build), Reinhard, ACES (curves as spelled out in cuda_details.cpp:36-41), or none.
 * bgra_host: width*height uint32, may be NULL (the device image is still produced). */
typedef enum trh_tonemap {
    TRH_TONEMAP_NONE = 0,
    TRH_TONEMAP_SRGB = 1,
    TRH_TONEMAP_REINHARD = 2,
    TRH_TONEMAP_ACES = 3
} trh_tonemap;
int trh_present_bgra8(trh_context* ctx, int tonemap /* trh_tonemap */, float exposure, uint32_t* bgra_host);

/* Parity probe (a): nearest hit of the un-jittered primary ray through each pixel centre
 * ((x+.5)/W, (y+.5)/H). ids[p] = (mesh << 24) | triangle (cpu_details.h:34), 0xffffffff on
 * miss. t and uv (2 floats per pixel: Moller-Trumbore u,v) are optional (may be NULL). */
int trh_primary_hits(trh_context* ctx, uint32_t* ids_host, float* t_host, float* uv_host);

/* Parity probe (b): radiance of ONE sample index for every pixel (no accumulation),
 * width*height*4 floats (.w = number of bounce iterations of that path). */
int trh_sample_radiance(trh_context* ctx, uint32_t sample_index, uint64_t seed, float* rgba_host);

/* Ray / sample counters. Replaces TracyModule::GetRayCount (src/module.h:53) with 64-bit
 * counters (the reference's u32 wraps at GPU rates). */
int trh_get_stats(trh_context* ctx, trh_stats* out, int reset);

/* Diagnostics of the instrumented kernel (trh_render_stats), since the last stats reset:
 * out[0..4] = wave-level inner-node iterations, wave-level triangle-loop iterations, wave-level
 * shade calls, wave-level generate calls, lane-level generate calls; out[5..9] = shader-clock cycles
 * the waves spent in inner steps, triangle steps, shade, refill/generate and in the kernel altogether;
 * out[10] = lifetime of the longest-lived wave, out[11] = number of waves, out[12..15] = earliest wave start,
 * latest wave end, sum of the wave starts, sum of the wave ends (device-wide 100 MHz clock), out[16..18] = lanes
 * standing on a leaf / waiting for shade / without a path, summed over the inner-step votes. Lane utilisation of a
 * phase = lane-level count / (64 * wave-level count), e.g. (box_tests/2) / (64*out[0]). */
int trh_get_debug_counters(trh_context* ctx, uint64_t* out, uint32_t n);

/* Info about the uploaded acceleration structure (for DESIGN/bench reporting). */
typedef struct trh_bvh_info {
    uint32_t node_count;
    uint32_t tri_count;       /* triangles stored (after dropping nothing: == scene triangles) */
    uint32_t max_depth;
    uint32_t max_leaf_size;
    uint64_t node_bytes;
    uint64_t tri_bytes;
    double   sah_cost;
    double   build_ms;
    double   fp16_inflation;  /* box-area growth of the 32-byte fp16 records; node_bytes / node_count tells which format is in use */
    uint32_t builder;         /* trh_builder that produced it */
    uint32_t from_cache;      /* 1: read from the on-disk cache (trh_set_bvh_cache_dir) instead of built */
    double   device_build_ms; /* device LBVH only: HIP-event time of its kernels + sort (build_ms is the wall time incl. the upload) */
} trh_bvh_info;
int trh_get_bvh_info(trh_context* ctx, trh_bvh_info* out);

/* Host-only: builds the acceleration structure of a scene description exactly as trh_scene_upload
 * does (same validation, same builder, no device needed), runs the builder's structural self-check
 * (every triangle in exactly one leaf, leaf boxes contain their triangles, every fp16 box contains
 * its fp32 box, limits respected) and returns the number of violations (0 = sound; trh_last_error()
 * names the first) plus an FNV-1a hash of the record arrays: the layout must not depend on the
 * number of builder threads (TRH_BUILD_THREADS). No counterpart in the reference (its kd-tree build,
 * kdtree.h:160-354, is not what the kernels traverse). */
int trh_bvh_build_host(const trh_scene_desc* scene, trh_bvh_info* info_out, uint64_t* hash_out, uint64_t* violations_out);

/* On-disk cache of host-built acceleration structures, process-wide: with a directory set, trh_scene_upload
 * and trh_bvh_build_host look there for a file named after a hash of the scene's triangles and use it instead
 * of building (after checking its header, sizes, checksum and the structural self-check above); on a miss they
 * build and write the file. NULL or "" switches the cache off (the default; the environment variable
 * TRH_BVH_CACHE sets the initial directory). The device-side builder does not use it. No counterpart in the
 * reference (it rebuilds its kd-tree at every start, cpu_details.cpp:26-86). */
int trh_set_bvh_cache_dir(const char* directory);

/* The same check for the device-side builder: builds the linear BVH of the scene on ctx's device, copies
 * the arrays back and runs the same structural self-check and hash on them. Leaves the context's resident
 * scene alone. Fails with TRH_ERR_LIMIT where trh_scene_upload would fall back to the host builder. */
int trh_bvh_build_device(trh_context* ctx, const trh_scene_desc* scene, trh_bvh_info* info_out, uint64_t* hash_out, uint64_t* violations_out);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one host thread drives the N GPUs of a node.
 *
 * The reference is one process with one host thread that calls Startup once and then
 * OnEvent* / OnUpdate / OnRender per frame (src/win_raytracer.cpp:480, 494-517); it has no multi-device
 * path of any kind. A group is that same call protocol over N devices: device g renders the sample indices
 * [g*S/N, (g+1)*S/N) of every trh_group_render call of S samples (the scene is replicated, paths are
 * independent), each into its own float4 sum buffer, and trh_group_resolve / trh_group_present_bgra8 run
 * the path's one exchange step -- ncclReduce(sum, float, width*height*4, root = device 0) over xGMI inside
 * ncclGroupStart/End, enqueued on the devices' render streams -- before the mean / tonemap on device 0.
 * The per-device sums are left as they are, so rendering goes on after a resolve. The sample set of a call
 * does not depend on N (the RNG is keyed by pixel, global sample index and seed); the image agrees with
 * the one-device image to float summation order.
 * A group of one device uses no RCCL and is the plain context, bit for bit (TRH_GROUP_FORCE_RCCL=1 runs the
 * reduce anyway, for testing). RCCL (librccl.so.1) is loaded when the first group of >= 2 devices is made.
 * Each function below is the trh_* call of the same name applied to every device; errors as there. */
typedef struct trh_group trh_group;

/* Startup, device half, for min(ngpus_requested, devices present) GPUs, devices 0..n-1 (ngpus_requested <= 0:
 * all of them). Extends trh_init / TracyModule::Startup (src/module.h:35). */
int trh_group_init(int ngpus_requested, int* ngpus_out, trh_group** group_out);
int trh_group_shutdown(trh_group* group);                                   /* TracyModule::Shutdown (src/module.h:38) */
int trh_group_size(const trh_group* group);
/* the single-device context of member `index` (0 = root), e.g. for trh_get_bvh_info or the parity probes */
int trh_group_context(trh_group* group, int index, trh_context** ctx_out);
int trh_group_set_builder(trh_group* group, int builder /* trh_builder */);
/* trh_scene_upload on every device; the host-side BVH build runs once and its arrays are copied to each GPU */
int trh_group_scene_upload(trh_group* group, const trh_scene_desc* scene);
int trh_group_set_camera(trh_group* group, const float eye[3], const float inv_view_proj[16]);
int trh_group_set_debug_view(trh_group* group, int view /* trh_debug_view */, const float view3x3[9]);
int trh_group_reset_accumulation(trh_group* group);                         /* eCameraCut (cpu_trace.cpp:72-82) */
/* THE HOT CALL on N devices (CpuTrace::OnUpdate, cpu_trace.cpp:33-70). Asynchronous: enqueues one trh_render
 * per device and returns. A call of fewer samples than devices leaves some devices idle. */
int trh_group_render(trh_group* group, uint32_t spp_begin, uint32_t spp_count, uint64_t seed);
int trh_group_synchronize(trh_group* group);
/* exchange step + trh_resolve on device 0; samples_out = samples per pixel over all devices since the last reset */
int trh_group_resolve(trh_group* group, float* rgba_host, uint64_t* samples_out);
/* exchange step + trh_present_bgra8 on device 0 (TracyModule::OnRender, cpu_details.cpp:201-243) */
int trh_group_present_bgra8(trh_group* group, int tonemap /* trh_tonemap */, float exposure, uint32_t* bgra_host);
/* counters summed over the devices; kernel_ms / launches are those of the slowest device (they run side by side) */
int trh_group_get_stats(trh_group* group, trh_stats* out, int reset);
/* the sample split used by trh_group_render (pure host arithmetic, no device needed): the range of device g of ngpus */
void trh_split_samples(uint32_t spp_begin, uint32_t spp_count, int ngpus, int g, uint32_t* begin_out, uint32_t* count_out);

#ifdef __cplusplus
}
#endif
#endif /* TRACY_HIP_H */
