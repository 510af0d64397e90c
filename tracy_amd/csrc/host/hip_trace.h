// hip_trace.h -- HipTrace: the `-kernel hip` render module, host side (C++, no HIP headers).
//
// A thin class over the C ABI of include/tracy_hip.h with the life cycle of the reference's
// modules: CpuTrace (src/kernels/raytracing/software/cpu_trace.{h,cpp}) and CUDATrace
// (src/kernels/raytracing/cuda/cuda_trace.cpp:28-62). It is a template on the scene type so that
// the same code serves
//   * the reference's `Scene` (src/scene.h): the generic FlattenScene below reads it through the
//     accessors GetObjectCount/GetObject/GetMaterials/GetTextures/GetCamera (scene.h:36-60,
//     mesh.h:83-103, material.h:54-101, texture.h:59-82, camera.h:57-67), and
//   * this repository's standalone host scene (host_scene.h), which already owns a trh_scene_desc.
// Error behaviour follows the reference: Startup returns false (main then logs "Kernel failed to
// initialize", win_raytracer.cpp:572-575); run-time device errors are logged and abort, like
// CUDAAssert (cuda_log.h:19-32). There is no CPU fallback.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#if __has_include("tracy_hip.h")
#include "tracy_hip.h"                      // on the include path (inside Tracy: ext/tracy_hip/include, integration/tracy_hip.cmake)
#else
#include "../../../include/tracy_hip.h"     // this repository's layout
#endif
#include "tracy_module.h"

namespace tracy_hip {

// Owns the POD arrays a trh_scene_desc points to.
struct FlatScene {
    std::vector<float> vertices;
    std::vector<uint32_t> indices;
    std::vector<trh_mesh> meshes;
    std::vector<trh_material> materials;
    std::vector<trh_texture> textures;
    trh_scene_desc desc{};
    float eye[3]{};
    float inv_view_proj[16]{};
    float view3[9]{};   // mat3(Camera::GetView()), column-major
};

// 4x4 column-major helpers (Camera keeps inverse(projection * view) private, camera.h:84)
inline void Mul4x4(const float* a, const float* b, float* r) {
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) r[c * 4 + k] = a[0 * 4 + k] * b[c * 4 + 0] + a[1 * 4 + k] * b[c * 4 + 1] + a[2 * 4 + k] * b[c * 4 + 2] + a[3 * 4 + k] * b[c * 4 + 3];
}
inline bool Invert4x4(const float* m, float* out) {
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return false;
    for (int i = 0; i < 16; ++i) out[i] = static_cast<float>(inv[i] / det);
    return true;
}

template <class SceneT>
bool FlattenCamera(const SceneT& scene, FlatScene& out);

// Generic flattening of a scene that offers the reference's accessors (src/scene.h:36-60).
// `Vertex` there is 14 packed floats (src/vertex.h:36-40) == TRH_VERTEX_FLOATS.
template <class SceneT>
bool FlattenScene(const SceneT& scene, FlatScene& out) {
    out = FlatScene{};
    for (uint32_t i = 0; i < scene.GetObjectCount(); ++i) {
        const auto& mesh = scene.GetObject(i);
        static_assert(sizeof(*mesh.GetVertices()) == TRH_VERTEX_FLOATS * sizeof(float), "Vertex must be 14 packed floats");
        trh_mesh m{};
        m.first_vertex = static_cast<uint32_t>(out.vertices.size() / TRH_VERTEX_FLOATS);
        m.vertex_count = mesh.GetVertexCount();
        m.first_index = static_cast<uint32_t>(out.indices.size());
        m.index_count = mesh.GetIndexCount();
        m.material = mesh.GetMaterial();
        const auto& bb = mesh.GetAABB();
        m.aabb_min[0] = bb.minbound.x; m.aabb_min[1] = bb.minbound.y; m.aabb_min[2] = bb.minbound.z;
        m.aabb_max[0] = bb.maxbound.x; m.aabb_max[1] = bb.maxbound.y; m.aabb_max[2] = bb.maxbound.z;
        const float* v = reinterpret_cast<const float*>(mesh.GetVertices());
        out.vertices.insert(out.vertices.end(), v, v + static_cast<size_t>(m.vertex_count) * TRH_VERTEX_FLOATS);
        out.indices.insert(out.indices.end(), mesh.GetIndices(), mesh.GetIndices() + m.index_count);
        out.meshes.push_back(m);
    }
    for (const auto& src : scene.GetMaterials()) {
        trh_material m{};
        const auto& al = src.GetAlbedo();
        const auto& em = src.GetEmissive();
        m.albedo[0] = al.x; m.albedo[1] = al.y; m.albedo[2] = al.z;
        m.roughness = src.GetRoughness();
        m.metalness = src.GetMetalness();
        m.ior = src.GetRefractiveIndex();
        m.emissive[0] = em.x; m.emissive[1] = em.y; m.emissive[2] = em.z;
        m.translucent = src.GetTranslucent();
        using MatT = typename std::decay<decltype(src)>::type;
        for (int k = 0; k < TRH_TEX_COUNT; ++k) m.tex[k] = src.GetTexture(static_cast<typename MatT::TextureID>(k));  // UINT32_MAX when unset
        out.materials.push_back(m);
    }
    for (const auto& t : scene.GetTextures()) {
        trh_texture d{};
        d.width = t.GetWidth();
        d.height = t.GetHeight();
        d.rgba = reinterpret_cast<const float*>(t.GetPixels());   // vec4* RGBA32F (texture.h:74-77)
        out.textures.push_back(d);
    }
    trh_scene_desc& d = out.desc;
    d.vertices = out.vertices.data();
    d.vertex_count = static_cast<uint32_t>(out.vertices.size() / TRH_VERTEX_FLOATS);
    d.indices = out.indices.data();
    d.index_count = static_cast<uint32_t>(out.indices.size());
    d.meshes = out.meshes.data();
    d.mesh_count = static_cast<uint32_t>(out.meshes.size());
    d.materials = out.materials.data();
    d.material_count = static_cast<uint32_t>(out.materials.size());
    d.textures = out.textures.data();
    d.texture_count = static_cast<uint32_t>(out.textures.size());
    d.width = scene.GetWidth();
    d.height = scene.GetHeight();
    d.max_bounces = 5;              // TRACY_MAX_BOUNCES (CMakeLists.txt:110)
    d.flags = TRH_FLAG_DEFAULT;     // TRACY_RUSSIAN_ROULETTE ON (CMakeLists.txt:113)
    return FlattenCamera(scene, out);
}

template <class SceneT>
bool FlattenCamera(const SceneT& scene, FlatScene& out) {
    const auto& cam = scene.GetCamera();
    const auto& e = cam.GetPosition();
    out.eye[0] = e.x; out.eye[1] = e.y; out.eye[2] = e.z;
    float pv[16];
    const float* view = reinterpret_cast<const float*>(&cam.GetView());
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) out.view3[c * 3 + r] = view[c * 4 + r];
    Mul4x4(reinterpret_cast<const float*>(&cam.GetProjection()), view, pv);
    return Invert4x4(pv, out.inv_view_proj);
}

#if defined(TRACY_HIP_USE_REFERENCE_MODULE_H)
// inside Tracy: the four types are the reference's own (integration/hip_module.h names them), no stand-ins exist
template <class SceneT, class Base, class WindowT, class EventT>
#else
template <class SceneT, class Base = TracyModuleT<SceneT>, class WindowT = WindowHandle, class EventT = TracyEvent>
#endif
class HipTraceT final : public Base {
public:
    // ngpus: GPUs of the node to use (devices 0..ngpus-1, <= 0 = all of them); one host thread drives them through
    // trh_group_* (one GPU: the group is the plain single-device context).
    // spp_per_update: samples traced per OnUpdate over all GPUs. The reference traces 1 (cpu_trace.cpp:55-65); at GPU
    // rates a 1-spp update is dominated by the launch's fixed cost (dragon 1080p: 0.77 ms at 1 spp = 2.7 Gsamples/s
    // against 11.5 Gsamples/s at 64 spp), so the default 0 means ADAPTIVE: one sample per GPU for the first update after
    // Startup / eCameraCut (the picture reacts at once), then doubling while an update stays under kTargetUpdateMs.
    explicit HipTraceT(int ngpus = 1, uint32_t spp_per_update = 0) : ngpus_requested_(ngpus), spp_per_update_(spp_per_update) {}

    bool Startup(const WindowT in_Window, const SceneT& in_Scene) override {
        (void)in_Window;
        if (trh_group_init(ngpus_requested_, &ngpus_, &group_) != TRH_OK) return Fail("trh_group_init");
        if (trh_group_set_builder(group_, builder_) != TRH_OK) return Fail("trh_group_set_builder");
        FlatScene flat;
        if (!FlattenScene(in_Scene, flat)) return Fail("FlattenScene");
        if (trh_group_scene_upload(group_, &flat.desc) != TRH_OK) return Fail("trh_group_scene_upload");
        if (trh_group_set_camera(group_, flat.eye, flat.inv_view_proj) != TRH_OK) return Fail("trh_group_set_camera");
        if (trh_group_set_debug_view(group_, debug_view_, flat.view3) != TRH_OK) return Fail("trh_group_set_debug_view");
        width_ = flat.desc.width;
        height_ = flat.desc.height;
        image_.assign(static_cast<size_t>(width_) * height_ * 4, 0.f);
        bitmap_.assign(static_cast<size_t>(width_) * height_, 0u);
        next_sample_ = 0;
        auto_spp_ = static_cast<uint32_t>(ngpus_);
        return true;
    }

    void Shutdown() override {
        if (group_) trh_group_shutdown(group_);
        group_ = nullptr;
    }

    void OnUpdate(const SceneT& in_Scene, float in_DeltaTime) override {
        (void)in_DeltaTime;
        if (camera_dirty_) {   // pattern of CUDADetails::Update (cuda_details.cpp:206-215)
            FlatScene cam;
            if (FlattenCamera(in_Scene, cam)) {
                Check(trh_group_set_camera(group_, cam.eye, cam.inv_view_proj), "trh_group_set_camera");
                Check(trh_group_set_debug_view(group_, debug_view_, cam.view3), "trh_group_set_debug_view");
            }
            Check(trh_group_reset_accumulation(group_), "trh_group_reset_accumulation");
            next_sample_ = 0;
            auto_spp_ = static_cast<uint32_t>(ngpus_);   // react at once, then ramp up again
            camera_dirty_ = false;
        }
        uint32_t spp = spp_per_update_ ? spp_per_update_ : auto_spp_;
        // sample indices are 32 bits (trh_group_render refuses beyond): a static camera left alone for hours reaches the end of
        // them at GPU rates (4096 spp per update). The last update is clamped and the picture then simply stays as it is.
        if (spp > 0xffffffffu - next_sample_) spp = 0xffffffffu - next_sample_;
        if (spp == 0) return;
        const auto t0 = std::chrono::steady_clock::now();
        Check(trh_group_render(group_, next_sample_, spp, kSeed), "trh_group_render");
        next_sample_ += spp;
        if (synchronous_update_) {
            // main() times this call (win_raytracer.cpp:511-515) and derives MRays/s from it: return when the pass is done
            Check(trh_group_synchronize(group_), "trh_group_synchronize");
            Adapt(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), spp);
        } else {
            // the launches stay in flight; OnRender / GetRayCount wait for them. A host that uses this must time
            // OnUpdate + OnRender together (tracy_hip_cli does).
            if (unmeasured_spp_ == 0) unmeasured_since_ = t0;
            unmeasured_spp_ += spp;
        }
    }

    // The reference modules tonemap into their Bitmap and paint the window here
    // (cpu_details.cpp:201-248). The picture is produced on the device by trh_present_bgra8
    // (8 MB at 1080p instead of the 33 MB float image) and kept for the caller (GetBitmap); a
    // host with a display blits it (XPutImage / StretchDIBits of bitmap.cpp:84-110). With several GPUs this is
    // where the one exchange step runs (RCCL reduce to device 0).
    void OnRender(const WindowT in_Window) override {
        (void)in_Window;
        if (unmeasured_spp_) {
            // asynchronous updates: wait for them HERE and adapt on the time the samples took, not on samples + present (round 2
            // measured both together: the 8 MB picture copy counted as render time, the updates came out too small and the mode
            // was slower than the synchronous one)
            Check(trh_group_synchronize(group_), "trh_group_synchronize");
            Adapt(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - unmeasured_since_).count(), unmeasured_spp_);
            unmeasured_spp_ = 0;
        }
        Check(trh_group_present_bgra8(group_, tonemap_, exposure_, bitmap_.data()), "trh_group_present_bgra8");
    }

    void OnEvent(EventT in_Event, const WindowT in_Window, const SceneT& in_Scene) override {
        (void)in_Window;
        (void)in_Scene;
        if (in_Event == EventT::eCameraCut) camera_dirty_ = true;   // cpu_trace.cpp:76-78
    }

    const char* GetModuleName() const override { return "HIP Raytracing [MI355X gfx950]"; }

    // u32 in the interface (src/module.h:53); saturates instead of wrapping
    uint32_t GetRayCount(bool in_ShouldReset = false) override {
        const uint64_t rays = GetRayCount64(in_ShouldReset);
        return rays > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(rays);
    }
    uint64_t GetRayCount64(bool in_ShouldReset = false) {
        trh_stats st{};
        Check(trh_group_get_stats(group_, &st, in_ShouldReset ? 1 : 0), "trh_group_get_stats");
        return st.rays;
    }

    // linear mean image (width*height*4 floats, row 0 = bottom row, .w = sample count), resolved on demand
    const float* GetImage() {
        uint64_t n = 0;
        Check(trh_group_resolve(group_, image_.data(), &n), "trh_group_resolve");
        return image_.data();
    }
    // picture of the last OnRender: width*height 0x00RRGGBB, row 0 = top row
    const uint32_t* GetBitmap() const { return bitmap_.data(); }
    // the reference fixes these at compile time (TRACY_TONEMAPPING / TRACY_EXPOSURE, CMakeLists.txt:92-103)
    void SetTonemap(int in_Tonemap, float in_Exposure) { tonemap_ = in_Tonemap; exposure_ = in_Exposure; }
    // the reference's DEBUG_SHOW_* build switches (CMakeLists.txt:23-35) at run time; call before Startup or send eCameraCut
    void SetDebugView(int in_View) { debug_view_ = in_View; camera_dirty_ = group_ != nullptr; }
    // acceleration-structure builder of the next Startup: TRH_BUILDER_HOST_SAH (default) or TRH_BUILDER_DEVICE_LBVH
    void SetBuilder(int in_Builder) { builder_ = in_Builder; }
    // false: OnUpdate only enqueues (see OnUpdate); true (default): OnUpdate returns when its samples are traced
    void SetSynchronousUpdate(bool in_Synchronous) { synchronous_update_ = in_Synchronous; }
    uint32_t GetWidth() const { return width_; }
    uint32_t GetHeight() const { return height_; }
    uint32_t GetSampleCount() const { return next_sample_; }
    uint32_t GetSamplesPerUpdate() const { return spp_per_update_ ? spp_per_update_ : auto_spp_; }
    int GetGpuCount() const { return ngpus_; }
    trh_group* GetGroup() { return group_; }
    trh_context* GetContext() {   // the root device's context (parity probes, BVH info)
        trh_context* c = nullptr;
        if (group_) trh_group_context(group_, 0, &c);
        return c;
    }

private:
    static constexpr uint64_t kSeed = 0xABCDEFull;   // initrand() (src/random.h:66-69)
    static constexpr double kTargetUpdateMs = 12.0;  // adaptive mode: an update of about one 60-Hz frame
    static constexpr uint32_t kMaxAutoSpp = 4096;

    // adaptive samples per update: `spp` samples took `ms`; grow towards kTargetUpdateMs, at most doubling per step,
    // in multiples of the GPU count (every device gets the same share)
    void Adapt(double ms, uint32_t spp) {
        if (spp_per_update_ != 0 || spp == 0 || !(ms > 0.0)) return;
        const double want = kTargetUpdateMs * static_cast<double>(spp) / ms;
        const uint32_t n = static_cast<uint32_t>(ngpus_);
        uint32_t next = want >= static_cast<double>(kMaxAutoSpp) ? kMaxAutoSpp : static_cast<uint32_t>(want);
        if (next > auto_spp_ * 2u) next = auto_spp_ * 2u;
        next -= next % n;
        auto_spp_ = next < n ? n : next;
    }

    bool Fail(const char* what) {
        fprintf(stderr, "[HipTrace] %s failed: %s\n", what, trh_last_error());
        if (group_) trh_group_shutdown(group_);
        group_ = nullptr;
        return false;
    }
    void Check(int rc, const char* what) {
        if (rc != TRH_OK) {   // as CUDAAssert (cuda_log.h:19-32): log and stop
            fprintf(stderr, "[HipTrace] %s failed (%d): %s\n", what, rc, trh_last_error());
            exit(EXIT_FAILURE);
        }
    }

    int ngpus_requested_ = 1, ngpus_ = 0;
    uint32_t spp_per_update_ = 0, auto_spp_ = 1;
    bool synchronous_update_ = true;
    uint32_t unmeasured_spp_ = 0;
    std::chrono::steady_clock::time_point unmeasured_since_{};
    trh_group* group_ = nullptr;
    uint32_t width_ = 0, height_ = 0, next_sample_ = 0;
    bool camera_dirty_ = false;
    std::vector<float> image_;
    std::vector<uint32_t> bitmap_;
    int tonemap_ = TRH_TONEMAP_SRGB;   // default build: USE_TONEMAP_SRGB, TRACY_EXPOSURE 1
    float exposure_ = 1.f;
    int debug_view_ = TRH_VIEW_NONE;
    int builder_ = TRH_BUILDER_HOST_SAH;
};

}  // namespace tracy_hip
