// host_scene.h -- C++ face of the standalone scene loader (include/tracy_scene.h) for hosts that
// run outside the Tracy tree (the headless CLI). Inside Tracy the reference's own Scene is used.
#pragma once
#include <cstring>
#include <string>

#include "../../../include/tracy_scene.h"
#include "hip_trace.h"

namespace tracy_hip {

class HostScene {
public:
    HostScene() = default;
    ~HostScene() { if (h_) tsc_free(h_); }
    HostScene(const HostScene&) = delete;
    HostScene& operator=(const HostScene&) = delete;

    // Scene::Init (src/scene.cpp:231); width/height > 0 override the OUT line before CAM is read
    bool Init(const char* scene_path, const char* asset_root, uint32_t& inout_width, uint32_t& inout_height, bool override_size) {
        h_ = tsc_load(scene_path, asset_root, override_size ? (int)inout_width : 0, override_size ? (int)inout_height : 0);
        if (!h_) { error_ = tsc_last_error(); return false; }
        // stbi_load / stbi_loadf of Scene::Init (src/scene.cpp:23-48): built-in JPEG and Radiance-HDR decoders;
        // data/sky.hdr is not shipped with the reference (.MISSING_LARGE_BLOBS): documented synthetic probe
        if (tsc_decode_textures(h_, 1) != TRH_OK) { error_ = tsc_last_error(); return false; }
        if (tsc_fill_desc(h_, &desc_) != TRH_OK) { error_ = tsc_last_error(); return false; }
        inout_width = desc_.width;
        inout_height = desc_.height;
        return true;
    }
    uint32_t GetWidth() const { return desc_.width; }
    uint32_t GetHeight() const { return desc_.height; }
    uint32_t GetObjectCount() const { return desc_.mesh_count; }
    uint32_t GetTriCount() const { return tsc_triangle_count(h_); }
    std::string GetName() const { return tsc_name(h_); }
    const trh_scene_desc& GetDesc() const { return desc_; }
    bool GetCamera(tsc_camera& c) const { return tsc_get_camera(h_, &c) == TRH_OK; }
    bool SetView(const float eye[3], const float center[3], const float up[3]) { return tsc_set_view(h_, eye, center, up) == TRH_OK; }
    const std::string& GetError() const { return error_; }

private:
    tsc_scene* h_ = nullptr;
    trh_scene_desc desc_{};
    std::string error_;
};

// HostScene already owns a trh_scene_desc: these overloads replace the generic accessor walk
inline bool FlattenCamera(const HostScene& scene, FlatScene& out) {
    tsc_camera c{};
    if (!scene.GetCamera(c)) return false;
    memcpy(out.eye, c.eye, sizeof(out.eye));
    memcpy(out.inv_view_proj, c.inv_view_proj, sizeof(out.inv_view_proj));
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) out.view3[col * 3 + row] = c.view[col * 4 + row];
    return true;
}
inline bool FlattenScene(const HostScene& scene, FlatScene& out) {
    out = FlatScene{};
    out.desc = scene.GetDesc();   // pointers stay owned by the HostScene, which outlives the module
    return FlattenCamera(scene, out);
}

using HipTrace = HipTraceT<HostScene>;

}  // namespace tracy_hip
