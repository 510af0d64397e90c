// ctx_internal.h -- what tracy_group.hip needs from the single-device context of tracy_hip.hip.
// Internal to libtracy_hip.so; nothing here is part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../../include/tracy_hip.h"
#include "../host/bvh_builder.h"

namespace trh_internal {

int fail(int code, const std::string& msg);   // sets trh_last_error(), returns code

int device_of(const trh_context* c);
hipStream_t stream_of(const trh_context* c);
int builder_of(const trh_context* c);
bool can_render(const trh_context* c);   // scene and camera are set: trh_render will not refuse for lack of either
// device pointer of the context's accumulation buffer (rgb sums + count) and its element count (floats)
float* accum_of(const trh_context* c, size_t* float_count);

// Host build of a scene's acceleration structure, once, for several devices; and trh_scene_upload with that
// structure handed in (ignored when the context's builder is the device-side one).
int build_host_shared(const trh_scene_desc* d, bvh::Result& out, bool& from_cache, double& build_ms);
int scene_upload_shared(trh_context* c, const trh_scene_desc* d, const bvh::Result* shared, bool from_cache, double build_ms);

// trh_resolve / trh_present_bgra8 reading the sums from `sums` (a float4 frame on the context's device) instead of
// the context's own accumulation buffer
int resolve_from(trh_context* c, const void* sums, float* rgba_host);
int present_from(trh_context* c, const void* sums, int tonemap, float exposure, uint32_t* bgra_host);

}  // namespace trh_internal
