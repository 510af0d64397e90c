// tracy_module.h -- the render-module interface of Tracy as a standalone declaration.
//
// Same virtuals, names, argument meaning and call protocol as the reference's TracyModule
// (src/module.h:17-54) and TracyEvent (src/module.h:12). Inside the Tracy tree this header is NOT
// used: hip_trace.h is included after the reference's own "module.h" (define
// TRACY_HIP_USE_REFERENCE_MODULE_H) and HipTrace derives from the real class.
#pragma once
#include <cstdint>

#if !defined(TRACY_HIP_USE_REFERENCE_MODULE_H)

namespace tracy_hip {

struct handle_t {            // the fields of the reference's handle_t a module reads (src/common.h:292-300)
    uint32_t width;
    uint32_t height;
};
using WindowHandle = handle_t*;

enum class TracyEvent { eCreate, eResize, eCameraCut, eDestroy };

template <class SceneT>
class TracyModuleT {
public:
    TracyModuleT() = default;
    virtual ~TracyModuleT() {}
    TracyModuleT(const TracyModuleT&) = delete;
    TracyModuleT& operator=(const TracyModuleT&) = delete;
    TracyModuleT(TracyModuleT&&) = delete;
    TracyModuleT& operator=(TracyModuleT&&) = delete;

    virtual bool Startup(const WindowHandle in_Window, const SceneT& in_Scene) = 0;
    virtual void Shutdown() = 0;
    virtual void OnUpdate(const SceneT& in_Scene, float in_DeltaTime) = 0;
    virtual void OnRender(const WindowHandle in_Window) = 0;
    virtual void OnEvent(TracyEvent in_Event, const WindowHandle in_Window, const SceneT& in_Scene) = 0;
    virtual const char* GetModuleName() const = 0;
    virtual uint32_t GetRayCount(bool in_ShouldReset = false) { (void)in_ShouldReset; return 0; }
};

}  // namespace tracy_hip

#endif
