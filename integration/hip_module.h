// hip_module.h -- the adapter a Tracy checkout adds as src/kernels/raytracing/hip/hip_module.h (INTEGRATION.md
// section 2): binds the generic HipTraceT of hip_trace.h to the reference's own Scene / TracyModule / WindowHandle /
// TracyEvent types (src/scene.h, src/module.h:12-54, src/common.h:253-329). Host C++ only, no HIP headers.
#pragma once
#include "module.h"                       // the reference's TracyModule / TracyEvent / WindowHandle
#include "scene.h"                        // the reference's Scene
#define TRACY_HIP_USE_REFERENCE_MODULE_H  // hip_trace.h must not declare its own interface
#include "hip_trace.h"

#if !defined(TRACY_HIP_GPUS)
 #define TRACY_HIP_GPUS 1                 // GPUs the module drives (0 = all GPUs of the node); set by tracy_hip.cmake
#endif

// HipTraceT<SceneT, Base, WindowT, EventT> reads Scene only through its public accessors (scene.h:36-60,
// mesh.h:83-103, material.h:54-101, texture.h:59-82, camera.h:57-67) and calls the C ABI of tracy_hip.h.
using HipTrace = tracy_hip::HipTraceT<Scene, TracyModule, WindowHandle, TracyEvent>;
