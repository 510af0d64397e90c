# tracy_hip.cmake -- the MODULE_HIP block of Tracy's CMakeLists.txt (included by integration/CMakeLists.patch
# next to the MODULE_CUDA / MODULE_OPENGL blocks, reference CMakeLists.txt:218-297).
#
# Layout expected under the Tracy tree:
#   ext/tracy_hip/tracy_hip.cmake                this file
#   ext/tracy_hip/include/tracy_hip.h            the C ABI
#   ext/tracy_hip/csrc/...                       tracy_amd/csrc (kernels + host builder), optional: only to build the library here
#   ext/tracy_hip/lib/libtracy_hip.so            prebuilt library, used if present
#   src/kernels/raytracing/hip/hip_module.h      integration/hip_module.h (the adapter)
#   src/kernels/raytracing/hip/hip_trace.h       tracy_amd/csrc/host/hip_trace.h
#   src/kernels/raytracing/hip/tracy_module.h    tracy_amd/csrc/host/tracy_module.h
#
# The host side of the module is plain C++ (no HIP headers) and is compiled by the project's own compiler; only the
# library holds device code, built by hipcc for gfx950 -- outside CMake's language support on purpose (Tracy enables
# CUDA that way, but a HIP language toolchain is not something a Tracy checkout can be assumed to have).

message(STATUS "Enabling HIP Raytracing (MI355X / gfx950)")
add_definitions(-DHIP_KERNEL=1)
set(TRACY_HIP_GPUS "1" CACHE STRING "GPUs the hip module drives (0 = all GPUs of the node)")
add_definitions(-DTRACY_HIP_GPUS=${TRACY_HIP_GPUS})

set(TRACY_HIP_ROOT "${CMAKE_CURRENT_LIST_DIR}")
include_directories(${TRACY_HIP_ROOT}/include)
include_directories(./src/kernels/raytracing/hip/)

set(hip_includes "./src/kernels/raytracing/hip/hip_module.h"
                 "./src/kernels/raytracing/hip/hip_trace.h"
                 "./src/kernels/raytracing/hip/tracy_module.h")
source_group("kernel\\raytracing\\hip\\include" FILES ${hip_includes})
list(APPEND tracy_includes ${hip_includes})

set(TRACY_HIP_LIBRARY "${TRACY_HIP_ROOT}/lib/libtracy_hip.so")
if(NOT EXISTS ${TRACY_HIP_LIBRARY})
	# build the library from the sources with hipcc (one command, the same tracy_amd/build.py runs)
	find_program(HIPCC hipcc HINTS /opt/rocm/bin ENV ROCM_PATH PATH_SUFFIXES bin REQUIRED)
	set(TRACY_HIP_LIBRARY "${CMAKE_BINARY_DIR}/libtracy_hip.so")
	set(hip_library_sources "${TRACY_HIP_ROOT}/csrc/hip/tracy_hip.hip"
	                        "${TRACY_HIP_ROOT}/csrc/hip/tracy_group.hip"
	                        "${TRACY_HIP_ROOT}/csrc/hip/lbvh_build.hip"
	                        "${TRACY_HIP_ROOT}/csrc/hip/tile_order.hip"
	                        "${TRACY_HIP_ROOT}/csrc/host/bvh_builder.cpp")
	add_custom_command(OUTPUT ${TRACY_HIP_LIBRARY}
	                   COMMAND ${HIPCC} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -shared
	                           ${hip_library_sources} -ldl -o ${TRACY_HIP_LIBRARY}
	                   DEPENDS ${hip_library_sources}
	                   COMMENT "hipcc: libtracy_hip.so (gfx950)")
	add_custom_target(tracy_hip_library ALL DEPENDS ${TRACY_HIP_LIBRARY})
	set(TRACY_HIP_NEEDS_BUILD ON)
endif()
list(APPEND tracy_libraries ${TRACY_HIP_LIBRARY} ${CMAKE_DL_LIBS})
# (CMakeLists.patch adds, after add_executable: add_dependencies(${PROJECT_NAME} tracy_hip_library) when TRACY_HIP_NEEDS_BUILD is set --
#  a full path in tracy_libraries creates no dependency on the custom command, and a parallel build could link too early)
