/*
 * tracy_scene.h -- C ABI of the host-side scene loader used by the hip module's
 * standalone hosts (bench.py, tests, the headless CLI).
 *
 * Inside Tracy itself this library is NOT needed: Scene::Init (src/scene.cpp:231-545)
 * stays unchanged and HipTrace flattens the reference `Scene` into a trh_scene_desc
 * (see INTEGRATION.md). Outside Tracy (GPU box without the reference tree) this loader
 * produces the very same arrays from the same .scn / .obj files: it follows
 * Scene::Init, AddSphere/AddBox/AddTriangle/AddMesh (src/scene.cpp:50-229) and
 * Mesh::Transform/ComputeNormals/ComputeTangentsAndBitangents/ComputeBoundingBox
 * (src/mesh.h:116-195) and Camera::Setup (src/camera.h:22-55).
 *
 * Pure host code (g++), no HIP.
 */
#ifndef TRACY_SCENE_H
#define TRACY_SCENE_H

#include "tracy_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsc_scene tsc_scene;

typedef struct tsc_camera {
    float eye[3];
    float center[3];
    float up[3];
    float fov_deg;
    float ratio;
    float view[16];           /* column-major, Camera::GetView       (src/camera.h:57) */
    float projection[16];     /* column-major, Camera::GetProjection (src/camera.h:59) */
    float inv_view_proj[16];  /* inverse(projection*view)            (src/camera.h:45) */
} tsc_camera;

typedef struct tsc_texture_info {
    const char* path;   /* as written in the .scn, relative to asset_root */
    int srgb;           /* SRGB flag of the TEX / SKY line (src/scene.cpp:354,378) */
    int is_hdr;         /* path ends in .hdr: float pixels, no /255 (src/scene.cpp:28-36) */
    uint32_t width, height;   /* 0 until pixels were supplied */
} tsc_texture_info;

const char* tsc_last_error(void);

/* Scene::Init (src/scene.cpp:231). Relative paths inside the .scn (data/...) resolve
 * against asset_root (the reference runs with the repository root as CWD).
 * override_width/height > 0 replace the OUT line BEFORE the CAM line is evaluated
 * (the camera ratio is fixed at CAM time, src/scene.cpp:290). Returns NULL on error. */
tsc_scene* tsc_load(const char* scn_path, const char* asset_root, int override_width, int override_height);
void tsc_free(tsc_scene* s);

const char* tsc_name(const tsc_scene* s);
uint32_t tsc_triangle_count(const tsc_scene* s);   /* Scene::GetTriCount (src/scene.h:78) */
int tsc_get_camera(const tsc_scene* s, tsc_camera* out);

/* Camera::UpdateView (src/camera.h:47-55): the realtime camera of the main loop. */
int tsc_set_view(tsc_scene* s, const float eye[3], const float center[3], const float up[3]);

/* Textures referenced by TEX / SKY lines. Image decoding is done by the caller (stb_image
 * in Tracy, PIL in the Python host); the loader applies Texture::Texture's conversion
 * (src/texture.h:93-108): /255 for 8-bit data, then sRGB->linear on all four channels when
 * the line carried SRGB. */
uint32_t tsc_texture_count(const tsc_scene* s);
int tsc_get_texture_info(const tsc_scene* s, uint32_t i, tsc_texture_info* out);
int tsc_set_texture_u8(tsc_scene* s, uint32_t i, uint32_t w, uint32_t h, const uint8_t* rgba8);
int tsc_set_texture_f32(tsc_scene* s, uint32_t i, uint32_t w, uint32_t h, const float* rgba);
/* Deterministic stand-in for the missing data/sky.hdr (formula: BASELINE.md section 3). */
int tsc_set_texture_synthetic_sky(tsc_scene* s, uint32_t i);

/* Texture file decoding without stb_image / PIL (replaces the stbi_load / stbi_loadf calls of
 * Scene::Init, src/scene.cpp:23-48, for hosts outside the Tracy tree): decodes every texture that
 * has no pixels yet from `asset_root`/<path> with the built-in decoders -- baseline JPEG
 * (RGBA8, grey replicated, alpha 255, as stbi_load(.., 4)) and Radiance RGBE .hdr (as stbi_loadf).
 * A missing .hdr becomes the synthetic sky if synthetic_missing_hdr != 0 (the reference's
 * data/sky.hdr is not shipped). Any other missing / undecodable file is an error. */
int tsc_decode_textures(tsc_scene* s, int synthetic_missing_hdr);
/* The decoders on their own (tests, tools). *_out buffers are malloc'ed; release with tsc_free_pixels. */
int tsc_decode_jpeg_file(const char* path, uint32_t* w_out, uint32_t* h_out, uint8_t** rgba8_out);
int tsc_decode_hdr_file(const char* path, uint32_t* w_out, uint32_t* h_out, float** rgba_out);
void tsc_free_pixels(void* p);

/* The OBJ reader on its own (tests, tools): what tinyobj::LoadObj returns to Scene::Init (src/scene.cpp:471-476) --
 * attrib.vertices / texcoords / normals and, per shape, the triangulated corners as (vertex_index, texcoord_index,
 * normal_index) with -1 for "absent" -- so that it can be compared with the real tiny_obj_loader's output
 * (tests/golden/ref_obj_*.npz). Every *_out may be NULL; buffers are malloc'ed, release with tsc_free_pixels.
 * corners_out holds counts->corners x 3 int32, the shapes one after the other (shape_sizes_out: corners per shape). */
typedef struct tsc_obj_counts {
    uint32_t v_floats, vt_floats, vn_floats;   /* array lengths in floats (3, 2, 3 per element) */
    uint32_t shapes, corners;
} tsc_obj_counts;
int tsc_parse_obj_file(const char* path, tsc_obj_counts* counts, float** v_out, float** vt_out, float** vn_out,
                       int32_t** shape_sizes_out, int32_t** corners_out);

/* Stress variant of a scene ("4x / 16x midpoint-subdivided dragon"): every
 * mesh with at least min_triangles triangles has each triangle split into four at its edge
 * midpoints, `levels` times. Shared edges share the new vertex; attributes are averaged; the surface
 * (and so the image) is unchanged, only the triangle count grows 4^levels-fold. Not part of the
 * reference; a fixed rule so that the larger workloads are reproducible. */
int tsc_subdivide(tsc_scene* s, uint32_t min_triangles, uint32_t levels);

/* Fills a trh_scene_desc whose pointers stay valid until tsc_free / the next tsc_set_*.
 * Fails if a referenced texture has no pixels yet. */
int tsc_fill_desc(const tsc_scene* s, trh_scene_desc* out);

#ifdef __cplusplus
}
#endif
#endif /* TRACY_SCENE_H */
