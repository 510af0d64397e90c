// ref_dump.cpp -- TEST INFRASTRUCTURE, build container only. Runs the reference's OWN third-party loaders, as they
// stand under /root/reference/ext (stb_image.h, tiny_obj_loader.h: the two parts of the reference's input side that
// are self-contained and compile here), exactly as the reference calls them, and dumps their results so that
// tests/golden/make_ref_fixtures.py can freeze them as fixtures:
//   jpeg <file> <out>   stbi_load(path, &w, &h, &bpp, 4)            (src/scene.cpp:39)      -> int32 w, h, bpp; u8 RGBA
//   hdr  <file> <out>   stbi_is_hdr + stbi_loadf(path, ..., 4)      (src/scene.cpp:28-30)   -> int32 w, h, bpp; f32 RGBA
//   obj  <file> <out>   tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, file)   (src/scene.cpp:471-476)
//                       -> int32 nv, nvt, nvn (floats), nshapes; the three float arrays; per shape int32 n, then n x
//                          (vertex_index, texcoord_index, normal_index)
// Nothing of the reference is copied: the headers are included from where they lie (-I/root/reference, see
// oracle/Makefile, target `ref`); the binary goes to oracle/_ref/ (git-ignored) and is never shipped or loaded by the
// product. The GPU box has no /root/reference: only the fixtures travel.
#define TINYOBJLOADER_IMPLEMENTATION
#include "ext/tiny_obj_loader.h"
#define STB_IMAGE_IMPLEMENTATION
#include "ext/stb_image.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static bool put(FILE* f, const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: ref_dump jpeg|hdr|obj <file> <out>\n"); return 2; }
    const std::string mode = argv[1];
    FILE* out = fopen(argv[3], "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    bool ok = true;
    if (mode == "jpeg") {
        int w, h, bpp;
        unsigned char* px = stbi_load(argv[2], &w, &h, &bpp, 4);
        if (!px) { fprintf(stderr, "stbi_load failed: %s\n", stbi_failure_reason()); return 1; }
        const int32_t hd[3] = {w, h, bpp};
        ok = put(out, hd, sizeof(hd)) && put(out, px, (size_t)w * h * 4);
        stbi_image_free(px);
    } else if (mode == "hdr") {
        if (!stbi_is_hdr(argv[2])) { fprintf(stderr, "stbi_is_hdr says no\n"); return 1; }
        int w, h, bpp;
        float* px = stbi_loadf(argv[2], &w, &h, &bpp, 4);
        if (!px) { fprintf(stderr, "stbi_loadf failed: %s\n", stbi_failure_reason()); return 1; }
        const int32_t hd[3] = {w, h, bpp};
        ok = put(out, hd, sizeof(hd)) && put(out, px, (size_t)w * h * 4 * sizeof(float));
        stbi_image_free(px);
    } else if (mode == "obj") {
        tinyobj::attrib_t attrib;
        std::vector<tinyobj::shape_t> shapes;
        std::vector<tinyobj::material_t> materials;
        std::string warn, err;
        if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, argv[2])) { fprintf(stderr, "LoadObj failed: %s\n", err.c_str()); return 1; }
        const int32_t hd[4] = {(int32_t)attrib.vertices.size(), (int32_t)attrib.texcoords.size(), (int32_t)attrib.normals.size(), (int32_t)shapes.size()};
        static_assert(sizeof(tinyobj::real_t) == sizeof(float), "the reference builds tinyobj with float reals");
        ok = put(out, hd, sizeof(hd)) && put(out, attrib.vertices.data(), attrib.vertices.size() * 4) &&
             put(out, attrib.texcoords.data(), attrib.texcoords.size() * 4) && put(out, attrib.normals.data(), attrib.normals.size() * 4);
        for (const tinyobj::shape_t& s : shapes) {
            const int32_t n = (int32_t)s.mesh.indices.size();
            ok = ok && put(out, &n, 4);
            for (const tinyobj::index_t& ix : s.mesh.indices) {
                const int32_t t[3] = {ix.vertex_index, ix.texcoord_index, ix.normal_index};
                ok = ok && put(out, t, sizeof(t));
            }
        }
    } else {
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }
    fclose(out);
    return ok ? 0 : 1;
}
