// decoder_fuzz.cpp -- robustness of the texture decoders and the .scn / OBJ readers against damaged files.
// Built by tests/test_host_sanitizers.py with -fsanitize=address,undefined (host code only) and run on
// deterministic mutations of the shipped assets: the decoders may refuse a file, they must not read or
// write out of bounds, overflow, or hang.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tracy_scene.h"
#include "../../tracy_amd/csrc/host/image_decode.h"

static uint32_t rng_state = 0x12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: decoder_fuzz file.jpg file.hdr rounds [scene.scn asset_root tmpdir]\n"); return 2; }
    const int rounds = atoi(argv[3]);
    std::vector<uint8_t> jpg, hdr;
    if (!imgdec::read_file(argv[1], jpg) || !imgdec::read_file(argv[2], hdr)) { fprintf(stderr, "cannot read inputs\n"); return 2; }
    int ok = 0, refused = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int which = 0; which < 2; ++which) {
            std::vector<uint8_t> b = which ? hdr : jpg;
            const int kind = r % 4;
            if (kind == 0) b.resize(rnd() % (b.size() + 1));                                        // truncation
            else if (kind == 1) for (int k = 0; k < 8; ++k) b[rnd() % b.size()] = (uint8_t)rnd();   // byte noise anywhere
            else if (kind == 2) for (int k = 0; k < 16; ++k) b[rnd() % std::min<size_t>(b.size(), 700)] = (uint8_t)rnd();  // header / table damage
            else { const size_t at = rnd() % b.size(); b.insert(b.begin() + at, (size_t)(rnd() % 64), (uint8_t)0xFF); }    // marker soup
            std::string err;
            bool good;
            if (which) { imgdec::ImageF img; good = imgdec::decode_hdr(b.data(), b.size(), img, err); if (good && img.rgba.size() != (size_t)img.width * img.height * 4) return 1; }
            else { imgdec::Image8 img; good = imgdec::decode_jpeg(b.data(), b.size(), img, err); if (good && img.rgba.size() != (size_t)img.width * img.height * 4) return 1; }
            good ? ++ok : ++refused;
            if (!good && err.empty()) { fprintf(stderr, "refusal without a message\n"); return 1; }
        }
    }
    // damaged scene descriptions: the loader must return a scene or an error, never crash
    int scenes = 0;
    if (argc >= 7) {
        std::vector<uint8_t> scn;
        if (!imgdec::read_file(argv[4], scn)) return 2;
        for (int r = 0; r < rounds; ++r) {
            std::vector<uint8_t> b = scn;
            for (int k = 0; k < 6; ++k) b[rnd() % b.size()] = (uint8_t)(" (),-0123456789.eESOBTCAMLKYX\n"[rnd() % 30]);
            if (r % 3 == 0) b.resize(rnd() % (b.size() + 1));
            const std::string path = std::string(argv[6]) + "/fuzz.scn";
            FILE* f = fopen(path.c_str(), "wb");
            if (!f) return 2;
            fwrite(b.data(), 1, b.size(), f);
            fclose(f);
            tsc_scene* s = tsc_load(path.c_str(), argv[5], 32, 18);
            if (s) {
                trh_scene_desc d;
                (void)tsc_decode_textures(s, 1);
                (void)tsc_fill_desc(s, &d);
                tsc_free(s);
                ++scenes;
            }
        }
    }
    // damaged OBJ files: face corners with index 0, indices beyond the arrays, relative indices beyond their start,
    // faces that name vertices defined later, polygons, truncation and byte noise. tsc_parse_obj_file may refuse a
    // file (with a message); under ASan/UBSan it must never read outside its arrays.
    int objs_ok = 0, objs_refused = 0;
    if (argc >= 7) {
        const std::string path = std::string(argv[6]) + "/fuzz.obj";
        const char* cases[] = {
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 0\n",                      // index 0
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n",                      // beyond nv
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 2147483647\n",             // far beyond nv
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -4\n",                   // relative index before the start
            "v 0 0 0\nv 1 0 0\nf 1 2 3\nv 0 1 0\n",                      // vertex defined after its face
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nf 1/2 2/1 3/1\n",       // texture coordinate beyond nvt
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//-3 3//1\n", // normal before the start
            "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv -1 .5 0\nf 1 2 3 4 5\nf 1 2 3 4\nf 5 4 3 2 1 1 1\n",   // polygons, repeated corners
            "v 1e400 -1e-400 .e5\nv 1e+ 2 3\nv - + .\nf 1 2 3\n",        // numbers tinyobj's parser rejects or overflows
            "f 1 2 3\n", "f\n", "f 1\nf 1 2\n", "v\nvn\nvt\nf 1// 1/ /1\n"};
        for (const char* c : cases) {
            FILE* f = fopen(path.c_str(), "wb");
            if (!f) return 2;
            fputs(c, f);
            fclose(f);
            tsc_obj_counts cnt;
            float *v = nullptr, *vt = nullptr, *vn = nullptr;
            int32_t *sizes = nullptr, *corners = nullptr;
            if (tsc_parse_obj_file(path.c_str(), &cnt, &v, &vt, &vn, &sizes, &corners) == 0) {
                for (uint32_t k = 0; k < cnt.corners; ++k)   // what a caller does next: index the arrays
                    if (corners[3 * k] < 0 || (uint32_t)corners[3 * k] * 3 + 2 >= cnt.v_floats) return 1;
                tsc_free_pixels(v); tsc_free_pixels(vt); tsc_free_pixels(vn); tsc_free_pixels(sizes); tsc_free_pixels(corners);
                ++objs_ok;
            } else {
                if (!*tsc_last_error()) { fprintf(stderr, "OBJ refusal without a message\n"); return 1; }
                ++objs_refused;
            }
        }
        std::vector<uint8_t> obj;
        const std::string teapot = std::string(argv[5]) + "/data/teapot.obj";
        if (!imgdec::read_file(teapot, obj)) return 2;
        obj.resize(std::min<size_t>(obj.size(), 40000));
        for (int r = 0; r < rounds; ++r) {
            std::vector<uint8_t> b = obj;
            for (int k = 0; k < 24; ++k) b[rnd() % b.size()] = (uint8_t)("vfntgo /-0123456789.e\n"[rnd() % 22]);
            if (r % 3 == 0) b.resize(rnd() % (b.size() + 1));
            FILE* f = fopen(path.c_str(), "wb");
            if (!f) return 2;
            fwrite(b.data(), 1, b.size(), f);
            fclose(f);
            tsc_obj_counts cnt;
            int32_t* corners = nullptr;
            if (tsc_parse_obj_file(path.c_str(), &cnt, nullptr, nullptr, nullptr, nullptr, &corners) == 0) {
                for (uint32_t k = 0; k < cnt.corners; ++k)
                    if (corners[3 * k] < 0 || (uint32_t)corners[3 * k] * 3 + 2 >= cnt.v_floats) return 1;
                tsc_free_pixels(corners);
                ++objs_ok;
            } else {
                ++objs_refused;
            }
        }
    }
    printf("decoded %d, refused %d, scenes loaded %d, objs parsed %d, objs refused %d\n", ok, refused, scenes, objs_ok, objs_refused);
    return 0;
}
