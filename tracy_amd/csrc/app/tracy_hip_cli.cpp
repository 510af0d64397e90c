// tracy_hip_cli.cpp -- headless twin of Tracy's main loop (src/win_raytracer.cpp:431-589) for the
// `-kernel hip` module: same command line (-scene, -kernel), same call protocol
// (Startup -> [OnEvent]* -> OnUpdate -> OnRender -> GetRayCount(true) -> Shutdown), same statistics
// text ("[%.2f MRays/s @ %.2f fps]", win_raytracer.cpp:532,566). No window: the GPU box has no X
// server. Extra flags: -width/-height (override OUT), -spp N (stop after N samples per pixel),
// -spu K|auto (samples per OnUpdate; auto = the module's adaptive default), -gpus N (GPUs of the node to use, 0 = all: the
// samples of every update are split across them and one RCCL reduce to GPU 0 makes the picture), -async (OnUpdate only
// enqueues; the frame is timed as OnUpdate + OnRender), -out file.pfm (linear) | file.ppm | file.png (tonemapped picture),
// -tonemap srgb|reinhard|aces|none and -exposure E (the reference's TRACY_TONEMAPPING / TRACY_EXPOSURE
// build options, CMakeLists.txt:92-103), -debug basecolor|normals|metalness|roughness|emissive (the DEBUG_SHOW_*
// build options, CMakeLists.txt:23-35), -builder sah|lbvh (host binned-SAH build, or the device-side linear BVH), -device D.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../host/host_scene.h"

using namespace tracy_hip;

static bool WritePFM(const char* path, const float* rgba, uint32_t w, uint32_t h) {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h);   // little endian, rows bottom to top == the module's row order
    for (size_t i = 0; i < (size_t)w * h; ++i) fwrite(rgba + i * 4, sizeof(float), 3, f);
    fclose(f);
    return true;
}

// the module's own picture (trh_present_bgra8: CPUDetails::Tonemap + Bitmap::SetPixel), 0x00RRGGBB, top row first
static bool WritePPM(const char* path, const uint32_t* bgra, uint32_t w, uint32_t h) {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const unsigned char o[3] = {(unsigned char)(bgra[i] >> 16), (unsigned char)(bgra[i] >> 8), (unsigned char)bgra[i]};
        fwrite(o, 1, 3, f);
    }
    fclose(f);
    return true;
}

// 8-bit RGB PNG with stored (uncompressed) deflate blocks: no zlib needed on the box
static bool WritePNG(const char* path, const uint32_t* bgra, uint32_t w, uint32_t h) {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    static uint32_t crc_table[256];
    if (!crc_table[1])
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
    auto be32 = [](unsigned char* p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; };
    auto chunk = [&](const char* type, const std::string& data) {
        unsigned char len[4], crc[4];
        be32(len, (uint32_t)data.size());
        uint32_t c = 0xffffffffu;
        for (int i = 0; i < 4; ++i) c = crc_table[(c ^ (unsigned char)type[i]) & 255u] ^ (c >> 8);
        for (unsigned char b : data) c = crc_table[(c ^ b) & 255u] ^ (c >> 8);
        be32(crc, c ^ 0xffffffffu);
        fwrite(len, 1, 4, f); fwrite(type, 1, 4, f); fwrite(data.data(), 1, data.size(), f); fwrite(crc, 1, 4, f);
    };
    fwrite("\x89PNG\r\n\x1a\n", 1, 8, f);
    std::string ihdr(13, '\0');
    be32((unsigned char*)&ihdr[0], w);
    be32((unsigned char*)&ihdr[4], h);
    ihdr[8] = 8; ihdr[9] = 2;   // 8 bits, colour type RGB
    chunk("IHDR", ihdr);
    std::string raw;
    raw.reserve((size_t)h * (1 + 3 * (size_t)w));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back('\0');   // filter: none
        for (uint32_t x = 0; x < w; ++x) {
            const uint32_t p = bgra[(size_t)y * w + x];
            raw.push_back((char)(p >> 16)); raw.push_back((char)(p >> 8)); raw.push_back((char)p);
        }
    }
    std::string z("\x78\x01", 2);
    uint32_t a = 1, b = 0;
    for (unsigned char c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        const bool last = off + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((char)(n & 255)); z.push_back((char)(n >> 8));
        z.push_back((char)(~n & 255)); z.push_back((char)((~n >> 8) & 255));
        z.append(raw, off, n);
        if (last) break;
    }
    unsigned char ad[4];
    be32(ad, (b << 16) | a);
    z.append((const char*)ad, 4);
    chunk("IDAT", z);
    chunk("IEND", std::string());
    fclose(f);
    return true;
}

int main(int argc, char** argv) {
    uint32_t WIDTH = 640, HEIGHT = 480;   // win_raytracer.cpp:439-440
    std::string scene_path = "data/default.scn", kernelname = "hip", out_path, asset_root = ".";
    uint32_t spp = 64, spu = 1;
    int device = -1, gpus = 1, tonemap = TRH_TONEMAP_SRGB, debug_view = TRH_VIEW_NONE, builder = TRH_BUILDER_HOST_SAH;
    float exposure = 1.f;
    bool override_size = false, async_update = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        auto next = [&](const char* def) { return (i + 1 < argc) ? argv[++i] : def; };
        if (a == "-scene") scene_path = next("");
        else if (a == "-kernel") kernelname = next("");
        else if (a == "-width") { WIDTH = (uint32_t)atoi(next("0")); override_size = true; }
        else if (a == "-height") { HEIGHT = (uint32_t)atoi(next("0")); override_size = true; }
        else if (a == "-spp") spp = (uint32_t)atoi(next("64"));
        else if (a == "-spu") { const std::string t = next("1"); spu = t == "auto" ? 0u : std::max(1u, (uint32_t)atoi(t.c_str())); }
        else if (a == "-gpus") gpus = atoi(next("1"));
        else if (a == "-async") async_update = true;
        else if (a == "-out") out_path = next("");
        else if (a == "-assets") asset_root = next(".");
        else if (a == "-device") device = atoi(next("0"));
        else if (a == "-exposure") exposure = (float)atof(next("1"));
        else if (a == "-debug") {   // the reference's DEBUG_SHOW_* build options (CMakeLists.txt:23-35)
            const std::string t = next("none");
            debug_view = t == "basecolor" ? TRH_VIEW_BASECOLOR : t == "normals" ? TRH_VIEW_NORMALS : t == "metalness" ? TRH_VIEW_METALNESS
                       : t == "roughness" ? TRH_VIEW_ROUGHNESS : t == "emissive" ? TRH_VIEW_EMISSIVE : TRH_VIEW_NONE;
        }
        else if (a == "-builder") builder = std::string(next("sah")) == "lbvh" ? TRH_BUILDER_DEVICE_LBVH : TRH_BUILDER_HOST_SAH;
        else if (a == "-tonemap") {
            const std::string t = next("srgb");
            tonemap = t == "none" ? TRH_TONEMAP_NONE : t == "reinhard" ? TRH_TONEMAP_REINHARD : t == "aces" ? TRH_TONEMAP_ACES : TRH_TONEMAP_SRGB;
        }
    }
    if (kernelname != "hip" && kernelname != "HIP") {   // KernelNameToID (win_raytracer.cpp:48-56)
        fprintf(stderr, "Invalid kernel choice: should be one in [hip]\n");
        return EXIT_FAILURE;
    }
    // a group uses devices 0..N-1; one particular GPU is chosen by making it the only visible one (before the first HIP call)
    if (device >= 0) setenv("HIP_VISIBLE_DEVICES", std::to_string(device).c_str(), 1);

    HostScene world;
    if (!world.Init(scene_path.c_str(), asset_root.c_str(), WIDTH, HEIGHT, override_size)) {
        fprintf(stderr, "Unable to load scene '%s': %s\n", scene_path.c_str(), world.GetError().c_str());
        return EXIT_FAILURE;
    }
    handle_t win{WIDTH, HEIGHT};
    HipTrace kernel(gpus, spu);
    kernel.SetSynchronousUpdate(!async_update);
    kernel.SetTonemap(tonemap, exposure);
    kernel.SetDebugView(debug_view);
    kernel.SetBuilder(builder);
    if (!kernel.Startup(&win, world)) {
        fprintf(stderr, "Kernel failed to initialize\n");   // win_raytracer.cpp:572-575
        return EXIT_FAILURE;
    }
    fprintf(stderr, "%s: '%s' %ux%u, %u objects, %u triangles, kernel '%s' on %d GPU(s)\n", argv[0], world.GetName().c_str(), WIDTH, HEIGHT,
            world.GetObjectCount(), world.GetTriCount(), kernel.GetModuleName(), kernel.GetGpuCount());

    using clock = std::chrono::steady_clock;
    double trace_s = 0.0, stat_s = 0.0;
    uint64_t rays_total = 0;
    uint32_t frames = 0, stat_frames = 0;
    while (kernel.GetSampleCount() < spp) {
        const auto t0 = clock::now();
        kernel.OnUpdate(world, 0.f);
        if (async_update) kernel.OnRender(&win);   // the update is still in flight: the frame ends when its picture is there
        const double dt = std::chrono::duration<double>(clock::now() - t0).count();
        trace_s += dt;
        stat_s += dt;
        ++frames;
        ++stat_frames;
        if (!async_update) kernel.OnRender(&win);
        if (stat_s > 1.0 || stat_frames > 100) {   // win_raytracer.cpp:521-553
            const uint64_t rays = kernel.GetRayCount64(true);
            rays_total += rays;
            fprintf(stderr, "[%.2f MRays/s @ %.2f fps]\n", rays * 1e-6 / stat_s, stat_frames / stat_s);
            stat_s = 0.0;
            stat_frames = 0;
        }
    }
    rays_total += kernel.GetRayCount64(true);
    kernel.OnRender(&win);
    fprintf(stderr, "*** Performance: %.2f MRays/s and %.2f fps on average\n", rays_total * 1e-6 / trace_s, frames / trace_s);
    fprintf(stderr, "*** %.2f Msamples/s over %u updates (%u spp, last update %u spp)\n", (double)WIDTH * HEIGHT * kernel.GetSampleCount() * 1e-6 / trace_s,
            frames, kernel.GetSampleCount(), kernel.GetSamplesPerUpdate());
    bool ok = true;
    if (!out_path.empty()) {
        const std::string ext = out_path.size() > 4 ? out_path.substr(out_path.size() - 4) : std::string();
        ok = ext == ".ppm"   ? WritePPM(out_path.c_str(), kernel.GetBitmap(), WIDTH, HEIGHT)
             : ext == ".png" ? WritePNG(out_path.c_str(), kernel.GetBitmap(), WIDTH, HEIGHT)
                             : WritePFM(out_path.c_str(), kernel.GetImage(), WIDTH, HEIGHT);
        if (!ok) fprintf(stderr, "cannot write %s\n", out_path.c_str());
    }
    kernel.Shutdown();
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
