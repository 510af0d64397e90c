// tracy_hip_cli.cpp -- headless twin of Tracy's main loop (src/win_raytracer.cpp:431-589) for the
// `-kernel hip` module: same command line (-scene, -kernel), same call protocol
// (Startup -> [OnEvent]* -> OnUpdate -> OnRender -> GetRayCount(true) -> Shutdown), same statistics
// text ("[%.2f MRays/s @ %.2f fps]", win_raytracer.cpp:532,566). No window: the GPU box has no X
// server. Extra flags: -width/-height (override OUT), -spp N (stop after N samples per pixel),
// -spu K (samples per OnUpdate), -out file.pfm|file.ppm, -device D.
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

// CPUDetails::Tonemap, sRGB variant (cpu_details.cpp:218-243) + Bitmap::SetPixel's Y flip (bitmap.cpp:58-69)
static bool WritePPM(const char* path, const float* rgba, uint32_t w, uint32_t h) {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const float* p = rgba + ((size_t)(h - 1 - y) * w + x) * 4;
            unsigned char o[3];
            for (int k = 0; k < 3; ++k) {
                float c = p[k] * 1.0f;   // TRACY_EXPOSURE = 1
                c = c <= 0.0031308f ? 12.92f * c : 1.055f * std::pow(c, 1.f / 2.4f) - 0.055f;
                c = 255.99f * c;
                o[k] = (unsigned char)(c < 0.f ? 0.f : (c > 255.f ? 255.f : c));
            }
            fwrite(o, 1, 3, f);
        }
    fclose(f);
    return true;
}

int main(int argc, char** argv) {
    uint32_t WIDTH = 640, HEIGHT = 480;   // win_raytracer.cpp:439-440
    std::string scene_path = "data/default.scn", kernelname = "hip", out_path, asset_root = ".";
    uint32_t spp = 64, spu = 1;
    int device = 0;
    bool override_size = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        auto next = [&](const char* def) { return (i + 1 < argc) ? argv[++i] : def; };
        if (a == "-scene") scene_path = next("");
        else if (a == "-kernel") kernelname = next("");
        else if (a == "-width") { WIDTH = (uint32_t)atoi(next("0")); override_size = true; }
        else if (a == "-height") { HEIGHT = (uint32_t)atoi(next("0")); override_size = true; }
        else if (a == "-spp") spp = (uint32_t)atoi(next("64"));
        else if (a == "-spu") spu = (uint32_t)atoi(next("1"));
        else if (a == "-out") out_path = next("");
        else if (a == "-assets") asset_root = next(".");
        else if (a == "-device") device = atoi(next("0"));
    }
    if (kernelname != "hip" && kernelname != "HIP") {   // KernelNameToID (win_raytracer.cpp:48-56)
        fprintf(stderr, "Invalid kernel choice: should be one in [hip]\n");
        return EXIT_FAILURE;
    }
    if (spu == 0) spu = 1;

    HostScene world;
    if (!world.Init(scene_path.c_str(), asset_root.c_str(), WIDTH, HEIGHT, override_size)) {
        fprintf(stderr, "Unable to load scene '%s': %s\n", scene_path.c_str(), world.GetError().c_str());
        return EXIT_FAILURE;
    }
    handle_t win{WIDTH, HEIGHT};
    HipTrace kernel(device, spu);
    if (!kernel.Startup(&win, world)) {
        fprintf(stderr, "Kernel failed to initialize\n");   // win_raytracer.cpp:572-575
        return EXIT_FAILURE;
    }
    fprintf(stderr, "%s: '%s' %ux%u, %u objects, %u triangles, kernel '%s'\n", argv[0], world.GetName().c_str(), WIDTH, HEIGHT,
            world.GetObjectCount(), world.GetTriCount(), kernel.GetModuleName());

    using clock = std::chrono::steady_clock;
    double trace_s = 0.0, stat_s = 0.0;
    uint64_t rays_total = 0;
    uint32_t frames = 0, stat_frames = 0;
    while (kernel.GetSampleCount() < spp) {
        const auto t0 = clock::now();
        kernel.OnUpdate(world, 0.f);
        const double dt = std::chrono::duration<double>(clock::now() - t0).count();
        trace_s += dt;
        stat_s += dt;
        ++frames;
        ++stat_frames;
        kernel.OnRender(&win);
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
    bool ok = true;
    if (!out_path.empty()) {
        const bool ppm = out_path.size() > 4 && out_path.substr(out_path.size() - 4) == ".ppm";
        ok = ppm ? WritePPM(out_path.c_str(), kernel.GetImage(), WIDTH, HEIGHT) : WritePFM(out_path.c_str(), kernel.GetImage(), WIDTH, HEIGHT);
        if (!ok) fprintf(stderr, "cannot write %s\n", out_path.c_str());
    }
    kernel.Shutdown();
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
