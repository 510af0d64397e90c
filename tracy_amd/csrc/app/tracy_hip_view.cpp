// tracy_hip_view.cpp -- interactive twin of Tracy's main loop (src/win_raytracer.cpp:431-589) for the
// `-kernel hip` module on X11: fixed-size window, progressive accumulation, the reference's camera keys
// (W/S forward/back, A/D left/right, Q/E down/up, ESC quits; win_raytracer.cpp:283-325), the statistics
// in the window title ("[%.2f MRays/s @ %.2f fps]", :521-553). Per frame, as the reference:
// [OnEvent(eCameraCut)] -> OnUpdate -> OnRender -> paint. OnRender leaves the 0x00RRGGBB picture of
// trh_present_bgra8 in HipTrace::GetBitmap(): that buffer IS the XImage (the ZPixmap layout of
// bitmap.cpp:28-39), so painting is one XPutImage, like Bitmap::Paint (bitmap.cpp:84-110).
// The GPU boxes of this project have no X server: the program is built and its start-up failure path is
// tested ("cannot open display"); the loop itself is exercised headless by tracy_hip_cli.
#include <X11/Xlib.h>
#include <X11/Xutil.h>
#include <X11/keysym.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../host/hip_trace.h"
#include "../host/host_scene.h"

using namespace tracy_hip;

int main(int argc, char** argv) {
    uint32_t WIDTH = 640, HEIGHT = 480;   // win_raytracer.cpp:439-440
    std::string scene_path = "data/default.scn", kernelname = "hip", asset_root = ".";
    uint32_t spu = 0;   // 0 = adaptive (the module default)
    int device = -1, gpus = 1;
    bool override_size = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        auto next = [&](const char* def) { return (i + 1 < argc) ? argv[++i] : def; };
        if (a == "-scene") scene_path = next("");
        else if (a == "-kernel") kernelname = next("");
        else if (a == "-width") { WIDTH = (uint32_t)atoi(next("0")); override_size = true; }
        else if (a == "-height") { HEIGHT = (uint32_t)atoi(next("0")); override_size = true; }
        else if (a == "-spu") { const std::string t = next("auto"); spu = t == "auto" ? 0u : (uint32_t)atoi(t.c_str()); }
        else if (a == "-gpus") gpus = atoi(next("1"));
        else if (a == "-assets") asset_root = next(".");
        else if (a == "-device") device = atoi(next("0"));
    }
    if (kernelname != "hip" && kernelname != "HIP") {
        fprintf(stderr, "Invalid kernel choice: should be one in [hip]\n");
        return EXIT_FAILURE;
    }
    if (device >= 0) setenv("HIP_VISIBLE_DEVICES", std::to_string(device).c_str(), 1);   // before the first HIP call

    Display* dpy = XOpenDisplay(nullptr);
    if (!dpy) {
        fprintf(stderr, "cannot open display (no X server; use tracy_hip_cli for headless rendering)\n");
        return EXIT_FAILURE;
    }
    HostScene world;
    if (!world.Init(scene_path.c_str(), asset_root.c_str(), WIDTH, HEIGHT, override_size)) {
        fprintf(stderr, "Unable to load scene '%s': %s\n", scene_path.c_str(), world.GetError().c_str());
        XCloseDisplay(dpy);
        return EXIT_FAILURE;
    }

    // ---- window (TracyCreateWindow, win_raytracer.cpp:183-206): fixed size, close button, key + expose events
    const int screen = DefaultScreen(dpy);
    Window win = XCreateSimpleWindow(dpy, RootWindow(dpy, screen), 0, 0, WIDTH, HEIGHT, 1, BlackPixel(dpy, screen), WhitePixel(dpy, screen));
    XSelectInput(dpy, win, KeyPressMask | StructureNotifyMask | ExposureMask);
    Atom wm_delete = XInternAtom(dpy, "WM_DELETE_WINDOW", False);
    XSetWMProtocols(dpy, win, &wm_delete, 1);
    XSizeHints hints{};
    hints.flags = PMinSize | PMaxSize;
    hints.min_width = hints.max_width = (int)WIDTH;
    hints.min_height = hints.max_height = (int)HEIGHT;
    XSetWMNormalHints(dpy, win, &hints);
    XStoreName(dpy, win, ".:: Tracy 2.0 [hip] ::. (collecting data...)");
    XMapWindow(dpy, win);
    GC gc = DefaultGC(dpy, screen);

    handle_t handle{WIDTH, HEIGHT};
    HipTrace kernel(gpus, spu);
    if (!kernel.Startup(&handle, world)) {
        fprintf(stderr, "Kernel failed to initialize\n");
        XDestroyWindow(dpy, win);
        XCloseDisplay(dpy);
        return EXIT_FAILURE;
    }
    // the module's picture buffer wrapped as an XImage: 32 bits per pixel, 0x00RRGGBB, top row first
    XImage* image = XCreateImage(dpy, DefaultVisual(dpy, screen), (unsigned)DefaultDepth(dpy, screen), ZPixmap, 0,
                                 reinterpret_cast<char*>(const_cast<uint32_t*>(kernel.GetBitmap())), WIDTH, HEIGHT, 32, 0);
    auto paint = [&]() {
        if (image) XPutImage(dpy, win, gc, image, 0, 0, 0, 0, WIDTH, HEIGHT);
        XFlush(dpy);
    };

    using clock = std::chrono::steady_clock;
    auto frame_start = clock::now();
    double stat_s = 0.0;
    uint32_t stat_frames = 0;
    float dt = 0.f;
    bool running = true;
    while (running) {
        // ---- TracyProcessMessages + TracyProcessInputs (win_raytracer.cpp:244-325)
        bool camera_cut = false;
        while (XPending(dpy)) {
            XEvent e;
            XNextEvent(dpy, &e);
            if (e.type == Expose) {
                paint();
            } else if (e.type == ClientMessage && (Atom)e.xclient.data.l[0] == wm_delete) {
                running = false;
            } else if (e.type == KeyPress) {
                const KeySym key = XLookupKeysym(&e.xkey, 0);
                if (key == XK_Escape) { running = false; continue; }
                tsc_camera cam{};
                if (!world.GetCamera(cam)) continue;
                float f[3], r[3], pos[3];
                for (int a = 0; a < 3; ++a) { f[a] = cam.center[a] - cam.eye[a]; pos[a] = cam.eye[a]; }
                r[0] = f[1] * cam.up[2] - f[2] * cam.up[1];   // normalize(cross(forward, up))
                r[1] = f[2] * cam.up[0] - f[0] * cam.up[2];
                r[2] = f[0] * cam.up[1] - f[1] * cam.up[0];
                const float rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                if (rl > 0.f) for (int a = 0; a < 3; ++a) r[a] /= rl;
                const float* dir = nullptr;
                float sign = 1.f;
                switch (key) {
                case XK_w: dir = f; break;
                case XK_s: dir = f; sign = -1.f; break;
                case XK_d: dir = r; break;
                case XK_a: dir = r; sign = -1.f; break;
                case XK_e: dir = cam.up; break;
                case XK_q: dir = cam.up; sign = -1.f; break;
                default: break;
                }
                if (dir) {
                    for (int a = 0; a < 3; ++a) pos[a] += sign * dt * dir[a];
                    world.SetView(pos, cam.center, cam.up);   // Camera::UpdateView (camera.h:47-55)
                    camera_cut = true;
                }
            }
        }
        if (!running) break;
        if (camera_cut) kernel.OnEvent(TracyEvent::eCameraCut, &handle, world);   // win_raytracer.cpp:502-505

        kernel.OnUpdate(world, dt);
        kernel.OnRender(&handle);
        paint();

        const auto now = clock::now();
        dt = std::chrono::duration<float>(now - frame_start).count();
        frame_start = now;
        stat_s += dt;
        ++stat_frames;
        if (stat_s > 1.0) {   // win_raytracer.cpp:521-553
            char title[160];
            snprintf(title, sizeof(title), ".:: Tracy 2.0 [%s] ::. '%s' %u spp [%.2f MRays/s @ %.2f fps]", kernel.GetModuleName(), world.GetName().c_str(),
                     kernel.GetSampleCount(), kernel.GetRayCount64(true) * 1e-6 / stat_s, stat_frames / stat_s);
            XStoreName(dpy, win, title);
            stat_s = 0.0;
            stat_frames = 0;
        }
    }

    if (image) {
        image->data = nullptr;   // the pixels belong to the module
        XDestroyImage(image);
    }
    kernel.Shutdown();
    XDestroyWindow(dpy, win);
    XCloseDisplay(dpy);
    return EXIT_SUCCESS;
}
