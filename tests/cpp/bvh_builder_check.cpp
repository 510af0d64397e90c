// bvh_builder_check.cpp -- the task-parallel BVH builder under ThreadSanitizer / ASan+UBSan (host only).
// Builds trees over synthetic triangle soups with several thread counts, validates them structurally and
// checks that the layout hash does not depend on the thread count.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../tracy_amd/csrc/host/bvh_builder.h"

static uint32_t st = 99u;
static float rnd() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.f; }

int main() {
    const size_t sizes[] = {0, 1, 7, 1000, 150000};
    for (size_t n : sizes) {
        std::vector<bvh::InputTriangle> tris(n);
        for (size_t i = 0; i < n; ++i) {
            const float cx = rnd() * 20.f - 10.f, cy = rnd() * 4.f, cz = rnd() * 20.f - 10.f, r = 0.01f + rnd() * 0.2f;
            bvh::InputTriangle& t = tris[i];
            for (int a = 0; a < 3; ++a) {
                t.v0[a] = (a == 0 ? cx : a == 1 ? cy : cz) + (rnd() - .5f) * r;
                t.v1[a] = (a == 0 ? cx : a == 1 ? cy : cz) + (rnd() - .5f) * r;
                t.v2[a] = (a == 0 ? cx : a == 1 ? cy : cz) + (rnd() - .5f) * r;
            }
            t.packed_id = (uint32_t)i;
        }
        uint64_t first_hash = 0;
        for (const char* threads : {"1", "3", "8"}) {
            setenv("TRH_BUILD_THREADS", threads, 1);
            bvh::Result r;
            bvh::build(tris, r);
            std::string why;
            uint64_t h = 0;
            const uint64_t bad = bvh::validate(tris, r, why, &h);
            if (bad) { fprintf(stderr, "n=%zu threads=%s: %llu violations, first: %s\n", n, threads, (unsigned long long)bad, why.c_str()); return 1; }
            if (threads[0] == '1') first_hash = h;
            else if (h != first_hash) { fprintf(stderr, "n=%zu: layout depends on the thread count\n", n); return 1; }
        }
    }
    puts("bvh builder ok");
    return 0;
}
