// valu_rate.hip -- development microbenchmark: how many cycles does one wave64 VALU instruction occupy a SIMD of
// gfx950, with 1 / 2 / 4 waves per SIMD? (independent v_fma_f32 chains; also v_pk_fma_f32, v_perm_b32, v_cndmask, v_min3)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.000001f, c = 0.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 1) {
            asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                         "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 2) {
            asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                         "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 3) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else {
            asm volatile("v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[1,0,0]\n"
                         "v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[1,0,0]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out)[1 << 20] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* d) {
    const int iters = 20000;
    for (int threads : {64, 256, 512, 1024}) {   // 1 block per CU: 1 wave on one SIMD; 1, 2, 4 waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<KIND><<<256, threads>>>(d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<256, threads>>>(d, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long cyc = 0;
        hipMemcpy(&cyc, reinterpret_cast<long long*>(d) + (1 << 20), 8, hipMemcpyDeviceToHost);
        const double waves_per_simd = threads <= 64 ? 1.0 : threads / 256.0;
        const double instr_per_simd = waves_per_simd * iters * 8.0;
        printf("%-14s %4d threads/CU (%.0f waves/SIMD): %.3f ms, in-kernel %lld cycles -> %.2f cycles per wave-instruction per SIMD (clock %.2f GHz)\n", name, threads,
               waves_per_simd, ms, cyc, (double)cyc / instr_per_simd, cyc / (ms * 1e6));
    }
}

int main() {
    float* d;
    hipMalloc(&d, (1 << 20) * 8 + 64);
    run<0>("v_fma_f32", d);
    run<1>("v_min3_f32", d);
    run<2>("v_perm_b32", d);
    run<3>("v_add_u32", d);
    run<4>("v_fma_mix_f32", d);
    return 0;
}
