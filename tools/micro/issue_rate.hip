// issue_rate.hip -- development microbenchmark (gfx950): issue cost of the instruction kinds the traversal loop is made
// of, at 1 / 2 / 4 waves per SIMD, every CU busy: cycles per wave-instruction per SIMD from the in-kernel shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue_rate.hip -o tools/micro/issue_rate && tools/micro/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned s0 = iters, s1 = iters + 1, s2 = iters + 2, s3 = iters + 3;
    const float m = 1.000001f, c = 0.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // 8 independent SALU
            asm volatile(REP8("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n") : "+s"(s0), "+s"(s1) : : "scc");
        } else if (KIND == 1) {   // 4 VALU + 4 SALU interleaved
            asm volatile(REP8("v_fma_f32 %0, %0, %4, %5\n s_add_u32 %2, %2, 3\n") : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : "v"(m), "v"(c) : "scc");
        } else if (KIND == 2) {   // v_cmp to an SGPR pair + v_cndmask using it
            asm volatile(REP8("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n") : "+v"(a0), "+v"(a1) : : "s20", "s21");
        } else if (KIND == 3) {   // exec save / restore pairs around one VALU
            asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n") : "+v"(a0), "+v"(a1) : : "s20", "s21", "vcc");
        } else if (KIND == 4) {   // census: v_cmp + s_bcnt1
            asm volatile(REP8("v_cmp_lt_i32_e64 s[20:21], -1, %0\n s_bcnt1_i32_b64 %1, s[20:21]\n") : "+v"(a0), "+s"(s0) : : "s20", "s21", "scc");
        } else if (KIND == 5) {   // dependent VALU chain
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(m), "v"(c));
        } else if (KIND == 6) {   // LDS read + dependent wait, as the stack pop
            asm volatile(REP8("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a1) : "v"(0));
        } else if (KIND == 7) {   // branch not taken + scalar compare
            asm volatile(REP8("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1\n s_nop 0\n") : : "s"(s0) : "scc");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + s0 + s1 + s2 + s3;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out)[1 << 20] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* d, int instr_per_iter) {
    const int iters = 8000;
    for (int threads : {256, 512, 1024}) {
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
        const double waves_per_simd = threads / 256.0;
        printf("%-34s %.0f waves/SIMD: %.3f ms, %8lld cycles, %.2f cycles per instruction for ONE wave, %.2f per instruction per SIMD, %.2f per instruction per CU\n", name,
               waves_per_simd, ms, cyc, (double)cyc / ((double)iters * instr_per_iter), (double)cyc / (waves_per_simd * iters * instr_per_iter),
               (double)cyc / (waves_per_simd * 4 * iters * instr_per_iter));
    }
}

int main() {
    float* d;
    hipMalloc(&d, (1 << 20) * 8 + 64);
    run<0>("16 SALU", d, 16);
    run<1>("8 VALU + 8 SALU interleaved", d, 16);
    run<2>("8 x (v_cmp->sgpr, v_cndmask)", d, 16);
    run<3>("8 x (v_cmp, saveexec, valu, s_or)", d, 32);
    run<4>("8 x (v_cmp->sgpr, s_bcnt1)", d, 16);
    run<5>("16 dependent v_fma", d, 16);
    run<6>("8 x (ds_read, wait)", d, 8);
    run<7>("8 x (s_cmp, branch not taken, nop)", d, 24);
    return 0;
}
