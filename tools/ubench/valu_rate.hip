// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for a few opcodes,
// at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = 1.0001f, d = 0.5f;
    const f2 pc = {c, c}, pd = {d, d};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // v_fma_f32, 8 independent chains
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 1) {  // v_pk_fma_f32, 4 independent chains (8 fma-lanes per "instruction pair")
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));)
        } else if (MODE == 2) {  // v_mul_f32 / v_add_f32 alternating
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                               "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 3) {  // integer: v_add_u32 / v_lshl_add_u32
            REP16(asm volatile("v_add_u32 %0, %0, %8\n v_lshl_add_u32 %1, %1, 1, %9\n v_add_u32 %2, %2, %8\n v_lshl_add_u32 %3, %3, 1, %9\n"
                               "v_add_u32 %4, %4, %8\n v_lshl_add_u32 %5, %5, 1, %9\n v_add_u32 %6, %6, %8\n v_lshl_add_u32 %7, %7, 1, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 4) {  // v_floor / v_cvt_i32_f32 / v_cndmask mix
            REP16(asm volatile("v_floor_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_floor_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n"
                               "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 6) {  // bf16 unpack the integer way: v_lshlrev_b32 16 / v_and_b32 0xffff0000
            REP16(asm volatile("v_lshlrev_b32 %0, 16, %1\n v_and_b32 %1, 0xffff0000, %2\n v_lshlrev_b32 %2, 16, %3\n v_and_b32 %3, 0xffff0000, %4\n"
                               "v_lshlrev_b32 %4, 16, %5\n v_and_b32 %5, 0xffff0000, %6\n v_lshlrev_b32 %6, 16, %7\n v_and_b32 %7, 0xffff0000, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 7) {  // bf16 unpack with gfx950's conversion: v_cvt_f32_bf16 (low half) / _sdwa WORD_1 (high half)
            REP16(asm volatile("v_cvt_f32_bf16 %0, %1\n v_cvt_f32_bf16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_cvt_f32_bf16 %2, %3\n v_cvt_f32_bf16_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_cvt_f32_bf16 %4, %5\n v_cvt_f32_bf16_sdwa %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_cvt_f32_bf16 %6, %7\n v_cvt_f32_bf16_sdwa %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 8) {  // v_perm_b32 / v_readfirstlane-free integer helpers: v_perm_b32, v_max3_u32, v_cmp+v_cndmask
            REP16(asm volatile("v_perm_b32 %0, %1, %2, %8\n v_max3_u32 %1, %2, %3, %4\n v_perm_b32 %2, %3, %4, %8\n v_max3_u32 %3, %4, %5, %6\n"
                               "v_perm_b32 %4, %5, %6, %8\n v_max3_u32 %5, %6, %7, %0\n v_perm_b32 %6, %7, %0, %8\n v_max3_u32 %7, %0, %1, %2\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 5) {  // v_pk_mul_f32 + v_pk_add_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                               "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
    float* out; long long* cyc;
    const int block = 64 * 4 * waves_per_simd;  // one block per CU: 4 SIMDs x waves
    hipMalloc(&out, 256 * block * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, block>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<MODE><<<256, block>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double inst_per_wave = (double)iters * 16 * 8;
    // s_memtime-style counter runs at 100 MHz on some parts; report wall-clock based rate too
    const double ns_per_inst_per_simd = ms * 1e6 / (inst_per_wave * waves_per_simd);
    printf("%-28s waves/SIMD=%d  time=%.3f ms  -> %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)  [cyclecounter delta %lld]\n",
           name, waves_per_simd, ms, ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4, c);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_mul/add_f32", w); run<3>("v_add_u32/lshl_add", w);
        run<4>("floor/cvt/mov", w); run<5>("v_pk_mul/add_f32", w);
        run<6>("v_lshlrev/v_and (bf16 unpack)", w); run<7>("v_cvt_f32_bf16 (+sdwa)", w); run<8>("v_perm_b32/v_max3_u32", w);
    }
    return 0;
}
