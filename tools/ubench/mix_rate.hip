// Issue rate of the 16-bit -> fp32 paths on gfx950 (wave64 instruction per SIMD), all 256 CUs busy:
//   v_fma_mix_f32 (f16 operand converted for free), v_cvt_f32_f16 (+sdwa WORD_1), v_cvt_pkrtz_f16_f32, v_cvt_pk_f16_f32,
//   v_dot2c_f32_f16, against v_fma_f32.  Build: hipcc --offload-arch=gfx950 -O3 mix_rate.hip -o bin/mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 1) {  // acc = w(f32) * t(f16 lo / hi) + acc(f32)
            REP16(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"
                               "v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"
                               "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"
                               "v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 2) {  // v_cvt_f32_f16 low half / sdwa high half
            REP16(asm volatile("v_cvt_f32_f16 %0, %1\n v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_cvt_f32_f16 %2, %3\n v_cvt_f32_f16_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_cvt_f32_f16 %4, %5\n v_cvt_f32_f16_sdwa %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_cvt_f32_f16 %6, %7\n v_cvt_f32_f16_sdwa %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 3) {
            REP16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %8\n v_cvt_pkrtz_f16_f32 %1, %2, %8\n v_cvt_pkrtz_f16_f32 %2, %3, %8\n v_cvt_pkrtz_f16_f32 %3, %4, %8\n"
                               "v_cvt_pkrtz_f16_f32 %4, %5, %8\n v_cvt_pkrtz_f16_f32 %5, %6, %8\n v_cvt_pkrtz_f16_f32 %6, %7, %8\n v_cvt_pkrtz_f16_f32 %7, %0, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 4) {
            REP16(asm volatile("v_cvt_pk_f16_f32 %0, %1, %8\n v_cvt_pk_f16_f32 %1, %2, %8\n v_cvt_pk_f16_f32 %2, %3, %8\n v_cvt_pk_f16_f32 %3, %4, %8\n"
                               "v_cvt_pk_f16_f32 %4, %5, %8\n v_cvt_pk_f16_f32 %5, %6, %8\n v_cvt_pk_f16_f32 %6, %7, %8\n v_cvt_pk_f16_f32 %7, %0, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 5) {
            REP16(asm volatile("v_dot2c_f32_f16 %0, %8, %9\n v_dot2c_f32_f16 %1, %8, %9\n v_dot2c_f32_f16 %2, %8, %9\n v_dot2c_f32_f16 %3, %8, %9\n"
                               "v_dot2c_f32_f16 %4, %8, %9\n v_dot2c_f32_f16 %5, %8, %9\n v_dot2c_f32_f16 %6, %8, %9\n v_dot2c_f32_f16 %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 6) {  // fma_mix with all-f32 operands (plain fma through the mix encoding)
            REP16(asm volatile("v_fma_mix_f32 %0, %8, %9, %0\n v_fma_mix_f32 %1, %8, %9, %1\n v_fma_mix_f32 %2, %8, %9, %2\n v_fma_mix_f32 %3, %8, %9, %3\n"
                               "v_fma_mix_f32 %4, %8, %9, %4\n v_fma_mix_f32 %5, %8, %9, %5\n v_fma_mix_f32 %6, %8, %9, %6\n v_fma_mix_f32 %7, %8, %9, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 8) {  // the integer class of the render kernels: shifts, and, add, perm
            REP16(asm volatile("v_lshlrev_b32 %0, 16, %1\n v_and_b32 %1, 0xffff0000, %2\n v_add_u32 %2, %3, %8\n v_perm_b32 %3, %4, %5, %8\n"
                               "v_lshlrev_b32 %4, 16, %5\n v_and_b32 %5, 0xffff0000, %6\n v_add_u32 %6, %7, %8\n v_perm_b32 %7, %0, %1, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 7) {  // the compositor's mix: 2 fma_f32 : 1 mul : 1 add (reference rate for a blend of classes)
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_mul_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_add_f32 %3, %3, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_mul_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_add_f32 %7, %7, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
    float* out;
    const int block = 64 * 4 * waves_per_simd;
    hipMalloc(&out, 256 * block * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, block>>>(out, 10);
    hipEventRecord(e0);
    k<MODE><<<256, block>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_wave = (double)iters * 16 * 8;
    printf("%-34s waves/SIMD=%d  %.3f ns per wave-instr per SIMD\n", name, waves_per_simd, ms * 1e6 / (inst_per_wave * waves_per_simd));
    hipFree(out);
}

int main() {
    {  // clock ramp: ~0.3 s of VALU work before anything is timed (a GPU coming from idle runs its first ~30 ms at low clocks)
        float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
        for (int i = 0; i < 60; ++i) k<0><<<256, 1024>>>(out, 2000);
        hipDeviceSynchronize(); hipFree(out);
    }
    for (int w : {2, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_fma_mix_f32 (f16 lo/hi src1)", w); run<6>("v_fma_mix_f32 (all f32)", w);
        run<2>("v_cvt_f32_f16 (+sdwa WORD_1)", w); run<3>("v_cvt_pkrtz_f16_f32", w); run<4>("v_cvt_pk_f16_f32", w);
        run<5>("v_dot2c_f32_f16", w); run<7>("fma/mul/fma/add f32", w); run<8>("v_lshlrev/v_and/v_add_u32/v_perm", w);
    }
    return 0;
}
