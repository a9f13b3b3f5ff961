// Shader clock under load: a VALU-dense loop on N of the 256 CUs, cycle counter (s_memtime) vs wall clock.
// If the time per instruction falls when fewer CUs are busy, the part is power-throttled at full occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0001f, d = 0.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else {
            REP16(asm volatile("v_add_u32 %0, %0, %8\n v_lshl_add_u32 %1, %1, 1, %9\n v_add_u32 %2, %2, %8\n v_lshl_add_u32 %3, %3, 1, %9\n"
                               "v_add_u32 %4, %4, %8\n v_lshl_add_u32 %5, %5, 1, %9\n v_add_u32 %6, %6, %8\n v_lshl_add_u32 %7, %7, 1, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, int blocks, int wps) {
    float* out; long long* cyc;
    const int block = 256 * wps;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, block>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<MODE><<<blocks, block>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double inst = (double)iters * 128 * wps;  // wave-instructions per SIMD
    printf("%-10s CUs=%3d waves/SIMD=%d  %.3f ms  %.3f ns/inst/SIMD  counter %lld ticks -> %.2f ticks/inst, %.2f GHz if ticks are shader cycles\n",
           name, blocks, wps, ms, ms * 1e6 / inst, c, c / inst, c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int b : {16, 64, 128, 256}) { run<0>("v_fma_f32", b, 4); run<1>("int add", b, 4); }
    run<0>("v_fma_f32", 256, 1); run<0>("v_fma_f32", 256, 2);
    return 0;
}
