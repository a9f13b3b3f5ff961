// LDS update rates on gfx950: ds_add_f32, ds_add_u32, plain read-modify-write, ds_write -- conflict-free addresses
// (lane i -> word i + k*64), 512-thread workgroups, 3 per CU.   hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 2048;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out) {
    __shared__ float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) buf[i] = 0.f;
    __syncthreads();
    const int base = threadIdx.x;  // conflict-free: consecutive lanes, consecutive banks
    float v = 1.0f + threadIdx.x;
#pragma unroll 16
    for (int it = 0; it < kIters; ++it) {
        const int a = (base + 512 * (it & 15)) & 8191;
        if (MODE == 0) atomicAdd(&buf[a], v);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&buf[a]), __float_as_uint(v));
        else if (MODE == 2) buf[a] = buf[a] + v;
        else buf[a] = v;
        v += 1.0f;
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = buf[threadIdx.x] + v;
}

template <int MODE> void run(const char* name, float* d) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int blocks = 256 * 3;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double ops = double(blocks) * 512 * kIters;
    printf("%-18s %.3f ms  %.1f G lane-ops/s  (%.2f ns per wave64 instruction per CU)\n", name, ms, ops / ms / 1e6,
           ms * 1e6 / (ops / 64 / 256));
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 3 * 512 * 4);
    run<0>("ds_add_f32", d);
    run<1>("ds_add_u32", d);
    run<2>("read+add+write", d);
    run<3>("ds_write_b32", d);
    return 0;
}
