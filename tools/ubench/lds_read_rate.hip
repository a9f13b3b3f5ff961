// LDS read/write rates on gfx950 (conflict-free, lane i -> consecutive words): ds_read_b32, ds_read2_b32 (x, x+1),
// ds_read_b64, ds_read_b128, ds_write_b128, and the render kernel's tap pattern (8 ds_read2_b32, rows 224 words apart).
// 512-thread workgroups, 3 per CU.   hipcc --offload-arch=gfx950 -O3 lds_read_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 1024;
constexpr int kWords = 12288;  // 48 KB

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int stride) {
    __shared__ float buf[kWords];
    for (int i = threadIdx.x; i < kWords; i += 512) buf[i] = i;
    __syncthreads();
    float acc = 0.f;
    int base = threadIdx.x % 448;
#pragma unroll 4
    for (int it = 0; it < kIters; ++it) {
        const int a = base + (it & 7) * stride;  // stride is a runtime value: keeps the loads in the loop
        if (MODE == 0) acc += buf[a];
        else if (MODE == 1) acc += buf[a] + buf[a + 1];                                        // ds_read2_b32
        else if (MODE == 2) { const float2 v = *reinterpret_cast<const float2*>(&buf[2 * (a & 2047)]); acc += v.x + v.y; }
        else if (MODE == 3) { const float4 v = *reinterpret_cast<const float4*>(&buf[4 * (a & 1023)]); acc += v.x + v.y + v.z + v.w; }
        else if (MODE == 4) { *reinterpret_cast<float4*>(&buf[4 * (a & 1023)]) = make_float4(acc, acc, acc, acc); acc += 1.f; }
        else {  // 16 taps of one pixel: 4 channels x 2 rows x (x, x+1)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += buf[a + c * 56] + buf[a + c * 56 + 1] + buf[a + 224 + c * 56] + buf[a + 224 + c * 56 + 1];
        }
        base = (base + 1) % 448;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MODE> void run(const char* name, float* d, double bytes_per_lane) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int blocks = 256 * 3 * 4;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, 512);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, 512);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double lanes = double(blocks) * 512 * kIters;
    printf("%-26s %.3f ms  %.2f ns per wave64 iteration per CU  %.0f B/ns/CU\n", name, ms, ms * 1e6 / (lanes / 64 / 256),
           lanes * bytes_per_lane / 256 / (ms * 1e6));
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 3 * 4 * 512 * 4);
    run<0>("ds_read_b32", d, 4);
    run<1>("ds_read2_b32 (x,x+1)", d, 8);
    run<2>("ds_read_b64", d, 8);
    run<3>("ds_read_b128", d, 16);
    run<4>("ds_write_b128", d, 16);
    run<5>("16 taps (8 ds_read2_b32)", d, 64);
    return 0;
}
