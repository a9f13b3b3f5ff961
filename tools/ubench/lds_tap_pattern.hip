// The render kernel's tap pattern on gfx950: a wave = 32 pixels x 2 pixel rows; every lane reads, for 4 channels and
// 2 texel rows, the dword pair (x0, x0+1) with ds_read2_b32 from a [row][channel][x] LDS tile.  Varies the x step per
// pixel (texels per pixel), the offset between the two pixel rows and the channel/row pitches, to see which bank
// aliasing costs time.   hipcc --offload-arch=gfx950 -O3 lds_tap_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 1024;
constexpr int kWords = 12288;

__global__ __launch_bounds__(512) void k(float* out, const int* lane_off, int chan_pitch, int row_pitch) {
    __shared__ float buf[kWords];
    for (int i = threadIdx.x; i < kWords; i += 512) buf[i] = i;
    __syncthreads();
    float acc = 0.f;
    const int base = lane_off[threadIdx.x & 63] + (threadIdx.x >> 6) * 8;
#pragma unroll 2
    for (int it = 0; it < kIters; ++it) {
        const int a = base + (it & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            acc += buf[a + c * chan_pitch] + buf[a + c * chan_pitch + 1] + buf[a + row_pitch + c * chan_pitch] + buf[a + row_pitch + c * chan_pitch + 1];
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    float* d;
    int* off;
    hipMalloc(&d, 256 * 3 * 4 * 512 * 4);
    hipMalloc(&off, 64 * 4);
    const int blocks = 256 * 3 * 4;
    struct Cfg { const char* name; float step; int second_row; int chan_pitch, row_pitch; };
    const Cfg cfgs[] = {
        {"64 lanes consecutive (one row)", 1.0f, 32, 56, 224},
        {"2 rows, row offset 224, step 1.0", 1.0f, 224, 56, 224},
        {"2 rows, row offset 224, step 0.9", 0.9f, 224, 56, 224},
        {"2 rows, row offset 224, step 1.05", 1.05f, 224, 56, 224},
        {"2 rows, row offset 240 (=16 mod 32), step 0.9", 0.9f, 240, 60, 240},
        {"2 rows, same texel row (offset 0), step 0.9", 0.9f, 0, 56, 224},
        {"2 rows, row offset 224, step 0.9, chan pitch 57", 0.9f, 228, 57, 228},
    };
    for (const Cfg& c : cfgs) {
        int h[64];
        for (int l = 0; l < 64; ++l) h[l] = static_cast<int>((l & 31) * c.step) + (l >> 5) * c.second_row;
        hipMemcpy(off, h, sizeof(h), hipMemcpyHostToDevice);
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, off, c.chan_pitch, c.row_pitch);
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, off, c.chan_pitch, c.row_pitch);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ms /= 5;
        const double waves = double(blocks) * 8 * kIters;
        printf("%-52s %.3f ms  %.2f ns per wave (16 taps) per CU\n", c.name, ms, ms * 1e6 / (waves / 256));
    }
    return 0;
}
