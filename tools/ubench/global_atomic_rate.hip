// Request rate of no-return fp32 global atomics (and plain stores) on gfx950, for the backward's flush (render_backward.hip): every wave adds
// LINES of `len` consecutive floats (one lane per float, lanes >= len idle) into a large zeroed buffer; successive lines of a wave are `pitch`
// floats apart (a box line of the gradient volume: the next texture row); the first line of a wave starts at a pseudo-random offset, aligned
// to `align` floats.  Reports lane-operations/s, wave instructions/s and 64-byte segments/s.
//   hipcc --offload-arch=gfx950 -O3 global_atomic_rate.hip -o bin/global_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kLines = 256;   // lines per wave

template <int MODE>  // 0 atomic add, 1 plain store
__global__ __launch_bounds__(256) void k(float* __restrict__ buf, const uint64_t n_floats, const int len, const int pitch, const int align, const int tiles_w) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    // waves walk the buffer like tiles of a plane: wave -> (tile row, tile column), a tile's lines `pitch` apart, neighbouring tiles `len - 2` apart
    const uint64_t ty = wave / tiles_w, tx = wave % tiles_w;
    uint64_t h = (static_cast<uint64_t>(wave) * 0x9E3779B97F4A7C15ull) >> 40;
    uint64_t base = (ty * kLines * static_cast<uint64_t>(pitch) + tx * static_cast<uint64_t>(len - 2) + (h % 16) * align) % (n_floats - static_cast<uint64_t>(kLines + 1) * pitch - 256);
    base = base / align * align;
    float* p = buf + base + lane;
    const float v = 1.0f + lane;
    if (lane < len) {
#pragma unroll 4
        for (int i = 0; i < kLines; ++i) {
            if (MODE == 0) atomicAdd(p + static_cast<uint64_t>(i) * pitch, v);
            else __builtin_nontemporal_store(v, p + static_cast<uint64_t>(i) * pitch);
        }
    }
}

template <int MODE> void run(const char* name, float* d, uint64_t n, int len, int pitch, int align) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int blocks = 256 * 32;   // 32768 waves
    const int tiles_w = pitch / (len - 2) > 0 ? pitch / (len - 2) : 1;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, n, len, pitch, align, tiles_w);
    hipEventRecord(a);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, n, len, pitch, align, tiles_w);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    const double instr = double(blocks) * 4 * kLines;
    const double lanes = instr * len;
    // segments per line: a line of len floats starting at an address aligned to `align` floats
    const double seg = (align % 16 == 0) ? (len + 15) / 16 : (len * 4.0 + 60.0) / 64.0 + 0.0;
    printf("%-8s len %2d pitch %5d align %2d : %7.3f ms  %7.1f G lane-ops/s  %6.2f G instr/s  ~%5.1f G segments/s  (%.0f GB/s of adds)\n", name, len, pitch, align, ms,
           lanes / ms / 1e6, instr / ms / 1e6, instr * seg / ms / 1e6, lanes * 4 / ms / 1e6);
}

__global__ __launch_bounds__(256) void kmix(float* __restrict__ buf, const uint64_t n_floats, const int mix) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    // lanes 0-15: segment 0 of a line, lanes 16-31: segment 1; lanes 32-63: the same for the wave's next line
    const uint64_t base = (static_cast<uint64_t>(wave) * kLines * 64) % (n_floats - kLines * 64 - 64);
    float* p = buf + base + lane;
    const float v = 1.0f + lane;
    const int seg = (lane >> 4) & 1, line_par = lane >> 5;
#pragma unroll 4
    for (int i = 0; i < kLines; ++i) {
        float* q = p + i * 64;
        bool st;
        if (mix == 0) st = false;
        else if (mix == 1) st = true;
        else if (mix == 2) st = line_par == 1;
        else st = seg == 1;
        if (st) __builtin_nontemporal_store(v, q);
        else atomicAdd(q, v);
    }
}
void run_mix(float* d, uint64_t n, int mix) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int blocks = 256 * 32;
    hipLaunchKernelGGL(kmix, dim3(blocks), dim3(256), 0, 0, d, n, mix);
    hipEventRecord(a);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kmix, dim3(blocks), dim3(256), 0, 0, d, n, mix);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    const double segs = double(blocks) * 4 * kLines * 4;
    const char* names[4] = {"all atomics", "all stores", "atomic LINES | stored LINES", "atomic segment | stored segment in every line"};
    printf("mix %-48s %7.3f ms  %6.1f G segments/s\n", names[mix], ms, segs / ms / 1e6);
}

int main() {
    const uint64_t n = 1ull << 29;  // 2 GiB of floats
    float* d;
    if (hipMalloc(&d, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, n * 4);
    for (int mode = 0; mode < 2; ++mode) {
        auto r = [&](int len, int pitch, int align) { if (mode == 0) run<0>("atomic", d, n, len, pitch, align); else run<1>("store", d, n, len, pitch, align); };
        r(64, 1024, 16);   // full waves, 256-byte aligned lines: 4 segments per instruction
        r(64, 1024, 1);    // ... unaligned: 4-5 segments
        r(34, 1024, 1);    // a box line of the backward: 34 texels, unaligned: 2.6 segments
        r(34, 1024, 16);   // ... aligned: 3 segments
        r(16, 1024, 16);   // one segment per instruction
        r(16, 1024, 1);
        r(8, 1024, 1);
    }
    // MIXES (would a flush that stores the segments only its own tile touches and adds the rest be faster?): every wave issues, per line, one
    // instruction of 16 lanes = one segment; a line is 32 floats = one 128-byte cache line.
    run_mix(d, n, 0);   // all atomics, both segments of every line
    run_mix(d, n, 1);   // all stores
    run_mix(d, n, 2);   // even LINES atomics, odd lines stores (no line gets both)
    run_mix(d, n, 3);   // first segment of every line atomic, second segment stored (every line gets both, no segment does)
    return 0;
}
