// band_loader -- the memory side of the tile decomposition in isolation, with the LDS-DMA loader of render_dma.hip:
// a workgroup owns a BW x BH pixel band of a 1024^2 view (frontal camera: texel = pixel), walks the 96 planes of a bf16
// [N, D, 4, 1024, 1024] volume, and per plane moves the band's box ((BW + 16) x (BH + 2) texels x 4 channels) into LDS with
// buffer_load_dwordx4 ... lds, one s_barrier per plane, PF planes ahead.  Nothing else (no taps, no arithmetic).
//   band_loader <BW> <BH> <threads> <PF> [wgs_per_cu_cap] [order]     order 0: XCD-contiguous row-major (the render kernels), 1: plain blockIdx,
//                                                                     2: as 0 with render_band.hip's lane map (64-pixel sub-blocks, 128 lanes each, 11-item line pitch)
// Build: hipcc --offload-arch=gfx950 -O3 band_loader.hip -o bin/band_loader
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(uint32_t voff, const u32x4& rsrc, uint32_t lds_dst, uint64_t mask) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(mask) : "memory");
}

constexpr int S = 1024, D = 96;

template <int NP>  // DMA passes per plane
__global__ void band_kernel(const uint16_t* __restrict__ vol, int bw, int bh, int pf, int lds_pad, int order, int n_bands, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nt = blockDim.x, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_xcd = (n_bands + 7) / 8;
    const int band = order != 1 ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
    if (band >= n_bands) return;
    const int bands_x = S / bw, bands_y = S / bh, per_view = bands_x * bands_y;
    const int n = band / per_view, rem = band - n * per_view, byi = rem / bands_x, bxi = rem - byi * bands_x;
    const int ipl = bw / 8 + 2, rows = bh + 2, items = ipl * rows * 4;
    const int qx0 = bxi * bw - 8, by0 = byi * bh - 1;
    uint32_t g_off[NP];
    uint64_t mask[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        if (order == 2) {  // sub-block map: 128 lanes per 64-pixel sub-block, lines of 11 item slots of which 9 are loaded
            const int sbk = tid >> 7, item = r * 128 + (tid & 127), line = item / 11, col = item - line * 11, row = line >> 2, ch = line & 3;
            const int x = bxi * bw + sbk * 64 - 8 + col * 8, y = by0 + row;
            const bool in_tex = x >= 0 && x < S && y >= 0 && y < S;
            g_off[r] = in_tex ? static_cast<uint32_t>(((static_cast<int64_t>(ch) * S + y) * S + x) * 2) : 0x80000000u;
            mask[r] = __ballot(col < 9 && row < rows);
            continue;
        }
        const int item = r * nt + tid, line = item / ipl, col = item - line * ipl, row = line >> 2, ch = line & 3;
        const int x = qx0 + col * 8, y = by0 + row;
        const bool in_tex = x >= 0 && x < S && y >= 0 && y < S;
        g_off[r] = in_tex ? static_cast<uint32_t>(((static_cast<int64_t>(ch) * S + y) * S + x) * 2) : 0x80000000u;
        mask[r] = __ballot(item < items);
    }
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem));
    const uint32_t buf_bytes = static_cast<uint32_t>(NP * nt * 16);  // (order 2: passes of 128 lanes per sub-block, the same bytes)
    const uint16_t* vbase = vol + static_cast<int64_t>(n) * D * 4 * S * S;
    auto issue = [&](int k) {
        const uint64_t a = reinterpret_cast<uint64_t>(vbase + static_cast<int64_t>(k) * 4 * S * S);
        const u32x4 rsrc = {static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a))), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(a >> 32))) & 0xffffu, 0x80000000u, 0x00020000u};
        uint32_t dst = base + static_cast<uint32_t>(k % (pf + 1)) * buf_bytes + static_cast<uint32_t>(wave) * 1024u;
        int stride = nt * 16;
        if (order == 2) dst = base + static_cast<uint32_t>(k % (pf + 1)) * buf_bytes + static_cast<uint32_t>(wave >> 1) * (NP * 2048u) + static_cast<uint32_t>(wave & 1) * 1024u, stride = 2048;
#pragma unroll
        for (int r = 0; r < NP; ++r) dma16(g_off[r], rsrc, dst + r * stride, mask[r]);
    };
    uint32_t acc = 0;
    if (order == 3) {  // ring of 4 slots, TWO planes per barrier: planes k + 2, k + 3 are in flight while k, k + 1 are read (pf must be 3: 4 buffers)
        issue(0), issue(1);
        for (int k = 0; k < D; k += 2) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (k + 2 < D) issue(k + 2), issue(k + 3);
            acc += *reinterpret_cast<volatile uint32_t*>(smem + (k % 4) * buf_bytes + tid * 4);
            acc += *reinterpret_cast<volatile uint32_t*>(smem + ((k + 1) % 4) * buf_bytes + tid * 4);
        }
        if (acc == 0x12345678u) sink[0] = acc;
        return;
    }
    for (int u = 0; u < pf; ++u) issue(u);
    if (pf == 3) {  // three planes ahead, one barrier per plane, partial waits
        for (int k = 0; k < D; ++k) {
            if (k + 2 < D) { if (NP == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (NP == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (k + pf < D) issue(k + pf);
            acc += *reinterpret_cast<volatile uint32_t*>(smem + (k % (pf + 1)) * buf_bytes + tid * 4);
        }
        if (acc == 0x12345678u) sink[0] = acc;
        return;
    }
    for (int k = 0; k < D; ++k) {
        if (pf == 2 && k + 1 < D) { if (NP == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if (NP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (k + pf < D) issue(k + pf);
        acc += *reinterpret_cast<volatile uint32_t*>(smem + (k % (pf + 1)) * buf_bytes + tid * 4);  // one LDS read per plane: the data is used
    }
    if (acc == 0x12345678u) sink[0] = acc;
    (void)lds_pad;
}

int main(int argc, char** argv) {
    const int bw = argc > 1 ? atoi(argv[1]) : 32, bh = argc > 2 ? atoi(argv[2]) : 16, nt = argc > 3 ? atoi(argv[3]) : 512, pf = argc > 4 ? atoi(argv[4]) : 1;
    const int cap = argc > 5 ? atoi(argv[5]) : 0, order = argc > 6 ? atoi(argv[6]) : 0;
    const int N = 4;
    const size_t elems = (size_t)N * D * 4 * S * S;
    uint16_t* vol; uint32_t* sink; CK(hipMalloc(&vol, elems * 2)); CK(hipMalloc(&sink, 16));
    {  // pseudo-random fill (data-dependent DVFS: do not time zeros)
        std::vector<uint32_t> h(1 << 20); uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 9) & 0x3f7f3f7fu; }
        for (size_t o = 0; o < elems * 2; o += h.size() * 4) CK(hipMemcpy((char*)vol + o, h.data(), std::min(h.size() * 4, elems * 2 - o), hipMemcpyHostToDevice));
    }
    const int ipl = bw / 8 + 2, rows = bh + 2, items = ipl * rows * 4, np = order == 2 ? (44 * rows + 127) / 128 : (items + nt - 1) / nt;
    const int n_bands = N * (S / bw) * (S / bh);
    int lds = (order == 2 ? np * 128 * 16 * (nt / 128) : np * nt * 16) * (pf + 1);
    if (cap > 0) lds = std::max(lds, 160 * 1024 / cap - 512);  // pad the allocation so that at most `cap` workgroups fit a CU
    if (np > 4) { printf("too many passes (%d)\n", np); return 1; }
    auto launch = [&]() {
        const dim3 grid(((n_bands + 7) / 8) * 8), block(nt);
        if (np == 1) band_kernel<1><<<grid, block, lds>>>(vol, bw, bh, pf, 0, order, n_bands, sink);
        else if (np == 2) band_kernel<2><<<grid, block, lds>>>(vol, bw, bh, pf, 0, order, n_bands, sink);
        else if (np == 3) band_kernel<3><<<grid, block, lds>>>(vol, bw, bh, pf, 0, order, n_bands, sink);
        else band_kernel<4><<<grid, block, lds>>>(vol, bw, bh, pf, 0, order, n_bands, sink);
    };
    CK(hipFuncSetAttribute((const void*)band_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)band_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)band_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)band_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 150; ++i) launch();  // clock ramp
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double staged = (double)n_bands * D * items * 16;
    printf("band %3dx%-2d threads %4d PF %d passes %d lds %6d B/wg (%d wg/CU by LDS) order %d: %.3f ms  volume %.2f TB/s  staged %.2f GB (%.2fx) %.2f TB/s  %s\n", bw, bh, nt, pf, np, lds,
           160 * 1024 / lds, order, ms, elems * 2.0 / ms / 1e9, staged / 1e9, staged / (elems * 2.0), staged / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
