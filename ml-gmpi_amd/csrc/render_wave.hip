// render_wave.hip -- GMPI_VARIANT_WAVE, the strip kernel: wave-private texel boxes, four pixels per lane, plane split.
//
// The counters of round 1's tile kernel (render_lds.hip; profiles/r02_issue_budget.txt) show two walls on MI355X:
//   * VALU issue: 98 vector instructions per wave and plane of 64 pixels, 40 % of them integer-class (16 shift/and per pixel to
//     unpack 16-bit taps, loader, addresses) at 1.73 ns per instruction and SIMD against 1.24 ns for fp32 mul/add/fma;
//   * the vector memory path: it holds a bounded number of wave-level load instructions per CU, the address unit takes a quad
//     of lanes per clock whatever the width, and a 32-pixel-wide tile uses a third of each 128-byte line of a 16-bit volume.
// This kernel is laid out for those two walls:
//   * a wavefront owns a 32x8 pixel STRIP (4 pixels per lane: rows y, y+2, y+4, y+6) and stages the texel box of the strip
//     privately: its own LDS region, its own prefetch registers, no workgroup barrier on the data path (LDS operations of
//     one wave execute in order, so the box of plane k+1 may overwrite plane k's without any wait); 4 strips side by side per
//     workgroup, 12 waves per CU (168 VGPRs, 13 KB of LDS each); launches of 1025-2048 strips run an 8-waves-per-CU build
//     (up to 256 VGPRs, no scratch), smaller ones split the planes of a strip over 3 or 6 waves (SPLIT, below);
//   * texels are stored in LDS as whole RGBA texels, converted once per staged texel (~1.5 per pixel) by the loader instead of
//     16x per pixel by the compositor: 54 vector instructions per pixel and plane (the exact coordinate chain 17, v_fract /
//     v_cvt_flr 4, weights 6, tap address 5, bilinear 16, blend 9, minus shared ones).  Two texel formats:
//       - fp32 RGBA, 16 bytes (fp32 volumes; the strict-order mode): a pixel's taps are four ds_read_b128;
//       - fp16 RGBA, 8 bytes (HALF: bf16 / fp16 volumes in default mode): half the LDS bytes and tap registers, boxes twice
//         as large fit (tilted cameras need no half-strip pass up to ~0.5 rad of yaw), and the fp32 conversion happens inside
//         the bilinear FMAs (v_fma_mix_f32).  bf16 -> fp16 is exact for 2^-17 <= |v| <= 65280 (smaller values are truncated
//         by < 2^-24, far below the 1e-5 bar; a plane holding a larger value, inf or NaN is detected by the range-check
//         maximum the loader keeps anyway and leaves the staged loop through the direct gather: correct for any input).
//         Measured (profiles/r02_variants.txt): LDS busy -17 %, waits for LDS data -80 %, bench poses 1.037 vs 1.034 ms for
//         the tile kernel, 1.18 vs 1.07 ms at tilted poses;
//   * 16-byte loads (8 half-precision / 4 fp32 texels of each of the four channel images per lane) at dword alignment:
//     4 load instructions per wave and plane, boxes as tight as the footprint (32 texels per row);
//   * LDS rows are padded by one (fp32 texels) or two (fp16 texels) slots per 8 texture columns -- the first holds a copy of
//     the next texel -- which makes the 16-byte stores of a wave conflict-free (lanes 144 / 80 bytes apart: 8 consecutive
//     lanes hit 8 different 16-byte bank groups) while the compositor still reads slot and slot + 1;
//   * everything that is uniform per plane (box origin, texture bounds, LDS address constant, plane constants and their
//     correctly rounded reciprocals) is computed by one lane per plane and corner, 16 planes per round, into a 32-entry ring
//     in the wave's LDS region and read back as two 16-byte broadcasts per plane, one plane ahead of their use;
//   * texture bounds along y are left to the buffer range check (one descriptor per channel image -- on gfx950 the scalar
//     offset of a buffer load takes part in the range check, so it cannot carry the channel): rows above/below the image
//     read as the zeros F.grid_sample's padding wants; only the x test is explicit, once per plane.
// The loader's lane -> (row, item) map is fixed per wave from the largest box over all planes, so the number of passes NP
// is a compile-time constant of the plane loop (one instance per NP = 1..3).  A strip whose boxes do not fit its LDS region
// (tilted views) is rendered as its upper and lower half in two passes; a half that still does not fit (textures much finer
// than the image, degenerate rays) takes the direct gather -- same arithmetic.
//
// Arithmetic contract: STRICT = op-for-op oracle/mpi_oracle.c (IEEE divisions, no FMA) -> bit-identical results.
// Default: the three divisions through correctly rounded reciprocals (div_by_recip, gmpi_device.hpp), FMA blend,
// depth as sum(w*s)*dot.
// Reference: gmpi/core/mpi.py:74-99 (chain), :136-142 (grid_sample), :411-434 (composite).
#include "gmpi_device.hpp"

#include <type_traits>

namespace gmpi {

constexpr int kSW = 32, kSH = 8;  // pixels per strip (one wavefront)
constexpr int kPX = 4;            // pixels per lane
constexpr int kRing = 32;         // planes in the per-wave table ring
constexpr int kGroup = 16;        // ... refilled this many at a time (one round of the quad-parallel box code)
constexpr int kEntry = 32;        // bytes per table entry (two 16-byte broadcasts)
// LDS bytes per wave: 16 waves per CU (4 per SIMD, <= 128 VGPRs) or 12 (3 per SIMD, <= 168 VGPRs) use all 160 KB
constexpr int wave_lds(int waves_per_simd) { return waves_per_simd >= 4 ? 10240 : waves_per_simd == 3 ? 13312 : 20480; }
// texel slots of a wave's box: 16 bytes (fp32 RGBA) or 8 bytes (fp16 RGBA, HALF); 16 bytes of front padding
constexpr int box_slots(int waves_per_simd, int slot_bytes) { return (wave_lds(waves_per_simd) - kRing * kEntry - 16) / slot_bytes; }
constexpr int kMaxNP = 3;
constexpr float kBoxSlack = 1.0f / 64;  // slack on the corner-derived box (fp32 error of ix is < 1e-3 texel)

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4_t lds_f32x4;
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;
typedef __attribute__((address_space(3))) u32x2_t lds_u32x2;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// One loader item = 16 bytes of one channel row = TPI texels (8 half-precision / 4 fp32): a wave-level load costs the
// texture-address unit 16 cycles whatever its width (4 lanes per clock), so the loader moves 16 bytes per lane and load.
// A lane stages the four channels of its TPI texels (4 loads) and writes TPI fp32 RGBA texels to LDS.
template <typename TexT> struct ItemIO;
template <> struct ItemIO<float> {
    static constexpr int kTPI = 4;
    static __device__ __forceinline__ void unpack(const u32x4_t& v, float (&o)[4]) {
        o[0] = __uint_as_float(v.x), o[1] = __uint_as_float(v.y), o[2] = __uint_as_float(v.z), o[3] = __uint_as_float(v.w);
    }
    // [0,1] <=> bit pattern <= 0x3f800000 (non-negative floats order like unsigned ints); -0.0 is legal (cold re-test)
    static __device__ __forceinline__ uint32_t fold(uint32_t m, const u32x4_t& v) { return max(max(m, v.x), max(max(v.y, v.z), v.w)); }
    static __device__ __forceinline__ bool suspicious(uint32_t m) { return m > 0x3f800000u; }
    static __device__ __forceinline__ bool bad(const u32x4_t& v) {
        auto ok = [](uint32_t e) { return e <= 0x3f800000u || e == 0x80000000u; };
        return !(ok(v.x) && ok(v.y) && ok(v.z) && ok(v.w));
    }
    static __device__ __forceinline__ void half_texels(const u32x4_t (&)[4], u32x4_t (&)[4]) {}  // (fp32 volumes are never staged as fp16)
    static __device__ __forceinline__ bool unsafe_half(const u32x4_t&) { return false; }
};
template <uint32_t ONE> struct ItemIO16 {
    static constexpr int kTPI = 8;
    static __device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b) {  // v_pk_max_u16
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
    }
    static __device__ __forceinline__ uint32_t fold(uint32_t m, const u32x4_t& v) { return pkmax(pkmax(m, v.x), pkmax(pkmax(v.y, v.z), v.w)); }
    static __device__ __forceinline__ bool suspicious(uint32_t m) { return max(m & 0xffffu, m >> 16) > ONE; }
    static __device__ __forceinline__ bool bad(const u32x4_t& v) {
        auto ok = [](uint32_t h) { return h <= ONE || h == 0x8000u; };
        auto ok2 = [&](uint32_t d) { return ok(d & 0xffffu) && ok(d >> 16); };
        return !(ok2(v.x) && ok2(v.y) && ok2(v.z) && ok2(v.w));
    }
};
// HALF: the 8 texels of an item as fp16 RGBA (two dwords per texel: R|G<<16, B|A<<16) from the four channel vectors.
//   bf16 -> fp16 through fp32 (v_cvt_pkrtz_f16_f32; exact for 2^-17 <= |v| <= 65280, below that truncated by < 2^-24);
//   values that fp16 cannot hold are caught by unsafe_half() and that plane is composited by the direct gather.
template <> struct ItemIO<bf16_t> : ItemIO16<0x3f80u> {
    static __device__ __forceinline__ void half_texels(const u32x4_t (&L)[4], u32x4_t (&o)[4]) {  // o[i] = texels 2i (.xy), 2i+1 (.zw)
        auto pk = [](uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(a), __uint_as_float(b))); };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i].x = pk(L[0][i] << 16, L[1][i] << 16), o[i].y = pk(L[2][i] << 16, L[3][i] << 16);
            o[i].z = pk(L[0][i] & 0xffff0000u, L[1][i] & 0xffff0000u), o[i].w = pk(L[2][i] & 0xffff0000u, L[3][i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ bool unsafe_half(const u32x4_t& v) {  // |v| > 65280 (the largest bf16 below fp16's maximum), inf, NaN
        auto big = [](uint32_t d) { return (d & 0x7fffu) > 0x477fu || ((d >> 16) & 0x7fffu) > 0x477fu; };
        return big(v.x) || big(v.y) || big(v.z) || big(v.w);
    }
    static __device__ __forceinline__ void unpack(const u32x4_t& v, float (&o)[8]) {
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) o[2 * i] = __uint_as_float(d[i] << 16), o[2 * i + 1] = __uint_as_float(d[i] & 0xffff0000u);
    }
};
template <> struct ItemIO<f16_t> : ItemIO16<0x3c00u> {
    static __device__ __forceinline__ void half_texels(const u32x4_t (&L)[4], u32x4_t (&o)[4]) {  // verbatim: two v_perm_b32 per texel
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i].x = __builtin_amdgcn_perm(L[1][i], L[0][i], 0x05040100u), o[i].y = __builtin_amdgcn_perm(L[3][i], L[2][i], 0x05040100u);
            o[i].z = __builtin_amdgcn_perm(L[1][i], L[0][i], 0x07060302u), o[i].w = __builtin_amdgcn_perm(L[3][i], L[2][i], 0x07060302u);
        }
    }
    static __device__ __forceinline__ bool unsafe_half(const u32x4_t&) { return false; }
    static __device__ __forceinline__ void unpack(const u32x4_t& v, float (&o)[8]) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const h2 h = __builtin_bit_cast(h2, d[i]);
            o[2 * i] = static_cast<float>(h.x), o[2 * i + 1] = static_cast<float>(h.y);
        }
    }
};

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}

// SPLIT > 1 (J1; launches that under-fill the chip): SPLIT wavefronts share a strip, each composites a
// contiguous range of planes front to back, and the partial (C, Z, T) triples are merged IN PLANE ORDER with the associative
// operator (C1, Z1, T1) (+) (C2, Z2, T2) = (C1 + T1 C2, Z1 + T1 Z2, T1 T2) of mpi.py:421-434 -- through LDS and one workgroup
// barrier at the very end (the partials live in different wavefronts).  Default mode only: the strict-order mode keeps the
// reference's sequential association (the two differ by rounding, ~1e-7).
template <typename TexT, bool AC, bool STRICT, int WPB, int WPS, int SPLIT, bool HALF, int GRPSEL>
__global__ __launch_bounds__(WPB * SPLIT * 64, WPS) void render_wave_kernel(const KParams p, const int tiles_x, const int tiles_y,
                                                                   const int n_tiles) {
    using IO = ItemIO<TexT>;
    constexpr int TPI = IO::kTPI;
    constexpr int ES = static_cast<int>(sizeof(TexT));
    constexpr int SS = HALF ? 8 : 16;  // bytes per texel slot in LDS
    constexpr int PADS = HALF ? 2 : 1; // padding slots per 8 texture columns (lanes 80 / 144 bytes apart: conflict-free 16-byte stores)
    constexpr int kWaveLds = wave_lds(WPS), kBoxSlots = box_slots(WPS, SS);
    static_assert(SPLIT == 1 || !STRICT, "plane split changes the association of the composite");
    static_assert(!HALF || (!STRICT && sizeof(TexT) == 2), "fp16 texels in LDS: 16-bit volumes, default mode");
    __shared__ __attribute__((aligned(16))) unsigned char smem[WPB * SPLIT * kWaveLds];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);

    // ---- blockIdx -> band of strips: XCD x (blockIdx % 8) gets the contiguous run [x*per, (x+1)*per) of bands, so bands that
    //      share halo rows meet in one L2; views that share one MPI are interleaved per band position (render_lds.hip) ----
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile_id = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile_id >= n_tiles) return;
    const int tiles_per_view = tiles_x * tiles_y;
    int n, trem;
    if (p.view_to_mpi == nullptr && p.views_per_mpi > 1) {
        const int group = tile_id / (tiles_per_view * p.views_per_mpi);
        const int first = group * p.views_per_mpi, size = min(p.views_per_mpi, p.N - first);
        const int r = tile_id - first * tiles_per_view;
        trem = r / size;
        n = first + (r - trem * size);
    } else {
        n = tile_id / tiles_per_view;
        trem = tile_id - n * tiles_per_view;
    }
    if (view_gated_out(p, n)) return;  // (AUTO: this view is the band kernel's)
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;

    uint32_t bad = 0;
    const int m = view_mpi(p, n, bad);  // (an index outside [0, M) is clamped and reported)
    const int D = p.D, Ht = p.Ht, Wt = p.Wt, H = p.H, W = p.W;
    const float* __restrict__ dhw = p.dhw + static_cast<int64_t>(m) * D * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const float zx = p.z_dir[3 * n + 0], zy = p.z_dir[3 * n + 1], zz = p.z_dir[3 * n + 2];
    const float cx = AC ? static_cast<float>(Wt - 1) * 0.5f : static_cast<float>(Wt);
    const float cy = AC ? static_cast<float>(Ht - 1) * 0.5f : static_cast<float>(Ht);
    const bool check_range = (p.flags & (1u << 3)) != 0;
    const bool check_last = (p.flags & (1u << 2)) != 0;
    const int64_t HW = static_cast<int64_t>(H) * W;
    const float* __restrict__ rdv = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    const int64_t s_chan = p.s_chan, s_row = p.s_row, s_plane = p.s_plane;

    if (p.status != nullptr && trem == 0 && threadIdx.x == 0) {  // mpi.py:70-72, once per view
        const float ez0 = p.eye_pos[2];
        bool behind = false;
        for (int k = 0; k < D; ++k) behind |= !(dhw[3 * k] >= ez0);
        if (behind) atomicOr(p.status, 4u);
    }

    // ---- this wave's strip and plane range (no workgroup barrier on the data path; with SPLIT > 1 one at the very end, so a
    //      strip past the right edge stays as a ghost that does nothing but reach it) ----
    const int part = SPLIT == 1 ? 0 : wv / WPB;  // which range of planes
    int sx0 = (txi * WPB + (wv - part * WPB)) * kSW;
    const int sy0 = tyi * kSH;
    const bool ghost = sx0 >= W;
    if (ghost) {
        if (SPLIT == 1) return;
        sx0 = 0;
    }
    const int planes_per_part = (D + SPLIT - 1) / SPLIT;
    const int k_begin = SPLIT == 1 ? 0 : min(part * planes_per_part, D), k_end = SPLIT == 1 ? D : min(k_begin + planes_per_part, D);
    const int lxp = lane & 31, lyp = lane >> 5;
    const int pxx = min(sx0 + lxp, W - 1);
    float rx[kPX], ry[kPX], rz[kPX], rrz[kPX];
    auto pixel_index = [&](int j) -> int64_t { return static_cast<int64_t>(min(sy0 + lyp + 2 * j, H - 1)) * W + pxx; };
#pragma unroll
    for (int j = 0; j < kPX; ++j) {
        const int64_t q = pixel_index(j);
        rx[j] = rdv[q], ry[j] = rdv[HW + q], rz[j] = rdv[2 * HW + q];
        rrz[j] = 1.0f / rz[j];  // correctly rounded, hoisted out of the plane loop (div_by_recip)
    }
    Accum A[kPX];
    auto ray_dot = [&](int j) -> float {  // einsum("nchw,nc->nhw") mpi.py:149
        float d = rx[j] * zx;
        d = d + ry[j] * zy;
        d = d + rz[j] * zz;
        return d;
    };

    lds_u32x4* const tab = reinterpret_cast<lds_u32x4*>(
        static_cast<uintptr_t>(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem)) +
                               static_cast<uint32_t>(wv * kWaveLds)));
    const uint32_t box_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab)) + kRing * kEntry + 16;  // (front padding slot)
    const int qc = lane & 3, qt = lane >> 2;  // box code: lane = (plane slot, corner)

    // The strip is rendered whole if the boxes of all planes fit the wave's LDS region (att 0); otherwise as its upper and
    // lower half, one after the other (att 1, 2: pixel pairs {0,1} / {2,3} of every lane, i.e. rows 0-3 / 4-7 -- tilted
    // cameras shear the box); a half that still does not fit (textures much finer than the image, degenerate rays) takes
    // the direct gather.  Everything below is wave-uniform control flow.
#pragma unroll 1
    for (int att = 0; att < 3 && !ghost && k_begin < k_end; ++att) {
        const int pmask = att == 0 ? 3 : att;  // pixel pairs this pass composites
        const int y_lo = min(sy0 + (att == 2 ? kSH / 2 : 0), H - 1);
        const int y_hi = min(sy0 + (att == 1 ? kSH / 2 - 1 : kSH - 1), H - 1);

        // ---- per-plane texel box from the four corner pixels: 16 planes per round, one corner ray per lane, min/max across
        //      the quad.  The box only has to CONTAIN the taps (slack 1/64 texel), so the corner coordinates go through
        //      hardware reciprocals (1 ulp) instead of IEEE divisions: error < 1e-3 texel ----
        float crx, cry, crz;
        {
            const int64_t q = static_cast<int64_t>((qc & 2) ? y_hi : y_lo) * W + ((qc & 1) ? min(sx0 + kSW - 1, W - 1) : sx0);
            crx = rdv[q], cry = rdv[HW + q], crz = rdv[2 * HW + q];
        }
        struct Box { int bx0, bx1, by0, nr; float zdiff, hw, hh; };  // columns [bx0, bx1], rows [by0, by0 + nr); bx1 < bx0: not finite
        const float crcp = __builtin_amdgcn_rcpf(crz);
        auto box_of = [&](int k) -> Box {  // k is the same in the four lanes of a quad
            Box b;
            const float d = dhw[3 * k + 0];
            b.hh = dhw[3 * k + 1] * 0.5f, b.hw = dhw[3 * k + 2] * 0.5f;  // exact halves: (2x)/w == x/(w/2)
            b.zdiff = d - ez;
            float ix, iy, s;
            plane_coord_recip<AC>(b.zdiff, b.hw, b.hh, __builtin_amdgcn_rcpf(b.hw), __builtin_amdgcn_rcpf(b.hh), ex, ey, crx, cry, crz, crcp,
                                  cx, cy, ix, iy, s);
            const float inf = __builtin_inff();
            // |coordinate| < 16384 texels or the strip takes the gather path: the corner chain goes through v_rcp (4e-7 relative: beyond ~45k
            // texels the error would exceed the box slack of 1/64 texel), and the box origin offset is formed in 32 bits.  NaN too.
            if (!(fabsf(ix) < 16384.0f)) ix = inf;
            if (!(fabsf(iy) < 16384.0f)) iy = inf;
            float mnx = ix, mxx = ix, mny = iy, mxy = iy;
#pragma unroll
            for (int o = 1; o < 4; o <<= 1) {
                mnx = fminf(mnx, __shfl_xor(mnx, o)), mxx = fmaxf(mxx, __shfl_xor(mxx, o));
                mny = fminf(mny, __shfl_xor(mny, o)), mxy = fmaxf(mxy, __shfl_xor(mxy, o));
            }
            if (mxx < inf && mxy < inf) {
                b.bx0 = static_cast<int>(floorf(mnx - kBoxSlack)), b.bx1 = static_cast<int>(floorf(mxx + kBoxSlack)) + 1;
                b.by0 = static_cast<int>(floorf(mny - kBoxSlack));
                b.nr = static_cast<int>(floorf(mxy + kBoxSlack)) + 1 - b.by0 + 1;
            } else {
                b.bx0 = b.by0 = 0, b.bx1 = -1, b.nr = 1;
            }
            return b;
        };

        // ---- loader map of the wave: lanes per row LPR = widest box (items), rows per pass, passes ----
        // Item alignment: a 16-byte load needs dword alignment only, so the box origin of 16-bit volumes is a multiple of 2
        // texels (tight boxes: ~32 instead of 40-48 texels per row) -- unless some plane's box reaches the left/right border
        // of the texture: then the origins of this wave are multiples of the item width, so that an item lies entirely
        // inside or outside the texture (zeros padding by one compare per item).  fp32 volumes: multiples of 4 texels.
        constexpr int kLooseAlign = ES == 2 ? 2 : TPI;
        int ni_loose = 1, ni_tight = 1, nr_max = 1, touch = 0;
        for (int g = k_begin; g < k_end; g += 16) {
            const Box b = box_of(min(g + qt, k_end - 1));
            if (b.bx1 < b.bx0) ni_loose = ni_tight = 1 << 20;
            ni_loose = max(ni_loose, (b.bx1 - (b.bx0 & ~(kLooseAlign - 1))) / TPI + 1);
            ni_tight = max(ni_tight, (b.bx1 - (b.bx0 & ~(TPI - 1))) / TPI + 1);
            nr_max = max(nr_max, b.nr);
            touch |= (b.bx0 < 0 || b.bx1 + TPI > Wt) ? 1 : 0;
        }
        const int align = wave_max(touch) ? TPI : kLooseAlign;
        const int LPR = wave_max(align == TPI ? ni_tight : ni_loose);
        const int NR = wave_max(nr_max);
        const bool wide = LPR > 64;
        const int rpp_max = wide ? 1 : 64 / LPR;
        const int NP = (NR + rpp_max - 1) / rpp_max;
        const int RPP = (NR + NP - 1) / NP;  // rows per pass, balanced
        // LDS box: texel x (texture column) of a row sits in slot x + x / 8 - const: one slot of padding per 8 texture
        // columns keeps the 16-byte stores of a wave conflict-free (the store path has 32 banks; with 8-texel items lanes
        // are 9 slots apart and rows 9 * LPR, so the slot of lane l is congruent to l modulo 8); the padding slot holds a
        // copy of the texel after it, so the compositor still reads slot and slot + 1.  (The copy of a row's first texel
        // lands in the last slot of the row above, which that phase leaves unused.)  Rows past the tallest box are not staged.
        const int pitch = TPI == 8 ? (8 + PADS) * LPR : TPI * LPR + (TPI * LPR + 7) / 8 + 1;  // slots per box row
        const bool fit = !wide && NP <= kMaxNP && pitch * NR <= kBoxSlots;
#ifdef GMPI_TUNE
        if (p.status != nullptr && lane == 0) {  // debug: passes per strip / strips that end in the gather
            if (p.flags & (1u << 27)) {  // sums over the strips: extra passes, lanes per row, rows
                if (fit) atomicAdd(p.status + 1, static_cast<uint32_t>(NP - 1)), atomicAdd(p.status + 2, static_cast<uint32_t>(LPR)), atomicAdd(p.status + 3, static_cast<uint32_t>(NR));
            } else {
                if (att == 1) atomicAdd(p.status + 2, 1u);
                if (att > 0 && !fit) atomicAdd(p.status + 3, 1u);
            }
        }
#endif
        if (att == 0 && !fit) continue;  // try the halves

        // ring entries of the planes [k0, k0 + 16) (slot k % 32): lo = {box origin (bytes from the plane's channel-0 image),
        // in-texture item columns clo | ncol << 8 | phase << 16 | rows << 24, LDS address constant, zdiff};
        // hi = {w/2, h/2, RN(1/(w/2)), RN(1/(h/2))}
        auto fill = [&](int k0) {
            const int k = k0 + qt;
            const Box b = box_of(min(k, k_end - 1));
            // correctly rounded reciprocals of the plane's half extents (div_by_recip needs RN(1/d)): one IEEE division per lane
            const float rv = 1.0f / ((qc & 1) ? b.hh : b.hw);
            const float ro = __shfl_xor(rv, 1);
            if (qc == 0 && k < k_end) {
                const int qx0 = b.bx0 & ~(align - 1), ni = (b.bx1 - qx0) / TPI + 1;
                const int clo = min(max(-qx0 / TPI, 0), ni), chi = min(max((Wt - qx0) / TPI, 0), ni);  // inside the texture AND the plane's box
                const int origin = (b.by0 * static_cast<int>(s_row) + qx0) * ES;
                const uint32_t cst = box_addr - static_cast<uint32_t>(SS) * static_cast<uint32_t>(b.by0 * pitch + qx0 + PADS * (qx0 >> 3));
                const int phase = qx0 & 7;  // position of the box origin inside its 8-column block (where the padding slots fall)
                u32x4_t lo, hi;
                lo.x = static_cast<uint32_t>(origin), lo.y = static_cast<uint32_t>(clo | (chi - clo) << 8 | phase << 16 | b.nr << 24), lo.z = cst,
                lo.w = __float_as_uint(b.zdiff);
                hi.x = __float_as_uint(b.hw), hi.y = __float_as_uint(b.hh), hi.z = __float_as_uint(rv), hi.w = __float_as_uint(ro);
                tab[2 * (k & (kRing - 1))] = lo;
                tab[2 * (k & (kRing - 1)) + 1] = hi;
            }
        };

        // ---- compositing of one plane, taps from the wave's LDS box.  Two pixels at a time: both coordinate chains, then all 8
        //      tap reads, then the arithmetic -- half as many waits for LDS data as one pixel at a time, at 32 registers of taps
        //      (all four pixels at once, 64 registers of taps, measured slower: 1.32 vs 1.19 ms) ----
        auto composite = [&](const u32x4_t& lo, const u32x4_t& hi, auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;  // all four pixels of a lane (compile time: no per-pixel branches)
            const float zdiff = __uint_as_float(lo.w), hw = __uint_as_float(hi.x), hh = __uint_as_float(hi.y);
            const float rw = __uint_as_float(hi.z), rh = __uint_as_float(hi.w);
            const uint32_t row_bytes = static_cast<uint32_t>(SS) * static_cast<uint32_t>(pitch);
            const uint32_t cst = lo.z, cst2 = lo.z + row_bytes;
            auto tap_ptr = [&](uint32_t base, int idx) {
                return reinterpret_cast<const lds_f32x4*>(static_cast<uintptr_t>(base + 16u * static_cast<uint32_t>(idx)));
            };
            // pixels whose taps are in flight together: 2 (32 registers of fp32 taps, 16 of fp16 taps; 4 pixels measured slower with
            // either texel type), 1 in the 128-VGPR build
            constexpr int GRP = GRPSEL > 0 ? GRPSEL : WPS >= 4 ? 1 : 2;
#pragma unroll
            for (int jp = 0; jp < kPX; jp += GRP) {
                if (!FULL && GRP <= 2 && !(pmask & (1 << (jp / 2)))) continue;
                if constexpr (STRICT) {
#pragma unroll
                    for (int j = jp; j < jp + GRP; ++j) {
                        float s, ix, iy, u, v, smp[4];
                        plane_coord<AC>(zdiff, hh + hh, hw + hw, ex, ey, rx[j], ry[j], rz[j], cx, cy, ix, iy, s, u, v);
                        const Footprint f = footprint(ix, iy, Ht, Wt);
                        const int lx = static_cast<int>(floorf(ix));
                        const int idx = __mul24(static_cast<int>(floorf(iy)), pitch) + lx + (lx >> 3);
                        const lds_f32x4 *t0 = tap_ptr(cst, idx), *t1 = tap_ptr(cst2, idx);
                        const f32x4_t q_nw = t0[0], q_ne = t0[1], q_sw = t1[0], q_se = t1[1];
#pragma unroll
                        for (int c = 0; c < 4; ++c) smp[c] = bilerp<true>(q_nw[c], q_ne[c], q_sw[c], q_se[c], f);
                        blend<true>(A[j], smp[0], smp[1], smp[2], smp[3], s, ray_dot(j));
                    }
                } else {
                    using TapT = std::conditional_t<HALF, u32x2_t, f32x4_t>;
                    float s[GRP], fx[GRP], fy[GRP];  // (the four weights are formed after the reads: 2 live values per pixel, not 4)
                    TapT q_nw[GRP], q_ne[GRP], q_sw[GRP], q_se[GRP];
#pragma unroll
                    for (int h = 0; h < GRP; ++h) {
                        const int j = jp + h;
                        if (!FULL && GRP > 2 && !(pmask & (1 << (j / 2)))) continue;
                        float ix, iy;
                        plane_coord_recip<AC>(zdiff, hw, hh, rw, rh, ex, ey, rx[j], ry[j], rz[j], rrz[j], cx, cy, ix, iy, s[h]);
                        fx[h] = __builtin_amdgcn_fractf(ix), fy[h] = __builtin_amdgcn_fractf(iy);
                        const int lx = floor_to_int(ix);
                        if constexpr (HALF) {
                            // slot = y * pitch + x + 2 * (x >> 3): v_ashrrev, v_lshl_add, v_mad_i32_i24, v_lshl_add, v_add
                            const int idx = __mul24(floor_to_int(iy), pitch) + (((lx >> 3) << 1) + lx);
                            const uint32_t a0 = cst + (static_cast<uint32_t>(idx) << 3);
                            const lds_u32x2 *t0 = reinterpret_cast<const lds_u32x2*>(static_cast<uintptr_t>(a0)),
                                            *t1 = reinterpret_cast<const lds_u32x2*>(static_cast<uintptr_t>(a0 + row_bytes));
                            q_nw[h] = t0[0], q_ne[h] = t0[1], q_sw[h] = t1[0], q_se[h] = t1[1];
                        } else {
                            const int idx = __mul24(floor_to_int(iy), pitch) + lx + (lx >> 3);
                            const lds_f32x4 *t0 = tap_ptr(cst, idx), *t1 = tap_ptr(cst2, idx);
                            q_nw[h] = t0[0], q_ne[h] = t0[1], q_sw[h] = t1[0], q_se[h] = t1[1];
                        }
                    }
#pragma unroll
                    for (int h = 0; h < GRP; ++h) {
                        const int j = jp + h;
                        if (!FULL && GRP > 2 && !(pmask & (1 << (j / 2)))) continue;
                        const float gx = 1.0f - fx[h], gy = 1.0f - fy[h];
                        const float w_nw = gx * gy, w_ne = fx[h] * gy, w_sw = gx * fy[h], w_se = fx[h] * fy[h];
                        float smp[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if constexpr (HALF) {
                                // fp16 -> fp32 inside the FMA (v_fma_mix_f32: the conversion is exact and free of a separate instruction)
                                auto pick = [&](const u32x2_t& q) -> float {
                                    const f16x2_t hv = __builtin_bit_cast(f16x2_t, c < 2 ? q.x : q.y);
                                    return static_cast<float>((c & 1) ? hv.y : hv.x);
                                };
                                float acc = __builtin_fmaf(pick(q_nw[h]), w_nw, 0.0f);
                                acc = __builtin_fmaf(pick(q_ne[h]), w_ne, acc);
                                acc = __builtin_fmaf(pick(q_sw[h]), w_sw, acc);
                                smp[c] = __builtin_fmaf(pick(q_se[h]), w_se, acc);
                            } else {
                                float acc = q_nw[h][c] * w_nw;
                                acc = __builtin_fmaf(q_ne[h][c], w_ne, acc);
                                acc = __builtin_fmaf(q_sw[h][c], w_sw, acc);
                                smp[c] = __builtin_fmaf(q_se[h][c], w_se, acc);
                            }
                        }
                        const float w = smp[3] * A[j].T;
                        A[j].r = __builtin_fmaf(w, smp[0], A[j].r);
                        A[j].g = __builtin_fmaf(w, smp[1], A[j].g);
                        A[j].b = __builtin_fmaf(w, smp[2], A[j].b);
                        A[j].z = __builtin_fmaf(w, s[h], A[j].z);
                        float om = 1.0f - smp[3];  // (T - w would cancel behind nearly opaque planes: 1 - a is exact, a*T is not)
                        om = om + 1e-10f;
                        A[j].T = A[j].T * om;
                    }
                }
            }
        };
        // one plane straight from global memory (boxes that do not fit; planes holding a value fp16 cannot represent)
        auto composite_gather = [&](int k) __attribute__((always_inline)) {
            const float d = dhw[3 * k + 0], ph = dhw[3 * k + 1], pw = dhw[3 * k + 2];
            const float zdiff = d - ez;
#pragma unroll
            for (int j = 0; j < kPX; ++j) {
                if (!(pmask & (1 << (j / 2)))) continue;
                float ix, iy, s, u, v, smp[4];
                plane_coord<AC>(zdiff, ph, pw, ex, ey, rx[j], ry[j], rz[j], cx, cy, ix, iy, s, u, v);
                gather_sample<TexT, STRICT>(vol + static_cast<int64_t>(k) * s_plane, s_chan, s_row, Ht, Wt, ix, iy, check_range, bad, smp);
                blend<STRICT>(A[j], smp[0], smp[1], smp[2], smp[3], s, ray_dot(j));
            }
        };

        if (fit) {
            // ---- loader role of this lane: row lr + q * RPP, item column lc of the box ----
            const int lr = lane / LPR, lc = lane - lr * LPR;
            const bool lane_ok = lr < RPP;
            // slot of the lane's first texel x0 = qx0 + TPI * lc relative to the slot of qx0: TPI * lc + (x0 >> 3) - (qx0 >> 3);
            // for 8-texel items that is 9 * lc, for 4-texel items 4 * lc + ((lc + odd) >> 1) with odd = (qx0 / 4) & 1
            const uint32_t dst0 = box_addr + static_cast<uint32_t>(SS) * static_cast<uint32_t>(lr * pitch + TPI * lc + (TPI == 8 ? PADS * lc : 0));
            const bool row_ok_last = lr + (NP - 1) * RPP < NR;  // the last pass may reach past the tallest box
            const uint32_t dst_step = static_cast<uint32_t>(SS) * static_cast<uint32_t>(RPP * pitch);
            const uint32_t goff0 = static_cast<uint32_t>((lr * static_cast<int>(s_row) + TPI * lc) * ES);
            const uint32_t goff_step = static_cast<uint32_t>(RPP * static_cast<int>(s_row) * ES);
            const int num_records = __builtin_amdgcn_readfirstlane(static_cast<int>(((Ht - 1) * s_row + Wt) * ES));

            auto run = [&](auto np, auto full_tag) -> int {
                constexpr int NPC = decltype(np)::value;
                // the loads of the next plane, in flight while the current one is composited.  (A second plane of loads in flight
                // gained < 1 % -- hipcc's waitcnt insertion waits for ALL outstanding loads on every other plane of the two-plane
                // loop, and the loads-only microbenchmark gets slower, not faster, with 8 instead of 4 buffer loads per wave in
                // flight: profiles/r02_variants.txt, tools/ubench/strip_loader.hip.)
                using LoadRegs = u32x4_t[NPC][4];
                LoadRegs La;
                auto issue = [&](const u32x4_t& lo, int k, LoadRegs& L) __attribute__((always_inline)) {
                    const uint32_t clo = lo.y & 0xffu, ncol = (lo.y >> 8) & 0xffu, nrk = lo.y >> 24;
                    const bool xok = lane_ok & (static_cast<uint32_t>(lc) - clo < ncol);
#ifdef GMPI_TUNE
                    const uint32_t sel = (xok && !(p.flags & (1u << 24))) ? lo.x + goff0 : 0x80000000u;  // ablation: no memory traffic
#else
                    const uint32_t sel = xok ? lo.x + goff0 : 0x80000000u;  // >= num_records: reads zeros, no memory access
#endif
                    // one descriptor per channel image of plane k (the scalar offset of a buffer load takes part in the
                    // range check on this part, so it cannot carry the channel); rows outside the texture fall outside
                    // num_records and read as zeros (padding_mode="zeros")
                    const TexT* pl = vol + static_cast<int64_t>(k) * s_plane;
                    __amdgpu_buffer_rsrc_t rsrc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        rsrc[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<TexT*>(pl + c * s_chan), 0, num_records, 0x00020000);
#pragma unroll
                    for (int q = 0; q < NPC; ++q) {
                        // rows past this plane's box are not fetched (the lane map covers the tallest box of all planes)
                        const uint32_t off = (static_cast<uint32_t>(lr + q * RPP) < nrk) ? sel + static_cast<uint32_t>(q) * goff_step : 0x80000000u;
#pragma unroll
                        for (int c = 0; c < 4; ++c) L[q][c] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc[c], off, 0, 0));
                    }
                };
                // storage -> fp32, range check, and the 16-byte stores of one pass.  PH = phase of the box origin in its
                // 8-column block (compile time: the slot offsets are immediates)
                u32x4_t hpair[4];  // HALF: the item's texels as fp16 RGBA, two per vector (converted once, stored by the phase's pattern)
                auto store_pass = [&](auto ph, uint32_t dst, bool dup, u32x4_t (&Lq)[4], uint32_t& mx) {
                    constexpr int PH = decltype(ph)::value;
                    if constexpr (HALF) {
                        // fp16 RGBA texels, 8 bytes each: texel t of the item sits in slot t + 2 * ((PH + t) >> 3) of the lane's
                        // 10-slot window; the first padding slot after an 8-column block holds a copy of the block's successor
                        // (so that slot, slot + 1 are always x, x + 1), the second is dead.  The window is written as five
                        // slot pairs = 16-byte stores of the texel pairs P0..P3 as converted (PH is even): the pair that starts
                        // a block is stored twice, once into the padding pair before it (copy + dead slot).
                        lds_u32x4* d = reinterpret_cast<lds_u32x4*>(static_cast<uintptr_t>(dst));
                        constexpr int HB = (8 - PH) / 2;  // first pair of the next 8-column block (4: this lane's pair 0 starts a block)
#pragma unroll
                        for (int i = 0; i < 4; ++i) d[i + (i >= HB ? 1 : 0)] = hpair[i];
                        if (HB == 4) d[-1] = hpair[0];
                        else d[HB] = hpair[HB];
                    } else {
                    float ch[4][TPI];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        IO::unpack(Lq[c], ch[c]);
                        mx = IO::fold(mx, Lq[c]);
                    }
                    lds_f32x4* d = reinterpret_cast<lds_f32x4*>(static_cast<uintptr_t>(dst));
#pragma unroll
                    for (int t = 0; t < TPI; ++t) {
                        f32x4_t tex;
                        tex[0] = ch[0][t], tex[1] = ch[1][t], tex[2] = ch[2][t], tex[3] = ch[3][t];
                        const int so = t + ((PH + t) >> 3);  // slot of texel t relative to the lane's first slot
                        d[so] = tex;
                        if (TPI == 8 ? ((PH + t) & 7) == 0 : (t == 0 && dup)) d[so - 1] = tex;  // copy into the padding slot before an 8-column block
                    }
                    }
                };
                auto stage = [&](const u32x4_t& lo_k, LoadRegs& L) __attribute__((always_inline)) {
                    uint32_t mx = 0;
                    const uint32_t phase = __builtin_amdgcn_readfirstlane((lo_k.y >> 16) & 7u);
                    uint32_t dstk = dst0;
                    bool dup = true;
                    if constexpr (TPI == 4) {  // (phase is 0 or 4: whether even or odd items start an 8-column block)
                        const uint32_t odd = phase >> 2;
                        dstk = dst0 + 16u * ((static_cast<uint32_t>(lc) + odd) >> 1);
                        dup = ((static_cast<uint32_t>(lc) + odd) & 1u) == 0;
                    }
#pragma unroll
                    for (int q = 0; q < NPC; ++q) {
#ifdef GMPI_TUNE
                        if (p.flags & (1u << 26)) {  // ablation: no LDS stores (the loads stay alive through the range check)
#pragma unroll
                            for (int c = 0; c < 4; ++c) mx = IO::fold(mx, L[q][c]);
                            continue;
                        }
#endif
                        if constexpr (HALF) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) mx = IO::fold(mx, L[q][c]);
                            IO::half_texels(L[q], hpair);
                        }
                        if (lane_ok && (q + 1 < NPC || row_ok_last)) {
                            const uint32_t dst = dstk + static_cast<uint32_t>(q) * dst_step;
                            if constexpr (TPI == 8) {
                                switch (phase >> 1) {
                                    case 0: store_pass(std::integral_constant<int, 0>{}, dst, dup, L[q], mx); break;
                                    case 1: store_pass(std::integral_constant<int, 2>{}, dst, dup, L[q], mx); break;
                                    case 2: store_pass(std::integral_constant<int, 4>{}, dst, dup, L[q], mx); break;
                                    default: store_pass(std::integral_constant<int, 6>{}, dst, dup, L[q], mx); break;
                                }
                            } else {
                                store_pass(std::integral_constant<int, 0>{}, dst, dup, L[q], mx);
                            }
                        }
                    }
                    bool unsafe = false;  // HALF: the plane holds a value fp16 cannot represent
                    if ((check_range || HALF) && __builtin_expect(IO::suspicious(mx), 0)) {
#pragma unroll
                        for (int q = 0; q < NPC; ++q)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (check_range && IO::bad(L[q][c])) bad |= 2u;
                                if (HALF && IO::unsafe_half(L[q][c])) unsafe = true;
                            }
                    }
                    return HALF && __any(unsafe);
                };
                // The ring entry of a plane (two 16-byte LDS broadcasts) is read one plane ahead: by the time the loader and the
                // compositor need it, the data has long arrived -- the plane loop has no wait for per-plane uniforms.
                auto entry = [&](int k, u32x4_t& lo, u32x4_t& hi) __attribute__((always_inline)) {
                    lo = tab[2 * (k & (kRing - 1))], hi = tab[2 * (k & (kRing - 1)) + 1];
                };
                fill(k_begin);
                u32x4_t lo, hi, lon, hin;
                entry(k_begin, lo, hi);
                issue(lo, k_begin, La);
#pragma unroll 1
                for (int g0 = k_begin; g0 < k_end; g0 += kGroup) {
                    if (g0 + kGroup < k_end) fill(g0 + kGroup);  // the slots of planes [g0 - 16, g0) are consumed
                    const int g1 = min(g0 + kGroup, k_end);
#pragma unroll 1
                    for (int k = g0; k < g1; ++k) {
                        const int kn = min(k + 1, k_end - 1);
                        entry(kn, lon, hin);
                        const bool by_gather = stage(lo, La);  // box of plane k -> LDS (after this wave's reads of plane k-1: LDS is in order)
                        issue(lon, kn, La);  // in flight while plane k is composited
#ifdef GMPI_TUNE
                        if (p.flags & (1u << 25)) { lo = lon, hi = hin; continue; }  // ablation: loader only
#endif
                        // HALF: a plane holding a value fp16 cannot represent ends the staged loop; the caller composites the
                        // rest of the planes by the direct gather (a side exit instead of a second path through the loop body:
                        // no copies of the accumulators where two paths would merge)
                        if (HALF && __builtin_expect(by_gather, 0)) return k;
                        composite(lo, hi, full_tag);
                        lo = lon, hi = hin;
                    }
                }
                return k_end;
            };
            int k_done = k_end;
            if (att == 0) {  // the whole strip: the common case, no per-pixel predicates
                switch (NP) {
                    case 1: k_done = run(std::integral_constant<int, 1>{}, std::true_type{}); break;
                    case 2: k_done = run(std::integral_constant<int, 2>{}, std::true_type{}); break;
                    default: k_done = run(std::integral_constant<int, 3>{}, std::true_type{}); break;
                }
            } else {
                switch (NP) {
                    case 1: k_done = run(std::integral_constant<int, 1>{}, std::false_type{}); break;
                    case 2: k_done = run(std::integral_constant<int, 2>{}, std::false_type{}); break;
                    default: k_done = run(std::integral_constant<int, 3>{}, std::false_type{}); break;
                }
            }
            if (HALF) {
#pragma unroll 1
                for (int k = k_done; k < k_end; ++k) composite_gather(k);
            }
        } else {
            // ---- direct gather (boxes do not fit): same arithmetic as render_gather.hip ----
            for (int k = k_begin; k < k_end; ++k) composite_gather(k);
        }
        if (att == 0) break;
    }

    // ---- merge of the plane ranges (SPLIT > 1): parts 1 .. SPLIT-1 hand their (T, C, Z) over through their own LDS region,
    //      part 0 folds them in plane order ----
    if constexpr (SPLIT > 1) {
        float* const mine = reinterpret_cast<float*>(smem + wv * kWaveLds);
        if (part > 0) {
#pragma unroll
            for (int j = 0; j < kPX; ++j) {
                const float v[5] = {A[j].T, A[j].r, A[j].g, A[j].b, A[j].z};
#pragma unroll
                for (int c = 0; c < 5; ++c) mine[(j * 5 + c) * 64 + lane] = v[c];
            }
        }
        __syncthreads();
        if (part > 0 || ghost) {
            report_status(p.status, bad);
            return;
        }
#pragma unroll 1
        for (int q = 1; q < SPLIT; ++q) {
            const float* theirs = reinterpret_cast<const float*>(smem + (wv + q * WPB) * kWaveLds);
#pragma unroll
            for (int j = 0; j < kPX; ++j) {
                const float T2 = theirs[(j * 5 + 0) * 64 + lane];
                A[j].r = __builtin_fmaf(A[j].T, theirs[(j * 5 + 1) * 64 + lane], A[j].r);
                A[j].g = __builtin_fmaf(A[j].T, theirs[(j * 5 + 2) * 64 + lane], A[j].g);
                A[j].b = __builtin_fmaf(A[j].T, theirs[(j * 5 + 3) * 64 + lane], A[j].b);
                A[j].z = __builtin_fmaf(A[j].T, theirs[(j * 5 + 4) * 64 + lane], A[j].z);
                A[j].T = A[j].T * T2;
            }
        }
    }

    // ---- epilogue ----
#pragma unroll
    for (int j = 0; j < kPX; ++j) {
        if (check_last) {  // assert_not_out_of_last_plane (mpi.py:381-395): u,v of the last plane, once per pixel
            const float d = dhw[3 * (D - 1) + 0], ph = dhw[3 * (D - 1) + 1], pw = dhw[3 * (D - 1) + 2];
            float ix, iy, s, u, v;
            plane_coord<AC>(d - ez, ph, pw, ex, ey, rx[j], ry[j], rz[j], cx, cy, ix, iy, s, u, v);
            if (!(u >= -1.0f && u <= 1.0f && v >= -1.0f && v <= 1.0f)) bad |= 1u;
        }
        float r = A[j].r, g = A[j].g, b = A[j].b;
        if (p.flags & (1u << 1)) {  // mpi_renderer.py:467  2*c - 1
            r = 2.0f * r - 1.0f;
            g = 2.0f * g - 1.0f;
            b = 2.0f * b - 1.0f;
        }
        if ((sx0 + lxp < W) && (sy0 + lyp + 2 * j < H)) {
            const int64_t q = pixel_index(j);
            float* __restrict__ out = p.rgb_out + static_cast<int64_t>(n) * 3 * HW + q;
            out[0] = r;
            out[HW] = g;
            out[2 * HW] = b;
            p.depth_out[static_cast<int64_t>(n) * HW + q] = finish_depth<STRICT>(A[j], ray_dot(j));
            if (p.T_out) p.T_out[static_cast<int64_t>(n) * HW + q] = A[j].T;
        }
    }
    report_status(p.status, bad);
}

// ---- host side ---------------------------------------------------------------------------------------------------
bool wave_variant_supports(const KParams& p, int dtype) {
    const int es = dtype == 0 ? 4 : 2;
    const int tpi = 16 / es;  // texels per 16-byte loader item
    if (p.Wt % tpi != 0) return false;  // items of a box that touches the border must not straddle it
    if (reinterpret_cast<uintptr_t>(p.rgba) % 16 != 0) return false;
    if (p.s_row % tpi != 0 || p.s_chan % tpi != 0 || p.s_plane % tpi != 0 || p.s_mpi % tpi != 0) return false;
    // offsets inside one channel image are kept in 31 bits
    if (static_cast<int64_t>(p.Ht + 64) * p.s_row * es >= (int64_t(1) << 30)) return false;
    return true;
}

template <typename TexT, int WPB, int WPS, int SPLIT, bool HALF, int GRPSEL, bool STRICT>
static hipError_t launch_wave_t(const KParams& p, hipStream_t stream) {
    const int tiles_x = (p.W + WPB * kSW - 1) / (WPB * kSW), tiles_y = (p.H + kSH - 1) / kSH;
    const int n_tiles = tiles_x * tiles_y * p.N;
    const dim3 grid(((n_tiles + 7) / 8) * 8), block(WPB * SPLIT * 64);
#ifdef GMPI_FAST_BUILD  // experiment builds (tools/build_tune.sh -DGMPI_FAST_BUILD): align_corners only
    if (!(p.flags & 1u)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((render_wave_kernel<TexT, true, STRICT, WPB, WPS, SPLIT, HALF, GRPSEL>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
#else
    if (p.flags & 1u) hipLaunchKernelGGL((render_wave_kernel<TexT, true, STRICT, WPB, WPS, SPLIT, HALF, GRPSEL>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else hipLaunchKernelGGL((render_wave_kernel<TexT, false, STRICT, WPB, WPS, SPLIT, HALF, GRPSEL>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
#endif
    return hipGetLastError();
}

// Texel format in LDS: fp32 volumes and the strict-order mode stage fp32 RGBA (16 bytes per texel); 16-bit volumes in default
// mode stage fp16 RGBA (HALF: 8 bytes per texel, the fp32 conversion folded into v_fma_mix_f32).
template <int WPB, int WPS, int SPLIT, int GRPSEL = 0>
static hipError_t launch_wave_d(const KParams& p, int dtype, hipStream_t stream, bool half = true) {
    if (dtype == 0) return launch_wave_t<float, WPB, WPS, SPLIT, false, GRPSEL, false>(p, stream);
#ifdef GMPI_TUNE
    if (!half) return dtype == 1 ? launch_wave_t<bf16_t, WPB, WPS, SPLIT, false, 0, false>(p, stream) : launch_wave_t<f16_t, WPB, WPS, SPLIT, false, 0, false>(p, stream);
#else
    (void)half;
#endif
    return dtype == 1 ? launch_wave_t<bf16_t, WPB, WPS, SPLIT, true, GRPSEL, false>(p, stream) : launch_wave_t<f16_t, WPB, WPS, SPLIT, true, GRPSEL, false>(p, stream);
}

static hipError_t launch_wave_strict(const KParams& p, int dtype, hipStream_t stream) {  // one configuration: 4 strips per workgroup, unsplit
    switch (dtype) {
        case 0: return launch_wave_t<float, 4, 3, 1, false, 0, true>(p, stream);
        case 1: return launch_wave_t<bf16_t, 4, 3, 1, false, 0, true>(p, stream);
        default: return launch_wave_t<f16_t, 4, 3, 1, false, 0, true>(p, stream);
    }
}

hipError_t launch_wave(const KParams& p0, int dtype, int tune, hipStream_t stream) {
    KParams p = p0;
#ifdef GMPI_TUNE
    p.flags |= static_cast<uint32_t>(tune & 0xf00) << 16;  // 256: no memory traffic, 512: loader only, 1024: no LDS stores, 2048: box statistics in status[1..3]
    if (p.flags & (1u << 4)) return launch_wave_strict(p, dtype, stream);
    switch (tune & 0xff) {
        case 1: return launch_wave_d<4, 3, 1>(p, dtype, stream, false);   // fp32 texels in LDS (round 2's first strip kernel)
        case 2: return launch_wave_d<4, 3, 3>(p, dtype, stream);
        case 4: return launch_wave_d<2, 3, 6>(p, dtype, stream);
        case 8: return launch_wave_d<4, 3, 1, 4>(p, dtype, stream);       // fp16 texels, 4 pixels of taps in flight
        case 9: return launch_wave_d<4, 3, 1, 2>(p, dtype, stream);       // fp16 texels, 2 pixels (the shipped configuration)
        case 25: return launch_wave_d<4, 2, 1, 2>(p, dtype, stream);      // 2 waves per SIMD (up to 256 VGPRs, 20 KB of LDS per wave)
        default: break;
    }
#else
    (void)tune;
#endif
    // 4 strips side by side per workgroup (a 128x8 pixel band), 3 waves per SIMD (168 VGPRs, 13 KB of LDS per wave).  A launch
    // with fewer strips than the chip has wave slots (12 x 256) is latency-bound on the plane loop of each wave (~1 us per
    // plane): SPLIT waves share a strip and its planes (default mode only; measured on MI355X, profiles/r02_variants.txt:
    // one 256^2 x 96 view 190 us unsplit, 72 us 3-way, 45 us 6-way, against 75 us for the tile kernel).  Between 1024 and 2048
    // strips (at most 2 waves per SIMD, config 2) the split does not pay; the 2-waves-per-SIMD build (no register pressure:
    // 213 VGPRs, no scratch) is 5-8 % faster there than the 168-VGPR one.
    const int64_t strips = static_cast<int64_t>(p.N) * ((p.W + kSW - 1) / kSW) * ((p.H + kSH - 1) / kSH);
    if (p.flags & (1u << 4)) return launch_wave_strict(p, dtype, stream);  // strict order: sequential association, fp32 texels
    if (strips <= 512 && p.D >= 12) return launch_wave_d<2, 3, 6>(p, dtype, stream);
    if (strips <= 1024 && p.D >= 6) return launch_wave_d<4, 3, 3>(p, dtype, stream);
    if (strips <= 2048) return launch_wave_d<4, 2, 1>(p, dtype, stream);
    return launch_wave_d<4, 3, 1>(p, dtype, stream);
}

}  // namespace gmpi
