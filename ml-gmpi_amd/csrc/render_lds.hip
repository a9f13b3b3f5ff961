// render_lds.hip -- GMPI_VARIANT_LDS: pixel tiles, per-plane texel boxes staged through LDS.
//
// Why: the direct gather issues 16 dword loads per pixel*plane through the vector L1 and stalls at
// ~13-25 % of the HBM roofline (profiles/r01_*).  Here a workgroup owns a 64 x TH pixel tile and
// walks the D planes front to back.  Because the warp is a homography of a small rectangle, the
// texels a tile needs on one plane form a small box whose extremes are at the tile's four corner
// pixels; the box (<= MAXR rows x 72 texels, 4 channels) is copied from HBM with 16-byte row loads
// (each lane 4 consecutive texels of one channel row: 288-byte contiguous runs), kept as fp32 in
// LDS in [row][channel][x] order, and every pixel then takes its 16 taps with 8 ds_read2_b32
// (x0,x1 pairs).  Texels outside the texture are stored as zeros, so the consumer needs no masks
// ("zeros" padding of F.grid_sample).  Two LDS buffers + register staging give a one-barrier-per-
// plane pipeline: loads of plane k+1 are in flight while plane k is composited.
//
// HBM traffic: each texel of the volume is read once per view (halo rows/columns are shared with the
// neighbouring tiles through the XCD's L2: the blockIdx -> tile map gives every XCD a contiguous run
// of tiles).  Algorithmic bytes: 16 B (fp32) / 8 B (bf16) per pixel*plane + 28-32 B per pixel.
//
// Planes whose box does not fit (texture much finer than the image, degenerate rays) fall back to
// the direct gather for that plane only -- same arithmetic, so results do not depend on the path.
#include "gmpi_device.hpp"

namespace gmpi {

constexpr int kTW = 64;               // tile width in pixels = lanes of a wavefront
constexpr int kMaxQ = 18;             // 16-byte texel quads per staged row
constexpr int kPitch = kMaxQ * 4;     // 72 floats per channel-row
constexpr int kChunk = 128;           // planes per geometry-table refill
constexpr int kNT = 512;              // threads per workgroup (8 wavefronts)
constexpr float kBoxEps = 1.0f / 64;  // slack on the corner-derived box (fp32 error of ix is < 1e-3 texel)

struct PlaneRec {  // 32 bytes, one per plane of the current chunk
    int qx0, by0, nq, nrows;  // box origin (texels; qx0 multiple of 4), quads per row, rows; nq < 0: does not fit
    float zdiff, ph, pw, pad;
};

template <int TH>
struct Cfg {
    static constexpr int kMaxR = TH + 3;                     // rows of the staged box
    static constexpr int kItems = kMaxR * 4 * kMaxQ;          // float4 items per plane box
    static constexpr int kNL = (kItems + kNT - 1) / kNT;      // items per thread
    static constexpr int kTileFloats = kItems * 4;
    static constexpr int kLdsBytes = kChunk * 32 + 2 * kTileFloats * 4;
};

// storage -> 4 floats ------------------------------------------------------------------------------------
template <typename TexT> struct Quad;
template <> struct Quad<float> {
    using raw = float4;
    static __device__ __forceinline__ raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float4 cvt(raw v) { return v; }
};
template <> struct Quad<bf16_t> {
    using raw = uint2;
    static __device__ __forceinline__ raw zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ float4 cvt(raw v) {
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
};
template <> struct Quad<f16_t> {
    using raw = uint2;
    static __device__ __forceinline__ raw zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ float4 cvt(raw v) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y);
        return make_float4(static_cast<float>(a.x), static_cast<float>(a.y), static_cast<float>(b.x), static_cast<float>(b.y));
    }
};

// [0,1] test on raw fp32 bit patterns: non-negative floats order like unsigned ints, so v in [0,1]
// <=> bits <= 0x3f800000; negative values (sign bit) and NaN/Inf compare above; -0.0 is allowed.
__device__ __forceinline__ bool quad_out_of_unit(const float4& q) {
    const uint32_t a = __float_as_uint(q.x), b = __float_as_uint(q.y), c = __float_as_uint(q.z), d = __float_as_uint(q.w);
    const uint32_t m = max(max(a, b), max(c, d));
    if (__builtin_expect(m <= 0x3f800000u, 1)) return false;
    auto ok = [](uint32_t e) { return e <= 0x3f800000u || e == 0x80000000u; };
    return !(ok(a) && ok(b) && ok(c) && ok(d));
}

template <typename TexT, bool AC, bool STRICT, int TH, int PPT>
__global__ __launch_bounds__(kNT) void render_lds_kernel(const KParams p, const int tiles_x, const int tiles_y,
                                                         const int n_tiles) {
    using C = Cfg<TH>;
    using Q = Quad<TexT>;
    static_assert(kTW * TH / PPT == kNT, "tile / thread mismatch");
    constexpr int kRowsPerPass = TH / PPT;

    __shared__ __attribute__((aligned(16))) unsigned char smem[C::kLdsBytes];
    PlaneRec* tab = reinterpret_cast<PlaneRec*>(smem);
    float* tile0 = reinterpret_cast<float*>(smem + kChunk * 32);

    // ---- blockIdx -> tile: XCD x (blockIdx % 8) gets the contiguous run [x*per, (x+1)*per) of tiles, so
    //      neighbouring tiles (shared halo texels) meet in one L2 ------------------------------------------
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile_id = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile_id >= n_tiles) return;
    const int tiles_per_view = tiles_x * tiles_y;
    const int n = tile_id / tiles_per_view;
    const int trem = tile_id - n * tiles_per_view;
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wrow = tid >> 6;
    const int m = p.view_to_mpi ? p.view_to_mpi[n] : n / p.views_per_mpi;
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

    uint32_t bad = 0;
    if (p.status != nullptr && tile_id == n * tiles_per_view && tid == 0) {  // mpi.py:70-72, once per view
        const float ez0 = p.eye_pos[2];
        bool behind = false;
        for (int k = 0; k < D; ++k) behind |= !(dhw[3 * k] >= ez0);
        if (behind) atomicOr(p.status, 4u);
    }

    // ---- this thread's pixels (out-of-image lanes shadow the last row/column) --------------------------------
    const int px = txi * kTW + lane;
    const int pxc = min(px, W - 1);
    float rx[PPT], ry[PPT], rz[PPT], dot[PPT];
    int64_t pix[PPT];
    bool active[PPT];
    Accum A[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int py = tyi * TH + wrow + j * kRowsPerPass;
        active[j] = px < W && py < H;
        pix[j] = static_cast<int64_t>(min(py, H - 1)) * W + pxc;
        rx[j] = rdv[pix[j]], ry[j] = rdv[HW + pix[j]], rz[j] = rdv[2 * HW + pix[j]];
        float d = rx[j] * zx;  // einsum("nchw,nc->nhw") mpi.py:149
        d = d + ry[j] * zy;
        d = d + rz[j] * zz;
        dot[j] = d;
    }

    // ---- tile corner rays (for the per-plane texel box) ------------------------------------------------------
    const int cx0 = txi * kTW, cx1 = min(cx0 + kTW - 1, W - 1);
    const int cy0 = tyi * TH, cy1 = min(cy0 + TH - 1, H - 1);

    // ---- loader role: item i = tid + r*kNT  <->  float4 slot i of the box = (row, channel, quad) -------------
    uint32_t g_off[C::kNL];  // element offset of the item inside a plane, relative to the box origin
    int it_row[C::kNL], it_col[C::kNL];
#pragma unroll
    for (int r = 0; r < C::kNL; ++r) {
        const int i = tid + r * kNT;
        const int rowc = i / kMaxQ;
        it_col[r] = i - rowc * kMaxQ;
        it_row[r] = rowc >> 2;
        const int c = rowc & 3;
        g_off[r] = static_cast<uint32_t>(c * s_chan + it_row[r] * s_row + 4 * it_col[r]);
        if (i >= C::kItems) it_row[r] = 1 << 20;  // never valid
    }
    typename Q::raw L[C::kNL];

    for (int kc = 0; kc < D; kc += kChunk) {
        const int kn = min(kChunk, D - kc);
        __syncthreads();  // previous chunk's table / tiles are no longer read
        // ---- per-plane geometry: texel box of this tile from its 4 corner pixels ---------------------------
        for (int t = tid; t < kn; t += kNT) {
            const int k = kc + t;
            PlaneRec rec;
            const float d = dhw[3 * k + 0];
            rec.ph = dhw[3 * k + 1], rec.pw = dhw[3 * k + 2];
            rec.zdiff = d - ez;
            rec.pad = 0.f;
            float mnx = __builtin_inff(), mxx = -__builtin_inff(), mny = mnx, mxy = mxx;
            bool finite = true;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t q = static_cast<int64_t>((c & 2) ? cy1 : cy0) * W + ((c & 1) ? cx1 : cx0);
                float ix, iy, s, u, v;
                plane_coord<AC>(rec.zdiff, rec.ph, rec.pw, ex, ey, rdv[q], rdv[HW + q], rdv[2 * HW + q], cx, cy, ix, iy, s, u, v);
                finite = finite && (fabsf(ix) < 1e6f) && (fabsf(iy) < 1e6f);  // false for NaN too
                mnx = fminf(mnx, ix), mxx = fmaxf(mxx, ix), mny = fminf(mny, iy), mxy = fmaxf(mxy, iy);
            }
            if (finite) {
                const int bx0 = static_cast<int>(floorf(mnx - kBoxEps)), bx1 = static_cast<int>(floorf(mxx + kBoxEps)) + 1;
                const int by0 = static_cast<int>(floorf(mny - kBoxEps)), by1 = static_cast<int>(floorf(mxy + kBoxEps)) + 1;
                rec.qx0 = bx0 & ~3;
                rec.by0 = by0;
                rec.nq = ((bx1 - rec.qx0) >> 2) + 1;
                rec.nrows = by1 - by0 + 1;
                if (rec.nq > kMaxQ || rec.nrows > C::kMaxR) rec.nq = -1;
            } else {
                rec.qx0 = rec.by0 = rec.nrows = 0;
                rec.nq = -1;
            }
            tab[t] = rec;
        }
        __syncthreads();

        // ---- register staging of one plane's box --------------------------------------------------------------
        auto issue_loads = [&](int t) {
            const int qx0 = __builtin_amdgcn_readfirstlane(tab[t].qx0), by0 = __builtin_amdgcn_readfirstlane(tab[t].by0);
            const int nq = __builtin_amdgcn_readfirstlane(tab[t].nq), nrows = __builtin_amdgcn_readfirstlane(tab[t].nrows);
            if (nq < 0) return;
            const TexT* __restrict__ base = vol + (static_cast<int64_t>(kc + t) * s_plane + static_cast<int64_t>(by0) * s_row + qx0);
#pragma unroll
            for (int r = 0; r < C::kNL; ++r) {
                const bool in_box = it_row[r] < nrows && it_col[r] < nq;
                const bool in_tex = static_cast<unsigned>(by0 + it_row[r]) < static_cast<unsigned>(Ht) &&
                                    static_cast<unsigned>(qx0 + 4 * it_col[r]) < static_cast<unsigned>(Wt);
                L[r] = Q::zero();
                if (in_box && in_tex) L[r] = *reinterpret_cast<const typename Q::raw*>(base + g_off[r]);
            }
        };
        auto store_box = [&](int t, float* tile) {
            const int nq = __builtin_amdgcn_readfirstlane(tab[t].nq), nrows = __builtin_amdgcn_readfirstlane(tab[t].nrows);
            if (nq < 0) return;
#pragma unroll
            for (int r = 0; r < C::kNL; ++r) {
                if (it_row[r] < nrows && it_col[r] < nq) {
                    const float4 q = Q::cvt(L[r]);
                    if (check_range && quad_out_of_unit(q)) bad |= 2u;
                    reinterpret_cast<float4*>(tile)[tid + r * kNT] = q;
                }
            }
        };

        issue_loads(0);
        for (int t = 0; t < kn; ++t) {
            float* tile = tile0 + (t & 1) * C::kTileFloats;
            store_box(t, tile);
            __syncthreads();  // box t visible; everybody is done reading box t-1 (other buffer is free for t+1)
            if (t + 1 < kn) issue_loads(t + 1);  // in flight while box t is composited

            const PlaneRec rec = tab[t];
            const bool last = (kc + t == D - 1);
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                float ix, iy, s, u, v;
                plane_coord<AC>(rec.zdiff, rec.ph, rec.pw, ex, ey, rx[j], ry[j], rz[j], cx, cy, ix, iy, s, u, v);
                if (check_last && last && !(u >= -1.0f && u <= 1.0f && v >= -1.0f && v <= 1.0f)) bad |= 1u;
                float smp[4];
                if (rec.nq >= 0) {
                    const Footprint f = footprint(ix, iy, Ht, Wt);
                    // box-relative corner; the box contains every tap of the tile, the clamp only keeps wild
                    // coordinates (NaN rays) inside the buffer.  (f.x0/f.y0 carry the gather path's -2
                    // out-of-range sentinel, so the integer corner is re-derived from the floor here.)
                    const int lx = min(max(static_cast<int>(floorf(ix)) - rec.qx0, 0), kPitch - 2);
                    const int ly = min(max(static_cast<int>(floorf(iy)) - rec.by0, 0), C::kMaxR - 2);
                    const float* __restrict__ t0 = tile + (ly * 4) * kPitch + lx;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float t_nw = t0[c * kPitch], t_ne = t0[c * kPitch + 1];
                        const float t_sw = t0[(4 + c) * kPitch], t_se = t0[(4 + c) * kPitch + 1];
                        smp[c] = bilerp<STRICT>(t_nw, t_ne, t_sw, t_se, f);
                    }
                } else {
                    gather_sample<TexT, STRICT>(vol + static_cast<int64_t>(kc + t) * s_plane, s_chan, s_row, Ht, Wt, ix, iy,
                                                check_range, bad, smp);
                }
                blend<STRICT>(A[j], smp[0], smp[1], smp[2], smp[3], s, dot[j]);
            }
        }
    }

    const bool pm1 = (p.flags & (1u << 1)) != 0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        float r = A[j].r, g = A[j].g, b = A[j].b;
        if (pm1) {  // mpi_renderer.py:467  2*c - 1
            r = 2.0f * r - 1.0f;
            g = 2.0f * g - 1.0f;
            b = 2.0f * b - 1.0f;
        }
        if (active[j]) {
            float* __restrict__ out = p.rgb_out + static_cast<int64_t>(n) * 3 * HW + pix[j];
            out[0] = r;
            out[HW] = g;
            out[2 * HW] = b;
            p.depth_out[static_cast<int64_t>(n) * HW + pix[j]] = A[j].z;
            if (p.T_out) p.T_out[static_cast<int64_t>(n) * HW + pix[j]] = A[j].T;
        }
    }
    report_status(p.status, bad);
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int elem_size(int dtype) { return dtype == 0 ? 4 : 2; }

bool lds_variant_supports(const KParams& p, int dtype) {
    const int es = elem_size(dtype);
    const int quad_bytes = 4 * es;  // one 4-texel quad
    if (p.Wt % 4 != 0) return false;
    if (reinterpret_cast<uintptr_t>(p.rgba) % quad_bytes != 0) return false;
    if (p.s_row % 4 != 0 || p.s_chan % 4 != 0 || p.s_plane % 4 != 0 || p.s_mpi % 4 != 0) return false;
    // the in-plane item offset is kept in 32 bits
    const int64_t span = 3 * p.s_chan + 32 * p.s_row + 128;
    if (span >= (int64_t(1) << 31) / es) return false;
    return true;
}

int lds_variant_query(int what) {
    switch (what) {
        case 3: return Cfg<16>::kLdsBytes;
        case 4: return kTW;
        case 5: return 16;
        default: return -1;
    }
}

template <typename TexT, int TH, int PPT>
static hipError_t launch_lds_t(const KParams& p, hipStream_t stream) {
    const int tiles_x = (p.W + kTW - 1) / kTW, tiles_y = (p.H + TH - 1) / TH;
    const int n_tiles = tiles_x * tiles_y * p.N;
    const dim3 grid(((n_tiles + 7) / 8) * 8), block(kNT);
    const bool ac = p.flags & 1u, strict = p.flags & (1u << 4);
    if (ac && strict) hipLaunchKernelGGL((render_lds_kernel<TexT, true, true, TH, PPT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else if (ac) hipLaunchKernelGGL((render_lds_kernel<TexT, true, false, TH, PPT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else if (strict) hipLaunchKernelGGL((render_lds_kernel<TexT, false, true, TH, PPT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else hipLaunchKernelGGL((render_lds_kernel<TexT, false, false, TH, PPT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    return hipGetLastError();
}

hipError_t launch_lds(const KParams& p, int dtype, hipStream_t stream) {
    switch (dtype) {
        case 0: return launch_lds_t<float, 16, 2>(p, stream);
        case 1: return launch_lds_t<bf16_t, 16, 2>(p, stream);
        default: return launch_lds_t<f16_t, 16, 2>(p, stream);
    }
}

}  // namespace gmpi
