// render_backward.hip -- gradient of the fused render w.r.t. the RGBA volume (the reference's G-step back-propagates
// through MPIRenderer.render into the generator: gmpi/train.py:740-779; the sampling grid itself carries no gradient,
// gmpi/core/mpi.py:65 `with torch.no_grad()`).
//
// Forward (mpi.py:421-434):  w_k = a_k T_k,  T_{k+1} = T_k om_k,  om_k = 1 - a_k + 1e-10,  C = sum_k w_k rgb_k,
// Z = sum_k w_k depth_k, with (rgb_k, a_k) the bilinear samples of plane k.  With upstream gradients gC (3), gZ and
// q_k = <gC, rgb_k> + gZ depth_k:
//     dL/drgb_k = gC * w_k
//     dL/da_k   = T_k q_k  -  S_k / om_k,      S_k = sum_{j>k} w_j q_j
// and every sample gradient is scattered to its four texels with the bilinear weights (atomicAdd, fp32).
//
// ONE sweep, back to front: S_k is accumulated directly (smallest terms first -- forming it as a difference of two
// front-to-back sums cancels catastrophically behind nearly opaque planes, where om_k is tiny and S_k/om_k is O(1)),
// and T_k = T_{k+1} / om_k starts from the final transmittance the forward wrote (GmpiRenderParams.transmittance_out).
// T is carried as mantissa x 2^exponent so that a product of several 1e-10 factors does not underflow; if the forward's
// value is missing or has underflowed (< 1e-30: four exactly opaque planes in a row) the pixel first walks the alpha
// channel front to back to rebuild it in that representation.
// Taps come straight from global memory (same addressing as the gather kernel): the backward runs at training sizes
// (D = 32, gmpi.yml:78).  The coordinate chain is the forward's (plane_coord), so both sample the same texels.
#include "gmpi_backward.hpp"

#include <cstdlib>
#include <type_traits>

namespace gmpi {

// One plane of the back-to-front sweep for one pixel: sample, T_k = T_{k+1}/om_k, gradients d_s[4] of the sample
// (r, g, b, alpha), suffix sum update.
struct BwdPixel {
    float gr, gg, gb, gz, dot;
    XT T;      // T_{k+1} on entry, T_k on exit
    float S;   // sum_{j>k} w_j q_j on entry, sum_{j>=k} on exit
    __device__ __forceinline__ void plane(const float (&smp)[4], float s, float (&d_s)[4]) {
        const float a = smp[3];
        const float om = (1.0f - a) + 1e-10f;
        T.m = T.m / om;
        T.renorm();
        const float Tk = T.value();
        const float q = gr * smp[0] + gg * smp[1] + gb * smp[2] + gz * (s * dot);
        const float w = a * Tk;
        d_s[0] = gr * w, d_s[1] = gg * w, d_s[2] = gb * w;
        d_s[3] = Tk * q - S / om;
        S += w * q;
    }
};

template <typename TexT, bool AC>
__global__ __launch_bounds__(256) void render_backward_kernel(const KParams p, const BwdParams b) {
    const int n = blockIdx.z;
    const int px = blockIdx.x * 64 + threadIdx.x;
    const int py = blockIdx.y * 4 + threadIdx.y;
    if (px >= p.W || py >= p.H) return;
    uint32_t bad_index = 0;  // (the forward reports a bad view index; here it is only clamped)
    const int m = view_mpi(p, n, bad_index);
    const float* __restrict__ dhw = p.dhw + static_cast<int64_t>(m) * p.D * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const int64_t HW = static_cast<int64_t>(p.H) * p.W;
    const int64_t pix = static_cast<int64_t>(py) * p.W + px;
    const float* __restrict__ rd = p.ray_dir + static_cast<int64_t>(n) * 3 * HW + pix;
    const float rx = rd[0], ry = rd[HW], rz = rd[2 * HW];
    const float zx = p.z_dir[3 * n + 0], zy = p.z_dir[3 * n + 1], zz = p.z_dir[3 * n + 2];
    float dot = rx * zx;
    dot = dot + ry * zy;
    dot = dot + rz * zz;
    const int Ht = p.Ht, Wt = p.Wt;
    const float cx = AC ? static_cast<float>(Wt - 1) * 0.5f : static_cast<float>(Wt);
    const float cy = AC ? static_cast<float>(Ht - 1) * 0.5f : static_cast<float>(Ht);
    const float scale = (p.flags & (1u << 1)) ? 2.0f : 1.0f;  // forward wrote 2*C-1 (mpi_renderer.py:467)
    const float* __restrict__ g = b.g_rgb + static_cast<int64_t>(n) * 3 * HW + pix;
    const float gr = scale * g[0], gg = scale * g[HW], gb = scale * g[2 * HW];
    const float gz = b.g_depth ? b.g_depth[static_cast<int64_t>(n) * HW + pix] : 0.0f;
    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    float* __restrict__ gvol = b.g_rgba + static_cast<int64_t>(m) * b.gs_mpi;

    // ---- back-to-front sweep: gradients, scattered with the bilinear weights --------------------------------------
    const float t_fwd = p.T_out ? p.T_out[static_cast<int64_t>(n) * HW + pix] : 0.0f;
    BwdPixel bp{gr, gg, gb, gz, dot, total_transmittance<TexT, AC>(p, dhw, vol, t_fwd, p.T_out != nullptr, ex, ey, ez, rx, ry, rz, cx, cy), 0.0f};
    uint32_t unused = 0;
    for (int k = p.D - 1; k >= 0; --k) {
        float ix, iy, s, u, v;
        plane_coord<AC>(dhw[3 * k] - ez, dhw[3 * k + 1], dhw[3 * k + 2], ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
        float smp[4], d_s[4];
        gather_sample<TexT, false>(vol + static_cast<int64_t>(k) * p.s_plane, p.s_chan, p.s_row, Ht, Wt, ix, iy, false, unused, smp);
        bp.plane(smp, s, d_s);

        Footprint f = footprint(ix, iy, Ht, Wt);
        const bool x0in = f.x0 >= 0 && f.x0 <= Wt - 1, x1in = f.x0 >= -1 && f.x0 <= Wt - 2;
        const bool y0in = f.y0 >= 0 && f.y0 <= Ht - 1, y1in = f.y0 >= -1 && f.y0 <= Ht - 2;
        float* __restrict__ gp = gvol + static_cast<int64_t>(k) * b.gs_plane;
        const int64_t oa = static_cast<int64_t>(f.y0) * b.gs_row + f.x0, ob = oa + b.gs_row;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* __restrict__ gc = gp + c * b.gs_chan;
            const float d = d_s[c];
            if (x0in && y0in) atomicAdd(gc + oa, d * f.nw);
            if (x1in && y0in) atomicAdd(gc + oa + 1, d * f.ne);
            if (x0in && y1in) atomicAdd(gc + ob, d * f.sw);
            if (x1in && y1in) atomicAdd(gc + ob + 1, d * f.se);
        }
    }
}

// ---- tile version: the scatter is staged in LDS ---------------------------------------------------------------------
// A 32x16 pixel tile touches a small texel box on every plane (same corner argument as render_lds.hip).  The 16 adds
// of a pixel go to a copy of that box in LDS; the box is then flushed with ONE global atomic per texel and channel,
// along rows (coalesced): 4.4 instead of 16 global atomics per pixel*plane for a frontal view.  A plane whose box does
// not fit (strong minification, degenerate rays) scatters straight to global memory.
//
// The LDS copy is 64-bit FIXED POINT: ds_add_f32 retires ~0.2 T lane-adds/s on this part, ds_add_u64/u32 ~9 T
// (tools/ubench/lds_atomic_rate.hip), and the float version of this kernel spent 75 % of its time in them.  Per plane
// the workgroup takes the largest |gradient| of its samples (one ds_max_u32 per pixel), scales by the power of two
// that puts it at 2^48 (exact), and accumulates integers: resolution 2^-48 of the largest term -- finer than a chain
// of fp32 adds -- with 2^14 terms of headroom (a tile has at most 2^11 taps).  Integer sums also make the staged part
// of the result independent of the order of the adds.
constexpr int kBwdThreads = 512, kBwdTW = 32, kBwdTH = 16;
constexpr int kBwdPitch = 56, kBwdRows = 27, kBwdCap = kBwdPitch * kBwdRows * 4;  // 64-bit words per box (47 KB)
constexpr int kBwdChunk = 96;
constexpr int kBwdFixBits = 48;

template <typename TexT, bool AC>
__global__ __launch_bounds__(kBwdThreads, 6) void render_backward_tile_kernel(const KParams p, const BwdParams b, const int tiles_x) {
    __shared__ int4 box[kBwdChunk];  // bx0, by0, nx (<= 0: not staged), ny
    __shared__ uint32_t gmax[kBwdChunk];  // per plane: largest |sample gradient| of the tile, as fp32 bits
    __shared__ unsigned long long acc[kBwdCap];
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int px = txi * kBwdTW + (tid % kBwdTW), py = tyi * kBwdTH + (tid / kBwdTW);
    const bool active = px < p.W && py < p.H;
    uint32_t bad_index = 0;  // (the forward reports a bad view index; here it is only clamped)
    const int m = view_mpi(p, n, bad_index);
    const float* __restrict__ dhw = p.dhw + static_cast<int64_t>(m) * p.D * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const int64_t HW = static_cast<int64_t>(p.H) * p.W;
    const int64_t pix = static_cast<int64_t>(min(py, p.H - 1)) * p.W + min(px, p.W - 1);
    const float* __restrict__ rdv = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
    const float rx = rdv[pix], ry = rdv[HW + pix], rz = rdv[2 * HW + pix];
    const float zx = p.z_dir[3 * n + 0], zy = p.z_dir[3 * n + 1], zz = p.z_dir[3 * n + 2];
    float dot = rx * zx;
    dot = dot + ry * zy;
    dot = dot + rz * zz;
    const int Ht = p.Ht, Wt = p.Wt;
    const float cx = AC ? static_cast<float>(Wt - 1) * 0.5f : static_cast<float>(Wt);
    const float cy = AC ? static_cast<float>(Ht - 1) * 0.5f : static_cast<float>(Ht);
    const float scale = (p.flags & (1u << 1)) ? 2.0f : 1.0f;  // forward wrote 2*C-1 (mpi_renderer.py:467)
    const float* __restrict__ g = b.g_rgb + static_cast<int64_t>(n) * 3 * HW + pix;
    const float gr = active ? scale * g[0] : 0.f, gg = active ? scale * g[HW] : 0.f, gb = active ? scale * g[2 * HW] : 0.f;
    const float gz = (active && b.g_depth) ? b.g_depth[static_cast<int64_t>(n) * HW + pix] : 0.0f;
    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    float* __restrict__ gvol = b.g_rgba + static_cast<int64_t>(m) * b.gs_mpi;

    for (int i = tid; i < kBwdCap; i += kBwdThreads) acc[i] = 0ull;

    const float t_fwd = (active && p.T_out) ? p.T_out[static_cast<int64_t>(n) * HW + pix] : 1.0f;
    BwdPixel bp{gr, gg, gb, gz, dot, XT{1.0f, 0}, 0.0f};
    if (active) bp.T = total_transmittance<TexT, AC>(p, dhw, vol, t_fwd, p.T_out != nullptr, ex, ey, ez, rx, ry, rz, cx, cy);
    uint32_t unused = 0;

    // ---- back-to-front sweep: gradients; scatter through the LDS boxes ------------------------------------------
    const int cx0 = txi * kBwdTW, cx1 = min(cx0 + kBwdTW - 1, p.W - 1);
    const int cy0 = tyi * kBwdTH, cy1 = min(cy0 + kBwdTH - 1, p.H - 1);
    for (int kend = p.D; kend > 0; kend -= kBwdChunk) {  // chunks of planes, last chunk first
        const int kc = max(kend - kBwdChunk, 0), kn = kend - kc;
        __syncthreads();  // previous chunk's table no longer read; (first pass) the zero fill is complete
        for (int t = tid; t < kn; t += kBwdThreads) {
            const int k = kc + t;
            const float zdiff = dhw[3 * k] - ez, ph = dhw[3 * k + 1], pw = dhw[3 * k + 2];
            float mnx = __builtin_inff(), mxx = -__builtin_inff(), mny = mnx, mxy = mxx;
            bool finite = true;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t q = static_cast<int64_t>((c & 2) ? cy1 : cy0) * p.W + ((c & 1) ? cx1 : cx0);
                float ix, iy, s, u, v;
                plane_coord<AC>(zdiff, ph, pw, ex, ey, rdv[q], rdv[HW + q], rdv[2 * HW + q], cx, cy, ix, iy, s, u, v);
                finite = finite && (fabsf(ix) < 1e6f) && (fabsf(iy) < 1e6f);
                mnx = fminf(mnx, ix), mxx = fmaxf(mxx, ix), mny = fminf(mny, iy), mxy = fmaxf(mxy, iy);
            }
            int4 bb = make_int4(0, 0, 0, 0);
            if (finite) {
                const float eps = 1.0f / 64;
                bb.x = static_cast<int>(floorf(mnx - eps)), bb.y = static_cast<int>(floorf(mny - eps));
                bb.z = static_cast<int>(floorf(mxx + eps)) + 2 - bb.x, bb.w = static_cast<int>(floorf(mxy + eps)) + 2 - bb.y;
                if (bb.z > kBwdPitch || bb.w > kBwdRows) bb.z = 0;
            }
            box[t] = bb;
            gmax[t] = 0u;
        }
        __syncthreads();
        for (int t = kn - 1; t >= 0; --t) {
            const int k = kc + t;
            const int4 bb = box[t];
            float* __restrict__ gp = gvol + static_cast<int64_t>(k) * b.gs_plane;
            float d_s[4] = {0.f, 0.f, 0.f, 0.f};
            Footprint f{};
            if (active) {
                float ix, iy, s, u, v;
                plane_coord<AC>(dhw[3 * k] - ez, dhw[3 * k + 1], dhw[3 * k + 2], ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
                float smp[4];
                gather_sample<TexT, false>(vol + static_cast<int64_t>(k) * p.s_plane, p.s_chan, p.s_row, Ht, Wt, ix, iy, false, unused, smp);
                bp.plane(smp, s, d_s);
                f = footprint(ix, iy, Ht, Wt);
                if (bb.z > 0) {  // non-negative floats order like their bit patterns; NaN/Inf end up on top
                    const float m = fmaxf(fmaxf(fabsf(d_s[0]), fabsf(d_s[1])), fmaxf(fabsf(d_s[2]), fabsf(d_s[3])));
                    atomicMax(&gmax[t], (m != m) ? 0x7fc00000u : __float_as_uint(m));
                }
            }
            __syncthreads();  // gmax[t] complete; the previous plane's flush is finished
            const uint32_t mb = gmax[t];
            // staged: the box fits and the gradients are finite (else: straight to global memory, fp32 atomics)
            const bool staged = bb.z > 0 && mb < 0x7f800000u;  // workgroup-uniform
            // scale = 2^(kBwdFixBits - floor(log2 max)), clamped to fp32's range (tiny maxima just use fewer bits)
            const int sh = min(kBwdFixBits - (static_cast<int>(mb >> 23) - 127), 126);
            const float scale = __builtin_amdgcn_ldexpf(1.0f, sh), inv_scale = __builtin_amdgcn_ldexpf(1.0f, -sh);
            if (active && (mb != 0u || bb.z <= 0)) {  // (mb == 0 with a box: every gradient of the tile is zero)
                const bool x0in = f.x0 >= 0 && f.x0 <= Wt - 1, x1in = f.x0 >= -1 && f.x0 <= Wt - 2;
                const bool y0in = f.y0 >= 0 && f.y0 <= Ht - 1, y1in = f.y0 >= -1 && f.y0 <= Ht - 2;
                // the box contains every in-texture tap of the tile; the extra test keeps wild coordinates (NaN rays) out
                const int lx = f.x0 - bb.x, ly = f.y0 - bb.y;
                if (staged && lx >= 0 && ly >= 0 && lx + 1 < bb.z && ly + 1 < bb.w) {
                    unsigned long long* __restrict__ l0 = acc + ly * (4 * kBwdPitch) + lx;
                    // |d * weight * scale| < 2^(kBwdFixBits+1): the product by a power of two is exact, the conversion
                    // rounds to the nearest integer
                    auto fix = [&](float v) { return static_cast<unsigned long long>(__double2ll_rn(static_cast<double>(v * scale))); };
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float d = d_s[c];
                        unsigned long long* __restrict__ lc = l0 + c * kBwdPitch;
#ifdef GMPI_TUNE
                        if (p.flags & (1u << 21)) continue;
#endif
                        if (x0in && y0in) atomicAdd(lc, fix(d * f.nw));
                        if (x1in && y0in) atomicAdd(lc + 1, fix(d * f.ne));
                        if (x0in && y1in) atomicAdd(lc + 4 * kBwdPitch, fix(d * f.sw));
                        if (x1in && y1in) atomicAdd(lc + 4 * kBwdPitch + 1, fix(d * f.se));
                    }
                } else {
                    const int64_t oa = static_cast<int64_t>(f.y0) * b.gs_row + f.x0, ob = oa + b.gs_row;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float* __restrict__ gc = gp + c * b.gs_chan;
                        const float d = d_s[c];
                        if (x0in && y0in) atomicAdd(gc + oa, d * f.nw);
                        if (x1in && y0in) atomicAdd(gc + oa + 1, d * f.ne);
                        if (x0in && y1in) atomicAdd(gc + ob, d * f.sw);
                        if (x1in && y1in) atomicAdd(gc + ob + 1, d * f.se);
                    }
                }
            }
            __syncthreads();  // box complete
            if (staged && mb != 0u) {  // flush + reset: line = (row, channel), x fastest -> coalesced global atomics
                const int n_items = bb.w * 4 * bb.z;
                for (int i = tid; i < n_items; i += kBwdThreads) {
                    const int line = i / bb.z, x = i - line * bb.z;
                    unsigned long long* __restrict__ src = acc + line * kBwdPitch + x;
                    const long long q = static_cast<long long>(*src);
                    if (q != 0) {
                        *src = 0ull;
                        const float v = static_cast<float>(static_cast<double>(q)) * inv_scale;
                        if (!(p.flags & (1u << 20))) atomicAdd(gp + (line & 3) * b.gs_chan + static_cast<int64_t>(bb.y + (line >> 2)) * b.gs_row + (bb.x + x), v);
                    }
                }
            }
        }
    }
}

// ---- tile version, round 5: ONE barrier per plane, wave roles ------------------------------------------------------------
// What the ablations of round 4 said (profiles/r04_band_variants.txt item 12): the atomics -- LDS and global together -- are a quarter of the
// kernel above; the rest is the chain  gather -> max -> barrier -> LDS atomics -> barrier -> flush  that every plane walks with exposed
// latencies.  This version keeps the decomposition (32 x 16 pixel tiles, the scatter staged in a copy of the tile's texel box in LDS, one
// global atomic per texel and channel) and removes the waits (profiles/r05_backward.txt has the measurements behind each point):
//   * software pipeline over the planes: the taps of plane t - 1 are in flight (registers) while plane t is scattered; the sample
//     gradients of plane t - 1 and their tile maximum are formed BEFORE the barrier that ends plane t, so the maximum costs no barrier of
//     its own; the flush of plane t + 1's box runs next to the scatter of plane t into the OTHER box: one barrier per plane;
//   * that barrier is `s_waitcnt lgkmcnt(0); s_barrier` -- it orders LDS only.  `__syncthreads()` is a workgroup-scope fence over EVERY
//     address space: on this part it drains vmcnt, i.e. every plane would wait for its own prefetch and for the flush's global atomics;
//   * two boxes fit because the staged sums are 32-bit fixed point (v_cvt_i32_f32 + ds_add_u32 instead of a double conversion +
//     ds_add_u64): per plane and tile the largest |sample gradient| M is scaled to 2^(30 - h), h = the bits of headroom for the taps that
//     can meet in one texel (from the pixel density of the tile on that plane: 4 bits at one pixel per texel, 11 = every tap of the tile
//     when the texture is much coarser than the image), i.e. <= 2^-27 M per add (rounded to nearest) at one pixel per texel -- finer than
//     the rounding of a chain of fp32 atomic adds (2^-24 of the running sum);
//   * WAVE ROLES: vector loads, stores and atomics of a wave share one counter (vmcnt) and retire in order, so a wave that flushes (global
//     atomics: a read-modify-write at L2 behind an HBM miss) and then gathers sits out its own atomics before it sees its taps.  The 8
//     pixel waves never issue an atomic on the staged path; 4 more waves of the workgroup do nothing but flush the box of plane t + 1
//     while the pixel waves work on plane t, and never wait for anything but their LDS reads;
//   * the coordinate chain takes its three divisions through the correctly rounded reciprocals (div_by_recip: the same quotients, hence
//     the same texels as the forward), the plane constants and their reciprocals come out of the LDS table; T / om likewise (one v_rcp_f32,
//     a Newton step, Markstein's correction); taps are fetched as (x, x + 1) pairs (8 loads of 8 bytes instead of 16 of 4 for fp32
//     volumes, borders by re-assigned weights) at 32-bit offsets from a uniform base; the direct-to-global scatter of a plane or pixel
//     that is not staged sits behind a wave-uniform branch.
constexpr int kB2Pix = 512;                           // pixel threads: one per pixel of the tile (8 waves)
constexpr int kB2Flush = 256;                         // + 4 waves that do nothing but flush
constexpr int kB2Threads = kB2Pix + kB2Flush;
constexpr int kB2Chunk = 96;
// The tile of a workgroup and its texel box in LDS (two boxes of Cap 32-bit words).  Round 5: 32 x 16 pixels.  Round 6: 64 x 8 -- a pixel wave is one
// pixel row, a box line is ~60 texels: fewer, longer lines for the flush, which is bound by the part's rate for atomic SEGMENTS of 64 bytes
// (profiles/r05_backward.txt): 191 -> 169 segments per 512 pixel-planes, 2.30 -> 2.12 ms at 1024^2 x 32 x 4, 0.352 -> 0.337 at 256^2 x 32 x 8
// (profiles/r06_backward.txt; the same file has the ownership scheme that was built on top -- plain stores for the aligned 128-byte lines a tile
// provably owns -- and why it lost).
template <int TW_, int TH_, int PITCH_, int ROWS_> struct B2Geo {
    static constexpr int TW = TW_, TH = TH_, Pitch = PITCH_, Rows = ROWS_, Cap = Pitch * Rows * 4;
    static_assert(TW * TH == kB2Pix && Pitch <= 96 && Cap % 2 == 0, "one pixel per pixel thread; a box line is at most two 64-column chunks (+ alignment)");
};
using B2Tall = B2Geo<32, 16, 56, 27>;   // 23.6 KB per box
using B2Wide = B2Geo<64, 8, 80, 19>;    // 23.8 KB per box
constexpr int kBwdDefaultGeo = 2;       // 1 = 32 x 16 tiles (round 5) | 2 = 64 x 8 tiles (profiles/r06_backward.txt)

// round to nearest (floor(x + 0.5)) in one instruction: the staged sums must not be biased -- with a texture much coarser than the image a hundred
// taps meet in one texel, and a truncating conversion adds up to half a unit of the fixed-point grid PER TAP in one direction
__device__ __forceinline__ int cvt_rpi(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Maximum of an unsigned word over the 64 lanes of a wave (EVERY lane must be enabled), as a scalar: four row shifts and two row broadcasts on
// the DPP path.  (Left to the compiler, `atomicMax` on a wave-uniform LDS address becomes a SCALAR loop over the lanes -- s_ff1 / v_readlane /
// s_max, 64 rounds of 7 instructions per wave and plane: it was two thirds of this kernel's pixel phase.)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    auto step = [&](auto ctrl, auto rows) {
        const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), decltype(ctrl)::value, decltype(rows)::value, 0xf, true));
        v = max(v, o);
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8   -> lane 15 of a row: the row's maximum
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1, 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2, 3 -> lane 63: the wave's maximum
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

template <typename TexT, bool AC, typename G>
__global__ __launch_bounds__(kB2Threads, 6) void render_backward_tile2_kernel(const KParams p, const BwdParams b, const int tiles_x) {
    constexpr int kB2TW = G::TW, kB2TH = G::TH, kB2Pitch = G::Pitch, kB2Rows = G::Rows, kB2Cap = G::Cap;
    __shared__ int4 box[kB2Chunk];        // bx0, by0, nx (<= 0: not staged), ny
    __shared__ float4 pcA[kB2Chunk];      // zdiff, w/2, h/2, RN(2/w)
    __shared__ float2 pcB[kB2Chunk];      // RN(2/h), headroom bits (as int bits)
    __shared__ uint32_t gmax[kB2Chunk];   // per plane: largest |sample gradient| of the tile, as fp32 bits
    __shared__ uint32_t acc[2][kB2Cap];
#ifdef GMPI_B2_PAD  // (experiment: LDS nobody uses, to hold the workgroups per CU down)
    __shared__ uint32_t lds_pad[GMPI_B2_PAD / 4];
    lds_pad[threadIdx.x] = 0u;
#endif
    constexpr int kES = static_cast<int>(sizeof(TexT));
    const int tid = threadIdx.x;
    // the role of a WAVE (kB2Pix is a multiple of 64), as a scalar: the two roles run separate loop nests behind a scalar branch -- as
    // divergent control flow the compiler would serialise both bodies in every wave and order the flush's atomics against the taps' loads
    const bool flusher = __builtin_amdgcn_readfirstlane(tid) >= kB2Pix;
    const int ptid = flusher ? 0 : tid, ftid = tid - kB2Pix;
    const int n = blockIdx.y;
    // XCD x = blockIdx.x % 8 (workgroups are dealt round-robin to the 8 XCDs) takes a contiguous, row-major run of the view's tiles: neighbours
    // share the halo rows of their boxes -- the taps they read and the gradient lines they add into -- in one L2 / from one XCD
    const int n_tiles = tiles_x * ((p.H + kB2TH - 1) / kB2TH);
    int tile = xcd_item_per_group(static_cast<int>(blockIdx.x), n_tiles, n_tiles);
#ifdef GMPI_TUNE
    if (p.flags & (1u << 24)) tile = blockIdx.x < static_cast<unsigned>(n_tiles) ? static_cast<int>(blockIdx.x) : n_tiles;   // GMPI_TUNE_SKIP=256: row-major order as dealt
#endif
    if (tile >= n_tiles) return;   // (whole workgroup: the grid is padded to a multiple of 8)
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int px = txi * kB2TW + (ptid % kB2TW), py = tyi * kB2TH + (ptid / kB2TW);
    const bool active = !flusher && px < p.W && py < p.H;
    uint32_t bad_index = 0;  // (the forward reports a bad view index; here it is only clamped)
    const int m = view_mpi(p, n, bad_index);
    const float* __restrict__ dhw = p.dhw + static_cast<int64_t>(m) * p.D * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const int64_t HW = static_cast<int64_t>(p.H) * p.W;
    const int64_t pix = static_cast<int64_t>(min(py, p.H - 1)) * p.W + min(px, p.W - 1);
    const float* __restrict__ rdv = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
#ifdef GMPI_TUNE
    const bool abl_noglobal = (p.flags & (1u << 20)) != 0, abl_nolds = (p.flags & (1u << 21)) != 0, abl_notaps = (p.flags & (1u << 23)) != 0;
#else
    constexpr bool abl_noglobal = false, abl_nolds = false, abl_notaps = false;
#endif
    const int Ht = p.Ht, Wt = p.Wt;
    const float cx = AC ? static_cast<float>(Wt - 1) * 0.5f : static_cast<float>(Wt);
    const float cy = AC ? static_cast<float>(Ht - 1) * 0.5f : static_cast<float>(Ht);
    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    float* __restrict__ gvol = b.g_rgba + static_cast<int64_t>(m) * b.gs_mpi;
    // (the launcher has checked that a plane's byte offsets fit 32 bits, for the volume and for its gradient)
    const uint32_t s_chan_b = static_cast<uint32_t>(p.s_chan) * kES, s_row_b = static_cast<uint32_t>(p.s_row) * kES;
    const uint32_t gs_chan = static_cast<uint32_t>(b.gs_chan), gs_row = static_cast<uint32_t>(b.gs_row);

    for (int i = tid; i < 2 * kB2Cap; i += kB2Threads) (&acc[0][0])[i] = 0u;

    // the pixel's own state: loaded by the PIXEL waves only (a load a flush wave issued and never used would stay "pending" for the
    // compiler's wait-count insertion, which then drains vmcnt -- the flush's own atomics -- in front of the first overwrite of its register)
    float rx = 0.0f, ry = 0.0f, rz = 1.0f, rrz = 1.0f, dot = 0.0f, gr = 0.0f, gg = 0.0f, gb = 0.0f, gz = 0.0f;
    XT T{1.0f, 0};
    float S = 0.0f;
    if (!flusher) {
        rx = rdv[pix], ry = rdv[HW + pix], rz = rdv[2 * HW + pix];
        rrz = 1.0f / rz;
        const float zx = p.z_dir[3 * n + 0], zy = p.z_dir[3 * n + 1], zz = p.z_dir[3 * n + 2];
        dot = rx * zx;
        dot = dot + ry * zy;
        dot = dot + rz * zz;
        const float oscale = (p.flags & (1u << 1)) ? 2.0f : 1.0f;  // forward wrote 2*C-1 (mpi_renderer.py:467)
        const float* __restrict__ g = b.g_rgb + static_cast<int64_t>(n) * 3 * HW + pix;
        gr = active ? oscale * g[0] : 0.f, gg = active ? oscale * g[HW] : 0.f, gb = active ? oscale * g[2 * HW] : 0.f;
        gz = (active && b.g_depth) ? b.g_depth[static_cast<int64_t>(n) * HW + pix] : 0.0f;
        const float t_fwd = (active && p.T_out) ? p.T_out[static_cast<int64_t>(n) * HW + pix] : 1.0f;
        if (active) T = total_transmittance<TexT, AC>(p, dhw, vol, t_fwd, p.T_out != nullptr, ex, ey, ez, rx, ry, rz, cx, cy);
    }

    // what a plane keeps between the issue of its taps and their use (Tap), and between its gradients and their scatter (Grad)
    struct Tap { float s, wx1, wy1; int x0, y0; float v[16]; };   // v: per channel (top p0, p1 | bottom p0, p1)
    struct Grad { float d[4]; float nw, ne, sw, se; int x0, y0; };   // (weights of taps outside the texture are 0)

    // coordinates of this pixel on plane (chunk-local index t) + its 8 pair loads
    auto fetch = [&](int t, int k, Tap& q) {
        const float4 a = pcA[t];
        const float2 c = pcB[t];
        float ix, iy;
        plane_coord_recip<AC>(a.x, a.y, a.z, a.w, c.x, ex, ey, rx, ry, rz, rrz, cx, cy, ix, iy, q.s);
        const float fx = floorf(ix), fy = floorf(iy);
        q.wx1 = ix - fx, q.wy1 = iy - fy;
        q.x0 = (fx >= -2.0f && fx <= static_cast<float>(Wt)) ? static_cast<int>(fx) : -2;   // (NaN / huge coordinates: out of range, all weights 0)
        q.y0 = (fy >= -2.0f && fy <= static_cast<float>(Ht)) ? static_cast<int>(fy) : -2;
        if (abl_notaps) return;
        const int xa = min(max(q.x0, 0), Wt - 2);
        const int ya = min(max(q.y0, 0), Ht - 1), yb = min(max(q.y0 + 1, 0), Ht - 1);
        const unsigned char* __restrict__ pl = reinterpret_cast<const unsigned char*>(vol + static_cast<int64_t>(k) * p.s_plane);
        const uint32_t oa = static_cast<uint32_t>(ya) * s_row_b + static_cast<uint32_t>(xa) * kES;
        const uint32_t ob = static_cast<uint32_t>(yb) * s_row_b + static_cast<uint32_t>(xa) * kES;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            load_pair<TexT>(pl, oa + c4 * s_chan_b, q.v[4 * c4 + 0], q.v[4 * c4 + 1]);
            load_pair<TexT>(pl, ob + c4 * s_chan_b, q.v[4 * c4 + 2], q.v[4 * c4 + 3]);
        }
    };
    // the taps have landed: bilinear samples, the plane's gradients (the arithmetic of BwdPixel::plane), the tile maximum
    auto grads = [&](int t, const Tap& q, Grad& gq) -> uint32_t {
        const int x0 = q.x0, y0 = q.y0;
        const bool x0in = x0 >= 0 && x0 <= Wt - 1, x1in = x0 >= -1 && x0 <= Wt - 2;
        const bool y0in = y0 >= 0 && y0 <= Ht - 1, y1in = y0 >= -1 && y0 <= Ht - 2;
        const float wx0 = x0in ? 1.0f - q.wx1 : 0.0f, wx1 = x1in ? q.wx1 : 0.0f;
        const float wy0 = y0in ? 1.0f - q.wy1 : 0.0f, wy1 = y1in ? q.wy1 : 0.0f;
        // the pair (p0, p1) sits at columns (xa, xa + 1), xa = clamp(x0, 0, Wt - 2): at the left border (x0 = -1) the tap x0 + 1 is p0, at the
        // right border (x0 = Wt - 1) the tap x0 is p1
        const int sh = x0 - min(max(x0, 0), Wt - 2);
        const float a0 = sh == 0 ? wx0 : (sh < 0 ? wx1 : 0.0f), a1 = sh == 0 ? wx1 : (sh > 0 ? wx0 : 0.0f);
        const float w00 = a0 * wy0, w01 = a1 * wy0, w10 = a0 * wy1, w11 = a1 * wy1;
        float smp[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            float acc_s = q.v[4 * c4 + 0] * w00;
            acc_s = __builtin_fmaf(q.v[4 * c4 + 1], w01, acc_s);
            acc_s = __builtin_fmaf(q.v[4 * c4 + 2], w10, acc_s);
            smp[c4] = __builtin_fmaf(q.v[4 * c4 + 3], w11, acc_s);
        }
        {   // T_k = T_{k+1} / om, the gradients, the suffix sum
            const float alpha = smp[3];
            const float om = (1.0f - alpha) + 1e-10f;
            float r = __builtin_amdgcn_rcpf(om);
            r = __builtin_fmaf(__builtin_fmaf(-om, r, 1.0f), r, r);      // Newton step: r = RN(1 / om) up to the last bit
            T.m = div_by_recip(T.m, om, r);
            T.renorm();
            const float Tk = T.value();
            const float qv = gr * smp[0] + gg * smp[1] + gb * smp[2] + gz * (q.s * dot);
            const float w = alpha * Tk;
            gq.d[0] = gr * w, gq.d[1] = gg * w, gq.d[2] = gb * w;
            gq.d[3] = Tk * qv - div_by_recip(S, om, r);
            S += w * qv;
        }
        {   // the scatter weights of the true taps (x0, y0) .. (x0 + 1, y0 + 1)
            const float ux0 = 1.0f - q.wx1, uy0 = 1.0f - q.wy1;
            gq.nw = (x0in && y0in) ? ux0 * uy0 : 0.0f, gq.ne = (x1in && y0in) ? q.wx1 * uy0 : 0.0f;
            gq.sw = (x0in && y1in) ? ux0 * q.wy1 : 0.0f, gq.se = (x1in && y1in) ? q.wx1 * q.wy1 : 0.0f;
        }
        gq.x0 = x0, gq.y0 = y0;
        // non-negative floats order like their bit patterns; NaN/Inf end up on top
        const float mx = fmaxf(fmaxf(fabsf(gq.d[0]), fabsf(gq.d[1])), fmaxf(fabsf(gq.d[2]), fabsf(gq.d[3])));
        return (mx != mx) ? 0x7fc00000u : __float_as_uint(mx);
    };
    // the tile maximum of a plane: one LDS atomic per wave (every lane of the wave must call this)
    auto tile_max = [&](int t, uint32_t lane_bits) {
        const uint32_t wmax = wave_max_u32(lane_bits);
        if (wmax != 0u && box[t].z > 0 && (tid & 63) == 0) atomicMax(&gmax[t], wmax);
    };
    // scatter of one plane's gradients: into the LDS box (fixed point) or, for a plane / a pixel that is not staged, straight to global memory
    auto scatter = [&](int t, int k, const Grad& gq, uint32_t* __restrict__ bx_acc) {
        const int4 bb = box[t];
        const uint32_t mb = gmax[t];
        const bool staged = bb.z > 0 && mb < 0x7f800000u;  // workgroup-uniform
        if (mb == 0u && bb.z > 0) return;                   // every gradient of the tile is zero (workgroup-uniform)
        const int x0 = gq.x0, y0 = gq.y0;
        const float nw = gq.nw, ne = gq.ne, sw = gq.sw, se = gq.se;
        const int lx = x0 - bb.x, ly = y0 - bb.y;
        const bool in_box = lx >= 0 && ly >= 0 && lx + 1 < bb.z && ly + 1 < bb.w;
        const bool any_w = nw != 0.0f || ne != 0.0f || sw != 0.0f || se != 0.0f;
        if (staged && in_box && active) {
            // scale = 2^(29 - h - floor(log2 M)): |d * weight * scale| < 2^(30 - h), and at most 2^h taps meet in one word
            const int hword = __float_as_int(pcB[t].y);
            const int hbits = hword & 0xff;
            const bool wide = (hword & 0x100) != 0;   // workgroup-uniform: two words per cell, |d * weight * scale| < 2^(42 - h)
            const int shf = min((wide ? 41 : 29) - hbits - (static_cast<int>(mb >> 23) - 127), 126);
            const float sc = __builtin_amdgcn_ldexpf(1.0f, shf);
            const float fnw = nw * sc, fne = ne * sc, fsw = sw * sc, fse = se * sc;   // (a power of two: exact)
            uint32_t* __restrict__ l0 = bx_acc + ly * (4 * kB2Pitch) + lx;
            if (wide && !abl_nolds) {
                auto add2 = [&](uint32_t* __restrict__ cell, float v) {   // v = hi * 2^12 + lo exactly (fp32: 24 significant bits), hi rounded to nearest
                    const int hi = cvt_rpi(v * (1.0f / 4096.0f));
                    const int lo = cvt_rpi(__builtin_fmaf(-static_cast<float>(hi), 4096.0f, v));
                    atomicAdd(cell, static_cast<uint32_t>(hi));
                    atomicAdd(cell + kB2Cap / 2, static_cast<uint32_t>(lo));
                };
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float d = gq.d[c4];
                    uint32_t* __restrict__ lc = l0 + c4 * kB2Pitch;
                    add2(lc, d * fnw), add2(lc + 1, d * fne), add2(lc + 4 * kB2Pitch, d * fsw), add2(lc + 4 * kB2Pitch + 1, d * fse);
                }
            } else if (!abl_nolds) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float d = gq.d[c4];
                    uint32_t* __restrict__ lc = l0 + c4 * kB2Pitch;
                    // (a tap outside the texture has weight 0 and adds 0 to a box cell outside the texture, which the flush never writes out;
                    //  round to nearest: at most half a unit of 2^-(30 - h) M per add, in either direction)
                    atomicAdd(lc, static_cast<uint32_t>(cvt_rpi(d * fnw)));
                    atomicAdd(lc + 1, static_cast<uint32_t>(cvt_rpi(d * fne)));
                    atomicAdd(lc + 4 * kB2Pitch, static_cast<uint32_t>(cvt_rpi(d * fsw)));
                    atomicAdd(lc + 4 * kB2Pitch + 1, static_cast<uint32_t>(cvt_rpi(d * fse)));
                }
            }
        }
        // cold: the plane is not staged (box too large, non-finite gradients), or this pixel's taps lie outside the box (wild coordinates)
        if (__any(active && any_w && !(staged && in_box))) {
            if (active && any_w && !(staged && in_box)) {
                float* __restrict__ gp = gvol + static_cast<int64_t>(k) * b.gs_plane;
                const int64_t oa = static_cast<int64_t>(y0) * b.gs_row + x0, ob = oa + b.gs_row;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    float* __restrict__ gc = gp + c4 * b.gs_chan;
                    const float d = gq.d[c4];
                    if (nw != 0.0f) atomicAdd(gc + oa, d * nw);        // (a weight that is not zero belongs to a tap inside the texture)
                    if (ne != 0.0f) atomicAdd(gc + oa + 1, d * ne);
                    if (sw != 0.0f) atomicAdd(gc + ob, d * sw);
                    if (se != 0.0f) atomicAdd(gc + ob + 1, d * se);
                }
            }
        }
    };
    // flush + reset of one plane's box by the FLUSH waves.  A wave takes whole lines (row, channel) of the box, a lane one texel column: what is
    // uniform per line -- the LDS line, the row and channel offsets of the gradient volume -- is scalar arithmetic, a lane spends an LDS read, a
    // test, a conversion and one global atomic per texel (consecutive lanes -> consecutive addresses); four lines' reads in flight per wave.
    // (Round 5's first version dealt the box out item by item: a 32-bit division by the box width and five quarter-rate integer multiplies
    // per item -- half of the kernel's VALU time.)
    constexpr int kFW = kB2Flush / 64;
    const int fw = __builtin_amdgcn_readfirstlane(ftid >> 6);
    const int flane = ftid & 63;
    // (Written without `break` / `continue` in the unrolled bodies: round 6's first version had them, the compiler turned them into per-lane state
    //  machines, and the flush -- the kernel's critical path -- took 38 % longer: profiles/r06_backward.txt.)
    auto flush = [&](int t, int k, uint32_t* __restrict__ bx_acc) {
        const int4 bb = box[t];
        const uint32_t mb = gmax[t];
        if (!(bb.z > 0 && mb < 0x7f800000u && mb != 0u)) return;
        const int hword = __float_as_int(pcB[t].y);
        const int hbits = hword & 0xff;
        const bool wide = __builtin_amdgcn_readfirstlane(hword & 0x100) != 0;
        const int shf = min((wide ? 41 : 29) - hbits - (static_cast<int>(mb >> 23) - 127), 126);
        const float inv = __builtin_amdgcn_ldexpf(1.0f, -shf);
        const int nx = __builtin_amdgcn_readfirstlane(bb.z), nlines = __builtin_amdgcn_readfirstlane(bb.w) * 4;
        const int bx = __builtin_amdgcn_readfirstlane(bb.x), by = __builtin_amdgcn_readfirstlane(bb.y);
        float* __restrict__ gp = gvol + static_cast<int64_t>(k) * b.gs_plane + static_cast<int64_t>(by) * b.gs_row + bx;
        if (wide) {   // two words per cell (see the tables): value = hi * 2^12 + lo.  (Textures several times coarser than the image: small boxes, atomics only.)
            for (int c0 = 0; c0 < nx; c0 += 64) {
                const bool on = c0 + flane < nx;
                uint32_t* __restrict__ src0 = bx_acc + c0 + flane;
                for (int line = fw; line < nlines; line += kFW) {
                    const int qh = on ? static_cast<int>(src0[line * kB2Pitch]) : 0, ql = on ? static_cast<int>(src0[line * kB2Pitch + kB2Cap / 2]) : 0;
                    if ((qh | ql) != 0) {
                        src0[line * kB2Pitch] = 0u, src0[line * kB2Pitch + kB2Cap / 2] = 0u;
                        const float v = __builtin_fmaf(static_cast<float>(qh), 4096.0f, static_cast<float>(ql)) * inv;
                        if (!abl_noglobal) atomicAdd(gp + (static_cast<uint32_t>(line & 3) * gs_chan + static_cast<uint32_t>(line >> 2) * gs_row + static_cast<uint32_t>(c0)) + flane, v);
                    }
                }
            }
            return;
        }
        for (int c0 = 0; c0 < nx; c0 += 64) {   // (one pass for boxes of up to 64 columns)
            const bool on = c0 + flane < nx;
            uint32_t* __restrict__ src0 = bx_acc + c0 + flane;
            for (int l0 = fw; l0 < nlines; l0 += 4 * kFW) {
                int q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int line = l0 + u * kFW;
                    q[u] = (on && line < nlines) ? static_cast<int>(src0[line * kB2Pitch]) : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int line = l0 + u * kFW;
                    if (q[u] != 0) {
                        src0[line * kB2Pitch] = 0u;
                        if (!abl_noglobal)
                            atomicAdd(gp + (static_cast<uint32_t>(line & 3) * gs_chan + static_cast<uint32_t>(line >> 2) * gs_row + static_cast<uint32_t>(c0)) + flane,
                                      static_cast<float>(q[u]) * inv);
                    }
                }
            }
        }
    };

    const int cx0 = txi * kB2TW, cx1 = min(cx0 + kB2TW - 1, p.W - 1);
    const int cy0 = tyi * kB2TH, cy1 = min(cy0 + kB2TH - 1, p.H - 1);
    const int npix = (cx1 - cx0 + 1) * (cy1 - cy0 + 1);
#ifdef GMPI_PROF  // per-phase shader-clock totals of one pixel wave and one flush wave (status words 8..15 | 16..23): tools/time_backward.py prints them
    uint32_t prof_acc[6] = {0, 0, 0, 0, 0, 0};
    uint64_t prof_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_last));
    const uint64_t prof_start = prof_last;
#define B2_STAMP(i) do { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); prof_acc[i] += static_cast<uint32_t>(now_ - prof_last); prof_last = now_; } while (0)
#else
#define B2_STAMP(i) do { } while (0)
#endif
    auto build_tables = [&](int kc, int kn) {
        for (int t = tid; t < kn; t += kB2Threads) {
            const int k = kc + t;
            const float zdiff = dhw[3 * k] - ez, ph = dhw[3 * k + 1], pw = dhw[3 * k + 2];
            float mnx = __builtin_inff(), mxx = -__builtin_inff(), mny = mnx, mxy = mxx;
            float cix[4], ciy[4];   // images of the tile's corner pixels (cx0, cy0), (cx1, cy0), (cx0, cy1), (cx1, cy1)
            bool finite = true;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t q = static_cast<int64_t>((c & 2) ? cy1 : cy0) * p.W + ((c & 1) ? cx1 : cx0);
                float ix, iy, s, u, v;
                plane_coord<AC>(zdiff, ph, pw, ex, ey, rdv[q], rdv[HW + q], rdv[2 * HW + q], cx, cy, ix, iy, s, u, v);
                finite = finite && (fabsf(ix) < 1e6f) && (fabsf(iy) < 1e6f);
                mnx = fminf(mnx, ix), mxx = fmaxf(mxx, ix), mny = fminf(mny, iy), mxy = fmaxf(mxy, iy);
                cix[c] = ix, ciy[c] = iy;
            }
            int4 bb = make_int4(0, 0, 0, 0);
            int hbits = 11;
            if (finite) {
                const float eps = 1.0f / 64;
                bb.x = static_cast<int>(floorf(mnx - eps)), bb.y = static_cast<int>(floorf(mny - eps));
                bb.z = static_cast<int>(floorf(mxx + eps)) + 2 - bb.x, bb.w = static_cast<int>(floorf(mxy + eps)) + 2 - bb.y;
                if (bb.z > kB2Pitch || bb.w > kB2Rows) bb.z = 0;
                else {
                    // taps that can meet in one texel: 4 x the tile's pixels per texel on average; 8 x + 8 is the bound used (a homography is
                    // smooth over a tile), 2^11 = every tap of the tile when the footprint is only a few texels.  The density comes from the AREA of
                    // the quadrilateral the tile's corner pixels map to (round 6; round 5 took the bounding box, which a rolled or sheared footprint
                    // fills only partly: its density is then higher than the box average)
                    const float quad2 = fabsf((cix[0] * ciy[1] - cix[1] * ciy[0]) + (cix[1] * ciy[3] - cix[3] * ciy[1]) + (cix[3] * ciy[2] - cix[2] * ciy[3]) +
                                              (cix[2] * ciy[0] - cix[0] * ciy[2]));
                    const int area = max(min(static_cast<int>(0.5f * quad2), (bb.z - 1) * (bb.w - 1)), 1);
                    const int bound = (8 * npix + area - 1) / area + 8;
                    hbits = min(11, 32 - __clz(bound - 1));
#ifdef GMPI_B2_HBITS  // (experiment: a fixed headroom)
                    hbits = GMPI_B2_HBITS;
#endif
                    // WIDE planes: with 9 and more bits of headroom (a texture several times coarser than the image: a hundred and more taps
                    // per texel) one 32-bit word leaves less than 2^-21 of the tile's largest gradient per add -- the round-1 kernel's 64-bit sums
                    // were measurably better there (tools/fuzz_backward_gpu.py: 1.2e-5 against 2.5e-6 of the largest gradient).  Such a box is
                    // small: when its rows fill at most half of the buffer, every cell gets a second word (the residual of the first, 12 bits
                    // finer) in the other half.
                    if (hbits >= 9 && bb.w * 4 * kB2Pitch <= kB2Cap / 2) hbits |= 0x100;
                }
            }
            box[t] = bb;
            gmax[t] = 0u;
            const float hw = pw * 0.5f, hh = ph * 0.5f;
            pcA[t] = make_float4(zdiff, hw, hh, 1.0f / hw);
            pcB[t] = make_float2(1.0f / hh, __int_as_float(hbits));
        }
    };
    // Both roles pass the same sequence of barriers: per chunk 3 (tables free | tables built | last plane's maximum) + one per plane.
    if (flusher) {
        for (int kend = p.D; kend > 0; kend -= kB2Chunk) {  // chunks of planes, last chunk first
            const int kc = max(kend - kB2Chunk, 0), kn = kend - kc;
            lds_barrier();
            build_tables(kc, kn);
            lds_barrier();
            lds_barrier();
            B2_STAMP(5);
            for (int t = kn - 1; t >= 0; --t) {
                if (t + 1 < kn) flush(t + 1, kc + t + 1, acc[(t + 1) & 1]);        // plane t + 1: box -> global atomics (complete since the last barrier)
                B2_STAMP(0);
                lds_barrier();
                B2_STAMP(3);
            }
            flush(0, kc, acc[0]);
        }
    } else {
        for (int kend = p.D; kend > 0; kend -= kB2Chunk) {
            const int kc = max(kend - kB2Chunk, 0), kn = kend - kc;
            lds_barrier();  // previous chunk's tables and boxes are done with; (first pass) the zero fill is complete
            build_tables(kc, kn);
            lds_barrier();
            Tap tq;
            Grad gq;
#pragma unroll
            for (int i = 0; i < 16; ++i) tq.v[i] = 0.0f;
            tq.s = tq.wx1 = tq.wy1 = 0.0f, tq.x0 = tq.y0 = -2;
            gq.d[0] = gq.d[1] = gq.d[2] = gq.d[3] = 0.0f, gq.nw = gq.ne = gq.sw = gq.se = 0.0f, gq.x0 = gq.y0 = -2;
            // prologue: gradients of the chunk's last plane, taps of the one in front of it in flight
            uint32_t mbits = 0u;
            if (active) {
                fetch(kn - 1, kc + kn - 1, tq);
                mbits = grads(kn - 1, tq, gq);
                if (kn >= 2) fetch(kn - 2, kc + kn - 2, tq);
            }
            tile_max(kn - 1, mbits);
            lds_barrier();  // gmax[kn - 1] complete
            B2_STAMP(5);
            for (int t = kn - 1; t >= 0; --t) {
                scatter(t, kc + t, gq, acc[t & 1]);                               // plane t: gradients -> box (gmax[t] complete since the last barrier)
                B2_STAMP(0);
                mbits = 0u;
                if (active && t >= 1) mbits = grads(t - 1, tq, gq);               // plane t - 1: taps landed -> gradients
                if (t >= 1) tile_max(t - 1, mbits);                               //              ... and their tile maximum
                B2_STAMP(1);
                if (active && t >= 2) fetch(t - 2, kc + t - 2, tq);               // plane t - 2: taps into flight
                B2_STAMP(2);
                lds_barrier();                                                    // box of plane t and gmax[t - 1] complete
                B2_STAMP(3);
            }
        }
    }
#ifdef GMPI_PROF
    if (p.status != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && (tid == 192 || tid == kB2Pix + 64)) {
        uint64_t now_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_));
        uint32_t* o = p.status + (flusher ? 16 : 8);
        for (int i = 0; i < 6; ++i) o[i] = prof_acc[i];
        o[6] = static_cast<uint32_t>(now_ - prof_start);
    }
#endif
}

template <typename TexT>
static hipError_t launch_backward_t(const KParams& p, const BwdParams& b, bool tiles, hipStream_t stream) {
    const bool ac = p.flags & 1u;
    if (tiles) {
        const int tiles_x = (p.W + kBwdTW - 1) / kBwdTW, tiles_y = (p.H + kBwdTH - 1) / kBwdTH;
        const dim3 grid(tiles_x * tiles_y, p.N), block(kBwdThreads), block2(kB2Threads);
        // the round-5 kernel wants two columns (pair loads) and a plane's offsets -- volume and gradient -- in 32 bits
        const int es = static_cast<int>(sizeof(TexT));
        bool v1 = p.Wt < 2 || (3 * p.s_chan + static_cast<int64_t>(p.Ht) * p.s_row + p.Wt) * es >= (int64_t(1) << 31) ||
                  (3 * b.gs_chan + static_cast<int64_t>(p.Ht) * b.gs_row + p.Wt) >= (int64_t(1) << 31) || p.s_chan < 0 || p.s_row < 0;
#ifdef GMPI_TUNE
        v1 = v1 || (p.flags & (1u << 22)) != 0;  // GMPI_TUNE_SKIP=64: the round-1 tile kernel (A/B)
#endif
        if (v1) {
            if (ac) hipLaunchKernelGGL((render_backward_tile_kernel<TexT, true>), grid, block, 0, stream, p, b, tiles_x);
            else hipLaunchKernelGGL((render_backward_tile_kernel<TexT, false>), grid, block, 0, stream, p, b, tiles_x);
            return hipGetLastError();
        }
        // Round 6: 64 x 8 pixel tiles (32 x 16 in round 5)
        int geo = kBwdDefaultGeo;
#ifdef GMPI_TUNE  // GMPI_TUNE_BWD: 1 = round 5's 32 x 16 tiles, 2 = 64 x 8 tiles
        static const int env_geo = [] { const char* e = getenv("GMPI_TUNE_BWD"); return e ? atoi(e) : 0; }();
        if (env_geo > 0) geo = env_geo;
#endif
        auto go = [&](auto geo_tag) {
            using G = decltype(geo_tag);
            const int tx = (p.W + G::TW - 1) / G::TW, ty = (p.H + G::TH - 1) / G::TH;
            const dim3 grid2(xcd_grid_per_group(tx * ty, tx * ty), p.N);
            if (ac) hipLaunchKernelGGL((render_backward_tile2_kernel<TexT, true, G>), grid2, block2, 0, stream, p, b, tx);
            else hipLaunchKernelGGL((render_backward_tile2_kernel<TexT, false, G>), grid2, block2, 0, stream, p, b, tx);
        };
        if (geo == 1) go(B2Tall{});
        else go(B2Wide{});
        return hipGetLastError();
    }
    const dim3 block(64, 4), grid((p.W + 63) / 64, (p.H + 3) / 4, p.N);
    if (ac) hipLaunchKernelGGL((render_backward_kernel<TexT, true>), grid, block, 0, stream, p, b);
    else hipLaunchKernelGGL((render_backward_kernel<TexT, false>), grid, block, 0, stream, p, b);
    return hipGetLastError();
}

// `tiles`: stage the scatter per pixel tile in LDS (default); false = one pixel per lane, 16 global atomics each
// (GMPI_VARIANT_GATHER: the simple kernel, kept as the cross-check)
bool backward_gather_supports(const KParams& p);                                                          // render_backward_gather.hip
uint64_t backward_gather_workspace_bytes(const KParams& p);                                                // render_backward_gather.hip
hipError_t launch_backward_gather(const KParams& p, int dtype, const BwdParams& b, bool overwrite, hipStream_t stream);   // render_backward_gather.hip

hipError_t launch_backward(const KParams& p0, int dtype, const float* g_rgb, const float* g_depth, float* g_rgba,
                           const int64_t* gstride, bool tiles, hipStream_t stream) {
    KParams p = p0;
#ifdef GMPI_TUNE  // profiling builds only: 16 = no global atomics in the flush, 32 = no LDS atomics, 64 = the round-1 tile kernel, 128 = no tap loads, 256 = tiles in row-major order over the XCDs
    static const unsigned skip = [] { const char* e = getenv("GMPI_TUNE_SKIP"); return e ? static_cast<unsigned>(atoi(e)) : 0u; }();
    p.flags |= skip << 16;
#endif
    BwdParams b;
    b.g_rgb = g_rgb, b.g_depth = g_depth, b.g_rgba = g_rgba;
    b.gs_mpi = gstride[0], b.gs_plane = gstride[1], b.gs_chan = gstride[2], b.gs_row = gstride[3];
    if (p.N > 65535) tiles = false;  // grid.y
    // Round 6: with a workspace for the sample gradients (gmpi_render_backward_workspace_bytes) the atomics-free pair -- pixel pass + texel gather,
    // render_backward_gather.hip -- takes the launch: every cell of the gradient is written once by its owner.
    bool gather = tiles && p.ws != nullptr && backward_gather_supports(p) && p.ws_bytes >= backward_gather_workspace_bytes(p) &&
                  reinterpret_cast<uintptr_t>(p.ws) % 256 == 0;
#ifdef GMPI_TUNE  // GMPI_TUNE_BWD = 1 | 2: the tile kernels even with a workspace
    {
        static const int env_geo = [] { const char* e = getenv("GMPI_TUNE_BWD"); return e ? atoi(e) : 0; }();
        if (env_geo == 1 || env_geo == 2) gather = false;
    }
#endif
    if (gather) return launch_backward_gather(p, dtype, b, (p.flags & (1u << 7)) != 0 /* GMPI_FLAG_GRAD_OVERWRITE */, stream);
    switch (dtype) {
        case 0: return launch_backward_t<float>(p, b, tiles, stream);
        case 1: return launch_backward_t<bf16_t>(p, b, tiles, stream);
        default: return launch_backward_t<f16_t>(p, b, tiles, stream);
    }
}

}  // namespace gmpi
