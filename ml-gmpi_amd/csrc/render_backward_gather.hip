// render_backward_gather.hip -- the backward of the render WITHOUT atomics (round 6): gradient of the fused render w.r.t. the RGBA volume
// (the reference's G-step back-propagates through MPIRenderer.render into the generator: gmpi/train.py:740-779; the sampling grid carries no
// gradient, gmpi/core/mpi.py:65).
//
// Why.  The tile kernel (render_backward.hip) scatters every pixel's sample gradients into a copy of its tile's texel box in LDS and flushes the box
// with one global atomic per texel: it runs AT the part's rate for atomic 64-byte segments (20.8 G/s: profiles/r05_backward.txt, r06_backward.txt --
// 2.12 ms at 1024^2 x 32 x 4 where the launch without its global writes takes 1.27), needs a zero-filled gradient volume (0.32 ms) and, with the
// write path behind the L2s as the shared resource, cannot be helped by storing what a tile owns (built and measured in round 6).  This file replaces the
// scatter by a GATHER, so that every gradient cell is written exactly once, by the one workgroup that owns it:
//   pass 1 (pixel_pass_kernel), pixel-stationary: the back-to-front sweep of render_backward.hip (same taps, same arithmetic), but instead of
//           scattering, every pixel WRITES its four sample gradients d_s = dL/d(r, g, b, alpha sample) per plane: G[n][k][py][px], 16 bytes,
//           coalesced -- no LDS boxes, no barriers in the plane loop, no flush waves;
//   pass 2 (texel_gather_kernel), texel-stationary: a workgroup owns a 64 x 16 texel tile of one plane of one MPI.  The pixels whose bilinear
//           footprint can touch the tile lie in the pre-image of the tile (grown by one texel) under the plane's homography; the workgroup stages
//           their exact sample positions -- recomputed with the forward's own coordinate chain -- and their d_s in LDS, and every texel sums
//           w(ix - x) w(iy - y) d_s over its candidates, w(t) = max(0, 1 - |t|): exactly the cells, weights and zeros padding of
//           F.grid_sample's backward, in a fixed order (the result is deterministic, which the atomics never were).  Membership is decided by the
//           exact positions; the homography only has to bound where to look.
// The homography pixel -> texel of a (view, plane) is solved in fp64 from the images of the four image-corner pixels (homography_kernel: a pinhole ray
// field maps pixel lines to lines -- the assumption every staged kernel of this library makes, include/gmpi_render.h); a (view, plane) whose solve is
// not trustworthy (degenerate rays, a plane behind the camera) is marked and its texel tiles look at EVERY pixel of the view: slow, still exact.
// align_corners = True only (mpi.py:98-99 makes the other mode's map discontinuous at |u| = 1: the tile kernel keeps that mode), uniform views per
// MPI (several views of one MPI are summed in registers), a workspace of N D H W 16 bytes lent by the caller (gmpi_render_backward_workspace_bytes).
#include "gmpi_backward.hpp"

#include <algorithm>
#include <cstdlib>

namespace gmpi {
namespace bwdg {

typedef float f32x4 __attribute__((ext_vector_type(4)));   // (the element types of the sample-gradient and sample-position buffers: the non-temporal
typedef float f32x2 __attribute__((ext_vector_type(2)));   //  builtins want vector types)

// per (view, plane): the INVERSE homography texel -> pixel,  px = (a x + b y + c) / den,  py = (d x + e y + f) / den,  den = g x + h y + 1
struct HRec {
    float a, b, c, d, e, f, g, h;
    float valid;   // 1: usable; 0: look at every pixel
    float pad[3];
};
static_assert(sizeof(HRec) == 48, "record");

// ---- 1. homography of every (view, plane) ------------------------------------------------------------------------------------------------------------
// Closed form (unit square -> quadrilateral, Heckbert 1989) in fp64 from the exact images of the four image-corner pixels, then the adjugate: straight-line
// code, a few microseconds for the whole launch.  (A first version solved the 8 x 8 DLT system by Gaussian elimination out of scratch memory: 81 us.)
template <bool AC>
__global__ __launch_bounds__(64) void homography_kernel(const KParams p, HRec* __restrict__ recs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N * p.D) return;
    const int n = i / p.D, k = i - n * p.D;
    uint32_t bad_index = 0;
    const int m = view_mpi(p, n, bad_index);
    const float* __restrict__ dhw = p.dhw + (static_cast<int64_t>(m) * p.D + k) * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const int64_t HW = static_cast<int64_t>(p.H) * p.W;
    const float* __restrict__ rdv = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
    const float cx = AC ? static_cast<float>(p.Wt - 1) * 0.5f : static_cast<float>(p.Wt), cy = AC ? static_cast<float>(p.Ht - 1) * 0.5f : static_cast<float>(p.Ht);
    HRec r{};
    r.valid = 0.0f;
    bool ok = p.W >= 2 && p.H >= 2;
    double X[4], Y[4];   // images of the pixels (0, 0), (W - 1, 0), (W - 1, H - 1), (0, H - 1): the unit square's corners in order
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int qx = (c == 1 || c == 2) ? p.W - 1 : 0, qy = (c >= 2) ? p.H - 1 : 0;
        const int64_t q = static_cast<int64_t>(qy) * p.W + qx;
        float ix, iy, s, u, v;
        plane_coord<AC>(dhw[0] - ez, dhw[1], dhw[2], ex, ey, rdv[q], rdv[HW + q], rdv[2 * HW + q], cx, cy, ix, iy, s, u, v);
        ok = ok && fabsf(ix) < 1e6f && fabsf(iy) < 1e6f && s > 0.0f;   // (NaN fails; s > 0: the corner pixel sees the plane in front of the camera)
        X[c] = ix, Y[c] = iy;
    }
    if (ok) {
        // (u, v) in the unit square -> texel:  X = (a u + b v + c) / (g u + h v + 1), Y = (d u + e v + f) / (g u + h v + 1)
        const double sx = X[0] - X[1] + X[2] - X[3], sy = Y[0] - Y[1] + Y[2] - Y[3];
        const double dx1 = X[1] - X[2], dx2 = X[3] - X[2], dy1 = Y[1] - Y[2], dy2 = Y[3] - Y[2];
        const double den = dx1 * dy2 - dx2 * dy1;
        ok = fabs(den) > 1e-12;
        if (ok) {
            const double g = (sx * dy2 - dx2 * sy) / den, hh = (dx1 * sy - sx * dy1) / den;
            // pixel coordinates: u = px / (W - 1), v = py / (H - 1)
            const double iu = 1.0 / (p.W - 1), iv_ = 1.0 / (p.H - 1);
            double h[9];
            h[0] = (X[1] - X[0] + g * X[1]) * iu, h[1] = (X[3] - X[0] + hh * X[3]) * iv_, h[2] = X[0];
            h[3] = (Y[1] - Y[0] + g * Y[1]) * iu, h[4] = (Y[3] - Y[0] + hh * Y[3]) * iv_, h[5] = Y[0];
            h[6] = g * iu, h[7] = hh * iv_, h[8] = 1.0;
            // the forward denominators at the four corners must be positive (no pixel line of the image passes through the map's pole)
            ok = ok && 1.0 > 1e-6 && g + 1.0 > 1e-6 && g + hh + 1.0 > 1e-6 && hh + 1.0 > 1e-6;
            // inverse by the adjugate
            const double det = h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) + h[2] * (h[3] * h[7] - h[4] * h[6]);
            double iv[9];
            iv[0] = h[4] * h[8] - h[5] * h[7], iv[1] = h[2] * h[7] - h[1] * h[8], iv[2] = h[1] * h[5] - h[2] * h[4];
            iv[3] = h[5] * h[6] - h[3] * h[8], iv[4] = h[0] * h[8] - h[2] * h[6], iv[5] = h[2] * h[3] - h[0] * h[5];
            iv[6] = h[3] * h[7] - h[4] * h[6], iv[7] = h[1] * h[6] - h[0] * h[7], iv[8] = h[0] * h[4] - h[1] * h[3];
            ok = ok && fabs(det) > 1e-300 && fabs(iv[8]) > 1e-300;
            if (ok) {
                const double s8 = 1.0 / iv[8];
#pragma unroll
                for (int j = 0; j < 9; ++j) iv[j] *= s8;
                // the inverse takes the corners' images back to the corner pixels, with positive denominators
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double wantx = (c == 1 || c == 2) ? p.W - 1 : 0, wanty = (c >= 2) ? p.H - 1 : 0;
                    const double dn = iv[6] * X[c] + iv[7] * Y[c] + 1.0;
                    ok = ok && dn > 1e-9;
                    const double bx = (iv[0] * X[c] + iv[1] * Y[c] + iv[2]) / dn, by = (iv[3] * X[c] + iv[4] * Y[c] + iv[5]) / dn;
                    ok = ok && fabs(bx - wantx) < 1e-3 && fabs(by - wanty) < 1e-3;
                }
                if (ok) {
                    r.a = static_cast<float>(iv[0]), r.b = static_cast<float>(iv[1]), r.c = static_cast<float>(iv[2]);
                    r.d = static_cast<float>(iv[3]), r.e = static_cast<float>(iv[4]), r.f = static_cast<float>(iv[5]);
                    r.g = static_cast<float>(iv[6]), r.h = static_cast<float>(iv[7]);
                    r.valid = 1.0f;
                }
            }
        }
    }
    recs[i] = r;
}

// ---- 2. pixel pass: the back-to-front sweep; every pixel writes its sample position and its four sample gradients per plane ---------------------------
constexpr int kPT = 512, kPTW = 64, kPTH = 8;   // one pixel per thread; a wave = one pixel row of the tile
constexpr int kPChunk = 96;

template <typename TexT, bool AC>
__global__ __launch_bounds__(kPT, 6) void pixel_pass_kernel(const KParams p, const BwdParams b, f32x2* __restrict__ P, f32x4* __restrict__ G, const int tiles_x) {
    __shared__ float4 pcA[kPChunk];      // zdiff, w/2, h/2, RN(2/w)
    __shared__ float pcB[kPChunk];       // RN(2/h)
    constexpr int kES = static_cast<int>(sizeof(TexT));
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int n_tiles = tiles_x * ((p.H + kPTH - 1) / kPTH);
    const int tile = xcd_item_per_group(static_cast<int>(blockIdx.x), n_tiles, n_tiles);   // XCD x = blockIdx % 8 takes a contiguous run of the view's tiles
    if (tile >= n_tiles) return;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int px = txi * kPTW + (tid % kPTW), py = tyi * kPTH + (tid / kPTW);
    const bool active = px < p.W && py < p.H;
    uint32_t bad_index = 0;
    const int m = view_mpi(p, n, bad_index);
    const float* __restrict__ dhw = p.dhw + static_cast<int64_t>(m) * p.D * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const int64_t HW = static_cast<int64_t>(p.H) * p.W;
    const int64_t pix = static_cast<int64_t>(min(py, p.H - 1)) * p.W + min(px, p.W - 1);
    const float* __restrict__ rdv = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
    const int Ht = p.Ht, Wt = p.Wt;
    const float cx = AC ? static_cast<float>(Wt - 1) * 0.5f : static_cast<float>(Wt);
    const float cy = AC ? static_cast<float>(Ht - 1) * 0.5f : static_cast<float>(Ht);
    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    const uint32_t s_chan_b = static_cast<uint32_t>(p.s_chan) * kES, s_row_b = static_cast<uint32_t>(p.s_row) * kES;

    const float rx = rdv[pix], ry = rdv[HW + pix], rz = rdv[2 * HW + pix];
    const float rrz = 1.0f / rz;
    float dot = rx * p.z_dir[3 * n + 0];
    dot = dot + ry * p.z_dir[3 * n + 1];
    dot = dot + rz * p.z_dir[3 * n + 2];
    const float oscale = (p.flags & (1u << 1)) ? 2.0f : 1.0f;  // forward wrote 2*C-1 (mpi_renderer.py:467)
    const float* __restrict__ g = b.g_rgb + static_cast<int64_t>(n) * 3 * HW + pix;
    const float gr = active ? oscale * g[0] : 0.f, gg = active ? oscale * g[HW] : 0.f, gb = active ? oscale * g[2 * HW] : 0.f;
    const float gz = (active && b.g_depth) ? b.g_depth[static_cast<int64_t>(n) * HW + pix] : 0.0f;
    const float t_fwd = (active && p.T_out) ? p.T_out[static_cast<int64_t>(n) * HW + pix] : 1.0f;
    XT T{1.0f, 0};
    if (active) T = total_transmittance<TexT, AC>(p, dhw, vol, t_fwd, p.T_out != nullptr, ex, ey, ez, rx, ry, rz, cx, cy);
    float S = 0.0f;

    struct Tap { float s, wx1, wy1, ix, iy; int x0, y0; float v[16]; };   // v: per channel (top p0, p1 | bottom p0, p1)
    auto fetch = [&](int t, int k, Tap& q) {   // coordinates of this pixel on plane k (chunk-local t) + its 8 pair loads
        const float4 a = pcA[t];
        float ix, iy;
        plane_coord_recip<AC>(a.x, a.y, a.z, a.w, pcB[t], ex, ey, rx, ry, rz, rrz, cx, cy, ix, iy, q.s);
        const float fx = floorf(ix), fy = floorf(iy);
        q.wx1 = ix - fx, q.wy1 = iy - fy, q.ix = ix, q.iy = iy;
        q.x0 = (fx >= -2.0f && fx <= static_cast<float>(Wt)) ? static_cast<int>(fx) : -2;   // (NaN / huge coordinates: out of range, all weights 0)
        q.y0 = (fy >= -2.0f && fy <= static_cast<float>(Ht)) ? static_cast<int>(fy) : -2;
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
    auto grads = [&](const Tap& q) -> float4 {   // the taps have landed: bilinear samples, the plane's gradients (the arithmetic of BwdPixel::plane)
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
        const float alpha = smp[3];
        const float om = (1.0f - alpha) + 1e-10f;
        float r = __builtin_amdgcn_rcpf(om);
        r = __builtin_fmaf(__builtin_fmaf(-om, r, 1.0f), r, r);      // Newton step: r = RN(1 / om) up to the last bit
        T.m = div_by_recip(T.m, om, r);
        T.renorm();
        const float Tk = T.value();
        const float qv = gr * smp[0] + gg * smp[1] + gb * smp[2] + gz * (q.s * dot);
        const float w = alpha * Tk;
        const float4 d = make_float4(gr * w, gg * w, gb * w, Tk * qv - div_by_recip(S, om, r));
        S += w * qv;
        return d;
    };

    f32x4* __restrict__ Gpix = G + static_cast<int64_t>(n) * p.D * HW + pix;   // + k * HW per plane
    f32x2* __restrict__ Ppix = P + static_cast<int64_t>(n) * p.D * HW + pix;
    for (int kend = p.D; kend > 0; kend -= kPChunk) {  // chunks of planes, last chunk first
        const int kc = max(kend - kPChunk, 0), kn = kend - kc;
        __syncthreads();
        for (int t = tid; t < kn; t += kPT) {
            const int k = kc + t;
            const float hw = dhw[3 * k + 2] * 0.5f, hh = dhw[3 * k + 1] * 0.5f;
            pcA[t] = make_float4(dhw[3 * k] - ez, hw, hh, 1.0f / hw);
            pcB[t] = 1.0f / hh;
        }
        __syncthreads();
        if (active) {   // (no barrier below: a pixel is on its own through the planes; the taps of plane t - 1 fly while plane t's gradients are formed)
            Tap tq;
            fetch(kn - 1, kc + kn - 1, tq);
            for (int t = kn - 1; t >= 0; --t) {
                const f32x2 pv = {tq.ix, tq.iy};   // the exact sample position (the forward's chain): the texel pass decides membership on it
                const float4 d = grads(tq);
                if (t >= 1) fetch(t - 1, kc + t - 1, tq);
                const f32x4 dv = {d.x, d.y, d.z, d.w};
                __builtin_nontemporal_store(pv, Ppix + static_cast<int64_t>(kc + t) * HW);   // (read once, by the texel pass: keep them out of the way of the taps)
                __builtin_nontemporal_store(dv, Gpix + static_cast<int64_t>(kc + t) * HW);
            }
        }
    }
}

// ---- 3. texel pass: every cell of the gradient volume is the sum over the pixels that sampled it -----------------------------------------------------
// A workgroup owns a 64 x 16 texel tile of one MPI over a run of planes; a thread the texels (tx, ty) and (tx, ty + 8).  Per (plane, view) the pixels that
// can reach the tile -- the pre-image of the tile grown by a texel, a convex quadrilateral, as a pixel box -- are staged in LDS (position 8 B + sample
// gradients 16 B) in chunks of up to kCap pixels; the loads of the NEXT chunk (registers) fly while the current one is gathered.
constexpr int kTT = 512, kTX = 64, kTY = 16;
constexpr int kCap = 2048;                          // staged pixels per chunk: 48 KB
constexpr int kSlots = kCap / kTT;                  // pixels a thread moves per chunk
constexpr int kChunkW = 512;                        // widest chunk (a pixel box wider than this is walked in column blocks)

struct Chunk {   // everything wave-uniform
    int k, v, n;             // plane, view of the MPI, view index
    int bx0, by0, cw, ch;    // pixel rectangle of this chunk
    int pxa, pxb, pya, pyb;  // pixel box of the (plane, view)
    int ncx, ncy;            // candidates per texel and axis (0: no usable homography -> every staged pixel)
    float rxu, ryu;          // the tile's radii
    HRec r;
};

template <bool OVERWRITE>
__global__ __launch_bounds__(kTT, 6) void texel_gather_kernel(const KParams p, const BwdParams b, const f32x2* __restrict__ P, const f32x4* __restrict__ G,
                                                              const HRec* __restrict__ recs, const int tiles_x, const int tiles_y, const int planes_per_wg) {
    __shared__ float2 sXY[kCap];
    __shared__ float4 sD[kCap];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = tiles_x * tiles_y;
    const int tile = xcd_item_per_group(static_cast<int>(blockIdx.x), n_tiles, n_tiles);
    if (tile >= n_tiles) return;
    const int m = blockIdx.z;
    const int k_first = blockIdx.y * planes_per_wg, k_last = min(k_first + planes_per_wg, p.D);   // [k_first, k_last)
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int X0 = txi * kTX, Y0 = tyi * kTY;
    const int x = X0 + lane, y0 = Y0 + wave, y1 = y0 + 8;   // this thread's two texels
    const float xf = static_cast<float>(x), yf0 = static_cast<float>(y0), yf1 = static_cast<float>(y1);
    const int Wt = p.Wt, Ht = p.Ht, W = p.W, H = p.H;
    const int64_t HW = static_cast<int64_t>(H) * W;

    auto inv_map = [&](const HRec& r, float tx, float ty, float& ppx, float& ppy, float& radx, float& rady) {
        // texel -> pixel: position and how far a texel's 2 x 2 support reaches in pixels (first order, with slack)
        const float den = __builtin_fmaf(r.g, tx, __builtin_fmaf(r.h, ty, 1.0f));
        const float rd = 1.0f / den;
        ppx = __builtin_fmaf(r.a, tx, __builtin_fmaf(r.b, ty, r.c)) * rd;
        ppy = __builtin_fmaf(r.d, tx, __builtin_fmaf(r.e, ty, r.f)) * rd;
        radx = (fabsf(r.a - r.g * ppx) + fabsf(r.b - r.h * ppx)) * fabsf(rd) * 1.03f + 0.08f;
        rady = (fabsf(r.d - r.g * ppy) + fabsf(r.e - r.h * ppy)) * fabsf(rd) * 1.03f + 0.08f;
    };
    // the pixel box of (plane k, view v): false when no pixel of the view reaches the tile
    auto open_view = [&](Chunk& c) -> bool {
        c.n = m * p.views_per_mpi + c.v;
        if (c.n >= p.N) return false;
        c.r = recs[static_cast<int64_t>(c.n) * p.D + c.k];
        int pxa = 0, pxb = W - 1, pya = 0, pyb = H - 1, ncx = 0, ncy = 0;
        float rxu = 0.0f, ryu = 0.0f;
        if (c.r.valid != 0.0f) {
            float mnx = 3e38f, mxx = -3e38f, mny = 3e38f, mxy = -3e38f, rxm = 0.0f, rym = 0.0f;
            bool okq = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // the tile grown by a texel on every side
                const float tx = static_cast<float>((q & 1) ? min(X0 + kTX, Wt) : X0 - 1), ty = static_cast<float>((q & 2) ? min(Y0 + kTY, Ht) : Y0 - 1);
                okq = okq && __builtin_fmaf(c.r.g, tx, __builtin_fmaf(c.r.h, ty, 1.0f)) > 1e-6f;
                float qx, qy, ax, ay;
                inv_map(c.r, tx, ty, qx, qy, ax, ay);
                mnx = fminf(mnx, qx), mxx = fmaxf(mxx, qx), mny = fminf(mny, qy), mxy = fmaxf(mxy, qy);
                rxm = fmaxf(rxm, ax), rym = fmaxf(rym, ay);
            }
            if (okq && mxx - mnx < 1e6f && mxy - mny < 1e6f && rxm < 64.0f && rym < 64.0f) {
                pxa = max(static_cast<int>(floorf(mnx - 0.25f)), 0), pxb = min(static_cast<int>(ceilf(mxx + 0.25f)), W - 1);
                pya = max(static_cast<int>(floorf(mny - 0.25f)), 0), pyb = min(static_cast<int>(ceilf(mxy + 0.25f)), H - 1);
                ncx = static_cast<int>(2.0f * rxm) + 1, ncy = static_cast<int>(2.0f * rym) + 1;   // integers in an interval of length 2 r: at most floor(2 r) + 1
                rxu = rxm, ryu = rym;
            }
            // (else: the inverse has a pole near the tile: every pixel is looked at, as for an invalid record)
        }
        c.pxa = __builtin_amdgcn_readfirstlane(pxa), c.pxb = __builtin_amdgcn_readfirstlane(pxb);
        c.pya = __builtin_amdgcn_readfirstlane(pya), c.pyb = __builtin_amdgcn_readfirstlane(pyb);
        c.ncx = __builtin_amdgcn_readfirstlane(ncx), c.ncy = __builtin_amdgcn_readfirstlane(ncy);
        c.rxu = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rxu)));
        c.ryu = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ryu)));
        if (c.pxa > c.pxb || c.pya > c.pyb) return false;
        c.bx0 = c.pxa, c.by0 = c.pya;
        c.cw = min(kChunkW, c.pxb - c.bx0 + 1);
        c.ch = min(kCap / c.cw, c.pyb - c.by0 + 1);
        return true;
    };
    // the first chunk at or after (plane c.k, view c.v); false: the planes of this workgroup are exhausted
    auto seek = [&](Chunk& c) -> bool {
        for (; c.k < k_last; ++c.k, c.v = 0)
            for (; c.v < p.views_per_mpi; ++c.v)
                if (open_view(c)) return true;
        return false;
    };
    auto next = [&](Chunk& c) -> bool {   // the chunk after c
        c.by0 += c.ch;
        if (c.by0 <= c.pyb) { c.ch = min(kCap / c.cw, c.pyb - c.by0 + 1); return true; }
        c.bx0 += c.cw;
        if (c.bx0 <= c.pxb) { c.cw = min(kChunkW, c.pxb - c.bx0 + 1); c.by0 = c.pya; c.ch = min(kCap / c.cw, c.pyb - c.by0 + 1); return true; }
        ++c.v;
        return seek(c);
    };
    f32x2 rp[kSlots];
    f32x4 rd[kSlots];
    auto load = [&](const Chunk& c) {   // the chunk's pixels into registers (row-major index i = tid + 512 s)
        const f32x2* __restrict__ Pv = P + (static_cast<int64_t>(c.n) * p.D + c.k) * HW;
        const f32x4* __restrict__ Gv = G + (static_cast<int64_t>(c.n) * p.D + c.k) * HW;
        const float inv_cw = 1.0f / static_cast<float>(c.cw);
        const int count = c.cw * c.ch;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int i = tid + s * kTT;
            const int ly = static_cast<int>((static_cast<float>(i) + 0.5f) * inv_cw);   // floor(i / cw): the fraction is at least 0.5 / cw away from an integer
            const int lx = i - ly * c.cw;
            const int64_t q = static_cast<int64_t>(c.by0 + ly) * W + (c.bx0 + lx);
            if (i < count) {
                rp[s] = __builtin_nontemporal_load(Pv + q);
                rd[s] = __builtin_nontemporal_load(Gv + q);
            }
        }
    };
    auto commit = [&](const Chunk& c) {   // registers -> LDS
        const int count = c.cw * c.ch;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int i = tid + s * kTT;
            if (i < count) {
                sXY[i] = make_float2(rp[s].x, rp[s].y);
                sD[i] = make_float4(rd[s].x, rd[s].y, rd[s].z, rd[s].w);
            }
        }
    };
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    auto visit = [&](int idx, bool inside, float tyf, float (&acc)[4]) {
        const float2 pos = sXY[idx];
        const float4 d = sD[idx];
        const float wxx = fmaxf(1.0f - fabsf(pos.x - xf), 0.0f), wyy = fmaxf(1.0f - fabsf(pos.y - tyf), 0.0f);
        float w = wxx * wyy;
        w = (inside && w == w) ? w : 0.0f;   // (a NaN position -- a NaN ray -- samples nothing)
        acc[0] = __builtin_fmaf(w, d.x, acc[0]), acc[1] = __builtin_fmaf(w, d.y, acc[1]);
        acc[2] = __builtin_fmaf(w, d.z, acc[2]), acc[3] = __builtin_fmaf(w, d.w, acc[3]);
    };
    auto gather = [&](const Chunk& c) {
        const int cw = c.cw, ch = c.ch;
        if (c.ncx > 0) {
            // this thread's candidate windows: [q - r, q + r] with the TILE's radius holds at most ncx integers
            float qx, qy, ax, ay;
            inv_map(c.r, xf, yf0, qx, qy, ax, ay);
            const int cxs0 = static_cast<int>(ceilf(qx - c.rxu)) - c.bx0, cys0 = static_cast<int>(ceilf(qy - c.ryu)) - c.by0;
            inv_map(c.r, xf, yf1, qx, qy, ax, ay);
            const int cxs1 = static_cast<int>(ceilf(qx - c.rxu)) - c.bx0, cys1 = static_cast<int>(ceilf(qy - c.ryu)) - c.by0;
            for (int jy = 0; jy < c.ncy; ++jy) {
                const int ly0 = cys0 + jy, ly1 = cys1 + jy;
                const bool iny0 = ly0 >= 0 && ly0 < ch, iny1 = ly1 >= 0 && ly1 < ch;
                for (int jx = 0; jx < c.ncx; ++jx) {
                    const int lx0 = cxs0 + jx, lx1 = cxs1 + jx;
                    const bool in0 = iny0 && lx0 >= 0 && lx0 < cw, in1 = iny1 && lx1 >= 0 && lx1 < cw;
                    visit(in0 ? ly0 * cw + lx0 : 0, in0, yf0, acc0);
                    visit(in1 ? ly1 * cw + lx1 : 0, in1, yf1, acc1);
                }
            }
        } else {   // no usable homography: every staged pixel is a candidate of every texel
            for (int idx = 0; idx < cw * ch; ++idx) {
                visit(idx, true, yf0, acc0);
                visit(idx, true, yf1, acc1);
            }
        }
    };
    auto store_plane = [&](int k) {   // every cell once; the sums restart
        float* __restrict__ gm = b.g_rgba + static_cast<int64_t>(m) * b.gs_mpi + static_cast<int64_t>(k) * b.gs_plane;
        if (x < Wt) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                if (y0 < Ht) {
                    float* cell = gm + c4 * b.gs_chan + static_cast<int64_t>(y0) * b.gs_row + x;
                    *cell = OVERWRITE ? acc0[c4] : *cell + acc0[c4];
                }
                if (y1 < Ht) {
                    float* cell = gm + c4 * b.gs_chan + static_cast<int64_t>(y1) * b.gs_row + x;
                    *cell = OVERWRITE ? acc1[c4] : *cell + acc1[c4];
                }
            }
        }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) acc0[c4] = acc1[c4] = 0.0f;
    };

    Chunk cur;
    cur.k = k_first, cur.v = 0;
    bool have = seek(cur);
    int k_done = k_first;   // planes [k_first, k_done) have been stored
    if (have) {
        load(cur);
        commit(cur);
    }
    while (have) {
        __syncthreads();   // cur's pixels are in LDS
        Chunk nxt = cur;
        const bool more = next(nxt);
        if (more) load(nxt);                 // in flight while cur is gathered
        for (; k_done < cur.k; ++k_done) store_plane(k_done);   // (planes nobody reaches: zeros; the plane before cur's: its sums)
        gather(cur);
        __syncthreads();   // everybody is done reading cur
        if (more) commit(nxt);
        cur = nxt;
        have = more;
    }
    for (; k_done < k_last; ++k_done) store_plane(k_done);
}

static uint64_t align256(uint64_t v) { return (v + 255) / 256 * 256; }
static uint64_t recs_bytes(const KParams& p) { return align256(static_cast<uint64_t>(p.N) * p.D * sizeof(HRec)); }

}  // namespace bwdg

// Can the atomics-free pair run this launch (workspace aside)?
bool backward_gather_supports(const KParams& p) {
    if (!(p.flags & 1u)) return false;                                   // align_corners = True only (see the head of this file)
    if (p.view_to_mpi != nullptr) return false;                          // uniform views per MPI
    if (p.N > 65535 || p.M > 65535 || p.D > 65535) return false;         // grid.y / grid.z
    if (p.Wt < 2) return false;                                          // pair loads
    if (static_cast<int64_t>(p.H) * p.W >= (int64_t(1) << 31)) return false;
    return true;
}
uint64_t backward_gather_workspace_bytes(const KParams& p) {
    if (!backward_gather_supports(p)) return 0;
    return bwdg::recs_bytes(p) + static_cast<uint64_t>(p.N) * p.D * p.H * p.W * 24u;   // sample gradients 16 B + sample positions 8 B per pixel and plane
}

template <typename TexT>
static hipError_t launch_gather_t(const KParams& p, const BwdParams& b, bool overwrite, hipStream_t stream) {
    using namespace bwdg;
    const uint64_t npp = static_cast<uint64_t>(p.N) * p.D * p.H * p.W;   // pixel-planes
    HRec* recs = static_cast<HRec*>(p.ws);
    f32x4* G = reinterpret_cast<f32x4*>(static_cast<unsigned char*>(p.ws) + recs_bytes(p));
    f32x2* P = reinterpret_cast<f32x2*>(static_cast<unsigned char*>(p.ws) + recs_bytes(p) + npp * 16u);
    const int nrec = p.N * p.D;
    hipLaunchKernelGGL((homography_kernel<true>), dim3((nrec + 63) / 64), dim3(64), 0, stream, p, recs);
    {
        const int tx = (p.W + kPTW - 1) / kPTW, ty = (p.H + kPTH - 1) / kPTH;
        const dim3 grid(xcd_grid_per_group(tx * ty, tx * ty), p.N);
        hipLaunchKernelGGL((pixel_pass_kernel<TexT, true>), grid, dim3(kPT), 0, stream, p, b, P, G, tx);
    }
    {
        const int tx = (p.Wt + kTX - 1) / kTX, ty = (p.Ht + kTY - 1) / kTY;
        // planes per workgroup: 4-8 measured best at the G-step shapes (1 ... 32: within 5 %, profiles/r06_backward.txt); fewer when the launch is small
        int ppw = std::min(p.D, 8);
        while (ppw > 1 && static_cast<int64_t>(tx) * ty * p.M * ((p.D + ppw - 1) / ppw) < 1024) ppw = (ppw + 1) / 2;
#ifdef GMPI_TUNE  // GMPI_TUNE_PPW: planes per workgroup of the texel pass
        static const int env_ppw = [] { const char* e = getenv("GMPI_TUNE_PPW"); return e ? atoi(e) : 0; }();
        if (env_ppw > 0) ppw = std::min(env_ppw, p.D);
#endif
        const dim3 grid(xcd_grid_per_group(tx * ty, tx * ty), (p.D + ppw - 1) / ppw, p.M);
        if (overwrite) hipLaunchKernelGGL((texel_gather_kernel<true>), grid, dim3(kTT), 0, stream, p, b, P, G, recs, tx, ty, ppw);
        else hipLaunchKernelGGL((texel_gather_kernel<false>), grid, dim3(kTT), 0, stream, p, b, P, G, recs, tx, ty, ppw);
    }
    return hipGetLastError();
}

// p.ws / p.ws_bytes: the caller's workspace (>= backward_gather_workspace_bytes).  overwrite: the gradient volume's content is not needed (every cell is
// WRITTEN); otherwise every cell is read, added to and written back -- by its one owner: no atomics either way.
hipError_t launch_backward_gather(const KParams& p, int dtype, const BwdParams& b, bool overwrite, hipStream_t stream) {
    switch (dtype) {
        case 0: return launch_gather_t<float>(p, b, overwrite, stream);
        case 1: return launch_gather_t<bf16_t>(p, b, overwrite, stream);
        default: return launch_gather_t<f16_t>(p, b, overwrite, stream);
    }
}

}  // namespace gmpi
