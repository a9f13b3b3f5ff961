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
#include "gmpi_device.hpp"

#include <cstdlib>

namespace gmpi {

struct BwdParams {
    const float* g_rgb;    // [N,3,H,W] gradient w.r.t. the colour the forward wrote ([0,1] or, with OUT_PM1, [-1,1])
    const float* g_depth;  // [N,1,H,W] or nullptr
    float* g_rgba;         // [M,D,4,Ht,Wt] fp32, accumulated into (caller zero-fills)
    int64_t gs_mpi, gs_plane, gs_chan, gs_row;
};

// transmittance as mantissa (in [0.5,1)) x 2^exponent
struct XT {
    float m;
    int e;
    __device__ __forceinline__ void renorm() {
        e += __builtin_amdgcn_frexp_expf(m);
        m = __builtin_amdgcn_frexp_mantf(m);
    }
    __device__ __forceinline__ float value() const { return __builtin_amdgcn_ldexpf(m, e); }
};

// Final transmittance of one pixel: the forward's value when it is usable, else a front-to-back walk of the alpha
// channel in the extended representation.
template <typename TexT, bool AC>
__device__ __forceinline__ XT total_transmittance(const KParams& p, const float* __restrict__ dhw, const TexT* __restrict__ vol,
                                                  float t_fwd, bool have_fwd, float ex, float ey, float ez, float rx, float ry,
                                                  float rz, float cx, float cy) {
    XT t{1.0f, 0};
    if (have_fwd && t_fwd >= 1e-30f) {
        t.m = t_fwd;
        t.renorm();
        return t;
    }
    uint32_t unused = 0;
    for (int k = 0; k < p.D; ++k) {
        float ix, iy, s, u, v;
        plane_coord<AC>(dhw[3 * k] - ez, dhw[3 * k + 1], dhw[3 * k + 2], ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
        float smp[4];
        gather_sample<TexT, false>(vol + static_cast<int64_t>(k) * p.s_plane, p.s_chan, p.s_row, p.Ht, p.Wt, ix, iy, false, unused, smp);
        t.m *= (1.0f - smp[3]) + 1e-10f;
        t.renorm();
    }
    return t;
}

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

template <typename TexT>
static hipError_t launch_backward_t(const KParams& p, const BwdParams& b, bool tiles, hipStream_t stream) {
    const bool ac = p.flags & 1u;
    if (tiles) {
        const int tiles_x = (p.W + kBwdTW - 1) / kBwdTW, tiles_y = (p.H + kBwdTH - 1) / kBwdTH;
        const dim3 grid(tiles_x * tiles_y, p.N), block(kBwdThreads);
        if (ac) hipLaunchKernelGGL((render_backward_tile_kernel<TexT, true>), grid, block, 0, stream, p, b, tiles_x);
        else hipLaunchKernelGGL((render_backward_tile_kernel<TexT, false>), grid, block, 0, stream, p, b, tiles_x);
        return hipGetLastError();
    }
    const dim3 block(64, 4), grid((p.W + 63) / 64, (p.H + 3) / 4, p.N);
    if (ac) hipLaunchKernelGGL((render_backward_kernel<TexT, true>), grid, block, 0, stream, p, b);
    else hipLaunchKernelGGL((render_backward_kernel<TexT, false>), grid, block, 0, stream, p, b);
    return hipGetLastError();
}

// `tiles`: stage the scatter per pixel tile in LDS (default); false = one pixel per lane, 16 global atomics each
// (GMPI_VARIANT_GATHER: the simple kernel, kept as the cross-check)
hipError_t launch_backward(const KParams& p0, int dtype, const float* g_rgb, const float* g_depth, float* g_rgba,
                           const int64_t* gstride, bool tiles, hipStream_t stream) {
    KParams p = p0;
#ifdef GMPI_TUNE  // profiling builds only: 16 = no global atomics in the flush, 32 = no LDS atomics
    static const unsigned skip = [] { const char* e = getenv("GMPI_TUNE_SKIP"); return e ? static_cast<unsigned>(atoi(e)) : 0u; }();
    p.flags |= skip << 16;
#endif
    BwdParams b;
    b.g_rgb = g_rgb, b.g_depth = g_depth, b.g_rgba = g_rgba;
    b.gs_mpi = gstride[0], b.gs_plane = gstride[1], b.gs_chan = gstride[2], b.gs_row = gstride[3];
    if (p.N > 65535) tiles = false;  // grid.y
    switch (dtype) {
        case 0: return launch_backward_t<float>(p, b, tiles, stream);
        case 1: return launch_backward_t<bf16_t>(p, b, tiles, stream);
        default: return launch_backward_t<f16_t>(p, b, tiles, stream);
    }
}

}  // namespace gmpi
