// render_backward.hip -- gradient of the fused render w.r.t. the RGBA volume (the reference's G-step back-propagates
// through MPIRenderer.render into the generator: gmpi/train.py:740-779; the sampling grid itself carries no gradient,
// gmpi/core/mpi.py:65 `with torch.no_grad()`).
//
// Forward (mpi.py:421-434):  w_k = a_k T_k,  T_{k+1} = T_k (1 - a_k + 1e-10),  C = sum_k w_k rgb_k,  Z = sum_k w_k depth_k,
// with (rgb_k, a_k) the bilinear samples of plane k.  With upstream gradients gC (3), gZ and q_k = <gC, rgb_k> + gZ depth_k:
//     dL/drgb_k = gC * w_k
//     dL/da_k   = T_k q_k  -  (sum_{j>k} w_j q_j) / (1 - a_k + 1e-10)
// and every sample gradient is scattered to its four texels with the bilinear weights (atomicAdd, fp32).
//
// One pixel per lane, two sweeps over the planes (sweep 1: Q = sum_j w_j q_j; sweep 2: prefix sums give the suffix
// sum as Q - P_k, nothing is stored per plane).  Taps come straight from global memory (same addressing as the
// gather kernel): the backward runs at training sizes (D = 32, gmpi.yml:78) where the scatter atomics, not the
// reads, dominate.  The coordinate chain is the forward's (plane_coord), so both sample the same texels.
#include "gmpi_device.hpp"

namespace gmpi {

struct BwdParams {
    const float* g_rgb;    // [N,3,H,W] gradient w.r.t. the colour the forward wrote ([0,1] or, with OUT_PM1, [-1,1])
    const float* g_depth;  // [N,1,H,W] or nullptr
    float* g_rgba;         // [M,D,4,Ht,Wt] fp32, accumulated into (caller zero-fills)
    int64_t gs_mpi, gs_plane, gs_chan, gs_row;
};

template <typename TexT, bool AC>
__global__ __launch_bounds__(256) void render_backward_kernel(const KParams p, const BwdParams b) {
    const int n = blockIdx.z;
    const int px = blockIdx.x * 64 + threadIdx.x;
    const int py = blockIdx.y * 4 + threadIdx.y;
    if (px >= p.W || py >= p.H) return;
    const int m = p.view_to_mpi ? p.view_to_mpi[n] : n / p.views_per_mpi;
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

    // ---- sweep 1: Q = sum_j w_j q_j ----------------------------------------------------------------------------
    float T = 1.0f, Q = 0.0f;
    uint32_t unused = 0;
    for (int k = 0; k < p.D; ++k) {
        float ix, iy, s, u, v;
        plane_coord<AC>(dhw[3 * k] - ez, dhw[3 * k + 1], dhw[3 * k + 2], ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
        float smp[4];
        gather_sample<TexT, false>(vol + static_cast<int64_t>(k) * p.s_plane, p.s_chan, p.s_row, Ht, Wt, ix, iy, false, unused, smp);
        const float depk = s * dot;
        const float q = gr * smp[0] + gg * smp[1] + gb * smp[2] + gz * depk;
        Q += smp[3] * T * q;
        T *= (1.0f - smp[3]) + 1e-10f;
    }

    // ---- sweep 2: gradients, scattered with the bilinear weights -----------------------------------------------
    T = 1.0f;
    float P = 0.0f;
    for (int k = 0; k < p.D; ++k) {
        float ix, iy, s, u, v;
        plane_coord<AC>(dhw[3 * k] - ez, dhw[3 * k + 1], dhw[3 * k + 2], ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
        const TexT* __restrict__ pl = vol + static_cast<int64_t>(k) * p.s_plane;
        float smp[4];
        gather_sample<TexT, false>(pl, p.s_chan, p.s_row, Ht, Wt, ix, iy, false, unused, smp);
        const float a = smp[3];
        const float depk = s * dot;
        const float q = gr * smp[0] + gg * smp[1] + gb * smp[2] + gz * depk;
        const float w = a * T;
        P += w * q;
        const float om = (1.0f - a) + 1e-10f;
        const float d_s[4] = {gr * w, gg * w, gb * w, T * q - (Q - P) / om};
        T *= om;

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

template <typename TexT>
static hipError_t launch_backward_t(const KParams& p, const BwdParams& b, hipStream_t stream) {
    const dim3 block(64, 4), grid((p.W + 63) / 64, (p.H + 3) / 4, p.N);
    if (p.flags & 1u) hipLaunchKernelGGL((render_backward_kernel<TexT, true>), grid, block, 0, stream, p, b);
    else hipLaunchKernelGGL((render_backward_kernel<TexT, false>), grid, block, 0, stream, p, b);
    return hipGetLastError();
}

hipError_t launch_backward(const KParams& p, int dtype, const float* g_rgb, const float* g_depth, float* g_rgba,
                           const int64_t* gstride, hipStream_t stream) {
    BwdParams b;
    b.g_rgb = g_rgb, b.g_depth = g_depth, b.g_rgba = g_rgba;
    b.gs_mpi = gstride[0], b.gs_plane = gstride[1], b.gs_chan = gstride[2], b.gs_row = gstride[3];
    switch (dtype) {
        case 0: return launch_backward_t<float>(p, b, stream);
        case 1: return launch_backward_t<bf16_t>(p, b, stream);
        default: return launch_backward_t<f16_t>(p, b, stream);
    }
}

}  // namespace gmpi
