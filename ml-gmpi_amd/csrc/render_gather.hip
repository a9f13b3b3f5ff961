// render_gather.hip -- GMPI_VARIANT_GATHER: one output pixel per lane, taps straight from global memory.
//
// This is the shape-agnostic kernel (any Ht/Wt/H/W, any rgba strides, any storage dtype): a
// wavefront covers 64 consecutive pixels of one image row, walks the D planes front to back
// (mpi.py:413 "the 1st plane is the closest one") and keeps colour/depth/transmittance in
// registers; nothing of the reference's [N*D, ...] temporaries (mpi.py:362-417) is materialised.
// Because the warp is near-identity (0.85-1.0 texel per pixel), the 64 lanes of a tap load touch
// 2-3 consecutive 128-byte lines -> coalesced through the vector L1; the 4x tap reuse is served by
// L1/L2, HBM sees each texel once per view.
#include "gmpi_device.hpp"

namespace gmpi {

constexpr int kGatherTileW = 64;  // one wavefront = 64 consecutive pixels of a row
constexpr int kGatherTileH = 4;   // 4 wavefronts per workgroup

template <typename TexT, bool AC, bool STRICT>
__global__ __launch_bounds__(kGatherTileW* kGatherTileH) void render_gather_kernel(const KParams p) {
    const int n = blockIdx.z;
    const int px = blockIdx.x * kGatherTileW + threadIdx.x;
    const int py = blockIdx.y * kGatherTileH + threadIdx.y;
    uint32_t bad = 0;
    const int m = view_mpi(p, n, bad);  // (an index outside [0, M) is clamped and reported)

    const float* __restrict__ dhw = p.dhw + static_cast<int64_t>(m) * p.D * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];

    // mpi.py:70-72 compares every plane distance with eye_z of the FIRST view; one lane per view checks it.
    if (p.status != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && threadIdx.y == 0) {
        const float ez0 = p.eye_pos[2];
        bool behind = false;
        for (int k = 0; k < p.D; ++k) behind |= !(dhw[3 * k] >= ez0);
        if (behind) atomicOr(p.status, 4u);
    }
    // edge tiles: out-of-image lanes shadow the last pixel (keeps the wave converged for the status reduce)
    const bool active = px < p.W && py < p.H;
    const int64_t HW = static_cast<int64_t>(p.H) * p.W;
    const int64_t pix = static_cast<int64_t>(min(py, p.H - 1)) * p.W + min(px, p.W - 1);
    const float* __restrict__ rd = p.ray_dir + static_cast<int64_t>(n) * 3 * HW + pix;
    const float rx = rd[0], ry = rd[HW], rz = rd[2 * HW];
    const float zx = p.z_dir[3 * n + 0], zy = p.z_dir[3 * n + 1], zz = p.z_dir[3 * n + 2];
    float dot = rx * zx;  // einsum("nchw,nc->nhw") mpi.py:149
    dot = dot + ry * zy;
    dot = dot + rz * zz;

    const int Ht = p.Ht, Wt = p.Wt;
    const float cx = AC ? static_cast<float>(Wt - 1) * 0.5f : static_cast<float>(Wt);
    const float cy = AC ? static_cast<float>(Ht - 1) * 0.5f : static_cast<float>(Ht);
    const bool check_range = (p.flags & (1u << 3)) != 0;
    const bool check_last = (p.flags & (1u << 2)) != 0;

    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    const int64_t s_chan = p.s_chan, s_row = p.s_row;

    Accum A;
#pragma unroll 2
    for (int k = 0; k < p.D; ++k) {
        const float d = dhw[3 * k + 0], ph = dhw[3 * k + 1], pw = dhw[3 * k + 2];
        const float zdiff = d - ez;
        float ix, iy, s, u, v;
        plane_coord<AC>(zdiff, ph, pw, ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
        if (check_last && k == p.D - 1 && !(u >= -1.0f && u <= 1.0f && v >= -1.0f && v <= 1.0f)) bad |= 1u;

        float smp[4];
        gather_sample<TexT, STRICT>(vol + static_cast<int64_t>(k) * p.s_plane, s_chan, s_row, Ht, Wt, ix, iy, check_range, bad, smp);
        blend<STRICT>(A, smp[0], smp[1], smp[2], smp[3], s, dot);
    }

    float r = A.r, g = A.g, b = A.b;
    if (p.flags & (1u << 1)) {  // mpi_renderer.py:467  2*c - 1
        r = 2.0f * r - 1.0f;
        g = 2.0f * g - 1.0f;
        b = 2.0f * b - 1.0f;
    }
    if (active) {
        float* __restrict__ out = p.rgb_out + static_cast<int64_t>(n) * 3 * HW + pix;
        out[0] = r;
        out[HW] = g;
        out[2 * HW] = b;
        p.depth_out[static_cast<int64_t>(n) * HW + pix] = finish_depth<STRICT>(A, dot);
        if (p.T_out) p.T_out[static_cast<int64_t>(n) * HW + pix] = A.T;
    }
    report_status(p.status, bad);
}

template <typename TexT>
static hipError_t launch_gather_t(const KParams& p, hipStream_t stream) {
    const dim3 block(kGatherTileW, kGatherTileH);
    const dim3 grid((p.W + kGatherTileW - 1) / kGatherTileW, (p.H + kGatherTileH - 1) / kGatherTileH, p.N);
    const bool ac = p.flags & 1u, strict = p.flags & (1u << 4);
    if (ac && strict) hipLaunchKernelGGL((render_gather_kernel<TexT, true, true>), grid, block, 0, stream, p);
    else if (ac) hipLaunchKernelGGL((render_gather_kernel<TexT, true, false>), grid, block, 0, stream, p);
    else if (strict) hipLaunchKernelGGL((render_gather_kernel<TexT, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((render_gather_kernel<TexT, false, false>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_gather(const KParams& p, int dtype, hipStream_t stream) {
    switch (dtype) {
        case 0: return launch_gather_t<float>(p, stream);
        case 1: return launch_gather_t<bf16_t>(p, stream);
        default: return launch_gather_t<f16_t>(p, stream);
    }
}

}  // namespace gmpi
