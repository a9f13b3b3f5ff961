// gmpi_backward.hpp -- what the backward kernels of the render share (render_backward.hip: the tile kernels with their atomics;
// render_backward_gather.hip: the atomics-free pair of round 6).
#pragma once
#include "gmpi_device.hpp"

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

template <typename TexT> __device__ __forceinline__ void load_pair(const unsigned char* __restrict__ base, uint32_t byte_off, float& a, float& b) {
    const TexT* __restrict__ q = reinterpret_cast<const TexT*>(base + byte_off);
    a = to_f32(q[0]), b = to_f32(q[1]);
}
template <> __device__ __forceinline__ void load_pair<float>(const unsigned char* __restrict__ base, uint32_t byte_off, float& a, float& b) {
    float v[2];
    __builtin_memcpy(v, base + byte_off, 8);  // (one global_load_dwordx2 at dword alignment, uniform base + 32-bit lane offset)
    a = v[0], b = v[1];
}

}  // namespace gmpi
