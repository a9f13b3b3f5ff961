// render_dma.hip -- GMPI_VARIANT_DMA: the tile kernel with the texel boxes moved HBM -> LDS by the LDS-DMA path.
//
// Why (round 3).  The round-1/2 tile kernel (render_lds.hip) spends 98 VALU instructions per wave and plane, 40 % of them
// in the slow issue class (profiles/r02_issue_budget.txt), and drives the vector memory path with four dword loads per lane
// and plane (26.5 M wave-level loads per launch at ~16 address-unit cycles each).  Measured this round
// (tools/ubench/r3_probe.hip, profiles/r03_probe.txt):
//   * `buffer_load_dwordx4 ... lds` writes ZEROS for lanes whose offset fails the descriptor's range check and leaves
//     inactive lanes' LDS slots untouched: "zeros" padding (F.grid_sample padding_mode) costs nothing, and the loader needs
//     no staging registers, no unpack, no ds_write;
//   * `ds_read_u16_d16_hi` returns a 16-bit LDS value in the HIGH half of the destination with the low half zeroed: a bf16
//     texel arrives as the fp32 value it denotes, so the compositor has no unpack instructions (16 of the 98);
//   * v_fma / v_mul / v_add / v_sub / v_and / v_mov issue in 1.0-1.1 ns per wave and SIMD, everything else the old kernel used
//     around the taps (v_cvt_*, v_fract, v_lshl_add, v_mad_u32_u24, v_cmp, v_max) in 1.7-1.9 ns: the tap address is now two
//     fp32 FMAs on floor(ix), floor(iy) and one v_cvt_u32_f32.
//
// Shape.  One workgroup of 512 threads = one 32x16 pixel tile, one pixel per thread, all D planes, front to back (as in
// render_lds.hip; the box / table / half-tile / gather-fallback logic is the same and results are identical bit for bit in
// strict-order mode).  LDS holds the RAW texels of a plane's box, planar, [texel row][channel][x]:
//   one loader item = 16 bytes of one channel row (8 bf16 / 4 fp32 texels); a (row, channel) line is kCols items (56 texels);
//   item i of the box lives at byte 16 i  -- the lane-linear image the DMA engine writes (LDS address = M0 + 16 lane), so
//   thread t of pass r moves item 512 r + t and the map thread -> (row, channel, column) is fixed for the whole launch.
// Per plane and wave the loader is: one 16-byte broadcast read of the plane's record, 4 v_readfirstlane, and 1 (bf16 frontal)
// to 3 (fp32 tilted) DMA instructions under an exec mask that selects the lanes of the chunk's largest box.  Planes whose box
// leaves the texture (zeros padding) take the predicated form: all lanes of the box rows active, lanes outside the texture get
// the out-of-range offset.  NBUF LDS buffers, one s_barrier per plane, the DMA runs NBUF-1 planes ahead; waits are counted
// s_waitcnt vmcnt(n) written by hand (the compiler does not see the DMA).
//
// The [0,1] range check of the texels (mpi.py:185-187; GMPI_FLAG_CHECK_RANGE) used to ride on the staging registers.  Here
// every loader lane reads its own landed item back from LDS (one ds_read_b128) and reduces the 8 (4) values with 4 (2)
// three-input max instructions.
#include "gmpi_device.hpp"

#include <type_traits>

namespace gmpi {
namespace dma {

constexpr int kNT = 512;              // threads per workgroup (8 wavefronts)
constexpr int kChunk = 96;            // planes per geometry-table refill
constexpr int kRecBytes = 48;         // per-plane record: three 16-byte LDS broadcasts
constexpr float kBoxEps = 1.0f / 64;  // slack on the corner-derived box (fp32 error of ix is < 1e-3 texel)
constexpr float kCoordLimit = 16384.0f;  // |ix|, |iy| of the box corners: tap addresses are formed in fp32 (exact below 2^24 bytes)
constexpr int TW = 32, TH = kNT / TW;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename TexT, int NBUF> struct Geo {
    static constexpr int kES = static_cast<int>(sizeof(TexT));
    static constexpr int kTPI = 16 / kES;                 // texels per 16-byte item
    static constexpr int kCols = 56 / kTPI;               // items per (row, channel) line: 56 texels
    static constexpr int kLineBytes = kCols * 16;         // 112 (bf16) / 224 (fp32)
    static constexpr int kRowBytes = 4 * kLineBytes;      // one texel row = 4 channel lines
    static constexpr int kIPR = 4 * kCols;                // items per texel row
    // rows per buffer: 4 (16-bit) / 3 (fp32, two buffers) workgroups per CU must fit 160 KB of LDS
    static constexpr int kMaxRows = kES == 2 ? (NBUF == 2 ? 27 : 26) : 27;
    static constexpr int kCapItems = kMaxRows * kIPR;
    static constexpr int kBufBytes = kCapItems * 16;
    static constexpr int kNP = (kCapItems + kNT - 1) / kNT;  // DMA passes per plane at most: 2 (bf16) / 3 (fp32)
    static constexpr int kLdsBytes = kChunk * kRecBytes + NBUF * kBufBytes;
};
static_assert(Geo<bf16_t, 2>::kLdsBytes * 4 <= 160 * 1024 && Geo<bf16_t, 3>::kLdsBytes * 4 <= 160 * 1024, "4 workgroups per CU (16-bit volumes)");
static_assert(Geo<float, 2>::kLdsBytes * 3 <= 160 * 1024, "3 workgroups per CU (fp32 volumes)");

// ---- the few instructions hipcc must not see or schedule ------------------------------------------------------------
// One DMA instruction: lanes of `mask` move 16 bytes each from (descriptor base + voff) to LDS byte M0 + 16 * lane.
// (s_nop: an s_mov to M0 needs one wait state before an LDS-DMA reads it.)
template <bool NT>
__device__ __forceinline__ void dma16(uint32_t voff, const u32x4& rsrc, uint32_t lds_dst, uint64_t mask) {
    uint64_t save;
    if (NT)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b64 exec, %0"
                     : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(mask) : "memory");
    else
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b64 exec, %0"
                     : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(mask) : "memory");
}
__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform and small
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}
__device__ __forceinline__ void wg_barrier() {  // bare s_barrier: the waits around it are explicit (a __syncthreads() would drain vmcnt)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// (measured on gfx950, tools/ubench/r3_probe.hip `sem`: a d16_hi load zeroes the low half of its destination -- the sramecc form of
//  the d16 loads; tests/test_hip_kernel_paths.py::test_dma_variant_bf16_bits pins it through the kernel's results)
template <int O> __device__ __forceinline__ void tap16(uint32_t& t, uint32_t a) {
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(t) : "v"(a), "i"(O));
}
template <int O> __device__ __forceinline__ void tap32x2(uint32_t& t0, uint32_t& t1, uint32_t a) {  // texels x, x + 1 of one fp32 line
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "i"(O), "i"(O + 1));
    t0 = v.x, t1 = v.y;
}

// TexT = bf16_t or float (fp16 volumes keep render_lds.hip: a d16 load yields the half's bits, not an fp32 value).
template <typename TexT, bool AC, bool STRICT, int NBUF, bool NT, int MINW>
__global__ __launch_bounds__(kNT, MINW) void render_dma_kernel(const KParams p, const int tiles_x, const int tiles_y, const int n_tiles) {
    using G = Geo<TexT, NBUF>;
    constexpr int kES = G::kES, kTPI = G::kTPI, kCols = G::kCols, kIPR = G::kIPR, kMaxRows = G::kMaxRows, kNP = G::kNP;
    constexpr int kLineBytes = G::kLineBytes, kRowBytes = G::kRowBytes, kBufBytes = G::kBufBytes;
    constexpr int PF = NBUF - 1;  // planes the DMA runs ahead of the compositor
    constexpr bool BF = kES == 2;

    __shared__ __attribute__((aligned(16))) unsigned char smem[G::kLdsBytes];
    int4* tabL = reinterpret_cast<int4*>(smem);
    float4* tabF = reinterpret_cast<float4*>(smem + kChunk * 16);
    float2* tabG = reinterpret_cast<float2*>(smem + kChunk * 32);  // (1/hh, tap address constant), 16-byte stride
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t tile_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte*)(smem + kChunk * kRecBytes)));

    // ---- blockIdx -> tile (render_lds.hip: XCD x = blockIdx % 8 gets a contiguous run of tiles) --------------------------
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
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t bad = 0;
    const int m = view_mpi(p, n, bad);
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

    if (p.status != nullptr && trem == 0 && tid == 0) {  // mpi.py:70-72, once per view
        const float ez0 = p.eye_pos[2];
        bool behind = false;
        for (int k = 0; k < D; ++k) behind |= !(dhw[3 * k] >= ez0);
        if (behind) atomicOr(p.status, 4u);
    }

    // ---- this thread's pixel ----------------------------------------------------------------------------------------------
    const int px = txi * TW + (tid % TW), py = tyi * TH + (tid / TW);
    const bool active = px < W && py < H;
    const int64_t pix = static_cast<int64_t>(min(py, H - 1)) * W + min(px, W - 1);
    const float rx = rdv[pix], ry = rdv[HW + pix], rz = rdv[2 * HW + pix];
    float dot = rx * zx;  // einsum("nchw,nc->nhw") mpi.py:149
    dot = dot + ry * zy;
    dot = dot + rz * zz;
    const float rcp_rz = 1.0f / rz;
    Accum A;
    const int cx0 = txi * TW, cx1 = min(cx0 + TW - 1, W - 1);
    const int cy0 = tyi * TH, cy1 = min(cy0 + TH - 1, H - 1);

    // ---- this thread's loader items: item 512 r + tid -> (texel row, channel, item column), fixed for the launch --------
    uint32_t g_off[kNP];   // byte offset of the item from the box origin
    uint32_t l_pos[kNP];   // column | (4 row + channel) << 8
#pragma unroll
    for (int r = 0; r < kNP; ++r) {
        const int item = r * kNT + tid, line = item / kCols, col = item - line * kCols;
        g_off[r] = static_cast<uint32_t>((line >> 2) * s_row + (line & 3) * s_chan + kTPI * col) * static_cast<uint32_t>(kES);
        l_pos[r] = static_cast<uint32_t>(col | line << 8);
    }

    for (int kc = 0; kc < D; kc += kChunk) {
        const int kn = min(kChunk, D - kc);
        // ---- per-plane geometry (as render_lds.hip): box of the pixel rows [y_lo, y_hi] from their 4 corner pixels ------
        auto build_table = [&](int y_lo, int y_hi) -> bool {
            __syncthreads();  // the previous table and the staging buffers are no longer read
            int4 ri = make_int4(0, 0, -1, 0);  // qx0, by0, nq (< 0: does not fit), nrows  (threads < kn: plane kc + tid)
            float zdiff = 0.0f, hw = 1.0f, hh = 1.0f;
            if (tid < kn) {
                const int k = kc + tid;
                const float d = dhw[3 * k + 0], ph = dhw[3 * k + 1], pw = dhw[3 * k + 2];
                zdiff = d - ez;
                hw = pw * 0.5f, hh = ph * 0.5f;  // exact halves: (2x)/w == x/(w/2)
                float mnx = __builtin_inff(), mxx = -__builtin_inff(), mny = mnx, mxy = mxx;
                bool finite = true;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t q = static_cast<int64_t>((c & 2) ? y_hi : y_lo) * W + ((c & 1) ? cx1 : cx0);
                    float ix, iy, s, u, v;
                    plane_coord<AC>(zdiff, ph, pw, ex, ey, rdv[q], rdv[HW + q], rdv[2 * HW + q], cx, cy, ix, iy, s, u, v);
                    finite = finite && (fabsf(ix) < kCoordLimit) && (fabsf(iy) < kCoordLimit);  // false for NaN too
                    mnx = fminf(mnx, ix), mxx = fmaxf(mxx, ix), mny = fminf(mny, iy), mxy = fmaxf(mxy, iy);
                }
                if (finite) {
                    const int bx0 = static_cast<int>(floorf(mnx - kBoxEps)), bx1 = static_cast<int>(floorf(mxx + kBoxEps)) + 1;
                    const int by0 = static_cast<int>(floorf(mny - kBoxEps)), by1 = static_cast<int>(floorf(mxy + kBoxEps)) + 1;
                    ri.x = bx0 & ~(kTPI - 1);
                    ri.y = by0;
                    ri.z = (bx1 - ri.x) / kTPI + 1;
                    ri.w = by1 - by0 + 1;
                    if (ri.z > kCols || ri.w > kMaxRows) ri.z = -1;
                }

                // dims = items per line | rows << 8, sign bit set when part of the box lies outside the texture (zeros padding: the
                // loader then takes the predicated form); ext = the in-texture item columns [clo, clo + ncol) and rows [rlo, rlo + nrow)
                int dims = 0, ext = 0;
                if (ri.z > 0) {  // (qx0 and Wt are multiples of the item width)
                    const int clo = min(max(-ri.x / kTPI, 0), ri.z), chi = min(max((Wt - ri.x) / kTPI, 0), ri.z);
                    const int rlo = min(max(-ri.y, 0), ri.w), rhi = min(max(Ht - ri.y, 0), ri.w);
                    const bool inside = clo == 0 && chi == ri.z && rlo == 0 && rhi == ri.w;
                    dims = ri.z | ri.w << 8 | (inside ? 0 : static_cast<int>(0x80000000u));
                    ext = clo | (chi - clo) << 8 | rlo << 16 | (rhi - rlo) << 24;
                }
                const uint64_t origin = reinterpret_cast<uint64_t>(vol + (static_cast<int64_t>(k) * s_plane + static_cast<int64_t>(ri.y) * s_row + ri.x));
                tabL[tid] = make_int4(static_cast<int>(origin & 0xffffffffu), static_cast<int>((origin >> 32) & 0xffffu), dims, ext);
                tabF[tid] = make_float4(zdiff, hw, hh, 1.0f / hw);
                // tap byte address = buffer + (iy0 - by0) * kRowBytes + (ix0 - qx0) * kES, formed in fp32 (all terms are integers
                // below 2^24): fma(floor(iy), kRowBytes, fma(floor(ix), kES, c0))
                const int c0 = static_cast<int>(tile_base) + (tid % NBUF) * kBufBytes - (ri.y * kRowBytes + ri.x * kES);
                tabG[2 * tid] = make_float2(1.0f / hh, static_cast<float>(c0));
            }
            return __syncthreads_or(tid < kn && ri.z < 0) != 0;  // table published; does some plane's box exceed the staging buffer?
        };

        // ---- last resort (texture much finer than the image, degenerate rays): direct gather, same arithmetic ----------
        auto gather_chunk = [&](bool mine) {
            if (!mine) return;
            for (int t = 0; t < kn; ++t) {
                const float4 rf = tabF[t];
                float ix, iy, s, u, v;
                plane_coord<AC>(rf.x, rf.z + rf.z, rf.y + rf.y, ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
                float smp[4];
                gather_sample<TexT, STRICT>(vol + static_cast<int64_t>(kc + t) * s_plane, s_chan, s_row, Ht, Wt, ix, iy, check_range, bad, smp);
                blend<STRICT>(A, smp[0], smp[1], smp[2], smp[3], s, dot);
            }
        };

#ifdef GMPI_TUNE
        const bool abl_noload = (p.flags & (1u << 16)) != 0, abl_nocomp = (p.flags & (1u << 17)) != 0;
#else
        constexpr bool abl_noload = false, abl_nocomp = false;
#endif

        auto run_staged = [&](bool mine) {
            // ---- per wave: the exec masks / pass count of the box shape last seen (a tile sees a handful of shapes per chunk), and per LDS
            //      buffer those of the plane it holds (the range check of plane t uses what the loader of plane t set) ----------------
            uint64_t m_cur[kNP], m_buf[NBUF][kNP];
            int np_cur = 0, dims_cur = -1, np_buf[NBUF];
#pragma unroll
            for (int r = 0; r < kNP; ++r) m_cur[r] = 0;
            const uint32_t wave_dst = tile_base + static_cast<uint32_t>(wave) * 1024u;
            const uint32_t a_item = tile_base + static_cast<uint32_t>(tid) * 16u;

            auto issue = [&](int tn, auto ub) {  // DMA of plane tn of the chunk into buffer U = tn % NBUF
                constexpr int U = decltype(ub)::value;
                const int4 rl = tabL[tn];
                const uint32_t b_lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(rl.x));
                const uint32_t b_hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(rl.y));
                const int dims = __builtin_amdgcn_readfirstlane(rl.z);
                const u32x4 rsrc = {b_lo, b_hi, 0x80000000u, 0x00020000u};  // raw buffer, num_records 2^31: only the explicit offset below is rejected
                const uint32_t dst = wave_dst + static_cast<uint32_t>(U * kBufBytes);
                if (dims >= 0) {  // the box lies inside the texture: lanes of the box load, the others are switched off
                    if (dims != dims_cur) {
                        const int nq = dims & 0xff, rows = dims >> 8;
                        dims_cur = dims;
                        np_cur = (kIPR * rows + kNT - 1) / kNT;
#pragma unroll
                        for (int r = 0; r < kNP; ++r) {
                            m_cur[r] = __ballot(static_cast<int>(l_pos[r] & 0xffu) < nq && static_cast<int>(l_pos[r] >> 10) < rows);
                            if (abl_noload) m_cur[r] = 0;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < kNP; ++r) {
                        if (r < np_cur) dma16<NT>(g_off[r], rsrc, dst + r * (kNT * 16), m_cur[r]);
                        m_buf[U][r] = m_cur[r];
                    }
                    np_buf[U] = np_cur;
                } else {  // zeros padding: every lane of the box rows is active, lanes outside the texture get the out-of-range offset
                    const int ext = __builtin_amdgcn_readfirstlane(rl.w);
                    const int rows = (dims >> 8) & 0xff;
                    const uint32_t clo = ext & 0xff, ncol = (ext >> 8) & 0xff, llo = 4 * ((ext >> 16) & 0xff), nline = 4 * ((ext >> 24) & 0xff);
                    const int npk = (kIPR * rows + kNT - 1) / kNT;
#pragma unroll
                    for (int r = 0; r < kNP; ++r) {
                        uint64_t mk = __ballot(static_cast<int>(l_pos[r] >> 10) < rows);
                        if (abl_noload) mk = 0;
                        if (r < npk) {
                            const bool ok = ((l_pos[r] & 0xffu) - clo < ncol) & ((l_pos[r] >> 8) - llo < nline);
                            dma16<NT>(ok ? g_off[r] : 0x80000000u, rsrc, dst + r * (kNT * 16), mk);
                        }
                        m_buf[U][r] = mk;
                    }
                    np_buf[U] = npk;
                }
            };

            // ---- [0,1] test of the landed items of a plane (mpi.py:185-187): every loader lane reads its own item back.  The read is
            //      issued before the compositor's coordinate chain and evaluated behind the taps' wait (one LDS round trip for both) ----
            auto check_read = [&](auto ub, u32x4 (&cq)[kNP]) {
                constexpr int U = decltype(ub)::value;
                const uint32_t a_it = a_item;  // (a generic lambda does not capture a variable that only an asm operand names)
#pragma unroll
                for (int r = 0; r < kNP; ++r)
                    if (r < np_buf[U]) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cq[r]) : "v"(a_it), "i"(U * kBufBytes + r * kNT * 16));
            };
            auto check_eval = [&](auto ub, u32x4 (&cq)[kNP]) {
                constexpr int U = decltype(ub)::value;
#pragma unroll
                for (int r = 0; r < kNP; ++r)
                    if (r < np_buf[U]) {
                        const u32x4 q = cq[r];
                        // non-negative patterns order like unsigned integers: in [0,1] <=> pattern <= that of 1.0; the sign bit and NaN/Inf
                        // compare above; -0.0 is legal: exact re-test on the cold path.  Lanes outside the plane's box hold stale bytes.
                        uint32_t mx;
                        uint64_t viol;
                        if (BF) {
                            asm volatile("v_max3_u16 %0, %1, %1, %2 op_sel:[0,1,0,0]\n\tv_max3_u16 %0, %0, %2, %3 op_sel:[0,1,0,0]\n\t"
                                         "v_max3_u16 %0, %0, %3, %4 op_sel:[0,1,0,0]\n\tv_max3_u16 %0, %0, %4, %4 op_sel:[0,1,0,0]"
                                         : "=&v"(mx) : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
                            viol = __ballot((mx & 0xffffu) > 0x3f80u) & m_buf[U][r];
                        } else {
                            asm volatile("v_max3_u32 %0, %1, %2, %3\n\tv_max_u32 %0, %0, %4" : "=&v"(mx) : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
                            viol = __ballot(mx > 0x3f800000u) & m_buf[U][r];
                        }
                        if (__builtin_expect(viol != 0, 0)) {
                            if ((viol >> (tid & 63)) & 1) {
                                const uint32_t d[4] = {q.x, q.y, q.z, q.w};
                                if (BF) {
                                    auto ok = [](uint32_t h) { return h <= 0x3f80u || h == 0x8000u; };
#pragma unroll
                                    for (int c = 0; c < 4; ++c)
                                        if (!(ok(d[c] & 0xffffu) && ok(d[c] >> 16))) bad |= 2u;
                                } else {
                                    auto ok = [](uint32_t e) { return e <= 0x3f800000u || e == 0x80000000u; };
                                    if (!(ok(d[0]) && ok(d[1]) && ok(d[2]) && ok(d[3]))) bad |= 2u;
                                }
                            }
                        }
                    }
            };
            // (the wait statement names the check registers so that it orders their asm loads as well)
            auto wait_lds = [&](u32x4 (&cq)[kNP]) {
                if constexpr (kNP == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[0]), "+v"(cq[1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[kNP - 1]));
            };

            auto composite = [&](int t, auto ub, u32x4 (&cq)[kNP]) {
                const float4 rf = tabF[t];
                const float2 rg = tabG[2 * t];
                float ix, iy, s;
                float nw, ne, sw, se;
                float fx, fy;
                if (STRICT) {
                    float u, v;
                    plane_coord<AC>(rf.x, rf.z + rf.z, rf.y + rf.y, ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
                    fx = floorf(ix), fy = floorf(iy);
                } else {
                    plane_coord_recip<AC>(rf.x, rf.y, rf.z, rf.w, rg.x, ex, ey, rx, ry, rz, rcp_rz, cx, cy, ix, iy, s);
                    fx = floorf(ix), fy = floorf(iy);
                }
                // LDS byte address of the north-west tap of channel 0: two exact fp32 FMAs and one saturating conversion (NaN -> 0;
                // an address past the allocation reads zeros): the box contains every tap of the tile (corner argument)
                const float af = __builtin_fmaf(fy, static_cast<float>(kRowBytes), __builtin_fmaf(fx, static_cast<float>(kES), rg.y));
                uint32_t a_tap = static_cast<uint32_t>(af);
                uint32_t q[16];
                if constexpr (BF) {
                    // a bf16 texel read into the high half of a register (low half zeroed by the load) IS its fp32 value
                    tap16<0 * kLineBytes>(q[0], a_tap), tap16<0 * kLineBytes + 2>(q[1], a_tap), tap16<0 * kLineBytes + kRowBytes>(q[2], a_tap), tap16<0 * kLineBytes + kRowBytes + 2>(q[3], a_tap);
                    tap16<1 * kLineBytes>(q[4], a_tap), tap16<1 * kLineBytes + 2>(q[5], a_tap), tap16<1 * kLineBytes + kRowBytes>(q[6], a_tap), tap16<1 * kLineBytes + kRowBytes + 2>(q[7], a_tap);
                    tap16<2 * kLineBytes>(q[8], a_tap), tap16<2 * kLineBytes + 2>(q[9], a_tap), tap16<2 * kLineBytes + kRowBytes>(q[10], a_tap), tap16<2 * kLineBytes + kRowBytes + 2>(q[11], a_tap);
                    tap16<3 * kLineBytes>(q[12], a_tap), tap16<3 * kLineBytes + 2>(q[13], a_tap), tap16<3 * kLineBytes + kRowBytes>(q[14], a_tap), tap16<3 * kLineBytes + kRowBytes + 2>(q[15], a_tap);
                } else {
                    const uint32_t a_bot = a_tap + kRowBytes;
                    tap32x2<0 * (kLineBytes / 4)>(q[0], q[1], a_tap), tap32x2<0 * (kLineBytes / 4)>(q[2], q[3], a_bot);
                    tap32x2<1 * (kLineBytes / 4)>(q[4], q[5], a_tap), tap32x2<1 * (kLineBytes / 4)>(q[6], q[7], a_bot);
                    tap32x2<2 * (kLineBytes / 4)>(q[8], q[9], a_tap), tap32x2<2 * (kLineBytes / 4)>(q[10], q[11], a_bot);
                    tap32x2<3 * (kLineBytes / 4)>(q[12], q[13], a_tap), tap32x2<3 * (kLineBytes / 4)>(q[14], q[15], a_bot);
                }
                // the bilinear weights while the taps are in flight
                if (STRICT) {
                    const float fx1 = fx + 1.0f, fy1 = fy + 1.0f;
                    const float wx1 = ix - fx, wx0 = fx1 - ix, wy1 = iy - fy, wy0 = fy1 - iy;
                    nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
                } else {
                    // ATen's vectorised CPU form of the weights: w1 = ix - floor(ix), w0 = 1 - w1
                    const float wx1 = ix - fx, wy1 = iy - fy;
                    const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
                    nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]),
                               "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]), "+v"(nw), "+v"(ne), "+v"(sw), "+v"(se));
                if (check_range) {
                    wait_lds(cq);
                    check_eval(ub, cq);
                }
                Footprint f;
                f.nw = nw, f.ne = ne, f.sw = sw, f.se = se;
                float smp[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    smp[c] = bilerp<STRICT>(__uint_as_float(q[4 * c]), __uint_as_float(q[4 * c + 1]), __uint_as_float(q[4 * c + 2]), __uint_as_float(q[4 * c + 3]), f);
                blend<STRICT>(A, smp[0], smp[1], smp[2], smp[3], s, dot);
            };

            // ---- the plane loop: one barrier per plane; the DMA of plane t + PF is issued right behind the barrier that
            //      retires the last readers of its buffer -----------------------------------------------------------------
#pragma unroll
            for (int u = 0; u < NBUF; ++u) {
                np_buf[u] = 0;
#pragma unroll
                for (int r = 0; r < kNP; ++r) m_buf[u][r] = 0;
            }
            auto stage = [&](int tt, auto ub) {  // plane tt of the chunk, held by buffer U
                constexpr int U = decltype(ub)::value;
                __builtin_amdgcn_s_setprio(2);
                int newer = 0;  // DMA instructions of the planes behind plane tt that may stay in flight
#pragma unroll
                for (int j = 1; j < PF; ++j)
                    if (tt + j < kn) newer += np_buf[(U + j) % NBUF];
                wait_vmcnt(newer);  // own DMA of plane tt has landed
                wg_barrier();       // everybody's has; everybody is done reading plane tt - 1
                if (tt + PF < kn) issue(tt + PF, std::integral_constant<int, (U + PF) % NBUF>{});
                __builtin_amdgcn_s_setprio(0);
                u32x4 cq[kNP];
                if (check_range) check_read(ub, cq);
                if (mine && !abl_nocomp) composite(tt, ub, cq);
                else if (check_range) {
                    wait_lds(cq);
                    check_eval(ub, cq);
                }
            };
            if (PF >= 1 && 0 < kn) issue(0, std::integral_constant<int, 0>{});
            if (PF >= 2 && 1 < kn) issue(1, std::integral_constant<int, 1 % NBUF>{});
            for (int t = 0; t < kn; t += NBUF) {
                stage(t, std::integral_constant<int, 0>{});
                if (t + 1 < kn) stage(t + 1, std::integral_constant<int, 1>{});
                if (NBUF > 2 && t + 2 < kn) stage(t + 2, std::integral_constant<int, 2 % NBUF>{});
            }
        };

        // ---- whole tile if every box fits; else its two 32x8 halves one after the other; else the direct gather --------
        const int half = (tid / TW) / (TH / 2);  // wave-uniform: waves 0-3 upper half, 4-7 lower half
#pragma unroll 1
        for (int h = -1; h < 2; ++h) {
            const int y_lo = h < 0 ? cy0 : cy0 + h * (TH / 2);
            if (y_lo > cy1) break;
            const int y_hi = h < 0 ? cy1 : min(y_lo + TH / 2 - 1, cy1);
            const bool unfit = build_table(y_lo, y_hi);
            if (h < 0 && unfit) continue;
            const bool mine = h < 0 || half == h;
            if (!unfit) run_staged(mine);
            else gather_chunk(mine);
            if (h < 0) break;
        }
    }

    // ---- assert_not_out_of_last_plane (mpi.py:381-395): u,v of the last plane, once per pixel --------------------------
    if (check_last) {
        const float d = dhw[3 * (D - 1) + 0], ph = dhw[3 * (D - 1) + 1], pw = dhw[3 * (D - 1) + 2];
        float ix, iy, s, u, v;
        plane_coord<AC>(d - ez, ph, pw, ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
        if (!(u >= -1.0f && u <= 1.0f && v >= -1.0f && v <= 1.0f)) bad |= 1u;
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

template <typename TexT, int NBUF, bool NT, int MINW>
static hipError_t launch_t(const KParams& p, hipStream_t stream) {
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int n_tiles = tiles_x * tiles_y * p.N;
    const dim3 grid(((n_tiles + 7) / 8) * 8), block(kNT);
    const bool ac = p.flags & 1u, strict = p.flags & (1u << 4);
    if (ac && strict) hipLaunchKernelGGL((render_dma_kernel<TexT, true, true, NBUF, NT, MINW>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else if (ac) hipLaunchKernelGGL((render_dma_kernel<TexT, true, false, NBUF, NT, MINW>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else if (strict) hipLaunchKernelGGL((render_dma_kernel<TexT, false, true, NBUF, NT, MINW>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else hipLaunchKernelGGL((render_dma_kernel<TexT, false, false, NBUF, NT, MINW>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    return hipGetLastError();
}

}  // namespace dma

bool dma_variant_supports(const KParams& p, int dtype) {
    // bf16 volumes only (fp16: render_lds.hip; the fp32 instance of this kernel does not pass its parity check yet and is not dispatched)
    if (dtype != 1) return false;
    const int es = 2, tpi = 16 / es;
    if (p.Wt % tpi != 0) return false;
    if (reinterpret_cast<uintptr_t>(p.rgba) % 16 != 0) return false;
    if (p.s_row % tpi != 0 || p.s_chan % tpi != 0 || p.s_plane % tpi != 0 || p.s_mpi % tpi != 0) return false;
    if (p.Ht > 8192 || p.Wt > 8192) return false;  // tap addresses are formed in fp32
    const int64_t span = 3 * p.s_chan + 28 * p.s_row + 128;  // the in-plane item offset is kept in 32 bits
    if (span >= (int64_t(1) << 31) / es) return false;
    return true;
}

hipError_t launch_dma(const KParams& p0, int dtype, int tune, hipStream_t stream) {
    KParams p = p0;
#ifdef GMPI_TUNE  // profiling builds: tune bits 8-9 = ablations (no memory traffic / no compositing), bits 0-3 select the instance
    p.flags |= static_cast<uint32_t>((tune >> 8) & 3) << 16;
    const int inst = tune & 15;
    if (dtype == 1) {
        switch (inst) {
            case 1: return dma::launch_t<bf16_t, 2, true, 8>(p, stream);
            case 2: return dma::launch_t<bf16_t, 3, false, 8>(p, stream);
            case 3: return dma::launch_t<bf16_t, 3, true, 8>(p, stream);
            default: return dma::launch_t<bf16_t, 2, false, 8>(p, stream);
        }
    }
    switch (inst) {
        case 1: return dma::launch_t<float, 2, true, 6>(p, stream);
        case 2: return dma::launch_t<float, 3, false, 4>(p, stream);
        case 3: return dma::launch_t<float, 3, true, 4>(p, stream);
        default: return dma::launch_t<float, 2, false, 6>(p, stream);
    }
#else
    (void)tune;
    if (dtype == 1) return dma::launch_t<bf16_t, 2, false, 8>(p, stream);
    return dma::launch_t<float, 2, false, 6>(p, stream);
#endif
}

}  // namespace gmpi
