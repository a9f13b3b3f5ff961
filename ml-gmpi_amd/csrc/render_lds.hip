// render_lds.hip -- GMPI_VARIANT_LDS: pixel tiles, per-plane texel boxes staged through LDS.
//
// Why: the direct gather issues 16 dword loads per pixel*plane through the vector L1 and stalls at
// ~13-25 % of the HBM roofline (profiles/r01_gather_v0_*).  Here a workgroup of 512 threads owns a
// TW x TH pixel tile (one pixel per thread) and walks the D planes front to back.  The warp is a
// homography of a small rectangle, so the texels a tile needs on one plane form a small box whose
// extremes are at the tile's four corner pixels.  The box is copied from HBM with 16-byte row loads
// (each lane 4 fp32 / 8 half-precision texels of one channel row), kept as fp32 in LDS in
// [row][channel][x] order, and every pixel takes its 16 taps with 8 ds_read2_b32 (x0,x1 pairs).
// Texels outside the texture are stored as zeros, so the consumer needs no masks ("zeros" padding of
// F.grid_sample).  Two LDS buffers + register staging give a one-barrier-per-plane pipeline: the loads
// of plane k+1 are in flight while plane k is composited.  (Deeper staging is implemented -- PF template
// parameter, GMPI_TUNE_PF -- but does not pay: the kernel is bound by its on-chip work, not by load
// latency, and two planes in flight cost 25 % on the D = 256 workload: profiles/r01_ablation.txt.)
//
// Box size.  Tilted cameras shear and stretch the footprint: a 32x16 pixel tile needs 33x17 texels
// for a frontal view, 37x20 at (yaw 0.3, pitch 0.1), 47x29 at the 2-sigma FFHQ pose.  Per chunk of
// planes the tile is staged whole if every box fits the buffer, else as its two 32x8 halves, else
// (texture much finer than the image, degenerate rays) the chunk takes the direct gather -- same
// arithmetic, so results do not depend on the path.  64x8 tiles (TileCfg<64>, GMPI_TUNE_TW=64) read
// longer lines but are no faster at any pose and far slower for tilted ones, so 32x16 is the default.
//
// Instruction count is what bounds the kernel (HBM traffic == algorithmic bytes; with the memory loads
// disabled it runs only 7 % (16-bit volumes) / 30 % (fp32) faster), so:
//  * the three IEEE divisions of the coordinate chain are mul+fma+fma against correctly rounded
//    reciprocals hoisted out of the plane loop (div_by_recip() in gmpi_device.hpp);
//  * everything that is uniform per plane -- box origin address, in-texture ranges, plane constants -- is
//    computed once per plane by one thread into an LDS table instead of 8x on the waves' scalar units;
//  * the loader's thread -> item map is re-derived per chunk from the largest box of the chunk, so a frontal
//    view needs 1 pass (16-bit) / 2 passes (fp32) over the box instead of the worst-case 2 / 3;
//  * tap addresses are kept opaque so that the 8 ds_read2_b32 use immediate offsets.
//  * 16-bit volumes keep their raw texels in LDS, interleaved per texel (LAYOUT 1 below): half the LDS bytes, two
//    ds_read2_b64 per pixel, and no bank wrap when a tilted view samples more than one texel per pixel along x (with
//    fp32 planes in LDS that wrap doubles the tap cost: tools/ubench/lds_tap_pattern.hip);
//  * the library is built with -fno-slp-vectorize: packed fp32 math (v_pk_*_f32) issues at 4.9 cycles per wave on this
//    part against 2.8 for scalar fp32, and the packing costs extra moves.
//
// HBM traffic: each texel of the volume is read once per view (halo rows/columns are shared with the
// neighbouring tiles through the XCD's L2: the blockIdx -> tile map gives every XCD a contiguous run
// of tiles).  Algorithmic bytes: 16 B (fp32) / 8 B (bf16) per pixel*plane + 28-32 B per pixel.
#include "gmpi_device.hpp"

#include <cstdlib>

#include <algorithm>

#include <type_traits>

namespace gmpi {

constexpr int kNT = 512;              // threads per workgroup (8 wavefronts)
constexpr int kChunk = 96;            // planes per geometry-table refill
constexpr int kRecBytes = 48;         // per-plane record: three 16-byte LDS broadcasts
constexpr int kMaxStagedItems = 4;    // 16-byte items (4 VGPRs each) a thread may hold in flight: 80 VGPRs at 6 waves/SIMD
constexpr float kBoxEps = 1.0f / 64;  // slack on the corner-derived box (fp32 error of ix is < 1e-3 texel)

// Staging-buffer geometry per tile shape.  kPitch = floats per (row,channel) line in LDS (a multiple of 8: the texel-row
// stride 4*kPitch floats is then a multiple of the 32 LDS banks, so taps of lanes on neighbouring texel rows do not
// conflict); kMaxLines = (row,channel) lines per staging buffer.  Two buffers + the table must stay <= 53 KB for
// 3 workgroups per CU.
//   32x16 pixels: boxes 33x17 texels frontal ... 47x29 at the 2-sigma FFHQ pose      -> 56 floats x 27 rows
//   64x8  pixels: boxes 65x9 frontal (longer lines: fewer partially used 128-byte
//                 cache lines per texel), taller and wider when the camera tilts       -> 96 floats x 14 rows
template <int TW> struct TileCfg;
template <> struct TileCfg<32> { static constexpr int kPitch = 56, kMaxLines = 108; };
template <> struct TileCfg<64> { static constexpr int kPitch = 96, kMaxLines = 56; };
template <int TW> constexpr int lds_bytes() { return kChunk * kRecBytes + 2 * TileCfg<TW>::kMaxLines * TileCfg<TW>::kPitch * 4; }
static_assert(lds_bytes<32>() <= 53 * 1024, "3 workgroups per CU");

// Loader geometry: one item = 16 bytes of storage = TPI texels; a box line holds kPitch/TPI items.
// (LPR = lines per texel row: 4 (row,channel) lines in the planar layout, 1 in the texel-interleaved one)
template <int TPI, int TW, int LPR> struct LoaderCfg {
    static constexpr int kCols = TileCfg<TW>::kPitch / TPI;        // 32-wide tiles: 14 (fp32) / 7 (16-bit) items per line
    static constexpr int kLinesPerPass = kNT / kCols;            // 36 / 73 lines per pass
    static constexpr int kNL = (TileCfg<TW>::kMaxLines / 4 * LPR + kLinesPerPass - 1) / kLinesPerPass;  // 3 / 2 passes at most
};

// Per-plane record of the current chunk (LDS), written once per plane by one thread so that the 8 waves do not repeat
// the address arithmetic on their scalar units:
//   tabL (loader):     byte address of the box origin (texel row by0, column qx0 of channel 0; qx0 a multiple of the item
//                      width) as two dwords = words 0,1 of the plane's buffer descriptor; cols = nq | clo << 8 | ncol << 16;
//                      lines = 4 nrows | llo << 8 | nline << 16.  nq items x nrows rows is the box; item columns
//                      [clo, clo + ncol) and (row,channel) lines [llo, llo + nline) of it lie inside the texture
//   tabF (compositor): zdiff = d - eye_z, hw = w/2, hh = h/2, RN(1/hw)
//   tabG (compositor): RN(1/hh), qx0, by0

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// storage -> fp32: one loader item is ALWAYS 16 bytes of storage (4 fp32 texels or 8 half-precision texels:
// the load path is request-bound, so 16-bit volumes move twice the texels per request) -----------------------
template <typename TexT> struct Quad;
template <> struct Quad<float> {
    static constexpr int kTexels = 4;
    static constexpr uint32_t kOneBits = 0;
    static __device__ __forceinline__ void unpack2(uint32_t, float&, float&) {}
    static __device__ __forceinline__ void cvt(const u32x4& v, float4 (&o)[1]) {
        o[0] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
};
template <> struct Quad<bf16_t> {
    static constexpr int kTexels = 8;
    static constexpr uint32_t kOneBits = 0x3f80u;  // 1.0 as a 16-bit pattern
    static __device__ __forceinline__ void unpack2(uint32_t v, float& lo, float& hi) {
        lo = __uint_as_float(v << 16), hi = __uint_as_float(v & 0xffff0000u);
    }
    static __device__ __forceinline__ void cvt(const u32x4& v, float4 (&o)[2]) {
        o[0] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
        o[1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16),
                           __uint_as_float(v.w & 0xffff0000u));
    }
};
template <> struct Quad<f16_t> {
    static constexpr int kTexels = 8;
    static constexpr uint32_t kOneBits = 0x3c00u;
    static __device__ __forceinline__ void unpack2(uint32_t v, float& lo, float& hi) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = __builtin_bit_cast(h2, v);
        lo = static_cast<float>(a.x), hi = static_cast<float>(a.y);
    }
    static __device__ __forceinline__ void cvt(const u32x4& v, float4 (&o)[2]) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const uint32_t vx = v.x, vy = v.y, vz = v.z, vw = v.w;  // (bit_cast of an ext-vector element lvalue reads element 0)
        const h2 a = __builtin_bit_cast(h2, vx), b = __builtin_bit_cast(h2, vy), c = __builtin_bit_cast(h2, vz), d = __builtin_bit_cast(h2, vw);
        o[0] = make_float4(static_cast<float>(a.x), static_cast<float>(a.y), static_cast<float>(b.x), static_cast<float>(b.y));
        o[1] = make_float4(static_cast<float>(c.x), static_cast<float>(c.y), static_cast<float>(d.x), static_cast<float>(d.y));
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

// TW x TH pixel tile, one pixel per thread: lane = (x = tid % TW, y = tid / TW).
// LAYOUT 0: LDS tile = fp32 planes [row][channel][x], one loader item = 16 bytes of one channel row.
// LAYOUT 1: LDS tile = the raw texels, interleaved [row][x][RGBA] (8 bytes per 16-bit texel, 16 per fp32 texel); one
//   loader item = 16 bytes of LDS = the four channels of a texel pair (16-bit: 4 dword loads, 4 v_perm) or of one texel
//   (fp32: 4 dword loads), one 16-byte store.  A pixel's 16 taps are two ds_read2_b64 (16-bit: 32 bytes instead of 64,
//   unpacked to fp32 in registers) or four ds_read_b128 (fp32: 8 lanes per LDS pass, so the bank wrap of a stretched
//   row only bites when a texel is skipped inside 8 pixels).
// The body of one workgroup's tile, as a function of the (virtual) block index: the kernels below call it once (render_lds_kernel) or in a
// loop over a small grid (render_lds_gated_kernel).
template <typename TexT, bool AC, bool STRICT, int TW, int MINW, int PF, int LAYOUT>
__device__ __forceinline__ void render_lds_tile(const KParams& p, const int tiles_x, const int tiles_y, const int n_tiles, const unsigned vblock,
                                                const uint32_t* view_bits = nullptr) {  // view_bits: the gate, one bit per view, already in LDS
    using Q = Quad<TexT>;
    constexpr int LPR = LAYOUT == 1 ? 1 : 4;
    constexpr int kTexelBytes = 4 * static_cast<int>(sizeof(TexT));  // interleaved layout: RGBA of one texel
    using LC = LoaderCfg<(LAYOUT == 1 ? 16 / kTexelBytes : Q::kTexels), TW, LPR>;
    constexpr int kPitch = TileCfg<TW>::kPitch, kMaxLines = TileCfg<TW>::kMaxLines, kMaxRows = kMaxLines / 4;
    // floats per staging buffer: fp32 planes / fp32 texels need 4 per texel-channel, raw 16-bit texels half of that
    constexpr int kCapFloats = (LAYOUT == 1 && sizeof(TexT) == 2) ? kMaxLines * kPitch / 2 : kMaxLines * kPitch;
    constexpr int kLdsBytes = kChunk * kRecBytes + 2 * kCapFloats * 4;
    constexpr int TPI = LAYOUT == 1 ? 16 / kTexelBytes : Q::kTexels, kCols = LC::kCols, kRowcPerPass = LC::kLinesPerPass, kNL = LC::kNL;
    constexpr int TH = kNT / TW;

    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];
    __shared__ int box_max[2];  // largest item count per row / row count of the chunk's boxes (ds_max_i32 reduction)
    int4* tabL = reinterpret_cast<int4*>(smem);
    float4* tabF = reinterpret_cast<float4*>(smem + kChunk * 16);
    int4* tabG = reinterpret_cast<int4*>(smem + kChunk * 32);
    float* tile0 = reinterpret_cast<float*>(smem + kChunk * kRecBytes);

    // ---- blockIdx -> tile: XCD x (blockIdx % 8) gets the contiguous run [x*per, (x+1)*per) of tiles, so
    //      neighbouring tiles (shared halo texels) meet in one L2 ------------------------------------------
    const int tiles_per_view = tiles_x * tiles_y;
    const int per_xcd = (n_tiles + 7) / 8;
    int tile_id = (vblock % 8) * per_xcd + vblock / 8;
    if (vblock >= static_cast<unsigned>(per_xcd * 8)) tile_id = n_tiles;
    // Per-group order (every view's tiles spread over all XCDs, the XCDs walk the views together):
    //  * a gated launch -- AUTO's fallback for the views the band kernel leaves -- renders only some of the views: with one run of all tiles
    //    per XCD a single view would run on the one or two XCDs that hold it;
    //  * fp32 volumes: measured 1.5 % (config 3 shape) to 3.7 % (config 5) faster than one run per XCD, 16-bit volumes 1 % slower
    //    (profiles/r03_xcd_order.txt) -- so 16-bit volumes keep the run per XCD.
    if (p.gate != nullptr || sizeof(TexT) == 4)
        tile_id = xcd_item_per_group(vblock, tiles_per_view * (p.view_to_mpi == nullptr ? p.views_per_mpi : 1), n_tiles, p.band_rot, p.band_split);
#ifdef GMPI_TUNE  // (experiment: flag bit 19 flips the order)
    if (p.flags & (1u << 19)) {
        tile_id = (p.gate != nullptr || sizeof(TexT) == 4) ? (vblock % 8) * per_xcd + vblock / 8
                                                           : xcd_item_per_group(vblock, tiles_per_view * (p.view_to_mpi == nullptr ? p.views_per_mpi : 1), n_tiles);
        if (vblock >= static_cast<unsigned>(per_xcd * 8) && (p.gate != nullptr || sizeof(TexT) == 4)) tile_id = n_tiles;
    }
#endif
    if (tile_id >= n_tiles) return;
    // Views that share one MPI (video paths: views_per_mpi > 1) are interleaved per tile position, so the workgroups
    // that need (nearly) the same texels of a plane run next to each other in time and on the same XCD: the volume is
    // then read from HBM about once per group of views instead of once per view (the rest hits in that XCD's L2).
    int n, trem;
    if (p.view_to_mpi == nullptr && p.views_per_mpi > 1) {
        const int group = tile_id / (tiles_per_view * p.views_per_mpi);           // full groups come first
        const int first = group * p.views_per_mpi, size = min(p.views_per_mpi, p.N - first);
        const int r = tile_id - first * tiles_per_view;
        trem = r / size;
        n = first + (r - trem * size);
    } else {
        n = tile_id / tiles_per_view;
        trem = tile_id - n * tiles_per_view;
    }
    if (view_bits != nullptr ? ((view_bits[n >> 5] >> (n & 31)) & 1u) == 0u : view_gated_out(p, n)) return;  // (AUTO: this view is the band kernel's)
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;

    const int tid = threadIdx.x;
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

    if (p.status != nullptr && trem == 0 && tid == 0) {  // mpi.py:70-72, once per view
        const float ez0 = p.eye_pos[2];
        bool behind = false;
        for (int k = 0; k < D; ++k) behind |= !(dhw[3 * k] >= ez0);
        if (behind) atomicOr(p.status, 4u);
    }

    // ---- this thread's pixel (out-of-image lanes shadow the last row/column) ---------------------------------
    const int px = txi * TW + (tid % TW), py = tyi * TH + (tid / TW);
    const bool active = px < W && py < H;
    const int64_t pix = static_cast<int64_t>(min(py, H - 1)) * W + min(px, W - 1);
    const float rx = rdv[pix], ry = rdv[HW + pix], rz = rdv[2 * HW + pix];
    float dot = rx * zx;  // einsum("nchw,nc->nhw") mpi.py:149
    dot = dot + ry * zy;
    dot = dot + rz * zz;
    const float rcp_rz = 1.0f / rz;  // correctly rounded; hoisted out of the plane loop (see div_by_recip)
    Accum A;

    // ---- tile corner pixels (for the per-plane texel box) ----------------------------------------------------
    const int cx0 = txi * TW, cx1 = min(cx0 + TW - 1, W - 1);
    const int cy0 = tyi * TH, cy1 = min(cy0 + TH - 1, H - 1);

    // ---- loader role, re-derived per chunk from the widest / tallest box of the chunk (set_loader_map below):
    //      thread -> item column `lcol` of the (row,channel) lines lrowc + perpass*r, r < npass <= kNL.  A frontal
    //      view needs 6 of the 7 (16-bit) / 10 of the 14 (fp32) item columns, so its lines fit in 1 instead of 2
    //      (16-bit) / 2 instead of 3 (fp32) passes -- the loader's VALU and request count shrink accordingly.
    int lcol = 0, lrowc = 0, perpass = kRowcPerPass, npass = kNL, dst_base = 0;
    uint32_t g_off[kNL];  // BYTE offset of item r inside a plane, relative to the box origin (32-bit voffset)
    auto set_loader_map = [&](int cols, int max_rows) {
        // (integer division runs on the VALU: readfirstlane tells the compiler the results are wave-uniform, so the
        //  `r < npass` tests below become scalar branches instead of exec-mask regions with vmcnt(0) at their ends)
        perpass = __builtin_amdgcn_readfirstlane(kNT / cols);
        npass = __builtin_amdgcn_readfirstlane((LPR * max_rows + perpass - 1) / perpass);  // <= kNL: fewer columns -> more lines per pass
        lrowc = tid / cols;
        lcol = tid - lrowc * cols;
        dst_base = LAYOUT == 1 ? lrowc * (kPitch / TPI) + lcol : lrowc * (kPitch / 4) + lcol * (TPI / 4);  // in 16-byte units
        if (lrowc >= perpass) lcol = 0x3fffffff;  // the last 512 % cols threads load nothing: every column test fails
#pragma unroll
        for (int r = 0; r < kNL; ++r) {
            const int rowc = lrowc + r * perpass;
            g_off[r] = static_cast<uint32_t>(LAYOUT == 1 ? rowc * s_row + TPI * lcol : (rowc & 3) * s_chan + (rowc >> 2) * s_row + TPI * lcol) *
                       static_cast<uint32_t>(sizeof(TexT));
        }
    };

    for (int kc = 0; kc < D; kc += kChunk) {
        const int kn = min(kChunk, D - kc);
        // ---- per-plane geometry: texel box of the pixel rows [y_lo, y_hi] of this tile from their 4 corner pixels;
        //      returns (workgroup-uniform) whether some plane's box exceeds the staging buffer ------------------
        //      and the largest item count per row / row count of the chunk (for the loader map) -----------------
        int max_nq = kCols, max_rows = kMaxRows;
        auto build_table = [&](int y_lo, int y_hi) -> bool {
            __syncthreads();  // the previous table / staging buffers / maxima are no longer read
            if (tid < 2) box_max[tid] = 0;
            __syncthreads();
            int nq_max = 0, row_max = 0;  // (a box that does not fit counts as 1 << 20 items)
            for (int t = tid; t < kn; t += kNT) {
                const int k = kc + t;
                const float d = dhw[3 * k + 0], ph = dhw[3 * k + 1], pw = dhw[3 * k + 2];
                const float zdiff = d - ez;
                float mnx = __builtin_inff(), mxx = -__builtin_inff(), mny = mnx, mxy = mxx;
                bool finite = true;
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t q = static_cast<int64_t>((c & 2) ? y_hi : y_lo) * W + ((c & 1) ? cx1 : cx0);
                    float ix, iy, s, u, v;
                    plane_coord<AC>(zdiff, ph, pw, ex, ey, rdv[q], rdv[HW + q], rdv[2 * HW + q], cx, cy, ix, iy, s, u, v);
                    finite = finite && (fabsf(ix) < 1e6f) && (fabsf(iy) < 1e6f);  // false for NaN too
                    mnx = fminf(mnx, ix), mxx = fmaxf(mxx, ix), mny = fminf(mny, iy), mxy = fmaxf(mxy, iy);
                }
                int4 ri = make_int4(0, 0, -1, 0);  // qx0, by0, nq (< 0: does not fit), nrows
                int cols = 0, lines = 0;
                if (finite) {
                    const int bx0 = static_cast<int>(floorf(mnx - kBoxEps)), bx1 = static_cast<int>(floorf(mxx + kBoxEps)) + 1;
                    const int by0 = static_cast<int>(floorf(mny - kBoxEps)), by1 = static_cast<int>(floorf(mxy + kBoxEps)) + 1;
                    ri.x = bx0 & ~(TPI - 1);
                    ri.y = by0;
                    ri.z = (bx1 - ri.x) / TPI + 1;
                    ri.w = by1 - by0 + 1;
                    if (ri.z > kCols || ri.w > kMaxRows) {
                        ri.z = -1;
                    } else {  // (qx0 and Wt are multiples of the item width)
                        const int clo = min(max(-ri.x / TPI, 0), ri.z), chi = min(max((Wt - ri.x) / TPI, 0), ri.z);
                        const int rlo = min(max(-by0, 0), ri.w), rhi = min(max(Ht - by0, 0), ri.w);
                        cols = ri.z | clo << 8 | (chi - clo) << 16;
                        lines = LPR * ri.w | LPR * rlo << 8 | LPR * (rhi - rlo) << 16;
                    }
                }
                nq_max = max(nq_max, ri.z < 0 ? 1 << 20 : ri.z);
                row_max = max(row_max, ri.w);
                const float hw = pw * 0.5f, hh = ph * 0.5f;  // exact halves: (2x)/w == x/(w/2)
                const uint64_t origin = reinterpret_cast<uint64_t>(vol + (static_cast<int64_t>(k) * s_plane + static_cast<int64_t>(ri.y) * s_row + ri.x));
                tabL[t] = make_int4(static_cast<int>(origin & 0xffffffffu), static_cast<int>((origin >> 32) & 0xffffu), cols, lines);
                tabF[t] = make_float4(zdiff, hw, hh, 1.0f / hw);
                tabG[t] = make_int4(__float_as_int(1.0f / hh), ri.x, ri.y, 0);
            }
            if (tid < kn) atomicMax(&box_max[0], nq_max), atomicMax(&box_max[1], row_max);
            __syncthreads();  // table and maxima published
            max_nq = __builtin_amdgcn_readfirstlane(box_max[0]);
            max_rows = __builtin_amdgcn_readfirstlane(box_max[1]);
            const bool unfit = max_nq >= (1 << 20);
            max_nq = max(min(max_nq, kCols), 1), max_rows = max(min(max_rows, kMaxRows), 1);
            return unfit;
        };

        // ---- last resort (texture much finer than the image, degenerate rays): direct gather, same arithmetic ----
        auto gather_chunk = [&](bool mine) {
            if (!mine) return;
            for (int t = 0; t < kn; ++t) {
                const float4 rf = tabF[t];
                float ix, iy, s, u, v;
                plane_coord<AC>(rf.x, rf.z + rf.z, rf.y + rf.y, ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
                float smp[4];
                gather_sample<TexT, STRICT>(vol + static_cast<int64_t>(kc + t) * s_plane, s_chan, s_row, Ht, Wt, ix, iy,
                                            check_range, bad, smp);
                blend<STRICT>(A, smp[0], smp[1], smp[2], smp[3], s, dot);
            }
        };

        // ---- register staging of one plane's box: branch-free raw buffer loads --------------------------------
        // One buffer resource per plane (its 4 channel images); items that fall outside the box or outside the
        // texture get the offset 0x80000000, which the hardware range check turns into zeros without touching
        // memory -- no exec masking, loads issue back to back and stay two planes ahead.
#ifdef GMPI_TUNE
        const int load_mask = (p.flags & (1u << 16)) ? 0 : 0xff, store_mask = (p.flags & (1u << 18)) ? 0 : 0xff;  // ablation bits
#else
        constexpr int load_mask = 0xff, store_mask = 0xff;
#endif
        const int chan_bytes = __builtin_amdgcn_readfirstlane(static_cast<int>(s_chan * static_cast<int64_t>(sizeof(TexT))));
        // (predicates are combined with bitwise ops on purpose: `&&` would be lowered to exec-mask control flow)
        auto issue_loads = [&](auto np, int t, u32x4 (&L)[decltype(np)::value], bool (&in_box)[decltype(np)::value]) {
            constexpr int NP = decltype(np)::value;  // passes of this chunk's loader map (compile-time: see run_staged)
            // Issued UNCONDITIONALLY, also for planes past the end of the chunk (all offsets out of range then: no
            // memory access): hipcc's waitcnt insertion only lets a load stay in flight across the next plane's
            // staged store ("vmcnt(NP)" instead of "vmcnt(0)") if every path issues the same number of loads.
            // integer masks (not bools): keeps the uniform liveness logic on the scalar unit
            const int live = t < kn ? load_mask : 0, live_store = t < kn ? store_mask : 0;
            const int4 rl = tabL[min(t, kn - 1)];
            // the descriptor must be PROVABLY wave-uniform or hipcc wraps every buffer op in a waterfall loop
            // (cdna_hip_programming.md T20): pass its inputs through readfirstlane.  Base = the box origin, so the
            // per-lane offsets are the chunk-invariant g_off[]; num_records 2^31: only the explicit out-of-range
            // offset below is rejected (every other lane is inside the texture by the range tests)
            const uint32_t b_lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(rl.x));
            const uint32_t b_hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(rl.y));
            const int cols = __builtin_amdgcn_readfirstlane(rl.z), lines = __builtin_amdgcn_readfirstlane(rl.w);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>((static_cast<uint64_t>(b_hi) << 32) | b_lo), 0, static_cast<int>(0x80000000u), 0x00020000);
            // item columns [clo, clo+ncol) and lines [llo, llo+nline) of the box lie inside the texture (one unsigned
            // compare each); everything else reads as zero ("zeros" padding) without touching memory
            // (a plane that is not live gets empty ranges: the uniform flags stay on the scalar unit)
            const int clo = (cols >> 8) & 0xff, ncol = (cols >> 16) & 0xff & live;
            const int llo = (lines >> 8) & 0xff, nline = (lines >> 16) & 0xff;
            const bool col_ok = static_cast<unsigned>(lcol - clo) < static_cast<unsigned>(ncol);
            // which LDS slots the staged store of this plane has to write: the box itself (texels of it that lie
            // outside the texture are written as the zeros the loads return); lanes outside the box stay idle --
            // LDS write time goes with the number of active lanes (tools/ubench/lds_read_rate.hip)
            const bool col_in_box = lcol < (cols & 0xff & live_store);
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                in_box[r] = col_in_box & (lrowc + r * perpass < (lines & 0xff));
                const bool ok = col_ok & (static_cast<unsigned>(lrowc + (r * perpass - llo)) < static_cast<unsigned>(nline));
                const uint32_t off = ok ? g_off[r] : 0x80000000u;  // == num_records: rejected; off+15 cannot wrap
                if constexpr (LAYOUT == 1) {  // the pair (x, x+1) of the four channel images: scalar offset = channel
                    L[r].x = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
                    L[r].y = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, chan_bytes, 0);
                    L[r].z = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 2 * chan_bytes, 0);
                    L[r].w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 3 * chan_bytes, 0);
                } else {
                    L[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                }
            }
        };
        auto store_box = [&](auto np, float* tile, u32x4 (&L)[decltype(np)::value], const bool (&in_box)[decltype(np)::value]) {
            constexpr int NP = decltype(np)::value;
            if constexpr (LAYOUT == 1 && sizeof(TexT) == 4) {
                // item = (R, G, B, A) of one texel, already in storage order: one 16-byte store, no VALU
                u32x4* dst = reinterpret_cast<u32x4*>(tile) + dst_base;
                const int pass_stride = perpass * kPitch;
                uint32_t mx = 0;
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    if (in_box[r]) dst[r * pass_stride] = L[r];
                    mx = max(max(mx, L[r].x), max(max(L[r].y, L[r].z), L[r].w));
                }
                if (check_range && __builtin_expect(mx > 0x3f800000u, 0)) {
#pragma unroll
                    for (int r = 0; r < NP; ++r)
                        if (quad_out_of_unit(make_float4(__uint_as_float(L[r].x), __uint_as_float(L[r].y), __uint_as_float(L[r].z), __uint_as_float(L[r].w)))) bad |= 2u;
                }
                return;
            } else if constexpr (LAYOUT == 1) {
                // item = (R, G, B, A) dwords of the texel pair (x, x+1): two v_perm per texel interleave them to
                // [r g | b a] (8 bytes per texel); the pair goes out as one 16-byte store
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                u32x4* dst = reinterpret_cast<u32x4*>(tile) + dst_base;
                const int pass_stride = perpass * (kPitch / 2);
                us2 m2 = {0, 0};
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    const uint32_t R = L[r].x, G = L[r].y, B = L[r].z, A = L[r].w;
                    u32x4 o;
                    o.x = __builtin_amdgcn_perm(G, R, 0x05040100u);  // texel x:   (g0 : r0)
                    o.y = __builtin_amdgcn_perm(A, B, 0x05040100u);  //            (a0 : b0)
                    o.z = __builtin_amdgcn_perm(G, R, 0x07060302u);  // texel x+1: (g1 : r1)
                    o.w = __builtin_amdgcn_perm(A, B, 0x07060302u);  //            (a1 : b1)
                    if (in_box[r]) dst[r * pass_stride] = o;
                    m2 = __builtin_elementwise_max(m2, __builtin_elementwise_max(__builtin_elementwise_max(__builtin_bit_cast(us2, R), __builtin_bit_cast(us2, G)),
                                                                                  __builtin_elementwise_max(__builtin_bit_cast(us2, B), __builtin_bit_cast(us2, A))));
                }
                // [0,1] test on the 16-bit patterns (ordered like unsigned ints for non-negative values); -0.0 is legal
                if (check_range && __builtin_expect(max(m2.x, m2.y) > Q::kOneBits, 0)) {
                    auto ok = [](uint32_t h) { return h <= Q::kOneBits || h == 0x8000u; };
#pragma unroll
                    for (int r = 0; r < NP; ++r) {
                        const uint32_t d[4] = {L[r].x, L[r].y, L[r].z, L[r].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (!(ok(d[c] & 0xffffu) && ok(d[c] >> 16))) bad |= 2u;
                    }
                }
                return;
            }
            // LDS slot of item r: line (lrowc + perpass*r), floats [TPI*lcol, TPI*lcol + TPI)
            float4* dst = reinterpret_cast<float4*>(tile) + dst_base;
            const int pass_stride = perpass * (kPitch / 4);
            uint32_t mx = 0;  // max of the fp32 bit patterns staged by this lane (lanes outside the box hold zeros)
            constexpr int NQ = Q::kTexels / 4;  // float4 per item
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                float4 q[NQ];
                Q::cvt(L[r], q);
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    mx = max(max(mx, __float_as_uint(q[h].x)), max(max(__float_as_uint(q[h].y), __float_as_uint(q[h].z)), __float_as_uint(q[h].w)));
                    if (in_box[r]) dst[r * pass_stride + h] = q[h];
                }
            }
            // [0,1] test on bit patterns: non-negative floats order like unsigned ints, so v in [0,1] <=> bits <=
            // 0x3f800000; negative values (sign bit) and NaN/Inf compare above.  -0.0 is legal: exact re-test (cold).
            if (check_range && __builtin_expect(mx > 0x3f800000u, 0)) {
#pragma unroll
                for (int r = 0; r < NP; ++r) {
                    float4 q[NQ];
                    Q::cvt(L[r], q);
#pragma unroll
                    for (int h = 0; h < NQ; ++h)
                        if (quad_out_of_unit(q[h])) bad |= 2u;
                }
            }
        };

        auto composite = [&](int t, const float* __restrict__ tile, bool mine) {
#ifdef GMPI_TUNE
            if (p.flags & (1u << 17)) return;  // ablation: no compositing
#endif
            if (t >= kn) return;  // workgroup-uniform: padding plane
            if (!mine) return;
            const float4 rf = tabF[t];
            const int4 rg = tabG[t];
            float ix, iy, s;
            Footprint f;
            if (STRICT) {
                float u, v;
                plane_coord<AC>(rf.x, rf.z + rf.z, rf.y + rf.y, ex, ey, rx, ry, rz, cx, cy, ix, iy, s, u, v);
                f = footprint(ix, iy, Ht, Wt);
            } else {
                plane_coord_recip<AC>(rf.x, rf.y, rf.z, rf.w, __int_as_float(rg.x), ex, ey, rx, ry, rz, rcp_rz, cx,
                                      cy, ix, iy, s);
                // ATen's vectorised CPU form of the weights: e = 1 - w (equals x1 - ix unless ix < 0); v_fract_f32 = ix - floor(ix)
                const float wx1 = __builtin_amdgcn_fractf(ix), wy1 = __builtin_amdgcn_fractf(iy);
                const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
                f.nw = wx0 * wy0, f.ne = wx1 * wy0, f.sw = wx0 * wy1, f.se = wx1 * wy1;
            }
            // the box contains every tap of the tile (corner argument above); the integer corner is taken from the
            // floor directly (Footprint::x0/y0 carry the gather path's out-of-range sentinel)
            const int lx = (STRICT ? static_cast<int>(floorf(ix)) : floor_to_int(ix)) - rg.y, ly = (STRICT ? static_cast<int>(floorf(iy)) : floor_to_int(iy)) - rg.z;
            // unsigned + clamped: keeps wild coordinates (NaN rays) inside the buffer and proves the base non-negative,
            // so the 8 tap-pair reads become ds_read2_b32 with immediate offsets
            float smp[4];
            if constexpr (LAYOUT == 1 && sizeof(TexT) == 4) {
                // texel (lx, ly) = 16 bytes at 16 * (ly * kPitch + lx): the four tap texels are four ds_read_b128
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef const f32x4 __attribute__((address_space(3))) lds_ctexel4;
                const uint32_t idx = min(static_cast<uint32_t>(__mul24(ly, kPitch) + lx), static_cast<uint32_t>(kMaxRows * kPitch - kPitch - 2));
                uint32_t tile_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const float __attribute__((address_space(3)))*)tile));
                asm volatile("" : "+s"(tile_addr));
                uint32_t a_tap = tile_addr + 16u * idx;
                asm volatile("" : "+v"(a_tap));
                lds_ctexel4* __restrict__ tp = reinterpret_cast<lds_ctexel4*>(static_cast<uintptr_t>(a_tap));
                const f32x4 q_nw = tp[0], q_ne = tp[1], q_sw = tp[kPitch], q_se = tp[kPitch + 1];
#pragma unroll
                for (int c = 0; c < 4; ++c) smp[c] = bilerp<STRICT>(q_nw[c], q_ne[c], q_sw[c], q_se[c], f);
            } else if constexpr (LAYOUT == 1) {
                // texel (lx, ly) sits at 8 * (ly * kPitch + lx) bytes: the four tap texels are two ds_read2_b64
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                typedef const u32x2 __attribute__((address_space(3))) lds_ctexel;
                const uint32_t idx = min(static_cast<uint32_t>(__mul24(ly, kPitch) + lx), static_cast<uint32_t>(kMaxRows * kPitch - kPitch - 2));
                uint32_t tile_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const float __attribute__((address_space(3)))*)tile));
                asm volatile("" : "+s"(tile_addr));
                uint32_t a_tap = tile_addr + 8u * idx;
                asm volatile("" : "+v"(a_tap));
                lds_ctexel* __restrict__ tp = reinterpret_cast<lds_ctexel*>(static_cast<uintptr_t>(a_tap));
                const u32x2 q_nw = tp[0], q_ne = tp[1], q_sw = tp[kPitch], q_se = tp[kPitch + 1];
                float t_nw[4], t_ne[4], t_sw[4], t_se[4];
                Q::unpack2(q_nw.x, t_nw[0], t_nw[1]), Q::unpack2(q_nw.y, t_nw[2], t_nw[3]);
                Q::unpack2(q_ne.x, t_ne[0], t_ne[1]), Q::unpack2(q_ne.y, t_ne[2], t_ne[3]);
                Q::unpack2(q_sw.x, t_sw[0], t_sw[1]), Q::unpack2(q_sw.y, t_sw[2], t_sw[3]);
                Q::unpack2(q_se.x, t_se[0], t_se[1]), Q::unpack2(q_se.y, t_se[2], t_se[3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) smp[c] = bilerp<STRICT>(t_nw[c], t_ne[c], t_sw[c], t_se[c], f);
            } else {
                const uint32_t idx = min(static_cast<uint32_t>(__mul24(ly, 4 * kPitch) + lx), static_cast<uint32_t>(kCapFloats - 7 * kPitch - 2));
                // LDS byte addresses of the two texel rows, made opaque to the optimiser: it would otherwise fold the
                // buffer's constant offset into every tap address and pay one v_add per ds_read2_b32 (the instruction's
                // offset fields are 8 bits of dwords: channel strides 0/56/112/168 fit, the staging area's base does not)
                typedef const float __attribute__((address_space(3))) lds_cfloat;
                uint32_t tile_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_cfloat*)tile));
                asm volatile("" : "+s"(tile_addr));
                uint32_t a_top = tile_addr + 4u * idx, a_bot = a_top + 16u * kPitch;
                asm volatile("" : "+v"(a_top), "+v"(a_bot));
                lds_cfloat* __restrict__ top = reinterpret_cast<lds_cfloat*>(static_cast<uintptr_t>(a_top));
                lds_cfloat* __restrict__ bot = reinterpret_cast<lds_cfloat*>(static_cast<uintptr_t>(a_bot));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float t_nw = top[c * kPitch], t_ne = top[c * kPitch + 1];
                    const float t_sw = bot[c * kPitch], t_se = bot[c * kPitch + 1];
                    smp[c] = bilerp<STRICT>(t_nw, t_ne, t_sw, t_se, f);
                }
            }
            blend<STRICT>(A, smp[0], smp[1], smp[2], smp[3], s, dot);
        };
        auto run_staged = [&](auto np, bool mine) {
            constexpr int NP = decltype(np)::value;
            constexpr int PFX = NP * PF <= kMaxStagedItems ? PF : (NP * 2 <= kMaxStagedItems ? 2 : 1);  // planes in flight
            u32x4 L[PFX][NP];  // staging registers: the loads run PFX planes ahead of the compositor
            bool in_box[PFX][NP];  // ... and the lanes that will store them
    #pragma unroll
            for (int u = 0; u < PFX; ++u) issue_loads(np, u, L[u], in_box[u]);
            // every stage runs for every t (a chunk is padded to a multiple of PFX planes: the padding planes load
            // and stage nothing and are not composited), so the memory operations are the same on every path
            for (int t = 0; t < kn; t += PFX) {
    #pragma unroll
                for (int u = 0; u < PFX; ++u) {
                    float* tile = tile0 + ((t + u) & 1) * kCapFloats;
                    // the loader phase runs at raised priority: with 8 waves per SIMD its few instructions otherwise queue behind the
                    // compositors of the other workgroups, and the barrier releases 8 waves late (config 3: -1.8 %, fp32 -2.3 %)
                    __builtin_amdgcn_s_setprio(2);
                    store_box(np, tile, L[u], in_box[u]);
                    __syncthreads();  // box t+u visible; everybody is done reading box t+u-1 (the other buffer)
                    issue_loads(np, t + u + PFX, L[u], in_box[u]);  // in flight while the PFX planes before it are composited
                    __builtin_amdgcn_s_setprio(0);
                    composite(t + u, tile, mine);
                }
            }
        };

        // ---- whole tile if every box fits; else its two 32x8 halves one after the other (tilted cameras shear the
        //      box: 47x29 texels at the 2-sigma FFHQ pose, 21 rows per half); else the direct gather ------------------
        // (one loop, one call site per lambda: a second inlined copy of the plane loop costs registers in the first)
        const int half = (tid / TW) / (TH / 2);  // wave-uniform: waves 0-3 upper half, 4-7 lower half
#pragma unroll 1
        for (int h = -1; h < 2; ++h) {             // h = -1: the whole tile; h = 0, 1: its halves
            const int y_lo = h < 0 ? cy0 : cy0 + h * (TH / 2);
            if (y_lo > cy1) break;
            const int y_hi = h < 0 ? cy1 : min(y_lo + TH / 2 - 1, cy1);
            const bool unfit = build_table(y_lo, y_hi);
            if (h < 0 && unfit) continue;          // try the halves
            const bool mine = h < 0 || half == h;
            if (!unfit) set_loader_map(max_nq, max_rows);
            // the pass count is a compile-time constant of the plane loop (a run-time trip count makes hipcc wait for
            // the prefetch, vmcnt(0), before compositing): one instance per count, selected per chunk
            if (!unfit) {
                if (kNL >= 3 && npass >= 3) run_staged(std::integral_constant<int, (kNL >= 3 ? 3 : 1)>{}, mine);
                else if (kNL >= 2 && npass == 2) run_staged(std::integral_constant<int, (kNL >= 2 ? 2 : 1)>{}, mine);
                else run_staged(std::integral_constant<int, 1>{}, mine);
            }
            else gather_chunk(mine);
            if (h < 0) break;
        }
    }

    // ---- assert_not_out_of_last_plane (mpi.py:381-395): u,v of the last plane, once per pixel ------------------
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


template <typename TexT, bool AC, bool STRICT, int TW, int MINW, int PF, int LAYOUT>
__global__ __launch_bounds__(kNT, MINW) void render_lds_kernel(const KParams p, const int tiles_x, const int tiles_y, const int n_tiles) {
    render_lds_tile<TexT, AC, STRICT, TW, MINW, PF, LAYOUT>(p, tiles_x, tiles_y, n_tiles, blockIdx.x);
}

// AUTO's fallback launch for the views the band kernel leaves (KParams::gate): usually NO view is left, and a grid of one workgroup per
// tile of every view that only exits costs 17 us for BASELINE config 3 (8192 workgroups: the dispatch rate).  So the gated launch has a
// fixed, small grid whose workgroups walk the virtual block indices round by round: 2 us when there is nothing to do.
// The grid (round 6; profiles/r06_band_order.txt): ONE round of resident workgroups -- 4 per CU with 16-bit volumes: 1024; 3 per CU with fp32 volumes (53 KB of
// LDS each): 768 (1024 there ran its last 256 alone in a second round: +22 % on a launch whose eight views all came through the gate).  Views that SHARE an MPI
// are interleaved per tile position (8 views: view = (vblock / 8) % 8), and a stride of 8 x 128 showed a workgroup the SAME view on every round -- two views of
// eight left to this kernel were rendered by a quarter of the workgroups, four tiles each (0.645 ms): for such launches the grid is a multiple of 8 whose eighth
// is ODD, 8 x 127 = 1016 / 8 x 95 = 760 (760 measured best there: 768 +8 %).  Everything else keeps the power of two: two gated views of 512^2 are 1024 tiles,
// which 1016 workgroups would render in two rounds (0.2135 ms against 0.1725).  Rounds 3-5 launched 1024 for everything AND kept a byte per view in LDS, which took
// the fp32 instances over a third of a CU's LDS: two workgroups per CU, 29 % slower than the plain tile kernel on the same tiles (below).  (Handing the tiles out
// by tickets -- an atomic counter per XCD -- was built and measured: 512 returning atomics per counter serialise at more than 1 us each: 1.27 ms for what the plain
// tile kernel does in 0.545.)
template <typename TexT> constexpr unsigned gated_grid(bool shared) { return sizeof(TexT) == 4 ? (shared ? 760u : 768u) : (shared ? 1016u : 1024u); }
constexpr int kGatedViews = 512;   // views whose gate a workgroup caches in LDS (kNT threads read one gate word each)
template <typename TexT, bool AC, bool STRICT, int TW, int MINW, int PF, int LAYOUT>
__global__ __launch_bounds__(kNT, MINW) void render_lds_gated_kernel(const KParams p, const int tiles_x, const int tiles_y, const int n_tiles,
                                                                     const unsigned n_vblocks) {
    // the gate words once per workgroup (a global load per tile would cost more than the dispatch it saves); nothing to do: exit.  ONE BIT per view: the fp32
    // instances stage 53 000 bytes per workgroup and three workgroups share a CU's 160 KB -- rounds 3-5 kept a BYTE per view here (512 bytes), which took the
    // allocation over a third of the LDS: the gated kernel ran two workgroups per CU, 29 % slower than the plain tile kernel on the same tiles (0.706 against
    // 0.548 ms for config 4's eight views, one tile per workgroup in both).
    __shared__ uint32_t view_bits[kGatedViews / 32];
    const bool cached = p.N <= kGatedViews;
    if (cached) {
        const bool mine = static_cast<int>(threadIdx.x) < p.N && !view_gated_out(p, static_cast<int>(threadIdx.x));
        if (!__syncthreads_or(mine ? 1 : 0)) return;   // (the usual case: one load, one barrier)
        if (threadIdx.x < kGatedViews / 32) view_bits[threadIdx.x] = 0u;
        __syncthreads();
        if (mine) atomicOr(&view_bits[threadIdx.x >> 5], 1u << (threadIdx.x & 31));
        __syncthreads();
    }
    for (unsigned vb = blockIdx.x; vb < n_vblocks; vb += gridDim.x) {
        render_lds_tile<TexT, AC, STRICT, TW, MINW, PF, LAYOUT>(p, tiles_x, tiles_y, n_tiles, vb, cached ? view_bits : nullptr);
        __syncthreads();  // the next tile reuses the staging buffers and the table
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int elem_size(int dtype) { return dtype == 0 ? 4 : 2; }

bool lds_variant_supports(const KParams& p, int dtype) {
    const int es = elem_size(dtype);
    const int tpi = 16 / es;  // texels per 16-byte loader item
    if (p.Wt % tpi != 0) return false;
    if (reinterpret_cast<uintptr_t>(p.rgba) % 16 != 0) return false;
    if (p.s_row % tpi != 0 || p.s_chan % tpi != 0 || p.s_plane % tpi != 0 || p.s_mpi % tpi != 0) return false;
    // the in-plane item offset is kept in 32 bits
    const int64_t span = 3 * p.s_chan + (TileCfg<32>::kMaxLines / 4 + 1) * p.s_row + 128;
    if (span >= (int64_t(1) << 31) / es) return false;
    return true;
}

constexpr int kTileW = 32;

int lds_variant_query(int what) {
    switch (what) {
        case 3: return lds_bytes<kTileW>();
        case 4: return kTileW;
        case 5: return kNT / kTileW;
        default: return -1;
    }
}

template <typename TexT, int TW, int MINW, int PF, int LAYOUT>
static hipError_t launch_lds_t(const KParams& p, hipStream_t stream) {
    constexpr int TH = kNT / TW;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int n_tiles = tiles_x * tiles_y * p.N;
    const unsigned grid_x = (p.gate != nullptr || sizeof(TexT) == 4 || (p.flags & (1u << 19))) ? xcd_grid_per_group(tiles_x * tiles_y * (p.view_to_mpi == nullptr ? p.views_per_mpi : 1), n_tiles)
                                              : static_cast<unsigned>(((n_tiles + 7) / 8) * 8);
    const dim3 grid(grid_x), block(kNT);
    const bool ac = p.flags & 1u, strict = p.flags & (1u << 4);
    if (p.gate != nullptr) {
        unsigned gg = std::min(grid_x, gated_grid<TexT>(p.view_to_mpi == nullptr && p.views_per_mpi > 1));
#ifdef GMPI_TUNE  // GMPI_TUNE_GGRID: the gated launch's grid (A/B)
        static const int env_gg = [] { const char* e = getenv("GMPI_TUNE_GGRID"); return e ? atoi(e) : 0; }();
        if (env_gg > 0) gg = std::min(grid_x, static_cast<unsigned>(env_gg));
#endif
        const dim3 ggrid(gg);
        if (ac && strict) hipLaunchKernelGGL((render_lds_gated_kernel<TexT, true, true, TW, MINW, PF, LAYOUT>), ggrid, block, 0, stream, p, tiles_x, tiles_y, n_tiles, grid_x);
        else if (ac) hipLaunchKernelGGL((render_lds_gated_kernel<TexT, true, false, TW, MINW, PF, LAYOUT>), ggrid, block, 0, stream, p, tiles_x, tiles_y, n_tiles, grid_x);
        else if (strict) hipLaunchKernelGGL((render_lds_gated_kernel<TexT, false, true, TW, MINW, PF, LAYOUT>), ggrid, block, 0, stream, p, tiles_x, tiles_y, n_tiles, grid_x);
        else hipLaunchKernelGGL((render_lds_gated_kernel<TexT, false, false, TW, MINW, PF, LAYOUT>), ggrid, block, 0, stream, p, tiles_x, tiles_y, n_tiles, grid_x);
        return hipGetLastError();
    }
    if (ac && strict) hipLaunchKernelGGL((render_lds_kernel<TexT, true, true, TW, MINW, PF, LAYOUT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else if (ac) hipLaunchKernelGGL((render_lds_kernel<TexT, true, false, TW, MINW, PF, LAYOUT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else if (strict) hipLaunchKernelGGL((render_lds_kernel<TexT, false, true, TW, MINW, PF, LAYOUT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    else hipLaunchKernelGGL((render_lds_kernel<TexT, false, false, TW, MINW, PF, LAYOUT>), grid, block, 0, stream, p, tiles_x, tiles_y, n_tiles);
    return hipGetLastError();
}

constexpr int kTileViewRotation = 1, kTileRunPieces = 2;   // (as the band kernel: profiles/r06_band_order.txt -- config 3 fp32 on the tile kernel 1.324 -> 1.309 ms, config 4 and tilted sets unchanged)

hipError_t launch_lds(const KParams& p0, int dtype, int tune, hipStream_t stream) {
    KParams p = p0;
    p.band_rot = kTileViewRotation, p.band_split = kTileRunPieces;   // (the XCD <-> tile-run assignment of the per-group order: gmpi_device.hpp xcd_item_per_group)
#ifdef GMPI_TUNE  // profiling builds: GMPI_TUNE_WAVE + 256 no memory traffic, + 512 no compositing (loader only), + 1024 no LDS stores
    p.flags |= static_cast<uint32_t>((tune >> 8) & 15) << 16;
    {
        static const int env_rot = [] { const char* e = getenv("GMPI_TUNE_ROT"); return e ? atoi(e) : -1; }();
        static const int env_split = [] { const char* e = getenv("GMPI_TUNE_SPLIT"); return e ? atoi(e) : 0; }();
        if (env_rot >= 0) p.band_rot = env_rot;
        if (env_split > 0) p.band_split = env_split;
    }
#endif
    // Shipped instances only: fp32 volumes keep fp32 planes in LDS (LAYOUT 0, 3 workgroups per CU), 16-bit volumes their raw
    // texels, interleaved (LAYOUT 1, 4 workgroups per CU); 32x16 pixel tiles, one plane of prefetch -- the values the round-1
    // ablations settled on (profiles/r01_ablation.txt; the other combinations are no longer instantiated).
    (void)tune;
    switch (dtype) {
        case 0: return launch_lds_t<float, kTileW, 6, 1, 0>(p, stream);
        case 1: return launch_lds_t<bf16_t, kTileW, 8, 1, 1>(p, stream);
        default: return launch_lds_t<f16_t, kTileW, 8, 1, 1>(p, stream);
    }
}

}  // namespace gmpi
