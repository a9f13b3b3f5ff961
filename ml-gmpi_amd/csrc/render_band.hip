// render_band.hip -- GMPI_VARIANT_BAND: 256 x 8 (bf16 volumes) / 128 x 8 (fp32 volumes) pixel bands, texel boxes moved HBM -> LDS by the LDS-DMA path.
//
// Why (round 3).  Two measurements decide the shape (profiles/r03_band_loader.txt, r03_probe.txt):
//  * the MEMORY side of a tile decomposition depends on how long the contiguous pieces are that a workgroup asks for in one plane
//    step: with nothing but the loads running, 32 x 16 pixel tiles (80-byte pieces of a bf16 volume) take 0.92 ms for BASELINE
//    config 3 however deep the prefetch and however the loads are issued (dword loads, 16-byte loads, LDS-DMA: the same), 128 x 8
//    bands 0.78-0.80 ms, 256 x 8 bands (528-byte pieces, the rows of a box fill whole 128-byte lines) 0.58 ms = the rate of a
//    streaming read -- at 2 workgroups of 1024 threads per CU;
//  * the COMPUTE side is bound by instruction issue and by what a plane step costs besides the pixels: the coordinate chain,
//    16 taps, bilinear and blend are 55 VALU instructions per pixel and plane (51 of them in the 1.0-1.1 ns class), but a 32 x 16
//    tile kernel pays 19 VALU + 41 SALU + a barrier per 64 pixels on top (loader, range check, table reads).
// So: a workgroup owns a band of sub-blocks of 64 x 8 pixels (struct Geo: bf16 1024 threads = 4 sub-blocks x 4 waves x 2 pixels per
// thread, fp32 512 threads = 2 sub-blocks x 4 waves x 2 pixels per thread -- the texel rows of a band's boxes are 576 bytes either way).  Each
// sub-block has its own texel box per plane (a tilted camera shears a 256-pixel band over up to 54 texel rows, a sub-block over
// 6-14), staged as raw texels, planar [row][channel][x], one DMA item = 16 bytes of a channel row, lane-linear LDS image,
// exec-masked DMA instructions, zeros padding by the buffer range check, two staging buffers, ONE s_barrier per plane, taps by
// ds_read_u16_d16_hi (bf16: the loaded half IS the fp32 value) / ds_read2_b32.  What is uniform per (band, plane, sub-block) comes
// from a table a small kernel writes into the caller's workspace in front of the render kernel (scalar loads, no LDS table).
// A band with a box that does not fit its buffers (strongly tilted camera) is rendered by the direct gather; GMPI_VARIANT_AUTO
// hands the whole VIEW to the tile kernel instead (KParams::gate, gmpi_abi.hip).
// Round 4 (profiles/r04_band_variants.txt): the plane step of the bf16 default-mode instances is software-pipelined -- a batch of 8 taps is
// issued one batch ahead of the wait that lands it -- and the loader offset lives in a vector register.  (Round 5: the [0,1] test folds the landed
// TAPS into a running maximum instead of reading every staged item back from LDS: the launch runs at the package power limit, see tap_fold8.)  What the measurements of that round say about the rest: a wave's own DMA is NOT what it waits for at the barrier (393 of
// 4600 cycles per plane with or without memory traffic); the wait is the skew between the 16 waves; the scalar loads of the next step's
// records belong BEHIND the last pixel (the waves that arrive first warm the scalar cache for the last one: issued at the top of the step
// they cost 5 %), and the records of a plane must share one cache line across the sub-blocks for that to work (a plane-major table: +10 %).
// Arithmetic: gmpi_device.hpp (bit-identical to the oracle in strict-order mode).
#include "gmpi_device.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>


namespace gmpi {
namespace band {

constexpr int kNT = 1024;              // threads per workgroup (16 wavefronts)
constexpr int SBW = 64, SBH = 8;       // sub-block: 64 x 8 pixels
constexpr float kBoxEps = 1.0f / 64;
// Band columns per XCD window (band_pos below), measured in round 6 (profiles/r06_band_order.txt: three boxes, alternating repeats, L2 -> fabric bytes
// from FETCH_SIZE passes).  Two columns: config 3 (bf16) 0.7945 -> 0.7790 ms with 0.987 x instead of 1.074 x the algorithmic bytes crossing the L2s,
// config 3 with an fp32 volume 1.170 -> 1.163 ms, 0.988 x instead of 1.081 x.  Deep fp32 stacks (config 5: 256 planes) used to be the exception -- one
// column was 1.5-4.6 % faster there although it moves more bytes (1.142 x against 1.075 x): with one column every XCD owns the full height of every view,
// with two the top or the bottom half of a window, and a launch lasts as long as its busiest XCD (below).  Since an XCD's run comes in two pieces from
// opposite regions (kRunPieces) that no longer holds: two columns are 0.1-1.7 % faster there too (two boxes, 3 + 6 alternating repeats), with 6 % fewer bytes.
constexpr int kWindowCols16 = 2, kWindowCols32 = 2, kWindowCols32Deep = 2, kDeepPlanes = 128;
// Per-view rotation of the XCD <-> run assignment (gmpi_device.hpp xcd_item_per_group; round 6).  Workgroups are dealt to the XCDs round-robin and stay there: a launch
// ends when the most loaded XCD is done, and with every XCD rendering the SAME region of every view the regional cost differences (keystone: taller boxes, a third DMA
// pass) add up over the views -- s_memtime stamps per workgroup (tools/kbench KB_STAMPS=1): the busiest XCD carries 3.6 % more than the mean on config 3.  Rotating the
// assignment by one XCD per view: config 3 bf16 0.7982 -> 0.7743 ms (-3.0 %) and 0.7856 -> 0.7676 (-2.3 %) on two boxes, four / three alternating repeats; fp32 -1.2 % / -0.8 %;
// config 5 -0.3 % / -1 %; a rotation by 3 is as good on 16-bit volumes and worse on config 5 (profiles/r06_band_order.txt).
constexpr int kViewRotation = 1;
// ... each XCD's run of a view cut into two pieces from opposite regions of the view (xcd_item_per_group `split`): what is expensive at one end of a view tends to be cheap at
// the other -- config 3 bf16 -0.9...-1.2 %, fp32 -0.4...-0.8 %, config 5 -0.2...-0.6 % on top of the rotation (two boxes, three / four alternating repeats; 4 or 8 pieces: no better).
constexpr int kRunPieces = 2;
// ... and the last bands of every XCD's last run handed out by tickets (see the kernel): -1 % on one box, nothing on another for the bench's pose draw (what is left after the
// rotation is mostly silicon: even and odd XCDs differ by ~4 % per band on some boxes), -13 % when a launch's bands differ a lot (explicit GMPI_VARIANT_BAND with bands on the
// direct-gather path); 64 per XCD is too many (+1 %: a stolen band's texels come from another XCD's neighbourhood).
constexpr int kTicketTail = 16;
constexpr float kCoordLimit = 16384.0f;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int V> using ic = std::integral_constant<int, V>;

// (two passes: boxes of up to 2 * kRPP rows -- the usual case)
template <int D0, int STEP>
__device__ __forceinline__ void dma16x2(uint32_t voff, uint32_t soff1, const u32x4& rsrc, uint32_t lds_dst, uint64_t m0, uint64_t m1);

// The band of a workgroup by storage type -- the same LDS budget (two staging buffers of 15 texel rows x 80 texels per sub-block = 76.8 KB, two
// workgroups per CU) buys
//   bf16: 4 sub-blocks = 256 x 8 pixels, 4 waves per sub-block, 2 pixels per thread (rows j, j + 4): 1024 threads;
//   fp32: 2 sub-blocks = 128 x 8 pixels, 4 waves per sub-block, 2 pixels per thread: 512 threads -- a texel row of a box is 576 bytes either way.
// Pixels per thread are a build parameter (round 4, profiles/r04_band_variants.txt item 14): with twice the pixels a workgroup has half the waves, i.e.
// half the per-wave skeleton of a plane step (barrier, DMA issue, record loads) and half the loader lanes per sub-block (5 DMA passes of 3 rows instead
// of 3 of 6).  fp32 gains 3-4 % from 2 pixels (config 3 with an fp32 volume 1.218 -> 1.184 ms, config 5 3.36 -> 3.23); bf16 LOSES 2 % with 4 (0.800 ->
// 0.815: four waves per SIMD no longer cover its 32 two-byte taps per pixel pair), so it stays at 2.
#ifndef GMPI_BAND_PPT16
#define GMPI_BAND_PPT16 2
#endif
#ifndef GMPI_BAND_PPT32
#define GMPI_BAND_PPT32 2
#endif
template <typename TexT> struct Geo {
    static constexpr int kES = static_cast<int>(sizeof(TexT));
    static constexpr int NSB = kES == 2 ? 4 : 2;            // sub-blocks per band
    static constexpr int PPT = kES == 2 ? GMPI_BAND_PPT16 : GMPI_BAND_PPT32;  // pixels per thread: rows j + WPS q of the sub-block (j = wave % WPS)
    static constexpr int WPS = SBH / PPT;                   // waves per sub-block
    static constexpr int kSubLanes = 64 * WPS;              // loader lanes per sub-block
    static constexpr int kThreads = NSB * kSubLanes;        // threads per workgroup: 1024 as shipped (512 with twice the pixels per thread)
    static constexpr int kWavesPerSimd = kThreads / 128;    // two workgroups per CU
    static_assert(kThreads <= kNT && SBH % PPT == 0, "waves");
    static constexpr int kTPI = 16 / kES;                   // texels per 16-byte item
    static constexpr int kCols = kES == 2 ? 10 : 20;        // items per (row, channel) line: 80 texels
    static constexpr int kLineBytes = kCols * 16;
    static constexpr int kRowBytes = 4 * kLineBytes;
    static constexpr int kIPR = 4 * kCols;                  // items per texel row
    static constexpr int kMaxRows = 15;                     // rows per sub-block buffer
    static constexpr int kCapItems = kMaxRows * kIPR;
    static constexpr int kSubBytes = kCapItems * 16;        // 9600 (bf16) / 19200 (fp32)
    static constexpr int kBufBytes = NSB * kSubBytes;
    // One DMA pass of a sub-block's lanes moves kRPP whole texel rows (lanes beyond kRPP * kIPR idle): a lane's item of pass r is its item
    // of pass 0 moved down by r * kRPP rows -- one per-lane offset register, the pass in the instruction's scalar offset.
    static constexpr int kRPP = kSubLanes / kIPR;           // 6 rows per pass
    static constexpr int kPassItems = kRPP * kIPR;          // 240 (bf16) / 480 (fp32) active lanes
    static constexpr int kNP = (kMaxRows + kRPP - 1) / kRPP;  // DMA passes per plane at most: 3 as shipped, 5 with half the loader lanes
    static_assert(kNP == 3 || kNP == 5, "passes");
    static constexpr int kOffBytes = kThreads * 4;          // per-lane loader offsets (kept in LDS: a VGPR through the plane loop is dearer)
    static constexpr int kLdsBytes = kOffBytes + 2 * kBufBytes;
};
static_assert(Geo<bf16_t>::kLdsBytes * 2 <= 160 * 1024 && Geo<f16_t>::kLdsBytes * 2 <= 160 * 1024 && Geo<float>::kLdsBytes * 2 <= 160 * 1024, "2 workgroups per CU");

// One DMA instruction: lanes of `mask` move 16 bytes each from (descriptor base + voff + soff) to LDS byte M0 + 16 * lane.
// (s_nop: an s_mov to M0 needs one wait state before an LDS-DMA reads it.)
__device__ __forceinline__ void dma16(uint32_t voff, uint32_t soff, const u32x4& rsrc, uint32_t lds_dst, uint64_t mask) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen lds\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(mask), "s"(soff) : "memory");
}
// The (up to) three passes of a plane's box in one statement: 11 scalar instructions.  Pass r: lanes m[r], scalar offset r * pass_off,
// LDS destination lds_dst + D0 + r * STEP (D0, STEP compile-time).  A pass the box does not need has an empty mask.
template <int D0, int STEP>
__device__ __forceinline__ void dma16x3(uint32_t voff, uint32_t soff1, uint32_t soff2, const u32x4& rsrc, uint32_t lds_dst, uint64_t m0, uint64_t m1, uint64_t m2) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %4\n\ts_add_i32 m0, %3, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                 "s_mov_b64 exec, %5\n\ts_add_i32 m0, %3, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %7 offen lds\n\t"
                 "s_mov_b64 exec, %6\n\ts_add_i32 m0, %3, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %8 offen lds\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(m0), "s"(m1), "s"(m2), "s"(soff1), "s"(soff2), "i"(D0), "i"(D0 + STEP), "i"(D0 + 2 * STEP)
                 : "memory", "scc");
}
template <int D0, int STEP>
__device__ __forceinline__ void dma16x2(uint32_t voff, uint32_t soff1, const u32x4& rsrc, uint32_t lds_dst, uint64_t m0, uint64_t m1) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %4\n\ts_add_i32 m0, %3, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                 "s_mov_b64 exec, %5\n\ts_add_i32 m0, %3, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %6 offen lds\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(m0), "s"(m1), "s"(soff1), "i"(D0), "i"(D0 + STEP)
                 : "memory", "scc");
}
template <int D0, int STEP>
__device__ __forceinline__ void dma16x4(uint32_t voff, uint32_t s1, uint32_t s2, uint32_t s3, const u32x4& rsrc, uint32_t lds_dst, uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %4\n\ts_add_i32 m0, %3, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                 "s_mov_b64 exec, %5\n\ts_add_i32 m0, %3, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %8 offen lds\n\t"
                 "s_mov_b64 exec, %6\n\ts_add_i32 m0, %3, %13\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %9 offen lds\n\t"
                 "s_mov_b64 exec, %7\n\ts_add_i32 m0, %3, %14\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %10 offen lds\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(s1), "s"(s2), "s"(s3),
                   "i"(D0), "i"(D0 + STEP), "i"(D0 + 2 * STEP), "i"(D0 + 3 * STEP)
                 : "memory", "scc");
}
template <int D0, int STEP>
__device__ __forceinline__ void dma16x5(uint32_t voff, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4, const u32x4& rsrc, uint32_t lds_dst, uint64_t m0, uint64_t m1, uint64_t m2,
                                        uint64_t m3, uint64_t m4) {
    uint64_t save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %4\n\ts_add_i32 m0, %3, %13\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                 "s_mov_b64 exec, %5\n\ts_add_i32 m0, %3, %14\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %9 offen lds\n\t"
                 "s_mov_b64 exec, %6\n\ts_add_i32 m0, %3, %15\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %10 offen lds\n\t"
                 "s_mov_b64 exec, %7\n\ts_add_i32 m0, %3, %16\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %11 offen lds\n\t"
                 "s_mov_b64 exec, %8\n\ts_add_i32 m0, %3, %17\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %12 offen lds\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(s1), "s"(s2), "s"(s3), "s"(s4),
                   "i"(D0), "i"(D0 + STEP), "i"(D0 + 2 * STEP), "i"(D0 + 3 * STEP), "i"(D0 + 4 * STEP)
                 : "memory", "scc");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (Measured and dropped, profiles/r03_band_variants.txt: the same rendezvous among the 4 waves of ONE sub-block only -- an LDS counter, arrive =
//  ds_add by lane 0, wait = poll with s_sleep -- is bit-exact but 7-9 % SLOWER than the workgroup barrier: the sub-blocks drift apart by planes,
//  their requests no longer form the 528-byte pieces per row the memory side likes, and the polling costs issue slots.)
// (a d16_hi load zeroes the low half of its destination on gfx950: tools/ubench/r3_probe.hip `sem`)
template <int O> __device__ __forceinline__ void tap16(uint32_t& t, uint32_t a) {
    asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "=v"(t) : "v"(a), "i"(O));
}
template <int O> __device__ __forceinline__ void tap32x2(uint32_t& t0, uint32_t& t1, uint32_t a) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "i"(O), "i"(O + 1));
    t0 = v.x, t1 = v.y;
}

// blockIdx / band index -> view n and band position inside the view.  Views that share one MPI (views_per_mpi > 1) are interleaved per band
// position, so that the workgroups that need (nearly) the same texels of a plane run next to each other in time and on the same XCD.
__device__ __forceinline__ void band_to_view(const KParams& p, int band_id, int per_view, int& n, int& brem) {
    if (p.view_to_mpi == nullptr && p.views_per_mpi > 1) {
        const int group = band_id / (per_view * p.views_per_mpi);
        const int first = group * p.views_per_mpi, size = min(p.views_per_mpi, p.N - first);
        const int r = band_id - first * per_view;
        brem = r / size;
        n = first + (r - brem * size);
    } else {
        n = band_id / per_view;
        brem = band_id - n * per_view;
    }
}

// band index inside the view -> (column, row) of bands.  The bands of a view are walked in WINDOWS of `wc` band columns: window by window, inside a
// window row by row (the wc bands of a row next to each other).  XCD x = blockIdx % 8 owns a contiguous run of that walk (xcd_item_per_group), so
// an XCD's share of a view is wc columns wide and the workgroups that are resident together in one L2 are neighbours in both directions.
//   wc = 1: column-major (round 5): an XCD walks DOWN one column of bands -- vertical neighbours share 2-3 of their ~10 box rows;
//   wc >= bands_x: row-major (rounds 3-4): horizontal neighbours share the partial 128-byte lines at every column border;
//   in between: both (round 6; the table order x {ms, L2 -> fabric bytes} is in profiles/r06_band_order.txt).
// No pixel's result depends on it (the strict-order tests compare every order against the same oracle).
__device__ __forceinline__ void band_pos(int brem, int bands_x, int bands_y, int wc, int& bxi, int& byi) {
    const int per_window = wc * bands_y;
    const int g = brem / per_window, r = brem - g * per_window;
    const int cols = min(wc, bands_x - g * wc);  // (the last window may be narrower)
    byi = r / cols;
    bxi = g * wc + (r - byi * cols);
}

// ---- the geometry table, written by a small kernel in front of the render kernel and read by the render kernel's waves through scalar loads
//      (no LDS table, no table builds between the planes, no v_readfirstlane):
//        recs[(band * D + plane) * NSB + sub-block] = uint4 { box origin address lo, hi | shape | gpart }          16 bytes
//        pl[(view * D + plane) * 2] = uint4 { zdiff, w/2, h/2, RN(2/w) }, uint4 { RN(2/h), -, -, - }               32 bytes per view and plane
//      shape = items per line (bits 0-4) | rows (5-8), and -- only when part of the box lies outside the texture (zeros padding: the loader
//      then takes the predicated form), marked by the sign bit -- the in-texture item columns [clo, clo + ncol) (bits 9-13, 14-18) and rows
//      [rlo, rlo + nrow) (19-22, 23-26); gpart = sub-block's LDS offset minus the box origin in LDS bytes (tap address = buffer + gpart +
//      iy0 * kRowBytes + ix0 * kES).  hdr[band] != 0: some box of the band does not fit its staging buffer -> the band takes the direct gather.
//      One workgroup per band: the header word is a workgroup reduction (no atomics, nothing to clear).  12.6 MB for BASELINE config 3.
constexpr int kPlU4 = 2;   // uint4 per (view, plane) record
__device__ __forceinline__ uint32_t shape_pack(int nq, int rows, int clo, int ncol, int rlo, int nrow, bool inside) {
    return inside ? static_cast<uint32_t>(nq | rows << 5)
                  : static_cast<uint32_t>(nq | rows << 5 | clo << 9 | ncol << 14 | rlo << 19 | nrow << 23) | 0x80000000u;
}
struct Shape { int nq, rows; uint32_t clo, ncol, rlo, nrow; };
__device__ __forceinline__ Shape shape_unpack(uint32_t w) {
    Shape h;
    h.nq = w & 31, h.rows = (w >> 5) & 15;
    const bool padded = (w >> 31) != 0;
    h.clo = padded ? (w >> 9) & 31 : 0u, h.ncol = padded ? (w >> 14) & 31 : static_cast<uint32_t>(h.nq);
    h.rlo = padded ? (w >> 19) & 15 : 0u, h.nrow = padded ? (w >> 23) & 15 : static_cast<uint32_t>(h.rows);
    return h;
}
template <typename TexT, bool AC>
__global__ __launch_bounds__(1024) void band_table_kernel(const KParams p, const int bands_x, const int bands_y, const int n_bands, const float cx, const float cy,
                                                         uint4* __restrict__ recs, uint4* __restrict__ pl, uint32_t* __restrict__ hdr, uint32_t* __restrict__ tickets) {
    using G = Geo<TexT>;
    constexpr int kES = G::kES, kTPI = G::kTPI, kCols = G::kCols, kMaxRows = G::kMaxRows, kRowBytes = G::kRowBytes, kSubBytes = G::kSubBytes;
    constexpr int NSB = G::NSB;
    static_assert(kCols < 32 && kMaxRows < 16, "shape_pack");
    const int band_id = blockIdx.x;
    int n, brem;
    band_to_view(p, band_id, bands_x * bands_y, n, brem);
    int bxi, byi;
    band_pos(brem, bands_x, bands_y, p.band_cols, bxi, byi);
    uint32_t ignore = 0;
    const int m = view_mpi(p, n, ignore);
    const int Ht = p.Ht, Wt = p.Wt, H = p.H, W = p.W;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const int64_t HW = static_cast<int64_t>(H) * W;
    const float* __restrict__ rdv = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    const int sx0 = min(bxi * (NSB * SBW), W - 1), sy0 = byi * SBH;
    const int by0p = min(sy0, H - 1), by1p = min(sy0 + SBH - 1, H - 1);
    bool unfit = false;
    for (int i = threadIdx.x; i < p.D * NSB; i += blockDim.x) {
        const int b = i % NSB, k = i / NSB;
        const float* __restrict__ dhw = p.dhw + (static_cast<int64_t>(m) * p.D + k) * 3;
        const float d = dhw[0], ph = dhw[1], pw = dhw[2];
        const float zdiff = d - ez;
        const float hw = pw * 0.5f, hh = ph * 0.5f;  // exact halves: (2x)/w == x/(w/2)
        const int bx0p = min(sx0 + b * SBW, W - 1), bx1p = min(sx0 + b * SBW + SBW - 1, W - 1);
        float mnx = __builtin_inff(), mxx = -__builtin_inff(), mny = mnx, mxy = mxx;
        bool finite = true;
        // (corner coordinates through v_rcp_f32 -- 4e-7 relative, i.e. below 0.007 texel inside the 16384-texel limit -- as in the strip
        //  kernel's box_of: the box only has to CONTAIN the taps, which the slack of 1/64 texel on every side covers; the pixels' own
        //  coordinates in the render kernel go through the exact chain)
        const float rw_a = __builtin_amdgcn_rcpf(hw), rh_a = __builtin_amdgcn_rcpf(hh);
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // the warp is a homography: the box of the sub-block's taps is spanned by its 4 corner pixels
            const int64_t q = static_cast<int64_t>((c & 2) ? by1p : by0p) * W + ((c & 1) ? bx1p : bx0p);
            float ix, iy, sc;
            const float crz = rdv[2 * HW + q];
            plane_coord_recip<AC>(zdiff, hw, hh, rw_a, rh_a, ex, ey, rdv[q], rdv[HW + q], crz, __builtin_amdgcn_rcpf(crz), cx, cy, ix, iy, sc);
            finite = finite && (fabsf(ix) < kCoordLimit) && (fabsf(iy) < kCoordLimit);  // false for NaN too
            mnx = fminf(mnx, ix), mxx = fmaxf(mxx, ix), mny = fminf(mny, iy), mxy = fmaxf(mxy, iy);
        }
        int qx0 = 0, by0 = 0, nq = -1, nrows = 0;
        if (finite) {
            const int bx0 = static_cast<int>(floorf(mnx - kBoxEps)), bx1 = static_cast<int>(floorf(mxx + kBoxEps)) + 1;
            by0 = static_cast<int>(floorf(mny - kBoxEps));
            const int by1 = static_cast<int>(floorf(mxy + kBoxEps)) + 1;
            qx0 = bx0 & ~(kTPI - 1);
            nq = (bx1 - qx0) / kTPI + 1;
            nrows = by1 - by0 + 1;
            if (nq > kCols || nrows > kMaxRows) nq = -1;
        }
        uint32_t shape = 0;
        if (nq > 0) {  // (qx0 and Wt are multiples of the item width)
            const int clo = min(max(-qx0 / kTPI, 0), nq), chi = min(max((Wt - qx0) / kTPI, 0), nq);
            const int rlo = min(max(-by0, 0), nrows), rhi = min(max(Ht - by0, 0), nrows);
            const bool inside = clo == 0 && chi == nq && rlo == 0 && rhi == nrows;
            shape = shape_pack(nq, nrows, clo, chi - clo, rlo, rhi - rlo, inside);
        } else {
            qx0 = 0, by0 = 0;
            unfit = true;
        }
        const uint64_t origin = reinterpret_cast<uint64_t>(vol + (static_cast<int64_t>(k) * p.s_plane + static_cast<int64_t>(by0) * p.s_row + qx0));
        const int gpart = b * kSubBytes - (by0 * kRowBytes + qx0 * kES);
        recs[(static_cast<int64_t>(band_id) * p.D + k) * NSB + b] =
            make_uint4(static_cast<uint32_t>(origin & 0xffffffffu), static_cast<uint32_t>((origin >> 32) & 0xffffu), shape, static_cast<uint32_t>(gpart));
        if (brem == 0 && b == 0) {  // the view's plane constants, once per view
            uint4* r = pl + (static_cast<int64_t>(n) * p.D + k) * kPlU4;
            r[0] = make_uint4(__float_as_uint(zdiff), __float_as_uint(hw), __float_as_uint(hh), __float_as_uint(1.0f / hw));
            r[1] = make_uint4(__float_as_uint(1.0f / hh), 0u, 0u, 0u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) tickets[threadIdx.x] = 0u;   // (the render kernel's ticket queues: one per XCD)
    const int any_unfit = __syncthreads_or(unfit ? 1 : 0);
    if (threadIdx.x == 0) {
        hdr[band_id] = static_cast<uint32_t>(any_unfit);
        if (any_unfit && p.gate != nullptr) {   // AUTO: the whole view goes to the tile kernel (gmpi_device.hpp) ...
            if (p.view_to_mpi == nullptr && p.views_per_mpi > 1) {   // ... and with it the views that share its MPI: the group stays together in ONE kernel, whose
                const int first = n / p.views_per_mpi * p.views_per_mpi, size = min(p.views_per_mpi, p.N - first);   // interleaved order reads the volume once for all of them
                for (int v = 0; v < size; ++v) p.gate[first + v] = p.gate_gen;
            } else {
                p.gate[n] = p.gate_gen;
            }
        }
    }
}

template <typename TexT, bool AC, bool STRICT, bool CHECK>
__global__ __launch_bounds__(Geo<TexT>::kThreads, Geo<TexT>::kWavesPerSimd) void render_band_kernel(const KParams p, const int bands_x, const int bands_y, const int n_bands, const float cx, const float cy,
                                                         const uint4* __restrict__ recs, const uint4* __restrict__ pl, const uint32_t* __restrict__ hdr, uint32_t* __restrict__ tickets) {
    using G = Geo<TexT>;
    constexpr int kES = G::kES, kTPI = G::kTPI, kCols = G::kCols, kNP = G::kNP;
    constexpr int NSB = G::NSB, PPT = G::PPT, WPS = G::WPS, kSubLanes = G::kSubLanes;
    constexpr int kLineBytes = G::kLineBytes, kRowBytes = G::kRowBytes, kSubBytes = G::kSubBytes, kBufBytes = G::kBufBytes;
    constexpr int kRPP = G::kRPP, kPassItems = G::kPassItems;
    constexpr bool BF = kES == 2;                              // 16-bit texels (bf16 or fp16): the two-pixel geometry, taps by ds_read_u16_d16_hi
    constexpr bool F16 = std::is_same<TexT, f16_t>::value;     // fp16: the loaded half is the value's fp16 pattern -- converted inside the FMA (v_fma_mix_f32)

    // `smem` is the ONLY __shared__ object of this kernel, i.e. it starts at LDS offset 0, and the staging buffers lead it: the [0,1] test below folds
    // every tap REGISTER, so every tap must read staged texels (or nothing).  A NaN coordinate saturates to LDS address 0 = the head of buffer 0
    // (texels: every box has >= 2 rows and >= 1 item, and buffer 0 holds a landed or landing plane at every step); an address past the allocation reads
    // zeros; addresses in between are in-box by the corner argument, which needs a pinhole ray field (include/gmpi_render.h).
    // tests/test_hip_band.py::test_band_range_check_has_no_false_alarm_on_nan_and_huge_rays.
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::kLdsBytes];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t tile_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte*)smem));  // the staging buffers lead the allocation (see coords(): a NaN coordinate reads THEM)

    // ---- blockIdx -> band.  XCD x = blockIdx % 8 gets a contiguous run of the bands of EVERY view (group of views that share an MPI): neighbours
    //      (column-major since round 5, see below) share halo rows in one L2, and the XCDs walk the views together (measured 4 % faster than one run of all bands per XCD,
    //      where different XCDs read different views at the same time) ----
    const int group_items = bands_x * bands_y * (p.view_to_mpi == nullptr ? p.views_per_mpi : 1);
    int band_id;
    {
        // The tail of the launch is handed out by TICKETS (round 6).  The dispatcher deals workgroups to the XCDs round-robin by index and nothing moves between XCDs
        // afterwards, so the launch ends when the slowest XCD -- the one whose regions cost most, or simply the slower silicon: even and odd XCDs differ by ~4 % in the
        // time a band takes on the boxes measured -- has worked off its share.  The last p.band_tail bands of every XCD's LAST run are therefore not tied to a block index:
        // the blocks that would have rendered them, plus as many EXTRA blocks per XCD appended to the grid, each draw a ticket -- from their own XCD's queue first (the
        // same bands in the same order as the static assignment), then from the others'.  An XCD that is done early finds its queue empty and takes what a late XCD has
        // not started yet; a block that finds every queue empty leaves.  Every band is drawn exactly once (one atomic per queue and block, for the tail blocks only).
        const int per_xcd = (group_items + 7) / 8, n_groups = (n_bands + group_items - 1) / group_items, static_blocks = per_xcd * 8 * n_groups;
        const int T = min(p.band_tail, per_xcd);
        const int jb = static_cast<int>(blockIdx.x) / 8, x = static_cast<int>(blockIdx.x) % 8;
        const bool ticketed = T > 0 && (static_cast<int>(blockIdx.x) >= static_blocks || (jb / per_xcd == n_groups - 1 && jb % per_xcd >= per_xcd - T));
        if (!ticketed) {
            band_id = blockIdx.x < static_cast<unsigned>(static_blocks) ? xcd_item_per_group(blockIdx.x, group_items, n_bands, p.band_rot, p.band_split) : n_bands;
        } else {
            int* word = reinterpret_cast<int*>(smem + 2 * kBufBytes);   // (the loader-offset area: written further down, behind a barrier)
            if (threadIdx.x == 0) {
                int got = n_bands;
                for (int i = 0; i < 8; ++i) {
                    const int q = (x + i) % 8;
                    const int t = static_cast<int>(atomicAdd(tickets + q, 1u));
                    if (t < T) {   // queue q, ticket t = the block (q, last group, position per_xcd - T + t) of the static assignment
                        got = xcd_item_per_group(q + 8 * ((n_groups - 1) * per_xcd + per_xcd - T + t), group_items, n_bands, p.band_rot, p.band_split);
                        break;
                    }
                }
                *word = got;
            }
            __syncthreads();
            band_id = __builtin_amdgcn_readfirstlane(*word);
            __syncthreads();   // (everybody has read the word before the loader offsets overwrite it)
        }
    }
#ifdef GMPI_TUNE  // (experiment: one contiguous run of ALL bands per XCD)
    if (p.flags & (1u << 19)) band_id = static_cast<int>(blockIdx.x % 8) * ((n_bands + 7) / 8) + static_cast<int>(blockIdx.x / 8);
#endif
    if (band_id >= n_bands) return;
    int n, brem;
    band_to_view(p, band_id, bands_x * bands_y, n, brem);
    if (view_gated_out(p, n)) return;  // (AUTO: a view with a box that does not fit is the tile kernel's)
    int bxi, byi;
    band_pos(brem, bands_x, bands_y, p.band_cols, bxi, byi);
#ifdef GMPI_TUNE  // GMPI_TUNE_WAVE + 16384: every workgroup leaves (start, end, XCC id) in status[64 + 3 blockIdx ...] (tools/kbench KB_STAMPS=1: who finishes when, per XCD)
    const bool stamp = (p.flags & (1u << 22)) != 0 && p.status != nullptr;
    uint64_t stamp_start = 0;
    if (stamp) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_start));   // (the 100 MHz reference clock: ONE epoch for the device; s_memtime counts per XCC / shader engine)
#endif

    const int tid = threadIdx.x;
    // Registers are the scarce resource (64 per lane for 8 waves per SIMD): values that only depend on the thread index are
    // recomputed from a laundered copy of it where they are needed, so that they do not live through the plane loop.
    auto fresh_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sb = wave / WPS, wj = wave % WPS;  // sub-block of this wave, its first pixel row in the sub-block
    const int lane = tid & 63;
    // status bits: kept wave-uniform (a scalar register) until the epilogue -- a per-lane word would cost a VGPR through the plane loop
    uint32_t bad_w = 0;
    const int m = view_mpi(p, n, bad_w);
    const int D = p.D, Ht = p.Ht, Wt = p.Wt, H = p.H, W = p.W;
    const float* __restrict__ dhw = p.dhw + static_cast<int64_t>(m) * D * 3;
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const float zx = p.z_dir[3 * n + 0], zy = p.z_dir[3 * n + 1], zz = p.z_dir[3 * n + 2];
    // cx, cy: (Wt - 1) / 2, (Ht - 1) / 2 (align_corners) or Wt, Ht -- kernel arguments, i.e. scalar registers: computed here they would sit in
    // a VGPR each through the plane loop
    constexpr bool check_range = CHECK;  // GMPI_FLAG_CHECK_RANGE (a template parameter: the test would sit in the plane loop)
    const bool check_last = (p.flags & (1u << 2)) != 0;
    const int64_t HW = static_cast<int64_t>(H) * W;
    const float* __restrict__ rdv = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
    const TexT* __restrict__ vol = static_cast<const TexT*>(p.rgba) + static_cast<int64_t>(m) * p.s_mpi;
    const int64_t s_chan = p.s_chan, s_row = p.s_row, s_plane = p.s_plane;

    if (p.status != nullptr && brem == 0 && tid == 0) {  // mpi.py:70-72, once per view
        const float ez0 = p.eye_pos[2];
        bool behind = false;
        for (int k = 0; k < D; ++k) behind |= !(dhw[3 * k] >= ez0);
        if (behind) atomicOr(p.status, 4u);
    }

    // ---- this thread's pixels: column px, rows py0 + WPS q (out-of-image pixels shadow the last row / column) ------------
    const int px = bxi * (NSB * SBW) + sb * SBW + lane;
    const int py0 = byi * SBH + wj;
    const int pxc = min(px, W - 1);
    float rx[PPT], ry[PPT], rz[PPT], rcp_rz[PPT];
    Accum A[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int64_t pix = static_cast<int64_t>(min(py0 + WPS * q, H - 1)) * W + pxc;
        rx[q] = rdv[pix], ry[q] = rdv[HW + pix], rz[q] = rdv[2 * HW + pix];
        rcp_rz[q] = 1.0f / rz[q];  // correctly rounded; hoisted out of the plane loop (div_by_recip)
    }
    auto ray_dot = [&](int q) {  // einsum("nchw,nc->nhw") mpi.py:149
        float dot = rx[q] * zx;
        dot = dot + ry[q] * zy;
        dot = dot + rz[q] * zz;
        return dot;
    };

    // ---- this thread's loader items: item kPassItems r + (tid % kSubLanes) of its sub-block's box -> (texel row, channel, item column) ----
    auto loader_pos = [&](int& l_col, int& l_line, int& l_row, bool& l_on) {  // pass 0: line = 4 row + channel
        const int t = fresh_tid() & (kSubLanes - 1);
        l_line = t / kCols, l_col = t - l_line * kCols, l_row = l_line >> 2, l_on = t < kPassItems;
    };
    {  // byte offset of this lane's pass-0 item from the box origin: parked in LDS, read back behind every plane's barrier
        int l_col, l_line, l_row;
        bool l_on;
        loader_pos(l_col, l_line, l_row, l_on);
        reinterpret_cast<uint32_t*>(smem + 2 * kBufBytes)[tid] = static_cast<uint32_t>(l_row * s_row + (l_line & 3) * s_chan + kTPI * l_col) * static_cast<uint32_t>(kES);
    }
    const uint32_t pass_off = static_cast<uint32_t>(kRPP * s_row) * static_cast<uint32_t>(kES), pass_off2 = 2 * pass_off;           // per pass
    const uint32_t pass_off3 = 3 * pass_off, pass_off4 = 4 * pass_off;  // (kNP == 5 only)
    const uint32_t sub_base = tile_base + static_cast<uint32_t>(sb * kSubBytes);
    const uint32_t wave_dst = sub_base + static_cast<uint32_t>(wj) * 1024u;  // pass r: + r * kPassItems * 16

#ifdef GMPI_TUNE
    const bool abl_noload = (p.flags & (1u << 16)) != 0, abl_nocomp = (p.flags & (1u << 17)) != 0, abl_noissue = (p.flags & (1u << 18)) != 0;
    // round 5, the "replay" of the plane step: bit 20 = no workgroup barrier, bit 21 = the records of the first planes stay (no scalar
    // loads in the loop) (KB_NOCHECK=1 in kbench drops the range check).  With 16 + 18 + 20 + 21 a wave runs the exact instruction stream of its pixels -- chain,
    // counted waits, 32 two-byte taps, bilinear, blend, the range check's fold -- and nothing that ties it to the other fifteen waves.
    const bool abl_nobar = (p.flags & (1u << 20)) != 0, abl_norec = (p.flags & (1u << 21)) != 0;
#else
    constexpr bool abl_noload = false, abl_nocomp = false, abl_noissue = false, abl_nobar = false, abl_norec = false;
#endif

#ifdef GMPI_PROF  // per-phase shader-clock totals of one wave (status words 8..): barrier | LDS burst | range check | DMA issue | - | pixel 0 | pixel 1
    uint32_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t prof_last, prof_start;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_start));
    prof_last = prof_start;
#define GMPI_STAMP(i) do { uint64_t now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); prof_acc[i] += static_cast<uint32_t>(now_ - prof_last); prof_last = now_; } while (0)
#else
#define GMPI_STAMP(i) do { } while (0)
#endif

    // this wave's records: recs[(band * D + t) * NSB + sb], pl[(n * D + t) * 2 + part]
    const uint4* __restrict__ myrec = recs + (static_cast<int64_t>(band_id) * D * NSB + sb);
    const uint4* __restrict__ mypl = pl + static_cast<int64_t>(n) * D * kPlU4;
    constexpr int kRecStep = NSB;  // uint4 from plane t to plane t + 1

    if (hdr[band_id] != 0) {
        // ---- last resort (a box does not fit: tilted camera, texture much finer than the image, degenerate rays): direct gather, same arithmetic ----
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const float dot = ray_dot(q);
            for (int t = 0; t < D; ++t) {
                const float d = dhw[3 * t + 0], ph = dhw[3 * t + 1], pw = dhw[3 * t + 2];
                float ix, iy, s, u, v;
                plane_coord<AC>(d - ez, ph, pw, ex, ey, rx[q], ry[q], rz[q], cx, cy, ix, iy, s, u, v);
                float smp[4];
                uint32_t gbad = 0;
                gather_sample<TexT, STRICT>(vol + static_cast<int64_t>(t) * s_plane, s_chan, s_row, Ht, Wt, ix, iy, check_range, gbad, smp);
                if (check_range && __any(gbad != 0)) bad_w |= 2u;
                blend<STRICT>(A[q], smp[0], smp[1], smp[2], smp[3], s, dot);
            }
        }
    } else {
        float dots[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) dots[q] = STRICT ? ray_dot(q) : 0.0f;  // (default mode applies the dot product once, at the end)
        // ---- per wave: exec masks / pass count of the box of the plane last issued (recomputed when the box shape changes: a handful of
        //      times per band).  (`three` of plane t is latched before plane t + 1 is issued: the range check of plane t reads it.) ----
        uint64_t m_cur[kNP];
        int dims_cur = 0;
        bool three = false;  // the box of the plane last issued needs the LAST pass (the kNP-th: the third as shipped)
        int np_cur = kNP - 1;  // kNP == 5: its number of passes (3, 4 or 5)
#pragma unroll
        for (int r = 0; r < kNP; ++r) m_cur[r] = 0;

        auto issue = [&](const uint4& rl, uint32_t g_off, auto ub) {  // DMA of the plane with record part L = rl into buffer U (this wave's part of its sub-block's box)
            constexpr int U = decltype(ub)::value;
            const int dims = static_cast<int>(rl.z);
            // raw buffer, num_records 2^31: only the explicit offset below is rejected.  (readfirstlane: the words are wave-uniform and the asm
            // operand MUST be scalar -- under scalar register pressure the compiler has been seen to keep them in vector registers and to print
            // those into the "s" operand; the intrinsic is free where they already are scalar)
            const u32x4 rsrc = {static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rl.x))),
                                static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rl.y))), 0x80000000u, 0x00020000u};
            if (dims >= 0) {  // the box lies inside the texture: lanes of the box load, the others are switched off
                if (dims != dims_cur) {
                    const int nq = dims & 31, rows = dims >> 5;
                    dims_cur = dims;
                    three = rows > (kNP - 1) * kRPP;
                    np_cur = max((rows + kRPP - 1) / kRPP, 3);
                    int l_col, l_line, l_row;
                    bool l_on;
                    loader_pos(l_col, l_line, l_row, l_on);
#pragma unroll
                    for (int r = 0; r < kNP; ++r) {
                        m_cur[r] = __ballot(l_on && l_col < nq && l_row + r * kRPP < rows);
                        if (abl_noload) m_cur[r] = 0;
                    }
                }
                if constexpr (kNP == 3) {
                    if (three) dma16x3<U * kBufBytes, kPassItems * 16>(g_off, pass_off, pass_off2, rsrc, wave_dst, m_cur[0], m_cur[1], m_cur[2]);
                    else dma16x2<U * kBufBytes, kPassItems * 16>(g_off, pass_off, rsrc, wave_dst, m_cur[0], m_cur[1]);
                } else {  // (a pass whose mask is empty still takes its turn on the CU's vector-memory issue path: issue what the box needs)
                    if (np_cur == 5) dma16x5<U * kBufBytes, kPassItems * 16>(g_off, pass_off, pass_off2, pass_off3, pass_off4, rsrc, wave_dst, m_cur[0], m_cur[1], m_cur[2], m_cur[3], m_cur[4]);
                    else if (np_cur == 4) dma16x4<U * kBufBytes, kPassItems * 16>(g_off, pass_off, pass_off2, pass_off3, rsrc, wave_dst, m_cur[0], m_cur[1], m_cur[2], m_cur[3]);
                    else dma16x3<U * kBufBytes, kPassItems * 16>(g_off, pass_off, pass_off2, rsrc, wave_dst, m_cur[0], m_cur[1], m_cur[2]);
                }
            } else {  // zeros padding: every lane of the box rows is active, lanes outside the texture get the out-of-range offset
                const Shape h = shape_unpack(static_cast<uint32_t>(dims));
                const int rows = h.rows;
                const uint32_t clo = h.clo, ncol = h.ncol, llo = 4 * h.rlo, nline = 4 * h.nrow;
                dims_cur = dims;
                three = rows > (kNP - 1) * kRPP;
                const int npk = (rows + kRPP - 1) / kRPP;
                np_cur = max(npk, 3);
                int l_col, l_line, l_row;
                bool l_on;
                loader_pos(l_col, l_line, l_row, l_on);
#pragma unroll
                for (int r = 0; r < kNP; ++r) {
                    m_cur[r] = __ballot(l_on && l_row + r * kRPP < rows);
                    if (abl_noload) m_cur[r] = 0;
                    if (r < npk) {
                        const bool ok = (static_cast<uint32_t>(l_col) - clo < ncol) & (static_cast<uint32_t>(l_line + 4 * r * kRPP) - llo < nline);
                        dma16(ok ? g_off : 0x80000000u, r * pass_off, rsrc, wave_dst + static_cast<uint32_t>(U * kBufBytes + r * (kPassItems * 16)), m_cur[r]);
                    }
                }
            }
        };

        // ---- [0,1] test (mpi.py:185-187) on the TAPS: every texel the render samples passes through a tap register, and a tap register orders
        //      like its value -- a 16-bit texel sits in the high half of a zeroed register (d16_hi), a 32-bit texel fills it; non-negative floats
        //      order like unsigned integers, the sign bit and NaN / Inf compare above 1.0 -- so the test is a running unsigned maximum over the
        //      tap registers: one v_max3_u32 per two taps, no compare, no branch in the plane loop; the verdict is taken once per band (below).
        //      (Rounds 3-4 read every staged ITEM back from LDS instead -- 2 ds_read_b128 + 8 half-rate v_pk_max_u16 per lane and plane in
        //      front of the pixels; round 5 found the launch running at the package power limit (1.39 of 1.40 kW, engine clock 2.02 of 2.4 GHz),
        //      and dropping the read-back is worth 2.7 % on bf16 volumes, 1-2.5 % on fp32 ones: profiles/r05_power.txt.)
        uint32_t chk_acc = 0;
        auto tap_fold8 = [&](const uint32_t (&q)[8]) {
            asm volatile("v_max3_u32 %0, %0, %1, %2\n\tv_max3_u32 %0, %0, %3, %4\n\tv_max3_u32 %0, %0, %5, %6\n\tv_max3_u32 %0, %0, %7, %8"
                         : "+v"(chk_acc) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
        };

        // ---- one pixel and plane, in three steps so that the taps of pixel q fly while the chain of pixel q + 1 issues ------------
        // a tap register -> fp32.  bf16: the d16_hi load put the texel into the high half of a zeroed register, which IS its fp32 value.  fp16
        // (round 4): the high half holds the fp16 pattern; the exact conversion folds into the bilinear FMAs (v_fma_mix_f32 with op_sel: no
        // instruction of its own in default mode; strict-order mode converts first, one v_cvt_f32_f16 per tap).  fp32: the register is the value.
        auto tapf = [&](uint32_t t) -> float {
            if constexpr (F16) {
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                return static_cast<float>(__builtin_bit_cast(h2_t, t).y);
            } else {
                return __uint_as_float(t);
            }
        };
        struct Coords { float s, nw, ne, sw, se; uint32_t a_tap; };
        auto coords = [&](int q, const float4& rf, const float2& rg, Coords& c) {
            float ix, iy, fx, fy;
            if (STRICT) {
                float u, v;
                plane_coord<AC>(rf.x, rf.z + rf.z, rf.y + rf.y, ex, ey, rx[q], ry[q], rz[q], cx, cy, ix, iy, c.s, u, v);
                fx = floorf(ix), fy = floorf(iy);
                const float fx1 = fx + 1.0f, fy1 = fy + 1.0f;
                const float wx1 = ix - fx, wx0 = fx1 - ix, wy1 = iy - fy, wy0 = fy1 - iy;
                c.nw = wx0 * wy0, c.ne = wx1 * wy0, c.sw = wx0 * wy1, c.se = wx1 * wy1;
            } else {
                plane_coord_recip<AC>(rf.x, rf.y, rf.z, rf.w, rg.x, ex, ey, rx[q], ry[q], rz[q], rcp_rz[q], cx, cy, ix, iy, c.s);
                fx = floorf(ix), fy = floorf(iy);
                // ATen's vectorised CPU form of the weights: w1 = ix - floor(ix), w0 = 1 - w1
                const float wx1 = ix - fx, wy1 = iy - fy;
                const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
                c.nw = wx0 * wy0, c.ne = wx1 * wy0, c.sw = wx0 * wy1, c.se = wx1 * wy1;
            }
            // LDS byte address of the north-west tap of channel 0: two exact fp32 FMAs and one saturating conversion: the box contains every tap of the
            // sub-block (corner argument).  The conversion is the SIGNED one (round 6): NaN -> 0 (the head of the staging buffers, i.e. texels), +inf /
            // huge -> 0x7fffffff and -inf -> 0x80000000, both far past the allocation even with the taps' immediate offsets added (such reads return
            // zeros) -- the unsigned conversion saturated +inf to 0xffffffff, which the offsets wrapped around to the bottom of LDS: an infinite ray
            // component read whatever lay there, and the range check below, which folds every tap register, raised a false "alpha out of [0, 1]"
            // (tests/test_hip_band.py::test_band_range_check_has_no_false_alarm_on_nan_and_huge_rays).  In-range addresses (< 2^24) are the same.
            const float af = __builtin_fmaf(fy, static_cast<float>(kRowBytes), __builtin_fmaf(fx, static_cast<float>(kES), rg.y));
            c.a_tap = static_cast<uint32_t>(static_cast<int32_t>(af));
        };
        // The 16 taps of a pixel are fetched as two halves (channels R, G | B, A): 8 tap registers instead of 16 -- what lets two pixels per
        // thread live in 64 VGPRs without a spill reload in the plane loop (a scratch load shares vmcnt with the DMA: it would drain it).
        auto taps = [&](auto hb, uint32_t a_tap, uint32_t (&q)[8]) {  // the 8 taps of two channels, landed
            constexpr int C0 = 2 * decltype(hb)::value;
            if constexpr (BF) {
                asm volatile("ds_read_u16_d16_hi %0, %8 offset:%9\n\tds_read_u16_d16_hi %1, %8 offset:%10\n\tds_read_u16_d16_hi %2, %8 offset:%11\n\tds_read_u16_d16_hi %3, %8 offset:%12\n\t"
                             "ds_read_u16_d16_hi %4, %8 offset:%13\n\tds_read_u16_d16_hi %5, %8 offset:%14\n\tds_read_u16_d16_hi %6, %8 offset:%15\n\tds_read_u16_d16_hi %7, %8 offset:%16\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                             : "v"(a_tap), "i"(C0 * kLineBytes), "i"(C0 * kLineBytes + 2), "i"(C0 * kLineBytes + kRowBytes), "i"(C0 * kLineBytes + kRowBytes + 2),
                               "i"((C0 + 1) * kLineBytes), "i"((C0 + 1) * kLineBytes + 2), "i"((C0 + 1) * kLineBytes + kRowBytes), "i"((C0 + 1) * kLineBytes + kRowBytes + 2));
            } else {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                u32x2 v0, v1, v2, v3;
                const uint32_t a_bot = a_tap + kRowBytes;
                asm volatile("ds_read2_b32 %0, %4 offset0:%6 offset1:%7\n\tds_read2_b32 %1, %5 offset0:%6 offset1:%7\n\t"
                             "ds_read2_b32 %2, %4 offset0:%8 offset1:%9\n\tds_read2_b32 %3, %5 offset0:%8 offset1:%9\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                             : "v"(a_tap), "v"(a_bot), "i"(C0 * (kLineBytes / 4)), "i"(C0 * (kLineBytes / 4) + 1), "i"((C0 + 1) * (kLineBytes / 4)), "i"((C0 + 1) * (kLineBytes / 4) + 1));
                q[0] = v0.x, q[1] = v0.y, q[2] = v1.x, q[3] = v1.y, q[4] = v2.x, q[5] = v2.y, q[6] = v3.x, q[7] = v3.y;
            }
        };
        // Software pipeline of the bf16 default-mode plane step: a batch of 8 taps is ISSUED one batch ahead of the wait that lands it, so the
        // LDS round trips overlap the wave's own FMAs instead of relying on the other waves of the SIMD.  LDS operations return in order:
        // with B younger LDS operations issued behind the wanted ones, `lgkmcnt(B)` lands the wanted ones.  Issue and wait are separate asm
        // statements with the tap registers as operands of both; tools/isa_pipe.py checks that the build has no copy of a tap register
        // between them (the compiler may copy an asm output as soon as its statement ends).
        auto taps_issue = [&](auto hb, uint32_t a_tap, uint32_t (&q)[8]) {
            constexpr int C0 = 2 * decltype(hb)::value;
            asm volatile("ds_read_u16_d16_hi %0, %8 offset:%9\n\tds_read_u16_d16_hi %1, %8 offset:%10\n\tds_read_u16_d16_hi %2, %8 offset:%11\n\tds_read_u16_d16_hi %3, %8 offset:%12\n\t"
                         "ds_read_u16_d16_hi %4, %8 offset:%13\n\tds_read_u16_d16_hi %5, %8 offset:%14\n\tds_read_u16_d16_hi %6, %8 offset:%15\n\tds_read_u16_d16_hi %7, %8 offset:%16"
                         : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                         : "v"(a_tap), "i"(C0 * kLineBytes), "i"(C0 * kLineBytes + 2), "i"(C0 * kLineBytes + kRowBytes), "i"(C0 * kLineBytes + kRowBytes + 2),
                           "i"((C0 + 1) * kLineBytes), "i"((C0 + 1) * kLineBytes + 2), "i"((C0 + 1) * kLineBytes + kRowBytes), "i"((C0 + 1) * kLineBytes + kRowBytes + 2));
        };
        auto taps_land = [&](auto nb, uint32_t (&q)[8]) {  // wait until at most N LDS operations (all younger than q's) are outstanding
            constexpr int N = decltype(nb)::value;
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "i"(N));
        };
        // (the same for fp32 volumes: a batch = the 8 taps of two channels as four ds_read2_b32)
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        auto taps32_issue = [&](auto hb, uint32_t a_tap, uint32_t a_bot, u32x2_t (&v)[4]) {
            constexpr int C0 = 2 * decltype(hb)::value;
            asm volatile("ds_read2_b32 %0, %4 offset0:%6 offset1:%7\n\tds_read2_b32 %1, %5 offset0:%6 offset1:%7\n\t"
                         "ds_read2_b32 %2, %4 offset0:%8 offset1:%9\n\tds_read2_b32 %3, %5 offset0:%8 offset1:%9"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                         : "v"(a_tap), "v"(a_bot), "i"(C0 * (kLineBytes / 4)), "i"(C0 * (kLineBytes / 4) + 1), "i"((C0 + 1) * (kLineBytes / 4)), "i"((C0 + 1) * (kLineBytes / 4) + 1));
        };
        auto taps32_land = [&](auto nb, u32x2_t (&v)[4]) {
            constexpr int N = decltype(nb)::value;
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "i"(N));
        };
        auto pixel = [&](int q, const float4& rf, const float2& rg) {
            Coords c;
            coords(q, rf, rg, c);
            Footprint f;
            f.nw = c.nw, f.ne = c.ne, f.sw = c.sw, f.se = c.se;
            uint32_t t[8];
            float smp[4];
            taps(ic<0>{}, c.a_tap, t);
            if (check_range) tap_fold8(t);
            smp[0] = bilerp<STRICT>(tapf(t[0]), tapf(t[1]), tapf(t[2]), tapf(t[3]), f);
            smp[1] = bilerp<STRICT>(tapf(t[4]), tapf(t[5]), tapf(t[6]), tapf(t[7]), f);
            taps(ic<1>{}, c.a_tap, t);
            if (check_range) tap_fold8(t);
            smp[2] = bilerp<STRICT>(tapf(t[0]), tapf(t[1]), tapf(t[2]), tapf(t[3]), f);
            smp[3] = bilerp<STRICT>(tapf(t[4]), tapf(t[5]), tapf(t[6]), tapf(t[7]), f);
            blend<STRICT>(A[q], smp[0], smp[1], smp[2], smp[3], c.s, dots[q]);
        };


        // One plane step: barrier | DMA of plane t + 1 | the pixels (their landed taps folded into the range check) | scalar
        // loads of the next step's records (box record of plane t + 2, plane constants of plane t + 1).
        // (Measured and not kept -- profiles/r03_band_variants.txt, r04_band_variants.txt: the loader offset fetched before the barrier;
        //  s_setprio around the DMA issue or by wave index; the record loads at the top of the step (+5 %: see the head of this file); the range
        //  check's former LDS read-back at the end of the step (-1.4 %; round 5 dropped the read-back); non-temporal DMA (+5 %); the DMA passes spread over the step.)
        uint4 Ln, Fc;      // wave-uniform: scalar registers
        uint32_t rhh_c, gp_c;
        uint32_t g_off = 0;  // this lane's loader offset: a vector register through the plane loop (round 3 re-read it from LDS behind every barrier)
#ifdef GMPI_PROF  // (the phase stamps wait for lgkmcnt(0): they would serialise the pipeline they are meant to time)
        constexpr bool piped = false, pipedG = false;
#else
        constexpr bool piped = BF && !STRICT && PPT == 2 && kNP == 3;
        constexpr bool pipedG = !STRICT && !piped;     // every other default-mode geometry (fp32 volumes as shipped: 2 pixels per thread, 5 DMA passes; any GMPI_BAND_PPT*): the same pipeline, written once for any pixel count
#endif
        auto stage = [&](int tt, auto ub) {  // plane tt, held by buffer U
            constexpr int U = decltype(ub)::value;
            if (!abl_nobar) wg_barrier();  // own DMA of plane tt has landed -> everybody's has; everybody is done reading plane tt - 1
            GMPI_STAMP(0);
            if (tt + 1 < D && !abl_noissue) issue(Ln, g_off, ic<1 - U>{});
            GMPI_STAMP(3);
            const float4 rf = make_float4(__uint_as_float(Fc.x), __uint_as_float(Fc.y), __uint_as_float(Fc.z), __uint_as_float(Fc.w));
            // tap address constant of this plane and buffer (an integer below 2^24, exact in fp32)
            const float2 rg = make_float2(__uint_as_float(rhh_c), static_cast<float>(static_cast<int>(gp_c) + static_cast<int>(tile_base) + U * kBufBytes));
            if constexpr (piped) {
                if (!abl_nocomp) {
                    // LDS operations of the step, in issue order:  b0 b1 (pixel 0: channels R G | B A) | b2 b3 (pixel 1); every wait but the last
                    // leaves the 8 taps of the batch behind it in flight.  A landed batch is folded into the range check's maximum (tap_fold8).
                    uint32_t ta[8], tb[8];
                    Coords p0, p1;
                    Footprint f;
                    float smp[4];
                    coords(0, rf, rg, p0);
                    taps_issue(ic<0>{}, p0.a_tap, ta);                                        // b0
                    taps_issue(ic<1>{}, p0.a_tap, tb);                                        // b1
                    coords(1, rf, rg, p1);                                                   // (the chain of pixel 1 issues while b0, b1 fly)
                    asm volatile("" : "+v"(p1.s), "+v"(p1.nw), "+v"(p1.ne), "+v"(p1.sw), "+v"(p1.se), "+v"(p1.a_tap));
                    taps_land(ic<8>{}, ta);
                    if (check_range) tap_fold8(ta);
                    f.nw = p0.nw, f.ne = p0.ne, f.sw = p0.sw, f.se = p0.se;
                    smp[0] = bilerp<false>(tapf(ta[0]), tapf(ta[1]), tapf(ta[2]), tapf(ta[3]), f);
                    smp[1] = bilerp<false>(tapf(ta[4]), tapf(ta[5]), tapf(ta[6]), tapf(ta[7]), f);
                    asm volatile("" : "+v"(smp[0]), "+v"(smp[1]));
                    taps_issue(ic<0>{}, p1.a_tap, ta);                                        // b2
                    taps_land(ic<8>{}, tb);
                    if (check_range) tap_fold8(tb);
                    smp[2] = bilerp<false>(tapf(tb[0]), tapf(tb[1]), tapf(tb[2]), tapf(tb[3]), f);
                    smp[3] = bilerp<false>(tapf(tb[4]), tapf(tb[5]), tapf(tb[6]), tapf(tb[7]), f);
                    blend<false>(A[0], smp[0], smp[1], smp[2], smp[3], p0.s, dots[0]);
                    asm volatile("" : "+v"(A[0].T), "+v"(A[0].r), "+v"(A[0].g), "+v"(A[0].b), "+v"(A[0].z));
                    taps_issue(ic<1>{}, p1.a_tap, tb);                                        // b3
                    taps_land(ic<8>{}, ta);
                    if (check_range) tap_fold8(ta);
                    f.nw = p1.nw, f.ne = p1.ne, f.sw = p1.sw, f.se = p1.se;
                    smp[0] = bilerp<false>(tapf(ta[0]), tapf(ta[1]), tapf(ta[2]), tapf(ta[3]), f);
                    smp[1] = bilerp<false>(tapf(ta[4]), tapf(ta[5]), tapf(ta[6]), tapf(ta[7]), f);
                    asm volatile("" : "+v"(smp[0]), "+v"(smp[1]));  // (pins the two samples in front of the last wait)
                    taps_land(ic<0>{}, tb);
                    if (check_range) tap_fold8(tb);
                    smp[2] = bilerp<false>(tapf(tb[0]), tapf(tb[1]), tapf(tb[2]), tapf(tb[3]), f);
                    smp[3] = bilerp<false>(tapf(tb[4]), tapf(tb[5]), tapf(tb[6]), tapf(tb[7]), f);
                    blend<false>(A[1], smp[0], smp[1], smp[2], smp[3], p1.s, dots[1]);
                }
            } else if constexpr (pipedG) {
                // any PPT: A0 B0 | A1 B1 | ... (A = channels R G of a pixel, B = B A); every wait but the last leaves one batch in flight
                constexpr int NB = BF ? 8 : 4;  // LDS operations per batch
                if (!abl_nocomp) {
                    uint32_t ta[8], tb[8];
                    u32x2_t va[4], vb[4];
                    Coords pc, pn;
                    Footprint f;
                    float smp[4];
                    auto issueA = [&](const Coords& c) {
                        if constexpr (BF) taps_issue(ic<0>{}, c.a_tap, ta);
                        else taps32_issue(ic<0>{}, c.a_tap, c.a_tap + kRowBytes, va);
                    };
                    auto issueB = [&](const Coords& c) {
                        if constexpr (BF) taps_issue(ic<1>{}, c.a_tap, tb);
                        else taps32_issue(ic<1>{}, c.a_tap, c.a_tap + kRowBytes, vb);
                    };
                    auto landA = [&](auto nb) {
                        if constexpr (BF) {
                            taps_land(nb, ta);
                            if (check_range) tap_fold8(ta);
                            smp[0] = bilerp<false>(tapf(ta[0]), tapf(ta[1]), tapf(ta[2]), tapf(ta[3]), f);
                            smp[1] = bilerp<false>(tapf(ta[4]), tapf(ta[5]), tapf(ta[6]), tapf(ta[7]), f);
                        } else {
                            taps32_land(nb, va);
                            if (check_range) { const uint32_t q8[8] = {va[0].x, va[0].y, va[1].x, va[1].y, va[2].x, va[2].y, va[3].x, va[3].y}; tap_fold8(q8); }
                            smp[0] = bilerp<false>(__uint_as_float(va[0].x), __uint_as_float(va[0].y), __uint_as_float(va[1].x), __uint_as_float(va[1].y), f);
                            smp[1] = bilerp<false>(__uint_as_float(va[2].x), __uint_as_float(va[2].y), __uint_as_float(va[3].x), __uint_as_float(va[3].y), f);
                        }
                        asm volatile("" : "+v"(smp[0]), "+v"(smp[1]));
                    };
                    auto landB = [&](auto nb) {
                        if constexpr (BF) {
                            taps_land(nb, tb);
                            if (check_range) tap_fold8(tb);
                            smp[2] = bilerp<false>(tapf(tb[0]), tapf(tb[1]), tapf(tb[2]), tapf(tb[3]), f);
                            smp[3] = bilerp<false>(tapf(tb[4]), tapf(tb[5]), tapf(tb[6]), tapf(tb[7]), f);
                        } else {
                            taps32_land(nb, vb);
                            if (check_range) { const uint32_t q8[8] = {vb[0].x, vb[0].y, vb[1].x, vb[1].y, vb[2].x, vb[2].y, vb[3].x, vb[3].y}; tap_fold8(q8); }
                            smp[2] = bilerp<false>(__uint_as_float(vb[0].x), __uint_as_float(vb[0].y), __uint_as_float(vb[1].x), __uint_as_float(vb[1].y), f);
                            smp[3] = bilerp<false>(__uint_as_float(vb[2].x), __uint_as_float(vb[2].y), __uint_as_float(vb[3].x), __uint_as_float(vb[3].y), f);
                        }
                    };
                    auto px_step = [&](auto qc) {
                        constexpr int Q = decltype(qc)::value;
                        constexpr bool more = Q + 1 < PPT;
                        if constexpr (more) {
                            coords(Q + 1, rf, rg, pn);  // (the chain of the next pixel issues while this pixel's batches fly)
                            asm volatile("" : "+v"(pn.s), "+v"(pn.nw), "+v"(pn.ne), "+v"(pn.sw), "+v"(pn.se), "+v"(pn.a_tap));
                        }
                        f.nw = pc.nw, f.ne = pc.ne, f.sw = pc.sw, f.se = pc.se;
                        landA(ic<NB>{});
                        if constexpr (more) issueA(pn);
                        landB(ic<(more ? NB : 0)>{});
                        blend<false>(A[Q], smp[0], smp[1], smp[2], smp[3], pc.s, dots[Q]);
                        if constexpr (more) {
                            asm volatile("" : "+v"(A[Q].T), "+v"(A[Q].r), "+v"(A[Q].g), "+v"(A[Q].b), "+v"(A[Q].z));
                            issueB(pn);
                            pc = pn;
                        }
                    };
                    coords(0, rf, rg, pc);
                    issueA(pc);
                    issueB(pc);
                    px_step(ic<0>{});
                    if constexpr (PPT > 1) px_step(ic<1>{});
                    if constexpr (PPT > 2) px_step(ic<2>{});
                    if constexpr (PPT > 3) px_step(ic<3>{});
                    static_assert(PPT <= 4, "pixel slots");
                }
            } else {
                GMPI_STAMP(2);
                if (!abl_nocomp) {
#pragma unroll
                    for (int q = 0; q < PPT; ++q) {
                        pixel(q, rf, rg);  // (the LDS round trips of the taps are covered by the other waves of the SIMD: 8 are resident)
                        GMPI_STAMP(5 + q);
                    }
                }
            }
            // The records of the next step (both tables are padded by two planes of records: no bounds tests), BEHIND the last pixel: the waves
            // that get here first pull the lines into the scalar cache, the workgroup's last wave -- the one the barrier waits for -- hits.
            if (!abl_norec) {
                gp_c = Ln.w;  // (Ln is still the record of plane tt + 1)
                Ln = myrec[static_cast<int64_t>(tt + 2) * kRecStep];
                Fc = mypl[(tt + 1) * kPlU4], rhh_c = mypl[(tt + 1) * kPlU4 + 1].x;
            }
            static_assert(PPT <= 4 && kNP <= 5, "pixel slots / check passes");
        };
        // (the per-pixel state must sit in registers through the plane loop: a reload there is a vector memory operation on the DMA's counter)
#pragma unroll
        for (int q = 0; q < PPT; ++q)
            asm volatile("" : "+v"(rx[q]), "+v"(ry[q]), "+v"(rz[q]), "+v"(rcp_rz[q]), "+v"(A[q].T), "+v"(A[q].r), "+v"(A[q].g), "+v"(A[q].b), "+v"(A[q].z));
        __syncthreads();  // (the loader offsets are in place)
        g_off = reinterpret_cast<const uint32_t*>(smem + 2 * kBufBytes)[fresh_tid()];
        issue(myrec[0], g_off, ic<0>{});  // plane 0
        Fc = mypl[0], rhh_c = mypl[1].x, gp_c = myrec[0].w, Ln = myrec[kRecStep];
        for (int t = 0; t < D; t += 2) {
            stage(t, ic<0>{});
            if (t + 1 < D) stage(t + 1, ic<1>{});
        }
        // ---- the verdict of the range check.  -0.0 is a legal value whose pattern sits above that of 1.0: a maximum of exactly that pattern
        //      says nothing about the values below it, so such a band re-tests its texels one by one (cold: never for generator output) ----
        if (check_range) {
            constexpr uint32_t kOne = F16 ? 0x3c00u : BF ? 0x3f80u : 0x3f800000u, kNegZero = BF ? 0x8000u : 0x80000000u;  // (the pattern of 1.0 / -0.0 in the storage type)
            const uint32_t mx = BF ? chk_acc >> 16 : chk_acc;   // (a 16-bit tap's register: the texel in the high half, zeros below)
#ifdef GMPI_TUNE
            if (p.status != nullptr && mx > kOne) { atomicMax(p.status + 1, mx); atomicMax(p.status + 2, static_cast<uint32_t>(band_id)); atomicMax(p.status + 3, static_cast<uint32_t>(tid)); }
#endif
            if (__any(mx > kOne)) {
                if (__any(mx > kOne && mx != kNegZero)) bad_w |= 2u;
                else {
                    int l_col, l_line, l_row;
                    bool l_on;
                    loader_pos(l_col, l_line, l_row, l_on);
                    const uint32_t g0 = reinterpret_cast<const uint32_t*>(smem + 2 * kBufBytes)[fresh_tid()];
                    bool lane_bad = false;
                    for (int t = 0; t < D; ++t) {
                        const uint4 rl = myrec[static_cast<int64_t>(t) * kRecStep];
                        const Shape h = shape_unpack(rl.z);
                        const int nq = h.nq;
                        const uint32_t clo = h.clo, ncol = h.ncol, llo = 4 * h.rlo, nline = 4 * h.nrow;
                        const unsigned char* org = reinterpret_cast<const unsigned char*>((static_cast<uint64_t>(rl.y) << 32) | rl.x);
                        for (int r = 0; r < kNP; ++r) {
                            const bool in = l_on && l_col < nq && (static_cast<uint32_t>(l_col) - clo < ncol) && (static_cast<uint32_t>(l_line + 4 * r * kRPP) - llo < nline);
                            if (!in) continue;
                            const uint4 q = *reinterpret_cast<const uint4*>(org + g0 + static_cast<uint32_t>(r) * pass_off);
                            const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (BF) lane_bad |= !((d[c] & 0xffffu) <= kOne || (d[c] & 0xffffu) == kNegZero) || !((d[c] >> 16) <= kOne || (d[c] >> 16) == kNegZero);
                                else lane_bad |= !(d[c] <= kOne || d[c] == kNegZero);
                            }
                        }
                    }
                    if (__any(lane_bad)) bad_w |= 2u;
                }
            }
        }
    }

    // ---- epilogue per pixel -------------------------------------------------------------------------------------------------
    uint32_t bad = bad_w;
    float dl = 0.0f, phl = 1.0f, pwl = 1.0f;
    if (check_last) dl = dhw[3 * (D - 1) + 0], phl = dhw[3 * (D - 1) + 1], pwl = dhw[3 * (D - 1) + 2];
    const int px_e = bxi * (NSB * SBW) + sb * SBW + (fresh_tid() & 63);
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int py = byi * SBH + wj + WPS * q;
        if (check_last) {  // assert_not_out_of_last_plane (mpi.py:381-395): u,v of the last plane, once per pixel
            float ix, iy, s, u, v;
            plane_coord<AC>(dl - ez, phl, pwl, ex, ey, rx[q], ry[q], rz[q], cx, cy, ix, iy, s, u, v);
            if (!(u >= -1.0f && u <= 1.0f && v >= -1.0f && v <= 1.0f)) bad |= 1u;
        }
        float r = A[q].r, g = A[q].g, b = A[q].b;
        if (p.flags & (1u << 1)) {  // mpi_renderer.py:467  2*c - 1
            r = 2.0f * r - 1.0f;
            g = 2.0f * g - 1.0f;
            b = 2.0f * b - 1.0f;
        }
        if (px_e < W && py < H) {
            const int64_t pix = static_cast<int64_t>(py) * W + px_e;
            float* __restrict__ out = p.rgb_out + static_cast<int64_t>(n) * 3 * HW + pix;
            out[0] = r;
            out[HW] = g;
            out[2 * HW] = b;
            p.depth_out[static_cast<int64_t>(n) * HW + pix] = finish_depth<STRICT>(A[q], ray_dot(q));
            if (p.T_out) p.T_out[static_cast<int64_t>(n) * HW + pix] = A[q].T;
        }
    }
    report_status(p.status, bad);
#ifdef GMPI_TUNE
    if (stamp && threadIdx.x == 0) {
        uint64_t stamp_end;
        uint32_t xcc;
        asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_end), "=s"(xcc));
        p.status[64 + 3 * blockIdx.x] = static_cast<uint32_t>(stamp_start), p.status[65 + 3 * blockIdx.x] = static_cast<uint32_t>(stamp_end), p.status[66 + 3 * blockIdx.x] = xcc & 15u;
    }
#endif
#ifdef GMPI_PROF
    if (p.status != nullptr && band_id == 700 && threadIdx.x == 320) {  // one wave in the middle of the launch
        uint64_t now_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_));
        for (int i = 0; i < 8; ++i) p.status[8 + i] = prof_acc[i];
        p.status[16] = static_cast<uint32_t>(now_ - prof_start);
    }
#endif
}

static int nsb_of(int dtype) { return dtype != 0 ? Geo<bf16_t>::NSB : Geo<float>::NSB; }  // (bf16 and fp16 share the 16-bit geometry)
static void band_grid(const KParams& p, int nsb, int& bands_x, int& bands_y, int& n_bands) {
    bands_x = (p.W + nsb * SBW - 1) / (nsb * SBW), bands_y = (p.H + SBH - 1) / SBH;
    n_bands = bands_x * bands_y * p.N;
}
// workspace: [n_bands] header words + [N] view gate words + 8 ticket counters | (N * D + 2) plane records of 32 bytes | (n_bands * D + 2) * NSB box records of 16 bytes (each part 256-aligned)
static uint64_t align256(uint64_t v) { return (v + 255) / 256 * 256; }
static uint64_t ws_hdr_bytes(int n_bands, int n_views) { return align256((static_cast<uint64_t>(n_bands) + n_views + 8) * 4); }   // header words | view gate words | 8 ticket counters
static uint64_t ws_pl_bytes(const KParams& p) { return align256((static_cast<uint64_t>(p.N) * p.D + 2) * kPlU4 * 16); }
static uint64_t ws_bytes(const KParams& p, int nsb) {
    int bx, by, nb;
    band_grid(p, nsb, bx, by, nb);
    return ws_hdr_bytes(nb, p.N) + ws_pl_bytes(p) + (static_cast<uint64_t>(nb) * p.D + 2) * nsb * 16;
}

template <typename TexT>
static hipError_t launch_t(const KParams& p, hipStream_t stream) {
    constexpr int NSB = Geo<TexT>::NSB;
    int bands_x, bands_y, n_bands;
    band_grid(p, NSB, bands_x, bands_y, n_bands);
    const int group_items = bands_x * bands_y * (p.view_to_mpi == nullptr ? p.views_per_mpi : 1);
    const int tail = std::min(p.band_tail, (group_items + 7) / 8);
    const dim3 grid(xcd_grid_per_group(group_items, n_bands) + 8u * static_cast<unsigned>(tail)), block(Geo<TexT>::kThreads);   // (+ the extra blocks that only draw tickets)
    uint32_t* tickets = static_cast<uint32_t*>(p.ws) + n_bands + p.N;
    const bool acf = p.flags & 1u;
    const float cx = acf ? static_cast<float>(p.Wt - 1) * 0.5f : static_cast<float>(p.Wt), cy = acf ? static_cast<float>(p.Ht - 1) * 0.5f : static_cast<float>(p.Ht);
    uint32_t* hdr = static_cast<uint32_t*>(p.ws);
    uint4* pl = reinterpret_cast<uint4*>(static_cast<unsigned char*>(p.ws) + ws_hdr_bytes(n_bands, p.N));
    uint4* recs = reinterpret_cast<uint4*>(static_cast<unsigned char*>(p.ws) + ws_hdr_bytes(n_bands, p.N) + ws_pl_bytes(p));
    // 1. the geometry table (one workgroup per band; writes every word the render kernel reads but the two planes of padding, whose content
    //    is never used)
    const dim3 tgrid(static_cast<unsigned>(n_bands)), tblock(static_cast<unsigned>(std::min(1024, (p.D * NSB + 63) / 64 * 64)));  // one record per thread up to 256 planes
    if (acf) hipLaunchKernelGGL((band_table_kernel<TexT, true>), tgrid, tblock, 0, stream, p, bands_x, bands_y, n_bands, cx, cy, recs, pl, hdr, tickets);
    else hipLaunchKernelGGL((band_table_kernel<TexT, false>), tgrid, tblock, 0, stream, p, bands_x, bands_y, n_bands, cx, cy, recs, pl, hdr, tickets);
    // 2. the render
    const int sel = (p.flags & 1u ? 4 : 0) | (p.flags & (1u << 4) ? 2 : 0) | (p.flags & (1u << 3) ? 1 : 0);  // align_corners, strict order, range check
    switch (sel) {
#define GMPI_BAND_CASE(I, AC_, ST_, CK_) \
    case I: hipLaunchKernelGGL((render_band_kernel<TexT, AC_, ST_, CK_>), grid, block, 0, stream, p, bands_x, bands_y, n_bands, cx, cy, recs, pl, hdr, tickets); break;
        GMPI_BAND_CASE(0, false, false, false) GMPI_BAND_CASE(1, false, false, true) GMPI_BAND_CASE(2, false, true, false) GMPI_BAND_CASE(3, false, true, true)
        GMPI_BAND_CASE(4, true, false, false) GMPI_BAND_CASE(5, true, false, true) GMPI_BAND_CASE(6, true, true, false) GMPI_BAND_CASE(7, true, true, true)
#undef GMPI_BAND_CASE
    }
    return hipGetLastError();
}

}  // namespace band

uint64_t band_workspace_bytes(const KParams& p, int dtype) { return band::ws_bytes(p, band::nsb_of(dtype)); }
int band_pixels_wide(int dtype) { return band::nsb_of(dtype) * band::SBW; }

// the view gate words of the workspace (KParams::gate of an AUTO launch)
uint32_t* band_gate_words(const KParams& p, int dtype) {
    int bx, by, nb;
    band::band_grid(p, band::nsb_of(dtype), bx, by, nb);
    return static_cast<uint32_t*>(p.ws) + nb;
}

bool band_variant_supports(const KParams& p, int dtype) {
    // fp32, bf16 and (round 4) fp16 volumes: a d16_hi load of a bf16 texel IS its fp32 value; an fp16 texel is converted inside the bilinear FMAs.
    if (dtype < 0 || dtype > 2) return false;
    const int nsb = band::nsb_of(dtype), bw = nsb * band::SBW;
    if (p.ws == nullptr || p.ws_bytes < band::ws_bytes(p, nsb) || reinterpret_cast<uintptr_t>(p.ws) % 256 != 0) return false;  // needs the caller's workspace
    if (static_cast<int64_t>(p.N) * p.D * ((p.W + bw - 1) / bw) * ((p.H + 7) / 8) > (int64_t(1) << 28)) return false;  // record indices stay in 32 bits
    const int es = dtype == 0 ? 4 : 2, tpi = 16 / es;
    if (p.Wt % tpi != 0) return false;
    if (reinterpret_cast<uintptr_t>(p.rgba) % 16 != 0) return false;
    if (p.s_row % tpi != 0 || p.s_chan % tpi != 0 || p.s_plane % tpi != 0 || p.s_mpi % tpi != 0) return false;
    if (p.Ht > 8192 || p.Wt > 8192) return false;  // tap addresses are formed in fp32
    const int64_t span = 3 * p.s_chan + 18 * p.s_row + 128;  // the in-plane item offset (row < 6, + 2 passes of 6 rows, 3 channels) is kept in 32 bits
    if (span >= (int64_t(1) << 31) / es) return false;
    return true;
}

hipError_t launch_band(const KParams& p0, int dtype, int tune, hipStream_t stream) {
    KParams p = p0;
    p.band_cols = dtype != 0 ? band::kWindowCols16 : p.D > band::kDeepPlanes ? band::kWindowCols32Deep : band::kWindowCols32;
    p.band_rot = band::kViewRotation, p.band_split = band::kRunPieces, p.band_tail = band::kTicketTail;
#ifdef GMPI_TUNE  // profiling builds: tune bits 8-9 = ablations (no memory traffic / no compositing); GMPI_TUNE_ORDER = band columns per XCD window
    p.flags |= static_cast<uint32_t>((tune >> 8) & 127) << 16;
    static const int env_order = [] { const char* e = getenv("GMPI_TUNE_ORDER"); return e ? atoi(e) : 0; }();
    if (env_order > 0) p.band_cols = env_order;
    static const int env_rot = [] { const char* e = getenv("GMPI_TUNE_ROT"); return e ? atoi(e) : -1; }();
    if (env_rot >= 0) p.band_rot = env_rot;
    static const int env_split = [] { const char* e = getenv("GMPI_TUNE_SPLIT"); return e ? atoi(e) : 0; }();
    if (env_split > 0) p.band_split = env_split;
    static const int env_tail = [] { const char* e = getenv("GMPI_TUNE_TAIL"); return e ? atoi(e) : -1; }();
    if (env_tail >= 0) p.band_tail = env_tail;
#else
    (void)tune;
#endif
    if (dtype == 1) return band::launch_t<bf16_t>(p, stream);
    if (dtype == 2) return band::launch_t<f16_t>(p, stream);
    if (dtype == 0) return band::launch_t<float>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace gmpi
