// conv_mfma.hip -- 3x3 convolution (stride 1, padding 1) of NHWC float16 activations as an implicit GEMM on the gfx950
// matrix cores: the convolutions of the Zero123 SDS step (SD-1.x UNet at batch 8 / 32^2 ... 4^2 latents, VAE encoder at batch
// 4 / 256^2 ... 32^2), where MIOpen's `igemm_fwd_gtcx35_nhwc_fp16` kernels reach 4-17 % of the dense fp16 peak
// (tools/conv_shapes.py, profiles/r03_zero123.md).
//
//     y[p][co] = bias[co] (+ res[p][co]) + sum_{tap, ci} x[pixel(p) + tap][ci] * w[co][tap][ci]
//
// GEMM view: M = N H W output pixels, N = C_out, K = 9 C_in; both operands are K-CONTIGUOUS in memory -- an input pixel's
// channels (NHWC) and a filter's taps x channels (torch's channels_last weight IS [C_out][3][3][C_in]) -- which is exactly the
// fragment the f16 MFMA wants from a lane (v_mfma_f32_32x32x16_f16: lane l holds 8 consecutive k of row l & 31).  So:
//   * a k-tile = one tap x 32 channels: 64 contiguous bytes per pixel / per filter, four 16-byte pieces;
//   * global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR round trip, no ds_write -- the write path of the LDS is four
//     times slower than its read path on this chip) through BUFFER DESCRIPTORS: a per-lane byte offset that never changes in the
//     loop + a wave-uniform SGPR offset per k-tile; a padding tap carries an out-of-range offset and the hardware writes zeros.
//     (The first version computed 64-bit addresses per tap: ~100 instructions per k-tile beside 8 MFMAs, and the loop was
//     bound by instruction issue, not by anything it moved.)
//   * the DMA lands 64 lanes x 16 B LINEARLY, but the per-lane SOURCE is free: lane -> (row, piece) is chosen so that piece c of row
//     r sits at slot 4 r + (c ^ ((r >> 2) & 3)), which makes every ds_read_b128 of a fragment (16 lanes = 16 rows, one piece)
//     hit 16 different 16-byte columns of the 256-byte LDS row: conflict-free (MI355X_MICROARCH.md, LDS lane groups);
//   * a 4-deep ring of k-tiles, raw s_barrier + counted vmcnt, the fragments of the next half k-tile read while the MFMAs of the
//     current one run (two register sets), ONE barrier per k-tile in its middle; the loop unrolled over the ring so that every
//     LDS address is a register + an immediate;
//   * workgroup = 4 waves, each a 64 x 64 block of the 128 x 128 output tile (2 x 2 MFMA tiles: 4 ds_read_b128 feed 4 MFMAs), two
//     workgroups per CU; the filter fragment is the MFMA's A operand so that a lane ends with 4 consecutive output channels, and
//     the epilogue goes through LDS to 16-byte stores (bias, residual there);
//   * small problems (the 16^2 .. 4^2 levels) split K over workgroups -- only until every CU has one workgroup; partial sums in
//     float32, reduced with the epilogue by a second kernel;
//   * a DIRECT variant for wide images / large problems keeps the input patch of a 256-pixel tile resident in LDS for the nine taps.
// The operator is the forward convolution; the data gradient of a stride-1 convolution is the same operator on the flipped,
// transposed filter (conv_mfma.py), which is how the VAE encoder's backward runs here.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCvBK = 32;            // channels per k-tile (one tap)

__device__ __attribute__((aligned(64))) uint4 g_conv_zero[4];     // the line the padding taps read (never written)

struct ConvDesc {
    int N, H, W, Cin, Cout, M;       // H, W: OUTPUT size; M = N H W output pixels
    int Hin, Win, stride, pad;       // input size; output (y, x), tap (ky, kx) reads input (stride y + ky - pad, stride x + kx - pad_x)
    int pad_x;                       // (= pad except for the 2 x 1 / 1 x 2 filters of the stride-2 data gradient)
    int ostep, oy0, ox0;             // ostep > 0: output pixel (n, y, x) is WRITTEN at (n, ostep y + oy0, ostep x + ox0) of an [N][ostep H][ostep W] image
    const _Float16 *x, *w, *bias, *res;
    _Float16 *y;
    float *partial;                  // [splits][M][Cout] when splits > 1
    int splits, kt_total, kt_per;    // k-tiles (taps x Cin / 32) in all / per split
    int act;                         // epilogue: 0 none | 1 GEGLU: tile columns [0, 64) x gelu(columns [64, 128)), y is [M][Cout / 2] (dm4d_linear_f16)
    int probe;                       // timing experiments only (-DDM4D_CONV_PROBE, env DM4D_CONV_PROBE): 1 no stores, 2 no MFMA, 4 no DMA
};
#ifdef DM4D_CONV_PROBE
#define CV_PROBE(bit) (cv_probe & (bit))
#else
#define CV_PROBE(bit) false
#endif

// slot (16-byte unit) of piece c of row r in a stage's operand tile
__device__ __forceinline__ int cv_slot(int r, int c) { return 4 * r + (c ^ ((r >> 2) & 3)); }

// WM x WN waves, each MB x NB MFMA tiles of 32 x 32: workgroup tile (32 MB WM) x (32 NB WN), kCvStages-deep ring.
// ---- epilogue shared by both kernels.  The FILTER fragment is the MFMA's A operand, so D[row][col] has col = lane & 31 = the
// wave's pixel row (tile row `row_of[i]`), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = output channel: a lane holds 4 consecutive
// channels of one pixel per register quad.  `pix(row)` maps a tile row to its output pixel (or -1).
//   split-K: float32 partial sums, one 16-byte store per quad.
//   else: float16 (+ bias, one rounding) through LDS as [tile row][channel] rows of BN halves + 16 bytes, so that every global
//   store (and residual load) is 16 bytes per lane, 16 lanes per 256-byte run of a pixel's channels; the residual is added there
//   (rounded again, like the separate residual add it replaces).
template <int BM, int BN, int NW, int MB, int NB, class Pix>
__device__ __forceinline__ void conv_epilogue(const ConvDesc &d, f32x16 (&acc)[MB][NB], const int (&row_of)[MB], int col0, int n0, char *stg, int tid,
                                              int lane, [[maybe_unused]] int cv_probe, Pix pix)
{
    if (d.splits > 1) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int px = pix(row_of[i]);
            if (px < 0) continue;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = n0 + col0 + 32 * j + 8 * q + 4 * (lane >> 5);
                    if (co >= d.Cout || CV_PROBE(1)) continue;
                    *reinterpret_cast<float4 *>(d.partial + ((size_t)blockIdx.z * d.M + px) * d.Cout + co) =
                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                }
        }
        return;
    }
    constexpr int kRowB = BN * 2 + 16;
    __builtin_amdgcn_s_barrier();                 // every wave is done with the operand buffers the staging tile overlays
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                const int cl = col0 + 32 * j + 8 * q + 4 * (lane >> 5);             // channel within the tile
                f16x4 h, bq;
#pragma unroll
                for (int k = 0; k < 4; ++k) bq[k] = (_Float16)0.f;
                if (d.bias && n0 + cl < d.Cout) bq = *reinterpret_cast<const f16x4 *>(d.bias + n0 + cl);
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = (_Float16)(acc[i][j][4 * q + k] + (float)bq[k]);
                *reinterpret_cast<f16x4 *>(stg + (size_t)row_of[i] * kRowB + 2 * cl) = h;
            }
    __syncthreads();
    if constexpr (BN == 128) {
        if (d.act == 1) {       // GEGLU on the float16-rounded projections (what the separate kernel read): value x gelu(gate), erf form
            constexpr int kOutPieces = 8, kRows = 64 * NW / kOutPieces;
            const int piece = tid % kOutPieces;
            const size_t ldy = (size_t)(d.Cout >> 1);
#pragma unroll 2
            for (int row = tid / kOutPieces; row < BM; row += kRows) {
                const int px = pix(row);
                if (px < 0 || CV_PROBE(1)) continue;
                const f16x8 v = *reinterpret_cast<const f16x8 *>(stg + (size_t)row * kRowB + 16 * piece);
                const f16x8 g = *reinterpret_cast<const f16x8 *>(stg + (size_t)row * kRowB + 16 * (piece + 8));
                f16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gg = (float)g[k];
                    o[k] = (_Float16)((float)v[k] * (0.5f * gg * (1.0f + erff(gg * 0.70710678118654752f))));
                }
                *reinterpret_cast<f16x8 *>(d.y + (size_t)px * ldy + (n0 >> 1) + 8 * piece) = o;
            }
            return;
        }
    }
    constexpr int kPieces = BN / 8;                         // 16-byte pieces per pixel row
    constexpr int kRowsPerPass = 64 * NW / kPieces;
    static_assert((64 * NW) % kPieces == 0, "tile width");
    const int piece = tid % kPieces, co = n0 + 8 * piece;
    if (co >= d.Cout) return;
#pragma unroll 4
    for (int row = tid / kPieces; row < BM; row += kRowsPerPass) {
        const int px = pix(row);
        if (px < 0 || CV_PROBE(1)) continue;
        f16x8 v = *reinterpret_cast<const f16x8 *>(stg + (size_t)row * kRowB + 16 * piece);
        const size_t o = (size_t)px * d.Cout + co;
        if (d.res) {
            const f16x8 rv = *reinterpret_cast<const f16x8 *>(d.res + o);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (_Float16)((float)v[k] + (float)rv[k]);
        }
        *reinterpret_cast<f16x8 *>(d.y + o) = v;
    }
}

// (the body is a __device__ function: the host pass cannot instantiate a __global__ template that uses the buffer builtins)
// The filter is KH x KW: 3 x 3; 1 x 1 = a GEMM y = x w^T over the [pixels][channels] view (dm4d_linear_f16); 2 x 2, 2 x 1, 1 x 2, 1 x 1 with
// a strided output = the four parity classes of the stride-2 data gradient (dm4d_conv3x3_s2_dgrad_nhwc_f16)
template <int WM, int WN, int MB, int NB, int kCvStages, int KH, int KW>
__device__ __forceinline__ void conv3x3_tile(const ConvDesc &d)
{
    constexpr int TAPS = KH * KW;
    constexpr int NW = WM * WN, BM = 32 * MB * WM, BN = 32 * NB * WN;
    constexpr int A_INSTR = (BM * 4 + 64 * NW - 1) / (64 * NW), B_INSTR = (BN * 4 + 64 * NW - 1) / (64 * NW);   // DMA instructions per wave and stage
    constexpr int A_SLOTS = A_INSTR * 64 * NW, B_SLOTS = B_INSTR * 64 * NW;         // 16-byte slots per stage (rows past BM / BN: padding)
    extern __shared__ __attribute__((aligned(1024))) uint4 smem[];                  // [stage][A_SLOTS + B_SLOTS]
    [[maybe_unused]] const int cv_probe = d.probe;
    if (CV_PROBE(8)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // (provably wave-uniform: LDS-DMA bases stay in SGPRs)
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kt0 = (d.splits > 1 ? (int)blockIdx.z : 0) * d.kt_per, kt1 = min(d.kt_total, kt0 + d.kt_per);   // (splits == 1: blockIdx.z is the caller's)
    const int HW = d.H * d.W, cpt = d.Cin / kCvBK;                        // k-tiles per tap
    constexpr unsigned kOob = 0x80000000u;                                // a voffset past every descriptor's range: the DMA writes zeros

    // ---- the DMA's addresses.  The loop must not spend VALU issue slots on them (measured: with MFMA, DMA and stores removed
    // the old loop still took a third of the kernel -- ~100 scalar / vector instructions per k-tile beside its 8 MFMAs), so:
    //   source = buffer descriptor (SGPRs) + per-lane byte offset (a VGPR that changes only when the TAP changes) + a
    //   wave-uniform byte offset (an SGPR, + 64 per k-tile).  The input's descriptor starts pad (W_in + 1) pixels BEFORE the tensor so
    //   that the uniform tap offset ((dy + 1) W + dx + 1) C_in is never negative; a lane whose tap falls outside the image (or
    //   whose row is past M / C_out) carries kOob and the hardware's range check fills its 16 bytes with zeros.
    const auto a_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(d.x) - (size_t)(d.pad * d.Win + d.pad_x) * d.Cin, 0,
        (int)(((size_t)d.N * d.Hin * d.Win + 2 * d.Win + 2) * d.Cin * 2), 0x00020000);
    const auto b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(d.w), 0, (int)((size_t)d.Cout * TAPS * d.Cin * 2), 0x00020000);
    unsigned a_vo[A_INSTR], a_taps[A_INSTR], a_cur[A_INSTR], b_vo[B_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int s = 64 * (wave * A_INSTR + i) + lane, r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
        const int p = m0 + r;
        const bool ok = r < BM && p < d.M;
        const int pp = ok ? p : 0, y = (pp % HW) / d.W, x = pp % d.W;
        a_vo[i] = (unsigned)(((pp / HW) * d.Hin + d.stride * y) * d.Win + d.stride * x) * (unsigned)(d.Cin * 2) + 16u * c;
        unsigned taps = 0;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int yy = d.stride * y + tap / KW - d.pad, xx = d.stride * x + tap % KW - d.pad_x;
            if (ok && (unsigned)yy < (unsigned)d.Hin && (unsigned)xx < (unsigned)d.Win) taps |= 1u << tap;
        }
        a_taps[i] = taps;
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int s = 64 * (wave * B_INSTR + i) + lane, r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
        const int co = n0 + r;
        b_vo[i] = (r < BN && co < d.Cout) ? (unsigned)co * (unsigned)(TAPS * d.Cin * 2) + 16u * c : kOob;
    }
    // the issue stream's position: tap, chunk and the two uniform offsets
    int is_tap = kt0 / cpt, is_chunk = kt0 % cpt;
    int is_a = ((is_tap / KW) * d.Win + is_tap % KW) * d.Cin * 2 + is_chunk * (kCvBK * 2);
    int is_b = (is_tap * d.Cin + is_chunk * kCvBK) * 2;
    auto set_tap = [&]() {
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) a_cur[i] = ((a_taps[i] >> is_tap) & 1u) ? a_vo[i] : kOob;
    };
    set_tap();
    constexpr int kStageSlots = A_SLOTS + B_SLOTS;
    auto issue = [&](int stage) {       // the next k-tile of the stream into `stage`
        if (!(CV_PROBE(4))) {
            uint4 *sa = smem + stage * kStageSlots, *sb = sa + A_SLOTS;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void *)(sa + 64 * (wave * A_INSTR + i)), 16, a_cur[i], is_a, 0, 0);
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (__attribute__((address_space(3))) void *)(sb + 64 * (wave * B_INSTR + i)), 16, b_vo[i], is_b, 0, 0);
        }
        is_a += kCvBK * 2;
        is_b += kCvBK * 2;
        if (++is_chunk == cpt) {        // next tap: same channels from the start, the pixel one to the right (or a row down)
            is_chunk = 0;
            ++is_tap;
            is_a = ((is_tap / KW) * d.Win + is_tap % KW) * d.Cin * 2;
            set_tap();
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop.  A k-tile is two MFMA k-steps (ks = 0, 1: 16 channels each).  The fragments of the NEXT half step are read
    // from LDS while the MFMAs of the current one run (two register sets).  One barrier per k-tile, in the MIDDLE of it: before
    // it every wave has issued all its reads of tile t and consumed those of tile t - 1, so after it the DMA of tile t + S - 1 may
    // overwrite the stage of tile t - 1, and tile t + 1 -- issued S - 2 tiles ago -- has landed for everyone.  The loop is
    // unrolled over the ring so that every LDS address is a register + an immediate.
    f16x8 ra[2][MB] = {}, rb[2][NB] = {};
    int a_row[MB], b_row[NB];
#pragma unroll
    for (int i = 0; i < MB; ++i) a_row[i] = 32 * (MB * wm + i) + (lane & 31);
#pragma unroll
    for (int j = 0; j < NB; ++j) b_row[j] = 32 * (NB * wn + j) + (lane & 31);
    unsigned fa[2][MB], fb[2][NB];                         // the lane's fragment addresses (LDS bytes) in stage 0
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)smem;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[ks][i] = lds0 + 16u * cv_slot(a_row[i], 2 * ks + (lane >> 5));
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[ks][j] = lds0 + 16u * (A_SLOTS + cv_slot(b_row[j], 2 * ks + (lane >> 5)));
    }
    auto read = [&](auto KS, auto STAGE) {
        constexpr int ks = decltype(KS)::value, stage = decltype(STAGE)::value;
        if (CV_PROBE(64)) return;
#pragma unroll
        for (int i = 0; i < MB; ++i) ra[ks][i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(fa[ks][i] + stage * (kStageSlots * 16));
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[ks][j] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(fb[ks][j] + stage * (kStageSlots * 16));
    };
    // the FILTER fragment is the MFMA's A operand: D[row = filter][col = pixel], so a lane ends up with 4 consecutive output
    // channels of one pixel per register quad (the epilogue packs them)
    auto mma = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        if (CV_PROBE(2)) { acc[0][0][0] += (float)ra[ks][0][0] + (float)rb[ks][0][0]; return; }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb[ks][j], ra[ks][i], acc[i][j], 0, 0, 0);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;

    constexpr int S = kCvStages;
    constexpr int kPerStage = A_INSTR + B_INSTR;          // DMA instructions a wave has in flight per k-tile
    static_assert(S >= 3 && S <= 6, "ring depth");
    const int nk = CV_PROBE(16) ? 1 : kt1 - kt0;
    // s_waitcnt vmcnt(tiles x kPerStage) for a run-time number of k-tiles allowed in flight (the immediate must be a constant)
    auto wait_tiles = [&](int fly) {
        switch (fly) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPerStage) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kPerStage) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * kPerStage) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * kPerStage) : "memory"); break;
        }
    };
    static_assert(4 * kPerStage < 64, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int p = 0; p < S - 1; ++p)
        if (p < nk) issue(p);
    wait_tiles(min(S - 2, nk - 1));                   // tile 0 has landed; tiles 1 .. S - 2 may be in flight
    __builtin_amdgcn_s_barrier();
    if (nk > 0) read(K0{}, std::integral_constant<int, 0>{});
    // tile t (not the last) in ring stage U = t % S
    auto body = [&](auto U, int t) {
        constexpr int u = decltype(U)::value;
        read(K1{}, U);
        __builtin_amdgcn_sched_barrier(0);      // reads first, THEN the MFMAs they overlap with (the scheduler otherwise sinks them to 1-2 MFMAs before their use)
        mma(K0{});
        wait_tiles(min(S - 3, nk - 2 - t));     // tile t + 1 has landed; the tiles issued after it may be in flight
        if (!CV_PROBE(32)) __builtin_amdgcn_s_barrier();
        if (t + S - 1 < nk) issue((u + S - 1) % S);
        read(K0{}, std::integral_constant<int, (u + 1) % S>{});
        __builtin_amdgcn_sched_barrier(0);
        mma(K1{});
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tail = [&](auto U) {           // the last tile: nothing left to fetch
        read(K1{}, U);
        mma(K0{});
        mma(K1{});
    };
    // tiles t .. t + n - 1 (n <= S) starting in ring stage 0; the last of them is the launch's last tile iff `last`
    auto run = [&](int t, int n, bool last) {
#define DM4D_STEP(u)                                                                                  \
        if constexpr ((u) < S) {                                                                      \
            if (n > (u)) {                                                                            \
                if (last && n == (u) + 1) tail(std::integral_constant<int, (u)>{});                  \
                else body(std::integral_constant<int, (u)>{}, t + (u));                               \
            }                                                                                         \
        }
        DM4D_STEP(0) DM4D_STEP(1) DM4D_STEP(2) DM4D_STEP(3) DM4D_STEP(4) DM4D_STEP(5)
#undef DM4D_STEP
    };
    int t = 0;
    for (; t + S < nk; t += S) run(t, S, false);
    if (nk > 0) run(t, nk - t, true);             // 1 .. S tiles left, starting in stage 0

    static_assert((size_t)BM * (BN * 2 + 16) <= (size_t)kCvStages * (A_SLOTS + B_SLOTS) * 16, "epilogue staging does not fit the ring");
    conv_epilogue<BM, BN, NW>(d, acc, a_row, 32 * NB * wn, n0, reinterpret_cast<char *>(smem), tid, lane, cv_probe,
                              [&](int row) {
                                  const int px = m0 + row;
                                  if (px >= d.M) return -1;
                                  if (d.ostep == 0) return px;
                                  const int n = px / HW, r = px - n * HW, y = r / d.W, x = r - y * d.W;
                                  return ((n * d.H + y) * d.ostep + d.oy0) * (d.W * d.ostep) + x * d.ostep + d.ox0;
                              });
}

template <int WM, int WN, int MB, int NB, int kCvStages, int KH = 3, int KW = KH>
__global__ __launch_bounds__(64 * WM * WN) void k_conv3x3(ConvDesc d)
{
    conv3x3_tile<WM, WN, MB, NB, kCvStages, KH, KW>(d);
}

// the four parity classes of the stride-2 data gradient in ONE launch (blockIdx.z = class; dm4d_conv3x3_s2_dgrad_nhwc_f16)
struct S2DgradDesc { ConvDesc d; const _Float16 *w[4]; };
template <int kUnused>       // (a template like k_conv3x3: the host pass must not instantiate the tile, whose buffer builtins exist on the device only)
__global__ __launch_bounds__(256) void k_conv_s2_dgrad(S2DgradDesc a)
{
    ConvDesc d = a.d;
    const int cls = blockIdx.z, py = cls >> 1, px = cls & 1, kh = py ? 1 : 2, kw = px ? 1 : 2;
    d.pad = kh - 1; d.pad_x = kw - 1; d.oy0 = py; d.ox0 = px;
    d.w = a.w[cls];
    d.kt_total = d.kt_per = kh * kw * d.Cin / kCvBK;
    switch (cls) {
    case 0: conv3x3_tile<2, 2, 2, 2, 4 + kUnused, 2, 2>(d); break;      // (+ kUnused: a DEPENDENT call, instantiated with the kernel)
    case 1: conv3x3_tile<2, 2, 2, 2, 4 + kUnused, 2, 1>(d); break;
    case 2: conv3x3_tile<2, 2, 2, 2, 4 + kUnused, 1, 2>(d); break;
    default: conv3x3_tile<2, 2, 2, 2, 3 + kUnused, 1, 1>(d); break;      // (3-deep: <.., 4, 1, 1> is dm4d_linear_f16's, and the host pass
                                                                           // rejects a second kernel instantiating the same device-only specialization)
    }
}

// ---------------------------------------------------------------------------------------- direct variant
// The implicit-GEMM kernel fetches every input pixel NINE times (once per tap), and what its loop is short of is DMA ISSUE
// slots: a 1 KB LDS-DMA piece costs a wave ~60-180 cycles of issue (MI355X_MICROARCH.md), a 128 x 128 x 32 k-tile is 16 of
// them for 8 MFMAs per wave.  Here the workgroup's 256 output pixels are a TH x TW block of the image (TW = min(W, 32), TH =
// 256 / TW; rows are GLOBAL rows g = n H + y, so at W = 8 the block spans 4 images) and their (TH + 2) x (TW + 2) input PATCH
// of one 32-channel chunk is brought into LDS once for all nine taps: a tap only shifts the patch pixel a fragment row reads.
// Per k-tile that is 8 pieces of filters + 1/9 of <= 22 KB of patch for 256 x 128 x 32 MACs: 10.4 pieces per 64 MFMAs instead
// of 24 (256 x 128 implicit GEMM).
//   8 waves (4 x 2), each 64 x 64; filters: ring of 9 (tap, chunk) tiles (stage = tap: compile time in the unrolled tap loop), fetched
//   4 k-tiles ahead (with one workgroup of 227 registers per CU nothing else hides the L2 latency);
//   patch: double buffer, the next chunk's 3 pieces per wave issued at taps 0, 1, 2 of the current chunk; fragments of the next
//   half k-tile are read while the MFMAs of the current one run; one barrier per k-tile, in its middle (see conv3x3_tile).
//   Addresses: buffer descriptors + per-lane offsets that never change + wave-uniform SGPR offsets -- no VALU in the loop for
//   the DMA.  A fragment row whose tap crosses the top / bottom of its image reads a ZERO region in front of each patch buffer
//   (its patch pixel index is set to -2: the +-1 of the horizontal taps stays inside the 256 zero bytes); left / right image
//   borders and rows past the tensor are zeros in the patch itself (the descriptor's range check).
//   W a power of two >= 8 (conv_plan sends everything else to the implicit GEMM).
constexpr int kDirPatchSlots = 1536;                            // patch: 384 pixels x 4 pieces >= 34 x 10
constexpr int kDirZeroSlots = 16;                               // 256 bytes of zeros in front of each patch buffer
// NWN: wave columns (the tile is 256 pixels x 64 NWN filters, 4 NWN waves); RING: stages of the filter ring (stage = tap % RING:
// 3 or 9, compile time in the unrolled tap loop); AHEAD: filter tiles in flight ahead of the one being multiplied (< RING)
template <int NWN, int RING> constexpr int dir_lds_slots() { return 2 * (kDirZeroSlots + kDirPatchSlots) + RING * 256 * NWN; }
// DMA instructions a wave issues in the iteration of tap `tap` (more: a chunk follows this one; PIECES patch pieces per wave)
constexpr int dir_issues(int tap, bool more, int PIECES, int AHEAD) { return ((tap < PIECES && more) ? 1 : 0) + ((tap + AHEAD < 9 || more) ? 1 : 0); }
// ... and how many it may leave in flight at the barrier of tap `tap`: everything issued after the filters of k-tile + 1, i.e. in the
// previous AHEAD - 2 iterations (those before tap 0 belong to the previous chunk, which had a successor)
constexpr int dir_in_flight(int tap, bool more, int PIECES, int AHEAD)
{
    int n = 0;
    for (int k = 1; k <= AHEAD - 2; ++k) n += tap - k >= 0 ? dir_issues(tap - k, more, PIECES, AHEAD) : dir_issues(tap - k + 9, true, PIECES, AHEAD);
    return n;
}

template <int NWN, int RING, int AHEAD>
__device__ __forceinline__ void conv3x3_direct_tile(const ConvDesc &d, int chunks_per_split, int lgTW)
{
    constexpr int BM = 256, BN = 64 * NWN, NW = 4 * NWN;
    constexpr int kDirPieces = kDirPatchSlots / (64 * NW), kDirBSlots = 256 * NWN, kDirRing = RING, kDirAhead = AHEAD;
    static_assert(AHEAD >= 2 && AHEAD < RING && 9 % RING == 0 && kDirPieces <= 9, "ring");
    extern __shared__ __attribute__((aligned(1024))) uint4 smem[];
    [[maybe_unused]] const int cv_probe = d.probe;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int W = d.W, TW = 1 << lgTW, PW = TW + 2, TH = BM >> lgTW, R = d.N * d.H;
    const int ncb = W >> lgTW, rblk = blockIdx.x / ncb, cb = blockIdx.x - rblk * ncb;
    const int g0 = rblk * TH, x0 = cb << lgTW, n0 = blockIdx.y * BN;
    const int cpt = d.Cin / kCvBK;
    const int ch0 = blockIdx.z * chunks_per_split, nch = min(cpt, ch0 + chunks_per_split) - ch0;
    const int patch_px = (TH + 2) * PW;
    constexpr unsigned kOob = 0x80000000u;
    // LDS: [zeros | patch 0 | zeros | patch 1 | filter ring]
    uint4 *const s_patch0 = smem + kDirZeroSlots;
    constexpr int kPatchStride = kDirZeroSlots + kDirPatchSlots;          // slots from patch 0 to patch 1
    uint4 *const s_b = smem + 2 * kPatchStride;                           // [kDirRing][kDirBSlots]
    if (tid < kDirZeroSlots) { smem[tid] = make_uint4(0u, 0u, 0u, 0u); smem[kPatchStride + tid] = make_uint4(0u, 0u, 0u, 0u); }

    const auto a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(d.x) - (size_t)(W + 1) * d.Cin, 0,
                                                        (int)(((size_t)d.M + 2 * W + 2) * d.Cin * 2), 0x00020000);
    const auto b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(d.w), 0, (int)((size_t)d.Cout * 9 * d.Cin * 2), 0x00020000);
    // patch piece i of this lane: pixel q of the patch, 16-byte piece c (swizzled like every operand tile)
    unsigned p_vo[kDirPieces];
#pragma unroll
    for (int i = 0; i < kDirPieces; ++i) {
        const int s = 64 * (wave * kDirPieces + i) + lane, q = s >> 2, c = (s & 3) ^ ((q >> 2) & 3);
        const int pr = q / PW, pc = q - pr * PW;
        const int g = g0 - 1 + pr, x = x0 - 1 + pc;
        const bool ok = q < patch_px && (unsigned)x < (unsigned)W && (unsigned)g < (unsigned)R;
        p_vo[i] = ok ? (unsigned)((g + 1) * W + x + 1) * (unsigned)(d.Cin * 2) + 16u * c : kOob;       // (descriptor starts W + 1 pixels early)
    }
    unsigned b_vo;
    {
        const int s = 64 * wave + lane, r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
        const int co = n0 + r;
        b_vo = co < d.Cout ? (unsigned)co * (unsigned)(9 * d.Cin * 2) + 16u * c : kOob;
    }
    const int cin2 = d.Cin * 2;
    auto issue_patch = [&](int piece, int buf, int chunk) {              // chunk: relative to ch0
        if (CV_PROBE(4)) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void *)(s_patch0 + buf * kPatchStride + 64 * (wave * kDirPieces + piece)),
                                                 16, p_vo[piece], (ch0 + chunk) * (kCvBK * 2), 0, 0);
    };
    auto issue_b = [&](int tap, int stage, int chunk) {
        if (CV_PROBE(4)) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (__attribute__((address_space(3))) void *)(s_b + stage * kDirBSlots + 64 * wave), 16, b_vo,
                                                 tap * cin2 + (ch0 + chunk) * (kCvBK * 2), 0, 0);
    };

    // ---- fragment rows of this lane: patch pixel for the three tap rows (dy = -1, 0, +1), -2 where the tap row is outside the image
    int row_of[2], qa[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 64 * wm + 32 * i + (lane & 31), tr = r >> lgTW, tc = r & (TW - 1);
        const int g = g0 + tr, y = g % d.H;
        const bool row_ok = g < R;
        const int q = (tr + 1) * PW + tc + 1;
        row_of[i] = r;
        qa[i][0] = (row_ok && y > 0) ? q - PW : -2;
        qa[i][1] = row_ok ? q : -2;
        qa[i][2] = (row_ok && y < d.H - 1) ? q + PW : -2;
    }
    const int hi = lane >> 5;
    unsigned fb[2][2];                                      // filter fragment byte addresses in ring stage 0
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            fb[ks][j] = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)(s_b + cv_slot(64 * wn + 32 * j + (lane & 31), 2 * ks + hi));
    const unsigned pb0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)s_patch0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 ra[2][2] = {}, rb[2][2] = {};
    // fragments of half step ks of tap TAP from patch buffer at byte address `pbase`, filter stage TAP % 3
    auto read = [&](auto KS, auto TAP, unsigned pbase) {
        constexpr int ks = decltype(KS)::value, tap = decltype(TAP)::value, dyi = tap / 3, dx = tap % 3 - 1;
        if (CV_PROBE(64)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = qa[i][dyi] + dx;
            const unsigned a = pbase + 16u * (unsigned)(4 * q + ((2 * ks + hi) ^ ((q >> 2) & 3)));
            ra[ks][i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(a);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            rb[ks][j] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(fb[ks][j] + (tap % kDirRing) * (kDirBSlots * 16));
    };
    auto mma = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        if (CV_PROBE(2)) { acc[0][0][0] += (float)ra[ks][0][0] + (float)rb[ks][0][0]; return; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb[ks][j], ra[ks][i], acc[i][j], 0, 0, 0);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
#define DM4D_TAP(n) std::integral_constant<int, n>{}

    // prologue: the first patch, the first kDirAhead filter tiles
    if (nch > 0) {
#pragma unroll
        for (int i = 0; i < kDirPieces; ++i) issue_patch(i, 0, 0);
#pragma unroll
        for (int k = 0; k < kDirAhead; ++k) issue_b(k, k % kDirRing, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDirAhead - 1) : "memory");
    }
    __syncthreads();                                        // (also publishes the zero regions)
    if (nch > 0) read(K0{}, DM4D_TAP(0), pb0);
    // k-tile (chunk ch, tap TAP), not the last one: its second half, the barrier, the DMA of k-tile + kDirAhead (and of a piece of the
    // next patch, BEFORE it: the DMA retires in order), the first half of k-tile + 1
    auto body = [&](auto TAP, int ch, unsigned pcur, unsigned pnext, bool more) {
        constexpr int tap = decltype(TAP)::value;
        read(K1{}, TAP, pcur);
        __builtin_amdgcn_sched_barrier(0);      // reads first, THEN the MFMAs they overlap with (the scheduler otherwise sinks them to 1-2 MFMAs before their use)
        mma(K0{});
        // k-tile + 1's filters (and every patch piece issued before them) have landed
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dir_in_flight(tap, true, kDirPieces, kDirAhead)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dir_in_flight(tap, false, kDirPieces, kDirAhead)) : "memory");
        if (!CV_PROBE(32)) __builtin_amdgcn_s_barrier();
        if (tap < kDirPieces && more) issue_patch(tap, (ch + 1) & 1, ch + 1);
        if (tap + kDirAhead < 9) issue_b(tap + kDirAhead, (tap + kDirAhead) % kDirRing, ch);
        else if (more) issue_b(tap + kDirAhead - 9, (tap + kDirAhead - 9) % kDirRing, ch + 1);
        read(K0{}, std::integral_constant<int, (tap + 1) % 9>{}, tap == 8 ? pnext : pcur);
        __builtin_amdgcn_sched_barrier(0);
        mma(K1{});
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int ch = 0; ch < nch; ++ch) {
        const unsigned pcur = pb0 + (unsigned)(ch & 1) * (kPatchStride * 16), pnext = pb0 + (unsigned)((ch + 1) & 1) * (kPatchStride * 16);
        const bool more = ch + 1 < nch;
        body(DM4D_TAP(0), ch, pcur, pnext, more);
        body(DM4D_TAP(1), ch, pcur, pnext, more);
        body(DM4D_TAP(2), ch, pcur, pnext, more);
        body(DM4D_TAP(3), ch, pcur, pnext, more);
        body(DM4D_TAP(4), ch, pcur, pnext, more);
        body(DM4D_TAP(5), ch, pcur, pnext, more);
        body(DM4D_TAP(6), ch, pcur, pnext, more);
        body(DM4D_TAP(7), ch, pcur, pnext, more);
        if (more) body(DM4D_TAP(8), ch, pcur, pnext, more);
        else {                                              // the last k-tile: nothing left to fetch
            read(K1{}, DM4D_TAP(8), pcur);
            mma(K0{});
            mma(K1{});
        }
    }
#undef DM4D_TAP
    static_assert((size_t)BM * (BN * 2 + 16) <= (size_t)dir_lds_slots<NWN, RING>() * 16, "epilogue staging does not fit");
    conv_epilogue<BM, BN, NW>(d, acc, row_of, 64 * wn, n0, reinterpret_cast<char *>(smem), tid, lane, cv_probe, [&](int row) {
        const int g = g0 + (row >> lgTW);
        return g < R ? g * W + x0 + (row & (TW - 1)) : -1;
    });
}

template <int NWN, int RING, int AHEAD>
__global__ __launch_bounds__(256 * NWN) void k_conv3x3_direct(ConvDesc d, int chunks_per_split, int lgTW)
{
    conv3x3_direct_tile<NWN, RING, AHEAD>(d, chunks_per_split, lgTW);
}

// ---------------------------------------------------------------------------------------- direct variant, PING-PONG waves (round 6)
// The 8-wave direct kernel above keeps both waves of a SIMD in step from barrier to barrier: both read fragments, then both want the
// SIMD's matrix pipe -- the three phases of a k-tile (fragment reads, operand DMA, 8 MFMAs) hardly overlap (1380 cycles per k-tile
// against 512 of MFMA per SIMD; profiles/r03_zero123.md).  Here the two waves of a SIMD (wave w and w + 4: the dispatcher deals a
// workgroup's waves to the SIMDs cyclically) run the SAME stream half a period apart (MI355X_MICROARCH.md, "Two waves per SIMD"):
//     LOAD(t):    issue the DMA of k-tile t + AHEAD (and a piece of the next chunk's patch), read the 8 fragments of k-tile t into
//                 registers, wait for them and for this wave's DMA pieces of k-tile t + 1, s_barrier
//     COMPUTE(t): the 8 MFMAs of k-tile t, s_barrier
// with waves 4-7 one barrier behind waves 0-3, so that every interval between two barriers pairs one wave's 8 MFMAs (256 cycles of the
// SIMD's matrix pipe) with its partner's LDS reads and DMA issue -- matrix beside memory.  One register set of fragments (a wave never
// reads while it multiplies), two barriers per k-tile.  Same tile algebra, LDS layout, DMA order and epilogue as conv3x3_direct_tile<2, 9, 4>:
// a k-tile's tiles are waited for by every wave at the end of its LOAD of the k-tile before and published by the barrier behind it.
// MB: 32-row MFMA tiles of pixels per wave -- 2: the 256 x 128 workgroup tile; 4 (round 6): 512 x 128, for the wide levels whose grid still fills the machine:
// a filter piece then feeds 16 MFMAs per wave instead of 8 (0.83 instead of 1.33 LDS-DMA instructions per 8 MFMAs: the DMA issue, ~130 cycles
// an instruction, is what an interval of this kernel is as long as), the DMA is issued in the LOAD interval (12 fragment reads + 1.6 DMA beside 16 MFMAs).
template <int MB> constexpr int pp_patch_slots() { return MB == 2 ? kDirPatchSlots : 2560; }      // 4: (16 + 2) x 34 = 612 patch pixels x 4 pieces, in whole pieces per wave
template <int MB> constexpr int pp_lds_slots() { return 2 * (kDirZeroSlots + pp_patch_slots<MB>()) + 9 * 512; }
template <int AHEAD, int MB>
__device__ __forceinline__ void conv3x3_direct_pp_tile(const ConvDesc &d, int chunks_per_split, int lgTW)
{
    constexpr int NWN = 2, RING = 9;
    constexpr int BM = 128 * MB, BN = 64 * NWN, NW = 4 * NWN;
    constexpr int kDirPatchSlots = pp_patch_slots<MB>();      // (shadows the 256-pixel tile's constant)
    constexpr int kDirPieces = kDirPatchSlots / (64 * NW), kDirBSlots = 256 * NWN, kDirRing = RING, kDirAhead = AHEAD;
    static_assert(AHEAD >= 2 && AHEAD < RING && kDirPieces <= 9, "ring");
    extern __shared__ __attribute__((aligned(1024))) uint4 smem[];
    [[maybe_unused]] const int cv_probe = d.probe;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int W = d.W, TW = 1 << lgTW, PW = TW + 2, TH = BM >> lgTW, R = d.N * d.H;
    const int ncb = W >> lgTW, rblk = blockIdx.x / ncb, cb = blockIdx.x - rblk * ncb;
    const int g0 = rblk * TH, x0 = cb << lgTW, n0 = blockIdx.y * BN;
    const int cpt = d.Cin / kCvBK;
    const int ch0 = blockIdx.z * chunks_per_split, nch = min(cpt, ch0 + chunks_per_split) - ch0;
    const int patch_px = (TH + 2) * PW;
    constexpr unsigned kOob = 0x80000000u;
    uint4 *const s_patch0 = smem + kDirZeroSlots;
    constexpr int kPatchStride = kDirZeroSlots + kDirPatchSlots;
    uint4 *const s_b = smem + 2 * kPatchStride;
    if (tid < kDirZeroSlots) { smem[tid] = make_uint4(0u, 0u, 0u, 0u); smem[kPatchStride + tid] = make_uint4(0u, 0u, 0u, 0u); }

    const auto a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(d.x) - (size_t)(W + 1) * d.Cin, 0,
                                                        (int)(((size_t)d.M + 2 * W + 2) * d.Cin * 2), 0x00020000);
    const auto b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(d.w), 0, (int)((size_t)d.Cout * 9 * d.Cin * 2), 0x00020000);
    unsigned p_vo[kDirPieces];
#pragma unroll
    for (int i = 0; i < kDirPieces; ++i) {
        const int s = 64 * (wave * kDirPieces + i) + lane, q = s >> 2, c = (s & 3) ^ ((q >> 2) & 3);
        const int pr = q / PW, pc = q - pr * PW;
        const int g = g0 - 1 + pr, x = x0 - 1 + pc;
        const bool ok = q < patch_px && (unsigned)x < (unsigned)W && (unsigned)g < (unsigned)R;
        p_vo[i] = ok ? (unsigned)((g + 1) * W + x + 1) * (unsigned)(d.Cin * 2) + 16u * c : kOob;
    }
    unsigned b_vo;
    {
        const int s = 64 * wave + lane, r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
        const int co = n0 + r;
        b_vo = co < d.Cout ? (unsigned)co * (unsigned)(9 * d.Cin * 2) + 16u * c : kOob;
    }
    const int cin2 = d.Cin * 2;
    auto issue_patch = [&](int piece, int buf, int chunk) {
        if (CV_PROBE(4)) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void *)(s_patch0 + buf * kPatchStride + 64 * (wave * kDirPieces + piece)),
                                                 16, p_vo[piece], (ch0 + chunk) * (kCvBK * 2), 0, 0);
    };
    auto issue_b = [&](int tap, int stage, int chunk) {
        if (CV_PROBE(4)) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (__attribute__((address_space(3))) void *)(s_b + stage * kDirBSlots + 64 * wave), 16, b_vo,
                                                 tap * cin2 + (ch0 + chunk) * (kCvBK * 2), 0, 0);
    };
    int row_of[MB], qa[MB][3];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int r = 32 * MB * wm + 32 * i + (lane & 31), tr = r >> lgTW, tc = r & (TW - 1);
        const int g = g0 + tr, y = g % d.H;
        const bool row_ok = g < R;
        const int q = (tr + 1) * PW + tc + 1;
        row_of[i] = r;
        qa[i][0] = (row_ok && y > 0) ? q - PW : -2;
        qa[i][1] = row_ok ? q : -2;
        qa[i][2] = (row_ok && y < d.H - 1) ? q + PW : -2;
    }
    const int hi = lane >> 5;
    unsigned fb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            fb[ks][j] = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)(s_b + cv_slot(64 * wn + 32 * j + (lane & 31), 2 * ks + hi));
    const unsigned pb0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)s_patch0;

    f32x16 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 ra[2][MB] = {}, rb[2][2] = {};
    // byte offset (from a patch buffer's start) of the ks = 0 fragment of tile row i for each of the nine taps; ks = 1 is the same slot
    // with piece ^ 2, i.e. offset ^ 32 (the buffers are 256-byte aligned): 18 registers instead of ~6 VALU operations per read -- a
    // LOAD interval is as long as its instruction stream (DMA issue, addresses, 8 reads, their latency), not as its bytes
    // (MB = 4: 36 of them do not fit beside 128 accumulator registers -- 256 VGPRs + 10 spilled; its LOAD interval lies beside 16 MFMAs and has
    //  the issue slots to compute the four addresses of a k-tile from qa[][], ~5 VALU operations each)
    constexpr int kPo = MB == 2 ? 9 : 1;
    unsigned po[MB][kPo];
    if constexpr (MB == 2) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int q = qa[i][tap / 3] + (tap % 3 - 1);
                po[i][tap] = 16u * (unsigned)(4 * q + (hi ^ ((q >> 2) & 3)));
            }
    }
    // every fragment of k-tile (tap TAP) from the patch buffer at byte address `pbase`, filter stage TAP % RING
    auto read_all = [&](auto TAP, unsigned pbase) {
        constexpr int tap = decltype(TAP)::value;
        if (CV_PROBE(64)) return;
        unsigned pa[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            if constexpr (MB == 2) pa[i] = po[i][tap];
            else {
                int q = qa[i][tap / 3];
                asm volatile("" : "+v"(q));          // (computed HERE: the loop-invariant code motion would keep all 36 addresses in registers again)
                q += tap % 3 - 1;
                pa[i] = 16u * (unsigned)(4 * q + (hi ^ ((q >> 2) & 3)));
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const unsigned a = pbase + (pa[i] ^ (ks ? 32u : 0u));
                ra[ks][i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(a);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                rb[ks][j] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(fb[ks][j] + (tap % kDirRing) * (kDirBSlots * 16));
        }
    };
#define DM4D_TAP(n) std::integral_constant<int, n>{}
    // prologue: the first patch, the first AHEAD filter tiles; k-tile 0's operands have landed behind the barrier
    if (nch > 0) {
#pragma unroll
        for (int i = 0; i < kDirPieces; ++i) issue_patch(i, 0, 0);
#pragma unroll
        for (int k = 0; k < kDirAhead; ++k) issue_b(k, k % kDirRing, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDirAhead - 1) : "memory");
    }
    __syncthreads();                                        // (also publishes the zero regions)
    if (wn == 1) __builtin_amdgcn_s_barrier();              // the second half starts one interval late: its LOAD beside the first half's COMPUTE
    // Measured alternatives (4 x 64^2, 512 -> 512, us per call; the lock-step kernel 78.7-81): the DMA issued in the LOAD interval 71.3; ONE
    // barrier per k-tile with the second half multiplying k-tile t - 1 BEFORE it loads k-tile t (an interval = one wave's LOAD + COMPUTE)
    // 73.5; this form 68.3.  The matrix stream alone (no reads, no DMA) takes 54 us of the lock-step kernel's 87 in the probe build: the
    // chip does not sustain the nominal 2.4 GHz x 1024 flop per SIMD cycle under it, so the ceiling is nearer 1.6 than 2.5 PFLOP/s.
    auto step = [&](auto TAP, int ch, unsigned pcur, bool more) {
        constexpr int tap = decltype(TAP)::value;
        // ---- LOAD: the fragments of this k-tile; this wave's pieces of k-tile + 1 (its filters and every patch piece issued before
        //      them) have landed -- what may stay in flight was issued behind them, in the AHEAD - 2 intervals before this one (and, MB = 4,
        //      in this one: the DMA of k-tile + AHEAD -- a piece of the next chunk's patch BEFORE it, the DMA retires in order -- leads the interval)
        constexpr bool kDmaInLoad = MB == 4;
        if constexpr (kDmaInLoad) {
            if (tap < kDirPieces && more) issue_patch(tap, (ch + 1) & 1, ch + 1);
            if (tap + kDirAhead < 9) issue_b(tap + kDirAhead, (tap + kDirAhead) % kDirRing, ch);
            else if (more) issue_b(tap + kDirAhead - 9, (tap + kDirAhead - 9) % kDirRing, ch + 1);
        }
        read_all(TAP, pcur);
        constexpr int kMine1 = kDmaInLoad ? dir_issues(tap, true, kDirPieces, kDirAhead) : 0, kMine0 = kDmaInLoad ? dir_issues(tap, false, kDirPieces, kDirAhead) : 0;
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dir_in_flight(tap, true, kDirPieces, kDirAhead) + kMine1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dir_in_flight(tap, false, kDirPieces, kDirAhead) + kMine0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!CV_PROBE(32)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- COMPUTE: 4 MB MFMAs; MB = 2: the DMA of k-tile + AHEAD is issued between them
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (CV_PROBE(2)) acc[0][0][0] += (float)ra[ks][i][0] + (float)rb[ks][j][0];
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb[ks][j], ra[ks][i], acc[i][j], 0, 0, 0);
                    if constexpr (!kDmaInLoad) {
                        if (ks == 0 && i == 0 && j == 0) {
                            if (tap < kDirPieces && more) issue_patch(tap, (ch + 1) & 1, ch + 1);
                        }
                        if (ks == 0 && i == 0 && j == 1) {
                            if (tap + kDirAhead < 9) issue_b(tap + kDirAhead, (tap + kDirAhead) % kDirRing, ch);
                            else if (more) issue_b(tap + kDirAhead - 9, (tap + kDirAhead - 9) % kDirRing, ch + 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        if (!CV_PROBE(32)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int ch = 0; ch < nch; ++ch) {
        const unsigned pcur = pb0 + (unsigned)(ch & 1) * (kPatchStride * 16);
        const bool more = ch + 1 < nch;
        step(DM4D_TAP(0), ch, pcur, more);
        step(DM4D_TAP(1), ch, pcur, more);
        step(DM4D_TAP(2), ch, pcur, more);
        step(DM4D_TAP(3), ch, pcur, more);
        step(DM4D_TAP(4), ch, pcur, more);
        step(DM4D_TAP(5), ch, pcur, more);
        step(DM4D_TAP(6), ch, pcur, more);
        step(DM4D_TAP(7), ch, pcur, more);
        step(DM4D_TAP(8), ch, pcur, more);
    }
#undef DM4D_TAP
    if (wn == 0) __builtin_amdgcn_s_barrier();              // (the second half's last COMPUTE)
    static_assert((size_t)BM * (BN * 2 + 16) <= (size_t)pp_lds_slots<MB>() * 16, "epilogue staging does not fit");
    conv_epilogue<BM, BN, NW>(d, acc, row_of, 64 * wn, n0, reinterpret_cast<char *>(smem), tid, lane, cv_probe, [&](int row) {
        const int g = g0 + (row >> lgTW);
        return g < R ? g * W + x0 + (row & (TW - 1)) : -1;
    });
}

template <int AHEAD, int MB>
__global__ __launch_bounds__(512) void k_conv3x3_direct_pp(ConvDesc d, int chunks_per_split, int lgTW)
{
    conv3x3_direct_pp_tile<AHEAD, MB>(d, chunks_per_split, lgTW);
}

// ---------------------------------------------------------------------------------------- direct variant, rolled tap loop
// The unrolled template above keeps every tap's swizzled fragment address in a register for the whole kernel (~130 of its 227 VGPRs):
// two waves per SIMD, whose LDS reads, DMA issue and MFMAs then hardly overlap.  Here the SAME tile algebra for a 256-pixel x 64-filter
// workgroup of 4 waves with the tap loop ROLLED over the three taps of a row (the row loop unrolled: which of a lane's three patch rows a
// tap reads stays a compile-time register choice), the fragment addresses computed at each use (~6 VALU operations per 16-byte read, beside
// 8 MFMAs per k-tile), a 3-stage filter ring (stage = column of the tap) fetched 2 ahead, and ONE patch buffer, re-fetched behind a
// barrier at the chunk boundary: 37 KB of LDS and <= 128 registers -- FOUR workgroups per CU, which hide each other's boundaries and
// whose LDS reads overlap with each other's MFMAs.
template <int kRing, int kAhead>
__device__ __forceinline__ void conv3x3_direct4_tile(const ConvDesc &d, int chunks_per_split, int lgTW)
{
    constexpr int BM = 256, BN = 64, NW = 4;
    constexpr int kPieces = kDirPatchSlots / (64 * NW), kBSlots = 256;
    static_assert(kAhead >= 2 && kAhead <= 9 && kRing >= kAhead + 1, "ring");
    extern __shared__ __attribute__((aligned(1024))) uint4 smem[];
    [[maybe_unused]] const int cv_probe = d.probe;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;
    const int W = d.W, TW = 1 << lgTW, PW = TW + 2, TH = BM >> lgTW, R = d.N * d.H;
    const int ncb = W >> lgTW, rblk = blockIdx.x / ncb, cb = blockIdx.x - rblk * ncb;
    const int g0 = rblk * TH, x0 = cb << lgTW, n0 = blockIdx.y * BN;
    const int cpt = d.Cin / kCvBK;
    const int ch0 = blockIdx.z * chunks_per_split, nch = min(cpt, ch0 + chunks_per_split) - ch0;
    const int patch_px = (TH + 2) * PW;
    constexpr unsigned kOob = 0x80000000u;
    // LDS: [zeros | patch | filter ring]
    uint4 *const s_patch = smem + kDirZeroSlots;
    uint4 *const s_b = smem + kDirZeroSlots + kDirPatchSlots;             // [kRing][kBSlots]
    if (tid < kDirZeroSlots) smem[tid] = make_uint4(0u, 0u, 0u, 0u);
    const auto a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(d.x) - (size_t)(W + 1) * d.Cin, 0,
                                                        (int)(((size_t)d.M + 2 * W + 2) * d.Cin * 2), 0x00020000);
    const auto b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(d.w), 0, (int)((size_t)d.Cout * 9 * d.Cin * 2), 0x00020000);
    unsigned p_vo[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int s = 64 * (wave * kPieces + i) + lane, q = s >> 2, c = (s & 3) ^ ((q >> 2) & 3);
        const int pr = q / PW, pc = q - pr * PW;
        const int g = g0 - 1 + pr, x = x0 - 1 + pc;
        const bool ok = q < patch_px && (unsigned)x < (unsigned)W && (unsigned)g < (unsigned)R;
        p_vo[i] = ok ? (unsigned)((g + 1) * W + x + 1) * (unsigned)(d.Cin * 2) + 16u * c : kOob;
    }
    unsigned b_vo;
    {
        const int s = 64 * wave + lane, r = s >> 2, c = (s & 3) ^ ((r >> 2) & 3);
        const int co = n0 + r;
        b_vo = co < d.Cout ? (unsigned)co * (unsigned)(9 * d.Cin * 2) + 16u * c : kOob;
    }
    const int cin2 = d.Cin * 2;
    auto issue_patch = [&](int piece, int chunk) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void *)(s_patch + 64 * (wave * kPieces + piece)), 16, p_vo[piece],
                                                 (ch0 + chunk) * (kCvBK * 2), 0, 0);
    };
    auto issue_b = [&](int tap, int stage, int chunk) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (__attribute__((address_space(3))) void *)(s_b + stage * kBSlots + 64 * wave), 16, b_vo,
                                                 tap * cin2 + (ch0 + chunk) * (kCvBK * 2), 0, 0);
    };
    int row_of[2], qa[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 64 * wm + 32 * i + (lane & 31), tr = r >> lgTW, tc = r & (TW - 1);
        const int g = g0 + tr, y = g % d.H;
        const bool row_ok = g < R;
        const int q = (tr + 1) * PW + tc + 1;
        row_of[i] = r;
        qa[i][0] = (row_ok && y > 0) ? q - PW : -2;
        qa[i][1] = row_ok ? q : -2;
        qa[i][2] = (row_ok && y < d.H - 1) ? q + PW : -2;
    }
    const int hi = lane >> 5;
    unsigned fb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            fb[ks][j] = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)(s_b + cv_slot(32 * j + (lane & 31), 2 * ks + hi));
    const unsigned pb0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint4 *)s_patch;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 ra[2][2] = {}, rb[2][2] = {};
    // fragments of half step ks: patch pixels q0 / q1 (tap shift applied), filter ring stage `stage`
    auto read = [&](auto KS, int q0, int q1, int stage) {
        constexpr int ks = decltype(KS)::value;
        const int qq[2] = {q0, q1};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = qq[i];
            const unsigned a = pb0 + 16u * (unsigned)(4 * q + ((2 * ks + hi) ^ ((q >> 2) & 3)));
            ra[ks][i] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(a);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            rb[ks][j] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(fb[ks][j] + (unsigned)stage * (kBSlots * 16));
    };
    auto mma = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb[ks][j], ra[ks][i], acc[i][j], 0, 0, 0);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;

    // k-tile t = (chunk t / 9, tap t % 9) lives in ring stage t % kRing; tiles t + 1 .. t + kAhead - 1 are in flight while t is multiplied
    const int T = nch * 9;
    if (nch > 0) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) issue_patch(i, 0);
#pragma unroll
        for (int k = 0; k < kAhead; ++k)
            if (k < T) issue_b(k % 9, k % kRing, k / 9);
        if (T >= kAhead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAhead - 1) : "memory");      // the patch and tile 0 have landed (in order)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                       // (also publishes the zero region)
    if (nch > 0) read(K0{}, qa[0][0] - 1, qa[1][0] - 1, 0);
    int t = 0, st_cur = 0, st_iss = kAhead % kRing, tap_iss = kAhead % 9, ch_iss = kAhead / 9;      // tile being multiplied; stage / (tap, chunk) of tile t + kAhead
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        // the three taps of patch row DY (compile time), rolled over the column dxi = 0, 1, 2 (tap = 3 DY + dxi, ring stage = dxi)
        auto row = [&](auto DY) {
            constexpr int dy = decltype(DY)::value;
#pragma unroll 1
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int tap = 3 * dy + dxi;
                const bool final_tile = !more && tap == 8;
                const int st_next = st_cur + 1 == kRing ? 0 : st_cur + 1;
                read(K1{}, qa[0][dy] + dxi - 1, qa[1][dy] + dxi - 1, st_cur);
                __builtin_amdgcn_sched_barrier(0);
                mma(K0{});
                if (!final_tile) {
                    // k-tile + 1's filters have landed: of the kAhead - 1 tiles in flight the youngest kAhead - 2 may stay (the DMA retires in
                    // order); near the end of the stream fewer are in flight and the count is no measure: wait for all
                    if (t + kAhead - 1 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAhead - 2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (t + kAhead < T) issue_b(tap_iss, st_iss, ch_iss);            // k-tile + kAhead goes into a stage every wave has left
                    if (tap != 8) {
                        constexpr int dyn = dy < 2 ? dy + 1 : 2;
                        const bool wrap = dxi == 2;
                        read(K0{}, wrap ? qa[0][dyn] - 1 : qa[0][dy] + dxi, wrap ? qa[1][dyn] - 1 : qa[1][dy] + dxi, st_next);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(K1{});
                __builtin_amdgcn_sched_barrier(0);
                ++t;
                st_cur = st_next;
                st_iss = st_iss + 1 == kRing ? 0 : st_iss + 1;
                if (++tap_iss == 9) { tap_iss = 0; ++ch_iss; }
            }
        };
        row(std::integral_constant<int, 0>{});
        row(std::integral_constant<int, 1>{});
        row(std::integral_constant<int, 2>{});
        if (more) {
            // every wave is done with the chunk's patch -> the next chunk's -> first fragments of its tap 0 (the CU's other workgroups
            // run their taps meanwhile)
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < kPieces; ++i) issue_patch(i, ch + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            read(K0{}, qa[0][0] - 1, qa[1][0] - 1, st_cur);
        }
    }
    static_assert((size_t)BM * (BN * 2 + 16) <= (size_t)(kDirZeroSlots + kDirPatchSlots + kRing * kBSlots) * 16, "epilogue staging does not fit");
    conv_epilogue<BM, BN, NW>(d, acc, row_of, 0, n0, reinterpret_cast<char *>(smem), tid, lane, cv_probe, [&](int row) {
        const int g = g0 + (row >> lgTW);
        return g < R ? g * W + x0 + (row & (TW - 1)) : -1;
    });
}

template <int kRing, int kAhead, int kOcc>
__global__ __launch_bounds__(256, kOcc) void k_conv3x3_direct4(ConvDesc d, int chunks_per_split, int lgTW)
{
    conv3x3_direct4_tile<kRing, kAhead>(d, chunks_per_split, lgTW);
}

// y = sum over splits of partial + bias (+ res), 8 outputs per thread
__global__ __launch_bounds__(256) void k_conv_reduce(ConvDesc d)
{
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8, total = (size_t)d.M * d.Cout;
    if (e >= total) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
    for (int s = 0; s < d.splits; ++s) {
        const float4 *p = reinterpret_cast<const float4 *>(d.partial + (size_t)s * total + e);
        const float4 a = p[0], b = p[1];
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    const int co = (int)(e % d.Cout);       // Cout is a multiple of 8: the 8 outputs share a pixel
    f16x8 out;
    f16x8 bv, rv;
#pragma unroll
    for (int k = 0; k < 8; ++k) { bv[k] = (_Float16)0.f; rv[k] = (_Float16)0.f; }
    if (d.bias) bv = *reinterpret_cast<const f16x8 *>(d.bias + co);
    if (d.res) rv = *reinterpret_cast<const f16x8 *>(d.res + e);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = (_Float16)(v[k] + (float)bv[k] + (float)rv[k]);
    *reinterpret_cast<f16x8 *>(d.y + e) = out;
}

// ---------------------------------------------------------------------------------------- 128 channels -> <= 4 channels
// The data gradient of the VAE encoder's first convolution (3 -> 128 channels at 256^2: dL/dimage from dL/dfeatures, the library
// spends ~0.5 ms of the 13 ms SDS step in a CK grouped-convolution kernel on it) is a convolution from 128 channels to THREE:
// 1.8 GFLOP over 67 MB -- memory bound, and no shape for a 128-wide MFMA tile.  VALU instead: 16 lanes share a pixel (8
// channels each: one coalesced 256-byte row per tap), the filters of a lane's channels stay in registers for the whole kernel
// (COUT x 9 x 8 halves), v_dot2_f32_f16 accumulates in float32, the 16 partial sums meet in a 4-step butterfly.
template <int COUT>
__global__ __launch_bounds__(256) void k_conv3x3_c128_small(int N, int H, int W, const _Float16 *__restrict__ x, const _Float16 *__restrict__ w,
                                                            _Float16 *__restrict__ y, int iters)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int M = N * H * W;
    h2 wr[COUT][9][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const uint4 v = *reinterpret_cast<const uint4 *>(w + (size_t)(co * 9 + tap) * 128 + 8 * sub);
            const h2 *h = reinterpret_cast<const h2 *>(&v);
#pragma unroll
            for (int k = 0; k < 4; ++k) wr[co][tap][k] = h[k];
        }
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int it = 0; it < iters; ++it) {
        const int p = (wave * iters + it) * 4 + grp;
        if (p >= M) break;                                       // (whole 16-lane groups leave together: the butterfly stays inside a group)
        const int xx0 = p % W, yy0 = (p / W) % H;
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
        uint4 v[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bool ok = (unsigned)(yy0 + dy) < (unsigned)H && (unsigned)(xx0 + dx) < (unsigned)W;
            v[tap] = ok ? *reinterpret_cast<const uint4 *>(x + (size_t)(p + dy * W + dx) * 128 + 8 * sub) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const h2 *h = reinterpret_cast<const h2 *>(&v[tap]);
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[co] = __builtin_amdgcn_fdot2(h[k], wr[co][tap][k], acc[co], false);
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) acc[co] += __shfl_xor(acc[co], m, 16);
        float mine = acc[0];
#pragma unroll
        for (int co = 1; co < COUT; ++co) mine = sub == co ? acc[co] : mine;
        if (sub < COUT) y[(size_t)p * COUT + sub] = (_Float16)mine;
    }
}

// tile configurations (DM4D_CONV_CFG; 3 and 7 are what conv_plan picks, the others are the measured alternatives of
// profiles/r03_zero123.md): 3: 128 x 128, 4 waves of 64 x 64, 4-deep ring | 0 / 10 / 11: the same 3- / 5- / 6-deep | 4 / 6: 128 x 128,
// 8 waves of 32 x 64, 3- / 4-deep | 7: the direct kernel, 256 x 128, 8 waves | 9: the direct kernel, 256 x 64, 4 waves, two per CU |
// 12: the rolled-tap direct kernel, 256 x 64, four per CU | 13 (round 6, the default direct kernel): 7's tile with the two waves of a SIMD half a period apart |
// 14: the same with a 512 x 128 tile (the plan's choice where its grid fills the machine).
// (256 x 128 implicit-GEMM tiles with 64 x 64 or 128 x 64 wave tiles were tried and removed: 256 VGPRs with spills.)
static void cfg_tile(int cfg, int &BM, int &BN)
{
    BM = cfg == 14 ? 512 : (cfg == 7 || cfg == 9 || cfg == 12 || cfg == 13) ? 256 : 128;
    BN = (cfg == 9 || cfg == 12) ? 64 : 128;
}
// (A/B switches, read once: the problem size from which the direct kernel takes the narrow images, the k-tiles a split must keep)
static double direct_gflop() { static const double v = [] { const char *e = getenv("DM4D_CONV_DIRECT_GFLOP"); return e ? atof(e) : 14.0; }(); return v; }
static int split_min_kt() { static const int v = [] { const char *e = getenv("DM4D_CONV_SPLIT_MIN_KT"); return e ? atoi(e) : 30; }(); return v; }
static int conv_plan(int M, int W, int Cout, int kt_total, int &cfg, int &splits)
{
    const char *force = getenv("DM4D_CONV_CFG");          // (A/B switches, read per call: tools/conv_probe.py flips them within a process)
    if (force) cfg = atoi(force);
    // the direct kernel takes the VAE encoder's wide images (W >= 64: 1.3-1.5x the implicit GEMM per shape) and, of the narrow ones, the
    // problems of >= 14 GFLOP.  That threshold is set IN THE STEP (tools/sds_ab.py, DM4D_CONV_DIRECT_GFLOP = 0 / 3 / 7 / 14 / 28 / 1000 on one
    // box: 10.60 / 10.51 / 10.59 / 10.42-10.49 / 10.83 / 10.78 ms per SDS step): timed alone (tools/conv_shapes.py) the split-K implicit GEMM
    // wins up to 28 GFLOP by up to 25 %, but in the step its float32 partial sums and second launch cost more than on an idle, cache-warm chip
    else if (W >= 8 && (W & (W - 1)) == 0 && (W >= 64 || (double)M * Cout * kt_total * kCvBK * 2.0 >= direct_gflop() * 1e9)) {
        // round 6: the ping-pong kernel (13) on every direct shape -- per call -4 ... -17 % against the lock-step 8-wave kernel (7) and
        // -7 / +1 / -6 % against the rolled 4-wave one (12) on its three shapes (tools/conv_cfg_vae.py); in the step 67.1 -> 63.6 ms of direct
        // convolutions per 21 steps under rocprofv3, 9.87 -> 9.78 ms per SDS step (tools/sds_ab.py, two alternating rounds)
        static const int dcfg = [] { const char *e = getenv("DM4D_CONV_DIRECT_CFG"); return e ? atoi(e) : 13; }();      // (A/B switch: 7, 9, 12 or 13)
        cfg = (dcfg == 7 || dcfg == 9 || dcfg == 12) ? dcfg : 13;
        // the rolled-tap variant (four 4-wave workgroups per CU) where the grid gives every CU at least four 256 x 64 tiles: the VAE
        // encoder's 256^2 and 128^2 levels (-12 / -10 / -4 % per call, tools/conv_cfg_vae.py); below that its chunk-boundary patch
        // fetch is exposed and the 8-wave kernel wins (64^2: +6 ... +10 %)
        static const long d4_min = [] { const char *e = getenv("DM4D_CONV_D4_MIN_WGS"); return e ? atol(e) : 1024L; }();      // (A/B switch; 0: never)
        if (cfg == 7 && d4_min > 0 && (long)((M + 255) / 256) * ((Cout + 63) / 64) >= d4_min) cfg = 12;
        // the 512-pixel ping-pong tile where its grid still gives every CU a workgroup (the VAE encoder's 256^2 and 128^2 levels: -8 ... -12 % per call
        // against the 256-pixel one; with half the machine filled, 64^2: +20 ... +40 %)
        static const long p14_min = [] { const char *e = getenv("DM4D_CONV_PP512_MIN_WGS"); return e ? atol(e) : 256L; }();      // (A/B switch; 0: never)
        if (cfg == 13 && p14_min > 0 && W >= 32 && (long)((M + 511) / 512) * ((Cout + 127) / 128) >= p14_min) cfg = 14;
    }
    else cfg = 3;
    int BM, BN;
    cfg_tile(cfg, BM, BN);
    const long tiles = (long)((M + BM - 1) / BM) * ((Cout + BN - 1) / BN);
    splits = 1;
    const char *fs = getenv("DM4D_CONV_SPLITS");
    if (fs) splits = atoi(fs);
    else {
        // split K only until every CU has ONE workgroup, and never below 30 k-tiles per workgroup: the float32 partial sums cost
        // splits x M x C_out x 8 bytes of traffic and a second launch (measured optimum on the UNet's 4^2 .. 16^2 levels,
        // tools/scratch/conv_splits.py: 12 / 24 splits at 4^2, 6 at 8^2, 3 at 16^2, none at 32^2)
        static const int wgs = [] { const char *e = getenv("DM4D_CONV_SPLIT_WGS"); return e ? atoi(e) : 256; }();      // (A/B switch)
        splits = (int)(wgs / tiles);
        if (splits > kt_total / split_min_kt()) splits = kt_total / split_min_kt();
    }
    static const int max_splits = [] { const char *e = getenv("DM4D_CONV_MAX_SPLITS"); return e ? atoi(e) : 64; }();      // (A/B switch)
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    return 0;
}

template <int WM, int WN, int MB, int NB, int kCvStages, int KH = 3, int KW = KH>
static int conv_launch(const ConvDesc &d, hipStream_t st)
{
    constexpr int NW = WM * WN, BM = 32 * MB * WM, BN = 32 * NB * WN;
    constexpr int A_INSTR = (BM * 4 + 64 * NW - 1) / (64 * NW), B_INSTR = (BN * 4 + 64 * NW - 1) / (64 * NW);
    const size_t lds = (size_t)kCvStages * (A_INSTR + B_INSTR) * 64 * NW * 16;
    const dim3 grid((d.M + BM - 1) / BM, (d.Cout + BN - 1) / BN, d.splits);
    static bool attr_set = false;          // (per instantiation; a second thread repeating the call is harmless)
    if (!attr_set) {
        DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3<WM, WN, MB, NB, kCvStages, KH, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv3x3<WM, WN, MB, NB, kCvStages, KH, KW>), grid, dim3(64 * NW), lds, st, d);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" {

// output size: stride 1 (pad 1): in;  stride 2, pad 1: floor((in + 2 - 3) / 2) + 1 = ceil(in / 2);  stride 2, pad 0 with ONE zero
// behind (F.pad(x, (0, 1, 0, 1)) + an unpadded convolution): floor((in + 1 - 3) / 2) + 1 = floor(in / 2)
static inline int conv_out(int in, int stride, int pad) { return stride == 1 ? in : (pad ? (in + 1) / 2 : in / 2); }
// the plan never picks the direct kernel for a strided convolution (its patch is an un-strided block of the image)
static int conv_plan_s(int M, int W, int Cout, int kt_total, int stride, int &cfg, int &splits)
{
    const int rc = conv_plan(M, stride == 1 ? W : 0, Cout, kt_total, cfg, splits);
    if (stride != 1 && (cfg == 7 || cfg == 9 || cfg == 12 || cfg == 13 || cfg == 14)) cfg = 3;
    return rc;
}

size_t dm4d_conv3x3_scratch_bytes(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout)
{
    return dm4d_conv3x3_strided_scratch_bytes(N, H, W, Cin, Cout, 1);
}

size_t dm4d_conv3x3_strided_scratch_bytes(int32_t N, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride)
{
    if (N <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0 || stride < 1 || stride > 2) return 256;
    // the launch plans from the REAL pad: for stride 2 with an odd H or W, pad 0 and pad 1 give different output sizes and
    // possibly different split counts -- the scratch covers the larger of the two plans
    size_t bytes = 256;
    for (int pad = (stride == 1 ? 1 : 0); pad <= 1; ++pad) {
        const int H = conv_out(Hin, stride, pad), W = conv_out(Win, stride, pad);
        if (H <= 0 || W <= 0) continue;
        int cfg, splits;
        conv_plan_s(N * H * W, W, Cout, 9 * Cin / kCvBK, stride, cfg, splits);
        const size_t b = splits > 1 ? (size_t)splits * N * H * W * Cout * 4 + 256 : 256;
        if (b > bytes) bytes = b;
    }
    return bytes;
}

int dm4d_conv3x3_nhwc_f16(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, const void *x, const void *w, const void *bias,
                          const void *residual, void *y, void *scratch, dm4d_stream_t stream)
{
    return dm4d_conv3x3_strided_nhwc_f16(N, H, W, Cin, Cout, 1, 1, x, w, bias, residual, y, scratch, stream);
}

int dm4d_conv3x3_strided_nhwc_f16(int32_t N, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t stride, int32_t pad, const void *x,
                                  const void *w, const void *bias, const void *residual, void *y, void *scratch, dm4d_stream_t stream)
{
    if (N < 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0) { set_error("conv3x3: bad shape"); return DM4D_ERR_INVALID; }
    if (stride < 1 || stride > 2 || pad < 0 || pad > 1 || (stride == 1 && pad != 1)) { set_error("conv3x3: stride 1 / pad 1, or stride 2 / pad 0 or 1 (got %d, %d)", stride, pad); return DM4D_ERR_UNSUPPORTED; }
    const int H = conv_out(Hin, stride, pad), W = conv_out(Win, stride, pad);
    if (H <= 0 || W <= 0) { set_error("conv3x3: empty output"); return DM4D_ERR_INVALID; }
    if (((int64_t)N * Hin * Win + 2 * Win + 2) * Cin * 2 >= 0x7FFF0000LL) { set_error("conv3x3: tensor too large for a 32-bit buffer descriptor"); return DM4D_ERR_UNSUPPORTED; }
    if (N == 0) return DM4D_OK;
    if (Cin % kCvBK != 0 || Cout % 32 != 0) { set_error("conv3x3: C_in and C_out must be multiples of %d (got %d, %d)", kCvBK, Cin, Cout); return DM4D_ERR_UNSUPPORTED; }
    if (((int64_t)N * H * W + 2 * W + 2) * Cin * 2 >= 0x7FFF0000LL || (int64_t)Cout * 9 * Cin * 2 >= 0x7FFF0000LL) { set_error("conv3x3: tensor too large for a 32-bit buffer descriptor"); return DM4D_ERR_UNSUPPORTED; }
    if ((int64_t)N * H * W > 0x7FFFFFFF / 4) { set_error("conv3x3: too many pixels"); return DM4D_ERR_UNSUPPORTED; }
    if (!x || !w || !y) { set_error("conv3x3: null tensor"); return DM4D_ERR_INVALID; }
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual) & 15) != 0) { set_error("conv3x3: tensors must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    ConvDesc d;
    d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.M = N * H * W;
    d.Hin = Hin; d.Win = Win; d.stride = stride; d.pad = pad; d.pad_x = pad;
    d.ostep = 0; d.oy0 = 0; d.ox0 = 0;
    d.x = (const _Float16 *)x; d.w = (const _Float16 *)w; d.bias = (const _Float16 *)bias; d.res = (const _Float16 *)residual;
    d.y = (_Float16 *)y;
    d.kt_total = 9 * Cin / kCvBK;
    d.act = 0;
    d.probe = 0;
#ifdef DM4D_CONV_PROBE
    if (const char *pe = getenv("DM4D_CONV_PROBE")) d.probe = atoi(pe);
#endif
    int cfg;
    conv_plan_s(d.M, W, Cout, d.kt_total, stride, cfg, d.splits);
    if ((cfg == 7 || cfg == 9 || cfg == 12 || cfg == 13 || cfg == 14) && (stride != 1 || pad != 1)) { set_error("conv3x3: the direct kernel takes stride 1, pad 1 only"); return DM4D_ERR_UNSUPPORTED; }
    d.kt_per = (d.kt_total + d.splits - 1) / d.splits;
    d.splits = (d.kt_total + d.kt_per - 1) / d.kt_per;
    d.partial = (float *)scratch;
    if (d.splits > 1 && !scratch) { set_error("conv3x3: this shape needs the split-K scratch (dm4d_conv3x3_scratch_bytes)"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (cfg) {
    case 0: rc = conv_launch<2, 2, 2, 2, 3>(d, st); break;
    case 3: rc = conv_launch<2, 2, 2, 2, 4>(d, st); break;
    case 10: rc = conv_launch<2, 2, 2, 2, 5>(d, st); break;      // 128 x 128, 4 waves, 5-deep (80 KB: two workgroups per CU)
    case 11: rc = conv_launch<2, 2, 2, 2, 6>(d, st); break;      // ... 6-deep (96 KB: one workgroup per CU)
    case 4: rc = conv_launch<4, 2, 1, 2, 3>(d, st); break;
    case 6: rc = conv_launch<4, 2, 1, 2, 4>(d, st); break;
    case 7: case 9: case 12: case 13: case 14: {      // direct: 256 pixels x 128 (7) / 64 (9) filters, the input patch resident in LDS for the nine taps
        const int W_ = d.W;
        if (W_ < 8 || (W_ & (W_ - 1)) != 0) { set_error("conv3x3 direct: W must be a power of two >= 8"); return DM4D_ERR_UNSUPPORTED; }
        int lgTW = 3;
        while ((1 << lgTW) < W_ && lgTW < 5) ++lgTW;                       // TW = min(W, 32)
        if (cfg == 14 && W_ < 32) { set_error("conv3x3 direct (512-pixel tile): W must be a power of two >= 32"); return DM4D_ERR_UNSUPPORTED; }
        const int TH = (cfg == 14 ? 512 : 256) >> lgTW, R = d.N * d.H, cpt = d.Cin / kCvBK;
        const int chunks_per_split = (cpt + d.splits - 1) / d.splits;
        d.splits = (cpt + chunks_per_split - 1) / chunks_per_split;
        const unsigned tiles_m = (unsigned)(((R + TH - 1) / TH) * (W_ >> lgTW));
        if (cfg == 14) {        // 8 waves, 512 x 128, ping-pong
            const size_t lds = (size_t)pp_lds_slots<4>() * 16;
            static bool attr_set = false;
            if (!attr_set) { DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3_direct_pp<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
            hipLaunchKernelGGL((k_conv3x3_direct_pp<4, 4>), dim3(tiles_m, (d.Cout + 127) / 128, d.splits), dim3(512), lds, st, d, chunks_per_split, lgTW);
        } else if (cfg == 13) {        // 8 waves, 256 x 128, the two waves of a SIMD half a period apart (ping-pong)
            const size_t lds = (size_t)pp_lds_slots<2>() * 16;
            static bool attr_set = false;
            if (!attr_set) { DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3_direct_pp<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
            hipLaunchKernelGGL((k_conv3x3_direct_pp<4, 2>), dim3(tiles_m, (d.Cout + 127) / 128, d.splits), dim3(512), lds, st, d, chunks_per_split, lgTW);
        } else if (cfg == 7) {         // 8 waves, 256 x 128, one workgroup per CU
            const size_t lds = (size_t)dir_lds_slots<2, 9>() * 16;
            static bool attr_set = false;
            if (!attr_set) { DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3_direct<2, 9, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
            hipLaunchKernelGGL((k_conv3x3_direct<2, 9, 4>), dim3(tiles_m, (d.Cout + 127) / 128, d.splits), dim3(512), lds, st, d, chunks_per_split, lgTW);
        } else if (cfg == 12) { // 4 waves, 256 x 64, rolled tap loop, one patch buffer: four workgroups per CU
            const size_t lds = (size_t)(kDirZeroSlots + kDirPatchSlots + 3 * 256) * 16;
            static bool attr_set = false;
            if (!attr_set) { DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3_direct4<3, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
            hipLaunchKernelGGL((k_conv3x3_direct4<3, 2, 4>), dim3(tiles_m, (d.Cout + 63) / 64, d.splits), dim3(256), lds, st, d, chunks_per_split, lgTW);
        } else {                // 4 waves, 256 x 64, two workgroups per CU (one fills the bubble around the other's barrier)
            const size_t lds = (size_t)dir_lds_slots<1, 3>() * 16;
            static bool attr_set = false;
            if (!attr_set) { DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3_direct<1, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
            hipLaunchKernelGGL((k_conv3x3_direct<1, 3, 2>), dim3(tiles_m, (d.Cout + 63) / 64, d.splits), dim3(256), lds, st, d, chunks_per_split, lgTW);
        }
        DM4D_HIP_CHECK(hipGetLastError());
        rc = DM4D_OK;
        break;
    }
    default: set_error("conv3x3: bad configuration %d", cfg); return DM4D_ERR_INVALID;
    }
    if (rc) return rc;
    if (d.splits > 1) {
        const size_t total = (size_t)d.M * Cout;
        hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, st, d);
        DM4D_HIP_CHECK(hipGetLastError());
    }
    return DM4D_OK;
}

// ---------------------------------------------------------------------------------------- linear layers (the same kernel, one tap)
// y = act(x w^T + bias) (+ residual): x [M][K], w [N][K] (an nn.Linear weight / a 1x1 convolution's filter as it lies), float16,
// float32 accumulation -- the transformer blocks' projections and the 1x1 convolutions of the Zero123 UNet (zero123.py), with
// what the library GEMM leaves to extra launches in the epilogue: the residual add, and GEGLU (attention.py:48-56) for the
// feed-forward's first projection.  Tiles: 128 x 128 (4 waves of 64 x 64) or, when that leaves most CUs idle, 64 x 64 (4 waves of
// 32 x 32); split K by the convolution's rule.
static void linear_plan(int64_t M, int N, int kt_total, int act, int &cfg, int &splits)
{
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    // (tools/linear_tune.py: the 64 x 64 tiles win while the 128 x 128 ones would leave CUs idle -- unless K is long, where the larger
    // tile's fewer operand fetches matter more)
    cfg = (act != 1 && tiles128 < 200 && (kt_total <= 24 || (tiles128 < 96 && !(tiles128 >= 32 && kt_total >= 128)))) ? 13 : 3;      // (K >= 4096 on >= 32 tiles: 128 x 128 + split K)
    if (const char *force = getenv("DM4D_LIN_CFG")) { const int f = atoi(force); if (act != 1 || f == 3) cfg = f; }
    const int B = cfg == 13 ? 64 : 128;
    const long tiles = (long)((M + B - 1) / B) * ((N + B - 1) / B);
    splits = (int)(256 / tiles);
    if (splits > kt_total / 30) splits = kt_total / 30;
    if (const char *fs = getenv("DM4D_LIN_SPLITS")) splits = atoi(fs);
    if (splits < 1 || act == 1) splits = 1;             // (the reduction kernel has no GEGLU)
    if (splits > 64) splits = 64;
}

size_t dm4d_linear_scratch_bytes(int64_t M, int32_t K, int32_t N)
{
    if (M <= 0 || K <= 0 || N <= 0) return 256;
    int cfg, splits;
    linear_plan(M, N, K / kCvBK, 0, cfg, splits);
    return splits > 1 ? (size_t)splits * M * N * 4 + 256 : 256;
}

int dm4d_linear_f16(int64_t M, int32_t K, int32_t N, const void *x, const void *w, const void *bias, const void *residual, void *y,
                    int32_t act, void *scratch, dm4d_stream_t stream)
{
    if (M < 0 || K <= 0 || N <= 0) { set_error("linear: bad shape"); return DM4D_ERR_INVALID; }
    if (act != 0 && act != 1) { set_error("linear: act must be 0 (none) or 1 (GEGLU), got %d", act); return DM4D_ERR_INVALID; }
    if (M == 0) return DM4D_OK;
    if (K % kCvBK != 0 || N % 8 != 0) { set_error("linear: K must be a multiple of %d and N of 8 (got %d, %d)", kCvBK, K, N); return DM4D_ERR_UNSUPPORTED; }
    if (act == 1 && (N % 128 != 0 || residual)) { set_error("linear: GEGLU needs N %% 128 == 0 (interleaved value / gate blocks of 64) and no residual"); return DM4D_ERR_UNSUPPORTED; }
    if ((3 * M + 2) * K * 2 >= 0x7FFF0000LL || (int64_t)N * K * 2 >= 0x7FFF0000LL || M > 0x7FFFFFFF / 4) { set_error("linear: tensor too large for a 32-bit buffer descriptor"); return DM4D_ERR_UNSUPPORTED; }
    if (!x || !w || !y) { set_error("linear: null tensor"); return DM4D_ERR_INVALID; }
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual) & 15) != 0) { set_error("linear: tensors must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    ConvDesc d;
    d.N = 1; d.H = 1; d.W = (int)M; d.Cin = K; d.Cout = N; d.M = (int)M;
    d.Hin = 1; d.Win = (int)M; d.stride = 1; d.pad = 0; d.pad_x = 0;
    d.ostep = 0; d.oy0 = 0; d.ox0 = 0;
    d.x = (const _Float16 *)x; d.w = (const _Float16 *)w; d.bias = (const _Float16 *)bias; d.res = (const _Float16 *)residual;
    d.y = (_Float16 *)y;
    d.kt_total = K / kCvBK;
    d.act = act;
    d.probe = 0;
    int cfg;
    linear_plan(M, N, d.kt_total, act, cfg, d.splits);
    d.kt_per = (d.kt_total + d.splits - 1) / d.splits;
    d.splits = (d.kt_total + d.kt_per - 1) / d.kt_per;
    d.partial = (float *)scratch;
    if (d.splits > 1 && !scratch) { set_error("linear: this shape needs the split-K scratch (dm4d_linear_scratch_bytes)"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (cfg) {
    case 3: rc = conv_launch<2, 2, 2, 2, 4, 1>(d, st); break;
    case 13: rc = conv_launch<2, 2, 1, 1, 4, 1>(d, st); break;
    default: set_error("linear: bad configuration %d", cfg); return DM4D_ERR_INVALID;
    }
    if (rc) return rc;
    if (d.splits > 1) {
        const size_t total = (size_t)d.M * N;
        hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, st, d);
        DM4D_HIP_CHECK(hipGetLastError());
    }
    return DM4D_OK;
}

// ---------------------------------------------------------------------------------------- data gradient of the stride-2 convolution
// y[n][oy][ox] = sum_{ky, kx} x[n][2 oy + ky][2 ox + kx] w[ky][kx] (pad 0 + one zero behind each axis: the VAE encoder's Downsample,
// model.py:85-100) -- dL/dx[iy][ix] gets, per axis, the taps k with (i - k) even: k in {0, 2} for an even coordinate (dy at i / 2 and
// i / 2 - 1), k = 1 for an odd one.  So the four parity classes (iy & 1, ix & 1) of the input pixels are four STRIDE-1 convolutions
// of dy with 2 x 2, 2 x 1, 1 x 2 and 1 x 1 filters (padding 1 on a 2-tap axis), each writing every second pixel of every second
// row of dx: 4 + 2 + 2 + 1 = the forward's nine taps, on the implicit-GEMM kernel (the library's transposed convolution of the
// padded shape took 100-170 us with its layout copies where the forward takes 36; profiles/r03_zero123.md).
// w_cls[c]: class c = 2 (iy & 1) + (ix & 1), [C_in][KH][KW][C_out] float16 with tap (ty, tx) = w[:, :, 2 - 2 ty (KH = 2) or 1, 2 - 2 tx or 1]
// transposed (conv_mfma.pack_weight_s2_dgrad).
int dm4d_conv3x3_s2_dgrad_nhwc_f16(int32_t N, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, const void *dy, const void *const *w_cls,
                                   void *dx, dm4d_stream_t stream)
{
    if (N < 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0) { set_error("s2 dgrad: bad shape"); return DM4D_ERR_INVALID; }
    if ((Hin | Win) & 1) { set_error("s2 dgrad: H_in and W_in must be even (got %d x %d)", Hin, Win); return DM4D_ERR_UNSUPPORTED; }
    if (Cout % kCvBK != 0 || Cin % 32 != 0) { set_error("s2 dgrad: C_in and C_out must be multiples of 32 (got %d, %d)", Cin, Cout); return DM4D_ERR_UNSUPPORTED; }
    if (N == 0) return DM4D_OK;
    const int H = Hin / 2, W = Win / 2;
    if (((int64_t)N * H * W + 2 * W + 2) * Cout * 2 >= 0x7FFF0000LL || (int64_t)Cin * 4 * Cout * 2 >= 0x7FFF0000LL || (int64_t)N * Hin * Win > 0x7FFFFFFF / 4) { set_error("s2 dgrad: tensor too large for a 32-bit buffer descriptor"); return DM4D_ERR_UNSUPPORTED; }
    if (!dy || !w_cls || !dx || !w_cls[0] || !w_cls[1] || !w_cls[2] || !w_cls[3]) { set_error("s2 dgrad: null tensor"); return DM4D_ERR_INVALID; }
    if ((((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)w_cls[0] | (uintptr_t)w_cls[1] | (uintptr_t)w_cls[2] | (uintptr_t)w_cls[3]) & 15) != 0) { set_error("s2 dgrad: tensors must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    S2DgradDesc a;
    ConvDesc &d = a.d;
    d.N = N; d.H = H; d.W = W; d.Cin = Cout; d.Cout = Cin; d.M = N * H * W;       // (the operator's input is dy: its channels are the forward's C_out)
    d.Hin = H; d.Win = W; d.stride = 1; d.pad = 1; d.pad_x = 1;                    // (pad, output offset, filter, k-tiles: per class, in the kernel)
    d.ostep = 2; d.oy0 = 0; d.ox0 = 0;
    d.x = (const _Float16 *)dy; d.w = nullptr; d.bias = nullptr; d.res = nullptr; d.y = (_Float16 *)dx;
    d.partial = nullptr; d.splits = 1; d.kt_total = d.kt_per = 0;
    d.act = 0; d.probe = 0;
    for (int c = 0; c < 4; ++c) a.w[c] = (const _Float16 *)w_cls[c];
    constexpr size_t lds = (size_t)4 * (2 + 2) * 256 * 16;                          // the 128 x 128 x 4-deep ring of conv_launch<2, 2, 2, 2, 4>
    static bool attr_set = false;
    if (!attr_set) { DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv_s2_dgrad<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
    hipLaunchKernelGGL(k_conv_s2_dgrad<0>, dim3((d.M + 127) / 128, (Cin + 127) / 128, 4), dim3(256), lds, st, a);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

int dm4d_conv3x3_c128_small_nhwc_f16(int32_t N, int32_t H, int32_t W, int32_t Cout, const void *x, const void *w, void *y, dm4d_stream_t stream)
{
    if (N < 0 || H <= 0 || W <= 0) { set_error("conv3x3_c128_small: bad shape"); return DM4D_ERR_INVALID; }
    if (Cout < 1 || Cout > 4) { set_error("conv3x3_c128_small: C_out must be 1 .. 4 (got %d)", Cout); return DM4D_ERR_UNSUPPORTED; }
    if (N == 0) return DM4D_OK;
    if ((int64_t)N * H * W > 0x7FFFFFFF / 4) { set_error("conv3x3_c128_small: too many pixels"); return DM4D_ERR_UNSUPPORTED; }
    if (!x || !w || !y) { set_error("conv3x3_c128_small: null tensor"); return DM4D_ERR_INVALID; }
    if ((((uintptr_t)x | (uintptr_t)w) & 15) != 0 || ((uintptr_t)y & 1) != 0) { set_error("conv3x3_c128_small: x, w must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    const int M = N * H * W, iters = 16;
    const int waves = (M + 4 * iters - 1) / (4 * iters);
    const dim3 grid((waves + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    const _Float16 *xp = (const _Float16 *)x, *wp = (const _Float16 *)w;
    _Float16 *yp = (_Float16 *)y;
    switch (Cout) {
    case 1: hipLaunchKernelGGL(k_conv3x3_c128_small<1>, grid, block, 0, st, N, H, W, xp, wp, yp, iters); break;
    case 2: hipLaunchKernelGGL(k_conv3x3_c128_small<2>, grid, block, 0, st, N, H, W, xp, wp, yp, iters); break;
    case 3: hipLaunchKernelGGL(k_conv3x3_c128_small<3>, grid, block, 0, st, N, H, W, xp, wp, yp, iters); break;
    default: hipLaunchKernelGGL(k_conv3x3_c128_small<4>, grid, block, 0, st, N, H, W, xp, wp, yp, iters); break;
    }
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // extern "C"
