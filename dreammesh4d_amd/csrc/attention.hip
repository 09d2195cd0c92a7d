// attention.hip -- softmax(q k^T / sqrt(d)) v for the self-attentions of the Zero123 UNet on the gfx950 matrix cores.
//
// Where it sits: BasicTransformerBlock.attn1 of every SpatialTransformer (extern/ldm_zero123/modules/attention.py:152-194; 16 per
// UNet forward, batch 8, 8 heads of d = C / 8 = 40 / 80 / 160 channels over L = 1024 / 256 / 64 tokens).  The library's flash
// kernel spends 77 us per call at L = 1024 (139 TFLOP/s of the nominal work: 0.31 ms of the 6 ms UNet forward).
//
// One workgroup = NW waves x 32 queries of one (batch, head); keys / values arrive in tiles of 64 (register-staged: the next
// tile's global loads are in flight while the current one is multiplied), K as [key][d] rows, V TRANSPOSED on its way into LDS
// ([d][key]: the only layout from which its MFMA operand is k-contiguous).  Per tile and wave, with v_mfma_f32_32x32x16_f16:
//   S^T[key][q] = K Q^T      (A = K rows, B = the wave's Q fragment, held in registers for the whole kernel): a lane ends up with
//                            ONE query (column lane & 31) and 16 of each 32 keys in its registers -- the softmax statistics of a
//                            query are two lanes' worth of register arithmetic + one exchange with lane ^ 32 per tile, and
//   O^T[d][q] += V^T P^T     takes P^T as its B operand STRAIGHT from those registers: the accumulator registers 8u .. 8u + 7 of a
//                            lane are keys {16u + 4h .. + 3} and {16u + 8 + 4h .. + 3} of the tile (h = lane >> 5), which is a legal
//                            choice of the 8 k-slots of that lane as long as the A operand (V^T) enumerates the keys the same way:
//                            two 8-byte reads instead of one 16-byte read.  No transposition of P, no trip through LDS.
// Online softmax in the exp2 domain (scale log2(e) folded into the exponent's FMA), float32 statistics and accumulators, float16
// probabilities (as the library's kernel).  d is padded to multiples of 16 (QK^T) / 32 (O^T rows) with zeros inside the kernel.
#include "common.h"
#include "../../include/dm4d.h"

namespace dm4d {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16a __attribute__((ext_vector_type(16)));

struct AttnArgs {
    int B, L, heads;
    const _Float16 *q, *k, *v;       // element (b, token, head, c) at base + b * batch_stride + token * tok_stride + head * D + c
    long batch_stride, tok_stride;
    _Float16 *out;                   // [B][L][heads * D]
    float scale_log2e;               // softmax scale x log2(e)
};

template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void k_attention(AttnArgs a)
{
    constexpr int KS = (D + 15) / 16;            // k-steps of QK^T (d padded to 16 KS with zeros)
    constexpr int MT = (D + 31) / 32;            // 32-row tiles of O^T
    constexpr int PK = D / 8;                    // real 16-byte pieces of a K / V row
    constexpr int KPITCH = (2 * KS + 1) * 16;    // bytes of a K row in LDS (+ one piece: conflict-free fragment reads)
    constexpr int VPITCH = 64 * 2 + 8;           // bytes of a V^T row (64 keys) in LDS
    constexpr int NT = 64 * NW;
    constexpr int NP = (64 * PK + NT - 1) / NT;  // pieces of a K (or V) tile per thread
    static_assert(D % 8 == 0 && D <= 160, "head dimension");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const s_k = smem;                                      // [2][64][KPITCH]
    char *const s_v = smem + 2 * 64 * KPITCH;                    // [2][32 MT][VPITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y, q0 = (blockIdx.x * NW + wave) * 32;
    const _Float16 *qp = a.q + (size_t)b * a.batch_stride + (size_t)head * D;
    const _Float16 *kp = a.k + (size_t)b * a.batch_stride + (size_t)head * D;
    const _Float16 *vp = a.v + (size_t)b * a.batch_stride + (size_t)head * D;

    // zero the padding of both K buffers (pieces PK .. 2 KS of every row) and the V^T rows D .. 32 MT - 1: never written again
    for (int e = tid; e < 2 * 64 * (2 * KS + 1 - PK); e += NT) {          // (>= 1 piece per row: the pitch's extra one)
        const int row = e / (2 * KS + 1 - PK), pc = PK + e % (2 * KS + 1 - PK);
        *reinterpret_cast<uint4 *>(s_k + (size_t)row * KPITCH + 16 * pc) = make_uint4(0u, 0u, 0u, 0u);
    }
    if constexpr (32 * MT > D) {
        constexpr int kPad = (32 * MT - D) * (VPITCH / 8);
        for (int e = tid; e < 2 * kPad; e += NT) {
            const int buf = e / kPad, r = e % kPad;
            *reinterpret_cast<uint2 *>(s_v + (size_t)buf * 32 * MT * VPITCH + (size_t)D * VPITCH + 8 * r) = make_uint2(0u, 0u);
        }
    }

    // the wave's Q fragment: lane = query col, k-slots 8 half .. + 7 of every k-step
    h16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int pc = 2 * s + half;
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[s][i] = (_Float16)0.f;
        if (pc < PK && q0 + col < a.L) qf[s] = *reinterpret_cast<const h16x8 *>(qp + (size_t)(q0 + col) * a.tok_stride + 8 * pc);
    }

    // register staging of a tile: piece e = tid + NT i of K and of V (key e / PK, piece e % PK)
    h16x8 rk[NP], rv[NP];
    auto fetch = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + NT * i;
            if (e < 64 * PK) {
                const int key = key0 + e / PK, pc = e % PK;
                const size_t off = (size_t)key * a.tok_stride + 8 * pc;
                rk[i] = *reinterpret_cast<const h16x8 *>(kp + off);
                rv[i] = *reinterpret_cast<const h16x8 *>(vp + off);
            }
        }
    };
    auto stash = [&](int buf) {
        char *dk = s_k + (size_t)buf * 64 * KPITCH, *dv = s_v + (size_t)buf * 32 * MT * VPITCH;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + NT * i;
            if (e < 64 * PK) {
                const int key = e / PK, pc = e % PK;
                *reinterpret_cast<h16x8 *>(dk + (size_t)key * KPITCH + 16 * pc) = rk[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<_Float16 *>(dv + (size_t)(8 * pc + j) * VPITCH + 2 * key) = rv[i][j];
            }
        }
    };

    f32x16a o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[m][r] = 0.f;
    float mrun = -1.0e30f, lrun = 0.f;            // running maximum (raw scores) of the lane's query, the lane's share of the sum

    const int ntiles = a.L / 64;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) fetch(64 * (t + 1));
        const char *bk = s_k + (size_t)buf * 64 * KPITCH, *bv = s_v + (size_t)buf * 32 * MT * VPITCH;
        // S^T: two tiles of 32 keys x the wave's 32 queries
        f32x16a s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const h16x8 kf = *reinterpret_cast<const h16x8 *>(bk + (size_t)(32 * kt + col) * KPITCH + 16 * (2 * ks + half));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kt], 0, 0, 0);
            }
        }
        // online softmax of the lane's query over these 64 keys (32 of them here, 32 in lane ^ 32)
        float mx = s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * a.scale_log2e);
        const float mc = -mnew * a.scale_log2e;
        mrun = mnew;
        float psum = 0.f;
        h16x8 pf[2][2];                           // P^T fragments: [key tile][16-key step]
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], a.scale_log2e, mc));
                psum += p;
                pf[kt][r >> 3][r & 7] = (_Float16)p;
            }
        lrun = lrun * alpha + psum;
        // (after the first few tiles a query's maximum rarely moves: alpha is exactly 1 in every lane, and the wave skips the rescale)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[m][r] *= alpha;
        }
        // O^T += V^T P^T: the keys of k-step (kt, u) in the order the P^T registers hold them
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const char *row = bv + (size_t)(32 * m + col) * VPITCH + 2 * (32 * kt + 16 * u + 4 * half);
                    const h16x4 lo = *reinterpret_cast<const h16x4 *>(row), hi = *reinterpret_cast<const h16x4 *>(row + 16);
                    h16x8 vf;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { vf[i] = lo[i]; vf[4 + i] = hi[i]; }
                    o[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kt][u], o[m], 0, 0, 0);
                }
        if (t + 1 < ntiles) stash(buf ^ 1);
        __syncthreads();
    }
    // normalise and store: the lane holds rows d = 32 m + 8 j + 4 half + i of column q
    const float ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.0f / ltot;
    if (q0 + col < a.L) {
        _Float16 *op = a.out + ((size_t)b * a.L + q0 + col) * ((size_t)a.heads * D) + (size_t)head * D;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d0 = 32 * m + 8 * j + 4 * half;
                if (d0 < D) {
                    h16x4 w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = (_Float16)(o[m][4 * j + i] * inv);
                    *reinterpret_cast<h16x4 *>(op + d0) = w;
                }
            }
    }
}

template <int D, int NW>
static int attn_launch(const AttnArgs &a, hipStream_t st)
{
    constexpr int KS = (D + 15) / 16, MT = (D + 31) / 32;
    constexpr size_t lds = 2 * 64 * (size_t)((2 * KS + 1) * 16) + 2 * 32 * MT * (size_t)(64 * 2 + 8);
    static bool attr_set = false;
    if (!attr_set) { DM4D_HIP_CHECK(hipFuncSetAttribute((const void *)k_attention<D, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
    hipLaunchKernelGGL((k_attention<D, NW>), dim3((a.L + 32 * NW - 1) / (32 * NW), a.heads, a.B), dim3(64 * NW), lds, st, a);
    DM4D_HIP_CHECK(hipGetLastError());
    return DM4D_OK;
}

}  // namespace dm4d

using namespace dm4d;

extern "C" int dm4d_attention_f16(int32_t B, int32_t L, int32_t heads, int32_t D, const void *q, const void *k, const void *v,
                                  int64_t batch_stride, int64_t tok_stride, void *out, float scale, dm4d_stream_t stream)
{
    if (B < 0 || L <= 0 || heads <= 0 || D <= 0) { set_error("attention: bad shape"); return DM4D_ERR_INVALID; }
    if (D != 40 && D != 80 && D != 160 && D != 64) { set_error("attention: head dimension %d (40, 64, 80 or 160)", D); return DM4D_ERR_UNSUPPORTED; }
    if (L % 64 != 0) { set_error("attention: L = %d must be a multiple of 64", L); return DM4D_ERR_UNSUPPORTED; }
    if (B == 0) return DM4D_OK;
    if (!q || !k || !v || !out) { set_error("attention: null tensor"); return DM4D_ERR_INVALID; }
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) != 0 || tok_stride % 8 != 0 || batch_stride % 8 != 0) { set_error("attention: tensors and strides must be 16-byte aligned"); return DM4D_ERR_INVALID; }
    AttnArgs a{B, L, heads, (const _Float16 *)q, (const _Float16 *)k, (const _Float16 *)v, (long)batch_stride, (long)tok_stride, (_Float16 *)out,
               scale * 1.44269504088896340736f};
    hipStream_t st = (hipStream_t)stream;
    const bool small = L < 128;
    switch (D) {
    case 40: return small ? attn_launch<40, 2>(a, st) : attn_launch<40, 4>(a, st);
    case 64: return small ? attn_launch<64, 2>(a, st) : attn_launch<64, 4>(a, st);
    case 80: return small ? attn_launch<80, 2>(a, st) : attn_launch<80, 4>(a, st);
    default: return small ? attn_launch<160, 2>(a, st) : attn_launch<160, 4>(a, st);
    }
}
