// common.h -- device helpers shared by the gfx950 kernels of libdm4d_hip.so.
// Compiled with -ffp-contract=off: FMAs appear only where __builtin_fmaf is written
// (the "arithmetic contract" of DESIGN.md that makes tile keys, radii, n_contrib and the
// forward image bit-comparable with the CPU checker).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DM4D_WAVE 64

#define DM4D_HIP_CHECK(expr)                                                         \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            dm4d::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                            __FILE__, __LINE__);                                     \
            return DM4D_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)

namespace dm4d {

void set_error(const char *fmt, ...);

// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
enum KernelId { kKPreprocess = 0, kKColscan, kKScatter, kKTileSort, kKRenderFwd, kKRenderBwd, kKGatherBwd,
                kKSkinFwd, kKSkinBwd, kKFaceFwd, kKFaceBwd, kKKnn, kKernelCount };
struct ProfScope {
    int id;
    hipStream_t st;
    void *slot;
    ProfScope(int id, hipStream_t st);
    ~ProfScope();
};

// fire-and-forget float add in LDS (ds_add_f32, no return value)
__device__ __forceinline__ void lds_fadd(float *p, float v)
{
#if defined(DM4D_LDS_ACC_STORE)
    *p = v;          // timing experiment only
#elif defined(DM4D_LDS_ACC_NONE)
    if (v == 123.456f) *p = v;
#else
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
#endif
}

__device__ __forceinline__ float as_f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __float_as_uint(f); }

// exp(x) for x <= 0, deterministic: 2^f minimax degree 6 on [-0.5, 0.5], <= 1.4 ulp.  Round 4: n = rint(x log2 e) as
// t = fma(x, log2e_hi, 1.5 * 2^23), n = t - 1.5 * 2^23 (one v_fma + one v_add instead of v_mul + v_rndne), and ldexp(p, n) as
// bits(p) + (bits(t) << 23) (one v_lshl_add_u32 instead of v_cvt_i32_f32 + v_ldexp_f32): -3 of the blend forward's ~37 VALU
// instructions per (entry, pixel) pair.  The SAME function, operation for operation, as oracle/raster_oracle.c::dm4d_expf.
__device__ __forceinline__ float det_expf(float x)
{
    const float L2E_HI = 0x1.715476p+0f;
    const float L2E_LO = 0x1.4ae0c0p-26f;
    const float MAGIC = 12582912.0f;
    x = fmaxf(x, -86.0f);
    const float t = __builtin_fmaf(x, L2E_HI, MAGIC);
    const float n = t - MAGIC;
    float f = __builtin_fmaf(x, L2E_HI, -n);
    f = __builtin_fmaf(x, L2E_LO, f);
    float p = 0x1.446c7ep-13f;
    p = __builtin_fmaf(p, f, 0x1.5f48c8p-10f);
    p = __builtin_fmaf(p, f, 0x1.3b29d8p-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aeccp-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbe0p-3f);
    p = __builtin_fmaf(p, f, 0x1.62e430p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(t) << 23));
}

// U independent det_expf evaluations written step-by-step across the U values, so that the instruction stream
// itself interleaves the U dependency chains (a lone wave issues dependent VALU ops far slower than
// independent ones).  Bit-identical to det_expf per element.
template <int U>
__device__ __forceinline__ void det_expf_n(const float (&xin)[U], float (&out)[U])
{
    const float L2E_HI = 0x1.715476p+0f;
    const float L2E_LO = 0x1.4ae0c0p-26f;
    const float MAGIC = 12582912.0f;
    float x[U], t[U], n[U], f[U], p[U];
#pragma unroll
    for (int j = 0; j < U; ++j) x[j] = fmaxf(xin[j], -86.0f);
#pragma unroll
    for (int j = 0; j < U; ++j) t[j] = __builtin_fmaf(x[j], L2E_HI, MAGIC);
#pragma unroll
    for (int j = 0; j < U; ++j) n[j] = t[j] - MAGIC;
#pragma unroll
    for (int j = 0; j < U; ++j) f[j] = __builtin_fmaf(x[j], L2E_HI, -n[j]);
#pragma unroll
    for (int j = 0; j < U; ++j) f[j] = __builtin_fmaf(x[j], L2E_LO, f[j]);
#pragma unroll
    for (int j = 0; j < U; ++j) p[j] = __builtin_fmaf(0x1.446c7ep-13f, f[j], 0x1.5f48c8p-10f);
#pragma unroll
    for (int j = 0; j < U; ++j) p[j] = __builtin_fmaf(p[j], f[j], 0x1.3b29d8p-7f);
#pragma unroll
    for (int j = 0; j < U; ++j) p[j] = __builtin_fmaf(p[j], f[j], 0x1.c6aeccp-5f);
#pragma unroll
    for (int j = 0; j < U; ++j) p[j] = __builtin_fmaf(p[j], f[j], 0x1.ebfbe0p-3f);
#pragma unroll
    for (int j = 0; j < U; ++j) p[j] = __builtin_fmaf(p[j], f[j], 0x1.62e430p-1f);
#pragma unroll
    for (int j = 0; j < U; ++j) p[j] = __builtin_fmaf(p[j], f[j], 1.0f);
#pragma unroll
    for (int j = 0; j < U; ++j) out[j] = __uint_as_float(__float_as_uint(p[j]) + (__float_as_uint(t[j]) << 23));
}

__device__ __forceinline__ int f2i_sat(float v)
{
    if (!(v > -1073741824.0f)) return -1073741824;
    if (v > 1073741824.0f) return 1073741824;
    return (int)v;
}

// ---- wave64 cross-lane helpers (DPP; gfx9 row_bcast forms) -------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v)
{
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(r);
}
// Sum over the 64 lanes; the total is valid in lanes 48..63 (row 3).
__device__ __forceinline__ float wave_sum_row3(float v)
{
    v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);       // row_half_mirror
    v = dpp_add<0x140>(v);       // row_mirror  -> every lane holds its row's sum
    v = dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast31 into rows 2,3
    return v;
}
// Wave-wide integer reductions, every lane (uniform result): four DPP row rotations leave each row's result in all
// of its lanes, the four rows are combined on the scalar unit -- no trips through the LDS crossbar (__shfl_xor).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
#define DM4D_ROW_ALLREDUCE(OP)                       \
    v = OP(v, dpp_u32<0x121>(v)); /* row_ror:1 */    \
    v = OP(v, dpp_u32<0x122>(v)); /* row_ror:2 */    \
    v = OP(v, dpp_u32<0x124>(v)); /* row_ror:4 */    \
    v = OP(v, dpp_u32<0x128>(v)); /* row_ror:8 */
#define DM4D_ROWS_COMBINE(OP)                                                                         \
    OP(OP((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)), \
       OP((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)))
__device__ __forceinline__ uint32_t op_add_u32(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t op_max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t op_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    DM4D_ROW_ALLREDUCE(op_add_u32)
    return DM4D_ROWS_COMBINE(op_add_u32);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    DM4D_ROW_ALLREDUCE(op_max_u32)
    return DM4D_ROWS_COMBINE(op_max_u32);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    DM4D_ROW_ALLREDUCE(op_min_u32)
    return DM4D_ROWS_COMBINE(op_min_u32);
}
// maximum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ uint32_t row_allmax_u32(uint32_t v)
{
    DM4D_ROW_ALLREDUCE(op_max_u32)
    return v;
}
#undef DM4D_ROW_ALLREDUCE
#undef DM4D_ROWS_COMBINE
// inclusive prefix sum across the wave
// Six DPP adds (row shifts by 1, 2, 4, 8 with zero fill, then lane 15 / lane 31 broadcast into the following rows),
// all on the VALU: the __shfl_up version this replaces went through the LDS crossbar six times, each a dependent
// round trip.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int /*lane*/)
{
#define DM4D_DPP_ADD(CTRL, ROWMASK) \
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false)
    DM4D_DPP_ADD(0x111, 0xf);   // row_shr:1
    DM4D_DPP_ADD(0x112, 0xf);   // row_shr:2
    DM4D_DPP_ADD(0x114, 0xf);   // row_shr:4
    DM4D_DPP_ADD(0x118, 0xf);   // row_shr:8
    DM4D_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1 and 3
    DM4D_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> rows 2 and 3
#undef DM4D_DPP_ADD
    return v;
}
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace dm4d
