// Micro-benchmark / layout probe of the multi-block f32 MFMA forms the blend kernels use (gfx950):
//   v_mfma_f32_4x4x1_16b_f32  (16 blocks of a 4x4 outer product, 4 result VGPRs)
//   v_mfma_f32_16x16x1_4b_f32 (4 blocks of a 16x16 outer product, 16 result VGPRs)
// 1. LAYOUT: A = lane id, B = 100 + lane id, C = 0  ->  every result value a*b identifies its (A lane, B lane): printed per (lane, register).
// 2. NUMERICS: accumulation chain against fmaf, bit for bit.
// 3. THROUGHPUT: wave-instruction cycles per SIMD with 1..8 waves per SIMD, alone and beside a VALU-only FMA stream in the same wave
//    (does the matrix pipe run beside the vector ALU?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_layout4(float *out)
{
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 100.f + (float)l, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void k_layout16(float *out)
{
    const int l = threadIdx.x;
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_16x16x1f32((float)l, 100.f + (float)l, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
// numerics: D = fma(a_k, b_k, D) chain, every lane the diagonal element of its block (i == j): lane l -> A lane l, B lane l, register l % 4
__global__ void k_chain4(const float *a, const float *b, float *out, int K)
{
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    float ref = 0.f;
    for (int k = 0; k < K; ++k) {
        const float av = a[k * 64 + l], bv = b[k * 64 + l];
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c, 0, 0, 0);
        ref = __builtin_fmaf(av, bv, ref);
    }
    out[l] = c[l & 3];
    out[64 + l] = ref;
}
template <int MODE>      // 0: 4x4x1 only, 1: 16x16x1 only, 2: VALU fma only, 3: 4x4x1 + VALU, 4: 16x16x1 + VALU
__global__ __launch_bounds__(512) void k_rate(int iters, float *out)
{
    const int l = threadIdx.x & 63;
    f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    f16v d0, d1;
    for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
    float a = 1.0f + l * 1e-3f, b = 0.5f + l * 1e-4f;
    float v0 = a, v1 = b, v2 = a + 1.f, v3 = b + 1.f, v4 = a + 2.f, v5 = b + 2.f, v6 = a + 3.f, v7 = b + 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0 || MODE == 3) {
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, a, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, b, c3, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 4) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x1f32(b, a, d1, 0, 0, 0);
            }
            if (MODE >= 2) {
                // 16 plain v_fma_f32 on 8 chains (inline asm: the compiler would SLP-pack them into v_pk_fma_f32, which the guide
                // lists as an anti-lever beside MFMAs)
#define F(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b))
                F(v0); F(v1); F(v2); F(v3); F(v4); F(v5); F(v6); F(v7);
                F(v0); F(v1); F(v2); F(v3); F(v4); F(v5); F(v6); F(v7);
#undef F
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int r = 0; r < 16; ++r) s += d0[r] + d1[r];
    if (s == 123.456f) out[0] = s;
}
template <int MODE>
static void rate(const char *name, int cus, double mhz, int waves_per_simd)
{
    float *out; CHECK(hipMalloc(&out, 4));
    const int iters = 4000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    const int threads = 64 * 4 * (waves_per_simd > 2 ? 2 : waves_per_simd), blocks_per_cu = waves_per_simd > 2 ? waves_per_simd / 2 : 1;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_rate<MODE>), dim3(cus * blocks_per_cu), dim3(threads), 0, 0, iters, out);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * mhz * 1e6 / ((double)iters * 4 * waves_per_simd);     // cycles per (wave, unrolled group) on one SIMD
    const int n_mfma = (MODE == 0 || MODE == 3) ? 4 : (MODE == 1 || MODE == 4) ? 2 : 0, n_valu = MODE >= 2 ? 16 : 0;
    printf("%-34s %d waves/SIMD  %8.1f us  %6.1f cycles per group of (%d MFMA + %d v_fma) per wave\n", name, waves_per_simd, best * 1e3, cyc, n_mfma, n_valu);
    CHECK(hipFree(out));
}
int main(int argc, char **argv)
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double mhz = p.clockRate * 1e-3;
    printf("%s: %d CUs, %.0f MHz\n", p.gcnArchName, cus, mhz);
    float *d; CHECK(hipMalloc(&d, 64 * 16 * 4));
    float h[64 * 16];
    hipLaunchKernelGGL(k_layout4, dim3(1), dim3(64), 0, 0, d); CHECK(hipMemcpy(h, d, 64 * 4 * 4, hipMemcpyDeviceToHost));
    printf("4x4x1_16b: result (lane, reg) = A lane x B lane\n");
    for (int l = 0; l < 64; ++l) {
        printf("  lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            // find (la, lb) with la * (100 + lb) == value
            int fa = -1, fb = -1;
            for (int la = 0; la < 64 && fa < 0; ++la) for (int lb = 0; lb < 64; ++lb) if ((float)la * (100.f + lb) == h[l * 4 + r] && (la || h[l * 4 + r] == 0.f)) { fa = la; fb = lb; break; }
            printf("  r%d=A%02d*B%02d", r, fa, fb);
        }
        printf("\n");
        if (l == 7) { printf("  ...\n"); l = 55; }
    }
    hipLaunchKernelGGL(k_layout16, dim3(1), dim3(64), 0, 0, d); CHECK(hipMemcpy(h, d, 64 * 16 * 4, hipMemcpyDeviceToHost));
    printf("16x16x1_4b: result (lane, reg) = A lane x B lane\n");
    const int show[] = {0, 1, 15, 16, 17, 33, 63};
    for (int l : show) {
        printf("  lane %2d:", l);
        for (int r = 0; r < 16; ++r) {
            int fa = -1, fb = -1;
            for (int la = 1; la < 64 && fa < 0; ++la) for (int lb = 0; lb < 64; ++lb) if ((float)la * (100.f + lb) == h[l * 16 + r]) { fa = la; fb = lb; break; }
            if (h[l * 16 + r] == 0.f) printf(" r%d=A00*B??", r); else printf(" r%d=A%02d*B%02d", r, fa, fb);
        }
        printf("\n");
    }
    // numerics
    const int K = 64;
    float ha[K * 64], hb[K * 64], ho[128];
    srand(1);
    for (int i = 0; i < K * 64; ++i) { ha[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.f; hb[i] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f * (1 + i % 7); }
    float *da, *db, *dout; CHECK(hipMalloc(&da, sizeof(ha))); CHECK(hipMalloc(&db, sizeof(hb))); CHECK(hipMalloc(&dout, sizeof(ho)));
    CHECK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_chain4, dim3(1), dim3(64), 0, 0, da, db, dout, K); CHECK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
    int same = 0; for (int l = 0; l < 64; ++l) same += memcmp(&ho[l], &ho[64 + l], 4) == 0;
    printf("numerics: 64-term chains, MFMA 4x4x1 vs fmaf: %d / 64 lanes bit-identical\n", same);
    for (int w : {1, 2, 4, 8}) {
        rate<0>("4x4x1_16b x4", cus, mhz, w);
        rate<1>("16x16x1_4b x2", cus, mhz, w);
        rate<2>("v_fma_f32 x16", cus, mhz, w);
        rate<3>("4x4x1_16b x4 + v_fma_f32 x16", cus, mhz, w);
        rate<4>("16x16x1_4b x2 + v_fma_f32 x16", cus, mhz, w);
    }
    return 0;
}
