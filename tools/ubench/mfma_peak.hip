// What the chip SUSTAINS on v_mfma_f32_32x32x16_f16 with nothing else in the way: every CU, W waves per SIMD, each wave a chain of
// 4 independent accumulators (the 64 x 64 wave tile of csrc/conv_mfma.hip), no memory traffic at all; random (non-zero) operands, because the
// power the matrix pipes draw -- and with it the clock the chip holds -- depends on the data.  Prints TFLOP/s against the nominal 2.5 PFLOP/s
// (256 CUs x 4 SIMDs x 1024 flop per cycle x 2.4 GHz) and the clock the rate implies.    hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k_mfma(const _Float16 *src, float *out, int iters, float scale)
{
    f16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) {
        a[i][k] = (_Float16)((float)src[(threadIdx.x * 16 + i * 8 + k) & 4095] * scale);
        b[i][k] = (_Float16)((float)src[(threadIdx.x * 16 + 2048 + i * 8 + k) & 4095] * scale);
    }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
}
int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    _Float16 *src; float *out;
    CHECK(hipMalloc(&src, 4096 * 2)); CHECK(hipMalloc(&out, 4));
    _Float16 h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    CHECK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (float scale : {1.0f, 0.0f})
        for (int wps : {1, 2})
            for (int iters : {2000, 20000, 200000}) {
                const int threads = 256 * wps;
                hipLaunchKernelGGL(k_mfma, dim3(cus), dim3(threads), 0, 0, src, out, 100, scale);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_mfma, dim3(cus), dim3(threads), 0, 0, src, out, iters, scale);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double flop = (double)cus * (threads / 64) * iters * 16.0 * 32 * 32 * 16 * 2;
                const double tf = flop / (ms * 1e-3) / 1e12;
                printf("operands %s, %d wave(s) per SIMD, %7.2f ms: %7.1f TFLOP/s = %.3f of 2500; implied clock %.2f GHz (%d CUs x 4 SIMDs x 1024 flop / cycle)\n",
                       scale != 0.f ? "random" : "zeros ", wps, ms, tf, tf / 2500.0, tf * 1e12 / (cus * 4.0 * 1024.0) / 1e9, cus);
            }
    return 0;
}
