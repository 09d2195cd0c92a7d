// Micro-benchmark: issue cost of plain vs packed FP32 VALU instructions on gfx950 (cycles per wave64 instruction
// per SIMD at full occupancy, and for a lone wave with / without dependent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2v __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float c = 1.0001f;
    const f2v c2 = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent scalar fma chains
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 1) {   // 8 independent packed fma chains
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                         "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (MODE == 2) {   // ONE dependent scalar chain
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                         "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
                         : "+v"(a0) : "v"(c));)
        } else if (MODE == 3) {   // ONE dependent packed chain
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n"
                         "v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n"
                         : "+v"(p0) : "v"(c2));)
        } else if (MODE == 4) {   // transcendental-ish: ldexp + rndne + cvt
            REP16(asm volatile("v_rndne_f32 %0, %0\n v_ldexp_f32 %1, %1, %8\n v_rndne_f32 %2, %2\n v_ldexp_f32 %3, %3, %8\n"
                         "v_rndne_f32 %4, %4\n v_ldexp_f32 %5, %5, %8\n v_rndne_f32 %6, %6\n v_ldexp_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1));)
        } else if (MODE == 5) {   // v_cndmask + v_cmp mix
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_gt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
        } else if (MODE == 7) {   // v_fmac_f32_dpp row_newbcast (VOP2 DPP, broadcast source), 8 independent accumulators
            REP16(asm volatile("v_fmac_f32_dpp %0, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %8, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %4, %8, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %6, %8, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 8) {   // v_mul_f32_dpp row_shr (scan step), 8 independent registers (each reads itself through DPP: needs the nops)
            REP16(asm volatile("v_mul_f32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_dpp %2, %2, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %8 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %5, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                         "v_mul_f32_dpp %6, %6, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %7, %8 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 9) {   // the scan idiom of the blend backward: two chains, s_nop between the steps
            REP16(asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 0\n"
                         "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 0\n"
                         "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 0\n"
                         "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1));)
        } else if (MODE == 10) {   // v_exp_f32
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 6) {   // v_exp_f32 / v_rcp_f32 (quarter rate?)
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE> double run(int blocks, int threads, int iters, float *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *out; hipMalloc(&out, 1 << 26);
    const int iters = 2000; const double instr = 128.0 * iters;   // per wave
    const double ghz = 2.4;
    const char *names[] = {"v_fma_f32 x8 independent", "v_pk_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_pk_fma_f32 dependent chain",
                           "v_rndne/v_ldexp", "v_cmp+v_cndmask", "v_rcp_f32", "v_fmac_f32_dpp row_newbcast x8", "v_mul_f32_dpp row_shr x8", "scan idiom (8 dpp + 4 nop per 8)", "v_exp_f32"};
    for (int mode = 0; mode < 11; ++mode) {
        for (int cfg = 0; cfg < 3; ++cfg) {
            // cfg 0: 1 wave on the whole GPU; cfg 1: 1 wave per SIMD (1024 waves, 256 blocks x 256 thr); cfg 2: 8 waves per SIMD
            const int blocks = cfg == 0 ? 1 : cfg == 1 ? 256 : 2048, threads = cfg == 0 ? 64 : 256;
            double ms = 0;
            switch (mode) {
                case 0: ms = run<0>(blocks, threads, iters, out); break; case 1: ms = run<1>(blocks, threads, iters, out); break;
                case 2: ms = run<2>(blocks, threads, iters, out); break; case 3: ms = run<3>(blocks, threads, iters, out); break;
                case 4: ms = run<4>(blocks, threads, iters, out); break; case 5: ms = run<5>(blocks, threads, iters, out); break;
                case 6: ms = run<6>(blocks, threads, iters, out); break; case 7: ms = run<7>(blocks, threads, iters, out); break;
                case 8: ms = run<8>(blocks, threads, iters, out); break; case 9: ms = run<9>(blocks, threads, iters, out); break; case 10: ms = run<10>(blocks, threads, iters, out); break;
            }
            const double waves_per_simd = cfg == 0 ? 1 : cfg == 1 ? 1 : 8;
            const double cyc = ms * 1e-3 * ghz * 1e9 / (instr * waves_per_simd);
            printf("%-32s cfg %d (%s): %.3f ms -> %.2f cycles per wave-instruction per SIMD (at %.1f GHz)\n", names[mode], cfg,
                   cfg == 0 ? "lone wave" : cfg == 1 ? "1 wave/SIMD" : "8 waves/SIMD", ms, cyc, ghz);
        }
    }
    return 0;
}
