// Micro-benchmark: what does it cost to get the blend backward's per-entry records out of the CUs?
// The blend backward of the bench step produces R = 12.75 M (Gaussian, cell) records of 9 floats per 8-view step, 64 per
// wave iteration (one per lane).  Compared here, for the same number of records and the same (seeded, window-local)
// targets:
//   store48      today's path: staged in LDS, lanes 4i .. 4i+2 write the three 16-byte parts of record i (unique slots)
//   atom_lane    every lane adds its 9 values to the per-GAUSSIAN accumulator (N x 12 floats per view) with 9
//                global_atomic_add_f32 (no return): 64 different lines per instruction
//   atom_staged  records staged in LDS; 12 consecutive lanes add 9 (of 12) consecutive dwords of one record: 5.3 records per
//                instruction, one 48-byte piece per record
//   atom_tile    as atom_staged, target = per-(Gaussian, tile) record (D records instead of N accumulators)
// The kernel does nothing else, so the time is the memory system's: if it is well below the blend backward's VALU time
// (~265 us) the atomics hide behind the arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// target of record r of wave w: a Gaussian within `window` of the wave's centre (cells hold neighbouring faces)
__device__ __forceinline__ uint32_t target(uint32_t wave, uint32_t iter, uint32_t lane, uint32_t n_targets, uint32_t window)
{
    const uint32_t centre = hash32(wave * 2654435761u + (iter >> 2)) % n_targets;
    const uint32_t off = hash32(wave * 64u * 1024u + iter * 64u + lane) % window;
    return (centre + off) % n_targets;
}

__device__ __forceinline__ void fadd(float *p, float v) { __builtin_amdgcn_global_atomic_fadd_f32(p, v); }

// mode 0: store48 (unique slot per record)  1: atom_lane  2: atom_staged
template <int MODE>
__global__ __launch_bounds__(64) void k_out(float *__restrict__ dst, uint32_t n_targets, uint32_t window, uint32_t iters, uint32_t total_records)
{
    __shared__ __attribute__((aligned(16))) float s_rec[64][12];
    __shared__ uint32_t s_slot[64];
    const uint32_t wave = blockIdx.x, lane = threadIdx.x;
    for (uint32_t it = 0; it < iters; ++it) {
        float acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = (float)(lane + i + it) * 1e-3f;
        uint32_t slot;
        if (MODE == 0) {
            const uint64_t r = ((uint64_t)wave * iters + it) * 64u + lane;
            slot = (uint32_t)((r * 2654435761ull) % total_records);     // scattered unique-ish slots
        } else slot = target(wave, it, lane, n_targets, window);
        if (MODE == 1) {
            float *p = dst + (size_t)slot * 12;
#pragma unroll
            for (int i = 0; i < 9; ++i) fadd(p + i, acc[i]);
        } else {
            __builtin_amdgcn_wave_barrier();
            float4 *mine = reinterpret_cast<float4 *>(s_rec[lane]);
            mine[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            mine[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            mine[2] = make_float4(acc[8], 0.f, 0.f, 0.f);
            s_slot[lane] = slot;
            __builtin_amdgcn_wave_barrier();
            if (MODE == 0) {
                const int part = lane & 3;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int entry = 16 * p + (lane >> 2);
                    const uint32_t sl = s_slot[entry];
                    if (part < 3) reinterpret_cast<float4 *>(dst + (size_t)sl * 12)[part] = reinterpret_cast<const float4 *>(s_rec[entry])[part];
                }
            } else {
                // 16 lanes per record (9 active): 4 records per instruction, 16 instructions
                const int v = lane & 15;
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int entry = 4 * p + (lane >> 4);
                    const uint32_t sl = s_slot[entry];
                    if (v < 9) fadd(dst + (size_t)sl * 12 + v, s_rec[entry][v]);
                }
            }
        }
    }
}

template <typename F>
static float best_ms(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const uint32_t R = 12750000u;              // records per step (8 views)
    const uint32_t waves = 32768u * 2u;        // blocks (the blend backward launches 32768 quadrant waves + wide blocks)
    const uint32_t iters = R / 64u / waves + 1u;
    const uint32_t total = waves * iters * 64u;
    const size_t rec_bytes = (size_t)total * 48;
    float *dst;
    CHECK(hipMalloc(&dst, rec_bytes));
    CHECK(hipMemset(dst, 0, rec_bytes));
    printf("records %u (%u waves x %u iterations x 64), record array %.0f MB\n", total, waves, iters, rec_bytes / 1e6);
    const float t0 = best_ms([&] { hipLaunchKernelGGL(k_out<0>, dim3(waves), dim3(64), 0, 0, dst, total, 1u, iters, total); });
    printf("store48      unique slots                        %8.1f us  (%.2f TB/s written)\n", t0 * 1e3, rec_bytes / (t0 * 1e-3) / 1e12);
    const uint32_t nG = 8u * 200000u, nD = 3760000u;
    for (uint32_t window : {64u, 512u, 4096u, 1600000u}) {
        const float t1 = best_ms([&] { hipLaunchKernelGGL(k_out<1>, dim3(waves), dim3(64), 0, 0, dst, nG, window, iters, total); });
        const float t2 = best_ms([&] { hipLaunchKernelGGL(k_out<2>, dim3(waves), dim3(64), 0, 0, dst, nG, window, iters, total); });
        const float t3 = best_ms([&] { hipLaunchKernelGGL(k_out<2>, dim3(waves), dim3(64), 0, 0, dst, nD, window, iters, total); });
        printf("window %7u: atom_lane (1.6 M accumulators) %8.1f us | atom_staged (1.6 M) %8.1f us | atom_staged (3.76 M tile records) %8.1f us\n",
               window, t1 * 1e3, t2 * 1e3, t3 * 1e3);
    }
    CHECK(hipFree(dst));
    return 0;
}
