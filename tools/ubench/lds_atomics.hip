// Micro-benchmark: throughput of LDS read-modify-write on gfx950 -- ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_add_f64 (no return)
// against plain ds_write_b32 and a non-atomic ds_read + add + ds_write, conflict-free addresses (lane -> its own bank) and a
// random scatter, 8 waves per CU on every CU.  Output: LDS cycles per wave64 instruction per CU (the LDS is shared by the CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) double lds_d;
template <int MODE, bool SCATTER>
__global__ __launch_bounds__(512) void k(int iters, float *out)
{
    __shared__ double s[4096];
    float *sf = reinterpret_cast<float *>(s);
    uint32_t *su = reinterpret_cast<uint32_t *>(s);
    unsigned long long *sl = reinterpret_cast<unsigned long long *>(s);
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) s[i] = 0.0;
    __syncthreads();
    uint32_t h = tid * 2654435761u;
    int idx = SCATTER ? 0 : tid; float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            if (SCATTER) { h = h * 1664525u + 1013904223u; idx = (h >> 12) & 4095; }
            else idx = (tid + 512 * (u % 8)) & 4095;
            if (MODE == 0) __builtin_amdgcn_ds_faddf((lds_f *)(sf + idx), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            if (MODE == 1) __hip_atomic_fetch_add(su + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_add(sl + idx, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) __hip_atomic_fetch_add(s + idx, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 4) sf[idx] = (float)it;
            if (MODE == 5) { const float v = sf[idx]; sf[idx] = v + 1.0f; }
            if (MODE == 6) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((uint32_t)(idx * 4)), "v"(1.0f) : "memory"); acc += r; }
            if (MODE == 7) { asm volatile("ds_add_f32 %0, %1" : : "v"((uint32_t)(idx * 4)), "v"(1.0f) : "memory"); }
            if (MODE == 8) { asm volatile("ds_pk_add_f16 %0, %1" : : "v"((uint32_t)(idx * 4)), "v"(0x3c003c00u) : "memory"); }
            if (MODE == 9) { asm volatile("ds_max_f32 %0, %1" : : "v"((uint32_t)(idx * 4)), "v"((float)it) : "memory"); }
        }
    }
    __syncthreads();
    if (sf[tid] + acc == 123.f) out[0] = 1.f;
}
template <int MODE, bool SCATTER>
static void run(const char *name, int cus, double mhz)
{
    float *out; CHECK(hipMalloc(&out, 4));
    const int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<MODE, SCATTER>), dim3(cus), dim3(512), 0, 0, iters, out);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    const double instr = 8.0 * iters * 9;      // wave instructions per CU
    printf("%-28s %-12s %8.1f us   %6.1f cycles per wave64 instruction per CU\n", name, SCATTER ? "scatter" : "conflict-free", best * 1e3, best * 1e-3 * mhz * 1e6 / instr);
    CHECK(hipFree(out));
}
int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1e3;
    printf("%d CUs, %.0f MHz\n", cus, mhz);
    run<0, false>("ds_add_f32", cus, mhz); run<0, true>("ds_add_f32", cus, mhz);
    run<1, false>("ds_add_u32", cus, mhz); run<1, true>("ds_add_u32", cus, mhz);
    run<2, false>("ds_add_u64", cus, mhz); run<2, true>("ds_add_u64", cus, mhz);
    run<3, false>("ds_add_f64", cus, mhz); run<3, true>("ds_add_f64", cus, mhz);
    run<4, false>("ds_write_b32", cus, mhz); run<4, true>("ds_write_b32", cus, mhz);
    run<5, false>("ds_read + add + ds_write", cus, mhz); run<5, true>("ds_read + add + ds_write", cus, mhz);
    run<6, false>("ds_add_rtn_f32 (asm)", cus, mhz); run<6, true>("ds_add_rtn_f32 (asm)", cus, mhz);
    run<7, false>("ds_add_f32 (asm)", cus, mhz); run<7, true>("ds_add_f32 (asm)", cus, mhz);
    run<8, false>("ds_pk_add_f16 (asm)", cus, mhz); run<8, true>("ds_pk_add_f16 (asm)", cus, mhz);
    run<9, false>("ds_max_f32 (asm)", cus, mhz); run<9, true>("ds_max_f32 (asm)", cus, mhz);
    return 0;
}
