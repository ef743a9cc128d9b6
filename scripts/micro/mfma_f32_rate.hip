// What does the fp32 matrix pipe of gfx950 SUSTAIN?  A bare chain of v_mfma_f32_32x32x2_f32 (no memory, no LDS), NACC
// independent accumulators per wave, one or two waves per SIMD, on every CU, for ~5 ms: ns per MFMA and SIMD, the clock that
// implies if an instruction is 64 cycles, and the TFLOP/s of the whole chip.  The nominal peak (157.3 TF) is 64 cycles at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mfma_f32_rate.hip -o /tmp/mfma_f32_rate && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma_chain(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;                     // keeps the chain alive
}

template <int NACC>
void run(int wg_per_cu, int n_cu, float* out) {
    const int iters = 8000 / NACC * 4;                   // 8 * NACC * iters MFMAs per wave
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_chain<NACC>, dim3(n_cu * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 1e-6f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_simd = 8.0 * NACC * iters * wg_per_cu;               // MFMAs per SIMD (one wave of each workgroup per SIMD)
        const double ns = ms * 1e6 / per_simd;
        if (rep == 2)
            printf("%d accumulators, %d wave(s) per SIMD, %d CUs: %.2f ms, %.2f ns per MFMA and SIMD (= 64 cycles at %.2f GHz), "
                   "%.1f TFLOP/s = %.3f of 157.3\n", NACC, wg_per_cu, n_cu, ms, ns, 64.0 / ns,
                   4096.0 * 4 * n_cu / ns / 1e3, 4096.0 * 4 * n_cu / ns / 1e3 / 157.3);
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    float* out; CK(hipMalloc(&out, 64));
    run<4>(1, n_cu, out); run<4>(2, n_cu, out); run<8>(1, n_cu, out); run<8>(2, n_cu, out);
    run<4>(1, 225, out);                                 // the batch-1 WaveFlow launch: 225 workgroups, one wave per SIMD
    return 0;
}
