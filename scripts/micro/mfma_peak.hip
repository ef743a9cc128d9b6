// Sustained MFMA issue rate on gfx950 with no memory traffic: what "peak" means on this box for the
// 32x32x16 bf16 and 32x32x2 f32 instructions the conv-GEMMs use.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BF>
__global__ __launch_bounds__(512) void spin(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * j); }
    const float fa = 0.001f * threadIdx.x, fb = 0.5f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
            }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bf = 1; bf >= 0; --bf)
        for (int threads : {256, 512})
            for (int rep = 0; rep < 3; ++rep) {
                const int wgs = 256, iters = bf ? 400000 : 400000;
                hipEventRecord(e0);
                if (bf) hipLaunchKernelGGL(spin<1>, dim3(wgs), dim3(threads), 0, 0, out, iters, clk);
                else hipLaunchKernelGGL(spin<0>, dim3(wgs), dim3(threads), 0, 0, out, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
                const double flop_per = bf ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
                const double flops = flop_per * 16.0 * iters * (threads / 64) * wgs;
                printf("%s waves/SIMD=%d: %.1f ms, %.1f TFLOP/s, shader clock %.0f MHz (cycles %llu / %.1f us), cycles per MFMA per SIMD %.2f\n",
                       bf ? "bf16 32x32x16" : "f32 32x32x2", threads / 256, ms, flops / ms / 1e9,
                       (double)h[0] / ((double)h[1] / 100.0), h[0], (double)h[1] / 100.0,
                       (double)h[0] / (16.0 * iters * (threads / 256)));
            }
    return 0;
}
