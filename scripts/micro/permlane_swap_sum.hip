// gfx950's v_permlane16_swap_b32 / v_permlane32_swap_b32 as the xor-16 / xor-32 steps of a cross-row sum: does
//   a = x, b = x; swap(a, b); a + b   equal   x + __shfl_xor(x, 16)   (resp. 32) bit for bit in every lane?
// (the persistent Tacotron decoder's pd_block_sum pays two ds_bpermute round trips through the LDS crossbar per sum; these are VALU)
//   hipcc --offload-arch=gfx950 -O3 permlane_swap_sum.hip -o permlane_swap_sum && ./permlane_swap_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k(const float* in, float* ref, float* out) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    float v = in[t], w = v;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    ref[t] = v;
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    w = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    out[t] = __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

int main() {
    const int n = 64 * 64;
    std::vector<float> h(n);
    unsigned s = 7u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1.37e-4f; }
    float *in, *ref, *out;
    hipMalloc(&in, n * 4); hipMalloc(&ref, n * 4); hipMalloc(&out, n * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, in, ref, out);
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), ref, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), out, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += std::memcmp(&a[i], &b[i], 4) != 0;
    printf("%d of %d lanes differ between the shfl_xor form and the permlane-swap form (first values %.6f / %.6f)\n", bad, n, a[0], b[0]);
    return bad != 0;
}
