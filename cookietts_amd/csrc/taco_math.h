// Device math shared by the Tacotron decoder kernels (tacotron_persistent.hip, tacotron_batched.h): ~2-ulp exp / sigmoid / tanh
// on the hardware exp2 / rcp, and 64-lane reductions on the DPP network.
#pragma once

#include <hip/hip_runtime.h>

namespace ctts {
namespace tmath {

// ~2-ulp forms for the cell updates, a third of libm's instructions: e^y on the hardware exp2 with the product y log2(e)
// carried in two parts (the plain product's rounding error grows with |y|), tanh as an odd polynomial below 0.625 (own
// least-squares fit of (tanh x - x) / x^3 in x^2, 1.3 ulp) and 1 - 2 / (1 + e^(2|x|)) above.  Measured on the
// rounding-amplifying trajectory (tests/test_tacotron_long.py): distance to the exact run in band 1 1.6e-4 with the
// fast forms, 1.1e-4 with libm or these; step time 32.2 / 34.1 (libm) us.
__device__ __forceinline__ float acc_exp(float y) {
    y = y > 88.0f ? 88.0f : y;                            // e^88 is finite: inf * (correction) would be NaN, and 1 / (1 + e^88) is 0 anyway
                                                          // (a select, not fminf: v_min_f32 returns the OTHER operand for a NaN - a NaN
                                                          // pre-activation would come out as a finite gate where the reference's is NaN)
    const float p = y * 1.4426950408889634f;
    const float r = fmaf(y, 1.4426950408889634f, -p) + y * 1.925963033500011e-08f;     // log2(e) = hi + lo
    const float e = __builtin_amdgcn_exp2f(p);
    return fmaf(e, r * 0.6931471805599453f, e);
}
__device__ __forceinline__ float acc_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + acc_exp(-x)); }
__device__ __forceinline__ float acc_tanh(float x) {
    const float ax = fabsf(x), u = x * x;
    float pl = fmaf(u, -0.005704042501747608f, 0.020637862384319305f);
    pl = fmaf(u, pl, -0.05373915657401085f); pl = fmaf(u, pl, 0.133314311504364f); pl = fmaf(u, pl, -0.3333328068256378f);
    const float small = fmaf(ax * u, pl, ax);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + acc_exp(2.0f * ax));
    return copysignf(ax < 0.625f ? small : big, x);
}

// 64-lane reductions on the DPP network (v_add_f32_dpp: quad_perm x2, row_half_mirror, row_mirror, row_bcast15,
// row_bcast31; the total lands in lanes 48..63 and is read back with v_readlane, so the result is wave-uniform): ~12
// issue slots per value instead of six ds_bpermute round trips through the LDS crossbar, which every phase of the step
// used to pay in sequence (360 ds_bpermute in the round-2 ISA).  N independent values interleave level by level.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v) {     // disabled rows / lanes see their own value
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false)));
}
template <int N>
__device__ __forceinline__ void wave_totals(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0xB1, 0xf>(v[i]);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x4E, 0xf>(v[i]);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x141, 0xf>(v[i]);     // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x140, 0xf>(v[i]);     // row_mirror: every lane holds its row's sum
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x142, 0xa>(v[i]);     // row_bcast15 into rows 1, 3
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x143, 0xc>(v[i]);     // row_bcast31 into rows 2, 3: row 3 = total
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i]), 63));
}
__device__ __forceinline__ float wave_total(float v) {
    float a[1] = {v};
    wave_totals<1>(a);
    return a[0];
}
__device__ __forceinline__ float wave_max(float v) {
    v = dpp_max<0xB1, 0xf>(v); v = dpp_max<0x4E, 0xf>(v); v = dpp_max<0x141, 0xf>(v); v = dpp_max<0x140, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v); v = dpp_max<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

}  // namespace tmath
}  // namespace ctts
