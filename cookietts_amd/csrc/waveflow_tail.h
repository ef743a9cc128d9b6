// End of a WaveFlow row (efficient_modules.py:61-62, glow_ax.py:558, 628) per column:
//   e = Wend . out + bend;  rows[row] = (rows[row] - e[1]) / exp(e[0]);  next row's X0[c] = ws[c] * rows[row] + bs[c]
// in ONE arithmetic shared by the two stand-alone kernels (wf_tail_kernel, wf_start_kernel: one launch each per row) and by the
// tail item of the row queue (gemm_f32_small.hip), so that the forms agree bit for bit: the end conv of a column is four fma
// chains over C / 4 ascending channels each, added as ((p0 + p1) + p2) + p3.
#pragma once

#include "common.h"

namespace ctts {

struct WfTailDesc {
    const float* out;            // skip sum [B][C][ld], valid columns [pad, pad + L)
    long long out_bstride;
    float* rows;                 // [B][G][Lr]; row `row` (the physical index of the NEXT logical row) is updated in place
    const float* Wend;           // [2][C]
    const float* bend;           // [2]
    const float* ws;             // start conv [C] / [C]
    const float* bs;
    float* x0;                   // X(0, slot of the next row) [B][C][ld], or NULL behind the flow's last row
    long long x0_bstride;
    int C, G, row, L, Lr, ld, pad;
};

#if defined(__HIPCC__)
__device__ __forceinline__ float wf_end_fma(float w, float v, float e) { return __builtin_fmaf(w, v, e); }
__device__ __forceinline__ float wf_row_update(float a, float e0, float e1, float b0, float b1) { return (a - (e1 + b1)) / expf(e0 + b0); }
__device__ __forceinline__ float wf_start_value(float w, float a, float bias) { return __builtin_fmaf(w, a, bias); }
#endif

}  // namespace ctts
