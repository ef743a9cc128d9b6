// Fused 1x1 stages of a separable WaveFlow layer (waveflow_sep.hip) - interface towards waveflow_api.hip.
#pragma once

#include "common.h"

namespace ctts {

constexpr int WF_MAX_KH = 8;

struct WfSlots { const float* p[WF_MAX_KH]; };   // ring slots of the height taps (padded [B][C][ld] tensors)

struct WfSepArgs {
    const float* dwout;                 // depthwise stage output [B][C][ld]
    const float* A1; const float* b1;                // packed pointwise image + bias (gate order)
    const float* A2; const float* b2;                // packed res/skip image + bias
    const float* cond;                  // upsampled conditioning of this layer [B][2C][ld]
    const float* xin;                   // x_i of the current row (= x.p[kh-1])
    float* xout;                        // x_{i+1} ring slot (NULL on the last layer)
    float* out;                         // skip accumulator [B][C][ld]
    int acc_out;                        // out += (layers > 0) or out = (layer 0)
    int rs_rows;                        // 256, or 128 on the last layer / with merge_res_skip (skip only)
    int gate;                           // GateKind (gemm_f32.h), 0 = GTU
    int split_bf16;                     // 1: split-bf16 main loop (the model's f32_gemm_mode resolved by the caller)
    int L, ld, pad, ntiles;             // ntiles = ceil(L / 64)
};

bool wf_sep_supported(int C);
// pw_w [2C][C], pw_b [2C], rs_w [rs_rows][C], rs_b [rs_rows] -> A1/A2 [128*256] floats each, b1/b2 [256]
int launch_wf_sep_pack(const float* pw_w, const float* pw_b, const float* rs_w, const float* rs_b, float* A1, float* b1,
                       float* A2, float* b2, int rs_rows, hipStream_t s);
int launch_wf_sep_layer(const WfSepArgs& a, int batch, hipStream_t s);

}  // namespace ctts
