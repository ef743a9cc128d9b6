// Launchers of the non-GEMM WaveGlow kernels (definitions in waveglow_kernels.hip).
#pragma once

#include "common.h"
#include "gemm_f32.h"

namespace ctts {

int launch_fold_weightnorm(const float* v, const float* g, float* w, int out_ch, int fan, hipStream_t s);
int launch_pack_a(float* dst, const float* src, int bm, int MB, int nch_total, int k_off, int ksrc, int epi, int C, int M,
                  long long src_row_off, long long src_row_stride, int src_k_stride, hipStream_t s, int k_group = 1,
                  int k_member = 0);
int launch_pack_bias(float* dst, int bm, int MB, const float* src0, long long off0, const float* src1, long long off1,
                     int epi, int C, int M, hipStream_t s);
// Wp: W in MFMA fragment order (launch_upsample_pack_mfma; only for shapes upsample_mfma_shape accepts) or nullptr = VALU kernels
bool upsample_mfma_shape(int n_mel, int win, int hop, int G);
int launch_upsample_pack_mfma(const float* W, float* Wp, hipStream_t s);
int launch_upsample_squeeze(const float* mel, const float* W, const float* Wp, const float* bias, float* spect, int batch,
                            int n_mel, int F, int win, int hop, int G, int ld, int pad, hipStream_t s);
int launch_wn_start(const float* audio, const float* Ws, const float* bs, float* x, int batch, int C, int G,
                    int ch_off, int n_half, int L, int ld, int pad, hipStream_t s);
int launch_flow_tail(const float* out, float* audio, float* wave, const float* Wend, const float* bend,
                     const float* Winv, int batch, int C, int G, int ch_off, int n_half, int L, int ld, int pad,
                     hipStream_t s);

}  // namespace ctts
