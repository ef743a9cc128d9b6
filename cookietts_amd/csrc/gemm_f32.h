// fp32 MFMA "conv-GEMM" used by every dense contraction of the vocoder / frontend paths.
//
//   D[b][m][n] = bias[m] + sum_seg sum_k A[m][koff_seg + k] * Bseg[b][k][n + shift_seg]
//
// A is the pre-packed weight matrix, the B operand is gathered from up to GEMM_MAX_SEG
// "segments" (dilated-conv taps of x at column shifts -d/0/+d, height taps of the WaveFlow
// row queue, the conditioning input) that all live in the padded activation layout, so no
// bounds checks are needed on loads.  Exact fp32: v_mfma_f32_32x32x2_f32 (one rounding per
// product).
#pragma once

#include "common.h"

namespace ctts {

constexpr int GEMM_KC = 16;    // K per LDS stage
constexpr int GEMM_MAX_SEG = 12;

// Two workgroup shapes (256 threads = 4 waves, wave tile always 128 x 64):
//   bm = 256: waves 2(M) x 2(N), block tile 256 x 128   (C >= 128 channel GEMMs)
//   bm = 128: waves 1(M) x 4(N), block tile 128 x 256   (WaveFlow's 64-channel GEMMs)
constexpr int GEMM_BM = 256;   // the bm = 256 shape
constexpr int GEMM_BN = 128;
inline int gemm_bn(int bm) { return bm == 256 ? 128 : 256; }

enum GemmEpilogue : int {
    // rows < split -> dst0[row] (= src0[row] + v when acc0), rows >= split -> dst1[row - split]
    GEMM_EPI_SPLIT = 0,
    // each wave-row holds (64 tanh, 64 sigmoid) rows: dst0[c] = tanh(u_t) * sigmoid(u_s)
    GEMM_EPI_GATE = 1,
    // same row pairing, (re, im) rows: dst0[c] = sqrt(re^2 + im^2)          (STFT magnitude)
    GEMM_EPI_MAG = 2,
    // dst0[row] = log(max(acc + bias, clip))                                 (mel projection + log)
    GEMM_EPI_LOG = 3,
    // dst0[row] = leaky_relu(acc + bias, clip as slope) / tanh(acc + bias)    (Tacotron encoder / postnet convs)
    GEMM_EPI_LRELU = 4,
    GEMM_EPI_TANH = 5,
    // GATE followed, in the same workgroup, by the res/skip 1x1 GEMM on the gated tile held in registers
    // (bm = 128 shape with all pairC <= 64 channels in one wave-row): the activations never touch HBM.
    //   rows < split of  rs_w . act + rs_b  -> dst0 = src0 + .   (next layer's input)
    //   rows >= split                       -> dst1 (+)= .       (skip sum)
    GEMM_EPI_GATE_RS = 6,
};

__host__ __device__ inline bool gemm_epi_is_pair(int epi) {
    return epi == GEMM_EPI_GATE || epi == GEMM_EPI_MAG || epi == GEMM_EPI_GATE_RS;
}

struct GemmSeg {
    const float* base;    // [B][rows][ld] padded layout
    long long bstride;    // floats between batch items
    int nch;              // K chunks (of GEMM_KC rows) this segment contributes
    int shift;            // column shift (time steps)
    int mb_rows;          // extra row offset per M-block (block-diagonal batched GEMMs)
    int reserved;
};

struct GemmArgs {
    const float* A;       // packed [MB][a_nch_alloc][GEMM_KC][bm]
    const float* bias;    // [MB*bm] in block-local row order
    GemmSeg seg[GEMM_MAX_SEG];
    int nseg;
    int interleave;       // G > 1: segments 0..G-1 (equal nch) are consumed round-robin, one chunk each, so the
                          // taps of one 16-channel slab are read back to back (L2 reuse); the rest sequentially
    int nch_total;        // K chunks used by this launch (sum of seg[].nch)
    int a_nch_alloc;      // K chunks per M-block in the packed A (>= a_ch_off + nch_total); 0 = nch_total
    int a_ch_off;         // first chunk of A to use (skips leading taps, e.g. WaveFlow rows 0/1)
    int bm;               // 256 or 128 (0 = 256)
    int ld, pad, L;       // row stride, left halo, valid columns of the B-operand tensors
    int ntiles, MB, batch;
    int M;                // valid rows; rows >= M of the last M-block are padding
    float* dst0; long long dst0_bstride; int acc0;
    const float* src0; long long src0_bstride;   // acc0 source (NULL = dst0 itself)
    float* dst1; long long dst1_bstride; int acc1;
    int split;            // GEMM_EPI_SPLIT row split (multiple of 32); pair epilogues: unused
    int pairC;            // pair epilogues: number of valid channels (dense rows c and pairC + c)
    int dst_ld, dst_pad;  // row stride / left pad of the destination tensors (usually == ld, pad)
    float clip;           // GEMM_EPI_LOG clamp / GEMM_EPI_LRELU negative slope
    // GATE / GATE_RS: optional per-element addend before the gate (conditioning computed elsewhere), padded layout
    // [B][2*pairC][ld] in DENSE row order (row c -> tanh input, row pairC + c -> sigmoid input), same ld / pad as B
    const float* addend; long long addend_bstride;
    int addend_ld, addend_pad;   // row stride / left pad of the addend tensor (0, 0 = ld, pad of the B operand)
    // GATE only: addend_frames = F > 0: the addend holds F columns per row at a lower rate and is linearly
    // interpolated to the L columns of the launch (align_corners=True: position n reads n * (F-1)/(L-1)), the
    // arithmetic of ATen's upsample_linear1d (glow_ax.py:362-373) - the sample-rate tensor is never materialised
    int addend_frames;
    int map_mode;         // block id -> (m-block, tile, batch) mapping, chosen by the launcher (see gemm_f32.hip)
    int exact_f32;        // 1 = always the fp32 MFMA main loop, whatever set_gemm_f32_mode says (STFT: its sums cancel,
                          // so the split-bf16 loop's 2^-17 operand error exceeds the 1e-4 log-mel bound)
    const float* rs_wT;   // GEMM_EPI_GATE_RS: res/skip weight transposed and row-padded: [64][128]
    const float* rs_bias; // [128] (rows >= rs_rows zero)
    int rs_rows;          // 128 (res + skip) or 64 (last layer: skip only)
};

// Row of the dense weight matrix held by block-local row r of M-block mb, or -1 for padding
// rows (their packed weights and bias are zero).  Pair epilogues (GATE, MAG) interleave so that
// each wave-row (128 rows) owns 64 first-half channels and the 64 matching second-half channels
// (dense rows c and C + c), which lets the epilogue combine them in registers.
__host__ __device__ inline int gemm_dense_row(int epi, int bm, int mb, int r, int C, int M) {
    if (gemm_epi_is_pair(epi)) {
        const int wm = r >> 7, rr = r & 127;
        const int c = (mb * (bm >> 7) + wm) * 64 + (rr & 63);
        if (c >= C) return -1;
        return (rr < 64) ? c : C + c;
    }
    const int row = mb * bm + r;
    return row < M ? row : -1;
}

int launch_gemm_f32(int epi, const GemmArgs& a, hipStream_t stream);

// Process-wide main-loop selection of launch_gemm_f32: 0 = fp32 MFMA (default), 1 = split bf16 (three bf16 MFMA products
// per fp32 operand pair, see conv_gemm_f32_kernel<..., X3>).  Set through ctts_set_f32_gemm_mode.
int set_gemm_f32_mode(int mode);
int get_gemm_f32_mode();

}  // namespace ctts
