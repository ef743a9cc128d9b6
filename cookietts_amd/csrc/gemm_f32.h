// fp32 MFMA "conv-GEMM" used by every dense contraction of the WN stack.
//
//   D[b][m][n] = bias[m] + sum_seg sum_k A[m][koff_seg + k] * Bseg[b][k][n + shift_seg]
//
// A is the pre-packed weight matrix, the B operand is gathered from up to four
// "segments" (dilated-conv taps of x at column shifts -d/0/+d, plus the conditioning
// hidden h) that all live in the padded activation layout, so no bounds checks are
// needed on loads.  Exact fp32: v_mfma_f32_32x32x2_f32 (one rounding per product).
#pragma once

#include "common.h"

namespace ctts {

constexpr int GEMM_BM = 256;   // rows per workgroup
constexpr int GEMM_BN = 128;   // time steps per workgroup
constexpr int GEMM_KC = 16;    // K per LDS stage
constexpr int GEMM_MAX_SEG = 4;

enum GemmEpilogue : int {
    // rows < split -> dst0[row], rows >= split -> dst1[row - split]; each optionally accumulates
    GEMM_EPI_SPLIT = 0,
    // block rows are (64 tanh, 64 sigmoid) per wave-row: dst0[c] = tanh(u_t) * sigmoid(u_s)
    GEMM_EPI_GATE = 1,
    // same row pairing, (re, im) rows: dst0[c] = sqrt(re^2 + im^2)          (STFT magnitude)
    GEMM_EPI_MAG = 2,
    // dst0[row] = log(max(acc + bias, clip))                                 (mel projection + log)
    GEMM_EPI_LOG = 3,
};

__host__ __device__ inline bool gemm_epi_is_pair(int epi) { return epi == GEMM_EPI_GATE || epi == GEMM_EPI_MAG; }

struct GemmSeg {
    const float* base;    // [B][rows][ld] padded layout
    long long bstride;    // floats between batch items
    int nch;              // K chunks (of GEMM_KC rows) this segment contributes
    int shift;            // column shift (time steps)
    int mb_rows;          // extra row offset per M-block (block-diagonal batched GEMMs)
    int aligned;          // 1 if (pad + shift) % 4 == 0 -> 16-byte loads
};

struct GemmArgs {
    const float* A;       // packed [MB][nch_total][GEMM_KC][GEMM_BM]
    const float* bias;    // [MB*GEMM_BM] in block-local row order
    GemmSeg seg[GEMM_MAX_SEG];
    int nseg;
    int nch_total;
    int ld, pad, L;       // row stride, left halo, valid columns
    int ntiles, MB, batch;
    int M;                // valid rows (multiple of 32); rows >= M of the last M-block are padding
    float* dst0; long long dst0_bstride; int acc0;
    float* dst1; long long dst1_bstride; int acc1;
    int split;            // GEMM_EPI_SPLIT row split; pair epilogues: unused
    int pairC;            // pair epilogues: number of valid channels (rows c and pairC + c of the dense matrix)
    int dst_ld, dst_pad;  // row stride / left pad of the destination tensors (usually == ld, pad)
    float clip;           // GEMM_EPI_LOG clamp
};

// Row of the dense weight matrix held by block-local row r of M-block mb.
// Pair epilogues (GATE, MAG) interleave so that each wave-row (128 rows) owns 64 tanh channels and
// the 64 matching sigmoid channels (dense rows c and C + c).
// Returns -1 for padding rows (their packed weights and bias are zero).
__host__ __device__ inline int gemm_dense_row(int epi, int mb, int r, int C, int M) {
    if (gemm_epi_is_pair(epi)) {
        const int wm = r >> 7, rr = r & 127;
        const int c = mb * 128 + wm * 64 + (rr & 63);
        if (c >= C) return -1;
        return (rr < 64) ? c : C + c;
    }
    const int row = mb * GEMM_BM + r;
    return row < M ? row : -1;
}

int launch_gemm_f32(int epi, const GemmArgs& a, hipStream_t stream);

}  // namespace ctts
