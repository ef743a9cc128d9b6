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
    // GATE with the unit chosen at run time (GemmArgs.gate != GATE_GTU): dst0[c] = gate_eval(gate, u_first, u_second)
    GEMM_EPI_GATEX = 7,
    // GATE followed, in the same workgroup, by the res/skip 1x1 GEMM on the gated tile held in registers
    // (bm = 128 shape with all pairC <= 64 channels in one wave-row): the activations never touch HBM.
    //   rows < split of  rs_w . act + rs_b  -> dst0 = src0 + .   (next layer's input)
    //   rows >= split                       -> dst1 (+)= .       (skip sum)
    GEMM_EPI_GATE_RS = 6,
};

__host__ __device__ inline bool gemm_epi_is_pair(int epi) {
    return epi == GEMM_EPI_GATE || epi == GEMM_EPI_MAG || epi == GEMM_EPI_GATE_RS || epi == GEMM_EPI_GATEX;
}

// The gated units of glow_ax.py:36-165 (get_gate_func :168-198): acts = f(in[:C]) * g(in[C:]).  Kind 0 (GTU) is
// the hot path with its own epilogue; the others go through GEMM_EPI_GATEX.
enum GateKind : int {
    GATE_GTU = 0,      // tanh * sigmoid
    GATE_GTRU = 1,     // tanh * relu
    GATE_GTLRU = 2,    // tanh * leaky_relu(0.01)
    GATE_GLU = 3,      // x * sigmoid
    GATE_TTU = 4,      // tanh * tanh
    GATE_STU = 5,      // tanh * selu
    GATE_GTSU = 6,     // tanhshrink * sigmoid
    GATE_SPTU = 7,     // tanh * softplus
    GATE_GSIU = 8,     // sin * sigmoid
    GATE_GSIRU = 9,    // sin(16 x) * sigmoid
    GATE_GTSRU = 10,   // tanhshrink * relu
    GATE_GSIRRU = 11,  // sin(16 x) * relu
    GATE_GSIRLRU = 12, // sin(16 x) * leaky_relu(0.01)
    GATE_GSIRRLRU = 13,// sin(16 x) * rrelu(0.01, 0.1) in eval mode = leaky_relu((0.01 + 0.1) / 2)
    GATE_KINDS = 14
};

#if defined(__HIPCC__)
// linear interpolation of the GATE epilogues' frame-rate addend, spelled out so that every kernel shape rounds alike
// (left to the compiler, l0 * a + l1 * b contracts to either fma(l0, a, l1 * b) or fma(l1, b, l0 * a))
__device__ __forceinline__ float gemm_lerp(float l0, float a, float l1, float b) { return __builtin_fmaf(l0, a, l1 * b); }

// compile-time unit; callers switch on the run-time kind OUTSIDE their element loops
template <int KIND>
__device__ __forceinline__ float gate_eval(float u0, float u1) {
    float f, g;
    if constexpr (KIND == GATE_GLU) f = u0;
    else if constexpr (KIND == GATE_GTSU || KIND == GATE_GTSRU) f = u0 - tanhf(u0);
    // __sinf: v_sin_f32 on x / 2pi (abs error ~1e-6 for the |arguments| < ~1e2 a WN layer produces); libm's sinf
    // drags a Payne-Hanek table into scratch
    else if constexpr (KIND == GATE_GSIU) f = __sinf(u0);
    else if constexpr (KIND == GATE_GSIRU || KIND == GATE_GSIRRU || KIND == GATE_GSIRLRU || KIND == GATE_GSIRRLRU) f = __sinf(16.0f * u0);
    else f = tanhf(u0);
    if constexpr (KIND == GATE_GTRU || KIND == GATE_GTSRU || KIND == GATE_GSIRRU) g = fmaxf(u1, 0.0f);
    else if constexpr (KIND == GATE_GTLRU || KIND == GATE_GSIRLRU) g = u1 > 0.0f ? u1 : 0.01f * u1;
    else if constexpr (KIND == GATE_GSIRRLRU) g = u1 >= 0.0f ? u1 : ((0.01f + 0.1f) / 2.0f) * u1;
    else if constexpr (KIND == GATE_TTU) g = tanhf(u1);
    else if constexpr (KIND == GATE_STU)
        g = 1.0507009873554804934193349852946f * (fmaxf(u1, 0.0f) + fminf(0.0f, 1.6732632423543772848170429916717f * expm1f(u1)));
    else if constexpr (KIND == GATE_SPTU) g = u1 > 20.0f ? u1 : log1pf(expf(u1));
    else g = 1.0f / (1.0f + expf(-u1));
    return f * g;
}
#endif

struct GemmSeg {
    const float* base;    // [B][rows][ld] padded layout
    long long bstride;    // floats between batch items
    int nch;              // K chunks (of GEMM_KC rows) this segment contributes
    int shift;            // column shift (time steps)
    int mb_rows;          // extra row offset per M-block (block-diagonal batched GEMMs)
    int fresh;            // wf_row_persistent_kernel only: rows written by other workgroups of the SAME launch (read at agent scope)
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
    int gate;             // GateKind of a GATE launch (0 = GTU); != 0 is routed to GEMM_EPI_GATEX by the launcher
    int gemm_mode;        // CTTS_GEMM_DEFAULT (0: the library default, ctts_set_f32_gemm_mode) | CTTS_GEMM_F32 | CTTS_GEMM_BF16X3:
                          // the caller's config struct carries it (ABI 4), so two models of one process can differ.  The
                          // STFT always asks for CTTS_GEMM_F32: its sums cancel, the split loop's 2^-17 operand error
                          // would exceed the 1e-4 log-mel bound
    const float* rs_wT;   // GEMM_EPI_GATE_RS: res/skip weight transposed and row-padded: [64][128]
    const float* rs_bias; // [128] (rows >= rs_rows zero)
    int rs_rows;          // 128 (res + skip) or 64 (last layer: skip only)
    int gt_limit;         // set by the launcher: column tiles (tile + ntiles * batch item) this launch covers, counted from 0 (the
                          // tiles behind it went to a separate small-shape launch: launch_gemm_f32, "round-aligned")
    long long shape_blocks;   // > 0: the launch is a PART of a larger one (a column region): choose the block shape as if the
                          // grid had this many large blocks, so that every part sums K in the order of the whole launch
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

// defaults launch_gemm_f32 fills in before a kernel sees the arguments (anything that launches a tile body itself does the same)
inline void gemm_apply_defaults(GemmArgs& a) {
    if (a.bm == 0) a.bm = 256;
    if (a.addend_ld == 0) { a.addend_ld = a.ld; a.addend_pad = a.pad; }
}
int gemm_check_args(int epi, const GemmArgs& a);      // the shape-independent argument checks of launch_gemm_f32 (defaults applied)
int launch_gemm_f32(int epi, const GemmArgs& a, hipStream_t stream);
// Small-problem shape (gemm_f32_small.hip): 128 x 64 blocks / 64 x 32 wave tiles on the SAME packed operands, bit-identical
// results; chosen by launch_gemm_f32 when the 256 x 128 shape would start fewer than two blocks per CU.
bool gemm_f32_small_applies(int epi, const GemmArgs& a);
int launch_gemm_f32_small(int epi, const GemmArgs& a, hipStream_t stream);

// WaveFlow row step as ONE launch (gemm_f32_small.hip, wf_row_persistent_kernel): the GEMM_EPI_GATE_RS layers of a row as items of
// a work queue, tile (layer i + 1, t) waiting for tiles t - 1, t, t + 1 of layer i only.  `layers_dev`: the row's GemmArgs in
// device memory, exactly what launch_gemm_f32(GEMM_EPI_GATE_RS, ...) would have been given layer by layer, with GemmSeg.fresh
// set on the segments the previous layer of the same row writes.  Needs |shift| <= 128 on fresh segments.
bool wf_row_persistent_supported(const GemmArgs& a);
// body 0: items = 128 x 128 tiles (same bits as the 128 x 128 / 128 x 256 per-layer shapes); body 1: 128 x 64 tiles of the split-K
// shape (same bits as THAT per-layer shape)
int wf_row_cus();                             // CUs of the current device
int wf_row_tiles(int L, int body);            // tiles per batch item (the flag array's inner extent)
// tails_dev != NULL: the whole-flow form - `nrows` rows chained in one launch, layers_dev = [nrows][nlayers], a tail stage (end conv,
// affine update, next row's start conv: waveflow_tail.h) behind every row; the flag array then holds nrows * (nlayers + 1) stages
struct WfTailDesc;
int launch_wf_row_persistent(const GemmArgs* layers_dev, const WfTailDesc* tails_dev, int nrows, int nlayers, int max_nseg, int L,
                             int batch, int body, unsigned int* counter, unsigned int* flags, unsigned int* abort_word,
                             unsigned int epoch, hipStream_t stream);

// Library DEFAULT of the main-loop selection (what CTTS_GEMM_DEFAULT resolves to), in the config structs' own encoding:
// CTTS_GEMM_F32 (initially), CTTS_GEMM_BF16X3 (three bf16 MFMA products per fp32 operand pair, see
// conv_gemm_f32_kernel<..., X3>) or CTTS_GEMM_BF16X6.  ctts_set_f32_gemm_mode (deprecated: prefer the per-model field).
int get_gemm_f32_mode();
// what the calling thread's most recent conv-GEMM launch ran (ctts_last_gemm_loop): bits 0-3 split level, 16 small shape, 32 split-K, 64 row queue, 128 its whole-flow form
void note_gemm_loop(int code);
int last_gemm_loop();
// true when a launch with this GemmArgs.gemm_mode / config f32_gemm_mode runs the split-bf16 main loop
bool gemm_mode_is_split(int gemm_mode);
// validates a config struct's f32_gemm_mode field
inline bool gemm_mode_valid(int m) { return m == CTTS_GEMM_DEFAULT || m == CTTS_GEMM_F32 || m == CTTS_GEMM_BF16X3 || m == CTTS_GEMM_BF16X6; }
// 0: fp32 MFMA; 3 / 6: bf16 products per operand pair of the split loops (the library default resolved)
int gemm_split_level(int gemm_mode);

}  // namespace ctts
