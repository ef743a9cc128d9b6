// bf16 MFMA conv-GEMM (fp32 accumulate) for the WN in-layer and res/skip contractions (config 3).
//
// Same contract as gemm_f32.h, different storage:
//   * activations are "K8-blocked" bf16: [B][C/8][ld][8] - the 8 channels of a group are the 16 bytes of
//     one time step, so an MFMA B fragment (8 consecutive k for one column) is ONE 16-byte load, a
//     dilated tap is a whole-unit column shift (always 16-byte aligned), and the 32x32 accumulator tile
//     stores 8-byte runs that pair up into contiguous 512-byte rows;
//   * weights are packed [MB][K/32][4][256][8] bf16 = the LDS image of one A stage.
// v_mfma_f32_32x32x16_bf16: bf16 inputs, exact fp32 accumulation.
#pragma once

#include "common.h"

namespace ctts {

constexpr int BGEMM_BM = 256;
constexpr int BGEMM_BN = 128;
constexpr int BGEMM_KC = 32;     // K per LDS stage (two 32x32x16 MFMA k-steps)
constexpr int BGEMM_MAX_SEG = 12;    // split-bf16 in-layer GEMM: 3 taps x 3 (hi/lo) products + 3 cond products
constexpr int BGEMM_MAX_CHUNKS = 253; // K chunks per launch (chunk -> address table in LDS)

enum BGemmEpilogue : int { BGEMM_EPI_SPLIT = 0, BGEMM_EPI_GATE = 1 };

typedef unsigned short bf16_t;   // raw bits

struct BGemmSeg {
    const bf16_t* base;   // K8-blocked [B][rows/8][ld][8]
    long long bstride;    // elements between batch items
    int nch;              // K chunks (32 channels each)
    int shift;            // column shift (time steps)
    int mb_rows;          // extra channel offset per M-block (multiple of 8; block-diagonal batched GEMMs)
};

struct BGemmArgs {
    const bf16_t* A;      // packed [MB][nch_total][4][256][8]
    const float* bias;    // fp32 [MB*256], block-local row order
    BGemmSeg seg[BGEMM_MAX_SEG];
    int nseg, interleave, nch_total;
    int ld, pad, L, ntiles, MB, batch;
    int M;                // valid rows (multiple of 32)
    bf16_t* dst0; long long dst0_bstride; int acc0;
    bf16_t* dst1; long long dst1_bstride; int acc1;
    int split;            // SPLIT: rows < split -> dst0, else dst1[row - split] (multiple of 32)
    int pairC;            // GATE: channels (dense rows c and pairC + c)
    int map_mode;         // block id -> (m-block, tile, batch) mapping, chosen by the launcher
    // Element format of A, of every K segment and of the destinations: 0 = bf16 (config 3), 1 = IEEE half ("f16": the
    // reference's own reduced-precision mode, glow.py:343; 11-bit significands - inside the 1e-3 waveform bound where one
    // bf16 product per MAC is not).  Same layouts, same kernels; v_mfma_f32_32x32x16_f16 instead of ..._bf16, fp32 accumulate.
    int f16;
    // Split-bf16 ("bf16x3") destinations: lo_off != 0 -> every destination tensor is a PAIR of planes, hi at dst and lo
    // at dst + lo_off (elements): the epilogue stores hi = bf16(v), lo = bf16(v - hi) and read-modify-write
    // destinations are read as hi + lo.  The K side of the split is expressed with segments (x_hi*W_hi + x_lo*W_hi +
    // x_hi*W_lo as three segments over the two planes and two packed weight parts).
    long long lo_off;
};

// dense weight row of block-local row r of M-block mb (same pairing as the fp32 kernel), -1 = padding
__host__ __device__ inline int bgemm_dense_row(int epi, int mb, int r, int C, int M) {
    if (epi == BGEMM_EPI_GATE) {
        const int wm = r >> 7, rr = r & 127;
        const int c = (mb * 2 + wm) * 64 + (rr & 63);
        if (c >= C) return -1;
        return (rr < 64) ? c : C + c;
    }
    const int row = mb * BGEMM_BM + r;
    return row < M ? row : -1;
}

int launch_gemm_bf16(int epi, const BGemmArgs& a, hipStream_t stream);

// fp32 dense weights -> packed bf16 A (same source addressing as launch_pack_a of the fp32 path; K slabs
// are 32 wide here, so an interleaved member's slab j lands at slab j*k_group + k_member)
// part: 0 = bf16(w) (round to nearest even), 1 = bf16(w - bf16(w)) (the low half of the split-bf16 form)
int launch_pack_a_bf16(bf16_t* dst, const float* src, int MB, int nch_total, int k_off, int ksrc, int epi, int C,
                       int M, long long src_row_off, long long src_row_stride, int src_k_stride, hipStream_t s,
                       int k_group = 1, int k_member = 0, int part = 0, int f16 = 0);

__host__ __device__ inline bf16_t f32_to_bf16_rne(float f) {
    union { float f; unsigned int u; } v;
    v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((v.u >> 16) | 0x40);   // NaN stays NaN
    return (bf16_t)((v.u + 0x7fffu + ((v.u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
    union { float f; unsigned int u; } v;
    v.u = (unsigned int)h << 16;
    return v.f;
}

// IEEE half (round to nearest even; overflow -> inf, subnormals kept): raw bits in the same 16-bit storage type.  Device only.
__device__ __forceinline__ bf16_t f32_to_f16_rne(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float f16_to_f32(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// two fp32 -> packed bf16 (round to nearest even): one v_cvt_pk_bf16_f32 on the device
__host__ __device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned int, b);
#else
    return (unsigned int)f32_to_bf16_rne(lo) | ((unsigned int)f32_to_bf16_rne(hi) << 16);
#endif
}

// two fp32 -> packed IEEE half (round to nearest even)
__device__ __forceinline__ unsigned int pack_f16x2(float lo, float hi) {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2h_t __attribute__((ext_vector_type(2)));
    const f32x2h_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_t));
}
// format-generic forms (F16: a compile-time choice of the kernel instantiation)
template <bool F16> __device__ __forceinline__ unsigned int pack_h2(float lo, float hi) {
    if constexpr (F16) return pack_f16x2(lo, hi); else return pack_bf16x2(lo, hi);
}
template <bool F16> __device__ __forceinline__ float h_to_f32(bf16_t h) {
    if constexpr (F16) return f16_to_f32(h); else return bf16_to_f32(h);
}
template <bool F16> __device__ __forceinline__ bf16_t f32_to_h(float f) {
    if constexpr (F16) return f32_to_f16_rne(f); else return f32_to_bf16_rne(f);
}

}  // namespace ctts
