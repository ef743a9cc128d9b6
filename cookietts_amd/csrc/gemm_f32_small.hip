// Small-problem shape of the fp32 MFMA conv-GEMM (same contract, same packed operands, same bits as gemm_f32.hip).
//
// Why: the 256 x 128 block of gemm_f32.hip gives every wave a 128 x 64 tile.  A launch with fewer than 1024 such wave
// tiles leaves SIMDs idle however its blocks are shaped, and a wave alone on its SIMD keeps the matrix pipe only ~68 %
// busy (nobody covers its LDS waits and chunk barriers).  Measured on the ax WaveGlow of the author's notebook at B = 1
// (profiles/r3_02_pmc_ax_notebook_b1.json): in-layer GEMM 184 blocks = 736 wave tiles, matrix pipe busy 0.49; res/skip
// GEMM 92 blocks, 0.29.  The Tacotron encoder / postnet convolutions, the conditioning stacks at frame rate and the
// STFT are smaller still.
//
// Here a block is 128 x 64 and a wave tile 64 x 32 (2 x 1 tiles of v_mfma_f32_32x32x2_f32): four times as many wave tiles,
// 40 KB of LDS per block, so four blocks share a CU and 3-4 waves a SIMD.
//   * Operands are NOT repacked: a block reads one 128-row half (wave-row `half` of M-block `mb`) of the 256-row packing.
//     For the pair epilogues that half holds [64 first-half channels | the 64 matching second-half channels]; wave wm
//     takes rows 32 wm .. +32 (first) and 64 + 32 wm .. +32 (second), so the pairing still happens in registers.
//   * Every output element sums the same chunks and k-steps in the same order with the same instruction as the large
//     shape: results are bit-identical (tests/test_conv1d_primitive.py, test_waveglow_ax.py).
//   * Staging: global -> LDS DMA (16 B per lane, per-lane source addresses: the A half is 16 runs of 512 B), three
//     stages, two chunks ahead, three DMA pieces per chunk per wave, counted vmcnt + s_barrier.
#include <algorithm>
#include <atomic>
#include <mutex>

#include <type_traits>

#include "gemm_f32.h"
#include "tuning.h"
#include "waveflow_tail.h"
#include "gemm_bf16.h"   // pack_bf16x2 (split-bf16 main loop)

namespace ctts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int S_BM = 128, S_BN = 64;
constexpr int S_ASTAGE = GEMM_KC * S_BM;                 // 2048 floats
constexpr int S_STAGE = GEMM_KC * (S_BM + S_BN);         // 3072 floats = 12 KiB
constexpr int S_MAX_CHUNKS = 384;                        // chunk -> B address table entries (same bound as gemm_f32.hip): 40 128 B of
                                                         // LDS per block still lets four blocks share a CU; the k = 5 cond convs
                                                         // of the author's WaveFlow stack are 290 chunks
constexpr int S_NST = 3;
// the fused WaveFlow layer takes the small shape below this many 128 x 256 blocks (set from the B = 1 / 2 / 8 measurements)
constexpr long long SMALL_BELOW_LARGE_BLOCKS = 2048;      // generic shape: taken below this many 256 x 128 blocks
// the same for the 128-row packing (128 x 256 blocks; WaveFlow above 64 channels, round 5): against the 1 x 4-wave 128 x 256 kernel the
// small shape was ahead at every size measured (profiles/r5_44): batch 1, 88 blocks (20 groups, 128 channels) 240 -> 113 ms,
// 72 blocks (50 groups, 256 channels) 1117 -> 534, 864 blocks (8 groups, 512 channels) 979 -> 854; 1728 blocks 1688 -> 1649;
// 2592 blocks (8 groups, 128 channels, batch 12) 779 -> 711
constexpr long long SMALL_BELOW_LARGE_BLOCKS_BM128 = 4096;
constexpr long long GATE_RS_SMALL_BELOW_BLOCKS = 256;
// ... and the split-K shape (128 x 64 blocks, K halves on wave pairs) up to this many 128 x 256 blocks (batch 1-2 of config 4)
constexpr long long GATE_RS_SPLITK_MAX_BLOCKS = 128;
constexpr int S_SEGTAB = S_NST * S_STAGE;
constexpr int S_CHTAB = S_SEGTAB + GEMM_MAX_SEG * 4;
constexpr int S_LDS_FLOATS = S_CHTAB + 2 * S_MAX_CHUNKS;

__device__ __forceinline__ float s_fast_sigmoid(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
}
__device__ __forceinline__ float s_fast_tanh(float u) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * 2.8853900817779268f));
}

typedef const __attribute__((address_space(1))) float* gfloat_ptr;
typedef __attribute__((address_space(3))) float* lds_fptr;
typedef const __attribute__((address_space(1))) char* gbyte_ptr;

// One 16-byte-per-lane global -> LDS DMA whose address is a wave-uniform 64-bit base (SGPR pair) + a 32-bit lane byte
// offset: global_load_lds_dwordx4 v, s[a:b].  The lane offset is laundered through an empty asm so that the compiler
// cannot fold it into a 64-bit VGPR address (global_load_lds_dwordx4 v[a:b], off): with that form every DMA keeps the
// matrix pipe from issuing for ~20 cycles (round 5, profiles/r5_07_bf16_mix_ceiling_dma_forms.txt).
#define CTTS_GLDS_U(ubase, lane_off, ldsdst, aux)                                                                \
    do {                                                                                                         \
        unsigned o_ = (lane_off);                                                                                \
        asm volatile("" : "+v"(o_));                                                                             \
        __builtin_amdgcn_global_load_lds((gfloat_ptr)((ubase) + o_), (ldsdst), 16, 0, (aux));                    \
    } while (0)

// A launch of these kernels starts with a cold scalar cache and a 600-byte argument struct that the prologue reads field by
// field: five to six DEPENDENT s_load batches, each a miss (~0.5 us), before the first operand request goes out (3.6 of a
// 37 us batch-1 WaveFlow layer: profiles/r5_31).  Touching one dword of every 64-byte line of the kernarg segment at entry
// turns them into one parallel miss; the later loads hit.
// (hipcc emits them as two batches, each ahead of a first use; a hand-written block of all ten loads does not come out earlier -
// the kernel's own argument loads are hoisted above it - so the plain form stays.)
template <int BYTES>
__device__ __forceinline__ unsigned warm_kernargs() {
    const unsigned __attribute__((address_space(4)))* kp =
        (const unsigned __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned x = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) x ^= kp[o / 4];
    return x;
}
#define CTTS_WARM_KERNARGS(T)                                                                                    \
    do {                                                                                                         \
        const unsigned w_ = warm_kernargs<(int)sizeof(T)>();                                                     \
        asm volatile("" ::"s"(w_));                                                                              \
    } while (0)

// split-bf16 main loop (X3), exactly as in conv_gemm_f32_kernel<..., X3>: the 8 k-values a lane reads per fragment and chunk
// become one bf16x8 operand pair hi = bf16(v), lo = bf16(v - hi); a 32x32 tile of the chunk is lo*hi + hi*lo + hi*hi on
// v_mfma_f32_32x32x16_bf16, in that order - so this shape stays bit-identical to the large one in split mode too
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int s_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s_split8(const float (&v)[8], s_u32x4& hi, s_u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int h = pack_bf16x2(v[2 * j], v[2 * j + 1]);
        hi[j] = h;
        lo[j] = pack_bf16x2(v[2 * j] - __builtin_bit_cast(float, h << 16), v[2 * j + 1] - __builtin_bit_cast(float, h & 0xffff0000u));
    }
}

// scripts/micro/small_gemm_timeline.hip compiles this file with CTTS_SMALL_GEMM_STAMPS to get a per-block timeline
// (s_memrealtime at entry / tables built / first chunk landed / main loop done / epilogue operands landed / stores
// acknowledged); the library build never defines it
#ifdef CTTS_SMALL_GEMM_STAMPS
__device__ unsigned long long* g_small_stamps;
#define S_STAMP(k)                                                                                               \
    do {                                                                                                         \
        if (threadIdx.x == 0) g_small_stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)
#define S_STAMP_DRAIN(k)                                                                                         \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
        S_STAMP(k);                                                                                              \
    } while (0)
// slot 6: HW_ID (cu_id [11:8], sh_id [12], se_id [15:13]) | XCC_ID << 32: which CU the block landed on
#define S_STAMP_WHERE()                                                                                          \
    do {                                                                                                         \
        if (threadIdx.x == 0)                                                                                    \
            g_small_stamps[(size_t)blockIdx.x * 8 + 6] =                                                         \
                (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                                  \
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);                          \
    } while (0)
#else
#define S_STAMP_WHERE()
#define S_STAMP(k)
#define S_STAMP_DRAIN(k)
#endif

__device__ __forceinline__ void s_split8x3(const float (&v)[8], s_u32x4& hi, s_u32x4& mid, s_u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int h = pack_bf16x2(v[2 * j], v[2 * j + 1]);
        const float r0 = v[2 * j] - __builtin_bit_cast(float, h << 16), r1 = v[2 * j + 1] - __builtin_bit_cast(float, h & 0xffff0000u);
        const unsigned int m = pack_bf16x2(r0, r1);
        hi[j] = h;
        mid[j] = m;
        lo[j] = pack_bf16x2(r0 - __builtin_bit_cast(float, m << 16), r1 - __builtin_bit_cast(float, m & 0xffff0000u));
    }
}

// XS: 0 = fp32 MFMA, 3 / 6 = split-bf16 products per operand pair (gemm_f32.hip, X3 / X6)
template <int EPI, int SEGS, int XS = 0>
__global__ __launch_bounds__(256, 4) void conv_gemm_f32_small_kernel(const GemmArgs a, const int ntiles_s) {
    constexpr bool X3 = XS != 0, X6 = XS == 6;
    __shared__ __attribute__((aligned(16))) float lds[S_LDS_FLOATS];
    constexpr bool PAIR = EPI == GEMM_EPI_GATE || EPI == GEMM_EPI_GATEX || EPI == GEMM_EPI_MAG;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    S_STAMP(0);
    S_STAMP_WHERE();
    // 128-row slice `mbs` of the output rows: half `half` of M-block mb in the 256-row packing, or the whole M-block mbs of the
    // 128-row packing (WaveFlow's GEMMs above 64 channels, round 5: at batch 1 their 128 x 256 launches were 44-176 blocks)
    int id = blockIdx.x;
    const bool bm128 = a.bm == 128;
    const int nmb = bm128 ? a.MB : 2 * a.MB;
    const int mbs = id % nmb;
    id /= nmb;
    const int tile = id % ntiles_s;
    const int b = id / ntiles_s;
    const int mb = bm128 ? mbs : mbs >> 1, half = bm128 ? 0 : mbs & 1;
    const int n0 = tile * S_BN;

    // SPLIT-family epilogues add the destination's previous contents (x += res, skip += ...).  Nothing in this launch
    // writes this block's tile before its own epilogue, so those 2 x 16 values per lane are requested FIRST: they travel
    // while the tables are built and the first chunks stream in (loads return in order, so the first counted vmcnt wait
    // of the main loop covers them) instead of as an exposed round trip of every block at the end of the launch - all
    // blocks of a short launch reach their epilogues together (scripts/micro/small_gemm_timeline.hip: 6.9 us of 41)
    float old[PAIR ? 1 : 2][16];
    if constexpr (!PAIR) {
        const int n = n0 + wn * 32 + l31;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int rbase = mbs * 128 + 32 * wm + 64 * mt;
            const bool second = rbase >= a.split;
            const float* dstc = second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride;
            const float* src = second ? dstc : (a.src0 ? a.src0 + (size_t)b * a.src0_bstride : dstc);
            const int accum = second ? a.acc1 : a.acc0;
            const int rdst = second ? rbase - a.split : rbase;
            // one unconditional form (two differently guarded forms would meet in register copies, which wait for the
            // loads): lanes past the last column / rows past M re-read the last valid one and are never stored
            if (accum && rbase < a.M) {
                const float* sp = src + (size_t)rdst * a.dst_ld + a.dst_pad + min(n, a.L - 1);
                const int rlast = a.M - 1 - rbase;
#pragma unroll
                for (int r = 0; r < 16; ++r) old[mt][r] = sp[(size_t)min((r & 3) + 8 * (r >> 2) + 4 * lhi, rlast) * a.dst_ld];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) old[mt][r] = 0.0f;
            }
        }
    }

    // GATE-family epilogues add a conditioning term (the ax core: frame-rate rows, linearly interpolated - glow_ax.py:362-373).
    // In the fp32 loop its operands are requested here for the same reason as `old` above (the split loops have no
    // registers to spare: they request them in the epilogue); the arithmetic stays in the epilogue either way.
    constexpr bool HAS_ADDEND = EPI == GEMM_EPI_GATE || EPI == GEMM_EPI_GATEX;
    constexpr bool HOIST_ADDEND = HAS_ADDEND && XS == 0;
    [[maybe_unused]] float adv[HAS_ADDEND ? 4 : 1][16];
    [[maybe_unused]] float ad_l0 = 1.0f, ad_l1 = 0.0f;
#define S_LOAD_ADDEND()                                                                                          \
    do {                                                                                                         \
        const int n_ = n0 + wn * 32 + l31;                                                                       \
        const int cb_ = mbs * 64 + 32 * wm;                                                                      \
        if (a.addend && cb_ < a.pairC) {                                                                         \
            const int nc_ = min(n_, a.L - 1);               /* lanes past the last column re-read it (never stored) */ \
            int i0_ = nc_, i1_ = nc_;                                                                            \
            if (a.addend_frames > 0) {                                                                           \
                const int F_ = a.addend_frames;                                                                  \
                const float scale_ = a.L > 1 ? (float)(F_ - 1) / (float)(a.L - 1) : 0.f;                         \
                const float real_ = scale_ * (float)n_;                                                          \
                i0_ = min((int)real_, F_ - 1);                                                                   \
                i1_ = i0_ + 1 < F_ ? i0_ + 1 : F_ - 1;                                                           \
                ad_l1 = real_ - (float)(int)real_;                                                               \
                ad_l0 = 1.0f - ad_l1;                                                                            \
            }                                                                                                    \
            const float* ad_ = a.addend + (size_t)b * a.addend_bstride + a.addend_pad;                           \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                     \
                const int c_ = min(cb_ + (r & 3) + 8 * (r >> 2) + 4 * lhi, a.pairC - 1);                         \
                const float* r0_ = ad_ + (size_t)c_ * a.addend_ld;                                               \
                const float* r1_ = ad_ + (size_t)(a.pairC + c_) * a.addend_ld;                                   \
                adv[0][r] = r0_[i0_]; adv[1][r] = r0_[i1_]; adv[2][r] = r1_[i0_]; adv[3][r] = r1_[i1_];          \
            }                                                                                                    \
        } else {                                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) adv[0][r] = adv[1][r] = adv[2][r] = adv[3][r] = 0.0f; \
        }                                                                                                        \
    } while (0)
    if constexpr (HOIST_ADDEND) S_LOAD_ADDEND();

    const int nalloc = a.a_nch_alloc ? a.a_nch_alloc : a.nch_total;
    // A: chunk c of M-block mb in the 256-row packing, rows [128 half, +128).  Piece p of this wave fills LDS floats
    // [wave * 256 + p * 1024, +256) of the [16][128] stage: k-row 2 wave + 8 p + (lane >> 5), columns 4 (lane & 31).
    // Every DMA address below is a wave-uniform 64-bit base + a 32-bit lane byte offset (CTTS_GLDS_U): the form
    // global_load_lds_dwordx4 v, s[a:b], which does not block the matrix pipe the way the 64-bit VGPR address form does.
    // (128-row packing: the [16][128] chunk is contiguous - k-row stride 128 floats - and copied linearly)
    const int a_ld = bm128 ? 128 : 256;                    // floats between the k-rows of a packed chunk
    const unsigned a_cstride = (unsigned)(GEMM_KC * a_ld * 4);
    const gbyte_ptr a_base = (gbyte_ptr)(a.A + ((size_t)mb * nalloc + a.a_ch_off) * (GEMM_KC * a_ld) + 128 * half + 2 * wave * a_ld);
    const unsigned dma_a_lane = (unsigned)(((lane >> 5) * a_ld + (lane & 31) * 4) * 4), dma_a_lane1 = dma_a_lane + 8 * a_ld * 4;   // piece 0 / 1
    // B: [16][64] stage, this wave's piece = k-rows 4 wave .. +4: k-row 4 wave + (lane >> 4), columns 4 (lane & 15)
    const unsigned dma_b_lane = (unsigned)(((size_t)(4 * wave + (lane >> 4)) * a.ld + (lane & 15) * 4) * 4);
    const unsigned long long* ctab = reinterpret_cast<const unsigned long long*>(lds + S_CHTAB);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

#define S_ISSUE_A(buf, c, p)                                                                                     \
    CTTS_GLDS_U(a_base + (size_t)(c) * a_cstride, (p) ? dma_a_lane1 : dma_a_lane,                                 \
                (lds_fptr)(lds + (buf) * S_STAGE + wave * 256 + (p) * 1024), 0)
#define S_ISSUE_B(buf, c)                                                                                        \
    do {                                                                                                         \
        const unsigned long long ub_ = ctab[c];                                                                  \
        const unsigned long long us_ =                                                                           \
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub_ >> 32)) << 32) |             \
            (unsigned)__builtin_amdgcn_readfirstlane((int)ub_);                                                  \
        CTTS_GLDS_U(reinterpret_cast<gbyte_ptr>(us_), dma_b_lane,                                                    \
                    (lds_fptr)(lds + (buf) * S_STAGE + S_ASTAGE + wave * 256), 0);                               \
    } while (0)

#define S_ISSUE_B_AT(buf, ub)                                                                                    \
    do {                                                                                                         \
        const unsigned long long ub_ = (ub);                                                                     \
        const unsigned long long us_ =                                                                           \
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub_ >> 32)) << 32) |             \
            (unsigned)__builtin_amdgcn_readfirstlane((int)ub_);                                                  \
        CTTS_GLDS_U(reinterpret_cast<gbyte_ptr>(us_), dma_b_lane,                                                    \
                    (lds_fptr)(lds + (buf) * S_STAGE + S_ASTAGE + wave * 256), 0);                               \
    } while (0)

    // Chunks 0 and 1 are requested BEFORE the tables are built (their B addresses follow from the first two segments by
    // the same rule as the table's): the first DMA round trip and the table build overlap instead of adding up.
    const int nch = a.nch_total;
    {
        const GemmSeg& g0 = a.seg[0];
        const GemmSeg& g1 = a.seg[1];
        const float* sb0 = g0.base + (size_t)b * g0.bstride + (size_t)(mb * g0.mb_rows) * a.ld + (a.pad + n0 + g0.shift);
        const float* sb1 = g1.base + (size_t)b * g1.bstride + (size_t)(mb * g1.mb_rows) * a.ld + (a.pad + n0 + g1.shift);
        const float* c1p = nch <= 1 ? sb0 : a.interleave > 1 ? sb1 : g0.nch > 1 ? sb0 + (size_t)GEMM_KC * a.ld : sb1;
        const int c1 = nch > 1 ? 1 : 0;
        S_ISSUE_A(0, 0, 0); S_ISSUE_A(0, 0, 1); S_ISSUE_B_AT(0, reinterpret_cast<unsigned long long>(sb0));
        S_ISSUE_A(1, c1, 0); S_ISSUE_A(1, c1, 1); S_ISSUE_B_AT(1, reinterpret_cast<unsigned long long>(c1p));
    }

    // chunk -> B address table, ONE barrier (round 5; until then: a segment table written by twelve serial branches, a barrier,
    // the chunk table through LDS reads of it, a barrier - 2.5 us, profiles/r5_32): every thread below nch_total finds the
    // segment of its chunk from the (uniform) chunk counts and reads that segment's base / batch stride / shift / row offset
    // with one lane-indexed load from the kernarg segment (the GemmArgs struct is the kernel's first parameter)
    {
        typedef const __attribute__((address_space(4))) char* s_kargp;
        const s_kargp kp = (s_kargp)__builtin_amdgcn_kernarg_segment_ptr();
        int nchs[SEGS];
#pragma unroll
        for (int k = 0; k < SEGS; ++k) nchs[k] = k < a.nseg ? a.seg[k].nch : 0x7fffffff;
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(lds + S_CHTAB);
        const int ilv0 = a.interleave > 1 ? a.interleave : 0;
        const int n_il = ilv0 * a.seg[0].nch;
        for (int c0 = t; c0 < a.nch_total; c0 += 256) {
            int c = c0, sg, loc;
            if (c < n_il) {
                sg = c % ilv0;
                loc = c / ilv0;
            } else {
                c -= n_il;
                sg = ilv0;
#pragma unroll
                for (int k = 0; k < SEGS - 1; ++k)                   // sequential walk over the non-interleaved segments
                    if (sg == k && k < a.nseg - 1 && c >= nchs[k]) { c -= nchs[k]; sg = k + 1; }
                loc = c;
            }
            const s_kargp sp = kp + (offsetof(GemmArgs, seg) + (size_t)sg * sizeof(GemmSeg));
            const unsigned long long sbase = *reinterpret_cast<const __attribute__((address_space(4))) unsigned long long*>(sp + offsetof(GemmSeg, base));
            const long long sbstr = *reinterpret_cast<const __attribute__((address_space(4))) long long*>(sp + offsetof(GemmSeg, bstride));
            const int sshift = *reinterpret_cast<const __attribute__((address_space(4))) int*>(sp + offsetof(GemmSeg, shift));
            const int smbr = *reinterpret_cast<const __attribute__((address_space(4))) int*>(sp + offsetof(GemmSeg, mb_rows));
            tab[c0] = sbase + (unsigned long long)(((long long)b * sbstr + (long long)(mb * smbr) * a.ld + (a.pad + n0 + sshift) +
                                                    (long long)loc * GEMM_KC * a.ld) * 4);
        }
    }
    __syncthreads();

    S_STAMP(1);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    S_STAMP(2);

    // fragment rows of this wave inside the 128-row half
    // (tile mt of wave wm = rows 32 wm + 64 mt for every epilogue: the pair epilogues need that pairing, and a fixed
    // 64-row distance lets one ds_read2st64_b32 fetch both A fragments of a k-step)
    const int arow0 = 32 * wm;
    const int arow1 = 64 + 32 * wm;
    int cur = 0;
    if constexpr (X3) {
        for (int ch = 0; ch < nch; ++ch) {
            const float* As = lds + cur * S_STAGE + l31;
            const float* Bs = lds + cur * S_STAGE + S_ASTAGE + wn * 32 + l31;
            const int nb = cur >= 1 ? cur - 1 : 2;              // (cur + 2) % 3: the stage chunk ch - 1 occupied
            const int cn = ch + 2 < nch ? ch + 2 : nch - 1;     // the last two iterations re-issue the final chunk
            s_u32x4 ah[2], al[2], bh, bl;
            [[maybe_unused]] s_u32x4 am[2], bm;
            {
                float v[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) v[ks] = Bs[(2 * ks + lhi) * S_BN];
                if constexpr (X6) s_split8x3(v, bh, bm, bl);
                else s_split8(v, bh, bl);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) v[ks] = As[(2 * ks + lhi) * S_BM + (mt == 0 ? arow0 : arow1)];
                if constexpr (X6) s_split8x3(v, ah[mt], am[mt], al[mt]);
                else s_split8(v, ah[mt], al[mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            S_ISSUE_A(nb, cn, 0); S_ISSUE_A(nb, cn, 1); S_ISSUE_B(nb, cn);
#define S_X3_P(A_, B_) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s_bf16x8, A_), __builtin_bit_cast(s_bf16x8, B_), acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                S_X3_P(al[mt], bh) S_X3_P(ah[mt], bl)
                if constexpr (X6) { S_X3_P(am[mt], bm) S_X3_P(am[mt], bh) S_X3_P(ah[mt], bm) }
                S_X3_P(ah[mt], bh)
            }
#undef S_X3_P
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");    // chunk ch + 1 landed, the newest still in flight
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == 2 ? 0 : cur + 1;
        }
    } else {
        // The MFMA stream runs THROUGH the chunk boundary: the wait-for-landing + barrier of chunk ch + 1 sits in front of
        // the last two k-steps of chunk ch, and the first two fragment reads of chunk ch + 1 are issued right behind
        // it, so a wave reaches the barrier with its k-step-5 MFMAs executing and starts the next chunk with operands that
        // were requested two MFMA pairs earlier.  Slot safety: at the barrier every wave has ISSUED all its reads of
        // chunk ch; the first DMA into that slot is issued a chunk later and its data arrive later still.
        //
        // The fragment reads and their waits are written by hand: with LDS-DMA in a loop hipcc turns every wait for an
        // LDS result into s_waitcnt lgkmcnt(0), i.e. each MFMA pair first drains the reads issued just in front of it
        // (scripts/micro/small_gemm_timeline.hip: main loop 79 us -> 60 us with reads, DMA issue and barrier removed).
        // LDS returns in order, so "k-step ks has arrived" is lgkmcnt(number of LDS instructions issued after it).
        // A k-step is two instructions: ds_read2st64_b32 (rows arow0 and arow0 + 64 of k-row 2 ks + lhi) and ds_read_b32.
        typedef float s_f32x2 __attribute__((ext_vector_type(2)));
        s_f32x2 av[GEMM_KC / 2];
        float bv[GEMM_KC / 2];
        const unsigned lds0 = (unsigned)(size_t)(lds_fptr)lds;
        const unsigned a_lane = lds0 + (unsigned)((lhi * S_BM + arow0 + l31) * 4);
        const unsigned b_lane = lds0 + (unsigned)((S_ASTAGE + lhi * S_BN + wn * 32 + l31) * 4);
#define S_READ_AT(ks, aaddr, baddr)                                                                              \
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(av[ks]) : "v"(aaddr), "n"(4 * (ks)), "n"(4 * (ks) + 1)); \
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bv[ks]) : "v"(baddr), "n"(2 * (ks) * S_BN * 4));
#define S_WAIT(n, ks) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(av[ks]), "+v"(bv[ks]));
#define S_MFMA(ks)                                                                                               \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][0], bv[ks], acc[0], 0, 0, 0);                       \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][1], bv[ks], acc[1], 0, 0, 0);
        S_READ_AT(0, a_lane, b_lane) S_READ_AT(1, a_lane, b_lane)
        for (int ch = 0; ch < nch; ++ch) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            const unsigned aa = a_lane + (unsigned)(cur * S_STAGE * 4), ba = b_lane + (unsigned)(cur * S_STAGE * 4);
            const unsigned an = a_lane + (unsigned)(nxt * S_STAGE * 4), bn = b_lane + (unsigned)(nxt * S_STAGE * 4);
            const int nb = cur >= 1 ? cur - 1 : 2;              // (cur + 2) % 3: the stage chunk ch - 1 occupied
            const int cn = ch + 2 < nch ? ch + 2 : nch - 1;     // the last two iterations re-issue the final chunk
            unsigned long long ub_next;                         // B address of chunk cn: one more in-order LDS read
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ub_next) : "v"((unsigned)(lds0 + cn * 8)), "n"(S_CHTAB * 4));
            // outstanding here: k-steps 0, 1 of this chunk (2 + 2) and the table entry (1)
            S_READ_AT(2, aa, ba) S_WAIT(5, 0) S_MFMA(0) S_ISSUE_A(nb, cn, 0);
            __builtin_amdgcn_sched_barrier(0);
            S_READ_AT(3, aa, ba) S_WAIT(5, 1) S_MFMA(1) S_ISSUE_A(nb, cn, 1);
            __builtin_amdgcn_sched_barrier(0);
            S_READ_AT(4, aa, ba) S_WAIT(4, 2)                   // (covers the older table entry)
            asm volatile("" : "+v"(ub_next));
            S_MFMA(2) S_ISSUE_B_AT(nb, ub_next);
            __builtin_amdgcn_sched_barrier(0);
            S_READ_AT(5, aa, ba) S_WAIT(4, 3) S_MFMA(3)
            __builtin_amdgcn_sched_barrier(0);
            S_READ_AT(6, aa, ba) S_WAIT(4, 4) S_MFMA(4)
            __builtin_amdgcn_sched_barrier(0);
            S_READ_AT(7, aa, ba) S_WAIT(4, 5) S_MFMA(5)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");    // chunk ch + 1 landed, the newest still in flight
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            S_READ_AT(0, an, bn) S_READ_AT(1, an, bn)           // (past the last chunk: a re-issued copy, never used)
            S_WAIT(6, 6) S_MFMA(6)
            __builtin_amdgcn_sched_barrier(0);
            S_WAIT(4, 7) S_MFMA(7)
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(bv[0]), "+v"(av[1]), "+v"(bv[1]));   // the two reads past the end
#undef S_READ_AT
#undef S_WAIT
#undef S_MFMA
    }
#undef S_ISSUE_A
#undef S_ISSUE_B
#undef S_ISSUE_B_AT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the re-issued tail DMAs still target LDS
    __builtin_amdgcn_s_barrier();
    S_STAMP(3);

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (t < S_BM) lds[t] = a.bias[mbs * 128 + t];
    __syncthreads();
    const int n = n0 + wn * 32 + l31;
    if constexpr (PAIR) {
        float* dst = a.dst0 + (size_t)b * a.dst0_bstride;
        const int cbase = mbs * 64 + 32 * wm;
        if (cbase < a.pairC && n < a.L) {
            float add0[16], add1[16];
            if constexpr (HAS_ADDEND && !HOIST_ADDEND) S_LOAD_ADDEND();
            if constexpr (HAS_ADDEND) {
                const bool interp = a.addend_frames > 0;    // (same arithmetic as the large shape, on operands requested at entry)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    add0[r] = interp ? gemm_lerp(ad_l0, adv[0][r], ad_l1, adv[1][r]) : adv[0][r];
                    add1[r] = interp ? gemm_lerp(ad_l0, adv[2][r], ad_l1, adv[3][r]) : adv[2][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) add0[r] = add1[r] = 0.0f;
            }
            S_STAMP_DRAIN(4);
            if constexpr (EPI == GEMM_EPI_GATEX) {
#define S_GATEX_LOOP(K)                                                                                           \
                case K:                                                                                           \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                              \
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;                                         \
                        const float u0 = acc[0][r] + lds[32 * wm + row] + add0[r];                                \
                        const float u1 = acc[1][r] + lds[64 + 32 * wm + row] + add1[r];                           \
                        if (cbase + row < a.pairC) dst[(size_t)(cbase + row) * a.dst_ld + a.dst_pad + n] = gate_eval<K>(u0, u1); \
                    }                                                                                             \
                    break;
                switch (a.gate) {
                    S_GATEX_LOOP(1) S_GATEX_LOOP(2) S_GATEX_LOOP(3) S_GATEX_LOOP(4) S_GATEX_LOOP(5) S_GATEX_LOOP(6) S_GATEX_LOOP(7)
                    S_GATEX_LOOP(8) S_GATEX_LOOP(9) S_GATEX_LOOP(10) S_GATEX_LOOP(11) S_GATEX_LOOP(12) S_GATEX_LOOP(13)
                    default:
                    S_GATEX_LOOP(0)
                }
#undef S_GATEX_LOOP
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float u0 = acc[0][r] + lds[32 * wm + row] + add0[r];
                    const float u1 = acc[1][r] + lds[64 + 32 * wm + row] + add1[r];
                    float v;
                    if constexpr (EPI == GEMM_EPI_GATE) v = s_fast_tanh(u0) * s_fast_sigmoid(u1);
                    else v = sqrtf(u0 * u0 + u1 * u1);
                    if (cbase + row < a.pairC) dst[(size_t)(cbase + row) * a.dst_ld + a.dst_pad + n] = v;
                }
            }
        }
    } else {
        // rows < split -> dst0 (= src0 + v when acc0), rows >= split -> dst1[row - split] (+= when acc1)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int rbase = mbs * 128 + 32 * wm + 64 * mt;                 // uniform per tile
            if (rbase >= a.M) continue;
            const bool second = rbase >= a.split;
            float* dst = second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride;
            const int rdst = second ? rbase - a.split : rbase;
            if (n < a.L) {
                if (mt == 1) { S_STAMP_DRAIN(4); }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float v = acc[mt][r] + lds[32 * wm + 64 * mt + row] + old[mt][r];
                    if constexpr (EPI == GEMM_EPI_LOG) v = logf(fmaxf(v, a.clip));
                    if constexpr (EPI == GEMM_EPI_LRELU) v = v > 0.f ? v : a.clip * v;
                    if constexpr (EPI == GEMM_EPI_TANH) v = tanhf(v);
                    if (rbase + row < a.M) dst[(size_t)(rdst + row) * a.dst_ld + a.dst_pad + n] = v;
                }
            }
        }
    }
    S_STAMP_DRAIN(5);
#undef S_LOAD_ADDEND
}


// ---- fused WaveFlow layer (GEMM_EPI_GATE_RS), small-problem shape --------------------------------------------------
// Block 128 rows (the <= 64 channels' first- and second-half rows: the bm = 128 packing as it is) x 128 columns, waves
// 1 x 4, wave tile 128 x 32 (4 x 1 MFMA tiles): half the columns per wave of gemm_f32.hip's 128 x 256 block, so a
// launch has twice as many waves of half the length.  WaveFlow's row recurrence makes a call ~960 DEPENDENT launches; at
// batch 1 a launch is 57 large blocks and its duration is one wave's serial time (113 us; B = 1 and B = 2 cost the same
// 109 ms per call, profiles/r3_11).  Same packed operands, same chunk / k-step order per element: bit-identical.
constexpr int R_BN = 128;
constexpr int R_STAGE = GEMM_KC * (S_BM + R_BN);         // 4096 floats = 16 KiB
constexpr int R_SEGTAB = S_NST * R_STAGE;
constexpr int R_CHTAB = R_SEGTAB + GEMM_MAX_SEG * 4;
constexpr int R_LDS_FLOATS = R_CHTAB + 2 * S_MAX_CHUNKS;
constexpr int R_AUX_SC1 = 16;                            // cache-policy immediate of the DMA builtin: sc1 (agent scope)
static_assert(64 * 128 + 128 <= S_NST * R_STAGE, "the res/skip weights are staged over the main loop's stages");

// One 128 x 128 tile (column tile `tile` of batch item `b`).  FRESH = false: the body of the per-layer launch.  FRESH = true: the
// tile is an item of wf_row_persistent_kernel below, where OTHER workgroups of the SAME launch - possibly on another XCD, whose
// L2 is not coherent with this one - produced part of what it reads and will read what it writes:
//   * segments marked GemmSeg.fresh, the residual source and the skip accumulator are read at agent scope (sc1: served by the
//     memory side, never by a line this XCD's L2 or this CU's L1 kept from an earlier layer),
//   * the results are stored at agent scope (write-through); the caller's release fence + flag make them visible.
// Everything else (weights, the ring slots of earlier rows, the conditioning) was written by earlier launches and is read as before.
// (ARGS: `const GemmArgs` in the kernel-argument segment, or the same struct in the constant address space when the descriptor is
// read from memory - then every field is a scalar load; pointers that come out of memory are generic to the compiler, so each
// one that is dereferenced is cast to the global address space: a flat load would also count on lgkmcnt and break the
// counted LDS waits of the main loop)
typedef __attribute__((address_space(1))) float* r_gptr;
typedef const __attribute__((address_space(1))) float* r_cgptr;
// uniform base + 32-bit BYTE offset per lane: global_load / global_store v, v, s[a:b] (an index that is scaled by 4 after the
// zero-extension is 64-bit math per lane again)
typedef const __attribute__((address_space(1))) char* r_cbptr;
typedef __attribute__((address_space(1))) char* r_bptr;
__device__ __forceinline__ float r_load_u(r_cgptr base, unsigned byte_off) { return *(r_cgptr)((r_cbptr)base + byte_off); }
__device__ __forceinline__ void r_store_u(r_gptr base, unsigned byte_off, float v) { *(r_gptr)((r_bptr)base + byte_off) = v; }
__device__ __forceinline__ float r_load_old(r_cgptr p, bool fresh) {
    return fresh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <int SEGS, bool FRESH, class ARGS>
__device__ __forceinline__ void gate_rs_small_tile(ARGS& a, const int tile, const int b) {
    __shared__ __attribute__((aligned(16))) float lds[R_LDS_FLOATS];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = tile * R_BN;

    // Everything the epilogue needs from memory is requested at kernel ENTRY (a batch-1 launch is one wave per SIMD: each
    // dependent round trip at the end is exposed in full): the destination's old values (x += res, skip += ...), the
    // res/skip weights on their way to LDS, both bias vectors.  Nothing in this launch writes this block's tile before its
    // own epilogue.  The launch shape is only taken below two blocks per CU, so the registers are there.
    typedef float r_f32x4 __attribute__((ext_vector_type(4)));
    float old[4][16];
    r_f32x4 rsw[8];
    {
        const int n_ = n0 + wn * 32 + l31;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int rbase = mt * 32;
            const bool second = rbase >= a.split;
            const r_cgptr dstc = (r_cgptr)(second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride);
            const r_cgptr src = second ? dstc : (a.src0 ? (r_cgptr)(a.src0 + (size_t)b * a.src0_bstride) : dstc);
            const int accum = second ? a.acc1 : a.acc0;
            const int rdst = second ? rbase - a.split : rbase;
            if (accum && rbase < a.rs_rows) {       // one unconditional form; rows / columns past the end re-read the last one
                const r_cgptr sp = src + (size_t)rdst * a.dst_ld + a.dst_pad + min(n_, a.L - 1);
                const int rlast = a.rs_rows - 1 - rbase;
#pragma unroll
                for (int r = 0; r < 16; ++r) old[mt][r] = r_load_old(sp + (size_t)min((r & 3) + 8 * (r >> 2) + 4 * lhi, rlast) * a.dst_ld, FRESH);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) old[mt][r] = 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) rsw[k] = *reinterpret_cast<const __attribute__((address_space(1))) r_f32x4*>((r_cgptr)a.rs_wT + t * 4 + k * 1024);
    }
    const float bias_pre = t < S_BM ? ((r_cgptr)a.bias)[t] : 0.0f;
    const float rsb_pre = t < 128 ? ((r_cgptr)a.rs_bias)[t] : 0.0f;

    const int nalloc = a.a_nch_alloc ? a.a_nch_alloc : a.nch_total;
    // A: the [16][128] chunk of the bm = 128 packing, copied linearly: piece p of this wave = floats [wave * 256 + p * 1024, +256)
    const gbyte_ptr a_base = (gbyte_ptr)(a.A + (size_t)a.a_ch_off * S_ASTAGE + wave * 256);
    const unsigned dma_a_lane = (unsigned)(lane * 16), dma_a_lane1 = dma_a_lane + 4096;   // piece 0 / 1
    (void)nalloc;
    // B: [16][128] stage, piece p of this wave = k-row 2 wave + 8 p + (lane >> 5), columns 4 (lane & 31)
    const unsigned dma_b_lane = (unsigned)(((size_t)(2 * wave + (lane >> 5)) * a.ld + (lane & 31) * 4) * 4);
    const unsigned dma_b_lane1 = dma_b_lane + (unsigned)(8 * a.ld * 4);   // piece 1: eight k-rows on

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

#define R_ISSUE_A(buf, c, p)                                                                                     \
    CTTS_GLDS_U(a_base + (size_t)(c) * (S_ASTAGE * 4), (p) ? dma_a_lane1 : dma_a_lane,                           \
                (lds_fptr)(lds + (buf) * R_STAGE + wave * 256 + (p) * 1024), 0)
    // a table entry = address of the chunk's B rows | (FRESH: bit 0 = the segment is marked fresh -> sc1 DMA); addresses are
    // 4-byte aligned, so the bit is free
#define R_ISSUE_B_AT(buf, ub, p)                                                                                 \
    do {                                                                                                         \
        const unsigned long long ub_ = (ub);                                                                     \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)ub_);                                 \
        const unsigned long long us_ =                                                                           \
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub_ >> 32)) << 32) |             \
            (FRESH ? (lo_ & ~1u) : lo_);                                                                         \
        if (FRESH && (lo_ & 1u))                                                                                 \
            CTTS_GLDS_U(reinterpret_cast<gbyte_ptr>(us_), (p) ? dma_b_lane1 : dma_b_lane,                                \
                        (lds_fptr)(lds + (buf) * R_STAGE + S_ASTAGE + wave * 256 + (p) * 1024), R_AUX_SC1);      \
        else                                                                                                     \
            CTTS_GLDS_U(reinterpret_cast<gbyte_ptr>(us_), (p) ? dma_b_lane1 : dma_b_lane,                                \
                        (lds_fptr)(lds + (buf) * R_STAGE + S_ASTAGE + wave * 256 + (p) * 1024), 0);              \
    } while (0)
    // chunks 0 and 1 before the tables are built (as in conv_gemm_f32_small_kernel)
    const int nch = a.nch_total;
    {
        const auto& g0 = a.seg[0];
        const auto& g1 = a.seg[1];
        const float* sb0 = g0.base + (size_t)b * g0.bstride + (a.pad + n0 + g0.shift);
        const float* sb1 = g1.base + (size_t)b * g1.bstride + (a.pad + n0 + g1.shift);
        const float* c1p = nch <= 1 ? sb0 : a.interleave > 1 ? sb1 : g0.nch > 1 ? sb0 + (size_t)GEMM_KC * a.ld : sb1;
        const int c1 = nch > 1 ? 1 : 0;
        const bool c1_is_seg1 = nch > 1 && (a.interleave > 1 || g0.nch <= 1);
        const unsigned long long u0 = reinterpret_cast<unsigned long long>(sb0) | (FRESH && g0.fresh ? 1u : 0u),
                                 u1 = reinterpret_cast<unsigned long long>(c1p) | (FRESH && (c1_is_seg1 ? g1.fresh : g0.fresh) ? 1u : 0u);
        R_ISSUE_A(0, 0, 0); R_ISSUE_A(0, 0, 1); R_ISSUE_B_AT(0, u0, 0); R_ISSUE_B_AT(0, u0, 1);
        R_ISSUE_A(1, c1, 0); R_ISSUE_A(1, c1, 1); R_ISSUE_B_AT(1, u1, 0); R_ISSUE_B_AT(1, u1, 1);
    }

#pragma unroll
    for (int sidx = 0; sidx < GEMM_MAX_SEG; ++sidx) {
        if (sidx < SEGS && t == sidx) {
            const auto& g = a.seg[sidx];
            unsigned int* e = reinterpret_cast<unsigned int*>(lds + R_SEGTAB + sidx * 4);
            if (sidx < a.nseg) {
                const float* base = g.base + (size_t)b * g.bstride + (a.pad + n0 + g.shift);
                const unsigned long long u = reinterpret_cast<unsigned long long>(base);
                e[0] = (unsigned int)u; e[1] = (unsigned int)(u >> 32); e[2] = (unsigned int)g.nch;
                e[3] = FRESH && g.fresh ? 1u : 0u;
            } else {
                e[0] = 0; e[1] = 0; e[2] = 0x7fffffffu; e[3] = 0;
            }
        }
    }
    __syncthreads();
    {
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(lds + R_CHTAB);
        const int ilv0 = a.interleave > 1 ? a.interleave : 0;
        const int n_il = ilv0 * a.seg[0].nch;
        for (int c0 = t; c0 < a.nch_total; c0 += 256) {
            int c = c0, sg, loc;
            if (c < n_il) {
                sg = c % ilv0;
                loc = c / ilv0;
            } else {
                c -= n_il;
                sg = ilv0;
                for (int k = 0; k < SEGS - 1; ++k) {
                    const int nck = (int)reinterpret_cast<const unsigned int*>(lds + R_SEGTAB + k * 4)[2];
                    if (sg == k && k < a.nseg - 1 && c >= nck) { c -= nck; sg = k + 1; }
                }
                loc = c;
            }
            const unsigned int* e = reinterpret_cast<const unsigned int*>(lds + R_SEGTAB + sg * 4);
            const unsigned long long base = ((unsigned long long)e[1] << 32) | e[0];
            tab[c0] = (base + (unsigned long long)loc * GEMM_KC * a.ld * sizeof(float)) | e[3];
        }
    }
    __syncthreads();

    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // Same hand-scheduled loop as conv_gemm_f32_small_kernel (counted LDS waits, MFMA stream through the chunk boundary);
    // here a k-step is three LDS instructions (row tiles 0|2 and 1|3 by ds_read2st64_b32, one B value) and four MFMAs, and
    // a launch at batch 1 has ONE wave per SIMD, so nothing else hides an exposed LDS round trip.
    typedef float r_f32x2 __attribute__((ext_vector_type(2)));
    r_f32x2 a02[2], a13[2];
    float bq[2];
    const unsigned lds0 = (unsigned)(size_t)(lds_fptr)lds;
    const unsigned a_lane = lds0 + (unsigned)((lhi * S_BM + l31) * 4);
    const unsigned b_lane = lds0 + (unsigned)((S_ASTAGE + lhi * R_BN + wn * 32 + l31) * 4);
#define R_READ_AT(ks, aaddr, baddr)                                                                              \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a02[(ks) & 1]) : "v"(aaddr), "n"(4 * (ks)), "n"(4 * (ks) + 1)); \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a13[(ks) & 1]) : "v"((aaddr) + 128u), "n"(4 * (ks)), "n"(4 * (ks) + 1)); \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bq[(ks) & 1]) : "v"(baddr), "n"(2 * (ks) * R_BN * 4));
#define R_WAIT(n, ks) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a02[(ks) & 1]), "+v"(a13[(ks) & 1]), "+v"(bq[(ks) & 1]));
#define R_MFMA(ks)                                                                                               \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a02[(ks) & 1][0], bq[(ks) & 1], acc[0], 0, 0, 0);               \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a13[(ks) & 1][0], bq[(ks) & 1], acc[1], 0, 0, 0);               \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a02[(ks) & 1][1], bq[(ks) & 1], acc[2], 0, 0, 0);               \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a13[(ks) & 1][1], bq[(ks) & 1], acc[3], 0, 0, 0);
    R_READ_AT(0, a_lane, b_lane)
    int cur = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        const unsigned aa = a_lane + (unsigned)(cur * R_STAGE * 4), ba = b_lane + (unsigned)(cur * R_STAGE * 4);
        const unsigned an = a_lane + (unsigned)(nxt * R_STAGE * 4), bn = b_lane + (unsigned)(nxt * R_STAGE * 4);
        const int nb = cur >= 1 ? cur - 1 : 2;
        const int cn = ch + 2 < nch ? ch + 2 : nch - 1;
        unsigned long long ub_next;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ub_next) : "v"((unsigned)(lds0 + cn * 8)), "n"(R_CHTAB * 4));
        // outstanding: k-step 0 (3 instructions) and the table entry (1)
        R_READ_AT(1, aa, ba) R_WAIT(4, 0) R_MFMA(0) R_ISSUE_A(nb, cn, 0);
        __builtin_amdgcn_sched_barrier(0);
        R_READ_AT(2, aa, ba) R_WAIT(3, 1)                   // (covers the older table entry)
        asm volatile("" : "+v"(ub_next));
        R_MFMA(1) R_ISSUE_A(nb, cn, 1);
        __builtin_amdgcn_sched_barrier(0);
        R_READ_AT(3, aa, ba) R_WAIT(3, 2) R_MFMA(2) R_ISSUE_B_AT(nb, ub_next, 0);
        __builtin_amdgcn_sched_barrier(0);
        R_READ_AT(4, aa, ba) R_WAIT(3, 3) R_MFMA(3) R_ISSUE_B_AT(nb, ub_next, 1);
        __builtin_amdgcn_sched_barrier(0);
        R_READ_AT(5, aa, ba) R_WAIT(3, 4) R_MFMA(4)
        __builtin_amdgcn_sched_barrier(0);
        R_READ_AT(6, aa, ba) R_WAIT(3, 5) R_MFMA(5)
        __builtin_amdgcn_sched_barrier(0);
        R_READ_AT(7, aa, ba) R_WAIT(3, 6) R_MFMA(6)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // chunk ch + 1 landed, the newest still in flight
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        R_READ_AT(0, an, bn)                                // (past the last chunk: a re-issued copy, never used)
        R_WAIT(3, 7) R_MFMA(7)
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a02[0]), "+v"(a13[0]), "+v"(bq[0]));
#undef R_READ_AT
#undef R_WAIT
#undef R_MFMA
#undef R_ISSUE_B_AT
#undef R_ISSUE_A
#undef R_ISSUE_B
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: gate in registers, res/skip 1x1 GEMM on the gated tile, read-modify-write (gemm_f32.hip GATE_RS)
    if (t < S_BM) lds[t] = bias_pre;
    __syncthreads();
    const int n = n0 + wn * 32 + l31;
    float actv[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const bool ok = mt * 32 + row < a.pairC;
            float u0 = acc[mt][r] + lds[mt * 32 + row];
            float u1 = acc[mt + 2][r] + lds[64 + mt * 32 + row];
            if (a.addend) {                                // uniform; columns >= L of a padded row are readable
                const r_cgptr ad = (r_cgptr)a.addend + (size_t)b * a.addend_bstride + a.addend_pad + n;
                const int c = min(mt * 32 + row, a.pairC - 1);
                u0 += ad[(size_t)c * a.addend_ld];
                u1 += ad[(size_t)(a.pairC + c) * a.addend_ld];
            }
            actv[mt][r] = ok ? s_fast_tanh(u0) * s_fast_sigmoid(u1) : 0.0f;
        }
    __syncthreads();                                       // everyone is done with the bias copy in LDS
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<r_f32x4*>(lds + t * 4 + k * 1024) = rsw[k];
    if (t < 128) lds[64 * 128 + t] = rsb_pre;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int r = s & 15;
        const int ch = (s >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float a2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a2[mt] = lds[ch * 128 + mt * 32 + l31];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[mt], actv[s >> 4][r], acc[mt], 0, 0, 0);
    }
    const float* rbias = lds + 64 * 128;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int rbase = mt * 32;
        if (rbase >= a.rs_rows) continue;
        const bool second = rbase >= a.split;
        const r_gptr dst = (r_gptr)(second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride);
        const int rdst = second ? rbase - a.split : rbase;
        if (n < a.L) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const float v = acc[mt][r] + rbias[rbase + row] + old[mt][r];
                if (rbase + row < a.rs_rows) {
                    const r_gptr dp = dst + (size_t)(rdst + row) * a.dst_ld + a.dst_pad + n;
                    if constexpr (FRESH) __hip_atomic_store(dp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *dp = v;
                }
            }
        }
    }
}

template <int SEGS>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_gate_rs_small_kernel(const GemmArgs a, const int ntiles_s) {
    gate_rs_small_tile<SEGS, false, const GemmArgs>(a, blockIdx.x % ntiles_s, blockIdx.x / ntiles_s);
}

// ---- fused WaveFlow layer (GEMM_EPI_GATE_RS), split-K shape for batch 1-2 -----------------------------------------------
// At batch 1 the 128 x 128 shape above is 113 workgroups with one wave per SIMD: fewer than half of the CUs work, and a
// launch lasts one wave's serial chain of 36 chunks x 32 MFMAs.  A wave must own all <= 64 channels (128 rows) of its 32
// columns for the gate -> res/skip GEMM fusion, so the rows cannot be split; the K axis can.  Here a workgroup is 128 rows
// x 64 columns, waves (wn = column tile, kh = K half): chunks are staged in PAIRS, the kh = 0 waves take the even chunk of
// a pair and the kh = 1 waves the odd one.  226 workgroups at batch 1 (88 % of the CUs), each wave half the chain.  After
// the loop the kh = 1 accumulators meet the kh = 0 ones through LDS, the kh = 0 waves gate, the gated tile goes through
// LDS to both halves, and the second GEMM is split by rows: kh = 0 the res rows, kh = 1 the skip rows.
// The sum over K is (even chunks) + (odd chunks): NOT the chunk order of the other shapes - equal to them within fp32
// summation noise, not bit for bit (tests/test_small_shape.py).
constexpr int K_BN = 64;
constexpr int K_CHUNK = GEMM_KC * (S_BM + K_BN);         // 3072 floats: A [16][128] | B [16][64]
constexpr int K_STAGE = 2 * K_CHUNK;                     // a pair of chunks
constexpr int K_NST = 3;
constexpr int K_SEGTAB = K_NST * K_STAGE;
constexpr int K_CHTAB = K_SEGTAB + GEMM_MAX_SEG * 4;
constexpr int K_LDS_FLOATS = K_CHTAB + 2 * S_MAX_CHUNKS; // 77 KB: dynamic LDS, two workgroups per CU
constexpr int K_RED = 0;                                 // epilogue: [2 wn][64][64] partial accumulators, then the res/skip weights
constexpr int K_ACT = 8192;                              //           [2 wn][32][64] gated tile
constexpr int K_BIAS = 12288;                            //           128 in-layer + 128 res/skip biases
static_assert(K_BIAS + 256 <= K_NST * K_STAGE, "the epilogue lives in the stage area");

// (FRESH / ARGS: as gate_rs_small_tile - the tile is an item of the row queue and reads / writes what other workgroups of the same
// launch write / read)
// WAIT (row queue only): called by every thread once the tile's data-INDEPENDENT prologue is under way - epilogue weights
// requested, segment and chunk tables built, the weight (A) pieces of the first two chunk pairs in flight - and returns when
// the item's dependencies are met (false: the launch is being aborted).  Everything that reads what other workgroups wrote
// (the read-modify-write operands, the B pieces) is requested after it.  At batch 1-2 every item of a stage is held by a
// workgroup that waits for the previous stage: the ~4 us prologue (profiles/r5_21) now runs inside that wait.
struct SplitkNoWait { __device__ __forceinline__ bool operator()() const { return true; } };
template <int SEGS, bool FRESH, class ARGS, class WAIT = SplitkNoWait>
__device__ __forceinline__ void gate_rs_splitk_tile(ARGS& a, const int tile, const int b, WAIT wait = WAIT()) {
    constexpr bool EARLY = !std::is_same<WAIT, SplitkNoWait>::value;      // prologue before the dependency wait
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave & 1, kh = wave >> 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = tile * K_BN;
    const int n = n0 + wn * 32 + l31;
    if constexpr (!FRESH) { S_STAMP(0); S_STAMP_WHERE(); }

    // epilogue operands requested at entry (see conv_gemm_f32_gate_rs_small_kernel): this wave stores row tiles 2 kh, 2 kh + 1
    typedef float k_f32x4 __attribute__((ext_vector_type(4)));
    float old[2][16];
    k_f32x4 rsw[8];
    auto load_old = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rbase = (2 * kh + j) * 32;
            const bool second = rbase >= a.split;
            const r_cgptr dstc = (r_cgptr)(second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride);
            const r_cgptr src = second ? dstc : (a.src0 ? (r_cgptr)(a.src0 + (size_t)b * a.src0_bstride) : dstc);
            const int accum = second ? a.acc1 : a.acc0;
            const int rdst = second ? rbase - a.split : rbase;
            if (accum && rbase < a.rs_rows) {
                const r_cgptr sp = src + (size_t)rdst * a.dst_ld + a.dst_pad + min(n, a.L - 1);
                const int rlast = a.rs_rows - 1 - rbase;
#pragma unroll
                for (int r = 0; r < 16; ++r) old[j][r] = r_load_old(sp + (size_t)min((r & 3) + 8 * (r >> 2) + 4 * lhi, rlast) * a.dst_ld, FRESH);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) old[j][r] = 0.0f;
            }
        }
    };
    if constexpr (!EARLY) load_old();
#pragma unroll
    for (int k = 0; k < 8; ++k) rsw[k] = *reinterpret_cast<const __attribute__((address_space(1))) k_f32x4*>((r_cgptr)a.rs_wT + t * 4 + k * 1024);
    const float bias_pre = t < S_BM ? ((r_cgptr)a.bias)[t] : 0.0f;
    const float rsb_pre = t < 128 ? ((r_cgptr)a.rs_bias)[t] : 0.0f;

    // segment table, then chunk -> B address (as in the other shapes)
#pragma unroll
    for (int sidx = 0; sidx < GEMM_MAX_SEG; ++sidx) {
        if (sidx < SEGS && t == sidx) {
            const auto& g = a.seg[sidx];
            unsigned int* e = reinterpret_cast<unsigned int*>(lds + K_SEGTAB + sidx * 4);
            if (sidx < a.nseg) {
                const float* base = g.base + (size_t)b * g.bstride + (a.pad + n0 + g.shift);
                const unsigned long long u = reinterpret_cast<unsigned long long>(base);
                e[0] = (unsigned int)u; e[1] = (unsigned int)(u >> 32); e[2] = (unsigned int)g.nch;
                e[3] = FRESH && g.fresh ? 1u : 0u;
            } else {
                e[0] = 0; e[1] = 0; e[2] = 0x7fffffffu; e[3] = 0;
            }
        }
    }
    __syncthreads();
    {
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(lds + K_CHTAB);
        const int ilv0 = a.interleave > 1 ? a.interleave : 0;
        const int n_il = ilv0 * a.seg[0].nch;
        for (int c0 = t; c0 < a.nch_total; c0 += 256) {
            int c = c0, sg, loc;
            if (c < n_il) {
                sg = c % ilv0;
                loc = c / ilv0;
            } else {
                c -= n_il;
                sg = ilv0;
                for (int k = 0; k < SEGS - 1; ++k) {
                    const int nck = (int)reinterpret_cast<const unsigned int*>(lds + K_SEGTAB + k * 4)[2];
                    if (sg == k && k < a.nseg - 1 && c >= nck) { c -= nck; sg = k + 1; }
                }
                loc = c;
            }
            const unsigned int* e = reinterpret_cast<const unsigned int*>(lds + K_SEGTAB + sg * 4);
            const unsigned long long base = ((unsigned long long)e[1] << 32) | e[0];
            tab[c0] = (base + (unsigned long long)loc * GEMM_KC * a.ld * sizeof(float)) | e[3];      // bit 0: fresh (sc1 DMA)
        }
    }
    __syncthreads();

    // DMA pieces of a chunk: A [16][128] = 8 pieces of 1 KiB (this wave: pieces wave, wave + 4), B [16][64] = 4 pieces (piece
    // wave = k-rows 4 wave .. +4: k-row 4 wave + (lane >> 4), columns 4 (lane & 15)).  Slot cs = 2 stage + chunk parity.
    const gbyte_ptr a_base = (gbyte_ptr)(a.A + (size_t)a.a_ch_off * S_ASTAGE + wave * 256);
    const unsigned dma_a_lane = (unsigned)(lane * 16), dma_a_lane1 = dma_a_lane + 4096;   // piece 0 / 1
    const unsigned dma_b_lane = (unsigned)(((size_t)(4 * wave + (lane >> 4)) * a.ld + (lane & 15) * 4) * 4);
    const unsigned lds0 = (unsigned)(size_t)(lds_fptr)lds;
#define K_ISSUE_A(cs, c, p)                                                                                      \
    CTTS_GLDS_U(a_base + (size_t)(c) * (S_ASTAGE * 4), (p) ? dma_a_lane1 : dma_a_lane,                           \
                (lds_fptr)(lds + (cs) * K_CHUNK + wave * 256 + (p) * 1024), 0)
#define K_ISSUE_B_AT(cs, ub)                                                                                     \
    do {                                                                                                         \
        const unsigned long long ub_ = (ub);                                                                     \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)ub_);                                 \
        const unsigned long long us_ =                                                                           \
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub_ >> 32)) << 32) |             \
            (FRESH ? (lo_ & ~1u) : lo_);                                                                         \
        if (FRESH && (lo_ & 1u))                                                                                 \
            CTTS_GLDS_U(reinterpret_cast<gbyte_ptr>(us_), dma_b_lane,                                                \
                        (lds_fptr)(lds + (cs) * K_CHUNK + S_ASTAGE + wave * 256), R_AUX_SC1);                    \
        else                                                                                                     \
            CTTS_GLDS_U(reinterpret_cast<gbyte_ptr>(us_), dma_b_lane,                                                \
                        (lds_fptr)(lds + (cs) * K_CHUNK + S_ASTAGE + wave * 256), 0);                            \
    } while (0)
    const unsigned long long* ctab = reinterpret_cast<const unsigned long long*>(lds + K_CHTAB);
    const int nch = a.nch_total;
    const int npairs = (nch + 1) / 2;
    if constexpr (!EARLY) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {                        // pairs 0 and 1 (chunks past the end: a copy of the last one)
            const int ce = min(2 * j, nch - 1), co = min(2 * j + 1, nch - 1);
            K_ISSUE_A(2 * j, ce, 0); K_ISSUE_A(2 * j, ce, 1); K_ISSUE_B_AT(2 * j, ctab[ce]);
            K_ISSUE_A(2 * j + 1, co, 0); K_ISSUE_A(2 * j + 1, co, 1); K_ISSUE_B_AT(2 * j + 1, ctab[co]);
        }
        if constexpr (!FRESH) S_STAMP(1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // pair 0 landed (the newest six pieces = pair 1 in flight)
    } else {
        // weights first: they do not depend on the previous stage ...
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ce = min(2 * j, nch - 1), co = min(2 * j + 1, nch - 1);
            K_ISSUE_A(2 * j, ce, 0); K_ISSUE_A(2 * j, ce, 1);
            K_ISSUE_A(2 * j + 1, co, 0); K_ISSUE_A(2 * j + 1, co, 1);
        }
        // ... then the dependencies, then what reads other workgroups' results.  Order of this thread's memory operations:
        // [8 A pieces] [read-modify-write operands] [B of pair 0: 2 pieces] [B of pair 1: 2 pieces]; "pair 0 landed" =
        // everything but the newest two.  The main loop's own counts (six pieces per pair) hold from its first pair on: what
        // is still in flight then is older than what it issues.
        if (!wait()) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may land in an LDS this workgroup has left
            return;
        }
        load_old();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ce = min(2 * j, nch - 1), co = min(2 * j + 1, nch - 1);
            K_ISSUE_B_AT(2 * j, ctab[ce]);
            K_ISSUE_B_AT(2 * j + 1, ctab[co]);
        }
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!FRESH) S_STAMP(2);

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // hand-scheduled loop over PAIRS (gemm_f32_small.hip above: counted LDS waits, MFMA stream through the pair boundary);
    // this wave reads the chunk of parity kh; six DMA pieces per wave and pair
    typedef float k_f32x2 __attribute__((ext_vector_type(2)));
    k_f32x2 a02[2], a13[2];
    float bq[2];
    const unsigned a_lane = lds0 + (unsigned)((kh * K_CHUNK + lhi * S_BM + l31) * 4);
    const unsigned b_lane = lds0 + (unsigned)((kh * K_CHUNK + S_ASTAGE + lhi * K_BN + wn * 32 + l31) * 4);
#define K_READ_AT(ks, aaddr, baddr)                                                                              \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a02[(ks) & 1]) : "v"(aaddr), "n"(4 * (ks)), "n"(4 * (ks) + 1)); \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a13[(ks) & 1]) : "v"((aaddr) + 128u), "n"(4 * (ks)), "n"(4 * (ks) + 1)); \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bq[(ks) & 1]) : "v"(baddr), "n"(2 * (ks) * K_BN * 4));
#define K_WAIT(n_, ks) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(a02[(ks) & 1]), "+v"(a13[(ks) & 1]), "+v"(bq[(ks) & 1]));
#ifdef CTTS_EXP_NO_LDSREAD    /* scripts/micro/wf_splitk_timeline.hip only: the loop's matrix work on stale registers */
#undef K_READ_AT
#undef K_WAIT
#define K_READ_AT(ks, aaddr, baddr)
#define K_WAIT(n_, ks) asm volatile("" : "+v"(a02[(ks) & 1]), "+v"(a13[(ks) & 1]), "+v"(bq[(ks) & 1]));
#endif
#ifdef CTTS_EXP_NO_MFMA       /* scripts/micro/wf_splitk_timeline.hip only: the loop without its matrix work */
#define K_MFMA_ON false
#else
#define K_MFMA_ON true
#endif
#define K_MFMA(ks)                                                                                               \
    if (K_MFMA_ON && active) {                                                                                                \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a02[(ks) & 1][0], bq[(ks) & 1], acc[0], 0, 0, 0);           \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a13[(ks) & 1][0], bq[(ks) & 1], acc[1], 0, 0, 0);           \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a02[(ks) & 1][1], bq[(ks) & 1], acc[2], 0, 0, 0);           \
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a13[(ks) & 1][1], bq[(ks) & 1], acc[3], 0, 0, 0);           \
    }
    K_READ_AT(0, a_lane, b_lane)
    int cur = 0;
    for (int i = 0; i < npairs; ++i) {
        const bool active = 2 * i + kh < nch;               // an odd chunk count leaves the last pair's odd half empty
        const int nxt = cur == 2 ? 0 : cur + 1;
        const unsigned aa = a_lane + (unsigned)(cur * K_STAGE * 4), ba = b_lane + (unsigned)(cur * K_STAGE * 4);
        const unsigned an = a_lane + (unsigned)(nxt * K_STAGE * 4), bn = b_lane + (unsigned)(nxt * K_STAGE * 4);
        const int nb = cur >= 1 ? cur - 1 : 2;              // (cur + 2) % 3: the stage pair i - 1 occupied
        const int ne = min(2 * (i + 2), nch - 1), no = min(2 * (i + 2) + 1, nch - 1);
        unsigned long long ub_e, ub_o;
        asm volatile("ds_read_b64 %0, %1" : "=v"(ub_e) : "v"((unsigned)(lds0 + K_CHTAB * 4 + ne * 8)));
        asm volatile("ds_read_b64 %0, %1" : "=v"(ub_o) : "v"((unsigned)(lds0 + K_CHTAB * 4 + no * 8)));
        // outstanding: k-step 0 (3 instructions) and the two table entries
#ifdef CTTS_EXP_NO_DMA        /* ... only: the loop without re-filling its stages (stale operands) */
#define K_LOOP_ISSUE_A(cs, c, p)
#define K_LOOP_ISSUE_B(cs, ub)
#else
#define K_LOOP_ISSUE_A(cs, c, p) K_ISSUE_A(cs, c, p)
#define K_LOOP_ISSUE_B(cs, ub) K_ISSUE_B_AT(cs, ub)
#endif
        K_READ_AT(1, aa, ba) K_WAIT(5, 0) K_MFMA(0) K_LOOP_ISSUE_A(2 * nb, ne, 0);
        __builtin_amdgcn_sched_barrier(0);
        K_READ_AT(2, aa, ba) K_WAIT(3, 1)                   // (covers the older table entries)
        asm volatile("" : "+v"(ub_e), "+v"(ub_o));
        K_MFMA(1) K_LOOP_ISSUE_A(2 * nb, ne, 1);
        __builtin_amdgcn_sched_barrier(0);
        K_READ_AT(3, aa, ba) K_WAIT(3, 2) K_MFMA(2) K_LOOP_ISSUE_B(2 * nb, ub_e);
        __builtin_amdgcn_sched_barrier(0);
        K_READ_AT(4, aa, ba) K_WAIT(3, 3) K_MFMA(3) K_LOOP_ISSUE_A(2 * nb + 1, no, 0);
        __builtin_amdgcn_sched_barrier(0);
        K_READ_AT(5, aa, ba) K_WAIT(3, 4) K_MFMA(4) K_LOOP_ISSUE_A(2 * nb + 1, no, 1);
        __builtin_amdgcn_sched_barrier(0);
        K_READ_AT(6, aa, ba) K_WAIT(3, 5) K_MFMA(5) K_LOOP_ISSUE_B(2 * nb + 1, ub_o);
        __builtin_amdgcn_sched_barrier(0);
        K_READ_AT(7, aa, ba) K_WAIT(3, 6) K_MFMA(6)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // pair i + 1 landed, the newest still in flight
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        K_READ_AT(0, an, bn)
        K_WAIT(3, 7) K_MFMA(7)
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a02[0]), "+v"(a13[0]), "+v"(bq[0]));
#undef K_READ_AT
#undef K_WAIT
#undef K_MFMA
#undef K_ISSUE_A
#undef K_ISSUE_B_AT
#undef K_LOOP_ISSUE_A
#undef K_LOOP_ISSUE_B
#undef K_MFMA_ON
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (!FRESH) S_STAMP(3);

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (kh == 1) {
        float* red = lds + K_RED + wn * 4096 + lane;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(mt * 16 + r) * 64] = acc[mt][r];
    }
    if (t < S_BM) { lds[K_BIAS + t] = bias_pre; lds[K_BIAS + 128 + t] = rsb_pre; }
    __syncthreads();
    float actv[2][16];
    if (kh == 0) {
        const float* red = lds + K_RED + wn * 4096 + lane;
        float* act = lds + K_ACT + wn * 2048 + lane;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const bool ok = mt * 32 + row < a.pairC;
                float u0 = (acc[mt][r] + red[(mt * 16 + r) * 64]) + lds[K_BIAS + mt * 32 + row];
                float u1 = (acc[mt + 2][r] + red[((mt + 2) * 16 + r) * 64]) + lds[K_BIAS + 64 + mt * 32 + row];
                if (a.addend) {                            // uniform; columns >= L of a padded row are readable
                    const r_cgptr ad = (r_cgptr)a.addend + (size_t)b * a.addend_bstride + a.addend_pad + n;
                    const int c = min(mt * 32 + row, a.pairC - 1);
                    u0 += ad[(size_t)c * a.addend_ld];
                    u1 += ad[(size_t)(a.pairC + c) * a.addend_ld];
                }
                actv[mt][r] = ok ? s_fast_tanh(u0) * s_fast_sigmoid(u1) : 0.0f;
                act[(mt * 16 + r) * 64] = actv[mt][r];
            }
    }
    __syncthreads();                                       // partial accumulators consumed, gated tile published
    if constexpr (!FRESH) S_STAMP(4);
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<k_f32x4*>(lds + K_RED + t * 4 + k * 1024) = rsw[k];
    if (kh == 1) {
        const float* act = lds + K_ACT + wn * 2048 + lane;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) actv[mt][r] = act[(mt * 16 + r) * 64];
    }
    __syncthreads();
    // res/skip GEMM on the gated tile, rows split between the K halves: row tiles 2 kh and 2 kh + 1
    f32x16 acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
        const int r = s2 & 15;
        const int ch = (s2 >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float w2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) w2[j] = lds[K_RED + ch * 128 + (2 * kh + j) * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[j], actv[s2 >> 4][r], acc2[j], 0, 0, 0);
    }
    if constexpr (!FRESH) S_STAMP(5);
    const float* rbias = lds + K_BIAS + 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rbase = (2 * kh + j) * 32;
        if (rbase >= a.rs_rows) continue;
        const bool second = rbase >= a.split;
        const r_gptr dst = (r_gptr)(second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride);
        const int rdst = second ? rbase - a.split : rbase;
        if (n < a.L) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const float v = acc2[j][r] + rbias[rbase + row] + old[j][r];
                if (rbase + row < a.rs_rows) {
                    const r_gptr dp = dst + (size_t)(rdst + row) * a.dst_ld + a.dst_pad + n;
                    if constexpr (FRESH) __hip_atomic_store(dp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *dp = v;
                }
            }
        }
    }
    if constexpr (!FRESH) S_STAMP_DRAIN(7);
}

template <int SEGS>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_gate_rs_splitk_kernel(const GemmArgs a, const int ntiles_s) {
    gate_rs_splitk_tile<SEGS, false, const GemmArgs>(a, blockIdx.x % ntiles_s, blockIdx.x / ntiles_s);
}

// ---- the split-K tile on EIGHT waves (round 5; per-layer launches only, i.e. batch 1) ------------------------------------
// At batch 1 the tile above runs ONE wave per SIMD on 226 CUs: nobody covers a wave's LDS reads, DMA issue and barrier waits, and
// the main loop takes 27.5 us for 20 us of matrix work (profiles/r5_21).  Same 128 x 64 tile, same stages, same K order, but
// waves (wn, mh, kh): wave mh of a (column tile, K half) owns the accumulator tiles mh and mh + 2 - the tanh rows and the sigmoid
// rows of channels 32 mh .. 32 mh + 31 - so a SIMD holds two waves (s and s + 4: the two K halves of one channel half) whose
// stalls overlap with each other's MFMAs.  Per pair and wave: one A piece of the even chunk, one of the odd chunk, one B piece
// (waves 0-3: the even chunk's, waves 4-7: the odd chunk's).  Epilogue: the kh = 1 partial sums meet the kh = 0 ones through LDS
// as before, four waves gate (32 channels x 32 columns each), every wave takes ONE 32-row tile of the res/skip GEMM over all 64
// channels.  Every output element is the same expression in the same order as in the four-wave tile: bit-identical
// (tests/test_small_shape.py).
// scripts/micro/wf_splitk_timeline.hip -DCTTS_PROLOGUE_STAMPS: the stamp slots 1-5 resolve the PROLOGUE instead of the later phases
#ifdef CTTS_PROLOGUE_STAMPS
#define K8_P_STAMP(k) S_STAMP(k)
#define K8_L_STAMP(k)
#else
#define K8_P_STAMP(k)
#define K8_L_STAMP(k) S_STAMP(k)
#endif
template <int SEGS>
__device__ __forceinline__ void gate_rs_splitk8_tile(const GemmArgs& a, const int tile, const int b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // waves w and w + 4 share a SIMD: they are the two K HALVES of one (column tile, channel half), so that in the epilogue the four
    // gating waves (kh = 0) sit on four different SIMDs (with the channel halves on a SIMD instead, two SIMDs gated and two idled)
    const int wn = wave & 1, mh = (wave >> 1) & 1, kh = wave >> 2;
    const int bp = wave >> 2;                                // B pieces: waves 0-3 request the even chunk of a pair, 4-7 the odd one
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n0 = tile * K_BN;
    const int n = n0 + wn * 32 + l31;
    S_STAMP(0);
    S_STAMP_WHERE();

    // ---- prologue (profiles/r5_32: 4.5 us of a 37 us layer in the order of the four-wave tile - epilogue operands, segment table,
    // barrier, chunk table, barrier, requests).  Here: (1) every thread below nch_total computes ITS chunk -> B address entry
    // straight from the argument struct - the segment it falls in from the (uniform) chunk counts, that segment's base / batch
    // stride / shift with one lane-indexed load from the kernarg segment - no segment table, one barrier; (2) while those loads
    // fly the weight pieces of pairs 0 and 1 are requested (they need nothing but the struct), then the epilogue operands;
    // (3) table entries written, barrier, the B pieces of pairs 0 and 1.
    typedef const __attribute__((address_space(4))) char* k_kargp;
    const k_kargp kp = (k_kargp)__builtin_amdgcn_kernarg_segment_ptr();      // the GemmArgs struct is the kernel's first parameter
    const int nch = a.nch_total;
    const int npairs = (nch + 1) / 2;
    unsigned long long my_entry[(S_MAX_CHUNKS + 511) / 512];
    {
        int nchs[SEGS];
#pragma unroll
        for (int k = 0; k < SEGS; ++k) nchs[k] = k < a.nseg ? a.seg[k].nch : 0x7fffffff;
        const int ilv0 = a.interleave > 1 ? a.interleave : 0;
        const int n_il = ilv0 * a.seg[0].nch;
#pragma unroll
        for (int q = 0; q < (S_MAX_CHUNKS + 511) / 512; ++q) {
            const int c0 = min(t + 512 * q, nch - 1);
            int c = c0, sg, loc;
            if (c < n_il) {
                sg = c % ilv0;
                loc = c / ilv0;
            } else {
                c -= n_il;
                sg = ilv0;
#pragma unroll
                for (int k = 0; k < SEGS - 1; ++k)
                    if (sg == k && k < a.nseg - 1 && c >= nchs[k]) { c -= nchs[k]; sg = k + 1; }
                loc = c;
            }
            const k_kargp sp = kp + (offsetof(GemmArgs, seg) + (size_t)sg * sizeof(GemmSeg));
            const unsigned long long sbase = *reinterpret_cast<const __attribute__((address_space(4))) unsigned long long*>(sp + offsetof(GemmSeg, base));
            const long long sbstr = *reinterpret_cast<const __attribute__((address_space(4))) long long*>(sp + offsetof(GemmSeg, bstride));
            const int sshift = *reinterpret_cast<const __attribute__((address_space(4))) int*>(sp + offsetof(GemmSeg, shift));
            my_entry[q] = sbase + (unsigned long long)(((long long)b * sbstr + (a.pad + n0 + sshift) + (long long)loc * GEMM_KC * a.ld) * 4);
        }
    }

    // DMA pieces: A [16][128] = 8 pieces of 1 KiB (piece = wave), B [16][64] = 4 pieces (piece bw = wave & 3: k-rows 4 bw .. + 4)
    const int bw = wave & 3;
    const gbyte_ptr a_base = (gbyte_ptr)(a.A + (size_t)a.a_ch_off * S_ASTAGE + wave * 256);
    const unsigned dma_a_lane = (unsigned)(lane * 16);
    const unsigned dma_b_lane = (unsigned)(((size_t)(4 * bw + (lane >> 4)) * a.ld + (lane & 15) * 4) * 4);
    const unsigned lds0 = (unsigned)(size_t)(lds_fptr)lds;
#define K8_ISSUE_A(cs, c) \
    CTTS_GLDS_U(a_base + (size_t)(c) * (S_ASTAGE * 4), dma_a_lane, (lds_fptr)(lds + (cs) * K_CHUNK + wave * 256), 0)
#define K8_ISSUE_B(cs, ub)                                                                                       \
    do {                                                                                                         \
        const unsigned long long ub_ = (ub);                                                                     \
        const unsigned long long us_ =                                                                           \
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub_ >> 32)) << 32) |             \
            (unsigned)__builtin_amdgcn_readfirstlane((int)ub_);                                                  \
        CTTS_GLDS_U(reinterpret_cast<gbyte_ptr>(us_), dma_b_lane,                                                \
                    (lds_fptr)(lds + (cs) * K_CHUNK + S_ASTAGE + bw * 256), 0);                                  \
    } while (0)
#pragma unroll
    for (int j = 0; j < 2; ++j) {                            // pairs 0 and 1 (chunks past the end: a copy of the last one)
        K8_ISSUE_A(2 * j, min(2 * j, nch - 1)); K8_ISSUE_A(2 * j + 1, min(2 * j + 1, nch - 1));
    }
    K8_P_STAMP(1);

    // epilogue operands: this wave stores row tile jt
    typedef float k_f32x4 __attribute__((ext_vector_type(4)));
    const int jt = 2 * kh + mh;
    float old[16];
    k_f32x4 rsw[4];
    {
        const int rbase = jt * 32;
        const bool second = rbase >= a.split;
        const r_cgptr dstc = (r_cgptr)(second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride);
        const r_cgptr src = second ? dstc : (a.src0 ? (r_cgptr)(a.src0 + (size_t)b * a.src0_bstride) : dstc);
        const int accum = second ? a.acc1 : a.acc0;
        const int rdst = second ? rbase - a.split : rbase;
        if (accum && rbase < a.rs_rows) {
            // a wave-uniform base + 32-bit lane offsets (global_load_dword v, v, s[a:b]): the 64-bit per-lane address math of the
            // plain form was most of the 0.9 us this block took (profiles/r5_33)
            const r_cgptr sp = src + (size_t)rdst * a.dst_ld + a.dst_pad;
            const int rlast = a.rs_rows - 1 - rbase;
            const unsigned col = (unsigned)min(n, a.L - 1), dld = (unsigned)a.dst_ld;
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = r_load_u(sp, ((unsigned)min((r & 3) + 8 * (r >> 2) + 4 * lhi, rlast) * dld + col) * 4u);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) rsw[k] = *reinterpret_cast<const __attribute__((address_space(1))) k_f32x4*>((r_cgptr)a.rs_wT + t * 4 + k * 2048);
    const float bias_pre = t < S_BM ? ((r_cgptr)a.bias)[t] : 0.0f;
    const float rsb_pre = t < 128 ? ((r_cgptr)a.rs_bias)[t] : 0.0f;
    K8_P_STAMP(2);

    {
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(lds + K_CHTAB);
#pragma unroll
        for (int q = 0; q < (S_MAX_CHUNKS + 511) / 512; ++q)
            if (t + 512 * q < nch) tab[t + 512 * q] = my_entry[q];
    }
    __syncthreads();
    K8_P_STAMP(3);
    const unsigned long long* ctab = reinterpret_cast<const unsigned long long*>(lds + K_CHTAB);
#pragma unroll
    for (int j = 0; j < 2; ++j) K8_ISSUE_B(2 * j + bp, ctab[min(2 * j + bp, nch - 1)]);
    K8_L_STAMP(1); K8_P_STAMP(4);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");         // pair 0 landed: everything but this wave's B piece of pair 1
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    K8_L_STAMP(2); K8_P_STAMP(5);
#ifdef CTTS_CLOCK_STAMPS        /* scripts/micro/wf_splitk_timeline.hip: slot 6 = shader-clock cycles of the main loop (s_memtime) */
    const unsigned long long clk0_ = __builtin_readcyclecounter();
#endif

    f32x16 acc[2];                                           // accumulator tiles mh, mh + 2
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // hand-scheduled loop over PAIRS: per k-step ONE ds_read2st64_b32 for the two A fragments (tiles mh, mh + 2), per TWO k-steps
    // one for the B values of both (rows 2 ks + lhi and 2 ks + 2 + lhi of the [16][64] stage are 128 floats apart): 12 LDS
    // instructions per chunk and wave; counted lgkmcnt waits (LDS operations complete in order)
    typedef float k_f32x2 __attribute__((ext_vector_type(2)));
    k_f32x2 a2[2], bq2[2];
    const unsigned a_lane = lds0 + (unsigned)((kh * K_CHUNK + lhi * S_BM + mh * 32 + l31) * 4);
    const unsigned b_lane = lds0 + (unsigned)((kh * K_CHUNK + S_ASTAGE + lhi * K_BN + wn * 32 + l31) * 4);
#define K8_RA(ks, aaddr) \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a2[(ks) & 1]) : "v"(aaddr), "n"(4 * (ks)), "n"(4 * (ks) + 1));
#define K8_RB(jj, baddr) \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(bq2[(jj) & 1]) : "v"(baddr), "n"(4 * (jj)), "n"(4 * (jj) + 2));
#define K8_WAIT(n_, ks) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(a2[(ks) & 1]), "+v"(bq2[((ks) >> 1) & 1]));
#ifdef CTTS_EXP_NO_LDSREAD    /* scripts/micro/wf_splitk_timeline.hip only (as in the four-wave tile) */
#undef K8_RA
#undef K8_RB
#undef K8_WAIT
#define K8_RA(ks, aaddr)
#define K8_RB(jj, baddr)
#define K8_WAIT(n_, ks) asm volatile("" : "+v"(a2[(ks) & 1]), "+v"(bq2[((ks) >> 1) & 1]));
#endif
#ifdef CTTS_EXP_NO_MFMA
#define K8_MFMA_ON false
#else
#define K8_MFMA_ON true
#endif
#define K8_MFMA(ks)                                                                                              \
    if (K8_MFMA_ON && active) {                                                                                  \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[(ks) & 1][0], bq2[((ks) >> 1) & 1][(ks) & 1], acc[0], 0, 0, 0); \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[(ks) & 1][1], bq2[((ks) >> 1) & 1][(ks) & 1], acc[1], 0, 0, 0); \
    }
    K8_RA(0, a_lane) K8_RB(0, b_lane)
    int cur = 0;
    for (int i = 0; i < npairs; ++i) {
        const bool active = 2 * i + kh < nch;               // an odd chunk count leaves the last pair's odd half empty
        const int nxt = cur == 2 ? 0 : cur + 1;
        const unsigned aa = a_lane + (unsigned)(cur * K_STAGE * 4), ba = b_lane + (unsigned)(cur * K_STAGE * 4);
        const unsigned an = a_lane + (unsigned)(nxt * K_STAGE * 4), bn = b_lane + (unsigned)(nxt * K_STAGE * 4);
        const int nb = cur >= 1 ? cur - 1 : 2;              // (cur + 2) % 3: the stage pair i - 1 occupied
        const int ne = min(2 * (i + 2), nch - 1), no = min(2 * (i + 2) + 1, nch - 1);
        unsigned long long ub;
        asm volatile("ds_read_b64 %0, %1" : "=v"(ub) : "v"((unsigned)(lds0 + K_CHTAB * 4 + (bp ? no : ne) * 8)));
#ifdef CTTS_EXP_NO_DMA
#define K8_LOOP_ISSUE_A(cs, c)
#define K8_LOOP_ISSUE_B(cs, ub)
#else
#define K8_LOOP_ISSUE_A(cs, c) K8_ISSUE_A(cs, c)
#define K8_LOOP_ISSUE_B(cs, ub) K8_ISSUE_B(cs, ub)
#endif
        // outstanding before every wait, oldest first (RAk: A of k-step k, BQj: B of k-steps 2 j, 2 j + 1, T: the table entry)
        K8_RA(1, aa) K8_WAIT(2, 0) K8_MFMA(0) K8_LOOP_ISSUE_A(2 * nb, ne);                       // RA0 BQ0 | T RA1
        __builtin_amdgcn_sched_barrier(0);
        K8_RA(2, aa) K8_RB(1, ba) K8_WAIT(2, 1)                                                  // T RA1 | RA2 BQ1
        asm volatile("" : "+v"(ub));
        K8_MFMA(1) K8_LOOP_ISSUE_A(2 * nb + 1, no);
        __builtin_amdgcn_sched_barrier(0);
        K8_RA(3, aa) K8_WAIT(1, 2) K8_MFMA(2) K8_LOOP_ISSUE_B(2 * nb + bp, ub);                  // RA2 BQ1 | RA3
        __builtin_amdgcn_sched_barrier(0);
        K8_RA(4, aa) K8_RB(2, ba) K8_WAIT(2, 3) K8_MFMA(3)                                       // RA3 | RA4 BQ2
        __builtin_amdgcn_sched_barrier(0);
        K8_RA(5, aa) K8_WAIT(1, 4) K8_MFMA(4)                                                    // RA4 BQ2 | RA5
        __builtin_amdgcn_sched_barrier(0);
        K8_RA(6, aa) K8_RB(3, ba) K8_WAIT(2, 5) K8_MFMA(5)                                       // RA5 | RA6 BQ3
        __builtin_amdgcn_sched_barrier(0);
        K8_RA(7, aa) K8_WAIT(1, 6) K8_MFMA(6)                                                    // RA6 BQ3 | RA7
        __builtin_amdgcn_sched_barrier(0);
#ifndef CTTS_EXP_NO_BARRIER   /* scripts/micro/wf_splitk_timeline.hip only: the loop without its per-pair synchronisation */
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");    // pair i + 1 landed, the newest still in flight
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        K8_RA(0, an) K8_RB(0, bn)
        K8_WAIT(2, 7) K8_MFMA(7)                                                                 // RA7 | RA0' BQ0'
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a2[0]), "+v"(bq2[0]));
#undef K8_RA
#undef K8_RB
#undef K8_WAIT
#undef K8_MFMA
#undef K8_ISSUE_A
#undef K8_ISSUE_B
#undef K8_LOOP_ISSUE_A
#undef K8_LOOP_ISSUE_B
#undef K8_MFMA_ON
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    K8_L_STAMP(3);
#ifdef CTTS_CLOCK_STAMPS
    if (threadIdx.x == 0) g_small_stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_readcyclecounter() - clk0_;
#endif

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (kh == 1) {
        float* red = lds + K_RED + wn * 4096 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            red[(mh * 16 + r) * 64] = acc[0][r];
            red[((mh + 2) * 16 + r) * 64] = acc[1][r];
        }
    }
    if (t < S_BM) { lds[K_BIAS + t] = bias_pre; lds[K_BIAS + 128 + t] = rsb_pre; }
    __syncthreads();
    if (kh == 0) {
        const float* red = lds + K_RED + wn * 4096 + lane;
        float* act = lds + K_ACT + wn * 2048 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const bool ok = mh * 32 + row < a.pairC;
            float u0 = (acc[0][r] + red[(mh * 16 + r) * 64]) + lds[K_BIAS + mh * 32 + row];
            float u1 = (acc[1][r] + red[((mh + 2) * 16 + r) * 64]) + lds[K_BIAS + 64 + mh * 32 + row];
            if (a.addend) {                                  // uniform; columns >= L of a padded row are readable
                const r_cgptr ad = (r_cgptr)a.addend + (size_t)b * a.addend_bstride + a.addend_pad + n;
                const int c = min(mh * 32 + row, a.pairC - 1);
                u0 += ad[(size_t)c * a.addend_ld];
                u1 += ad[(size_t)(a.pairC + c) * a.addend_ld];
            }
            act[(mh * 16 + r) * 64] = ok ? s_fast_tanh(u0) * s_fast_sigmoid(u1) : 0.0f;
        }
    }
    __syncthreads();                                         // partial accumulators consumed, gated tile published
    K8_L_STAMP(4);
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<k_f32x4*>(lds + K_RED + t * 4 + k * 2048) = rsw[k];
    float actv[2][16];
    {
        const float* act = lds + K_ACT + wn * 2048 + lane;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) actv[mt][r] = act[(mt * 16 + r) * 64];
    }
    __syncthreads();
    // res/skip GEMM on the gated tile: this wave's row tile jt over all 64 channels
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
        const int r = s2 & 15;
        const int ch = (s2 >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[K_RED + ch * 128 + jt * 32 + l31], actv[s2 >> 4][r], acc2, 0, 0, 0);
    }
    K8_L_STAMP(5);
    const float* rbias = lds + K_BIAS + 128;
    {
        const int rbase = jt * 32;
        if (rbase < a.rs_rows && n < a.L) {
            const bool second = rbase >= a.split;
            const int rdst = second ? rbase - a.split : rbase;
            const r_gptr dst = (r_gptr)(second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride) +
                               ((size_t)rdst * a.dst_ld + a.dst_pad);
            const unsigned dld = (unsigned)a.dst_ld;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const float v = acc2[r] + rbias[rbase + row] + old[r];
                if (rbase + row < a.rs_rows) r_store_u(dst, ((unsigned)row * dld + (unsigned)n) * 4u, v);
            }
        }
    }
    S_STAMP_DRAIN(7);
}

template <int SEGS>
__global__ __launch_bounds__(512) void conv_gemm_f32_gate_rs_splitk8_kernel(const GemmArgs a, const int ntiles_s) {
    CTTS_WARM_KERNARGS(GemmArgs);
    gate_rs_splitk8_tile<SEGS>(a, blockIdx.x % ntiles_s, blockIdx.x / ntiles_s);
}

// ---- WaveFlow row step as ONE launch: the fused layers of a row free-run through a work queue ---------------------------
//   A row of the WaveFlow recurrence is n_layers dependent fused layers; launched one by one, every layer
// waits for the slowest workgroup of the one before it, although tile t of layer i + 1 only needs tiles t - 1, t, t + 1 of
// layer i (|column shift| <= 128 = one tile).  Here the row is ONE launch of resident workgroups that take ITEMS (layer, batch
// item, tile) from an atomic counter in layer-major order and run gate_rs_small_tile on each:
//   * an item waits for the <= 3 flags of its neighbours in the previous layer ({epoch} words), computes, stores at agent scope,
//     waits for the stores' acknowledgements, sets its own flag;
//   * items are CLAIMED IN ORDER, one at a time, and every dependency of an item precedes it in that order: the oldest
//     unfinished item is always being worked on with every dependency finished - no co-residency requirement, no deadlock
//     by construction.  The wait is bounded all the same (s_memrealtime): on expiry the abort word is set, every workgroup leaves,
//     and the call fills its output with NaN;
//   * with 2 workgroups per CU and ~900 items per layer the dependencies of a freshly claimed item are one whole layer of items
//     behind the running window: nobody waits, the CUs never drain between layers, and a CU that is faster simply takes more
//     items (the per-layer launch quantises 456 blocks on 512 slots).
// Same tile body, same chunk order: bit-identical to the per-layer launches (tests/test_waveflow.py, test_full_size.py).
// Whole-FLOW form (tails != NULL): the rows of a flow are chained in the same launch.  A row's stages are its n_layers fused
// layers and then a TAIL stage per tile (end conv -> affine update of the next latent row -> the next row's start conv, all
// per column: waveflow_tail.h), which depends on the last layer of its own tile only; the first layer of the next row depends
// on the tail stage of the neighbouring tiles.  No launch boundary is left inside a flow, so EVERYTHING written inside the
// launch is read at agent scope: the host marks every X segment fresh, the tail reads the skip sum at agent scope.
struct WfRowArgs {
    const GemmArgs* layers;            // [nrows][nlayers] in device memory
    const WfTailDesc* tails;           // [nrows], or NULL: one row per launch, no tail stage
    int nrows;
    int nlayers, ntiles_s, batch;
    unsigned int* counter;             // this launch's item counter (zeroed by the host at the start of the call)
    unsigned int* flags;               // [nlayers][batch][ntiles_s], value = epoch of the launch that last finished the item
    unsigned int* abort_word;          // != 0: a bounded wait expired somewhere in this call
    unsigned int epoch;                // > 0, unique per launch within a call
    unsigned int timeout_ticks;        // of s_memrealtime (100 MHz)
    int ntiles_k;                      // 64-column tiles per batch item (the split-K body's items)
    int debug;                         // CTTS_WF_QUEUE_DEBUG (diagnosis only): 1 no dependency waits, 2 no tile body, 8 release fence (L2 write-back), 64 two workgroups per CU at every size
};

// BODY 0: items are 128 x 128 tiles (gate_rs_small_tile, neighbours t - 1 .. t + 1); BODY 1: 128 x 64 tiles of the split-K body
// (gate_rs_splitk_tile: half the serial chain per item, twice the items - the sizes at which a layer has fewer items than
// the chip has workgroup slots; a 128-column shift reaches tiles t - 2 .. t + 2)
// (Two workgroups per CU.  Three - the 128 x 128 body with every epilogue operand loaded late, 168 registers - gave 168.2 ms at
// batch 8 against 166.3, and 136 against 100 at batch 4 where a layer has fewer items than slots: profiles/r4_12.)
// tail stage of one tile of NCOL columns (256 threads: wave wv = channel quarter, lane = column (+ 64))
template <int NCOL, class DESC>
__device__ __forceinline__ void wf_tail_start_tile(DESC& d, const int tile, const int b, float* part /* [3][2][NCOL] + [NCOL] */) {
    constexpr int CPL = NCOL / 64;                       // columns per lane
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n0 = tile * NCOL;
    const int C = d.C, cq = C / 4, cbeg = wv * cq;
    float e0[CPL], e1[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) e0[k] = e1[k] = 0.f;
    const r_cgptr ob = (r_cgptr)d.out + (size_t)b * d.out_bstride + d.pad + n0 + lane;
    const r_cgptr we = (r_cgptr)d.Wend;
    for (int c = cbeg; c < cbeg + cq; ++c) {
        const float w0 = we[c], w1 = we[C + c];
#pragma unroll
        for (int k = 0; k < CPL; ++k) {                  // (columns >= L of a padded row are readable)
            const float v = __hip_atomic_load(ob + (size_t)c * d.ld + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            e0[k] = wf_end_fma(w0, v, e0[k]);
            e1[k] = wf_end_fma(w1, v, e1[k]);
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            part[((wv - 1) * 2 + 0) * NCOL + lane + 64 * k] = e0[k];
            part[((wv - 1) * 2 + 1) * NCOL + lane + 64 * k] = e1[k];
        }
    }
    __syncthreads();
    float* anew = part + 6 * NCOL;
    if (wv == 0) {
        const float b0 = ((r_cgptr)d.bend)[0], b1 = ((r_cgptr)d.bend)[1];
        const r_gptr rp = (r_gptr)d.rows + ((size_t)b * d.G + d.row) * d.Lr;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int col = lane + 64 * k, n = n0 + col;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                e0[k] += part[(q * 2 + 0) * NCOL + col];
                e1[k] += part[(q * 2 + 1) * NCOL + col];
            }
            float a = 0.f;
            if (n < d.L) a = wf_row_update(rp[n], e0[k], e1[k], b0, b1);
            if (n < d.Lr) __hip_atomic_store(rp + n, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (row tail stays zero)
            anew[col] = a;
        }
    }
    __syncthreads();
    if (d.x0) {                                          // the next row's start conv; columns >= L are halo: zero
        const int col = t % NCOL, n = n0 + col;
        const int L4 = (d.L + 3) & ~3;                   // (what wf_start_kernel writes: whole float4s that begin below L)
        if (n < L4) {
            const float a = anew[col];
            const r_gptr xb = (r_gptr)d.x0 + (size_t)b * d.x0_bstride + d.pad + n;
            const r_cgptr ws = (r_cgptr)d.ws, bs = (r_cgptr)d.bs;
            for (int c = t / NCOL; c < C; c += 256 / NCOL)
                __hip_atomic_store(xb + (size_t)c * d.ld, n < d.L ? wf_start_value(ws[c], a, bs[c]) : 0.f, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int SEGS, int BODY>
__global__ __launch_bounds__(256, 2) void wf_row_persistent_kernel(const WfRowArgs w) {
    __shared__ int s_item, s_abort;
    __shared__ float s_tail[7 * 128];
    constexpr int HALO = BODY == 1 ? 2 : 1;
    const int t = threadIdx.x;
    const int ntiles = BODY == 1 ? w.ntiles_k : w.ntiles_s;
    const int spr = w.nlayers + (w.tails ? 1 : 0);       // stages per row
    const int per_layer = ntiles * w.batch, total = w.nrows * spr * per_layer;
    // ONE `t == 0` region per iteration, between two barriers, and every branch that contains a barrier on a readfirstlane'd
    // (provably uniform) value: with the claim at the top of the loop and the flag store at its bottom the compiler threaded
    // the two `t == 0` regions together across the back edge and lane 0 left the loop's barriers to the other 63 lanes of its
    // wave - the launch never ended (first version of this kernel, profiles/HISTORY.md round 4)
    auto claim = [&]() -> int {
        return __hip_atomic_load(w.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0
                   ? total
                   : (int)__hip_atomic_fetch_add(w.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (t == 0) { s_abort = 0; s_item = claim(); }
    __syncthreads();
    for (;;) {
        const int q = __builtin_amdgcn_readfirstlane(s_item);
        if (q >= total) break;
        const int gs = q / per_layer, rem = q - gs * per_layer;              // global stage (row-major), item within it
        const int row = gs / spr, layer = gs - row * spr;                    // layer == nlayers: the row's tail stage
        const int b = rem / ntiles, tile = rem - b * ntiles;
        const int halo = layer == w.nlayers ? 0 : HALO;                      // a tail item needs its own tile only
        // Every thread calls it once per item: the (<= 2 HALO + 1) polling threads spin on their neighbour flags of the previous
        // stage, the barrier closes the wait.  false: the launch is being aborted.
        auto wait_deps = [&]() -> bool {
            if (gs > 0 && t < 2 * HALO + 1 && !(w.debug & 1)) {
                const int tt = tile + t - HALO;
                if (tt >= tile - halo && tt <= tile + halo && tt >= 0 && tt < ntiles) {
                    const unsigned int* f = w.flags + (size_t)(gs - 1) * per_layer + b * ntiles + tt;
                    // The bound is on time WITHOUT PROGRESS, not on wall time: s_memrealtime keeps running while the process is
                    // preempted or shares the GPU, so the first expiry only takes a snapshot of the launch's item counter and opens a
                    // second period (everybody was frozen together; the others need a moment to be seen moving again); the wait
                    // aborts when a whole further period passes in which no workgroup claimed an item.
                    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    unsigned int seen = 0;
                    bool have_seen = false;                                // (the counter is only looked at once a period has expired)
                    for (unsigned spins = 0; __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != w.epoch; ++spins) {
                        __builtin_amdgcn_s_sleep(8);
                        if ((spins & 63u) == 63u) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                            if (now - t0 > w.timeout_ticks) {
                                const unsigned int c = __hip_atomic_load(w.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (have_seen && c == seen) {              // a whole period in which nobody claimed an item
                                    __hip_atomic_store(w.abort_word, 1u + (unsigned)gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    s_abort = 1;
                                    break;
                                }
                                seen = c;
                                have_seen = true;
                                t0 = now;
                            }
                            if (__hip_atomic_load(w.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { s_abort = 1; break; }
                        }
                    }
                }
            }
            __syncthreads();                               // dependencies met
            if (__builtin_amdgcn_readfirstlane(s_abort)) return false;
            // Acquire.  Everything another workgroup wrote inside this launch is read with sc1 loads / sc1 DMA (GemmSeg.fresh, the
            // tail's atomic loads), which is what makes the relaxed flag protocol correct today; the invalidate (buffer_inv sc1)
            // makes it correct for a plain load of such data too, should one ever be added.  CTTS_WF_QUEUE_DEBUG=256
            // leaves it out (A/B: profiles/r5_17_wf_queue_acquire_ab.txt).
            if (!(w.debug & 256)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return true;
        };
        typedef const __attribute__((address_space(4))) GemmArgs const_args;         // scalar loads of the descriptor
        if (BODY == 1 && layer < w.nlayers && !(w.debug & (2 | 512))) {
            // split-K item: the tile's data-independent prologue runs BEFORE the wait (CTTS_WF_QUEUE_DEBUG=512: after it, as the
            // 128 x 128 items and the tail stage do)
            const_args& a = *((const_args*)w.layers + (size_t)row * w.nlayers + layer);
            if constexpr (BODY == 1) gate_rs_splitk_tile<SEGS, true, const_args>(a, tile, b, wait_deps);
            if (__builtin_amdgcn_readfirstlane(s_abort)) break;
        } else {
            if (!wait_deps()) break;
            if (layer == w.nlayers) {
                typedef const __attribute__((address_space(4))) WfTailDesc const_tail;
                wf_tail_start_tile<BODY == 1 ? 64 : 128, const_tail>(*((const_tail*)w.tails + row), tile, b, s_tail);
            } else if (!(w.debug & 2)) {
                const_args& a = *((const_args*)w.layers + (size_t)row * w.nlayers + layer);
                if constexpr (BODY == 1) gate_rs_splitk_tile<SEGS, true, const_args>(a, tile, b);
                else gate_rs_small_tile<SEGS, true, const_args>(a, tile, b);
            }
        }
        // Release.  Every result was stored at agent scope (sc1: written THROUGH this XCD's L2), so a store is visible to the
        // agent once it is acknowledged: vmcnt(0) of every thread, then the barrier, then the flag.  The formal release fence
        // adds `buffer_wbl2 sc1`, a walk of the whole L2 for dirty lines that are not there: measured 226.9 ms per call with it
        // against 166.0 without (config 4, B = 8; CTTS_WF_QUEUE_DEBUG=8 brings it back for A/B).
        if (w.debug & 8) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // ... for every thread of the item; the LDS is free; s_item has been read
        if (t == 0) {
            __hip_atomic_store(w.flags + (size_t)gs * per_layer + rem, w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (claimed only now: a workgroup that claimed its next item early - to hide the atomic's round trip - kept it from
            //  the workgroups that were idle in the meantime: 106 -> 117 ms at batch 5, 38.5 -> 47 at batch 1)
            s_item = claim();
        }
        __syncthreads();
    }
}

template <int EPI, int XS>
void launch_small_xs(dim3 grid, hipStream_t stream, const GemmArgs& a, int ntiles_s) {
    if (a.nseg <= 4) hipLaunchKernelGGL((conv_gemm_f32_small_kernel<EPI, 4, XS>), grid, dim3(256), 0, stream, a, ntiles_s);
    else hipLaunchKernelGGL((conv_gemm_f32_small_kernel<EPI, GEMM_MAX_SEG, XS>), grid, dim3(256), 0, stream, a, ntiles_s);
}

template <int EPI>
void launch_small(dim3 grid, hipStream_t stream, const GemmArgs& a, int ntiles_s) {
    switch (gemm_split_level(a.gemm_mode)) {
        case 6: launch_small_xs<EPI, 6>(grid, stream, a, ntiles_s); break;
        case 3: launch_small_xs<EPI, 3>(grid, stream, a, ntiles_s); break;
        default: launch_small_xs<EPI, 0>(grid, stream, a, ntiles_s); break;
    }
}

}  // namespace

static inline long long gate_rs_blocks(const GemmArgs& a) { return a.shape_blocks > 0 ? a.shape_blocks : (long long)a.ntiles * a.batch; }

bool gemm_f32_small_applies(int epi, const GemmArgs& a) {
    if (a.nch_total > S_MAX_CHUNKS) return false;
    const Tuning tune = tuning();
    if (tune.f32_no_glds || tune.f32_no_small) return false;
    if (epi == GEMM_EPI_GATE_RS) {     // fused WaveFlow layer (bm = 128, one M-block): 128 x 128 blocks of 128 x 32 wave tiles
        if (!(a.bm == 128 && a.MB == 1 && a.gate == GATE_GTU)) return false;
        // under the split-bf16 loops only the split-K range: there the launch is latency-bound and the fp32 split-K shape
        // (exact products: never less accurate than what was asked for) beats the split loop of the 128 x 256 shape
        // (config 4 at batch 1: 38.8 ms against 67 / 83 ms)
        if (gemm_mode_is_split(a.gemm_mode))
            return !tune.f32_no_splitk && gate_rs_blocks(a) <= GATE_RS_SPLITK_MAX_BLOCKS;
        return gate_rs_blocks(a) < GATE_RS_SMALL_BELOW_BLOCKS || tune.f32_force_small;
    }
    if (a.bm != 256 && a.bm != 128) return false;
    if (!(epi == GEMM_EPI_SPLIT || epi == GEMM_EPI_GATE || epi == GEMM_EPI_GATEX || epi == GEMM_EPI_MAG || epi == GEMM_EPI_LOG ||
          epi == GEMM_EPI_LRELU || epi == GEMM_EPI_TANH))
        return false;
    // Below four rounds of 256 x 128 blocks (two per CU) the small shape wins since its main loop was written by hand:
    // measured on the final tree (profiles/r3_30_shape_crossover.txt) - ax notebook B=8 (1472 large blocks) 332.9 -> 317.9 ms,
    // untts B=4 (1104) 254.8 -> 239.9, WaveGlow 12 x 512 at B=2 (1800) 219.7 -> 209.3 and B=1 113.2 -> 109.9; from B=3 (2700
    // blocks) on the large shape is ahead (303.5 vs 308.4 ms; headline B=8: 0.857 vs 0.783 of the MFMA peak)
    // (the 128-row packing - WaveFlow above 64 channels: bound measured in round 5, see the constant)
    const long long below = a.bm == 128 ? SMALL_BELOW_LARGE_BLOCKS_BM128 : SMALL_BELOW_LARGE_BLOCKS;
    return (long long)a.MB * a.ntiles * a.batch < below || tune.f32_force_small;
}

int launch_gemm_f32_small(int epi, const GemmArgs& a, hipStream_t stream) {
    if (epi == GEMM_EPI_GATE_RS && !tuning().f32_no_splitk && gate_rs_blocks(a) <= GATE_RS_SPLITK_MAX_BLOCKS) {
        const int nt = (a.L + K_BN - 1) / K_BN;
        const long long nblk = (long long)nt * a.batch;
        CTTS_CHECK_ARG(nblk > 0 && nblk < (1ll << 31), "gemm (split-K fused shape): grid %lld", nblk);
        constexpr size_t LDS = K_LDS_FLOATS * sizeof(float);         // above the 64 KiB default: opt in once per kernel
        const bool w8 = !tuning().f32_splitk_w4;                     // eight waves per tile (two per SIMD), CTTS_F32_SPLITK_W4: four
        const int vi = (a.nseg <= 4 ? 0 : 1) + (w8 ? 2 : 0);
        const void* fns[4] = {reinterpret_cast<const void*>(conv_gemm_f32_gate_rs_splitk_kernel<4>),
                              reinterpret_cast<const void*>(conv_gemm_f32_gate_rs_splitk_kernel<GEMM_MAX_SEG>),
                              reinterpret_cast<const void*>(conv_gemm_f32_gate_rs_splitk8_kernel<4>),
                              reinterpret_cast<const void*>(conv_gemm_f32_gate_rs_splitk8_kernel<GEMM_MAX_SEG>)};
        {   // once per process and kernel (one process per GPU); a failure is reported by every call that meets it
            static std::mutex mu;
            static bool attr_set[4] = {false, false, false, false};
            std::lock_guard<std::mutex> lk(mu);
            if (!attr_set[vi]) {
                CTTS_CHECK_HIP(hipFuncSetAttribute(fns[vi], hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
                attr_set[vi] = true;
            }
        }
        if (vi == 0) hipLaunchKernelGGL((conv_gemm_f32_gate_rs_splitk_kernel<4>), dim3((unsigned)nblk), dim3(256), LDS, stream, a, nt);
        else if (vi == 1) hipLaunchKernelGGL((conv_gemm_f32_gate_rs_splitk_kernel<GEMM_MAX_SEG>), dim3((unsigned)nblk), dim3(256), LDS, stream, a, nt);
        else if (vi == 2) hipLaunchKernelGGL((conv_gemm_f32_gate_rs_splitk8_kernel<4>), dim3((unsigned)nblk), dim3(512), LDS, stream, a, nt);
        else hipLaunchKernelGGL((conv_gemm_f32_gate_rs_splitk8_kernel<GEMM_MAX_SEG>), dim3((unsigned)nblk), dim3(512), LDS, stream, a, nt);
        note_gemm_loop(16 | 32);                                     // fp32 MFMA whatever the mode (see cookietts_hip.h)
        CTTS_CHECK_LAUNCH("conv_gemm_f32_gate_rs_splitk");
        return CTTS_OK;
    }
    if (epi == GEMM_EPI_GATE_RS) {
        const int nt = (a.L + R_BN - 1) / R_BN;
        const long long nblk = (long long)nt * a.batch;
        CTTS_CHECK_ARG(nblk > 0 && nblk < (1ll << 31), "gemm (small fused shape): grid %lld", nblk);
        if (a.nseg <= 4) hipLaunchKernelGGL((conv_gemm_f32_gate_rs_small_kernel<4>), dim3((unsigned)nblk), dim3(256), 0, stream, a, nt);
        else hipLaunchKernelGGL((conv_gemm_f32_gate_rs_small_kernel<GEMM_MAX_SEG>), dim3((unsigned)nblk), dim3(256), 0, stream, a, nt);
        note_gemm_loop(16);
        CTTS_CHECK_LAUNCH("conv_gemm_f32_gate_rs_small");
        return CTTS_OK;
    }
    const int ntiles_s = (a.L + S_BN - 1) / S_BN;
    const long long blocks = (a.bm == 128 ? 1ll : 2ll) * a.MB * ntiles_s * a.batch;
    CTTS_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "gemm (small shape): grid %lld", blocks);
    const dim3 grid((unsigned)blocks);
    switch (epi) {
        case GEMM_EPI_GATEX: launch_small<GEMM_EPI_GATEX>(grid, stream, a, ntiles_s); break;
        case GEMM_EPI_GATE: launch_small<GEMM_EPI_GATE>(grid, stream, a, ntiles_s); break;
        case GEMM_EPI_MAG: launch_small<GEMM_EPI_MAG>(grid, stream, a, ntiles_s); break;
        case GEMM_EPI_LOG: launch_small<GEMM_EPI_LOG>(grid, stream, a, ntiles_s); break;
        case GEMM_EPI_LRELU: launch_small<GEMM_EPI_LRELU>(grid, stream, a, ntiles_s); break;
        case GEMM_EPI_TANH: launch_small<GEMM_EPI_TANH>(grid, stream, a, ntiles_s); break;
        case GEMM_EPI_SPLIT: launch_small<GEMM_EPI_SPLIT>(grid, stream, a, ntiles_s); break;
        default: set_error("gemm (small shape): epilogue %d", epi); return CTTS_E_ARG;
    }
    note_gemm_loop(16 | gemm_split_level(a.gemm_mode));
    CTTS_CHECK_LAUNCH("conv_gemm_f32_small");
    return CTTS_OK;
}

// ---- WaveFlow row step as one launch (wf_row_persistent_kernel) -------------------------------------------------------------
bool wf_row_persistent_supported(const GemmArgs& a) {
    if (!(a.bm == 128 && a.MB == 1 && a.gate == GATE_GTU && a.pairC <= 64 && a.rs_wT && a.rs_bias)) return false;
    if (a.nch_total > S_MAX_CHUNKS || gemm_mode_is_split(a.gemm_mode)) return false;
    const Tuning tune = tuning();
    return !(tune.f32_no_glds || tune.f32_no_small);
}

int wf_row_cus() {                                  // CUs of the current device (one process per GPU: asked once)
    static std::atomic<int> n_cu{0};
    int n = n_cu.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
        n_cu.store(n, std::memory_order_relaxed);
    }
    return n;
}

int wf_row_tiles(int L, int body) { return body == 1 ? (L + K_BN - 1) / K_BN : (L + R_BN - 1) / R_BN; }

int launch_wf_row_persistent(const GemmArgs* layers_dev, const WfTailDesc* tails_dev, int nrows, int nlayers, int max_nseg, int L,
                             int batch, int body, unsigned int* counter, unsigned int* flags, unsigned int* abort_word,
                             unsigned int epoch, hipStream_t stream) {
    WfRowArgs w{};
    w.layers = layers_dev; w.tails = tails_dev; w.nrows = nrows; w.nlayers = nlayers; w.ntiles_s = wf_row_tiles(L, 0); w.ntiles_k = wf_row_tiles(L, 1); w.batch = batch;
    w.counter = counter; w.flags = flags; w.abort_word = abort_word; w.epoch = epoch;
    w.timeout_ticks = 100u * 1000u * 1000u;                             // 1 s without a single item being claimed, twice in a row (a whole call is ~0.2 s)
    w.debug = tuning().wf_queue_debug;
    CTTS_CHECK_ARG(layers_dev && counter && flags && abort_word && epoch > 0 && nlayers >= 1 && nrows >= 1 && batch >= 1 &&
                       w.ntiles_s >= 1 && (body == 0 || body == 1), "waveflow row launch: bad argument");
    const int n_cu = wf_row_cus();
    // Two resident workgroups per CU, never more than the items of two layers (the rest could only wait).  A layer of no more
    // items than CUs gets ONE workgroup per item: a second workgroup on a CU would run the next layer's item next to the
    // straggler it waits for and halve that one's matrix pipe (config 4, batch 1: 55 ms per call with 2 per CU)
    const long long per_layer = (long long)(body == 1 ? w.ntiles_k : w.ntiles_s) * batch;
    long long want = std::min<long long>(2ll * n_cu, std::min<long long>((long long)nrows * nlayers * per_layer, 2 * per_layer));
    if (per_layer <= n_cu && !(w.debug & 64)) want = per_layer;
    const dim3 grid((unsigned)std::max<long long>(want, 1));
    const int vi = max_nseg <= 4 ? 0 : 1;
    if (body == 1) {
        constexpr size_t LDS = K_LDS_FLOATS * sizeof(float);            // above the 64 KiB default: opt in once per kernel
        {
            static std::mutex mu;
            static bool attr_set[2] = {false, false};
            std::lock_guard<std::mutex> lk(mu);
            if (!attr_set[vi]) {
                const void* fn = vi == 0 ? reinterpret_cast<const void*>(wf_row_persistent_kernel<4, 1>)
                                         : reinterpret_cast<const void*>(wf_row_persistent_kernel<GEMM_MAX_SEG, 1>);
                CTTS_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
                attr_set[vi] = true;
            }
        }
        if (vi == 0) hipLaunchKernelGGL((wf_row_persistent_kernel<4, 1>), grid, dim3(256), LDS, stream, w);
        else hipLaunchKernelGGL((wf_row_persistent_kernel<GEMM_MAX_SEG, 1>), grid, dim3(256), LDS, stream, w);
        note_gemm_loop(16 | 32 | 64 | (tails_dev ? 128 : 0));
    } else {
        if (vi == 0) hipLaunchKernelGGL((wf_row_persistent_kernel<4, 0>), grid, dim3(256), 0, stream, w);
        else hipLaunchKernelGGL((wf_row_persistent_kernel<GEMM_MAX_SEG, 0>), grid, dim3(256), 0, stream, w);
        note_gemm_loop(16 | 64 | (tails_dev ? 128 : 0));
    }
    CTTS_CHECK_LAUNCH("wf_row_persistent");
    return CTTS_OK;
}

}  // namespace ctts
