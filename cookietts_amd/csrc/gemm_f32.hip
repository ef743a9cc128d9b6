// fp32 MFMA conv-GEMM for gfx950 (see gemm_f32.h for the contract).
//
// Tiling (CDNA4-first, 64-wide waves):
//   workgroup = 256 threads = 4 waves; wave tile 128 x 64 = 4 x 2 tiles of
//   v_mfma_f32_32x32x2_f32 -> 128 accumulator VGPRs, ~200 VGPRs total, 2 waves per SIMD
//   (2 workgroups per CU) so one wave's LDS/barrier time is covered by the other
//   wave's MFMAs; the f32 MFMA pipe (64 cycles per 32x32x2) is the bound.
//   Block tile 256 x 128 (waves 2x2) or 128 x 256 (waves 1x4); K chunk 16.
//   Default kernel (GLDS = true): 3 LDS stages x 24 KiB filled by direct global->LDS DMA
//   (global_load_lds, 16 B per lane), issued two chunks ahead, one DMA piece per k-step behind that
//   k-step's eight MFMAs, counted vmcnt + raw s_barrier; a chunk -> B-address table built once per
//   workgroup keeps the loop body one basic block.  Both operands are k-major so a fragment pair is
//   one conflict-free ds_read2_b32 (lane l: row/col l & 31, k = l >> 5); the LDS->MFMA software
//   pipeline is pinned with sched_group_barrier.
//   GLDS = false (CTTS_F32_NO_GLDS=1, or more K chunks than the address table holds): the older
//   2-stage variant that stages global->LDS through registers one chunk ahead.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "gemm_f32.h"
#include "tuning.h"
#include "gemm_bf16.h"   // pack_bf16x2 (split-bf16 main loop)

namespace ctts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int STAGE = GEMM_KC * (256 + 128);            // 6144 floats = 24 KiB for both shapes
// (the segment table, GEMM_MAX_SEG x 4 dwords, sits behind the last stage)
constexpr int GEMM_GLDS_MAX_CHUNKS = 384;               // DMA-staged kernels: chunk address table entries

// Gate math on the hardware transcendental unit: exp via v_exp_f32 (2^x), reciprocal via
// v_rcp_f32 (1 ulp).  Absolute error of tanh/sigmoid <= ~3e-7, far inside the parity budget.
__device__ __forceinline__ float fast_sigmoid(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
}
__device__ __forceinline__ float fast_tanh(float u) {
    // tanh(u) = 1 - 2 / (1 + e^{2u}); saturates correctly at +-inf
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * 2.8853900817779268f));
}

// 16 bytes from a 4-byte-aligned address (dilation 1 and 2 taps): the backend emits one
// global_load_dwordx4, which gfx950 serves unaligned.
// The pointer is rebuilt from the LDS segment table, so it is cast to the global address space
// explicitly: a generic (flat) load would also count on lgkmcnt and every LDS fragment wait in the
// MFMA loop would then drain the prefetch of the next chunk.
typedef const __attribute__((address_space(1))) float* gfloat_ptr;
__device__ __forceinline__ float4 load4u(gfloat_ptr p) {
    float4 v;
    v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
    return v;
}

// Store / read-modify-write epilogue of one wave tile (4 x 2 MFMA tiles = 128 rows x 64 columns).
// rows < split -> dst0 (= src0 + v when acc0), rows >= split -> dst1[row - split] (+= when acc1).
template <int EPI>
__device__ __forceinline__ void split_epilogue(const GemmArgs& a, f32x16 (&acc)[4][2], const float* bias, int M, int row0,
                                               int b, int ncol0, int l31, int lhi) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int rbase = row0 + mt * 32;                        // uniform per tile
        if (rbase >= M) continue;                                // zero-padded rows of a ragged M
        const bool second = rbase >= a.split;                    // split is a multiple of 32
        float* dst = second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride;
        const float* src = second ? dst : (a.src0 ? a.src0 + (size_t)b * a.src0_bstride : dst);
        const int accum = second ? a.acc1 : a.acc0;
        const int rdst = second ? rbase - a.split : rbase;
        // read-modify-write: issue all 32 loads of this row-tile before the first store so
        // the wave pays one memory latency per tile, not one per element (the compiler must
        // otherwise order every load behind the previous, possibly aliasing, store).
        float old[2][16];
        if (accum && rbase + 32 <= M) {   // uniform; columns >= L of a padded row are readable, so no per-lane guard
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    old[nt][r] = src[(size_t)(rdst + row) * a.dst_ld + a.dst_pad + ncol0 + nt * 32 + l31];
                }
        } else if (accum) {               // last row tile of a ragged M (the postnet's 80 mel rows): rows >= M do not
                                          // exist in the destination tensor - reading them ran past the end of the last
                                          // batch item's allocation (a memory fault when it ends at a mapping boundary)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    old[nt][r] = rbase + row < M ? src[(size_t)(rdst + row) * a.dst_ld + a.dst_pad + ncol0 + nt * 32 + l31] : 0.0f;
                }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) old[nt][r] = 0.0f;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = ncol0 + nt * 32 + l31;
            if (n < a.L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float v = acc[mt][nt][r] + bias[(rbase - row0) + row] + old[nt][r];
                    if constexpr (EPI == GEMM_EPI_LOG) v = logf(fmaxf(v, a.clip));
                    if constexpr (EPI == GEMM_EPI_LRELU) v = v > 0.f ? v : a.clip * v;
                    if constexpr (EPI == GEMM_EPI_TANH) v = tanhf(v);
                    if (rbase + row < M) dst[(size_t)(rdst + row) * a.dst_ld + a.dst_pad + n] = v;
                }
            }
        }
    }
}

// SEGS = 4: segment bases live in registers and are picked with a scalar-compare select chain (no LDS
// round trip at the top of a chunk); SEGS = GEMM_MAX_SEG: bases come from the LDS segment table.
// GLDS: global -> LDS staging by direct DMA (global_load_lds, 16 B per lane), three stages, two chunks ahead, counted
// vmcnt + raw s_barrier.  Measured on the config-2 in-layer launch: with staging through registers (6 x 16-byte
// global loads + 6 x ds_write_b128 per thread per chunk) the kernel ran at 81.6 % of the MFMA peak, with the staging
// removed altogether (stale operands) at 89.9 %: the LDS write port and the VGPR round trip were delaying the
// fragment reads that feed the matrix pipe.
//
// X3 (split-bf16 main loop, GLDS only): the SAME staging, LDS image and fragment reads as the fp32 loop - lane (l31,
// lhi) reads the 8 values k = 2 ks + lhi of its row / column - but the 8 values become ONE v_mfma_f32_32x32x16_bf16
// operand: hi = bf16(v), lo = bf16(v - hi), and a 32x32 tile of the chunk is three bf16 MFMAs hi*hi + hi*lo + lo*hi
// (96 matrix-pipe cycles) instead of eight fp32 MFMAs (512).  Operands carry 16 mantissa bits, accumulation is fp32;
// tensors, packed weights and epilogues are untouched, so every fp32 path of the library can run on it.
// X6 (XS = 6): a three-way split hi + mid + lo (3 x 8 = 24 mantissa bits = all of an fp32 operand) and the six products
// whose weight is >= 2^-16 of the leading one: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, accumulated in that order
// (smallest first); the three dropped products are <= 2^-24 relative.  192 matrix-pipe cycles per tile and chunk.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8(const float (&v)[8], u32x4_t& hi, u32x4_t& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int h = pack_bf16x2(v[2 * j], v[2 * j + 1]);
        hi[j] = h;
        lo[j] = pack_bf16x2(v[2 * j] - __builtin_bit_cast(float, h << 16), v[2 * j + 1] - __builtin_bit_cast(float, h & 0xffff0000u));
    }
}

__device__ __forceinline__ void split8x3(const float (&v)[8], u32x4_t& hi, u32x4_t& mid, u32x4_t& lo) {
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

template <int EPI, int WM, int SEGS, bool GLDS, int XS = 0>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_kernel(const GemmArgs a) {
    constexpr bool X3 = XS != 0;                           // any split-bf16 main loop (XS = 3 or 6 products)
    constexpr bool X6 = XS == 6;
    static_assert(XS == 0 || XS == 3 || XS == 6, "main loop: fp32 MFMA, 3 or 6 bf16 products");
    static_assert(!X3 || GLDS, "the split-bf16 loop is built on the DMA-staged pipeline");
    constexpr int NST = GLDS ? 3 : 2;
    constexpr int SEGTAB = NST * STAGE;
    constexpr int CHTAB = SEGTAB + GEMM_MAX_SEG * 4;         // GLDS: chunk -> B base address table (8 B per chunk)
    constexpr int LDS_FLOATS = CHTAB + (GLDS ? 2 * GEMM_GLDS_MAX_CHUNKS : 0);
    constexpr int BM = 128 * WM;
    constexpr int WN = 4 / WM;
    constexpr int BN = 64 * WN;
    constexpr int A_STAGE = GEMM_KC * BM;
    constexpr int NA = BM / 64;                          // float4 loads per thread per A stage
    constexpr int NB = BN / 64;                          // float4 loads per thread per B stage
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = GLDS ? __builtin_amdgcn_readfirstlane(t >> 6) : (t >> 6);   // scalar for the DMA's LDS bases
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // block -> (m-block, n-tile, batch).  Dispatch places block id on XCD id % 8, so with
    // mb = id % MB an XCD keeps re-using the same 1-2 weight slices in its private L2.
    int id = blockIdx.x;
    int mb, tile, b;
    if (a.map_mode == 1) {
        // MB == 4 variant: XCD x owns the m-block PAIR {2(x&1), 2(x&1)+1} of the column tiles t = 4q + (x>>1); the
        // two m-blocks of one tile are consecutive ids of that XCD, so the B tile is fetched into 2 L2s, not 4.
        const int x = id & 7, j = id >> 3;
        mb = 2 * (x & 1) + (j & 1);
        const int gt = (j >> 1) * 4 + (x >> 1);
        if (gt >= a.gt_limit) return;                    // whole workgroup: grid is rounded up to 4 tiles
        tile = gt % a.ntiles;
        b = gt / a.ntiles;
    } else {
        mb = id % a.MB;
        id /= a.MB;
        tile = id % a.ntiles;
        b = id / a.ntiles;
    }
    const int n0 = tile * BN;

    // segment bases: everything except the k-row of the chunk and the per-thread (row, column) offset
    gfloat_ptr sbase[4];
    int snch[4];
    if constexpr (SEGS == 4) {
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {                 // static kernarg indices only
            const GemmSeg& g = a.seg[sidx];
            sbase[sidx] = (gfloat_ptr)(g.base + (size_t)b * g.bstride + (size_t)(mb * g.mb_rows) * a.ld +
                                       (a.pad + n0 + g.shift));
            snch[sidx] = sidx < a.nseg ? g.nch : 0x7fffffff;
        }
    } else {
        // segment table -> LDS (a dynamically indexed kernarg struct would be copied to scratch)
#pragma unroll
        for (int sidx = 0; sidx < GEMM_MAX_SEG; ++sidx) {      // static kernarg indices only
            if (t == sidx) {
                const GemmSeg& g = a.seg[sidx];
                unsigned int* e = reinterpret_cast<unsigned int*>(lds + SEGTAB + sidx * 4);
                if (sidx < a.nseg) {
                    const float* base = g.base + (size_t)b * g.bstride + (size_t)(mb * g.mb_rows) * a.ld +
                                        (a.pad + n0 + g.shift);
                    const unsigned long long u = reinterpret_cast<unsigned long long>(base);
                    e[0] = (unsigned int)u;
                    e[1] = (unsigned int)(u >> 32);
                    e[2] = (unsigned int)g.nch;
                } else {
                    e[0] = 0; e[1] = 0; e[2] = 0x7fffffffu;
                }
                e[3] = 0;
            }
        }
    }

    const int nalloc = a.a_nch_alloc ? a.a_nch_alloc : a.nch_total;
    const float* ap = a.A + ((size_t)mb * nalloc + a.a_ch_off) * A_STAGE + t * 4;

    // register staging: thread -> (k row t/16, columns 4(t%16) + 64j).  DMA staging: a wave instruction fills 1 KiB of
    // LDS linearly in lane order = 256/BN whole k-rows, piece (wave + 4j) of the stage.
    constexpr int UPR = BN / 4;                              // 16-byte units per k-row of the B stage
    constexpr int RPP = 64 / UPR;                            // k-rows per DMA piece (2 or 1)
    const int brow = GLDS ? wave * RPP + lane / UPR : t >> 4;
    const int bcol = GLDS ? (lane % UPR) * 4 : (t & 15) * 4;
    const size_t thread_off = (size_t)brow * a.ld + bcol;
    const size_t piece_stride = (size_t)4 * RPP * a.ld;      // DMA: next piece of the same thread
    const size_t chunk_rows = (size_t)GEMM_KC * a.ld;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // named registers (arrays indexed inside the pipelined loop end up in scratch)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    int seg = 0, local = 0;
    const int ilv = a.interleave > 1 ? a.interleave : 0;
    bool in_ilv = ilv > 0;
    if constexpr (SEGS != 4) __syncthreads();   // segment table visible
    if constexpr (GLDS) {
        // chunk c -> address of its B rows (k row 0, column n0 + shift): the segment / interleave sequencing is
        // resolved once, so the loop body below is one basic block the scheduler can weave DMA issues into
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(lds + CHTAB);
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
                for (int k = 0; k < GEMM_MAX_SEG - 1; ++k)   // static kernarg indices only
                    if (k < SEGS - 1 && sg == k && k < a.nseg - 1 && c >= a.seg[k].nch) { c -= a.seg[k].nch; sg = k + 1; }
                loc = c;
            }
            const float* base = nullptr;
#pragma unroll
            for (int k = 0; k < GEMM_MAX_SEG; ++k)
                if (k < SEGS && sg == k)
                    base = a.seg[k].base + (size_t)b * a.seg[k].bstride + (size_t)(mb * a.seg[k].mb_rows) * a.ld +
                           (a.pad + n0 + a.seg[k].shift);
            tab[c0] = reinterpret_cast<unsigned long long>(base + (size_t)loc * GEMM_KC * a.ld);
        }
        __syncthreads();
    }

#define CTTS_ISSUE_LOADS()                                                                      \
    do {                                                                                        \
        ra0 = *reinterpret_cast<const float4*>(ap);                                             \
        ra1 = *reinterpret_cast<const float4*>(ap + 1024);                                      \
        if constexpr (NA > 2) {                                                                 \
            ra2 = *reinterpret_cast<const float4*>(ap + 2048);                                  \
            ra3 = *reinterpret_cast<const float4*>(ap + 3072);                                  \
        }                                                                                       \
        ap += A_STAGE;                                                                          \
        gfloat_ptr sb_;                                                                         \
        int sn_;                                                                                \
        if constexpr (SEGS == 4) {                                                              \
            sb_ = seg == 0 ? sbase[0] : seg == 1 ? sbase[1] : seg == 2 ? sbase[2] : sbase[3];   \
            sn_ = seg == 0 ? snch[0] : seg == 1 ? snch[1] : seg == 2 ? snch[2] : snch[3];       \
        } else {                                                                                \
            const uint4 e = *reinterpret_cast<const uint4*>(lds + SEGTAB + seg * 4);            \
            sb_ = reinterpret_cast<gfloat_ptr>(((unsigned long long)e.y << 32) | e.x);          \
            sn_ = (int)e.z;                                                                     \
        }                                                                                       \
        gfloat_ptr bp = sb_ + (size_t)local * chunk_rows + thread_off;                          \
        rb0 = load4u(bp);                                                                       \
        rb1 = load4u(bp + 64);                                                                  \
        if constexpr (NB > 2) {                                                                 \
            rb2 = load4u(bp + 128);                                                             \
            rb3 = load4u(bp + 192);                                                             \
        }                                                                                       \
        if (in_ilv) {                                                                           \
            if (++seg == ilv) { seg = 0; if (++local == sn_) { local = 0; seg = ilv; in_ilv = false; } } \
        } else if (++local == sn_) { local = 0; ++seg; }                                        \
    } while (0)

#define CTTS_STORE_LDS(buf)                                                                     \
    do {                                                                                        \
        float* As_ = lds + (buf) * STAGE + t * 4;                                               \
        float* Bs_ = lds + (buf) * STAGE + A_STAGE + brow * BN + bcol;                          \
        *reinterpret_cast<float4*>(As_) = ra0;                                                  \
        *reinterpret_cast<float4*>(As_ + 1024) = ra1;                                           \
        if constexpr (NA > 2) {                                                                 \
            *reinterpret_cast<float4*>(As_ + 2048) = ra2;                                       \
            *reinterpret_cast<float4*>(As_ + 3072) = ra3;                                       \
        }                                                                                       \
        *reinterpret_cast<float4*>(Bs_) = rb0;                                                  \
        *reinterpret_cast<float4*>(Bs_ + 64) = rb1;                                             \
        if constexpr (NB > 2) {                                                                 \
            *reinterpret_cast<float4*>(Bs_ + 128) = rb2;                                        \
            *reinterpret_cast<float4*>(Bs_ + 192) = rb3;                                        \
        }                                                                                       \
    } while (0)

    typedef __attribute__((address_space(3))) float* lds_fptr;
    typedef const __attribute__((address_space(1))) char* gbyte_ptr;
    // wave-uniform A base of this m-block and the 32-bit per-lane byte offsets of the two operands
    const gbyte_ptr apu = (gbyte_ptr)(a.A + ((size_t)mb * nalloc + a.a_ch_off) * A_STAGE);
    unsigned piece_lane[6];
#pragma unroll
    for (int p_ = 0; p_ < 6; ++p_)
        piece_lane[p_] = p_ < NA ? (unsigned)(t * 16 + 4096 * p_) : (unsigned)(thread_off * 4 + (size_t)(p_ - NA) * (piece_stride * 4));
    const unsigned long long* ctab = reinterpret_cast<const unsigned long long*>(lds + CHTAB);
    // DMA piece p (0..5) of chunk c into stage buf: pieces [0, NA) = A, [NA, 6) = B.  Every address is a wave-uniform
    // 64-bit base (ac_ / bp_, SGPR pair) + a 32-bit lane offset, laundered through an empty asm so that the compiler keeps
    // that form (global_load_lds_dwordx4 v, s[a:b]) instead of folding it into a loop-carried 64-bit VGPR address
    // (v[a:b], off): with the 64-bit form every DMA blocks the matrix pipe for ~20 cycles (round 5,
    // profiles/r5_07_bf16_mix_ceiling_dma_forms.txt).
#define CTTS_GLDS_PIECE(p, la_, ac_, bp_)                                                                   \
    do {                                                                                                    \
        unsigned o_ = piece_lane[(p)];                     /* lane offset incl. the piece's own offset: nothing to reassociate */ \
        asm volatile("" : "+v"(o_));                                                                        \
        if constexpr ((p) < NA) __builtin_amdgcn_global_load_lds((gfloat_ptr)((ac_) + o_), (la_) + 1024 * (p), 16, 0, 0); \
        else __builtin_amdgcn_global_load_lds((gfloat_ptr)((bp_) + o_), (la_) + A_STAGE + 1024 * ((p) - NA), 16, 0, 0); \
    } while (0)
#define CTTS_GLDS_ADDR_A(buf, c)                                                                            \
    lds_fptr la_ = (lds_fptr)(lds + (buf) * STAGE + wave * 256);                                            \
    const gbyte_ptr ac_ = apu + (size_t)(c) * (A_STAGE * 4);                                                \
    const unsigned long long ub_ = ctab[c];
#define CTTS_GLDS_ADDR_B()                                                                                  \
    const unsigned long long us_ =                                                                          \
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub_ >> 32)) << 32) |            \
        (unsigned)__builtin_amdgcn_readfirstlane((int)ub_);                                                 \
    const gbyte_ptr bp_ = reinterpret_cast<gbyte_ptr>(us_);
#define CTTS_GLDS_ADDR(buf, c) CTTS_GLDS_ADDR_A(buf, c) CTTS_GLDS_ADDR_B()
#define CTTS_ISSUE_GLDS(buf, c)                                                                             \
    do {                                                                                                    \
        CTTS_GLDS_ADDR(buf, c)                                                                              \
        CTTS_GLDS_PIECE(0, la_, ac_, bp_); CTTS_GLDS_PIECE(1, la_, ac_, bp_); CTTS_GLDS_PIECE(2, la_, ac_, bp_); \
        CTTS_GLDS_PIECE(3, la_, ac_, bp_); CTTS_GLDS_PIECE(4, la_, ac_, bp_); CTTS_GLDS_PIECE(5, la_, ac_, bp_); \
    } while (0)

    const int nch = a.nch_total;
    if constexpr (GLDS) {
        // six DMAs per thread per chunk, in order: vmcnt(6) = "everything but the newest chunk has landed"
        // (the last two iterations re-issue the final chunk into a stage nobody reads any more: the body stays
        // branch-free and the DMA count per iteration constant)
        CTTS_ISSUE_GLDS(0, 0);
        CTTS_ISSUE_GLDS(1, nch > 1 ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    } else {
        CTTS_ISSUE_LOADS();
        CTTS_STORE_LDS(0);
        __syncthreads();
    }

    int cur = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        const float* As = lds + cur * STAGE + wm * 128 + l31;
        const float* Bs = lds + cur * STAGE + A_STAGE + wn * 64 + l31;
        [[maybe_unused]] float av[GEMM_KC / 2][4], bv[GEMM_KC / 2][2];
        if constexpr (X3) {
            const int nb = cur >= 1 ? cur - 1 : 2;          // (cur + 2) % 3: the stage of chunk ch-1
            const int cn = ch + 2 < nch ? ch + 2 : nch - 1;
            CTTS_GLDS_ADDR(nb, cn)
            // B fragments first, then per row tile: [read + split A(mt + 1) | 6 MFMAs of mt | DMA pieces]: the VALU work
            // of the next row tile sits in the shadow of this row tile's 192 matrix-pipe cycles
            u32x4_t ah[2], al[2], bh[2], bl[2];
            [[maybe_unused]] u32x4_t am[2], bm[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float v[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) v[ks] = Bs[(2 * ks + lhi) * BN + nt * 32];
                if constexpr (X6) split8x3(v, bh[nt], bm[nt], bl[nt]);
                else split8(v, bh[nt], bl[nt]);
            }
#define CTTS_X3_A(mt, slot)                                                                     \
            {                                                                                   \
                float v[8];                                                                     \
                _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) v[ks] = As[(2 * ks + lhi) * BM + (mt) * 32]; \
                if constexpr (X6) split8x3(v, ah[slot], am[slot], al[slot]);                    \
                else split8(v, ah[slot], al[slot]);                                             \
            }
#define CTTS_X3_P(A_, B_, mt, nt)                                                               \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A_), __builtin_bit_cast(bf16x8_t, B_), acc[mt][nt], 0, 0, 0);
#define CTTS_X3_MFMA6(mt, slot)                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                  \
                CTTS_X3_P(al[slot], bh[nt], mt, nt)                                             \
                CTTS_X3_P(ah[slot], bl[nt], mt, nt)                                             \
                if constexpr (X6) {                                                             \
                    CTTS_X3_P(am[slot], bm[nt], mt, nt)                                         \
                    CTTS_X3_P(am[slot], bh[nt], mt, nt)                                         \
                    CTTS_X3_P(ah[slot], bm[nt], mt, nt)                                         \
                }                                                                               \
                CTTS_X3_P(ah[slot], bh[nt], mt, nt)                                             \
            }
            CTTS_X3_A(0, 0)
            __builtin_amdgcn_sched_barrier(0);
            CTTS_X3_A(1, 1)
            CTTS_X3_MFMA6(0, 0)
            CTTS_GLDS_PIECE(0, la_, ac_, bp_); CTTS_GLDS_PIECE(1, la_, ac_, bp_);
            __builtin_amdgcn_sched_barrier(0);
            CTTS_X3_A(2, 0)
            CTTS_X3_MFMA6(1, 1)
            CTTS_GLDS_PIECE(2, la_, ac_, bp_); CTTS_GLDS_PIECE(3, la_, ac_, bp_);
            __builtin_amdgcn_sched_barrier(0);
            CTTS_X3_A(3, 1)
            CTTS_X3_MFMA6(2, 0)
            CTTS_GLDS_PIECE(4, la_, ac_, bp_); CTTS_GLDS_PIECE(5, la_, ac_, bp_);
            __builtin_amdgcn_sched_barrier(0);
            CTTS_X3_MFMA6(3, 1)
#undef CTTS_X3_A
#undef CTTS_X3_P
#undef CTTS_X3_MFMA6
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                // chunk ch+1 landed, the newest in flight
            __builtin_amdgcn_s_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        } else if constexpr (GLDS) {
            // One region per k-step, fenced: [fragments of k-step ks+1 | 8 MFMAs of k-step ks | one DMA piece of chunk
            // ch+2].  The DMA issue (~60 cycles of this wave's instruction stream) sits behind eight 64-cycle MFMAs
            // already queued on the matrix pipe instead of in front of the chunk.
            const int nb = cur >= 1 ? cur - 1 : 2;          // (cur + 2) % 3: the stage of chunk ch-1
            const int cn = ch + 2 < nch ? ch + 2 : nch - 1;
            CTTS_GLDS_ADDR_A(nb, cn)        // (the table entry is only needed by the B pieces: resolved after region 1)
#define CTTS_READ_FRAGS(ks)                                                                     \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) av[ks][mt] = As[(2 * (ks) + lhi) * BM + mt * 32]; \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) bv[ks][nt] = Bs[(2 * (ks) + lhi) * BN + nt * 32];
#define CTTS_MFMA8(ks)                                                                          \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                    \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                \
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][mt], bv[ks][nt], acc[mt][nt], 0, 0, 0);
#define CTTS_REGION(ks, p, BP)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                  \
            if constexpr ((ks) + 1 < GEMM_KC / 2) { CTTS_READ_FRAGS((ks) + 1) }                 \
            CTTS_MFMA8(ks)                                                                      \
            if constexpr ((p) >= 0) CTTS_GLDS_PIECE((p) < 0 ? 0 : (p), la_, ac_, BP);
            CTTS_READ_FRAGS(0)
            // pieces 0, 1 are A pieces for both block shapes; the B address comes out of the table after them
            CTTS_REGION(0, 0, ac_) CTTS_REGION(1, 1, ac_)
            __builtin_amdgcn_sched_barrier(0);
            CTTS_GLDS_ADDR_B()
            CTTS_REGION(2, 2, bp_) CTTS_REGION(3, 3, bp_)
            CTTS_REGION(4, 4, bp_) CTTS_REGION(5, 5, bp_) CTTS_REGION(6, -1, bp_) CTTS_REGION(7, -1, bp_)
#undef CTTS_REGION
#undef CTTS_MFMA8
#undef CTTS_READ_FRAGS
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                // chunk ch+1 landed, the newest in flight
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == 2 ? 0 : cur + 1;
        } else {
            if (more) CTTS_ISSUE_LOADS();
            // k-step ks+1's fragments are read from LDS while ks runs on the MFMA pipe; the
            // sched_group_barrier sequence pins that software pipeline (hipcc otherwise sinks every
            // ds_read to just before its first use and exposes the LDS latency 16x per chunk).
#pragma unroll
            for (int ks = 0; ks < GEMM_KC / 2; ++ks) {
                const int krow = 2 * ks + lhi;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) av[ks][mt] = As[krow * BM + mt * 32];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bv[ks][nt] = Bs[krow * BN + nt * 32];
            }
#pragma unroll
            for (int ks = 0; ks < GEMM_KC / 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][mt], bv[ks][nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);      // 3 x ds_read2_b32: fragments of k-step 0
#pragma unroll
            for (int ks = 0; ks < GEMM_KC / 2 - 1; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // fragments of k-step ks+1
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // 8 MFMAs of k-step ks
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            if (more) CTTS_STORE_LDS(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
#undef CTTS_ISSUE_GLDS
#undef CTTS_GLDS_ADDR
#undef CTTS_GLDS_ADDR_A
#undef CTTS_GLDS_ADDR_B
#undef CTTS_GLDS_PIECE
    if constexpr (GLDS) {                                   // the re-issued tail DMAs still target LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef CTTS_ISSUE_LOADS
#undef CTTS_STORE_LDS

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // bias via LDS: a global bias load between the stores would force vmcnt(0) (which on gfx9
    // also drains the stores) once per element.
    if (t < BM) lds[t] = a.bias[mb * BM + t];
    __syncthreads();
    const float* bias = lds + wm * 128;
    if constexpr (EPI == GEMM_EPI_GATE || EPI == GEMM_EPI_MAG || EPI == GEMM_EPI_GATEX) {
        float* dst = a.dst0 + (size_t)b * a.dst0_bstride;
        const int cbase = (mb * WM + wm) * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (cbase + mt * 32 >= a.pairC) continue;             // uniform: whole tile is channel padding
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = n0 + wn * 64 + nt * 32 + l31;
                if (n < a.L) {
                    float add0[16], add1[16];
                    if ((EPI == GEMM_EPI_GATE || EPI == GEMM_EPI_GATEX) && a.addend && a.addend_frames > 0) {   // uniform: interpolated addend
                        const int F = a.addend_frames;
                        const float scale = a.L > 1 ? (float)(F - 1) / (float)(a.L - 1) : 0.f;
                        const float real = scale * (float)n;
                        const int i0 = (int)real;
                        const int i1 = i0 + 1 < F ? i0 + 1 : F - 1;
                        const float l1 = real - (float)i0;
                        const float l0 = 1.0f - l1;
                        const float* ad = a.addend + (size_t)b * a.addend_bstride + a.addend_pad;
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {            // two halves: 32 loads in flight, not 64
                            float s00[8], s01[8], s10[8], s11[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int r = hf * 8 + q;
                                const int c = min(cbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, a.pairC - 1);
                                const float* r0 = ad + (size_t)c * a.addend_ld;
                                const float* r1 = ad + (size_t)(a.pairC + c) * a.addend_ld;
                                s00[q] = r0[i0]; s01[q] = r0[i1]; s10[q] = r1[i0]; s11[q] = r1[i1];
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                add0[hf * 8 + q] = gemm_lerp(l0, s00[q], l1, s01[q]);
                                add1[hf * 8 + q] = gemm_lerp(l0, s10[q], l1, s11[q]);
                            }
                        }
                    } else if ((EPI == GEMM_EPI_GATE || EPI == GEMM_EPI_GATEX) && a.addend) {   // uniform
                        const float* ad = a.addend + (size_t)b * a.addend_bstride + a.addend_pad + n;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int c = min(cbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, a.pairC - 1);
                            add0[r] = ad[(size_t)c * a.addend_ld];
                            add1[r] = ad[(size_t)(a.pairC + c) * a.addend_ld];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) add0[r] = add1[r] = 0.0f;
                    }
                    if constexpr (EPI == GEMM_EPI_GATEX) {
                        // one textual copy of the element loop per unit (a lambda capturing the accumulators by
                        // reference, or a run-time switch inside the loop, sends the accumulator arrays to scratch)
#define CTTS_GATEX_LOOP(K)                                                                                        \
                        case K:                                                                                   \
                            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                      \
                                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;                                 \
                                const int c = cbase + mt * 32 + row;                                              \
                                const float u0 = acc[mt][nt][r] + bias[mt * 32 + row] + add0[r];                  \
                                const float u1 = acc[mt + 2][nt][r] + bias[64 + mt * 32 + row] + add1[r];         \
                                if (c < a.pairC) dst[(size_t)c * a.dst_ld + a.dst_pad + n] = gate_eval<K>(u0, u1); \
                            }                                                                                     \
                            break;
                        switch (a.gate) {
                            CTTS_GATEX_LOOP(1) CTTS_GATEX_LOOP(2) CTTS_GATEX_LOOP(3) CTTS_GATEX_LOOP(4) CTTS_GATEX_LOOP(5)
                            CTTS_GATEX_LOOP(6) CTTS_GATEX_LOOP(7) CTTS_GATEX_LOOP(8) CTTS_GATEX_LOOP(9) CTTS_GATEX_LOOP(10)
                            CTTS_GATEX_LOOP(11) CTTS_GATEX_LOOP(12) CTTS_GATEX_LOOP(13)
                            default:
                            CTTS_GATEX_LOOP(0)
                        }
#undef CTTS_GATEX_LOOP
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                            const int c = cbase + mt * 32 + row;
                            const float u0 = acc[mt][nt][r] + bias[mt * 32 + row] + add0[r];
                            const float u1 = acc[mt + 2][nt][r] + bias[64 + mt * 32 + row] + add1[r];
                            float v;
                            if constexpr (EPI == GEMM_EPI_GATE) v = fast_tanh(u0) * fast_sigmoid(u1);
                            else v = sqrtf(u0 * u0 + u1 * u1);
                            if (c < a.pairC) dst[(size_t)c * a.dst_ld + a.dst_pad + n] = v;
                        }
                    }
                }
            }
        }
    } else if constexpr (EPI == GEMM_EPI_GATE_RS) {
        static_assert(EPI != GEMM_EPI_GATE_RS || WM == 1, "fused res/skip needs the 128-row block shape");
        // 1. gated activations of this wave's 64 channels x 64 columns, kept in registers
        float actv[2][2][16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const bool ok = mt * 32 + row < a.pairC;
                    float u0 = acc[mt][nt][r] + bias[mt * 32 + row];
                    float u1 = acc[mt + 2][nt][r] + bias[64 + mt * 32 + row];
                    if (a.addend) {                                // uniform; columns >= L of a padded row are readable
                        const float* ad = a.addend + (size_t)b * a.addend_bstride + a.addend_pad + n0 + wn * 64 + nt * 32 + l31;
                        const int c = min(mt * 32 + row, a.pairC - 1);
                        u0 += ad[(size_t)c * a.addend_ld];
                        u1 += ad[(size_t)(a.pairC + c) * a.addend_ld];
                    }
                    actv[mt][nt][r] = ok ? fast_tanh(u0) * fast_sigmoid(u1) : 0.0f;
                }
        __syncthreads();                                   // everyone is done with the bias copy in LDS
        // 2. res/skip weights (transposed, [channel][128 rows]) and bias -> LDS
        for (int i = t * 4; i < 64 * 128; i += 1024)
            *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(a.rs_wT + i);
        if (t < 128) lds[64 * 128 + t] = a.rs_bias[t];
        __syncthreads();
        // 3. second GEMM, wave-local: B operand = the activation registers.  The MFMA k index is free to be any
        // permutation of the channels as long as A agrees: k-step s pairs the channel each half-wave already holds
        // in accumulator register s of its C/D layout, so no cross-lane movement is needed.
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int r = s & 15;
            const int ch = (s >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            float a2[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a2[mt] = lds[ch * 128 + mt * 32 + l31];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[mt], actv[s >> 4][nt][r], acc[mt][nt], 0, 0, 0);
        }
        // 4. residual / skip epilogue
        split_epilogue<GEMM_EPI_SPLIT>(a, acc, lds + 64 * 128, a.rs_rows, /*row0=*/0, b, n0 + wn * 64, l31, lhi);
    } else {
        split_epilogue<EPI>(a, acc, bias, a.M, mb * BM + wm * 128, b, n0 + wn * 64, l31, lhi);
    }
}

// CTTS_GEMM_DEFAULT in a config struct = fp32 MFMA.  (Until ABI 5 a process-wide default could be set; it is gone: the
// mode travels in the config structs only.)
inline int gemm_f32_mode() { return CTTS_GEMM_F32; }

template <int EPI, int XS>
void launch_shape_xs(int bm, dim3 grid, hipStream_t stream, const GemmArgs& a) {
    const bool few = a.nseg <= 4;
    if (bm == 128) {
        if (few) hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 1, 4, true, XS>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 1, GEMM_MAX_SEG, true, XS>), grid, dim3(256), 0, stream, a);
    } else {
        if (few) hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 2, 4, true, XS>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 2, GEMM_MAX_SEG, true, XS>), grid, dim3(256), 0, stream, a);
    }
}

template <int EPI, bool GLDS>
void launch_shape_g(int bm, dim3 grid, hipStream_t stream, const GemmArgs& a) {
    const bool few = a.nseg <= 4;
    if (bm == 128) {
        if (few) hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 1, 4, GLDS>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 1, GEMM_MAX_SEG, GLDS>), grid, dim3(256), 0, stream, a);
    } else {
        if (few) hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 2, 4, GLDS>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv_gemm_f32_kernel<EPI, 2, GEMM_MAX_SEG, GLDS>), grid, dim3(256), 0, stream, a);
    }
}

template <int EPI>
void launch_shape(int bm, dim3 grid, hipStream_t stream, const GemmArgs& a) {
    if (tuning().f32_no_glds || a.nch_total > GEMM_GLDS_MAX_CHUNKS) launch_shape_g<EPI, false>(bm, grid, stream, a);
    else if (gemm_split_level(a.gemm_mode) == 6) launch_shape_xs<EPI, 6>(bm, grid, stream, a);
    else if (gemm_split_level(a.gemm_mode) == 3) launch_shape_xs<EPI, 3>(bm, grid, stream, a);
    else launch_shape_g<EPI, true>(bm, grid, stream, a);
}

}  // namespace

int get_gemm_f32_mode() { return gemm_f32_mode(); }
namespace { thread_local int t_last_loop = 0; }
void note_gemm_loop(int code) { t_last_loop = code; }
int last_gemm_loop() { return t_last_loop; }
int gemm_split_level(int m) {
    if (m == CTTS_GEMM_DEFAULT) m = gemm_f32_mode();
    if (m == CTTS_GEMM_BF16X3) return 3;
    if (m == CTTS_GEMM_BF16X6) return 6;
    return 0;
}
bool gemm_mode_is_split(int m) { return gemm_split_level(m) != 0; }

namespace {
std::mutex g_tune_mu;
Tuning g_tune{};
bool g_tune_loaded = false;
void load_tuning_locked() {
    auto on = [](const char* n) { return getenv(n) != nullptr; };
    auto num = [](const char* n, int d) { const char* v = getenv(n); return v ? atoi(v) : d; };
    g_tune.f32_no_glds = on("CTTS_F32_NO_GLDS");
    g_tune.f32_no_small = on("CTTS_F32_NO_SMALL");
    g_tune.f32_force_small = on("CTTS_F32_FORCE_SMALL");
    g_tune.f32_no_splitk = on("CTTS_F32_NO_SPLITK");
    g_tune.f32_splitk_w4 = on("CTTS_F32_SPLITK_W4");
    g_tune.f32_no_round_split = on("CTTS_F32_NO_ROUND_SPLIT");
    g_tune.no_xcd_pair = on("CTTS_GEMM_NO_XCD_PAIR");
    g_tune.bf16_no_glds = on("CTTS_BF16_NO_GLDS");
    g_tune.bf16_no_wide = on("CTTS_BF16_NO_WIDE");
    g_tune.bf16_no_pp = on("CTTS_BF16_NO_PP");
    // (512 until round 5.  One utterance of config 2 is 452 wide tiles: 17.7 -> 15.7 ms per call in the half mode with the wide block;
    //  450 frames = 228 tiles 11.0 -> 10.4; at 225 frames = 116 tiles the narrow block is ahead, 8.8 vs 9.2: profiles/r5_61)
    g_tune.bf16_wide_min = num("CTTS_BF16_WIDE_MIN", 192);
    g_tune.bf16_w4 = on("CTTS_BF16_W4");
    g_tune.bf16_pp_stages = num("CTTS_BF16_PP_STAGES", 3);
    g_tune.bf16_map = num("CTTS_BF16_MAP", 0);
    g_tune.bf16_ps = on("CTTS_BF16_PS");
    g_tune.bf16_no_ps = on("CTTS_BF16_NO_PS");
    g_tune.bf16_ps_stages = num("CTTS_BF16_PS_STAGES", 4) == 3 ? 3 : 4;
    g_tune.wf_no_fuse = on("CTTS_WF_NO_FUSE");
    g_tune.taco_no_fuse = on("CTTS_TACO_NO_FUSE");
    g_tune.taco_valu = on("CTTS_TACO_VALU");
    g_tune.up_no_mfma = on("CTTS_UP_NO_MFMA");
    g_tune.taco_bg_no_pipe = on("CTTS_TACO_BG_NO_PIPE");
    { const char* e = getenv("CTTS_TACO_BG_SHAPE"); g_tune.taco_bg_shape = e ? atoi(e) : 0; }
    { const char* e = getenv("CTTS_TACO_BG_DEBUG"); g_tune.taco_bg_debug = e ? atoi(e) : 0; }
    {
        // measured: profiles/r5_59_taco_poll_delay.txt, r5_62 (0x10000 = straight to the full sweep; ctx: 128 is 0.3 us faster still but
        // 144 is already behind - the attention workgroups' answer must not beat the delay - so it stays a quarter below that edge)
        static const int dflt[6] = {0x10000 | 36, 0x10000 | 96, 0x10000 | 12, 0x10000 | 20, 0x10000 | 8, 0x10000 | 8};   // (h1 / prenet: 24 / 28 until the early products of phase D moved in front of their gathers, r5_68)
        int v[6];
        for (int i = 0; i < 6; ++i) v[i] = dflt[i];
        const char* e = getenv("CTTS_TACO_POLL_DELAY");
        if (e && sscanf(e, "%d,%d,%d,%d,%d,%d", v, v + 1, v + 2, v + 3, v + 4, v + 5) != 6)
            for (int i = 0; i < 6; ++i) v[i] = dflt[i];
        for (int i = 0; i < 6; ++i) g_tune.taco_poll_delay[i] = v[i] < 0 ? 0 : v[i];
        g_tune.taco_poll_delay_set = e != nullptr;
    }
    g_tune.wf_no_vec_interp = on("CTTS_WF_NO_VEC_INTERP");
    g_tune.wf_no_region_split = on("CTTS_WF_NO_REGION_SPLIT");
    g_tune.wf_no_row_queue = on("CTTS_WF_NO_ROW_QUEUE");
    g_tune.wf_row_queue_min = num("CTTS_WF_ROW_QUEUE_MIN", -1);
    g_tune.wf_inject_abort = num("CTTS_WF_INJECT_ABORT", 0);
    g_tune.wf_queue_debug = num("CTTS_WF_QUEUE_DEBUG", 0);
    g_tune.f32_no_defer_skip = on("CTTS_F32_NO_DEFER_SKIP");
    g_tune.w4_debug = num("CTTS_BF16_W4_DEBUG", 0);
    g_tune_loaded = true;
}
}  // namespace

Tuning tuning() {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (!g_tune_loaded) load_tuning_locked();
    return g_tune;
}
void reload_tuning() {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    load_tuning_locked();
}

// every argument check that does not depend on the launch shape (a: defaults applied)
int gemm_check_args(int epi, const GemmArgs& a) {
    CTTS_CHECK_ARG(a.bm == 256 || a.bm == 128, "gemm: bm=%d", a.bm);
    CTTS_CHECK_ARG(gemm_mode_valid(a.gemm_mode), "gemm: f32_gemm_mode %d (0 default, 1 fp32 MFMA, 2 split bf16)", a.gemm_mode);
    const int bn = gemm_bn(a.bm);
    CTTS_CHECK_ARG(a.nseg >= 1 && a.nseg <= GEMM_MAX_SEG, "gemm: nseg=%d", a.nseg);
    int nch = 0;
    for (int s = 0; s < a.nseg; ++s) {
        CTTS_CHECK_ARG(a.seg[s].nch > 0 && a.seg[s].base, "gemm: empty segment %d", s);
        CTTS_CHECK_ARG(a.seg[s].shift >= -a.pad && a.seg[s].shift <= a.pad,
                       "gemm: shift %d exceeds halo %d", a.seg[s].shift, a.pad);
        nch += a.seg[s].nch;
    }
    CTTS_CHECK_ARG(nch == a.nch_total, "gemm: chunk count mismatch %d vs %d", nch, a.nch_total);
    if (a.interleave > 1) {
        CTTS_CHECK_ARG(a.interleave <= a.nseg, "gemm: interleave %d > nseg %d", a.interleave, a.nseg);
        for (int s = 1; s < a.interleave; ++s)
            CTTS_CHECK_ARG(a.seg[s].nch == a.seg[0].nch, "gemm: interleaved segments must have equal length");
    }
    CTTS_CHECK_ARG(a.a_nch_alloc == 0 || a.a_ch_off + a.nch_total <= a.a_nch_alloc, "gemm: A chunk window");
    CTTS_CHECK_ARG(a.ld % 4 == 0 && a.ntiles * bn + 2 * a.pad <= a.ld && a.L <= a.ntiles * bn,
                   "gemm: bad geometry ld=%d pad=%d L=%d ntiles=%d bn=%d", a.ld, a.pad, a.L, a.ntiles, bn);
    CTTS_CHECK_ARG(gemm_epi_is_pair(epi) || a.split % 32 == 0, "gemm: split %d not a multiple of 32", a.split);
    const int cpb = a.bm / 2;   // pair channels per M-block
    CTTS_CHECK_ARG(gemm_epi_is_pair(epi) ? (a.pairC > (a.MB - 1) * cpb && a.pairC <= a.MB * cpb)
                                         : (a.M > (a.MB - 1) * a.bm && a.M <= a.MB * a.bm),
                   "gemm: M=%d pairC=%d MB=%d bm=%d", a.M, a.pairC, a.MB, a.bm);
    CTTS_CHECK_ARG(a.dst_ld > 0 && a.dst0, "gemm: destination not set");
    CTTS_CHECK_ARG(a.addend_frames == 0 || ((epi == GEMM_EPI_GATE || epi == GEMM_EPI_GATEX) && a.addend && a.addend_frames <= a.addend_ld - a.addend_pad),
                   "gemm: interpolated addend needs the GATE epilogue (frames=%d)", a.addend_frames);
    CTTS_CHECK_ARG(a.gate >= 0 && a.gate < GATE_KINDS && (a.gate == 0 || epi == GEMM_EPI_GATE), "gemm: gate=%d with epilogue %d",
                   a.gate, epi);
    // the fused res/skip epilogue is validated BEFORE any shape is chosen: the small and split-K shapes (and the row queue's tile
    // bodies) take the same arguments
    CTTS_CHECK_ARG(epi != GEMM_EPI_GATE_RS ||
                       (a.bm == 128 && a.pairC <= 64 && a.MB == 1 && a.rs_wT && a.rs_bias && (a.rs_rows == 64 || a.rs_rows == 128)),
                   "gemm: fused res/skip needs bm=128, <= 64 channels, rs_wT / rs_bias and 64 or 128 res/skip rows");
    return CTTS_OK;
}

int launch_gemm_f32(int epi, const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    gemm_apply_defaults(a);
    if (int rc = gemm_check_args(epi, a)) return rc;
    if (epi == GEMM_EPI_GATE && a.gate != GATE_GTU) epi = GEMM_EPI_GATEX;
    if (gemm_f32_small_applies(epi, a)) return launch_gemm_f32_small(epi, a, stream);
    long long tiles = (long long)a.ntiles * a.batch;         // column tiles of the launch, tile + ntiles * batch item
    // Round-aligned launch.  Two workgroups share a CU, so a launch runs in rounds of 2 x CUs workgroups; one whose count is a
    // little above a whole number of rounds ends with a round that a few CUs run alone.  Measured on the headline's in-layer
    // launch (scripts/micro/headline_gemm.hip): 7168 workgroups = 14 rounds 5.99 ms, 7200 = 14.06 rounds 6.20 ms - 0.21 ms for
    // 0.45 % more work.  The tiles beyond the last whole round (they lie at the end of the last batch item) go to a launch of
    // the small-problem shape FIRST - a quarter of the tile per workgroup, so they spread over every CU; same packed operands,
    // same K order: bit-identical - and this launch covers the rest.  CTTS_F32_NO_ROUND_SPLIT = one launch.
    if (a.bm == 256 && epi != GEMM_EPI_GATE_RS && a.addend_frames == 0 && a.shape_blocks == 0 && !tuning().f32_no_round_split &&
        !tuning().f32_no_glds && !tuning().f32_no_small && a.nch_total <= GEMM_GLDS_MAX_CHUNKS) {
        const long long slots = 2ll * wf_row_cus();
        const long long rounds = a.MB * tiles / slots;
        const long long rem_tiles = (a.MB * tiles - rounds * slots + a.MB - 1) / a.MB;
        // (the peeled tiles must hold valid columns: a caller may over-provision ntiles, L <= (ntiles - rem_tiles) * bn)
        if (rounds >= 2 && rem_tiles > 0 && rem_tiles * a.MB <= slots * 3 / 10 && rem_tiles < a.ntiles &&
            (long long)(a.ntiles - rem_tiles) * gemm_bn(a.bm) < a.L) {
            GemmArgs r = a;                                    // the last rem_tiles column tiles of the last batch item
            const int bn = gemm_bn(a.bm);
            const long long co = (long long)(a.ntiles - rem_tiles) * bn;
            const long long bo = a.batch - 1;
            for (int j = 0; j < r.nseg; ++j) r.seg[j].base += bo * r.seg[j].bstride + co;
            if (r.dst0) r.dst0 += bo * r.dst0_bstride + co;
            if (r.dst1) r.dst1 += bo * r.dst1_bstride + co;
            if (r.src0) r.src0 += bo * r.src0_bstride + co;
            if (r.addend) r.addend += bo * r.addend_bstride + co;
            r.L = a.L - (int)co;
            r.ntiles = (int)rem_tiles;
            r.batch = 1;
            if (int rc = launch_gemm_f32_small(epi, r, stream)) return rc;
            tiles -= rem_tiles;
        }
    }
    a.gt_limit = (int)tiles;
    long long blocks = (long long)a.MB * tiles;
    a.map_mode = 0;
    // measured on config 2 (PMC FETCH_SIZE per in-layer launch): 4.2 GB -> 2.5 GB at unchanged speed
    if (a.MB == 4 && epi != GEMM_EPI_GATE_RS && !tuning().no_xcd_pair) {
        a.map_mode = 1;
        blocks = 16ll * ((tiles + 3) / 4);
    }
    CTTS_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "gemm: grid %lld", blocks);
    dim3 grid((unsigned)blocks);
    note_gemm_loop((tuning().f32_no_glds || a.nch_total > GEMM_GLDS_MAX_CHUNKS) ? 0 : gemm_split_level(a.gemm_mode));
    switch (epi) {
        case GEMM_EPI_GATEX: launch_shape<GEMM_EPI_GATEX>(a.bm, grid, stream, a); break;
        case GEMM_EPI_GATE: launch_shape<GEMM_EPI_GATE>(a.bm, grid, stream, a); break;
        case GEMM_EPI_GATE_RS:
            if (tuning().f32_no_glds || a.nch_total > GEMM_GLDS_MAX_CHUNKS) {
                if (a.nseg <= 4) hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, 4, false>), grid, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, GEMM_MAX_SEG, false>), grid, dim3(256), 0, stream, a);
            } else if (gemm_split_level(a.gemm_mode) == 6) {
                if (a.nseg <= 4) hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, 4, true, 6>), grid, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, GEMM_MAX_SEG, true, 6>), grid, dim3(256), 0, stream, a);
            } else if (gemm_split_level(a.gemm_mode) == 3) {
                if (a.nseg <= 4) hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, 4, true, 3>), grid, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, GEMM_MAX_SEG, true, 3>), grid, dim3(256), 0, stream, a);
            } else {
                if (a.nseg <= 4) hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, 4, true>), grid, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((conv_gemm_f32_kernel<GEMM_EPI_GATE_RS, 1, GEMM_MAX_SEG, true>), grid, dim3(256), 0, stream, a);
            }
            break;
        case GEMM_EPI_MAG: launch_shape<GEMM_EPI_MAG>(a.bm, grid, stream, a); break;
        case GEMM_EPI_LOG: launch_shape<GEMM_EPI_LOG>(a.bm, grid, stream, a); break;
        case GEMM_EPI_LRELU: launch_shape<GEMM_EPI_LRELU>(a.bm, grid, stream, a); break;
        case GEMM_EPI_TANH: launch_shape<GEMM_EPI_TANH>(a.bm, grid, stream, a); break;
        case GEMM_EPI_SPLIT: launch_shape<GEMM_EPI_SPLIT>(a.bm, grid, stream, a); break;
        default: set_error("gemm: unknown epilogue %d", epi); return CTTS_E_ARG;
    }
    CTTS_CHECK_LAUNCH("conv_gemm_f32");
    return CTTS_OK;
}

}  // namespace ctts
