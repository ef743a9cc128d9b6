// Device code of the bf16 MFMA conv-GEMM (contract: gemm_bf16.h; launchers: gemm_bf16.hip).  A header so that the
// stand-alone harness scripts/micro/bf16_gemm_harness.hip launches exactly the product kernels on synthetic arguments.
//
// bf16 MFMA conv-GEMM for gfx950 (contract: gemm_bf16.h).
//
// 256 threads = 4 waves as 2(M) x 2(N); block tile 256 x 128, K chunk 32; wave tile 128 x 64 =
// 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator VGPRs).  LDS stage = A [4][256] + B [4][128]
// 16-byte units (24 KiB), double buffered; every fragment is one ds_read_b128 per lane with consecutive
// lanes on consecutive units (conflict-free).  Global->LDS staging through registers one chunk ahead.
#pragma once
#include <cstdlib>

#include "gemm_bf16.h"

namespace ctts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int A_UNITS = 4 * BGEMM_BM;                   // 1024 x 16 B
constexpr int BGEMM_PP_MAX_CHUNKS = 256;                // chunk address table entries (8 B each, in LDS)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte K8 unit
typedef const __attribute__((address_space(1))) u32x4* gunit_ptr;

__device__ __forceinline__ float fast_sigmoid(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
}
__device__ __forceinline__ float fast_tanh(float u) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * 2.8853900817779268f));
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// one v_mfma_f32_32x32x16 on two 16-byte K8 units: bf16 or IEEE-half operands, fp32 accumulate
template <bool F16>
__device__ __forceinline__ f32x16 mfma_k16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16 = false>
__device__ __forceinline__ unsigned int pack2(float lo, float hi) { return pack_h2<F16>(lo, hi); }
// tanh(u0) * sigmoid(u1) = (e^{2 u0} - 1) / ((e^{2 u0} + 1)(1 + e^{-u1})): three transcendentals instead of four.
// u0 is clamped to +-10 (tanh is 1 to fp32 precision there) so the numerator stays finite; a huge e^{-u1} drives
// the reciprocal to 0, which is the limit.
__device__ __forceinline__ float fast_gate(float u0, float u1) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(u0, -10.0f, 10.0f) * 2.8853900817779268f);
    const float q = __builtin_amdgcn_exp2f(u1 * -1.4426950408889634f);
    return (e - 1.0f) * __builtin_amdgcn_rcpf((e + 1.0f) * (1.0f + q));
}

// split-bf16: low halves of two values whose high halves are the packed pair `hi` (lo = bf16(v - float(hi)))
__device__ __forceinline__ unsigned int pack2_residual(float v0, float v1, unsigned int hi) {
    return pack2<false>(v0 - __builtin_bit_cast(float, hi << 16), v1 - __builtin_bit_cast(float, hi & 0xffff0000u));
}

// Half-wave exchange (v_permlane32_swap): lanes 32..63 of x swap with lanes 0..31 of y.
__device__ __forceinline__ void swap_halves(unsigned int& x, unsigned int& y) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
__device__ __forceinline__ void swap_halves(float& x, float& y) {
    unsigned int a = __builtin_bit_cast(unsigned int, x), b = __builtin_bit_cast(unsigned int, y);
    swap_halves(a, b);
    x = __builtin_bit_cast(float, a);
    y = __builtin_bit_cast(float, b);
}

// Chunk c -> global byte address of its B rows (channel group 0, column n0 + shift) for this workgroup: the segment /
// tap sequencing is resolved ONCE per workgroup into an LDS table, so the steady-state loads are a table read plus a
// per-thread constant.  Static segment indices only: a dynamically indexed kernarg goes to scratch.
// rel = the chunk's address for batch item 0, column 0; bsb = bytes between batch items (segment dependent)
__device__ __forceinline__ void chunk_rel(const BGemmArgs& a, int t, int mb, unsigned long long& rel, unsigned long long& bsb) {
    const int ilv = a.interleave > 1 ? a.interleave : 0;
    const int n_il = ilv * a.seg[0].nch;
    int c = t, s, local;
    if (c < n_il) {
        s = c % ilv;
        local = c / ilv;
    } else {
        c -= n_il;
        s = ilv;
#pragma unroll
        for (int k = 0; k < BGEMM_MAX_SEG - 1; ++k)
            if (s == k && k < a.nseg - 1 && c >= a.seg[k].nch) { c -= a.seg[k].nch; s = k + 1; }
        local = c;
    }
    static_assert(BGEMM_MAX_SEG == 12, "segment select chain below");
#define CTTS_SEGF(f)                                                                                                  \
    (s == 0 ? a.seg[0].f : s == 1 ? a.seg[1].f : s == 2 ? a.seg[2].f : s == 3 ? a.seg[3].f : s == 4 ? a.seg[4].f :    \
     s == 5 ? a.seg[5].f : s == 6 ? a.seg[6].f : s == 7 ? a.seg[7].f : s == 8 ? a.seg[8].f : s == 9 ? a.seg[9].f :    \
     s == 10 ? a.seg[10].f : a.seg[11].f)
    const bf16_t* base = CTTS_SEGF(base);
    const long long bstride = CTTS_SEGF(bstride);
    const int shift = CTTS_SEGF(shift), mbr = CTTS_SEGF(mb_rows);
#undef CTTS_SEGF
    rel = (unsigned long long)base + 16ull * ((size_t)(mb * (mbr / 8) + 4 * local) * a.ld + a.pad + shift);
    bsb = 2ull * (unsigned long long)bstride;
}
__device__ __forceinline__ void build_chunk_table(const BGemmArgs& a, unsigned long long* tab, int t, int mb, int b, int n0) {
    if (t >= a.nch_total) return;
    unsigned long long rel, bsb;
    chunk_rel(a, t, mb, rel, bsb);
    tab[t] = rel + (unsigned long long)b * bsb + 16ull * (unsigned long long)n0;
}

// Block id -> (m-block, column tile, batch item).  Workgroup ids go round-robin over the 8 XCDs (id % 8), each with a
// private L2, and all m-blocks of a column tile read the same B tile:
//   map_mode 1 (MB == 4): XCD x owns the m-block pair {2(x&1), 2(x&1)+1} of the column tiles 4q + (x>>1), the two
//     m-blocks of a tile on ids 8 apart: the B tile goes through 2 private L2s instead of 4 (and each L2 holds half of A);
//   map_mode 2 (MB == 2): XCD x owns BOTH m-blocks of the column tiles 8q + x, on ids 8 apart: the B tile of the
//     memory-bound res / skip GEMMs is fetched from HBM once instead of twice (round 5: FETCH_SIZE of these launches was
//     the B bytes twice plus the read-modify-write destination, profiles/r3_01_pmc_config3_bf16_b32.json);
//   map_mode 3 (MB == 4, experiment CTTS_BF16_MAP=2): all four m-blocks of a tile on one XCD.
// Returns false for the ids beyond the last tile (the grid is rounded up to whole groups).
__device__ __forceinline__ bool block_map(const BGemmArgs& a, int id, int& mb, int& tile, int& b) {
    int gt;
    if (a.map_mode == 1) {
        const int x = id & 7, j = id >> 3;
        mb = 2 * (x & 1) + (j & 1);
        gt = (j >> 1) * 4 + (x >> 1);
    } else if (a.map_mode == 2) {
        const int x = id & 7, j = id >> 3;
        mb = j & 1;
        gt = (j >> 1) * 8 + x;
    } else if (a.map_mode == 3) {
        const int x = id & 7, j = id >> 3;
        mb = j & 3;
        gt = (j >> 2) * 8 + x;
    } else {
        mb = id % a.MB;
        gt = id / a.MB;
    }
    if (gt >= a.ntiles * a.batch) return false;
    tile = gt % a.ntiles;
    b = gt / a.ntiles;
    return true;
}

// Epilogue shared by the block shapes.  32x32 C/D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5):
// for a fixed register group q = r>>2 the lane holds 4 consecutive channels (8q + 4*lhi + 0..3) of one column,
// i.e. half of a 16-byte K8 unit, lanes l and l+32 complete the unit.  The natural 8-byte accesses are
// issue-bound (16 per lane per tile), so groups are handled in pairs (q, q+1) with one half-wave exchange per
// dword: afterwards lane l holds all 8 channels of unit q and lane l+32 all 8 of unit q+1 for column l, and every
// global access is one 16-byte unit per lane (consecutive lanes -> consecutive units).
// STAGED: `lds` already holds this m-block's 256 bias values (the persistent kernel stages them once per workgroup and
// must not pass a barrier here: its LDS-DMA of the next tile is in flight).
template <int EPI, bool STAGED = false, bool F16 = false>
__device__ __forceinline__ void bf16_epilogue(const BGemmArgs& a, f32x16 (&acc)[4][2], u32x4* lds, int t, int mb, int wm,
                                              int wn, int b, int n0, int l31, int lhi) {
    float* bias_s = reinterpret_cast<float*>(lds);
    if constexpr (!STAGED) {
        if (t < BGEMM_BM) bias_s[t] = a.bias[mb * BGEMM_BM + t];
        __syncthreads();
    }
    const float* bias = bias_s + wm * 128;
    if constexpr (EPI == BGEMM_EPI_GATE) {
        bf16_t* dst = a.dst0 + (size_t)b * a.dst0_bstride;
        const int cbase = (mb * 2 + wm) * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (cbase + mt * 32 >= a.pairC) continue;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    unsigned int pk[2][2], pl[2][2];         // [group of the pair][dword]: high halves, low halves
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = 2 * qp + h;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int row = 8 * q + 4 * lhi + j;
                            const float u0 = acc[mt][nt][4 * q + j] + bias[mt * 32 + row];
                            const float u1 = acc[mt + 2][nt][4 * q + j] + bias[64 + mt * 32 + row];
                            v[j] = fast_gate(u0, u1);
                        }
                        pk[h][0] = pack2<F16>(v[0], v[1]);
                        pk[h][1] = pack2<F16>(v[2], v[3]);
                        if (a.lo_off) {                      // uniform
                            pl[h][0] = pack2_residual(v[0], v[1], pk[h][0]);
                            pl[h][1] = pack2_residual(v[2], v[3], pk[h][1]);
                        }
                    }
                    swap_halves(pk[0][0], pk[1][0]);
                    swap_halves(pk[0][1], pk[1][1]);
                    const int cg = (cbase + mt * 32) / 8 + 2 * qp + lhi;
                    bf16_t* du = dst + ((size_t)cg * a.ld + a.pad + n) * 8;
                    if (n < a.L) *reinterpret_cast<u32x4*>(du) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                    if (a.lo_off) {
                        swap_halves(pl[0][0], pl[1][0]);
                        swap_halves(pl[0][1], pl[1][1]);
                        if (n < a.L) *reinterpret_cast<u32x4*>(du + a.lo_off) = u32x4{pl[0][0], pl[0][1], pl[1][0], pl[1][1]};
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int rbase = mb * BGEMM_BM + wm * 128 + mt * 32;
            if (rbase >= a.M) continue;
            const bool second = rbase >= a.split;
            bf16_t* dst = second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride;
            const int accum = second ? a.acc1 : a.acc0;
            const int cg0 = (second ? rbase - a.split : rbase) / 8;
            // read-modify-write: all four 16-byte loads of the row tile are issued before the first store
            u32x4 old[2][2], oldl[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const bf16_t* du = dst + ((size_t)(cg0 + 2 * qp + lhi) * a.ld + a.pad + n0 + wn * 64 + nt * 32 + l31) * 8;
                    if (accum)   // uniform; columns >= L of a padded row are readable
                        old[nt][qp] = *reinterpret_cast<const u32x4*>(du);
                    else
                        old[nt][qp] = u32x4{0u, 0u, 0u, 0u};
                    if (accum && a.lo_off) oldl[nt][qp] = *reinterpret_cast<const u32x4*>(du + a.lo_off);
                    else oldl[nt][qp] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    float v[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            v[h][j] = acc[mt][nt][4 * (2 * qp + h) + j] + bias[mt * 32 + 8 * (2 * qp + h) + 4 * lhi + j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) swap_halves(v[0][j], v[1][j]);
                    // v[0] = channels 0..3, v[1] = channels 4..7 of this lane's unit
                    const u32x4 o = old[nt][qp], ol = oldl[nt][qp];
                    unsigned int pk[4], pl[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        // even / odd channel of the dword; the old value of a split destination is hi + lo
                        const float ev = v[d >> 1][2 * (d & 1)] +
                                         (h_to_f32<F16>((bf16_t)(o[d] & 0xffff)) + bf16_to_f32((bf16_t)(ol[d] & 0xffff)));
                        const float od = v[d >> 1][2 * (d & 1) + 1] +
                                         (h_to_f32<F16>((bf16_t)(o[d] >> 16)) + bf16_to_f32((bf16_t)(ol[d] >> 16)));
                        pk[d] = pack2<F16>(ev, od);
                        pl[d] = a.lo_off ? pack2_residual(ev, od, pk[d]) : 0u;
                    }
                    bf16_t* du = dst + ((size_t)(cg0 + 2 * qp + lhi) * a.ld + a.pad + n) * 8;
                    if (n < a.L) *reinterpret_cast<u32x4*>(du) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    if (a.lo_off && n < a.L) *reinterpret_cast<u32x4*>(du + a.lo_off) = u32x4{pl[0], pl[1], pl[2], pl[3]};
                }
            }
        }
    }
}

// NW = waves along N: 2 -> 256 threads, block tile 256 x 128; 4 -> 512 threads, block tile 256 x 256 (one
// workgroup per CU).  The wide tile stages 1/3 fewer bytes per FLOP: at bf16 MFMA rates the CU's vector-memory
// path (64 B/clk) is the co-bottleneck of the narrow tile (PMC: MFMA busy 44 %, issue-stalled 48 %).
template <int EPI, bool GLDS, int NW, bool F16 = false>
__global__ __launch_bounds__(128 * NW, 2) void conv_gemm_bf16_kernel(const BGemmArgs a) {
    constexpr int NT = 128 * NW;                            // threads
    constexpr int BN = 64 * NW;
    constexpr int B_UNITS = 4 * BN;
    constexpr int STAGE_UNITS = A_UNITS + B_UNITS;
    constexpr int NA = A_UNITS / NT;                        // 16-byte units per thread per stage (A)
    constexpr int NB_ = B_UNITS / NT;                       // (B) == 2 for both shapes
    static_assert(NB_ == 2, "B staging assumes 2 units per thread");
    constexpr int NSTAGE = GLDS ? 3 : 2;
    __shared__ __attribute__((aligned(16))) u32x4 lds[NSTAGE * STAGE_UNITS + BGEMM_PP_MAX_CHUNKS / 2];
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(lds + NSTAGE * STAGE_UNITS);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / NW, wn = wave % NW;
    const int l31 = lane & 31, lhi = lane >> 5;

    int mb, tile, b;
    if (!block_map(a, blockIdx.x, mb, tile, b)) return;     // whole workgroup
    const int n0 = tile * BN;

    // per-thread B staging: units (g, n) with g = t / BN (+2), n = t % BN, relative to the chunk's table entry
    const int bg = t / BN, bn = t % BN;
    build_chunk_table(a, tab, t, mb, b, n0);
    __syncthreads();
    const size_t boff_units = (size_t)bg * a.ld + bn;
    const size_t g2_units = (size_t)2 * a.ld;
    int ich = 0;                                            // next chunk to stage

    gunit_ptr ap = (gunit_ptr)a.A + (size_t)mb * a.nch_total * A_UNITS + t;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    u32x4 ra0, ra1, ra2, ra3, rb0, rb1;

#define CTTS_ISSUE_LOADS()                                                                      \
    do {                                                                                        \
        ra0 = ap[0]; ra1 = ap[NT];                                                              \
        if constexpr (NA > 2) { ra2 = ap[2 * NT]; ra3 = ap[3 * NT]; }                           \
        ap += A_UNITS;                                                                          \
        gunit_ptr bp = (gunit_ptr)tab[ich++] + boff_units;                                      \
        rb0 = bp[0];                                                                            \
        rb1 = bp[g2_units];                                                                     \
    } while (0)

#define CTTS_STORE_LDS(buf)                                                                     \
    do {                                                                                        \
        u32x4* As_ = lds + (buf) * STAGE_UNITS + t;                                             \
        u32x4* Bs_ = lds + (buf) * STAGE_UNITS + A_UNITS + t;                                   \
        As_[0] = ra0; As_[NT] = ra1;                                                            \
        if constexpr (NA > 2) { As_[2 * NT] = ra2; As_[3 * NT] = ra3; }                         \
        Bs_[0] = rb0; Bs_[NT] = rb1;                                                            \
    } while (0)

    // Direct global->LDS staging (global_load_lds_dwordx4): no VGPR round trip, no ds_write pass.  The LDS image
    // is lane-linear by construction (unit index == thread index + 256 j), which is what the DMA needs: the
    // destination is a wave-uniform base + lane * 16 B.
    typedef __attribute__((address_space(3))) u32x4* lds_ptr;
#define CTTS_ISSUE_GLDS(buf)                                                                    \
    do {                                                                                        \
        lds_ptr la_ = (lds_ptr)(lds + (buf) * STAGE_UNITS + (t & ~63));                         \
        __builtin_amdgcn_global_load_lds(ap, la_, 16, 0, 0);                                    \
        __builtin_amdgcn_global_load_lds(ap + NT, la_ + NT, 16, 0, 0);                          \
        if constexpr (NA > 2) {                                                                 \
            __builtin_amdgcn_global_load_lds(ap + 2 * NT, la_ + 2 * NT, 16, 0, 0);              \
            __builtin_amdgcn_global_load_lds(ap + 3 * NT, la_ + 3 * NT, 16, 0, 0);              \
        }                                                                                       \
        ap += A_UNITS;                                                                          \
        gunit_ptr bp = (gunit_ptr)tab[ich++] + boff_units;                                      \
        __builtin_amdgcn_global_load_lds(bp, la_ + A_UNITS, 16, 0, 0);                          \
        __builtin_amdgcn_global_load_lds(bp + g2_units, la_ + A_UNITS + NT, 16, 0, 0);          \
    } while (0)

    const int nch = a.nch_total;
    if constexpr (GLDS) {
        // 3 LDS stages, DMA issued TWO chunks ahead: a bf16 chunk is only ~0.5k MFMA cycles per wave, far less than
        // the loaded-memory latency, so a one-chunk prefetch leaves the matrix pipe waiting on vmcnt.  Six DMAs per
        // thread per chunk -> `vmcnt(6)` = "everything but the newest chunk has landed".  Raw s_barrier: a
        // __syncthreads() would add vmcnt(0) and drain the DMA queue.
        CTTS_ISSUE_GLDS(0);
        if (nch > 1) CTTS_ISSUE_GLDS(1);
        if (nch > 1) { if constexpr (NA > 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    } else {
        CTTS_ISSUE_LOADS();
        CTTS_STORE_LDS(0);
        __syncthreads();
    }

    int cur = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = GLDS ? ch + 2 < nch : ch + 1 < nch;
        if (more) {
            if constexpr (GLDS) { const int nb = cur >= 1 ? cur - 1 : 2; CTTS_ISSUE_GLDS(nb); }   // (cur + 2) % 3
            else CTTS_ISSUE_LOADS();
        }
        const u32x4* As = lds + cur * STAGE_UNITS + wm * 128 + l31;
        const u32x4* Bs = lds + cur * STAGE_UNITS + A_UNITS + wn * 64 + l31;
        u32x4 av[2][4], bv[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int grp = 2 * ks + lhi;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) av[ks][mt] = As[grp * BGEMM_BM + mt * 32];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bv[ks][nt] = Bs[grp * BN + nt * 32];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = mfma_k16<F16>(av[ks][mt], bv[ks][nt], acc[mt][nt]);
        // pin the LDS->MFMA pipeline: fragments of k-step 1 are read while k-step 0 runs on the matrix pipe
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if constexpr (GLDS) {
            if (more) { if constexpr (NA > 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }   // chunk ch+1 landed, ch+2 in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == 2 ? 0 : cur + 1;
        } else {
            if (more) CTTS_STORE_LDS(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
#undef CTTS_ISSUE_LOADS
#undef CTTS_STORE_LDS
#undef CTTS_ISSUE_GLDS

    bf16_epilogue<EPI, false, F16>(a, acc, lds, t, mb, wm, wn, b, n0, l31, lhi);
}

// Skewed ("ping-pong") form of the 256 x 256 block (512 threads = 8 waves, one workgroup per CU, two waves per
// SIMD).  ONE barrier per K chunk; inside a barrier interval the waves of rows 0..127 (waves 0-3, one per SIMD) run
// [LOAD k | COMPUTE k] while the waves of rows 128..255 (waves 4-7) run [COMPUTE k-1 | LOAD k]: each SIMD always
// has one wave issuing its 16 MFMAs (512 matrix-pipe cycles) out of registers while its partner reads its 12
// fragments from LDS and issues its share of the DMA for chunk k+2.  Measured with s_memtime stamps: LOAD ~570 and
// COMPUTE ~520 cycles, a barrier release ~100; the barrier-per-phase form spent 1640 cycles per chunk, the
// barrier-per-chunk form with both halves in phase 2400.
// LDS: NS stages x 32 KiB (NS - 1 chunks of DMA in flight).  Both halves read chunk k inside interval k; the DMA
// issued in interval k (chunk k+NS-1) rewrites the buffer of chunk k-1, read by both before the closing barrier of
// interval k-1; a thread passes that barrier only after its own DMAs of chunk k have landed (counted vmcnt).
template <int EPI, int NS, bool F16 = false>
__global__ __launch_bounds__(512, 2) void conv_gemm_bf16_pp_kernel(const BGemmArgs a) {
    constexpr int NT = 512, BN = 256;
    constexpr int B_UNITS = 4 * BN;
    constexpr int STAGE_UNITS = A_UNITS + B_UNITS;
    // NS stages + the chunk address table (8 B per chunk, <= BGEMM_PP_MAX_CHUNKS)
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * STAGE_UNITS + BGEMM_PP_MAX_CHUNKS / 2];
    typedef unsigned long long u64;
    u64* tab = reinterpret_cast<u64*>(lds + NS * STAGE_UNITS);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform -> SALU address math, scalar branches
    const int wm = wave >> 2, wn = wave & 3;                // wm = which half (leading 0 / lagging 1)
    const int l31 = lane & 31, lhi = lane >> 5;

    int mb, tile, b;
    if (!block_map(a, blockIdx.x, mb, tile, b)) return;     // whole workgroup
    const int n0 = tile * BN;
    const int nch = a.nch_total;

    build_chunk_table(a, tab, t, mb, b, n0);
    // per-thread byte offsets inside a chunk: B unit (g, n) with g = t / 256 (+2), n = t % 256; A unit t (+512)
    const unsigned boff0 = (unsigned)(((t >> 8) * a.ld + (t & 255)) * 16);
    const unsigned boff1 = boff0 + (unsigned)(2 * a.ld * 16);
    const unsigned aoff0 = (unsigned)(t * 16), aoff1 = aoff0 + NT * 16;
    typedef const __attribute__((address_space(1))) char* gbyte_ptr;
    const gbyte_ptr abase = (gbyte_ptr)a.A + (size_t)mb * nch * (A_UNITS * 16);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    typedef __attribute__((address_space(3))) u32x4* lds_ptr;
    // chunk c -> LDS stage buf; ub = wave-uniform address of the chunk's B rows
#define CTTS_PP_DMA(buf, c, ub)                                                                 \
    do {                                                                                        \
        lds_ptr la_ = (lds_ptr)(lds + (buf) * STAGE_UNITS + wave * 64);                         \
        const gbyte_ptr ac_ = abase + (size_t)(c) * (A_UNITS * 16);                             \
        unsigned a0_ = aoff0, a1_ = aoff1, b0_ = boff0, b1_ = boff1;   /* SGPR base + 32-bit lane offset: see the persistent kernel */ \
        asm volatile("" : "+v"(a0_), "+v"(a1_), "+v"(b0_), "+v"(b1_));                         \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(ac_ + a0_), la_, 16, 0, 0);                \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(ac_ + a1_), la_ + NT, 16, 0, 0);           \
        const gbyte_ptr bc_ = (gbyte_ptr)(ub);                                                  \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(bc_ + b0_), la_ + A_UNITS, 16, 0, 0);      \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(bc_ + b1_), la_ + A_UNITS + NT, 16, 0, 0); \
    } while (0)
#define CTTS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define CTTS_UNIFORM64(v) \
    (((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)((v) >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(v)))

    __syncthreads();                                        // table visible
    // prologue: chunks 0 .. NS-2 in flight, chunk 0 landed.  Every wait below is "all but the newest N DMAs":
    // 4 DMAs per thread per chunk, in order.
    u64 ub;
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nch) { ub = CTTS_UNIFORM64(tab[c]); CTTS_PP_DMA(c, c, ub); }
    if (nch >= NS - 1) CTTS_WAIT_VM(4 * (NS - 2));
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ub = CTTS_UNIFORM64(tab[NS - 1]);                       // (entries >= nch are never used)
    __builtin_amdgcn_s_barrier();                           // chunk 0 is in LDS
    if (wm) __builtin_amdgcn_s_setprio(1);                  // the later-dispatched half loses every arbitration otherwise
    __builtin_amdgcn_sched_barrier(0);

    u32x4 av[2][4], bv[2][2];
    // fragments of this wave's 128 x 64 tile of chunk ch (stage cur) -> registers
#define CTTS_LOAD_FRAGS()                                                                       \
    do {                                                                                        \
        const u32x4* As = lds + cur * STAGE_UNITS + wm * 128 + l31;                             \
        const u32x4* Bs = lds + cur * STAGE_UNITS + A_UNITS + wn * 64 + l31;                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                      \
            const int grp = 2 * ks + lhi;                                                       \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) av[ks][mt] = As[grp * BGEMM_BM + mt * 32]; \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) bv[ks][nt] = Bs[grp * BN + nt * 32];      \
        }                                                                                       \
    } while (0)
#define CTTS_MFMA16()                                                                           \
    do {                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                    \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                \
                    acc[mt][nt] = mfma_k16<F16>(av[ks][mt], bv[ks][nt], acc[mt][nt]);          \
    } while (0)
    // DMA of chunk ch+NS-1 into the buffer of chunk ch-1, then: own DMAs of chunk ch+1 landed
#define CTTS_DMA_AND_WAIT()                                                                     \
    do {                                                                                        \
        if (ch + NS - 1 < nch) {                                                                \
            const int nb = cur >= 1 ? cur - 1 : NS - 1;     /* (cur + NS - 1) % NS */           \
            CTTS_PP_DMA(nb, ch + NS - 1, ub);                                                   \
        }                                                                                       \
    } while (0)
#define CTTS_WAIT_NEXT_CHUNK()                                                                  \
    do {                                                                                        \
        if (ch + NS - 1 < nch) CTTS_WAIT_VM(4 * (NS - 2));                                      \
        else if (NS > 3 && nch - ch - 2 == 1) CTTS_WAIT_VM(4);   /* tail: one chunk was issued after ch+1 */ \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
    } while (0)

    int cur = 0;
    if (!wm) {
        // leading half: [LOAD ch | COMPUTE ch] per interval
        for (int ch = 0; ch < nch; ++ch) {
            CTTS_LOAD_FRAGS();
            const u64 tnext = tab[ch + NS < BGEMM_PP_MAX_CHUNKS ? ch + NS : 0];
            CTTS_DMA_AND_WAIT();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            CTTS_MFMA16();
            ub = CTTS_UNIFORM64(tnext);
            __builtin_amdgcn_sched_barrier(0);
            CTTS_WAIT_NEXT_CHUNK();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == NS - 1 ? 0 : cur + 1;
        }
    } else {
        // lagging half: [COMPUTE ch-1 | LOAD ch] per interval; its last COMPUTE runs beside the leading half's epilogue
        for (int ch = 0; ch < nch; ++ch) {
            if (ch > 0) CTTS_MFMA16();
            __builtin_amdgcn_sched_barrier(0);
            CTTS_LOAD_FRAGS();
            const u64 tnext = tab[ch + NS < BGEMM_PP_MAX_CHUNKS ? ch + NS : 0];
            CTTS_DMA_AND_WAIT();
            CTTS_WAIT_NEXT_CHUNK();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this half's reads of the stage are complete
            ub = CTTS_UNIFORM64(tnext);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == NS - 1 ? 0 : cur + 1;
        }
        CTTS_MFMA16();
    }
#undef CTTS_LOAD_FRAGS
#undef CTTS_MFMA16
#undef CTTS_DMA_AND_WAIT
#undef CTTS_WAIT_NEXT_CHUNK
#undef CTTS_PP_DMA
#undef CTTS_UNIFORM64
#undef CTTS_WAIT_VM
    if (wm) __builtin_amdgcn_s_setprio(0);
    bf16_epilogue<EPI, false, F16>(a, acc, lds, t, mb, wm, wn, b, n0, l31, lhi);
}

// Persistent form of the skewed 256 x 256 block (round 5).  The per-tile kernel above pays, per workgroup and with nothing
// else resident on the CU (one workgroup of 100 KiB LDS per CU): the first DMA latency (~4 k cycles), the drain of the last
// chunks and an epilogue, ~18 k of its ~84 k cycles per tile on config 3's in-layer launch (profiles/r5_08).  Here ONE
// workgroup per CU walks a sequence of column tiles of one m-block (XCD-aware: the m-blocks of a column tile run on one XCD
// pair / one XCD at the same time) and the chunk stream never stops at a tile boundary:
//   * intervals are those of the skewed kernel, numbered over the whole stream: leading half [LOAD v | COMPUTE v], lagging
//     half [COMPUTE v-1 | LOAD v], one barrier; the DMA of chunk v + NS - 1 is issued in interval v whichever tile it
//     belongs to, so the next tile's first chunks land while this tile's epilogue runs;
//   * a half runs its epilogue inside the interval that follows its last COMPUTE of the tile, after issuing that
//     interval's DMA: the two halves' epilogues overlap each other and the lagging half's last COMPUTE;
//   * bias staged once per workgroup (the m-block is fixed), chunk -> address tables in THREE slots (tile % 3), built one
//     tile ahead: when the leading half's DMA cursor enters tile k + 1 it writes the table of tile k + 2 over that of
//     k - 1.  Three, not two: in that same interval the lagging half still reads tile k's last entry (its cursor follows
//     one interval behind, no barrier in between), so the slot being written must be neither k's nor k + 1's; slot k - 1
//     was last read nch > NS intervals (and as many barriers) earlier, and the new table is first read nch intervals later.
// Arithmetic and summation order are those of the per-tile kernel: bit-identical results.
// (An earlier form with every wave a self-contained software-pipelined stream - fragments double-buffered per k-step, DMA
// issued between MFMAs - measured 39-41 cycles per MFMA and SIMD against this structure's 36.5: a DMA issued inside an
// MFMA stream costs ~19 matrix-pipe cycles, in a LOAD phase ~2; profiles/r5_08_bf16_harness_stream_form.txt.)
// vmcnt: a thread issues 4 DMAs per interval in stream order; "all but the newest 4 (NS - 2)" = chunk v + 1 has landed.
// The epilogue's stores are younger than the interval's DMA and only ADD to the outstanding count: the waits stay safe.
__device__ unsigned long long g_ps_stamps[32];   // DBG: [wave half][{tile cycles, epilogue cycles, 100 MHz ticks of the tile, tiles}]

template <int EPI, int NS, int DBG = 0, bool F16 = false>   // DBG (harness only): 1 = s_memtime stamps of workgroup 100, third tile
__global__ __launch_bounds__(512) void conv_gemm_bf16_ps_kernel(const BGemmArgs a) {
    constexpr int NT = 512, BN = 256, MAXC = BGEMM_PP_MAX_CHUNKS;
    constexpr int B_UNITS = 4 * BN;
    constexpr int SU = A_UNITS + B_UNITS;
    typedef unsigned long long u64;
    // stages | tabs[3][MAXC] (absolute chunk addresses of the DMA cursor's tile, slot = tile % 3) | rel[MAXC] | bsb[MAXC] | bias[256]
    static_assert(MAXC % 2 == 0, "table slots are whole 16-byte units");
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * SU + 3 * MAXC / 2 + MAXC + BGEMM_BM / 4];
    u64* tabs = reinterpret_cast<u64*>(lds + NS * SU);
    u64* rel = tabs + 3 * MAXC;
    u64* bsb = rel + MAXC;
    u32x4* bias_lds = reinterpret_cast<u32x4*>(bsb + MAXC);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;                // wm = which half (leading 0 / lagging 1)
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nch = a.nch_total;

    // this workgroup's tile sequence: m-block mb, column tiles gt = first + k * step (see block_map for the XCD layouts)
    const int G = gridDim.x, id = blockIdx.x;
    int mb, first, step;
    if (a.map_mode == 1) {
        const int x = id & 7, j = id >> 3;
        mb = 2 * (x & 1) + (j & 1);
        first = (j >> 1) * 4 + (x >> 1);
        step = 4 * (G >> 4);
    } else if (a.map_mode == 2) {
        const int x = id & 7, j = id >> 3;
        mb = j & 1;
        first = (j >> 1) * 8 + x;
        step = 8 * (G >> 4);
    } else {
        mb = id % a.MB;
        first = id / a.MB;
        step = G / a.MB;
    }
    const int T = a.ntiles * a.batch;
    if (first >= T) return;                                 // whole workgroup
    const int my_tiles = (T - first + step - 1) / step;
    const int total = my_tiles * nch;                       // chunks of this workgroup's stream

    if (t < nch) {
        u64 r, s;
        chunk_rel(a, t, mb, r, s);
        rel[t] = r;
        bsb[t] = s;
        tabs[t] = r + (u64)(first / a.ntiles) * s + 16ull * (u64)((first % a.ntiles) * BN);
        if (my_tiles > 1) {
            const int g1 = first + step;
            tabs[MAXC + t] = r + (u64)(g1 / a.ntiles) * s + 16ull * (u64)((g1 % a.ntiles) * BN);
        }
    }
    if (t < BGEMM_BM) reinterpret_cast<float*>(bias_lds)[t] = a.bias[mb * BGEMM_BM + t];

    const unsigned boff0 = (unsigned)(((t >> 8) * a.ld + (t & 255)) * 16);
    const unsigned boff1 = boff0 + (unsigned)(2 * a.ld * 16);
    const unsigned aoff0 = (unsigned)(t * 16), aoff1 = aoff0 + NT * 16;
    typedef const __attribute__((address_space(1))) char* gbyte_ptr;
    const gbyte_ptr abase = (gbyte_ptr)a.A + (size_t)mb * nch * (A_UNITS * 16);

    f32x16 acc[4][2];
#define CTTS_PS_ZERO()                                                                              \
    do {                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                            \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                        \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.0f;           \
    } while (0)
    CTTS_PS_ZERO();

    typedef __attribute__((address_space(3))) u32x4* lds_ptr;
    // The address is a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane offset: global_load_lds_dwordx4 v, s[a:b].
    // The offsets are laundered through an empty asm so that the compiler cannot fold them into loop-carried 64-bit VGPR
    // addresses (global_load_lds v[a:b], off): that form costs ~4 more matrix-pipe cycles per MFMA in this loop
    // (profiles/r5_07_bf16_mix_ceiling_dma_forms.txt: 41.0 -> 36.6 cycles per MFMA and SIMD).
    // chunk c (of the cursor's tile, B rows at ub) -> LDS stage buf
#define CTTS_PS_DMA(buf, c, ub)                                                                     \
    do {                                                                                            \
        lds_ptr la_ = (lds_ptr)(lds + (buf) * SU + wave * 64);                                      \
        const gbyte_ptr ac_ = abase + (size_t)(c) * (A_UNITS * 16);                                 \
        unsigned a0_ = aoff0, a1_ = aoff1, b0_ = boff0, b1_ = boff1;                                \
        asm volatile("" : "+v"(a0_), "+v"(a1_), "+v"(b0_), "+v"(b1_));                             \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(ac_ + a0_), la_, 16, 0, 0);                    \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(ac_ + a1_), la_ + NT, 16, 0, 0);               \
        const gbyte_ptr bc_ = (gbyte_ptr)(ub);                                                      \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(bc_ + b0_), la_ + A_UNITS, 16, 0, 0);          \
        __builtin_amdgcn_global_load_lds((gunit_ptr)(bc_ + b1_), la_ + A_UNITS + NT, 16, 0, 0);     \
    } while (0)
#define CTTS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define CTTS_UNIFORM64(v) \
    (((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)((v) >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(v)))

    __syncthreads();                                        // tables, bias visible (no DMA in flight yet)
    // prologue: chunks 0 .. NS-2 of the first tile in flight (the launcher guarantees nch > NS), chunk 0 landed
#pragma unroll
    for (int c = 0; c < NS - 1; ++c) {
        const u64 ub0 = CTTS_UNIFORM64(tabs[c]);
        CTTS_PS_DMA(c, c, ub0);
    }
    int dk = 0, dch = NS - 1, dslot = 0;                    // DMA cursor: next chunk to issue = chunk dch of tile dk (table slot dslot = dk % 3) ...
    u64 ub = CTTS_UNIFORM64(tabs[NS - 1]);                  // ... whose B rows start at ub
    CTTS_WAIT_VM(4 * (NS - 2));
    __builtin_amdgcn_s_barrier();                           // chunk 0 is in LDS
    if (wm) __builtin_amdgcn_s_setprio(1);                  // the later-dispatched half loses every arbitration otherwise
    __builtin_amdgcn_sched_barrier(0);

    u32x4 av[2][4], bv[2][2];
#define CTTS_LOAD_FRAGS()                                                                           \
    do {                                                                                            \
        const u32x4* As = lds + cur * SU + wm * 128 + l31;                                          \
        const u32x4* Bs = lds + cur * SU + A_UNITS + wn * 64 + l31;                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
            const int grp = 2 * ks + lhi;                                                           \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) av[ks][mt] = As[grp * BGEMM_BM + mt * 32]; \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) bv[ks][nt] = Bs[grp * BN + nt * 32];   \
        }                                                                                           \
    } while (0)
    // The 16 MFMAs of a chunk.  FETCH: the address of the DMA cursor's chunk is read from the table behind MFMA 1 and made
    // wave-uniform behind MFMA 11, in the shadow of the matrix pipe: nothing of it is left in a LOAD phase, which is the
    // phase the interval waits for (12 fragment reads + 4 DMA issues against the partner's 512 matrix-pipe cycles).
#define CTTS_MFMA16(FETCH)                                                                          \
    do {                                                                                            \
        u64 tnext_ = 0;                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                        \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                  \
                    acc[mt][nt] = mfma_k16<F16>(av[ks][mt], bv[ks][nt], acc[mt][nt]);              \
                    if (FETCH && ks * 8 + mt * 2 + nt == 1) {                                       \
                        __builtin_amdgcn_sched_barrier(0);                                          \
                        tnext_ = tabs[dslot * MAXC + dch];                                          \
                        __builtin_amdgcn_sched_barrier(0);                                          \
                    }                                                                               \
                    if (FETCH && ks * 8 + mt * 2 + nt == 11) {                                      \
                        __builtin_amdgcn_sched_barrier(0);                                          \
                        ub = CTTS_UNIFORM64(tnext_);                                                \
                        __builtin_amdgcn_sched_barrier(0);                                          \
                    }                                                                               \
                }                                                                                   \
    } while (0)
    // interval v's DMA (chunk v + NS - 1 of the stream) into the stage of chunk v - 1; then the cursor moves on, the table
    // of the tile after the one it enters is built, and the next chunk's address is fetched (tnext -> ub after the MFMAs)
#define CTTS_PS_ISSUE()                                                                             \
    do {                                                                                            \
        const int nb = cur >= 1 ? cur - 1 : NS - 1;        /* (cur + NS - 1) % NS */                \
        CTTS_PS_DMA(nb, dch, ub);                                                                   \
        if (++dch == nch) {                                                                         \
            dch = 0;                                                                                \
            ++dk;                                                                                   \
            dslot = dslot == 2 ? 0 : dslot + 1;                                                     \
            if (dk + 1 < my_tiles && t < nch) {                                                     \
                const int g2 = first + (dk + 1) * step;                                             \
                const int wslot = dslot == 2 ? 0 : dslot + 1;   /* (dk + 1) % 3: neither the slot entered nor the one left */ \
                tabs[wslot * MAXC + t] = rel[t] + (u64)(g2 / a.ntiles) * bsb[t] + 16ull * (u64)((g2 % a.ntiles) * BN); \
            }                                                                                       \
        }                                                                                           \
    } while (0)
    // own DMAs of chunk v + 1 landed: everything but the chunks issued after it
#define CTTS_PS_WAIT_NEXT()                                                                         \
    do {                                                                                            \
        const int after = total - 2 - v;                   /* chunks of the stream beyond v + 1 */   \
        if (after >= NS - 2) CTTS_WAIT_VM(4 * (NS - 2));                                            \
        else if (NS > 3 && after == 1) CTTS_WAIT_VM(4);                                             \
        else CTTS_WAIT_VM(0);                                                                       \
    } while (0)
#define CTTS_PS_EPILOGUE()                                                                          \
    do {                                                                                            \
        const unsigned long long e0_ = (DBG & 1) ? __builtin_readcyclecounter() : 0;                \
        /* lane ids laundered: keeps the epilogue's address arithmetic out of the chunk loop's live registers (LICM) */ \
        int l31_ = l31, lhi_ = lhi;                                                                 \
        asm volatile("" : "+v"(l31_), "+v"(lhi_));                                                  \
        bf16_epilogue<EPI, true, F16>(a, acc, bias_lds, t, mb, wm, wn, gt_done / a.ntiles, (gt_done % a.ntiles) * BN, l31_, lhi_); \
        CTTS_PS_ZERO();                                                                             \
        if ((DBG & 1) && blockIdx.x == 100 && tiles_done == 2 && lane == 0 && wn == 0) {            \
            unsigned long long* o_ = g_ps_stamps + 4 * wm;                                          \
            const unsigned long long now_ = __builtin_readcyclecounter();                           \
            o_[0] = now_ - st_tile; o_[1] = now_ - e0_; o_[2] = __builtin_amdgcn_s_memrealtime() - rt_tile; o_[3] = my_tiles; \
        }                                                                                           \
        if (DBG & 1) { st_tile = __builtin_readcyclecounter(); rt_tile = __builtin_amdgcn_s_memrealtime(); } \
        ++tiles_done;                                                                               \
    } while (0)

    static_assert(NS == 3 || NS == 4, "tail waits above");
    int cur = 0, ch = 0, gt = first, gt_done = first, tiles_done = 0;
    unsigned long long st_tile = 0, rt_tile = 0;
    if (!wm) {
        // leading half: [LOAD v | COMPUTE v]; the epilogue of a tile opens the interval after its last chunk
        for (int v = 0;; ++v) {
            const bool issue = v + NS - 1 < total;
            const bool boundary = v > 0 && ch == 0;
            if (boundary) {
                if (issue) CTTS_PS_ISSUE();
                CTTS_PS_EPILOGUE();
                if (v == total) break;
            }
            CTTS_LOAD_FRAGS();
            if (issue && !boundary) CTTS_PS_ISSUE();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            CTTS_MFMA16(true);                              // + the next interval's DMA address
            __builtin_amdgcn_sched_barrier(0);
            CTTS_PS_WAIT_NEXT();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == NS - 1 ? 0 : cur + 1;
            if (++ch == nch) { ch = 0; gt_done = gt; gt += step; }
        }
    } else {
        // lagging half: [COMPUTE v-1 | LOAD v]; its epilogue follows its last COMPUTE of the tile
        for (int v = 0;; ++v) {
            if (v > 0) CTTS_MFMA16(true);                   // + this interval's DMA address (the cursor moved in interval v - 1)
            __builtin_amdgcn_sched_barrier(0);
            const bool issue = v + NS - 1 < total;
            const bool boundary = v > 0 && ch == 0;
            if (boundary) {
                if (issue) CTTS_PS_ISSUE();
                CTTS_PS_EPILOGUE();
                if (v == total) break;
            }
            CTTS_LOAD_FRAGS();
            if (issue && !boundary) CTTS_PS_ISSUE();
            CTTS_PS_WAIT_NEXT();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this half's reads of the stage are complete
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == NS - 1 ? 0 : cur + 1;
            if (++ch == nch) { ch = 0; gt_done = gt; gt += step; }
        }
    }
    if (wm) __builtin_amdgcn_s_setprio(0);
#undef CTTS_PS_EPILOGUE
#undef CTTS_PS_WAIT_NEXT
#undef CTTS_PS_ISSUE
#undef CTTS_MFMA16
#undef CTTS_LOAD_FRAGS
#undef CTTS_PS_DMA
#undef CTTS_PS_ZERO
#undef CTTS_UNIFORM64
#undef CTTS_WAIT_VM
}

// Four-wave form of the 256 x 256 block: 256 threads = 2(M) x 2(N) waves, ONE wave per SIMD, wave tile 128 x 128 =
// 4 x 4 tiles of v_mfma_f32_32x32x16_bf16 (256 accumulator registers, the whole AGPR half of the 512-register file).
// Why: a fragment is one ds_read_b128 per lane = 1 KiB per wave, and the CU's LDS moves 128 B/clk.  The 128 x 64 wave
// tile of the kernels above reads 12 fragments per 16 MFMAs: 8 waves x 12 KiB = 96 KiB of LDS reads plus 32 KiB of
// DMA writes per K chunk = 1024 LDS cycles, exactly the 1024 matrix-pipe cycles of the chunk - both pipes would have
// to run at 100 % at once, and the measured interval is ~1950 cycles.  The 128 x 128 wave tile reads 16 fragments per
// 32 MFMAs: 64 + 32 KiB = 768 LDS cycles against the same 1024 MFMA cycles.
// Pipeline: NS = 4 stages of 32 KiB filled by global_load_lds DMA three chunks ahead; per chunk ONE barrier, placed
// between the two k-steps: [ds_read k-step 1 | 16 MFMAs of k-step 0] barrier [DMA chunk+4, ds_read k-step 0 of the
// next chunk | 16 MFMAs of k-step 1], so every fragment read has 512 matrix-pipe cycles to land and the stage of
// chunk c is free for DMA as soon as the barrier inside chunk c has passed (its k-step-1 fragments are in registers).
__device__ unsigned long long g_w4_stamps[8];   // DBG == 3: {loop cycles, loop 100 MHz ticks, epilogue cycles, prologue cycles}

template <int EPI, int DBG = 0>   // DBG bits (timing experiments only): 1 = no DMA after the prologue, 2 = no MFMA (both: wrong results), 4 = stamps
__global__ __launch_bounds__(256) void conv_gemm_bf16_w4_kernel(const BGemmArgs a) {
    constexpr int NT = 256, BN = 256, NS = 4;
    constexpr int B_UNITS = 4 * BN;
    constexpr int STAGE_UNITS = A_UNITS + B_UNITS;
    __shared__ __attribute__((aligned(16))) u32x4 lds[NS * STAGE_UNITS + BGEMM_PP_MAX_CHUNKS / 2];
    typedef unsigned long long u64;
    u64* tab = reinterpret_cast<u64*>(lds + NS * STAGE_UNITS);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    int mb, tile, b;
    if (!block_map(a, blockIdx.x, mb, tile, b)) return;
    const int n0 = tile * BN;
    const int nch = a.nch_total;

    build_chunk_table(a, tab, t, mb, b, n0);
    // per-thread byte offsets inside a chunk: B unit (g, n) = (j, t), A unit t + 256 j, j = 0..3
    const unsigned boff = (unsigned)(t * 16), bstep = (unsigned)(a.ld * 16);
    const unsigned aoff = (unsigned)(t * 16);
    typedef const __attribute__((address_space(1))) char* gbyte_ptr;
    const gbyte_ptr abase = (gbyte_ptr)a.A + (size_t)mb * nch * (A_UNITS * 16);

    f32x16 acc[2][4][2];                                    // [64-column half][mt][nt]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.0f;

    typedef __attribute__((address_space(3))) u32x4* lds_ptr;
#define CTTS_W4_DMA(buf, c, ub)                                                                     \
    do {                                                                                            \
        lds_ptr la_ = (lds_ptr)(lds + (buf) * STAGE_UNITS + wave * 64);                             \
        const gbyte_ptr ac_ = abase + (size_t)(c) * (A_UNITS * 16) + aoff;                          \
        const gbyte_ptr bc_ = (gbyte_ptr)(ub) + boff;                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                            \
            __builtin_amdgcn_global_load_lds((gunit_ptr)(ac_ + j_ * (NT * 16)), la_ + j_ * NT, 16, 0, 0); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                            \
            __builtin_amdgcn_global_load_lds((gunit_ptr)(bc_ + j_ * bstep), la_ + A_UNITS + j_ * NT, 16, 0, 0); \
    } while (0)
#define CTTS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define CTTS_UNIFORM64(v) \
    (((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)((v) >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(v)))
    // wait until at most `k` chunks (8 DMAs each, in order) of this thread's DMAs are still in flight
#define CTTS_W4_WAIT_CHUNKS(k)                                                                      \
    do {                                                                                            \
        if ((k) >= 3) CTTS_WAIT_VM(24);                                                             \
        else if ((k) == 2) CTTS_WAIT_VM(16);                                                        \
        else if ((k) == 1) CTTS_WAIT_VM(8);                                                         \
        else CTTS_WAIT_VM(0);                                                                       \
    } while (0)

    const unsigned long long st0 = (DBG & 4) ? __builtin_readcyclecounter() : 0;
    __syncthreads();                                        // table visible
    {
        const int npro = nch < NS ? nch : NS;
        for (int c = 0; c < npro; ++c) {
            const u64 ub = CTTS_UNIFORM64(tab[c]);
            CTTS_W4_DMA(c, c, ub);
        }
        CTTS_W4_WAIT_CHUNKS(npro - 1);                      // chunk 0 landed (own DMAs), the rest in flight
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    u32x4 fa[2][4], fb[2][4];                               // [k-step][tile] fragments
    // With one wave per SIMD nothing else fills the matrix pipe while this wave issues other instructions: every
    // ds_read / DMA is issued in the 32-cycle shadow of ONE MFMA (order pinned by sched_barrier), never in a burst.
#define CTTS_W4_FRAG1(ks, i, buf)                                                                   \
    do {                                                                                            \
        if ((i) < 4) fa[ks][(i)] = lds[(buf) * STAGE_UNITS + (2 * (ks) + lhi) * BGEMM_BM + wm * 128 + l31 + (i) * 32]; \
        else fb[ks][(i) - 4] = lds[(buf) * STAGE_UNITS + A_UNITS + (2 * (ks) + lhi) * BN + wn * 128 + l31 + ((i) - 4) * 32]; \
    } while (0)
#define CTTS_W4_MFMA1(ks, mt, nt)                                                                   \
    do {                                                                                            \
        if ((DBG & 2)) acc[(nt) >> 1][mt][(nt) & 1][0] += __builtin_bit_cast(float, fa[ks][mt][0] ^ fb[ks][nt][0]); \
        else acc[(nt) >> 1][mt][(nt) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                \
            *reinterpret_cast<const bf16x8*>(&fa[ks][mt]),                                          \
            *reinterpret_cast<const bf16x8*>(&fb[ks][nt]), acc[(nt) >> 1][mt][(nt) & 1], 0, 0, 0);  \
    } while (0)
    // SGPR base + 32-bit lane offset per piece (see the skewed kernel's DMA)
    unsigned w4_lane[8];
#pragma unroll
    for (int j_ = 0; j_ < 8; ++j_) w4_lane[j_] = j_ < 4 ? aoff + (unsigned)(j_ * (NT * 16)) : boff + (unsigned)(j_ - 4) * bstep;
#define CTTS_W4_DMA1(buf, c, ub, j)                                                                 \
    do {                                                                                            \
        lds_ptr la_ = (lds_ptr)(lds + (buf) * STAGE_UNITS + wave * 64);                             \
        unsigned o_ = w4_lane[(j)];                                                                 \
        asm volatile("" : "+v"(o_));                                                                \
        if ((j) < 4)                                                                                \
            __builtin_amdgcn_global_load_lds((gunit_ptr)(abase + (size_t)(c) * (A_UNITS * 16) + o_), la_ + (j) * NT, 16, 0, 0); \
        else                                                                                        \
            __builtin_amdgcn_global_load_lds((gunit_ptr)((gbyte_ptr)(ub) + o_), la_ + A_UNITS + ((j) - 4) * NT, 16, 0, 0); \
    } while (0)
#define CTTS_SB() __builtin_amdgcn_sched_barrier(0)

#pragma unroll
    for (int i = 0; i < 8; ++i) CTTS_W4_FRAG1(0, i, 0);
    int cur = 0;
    const unsigned long long st1 = (DBG & 4) ? __builtin_readcyclecounter() : 0;
    const unsigned long long rt1 = (DBG & 4) ? __builtin_amdgcn_s_memrealtime() : 0;
    // One chunk.  STEADY: chunk ch + NS exists (DMA issued, constant vmcnt); otherwise the tail (no DMA, draining waits).
#define CTTS_W4_CHUNK(STEADY)                                                                       \
    do {                                                                                            \
        const int nxt = cur == NS - 1 ? 0 : cur + 1;                                                \
        /* k-step 0: MFMA m (row-major over the 4 x 4 tiles); k-step-1 fragment i is read behind MFMA i + 1 */ \
        u64 tnext = 0;                                                                              \
        _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                            \
            CTTS_W4_MFMA1(0, m >> 2, m & 3);                                                        \
            CTTS_SB();                                                                              \
            if (m >= 1 && m <= 8) CTTS_W4_FRAG1(1, m - 1, cur);                                     \
            if (STEADY && m == 9) tnext = tab[ch + NS];                                             \
            CTTS_SB();                                                                              \
        }                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* k-step-1 fragments, table entry */    \
        if ((DBG & 1)) CTTS_WAIT_VM(0);                                                              \
        else if (STEADY) CTTS_WAIT_VM(8 * (NS - 2));       /* own DMAs of chunk ch+1 landed */      \
        else CTTS_W4_WAIT_CHUNKS(nch - ch - 2);                                                     \
        __builtin_amdgcn_s_barrier();                      /* chunk ch+1 visible; stage `cur` is free */ \
        CTTS_SB();                                                                                  \
        /* k-step 1: next chunk's k-step-0 fragments behind MFMAs 1..8, the DMA of chunk ch + NS behind 8..15 */ \
        const bool more = STEADY || ch + 1 < nch;                                                   \
        const u64 ub = STEADY ? CTTS_UNIFORM64(tnext) : 0;                                          \
        _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                            \
            CTTS_W4_MFMA1(1, m >> 2, m & 3);                                                        \
            CTTS_SB();                                                                              \
            if (m >= 1 && m <= 8 && more) CTTS_W4_FRAG1(0, m - 1, nxt);                             \
            if (STEADY && !(DBG & 1) && m >= 8) CTTS_W4_DMA1(cur, ch + NS, ub, m - 8);                \
            CTTS_SB();                                                                              \
        }                                                                                           \
        cur = nxt;                                                                                  \
    } while (0)
    int ch = 0;
    for (; ch + NS < nch; ++ch) CTTS_W4_CHUNK(true);
    for (; ch < nch; ++ch) CTTS_W4_CHUNK(false);
#undef CTTS_W4_CHUNK
#undef CTTS_W4_FRAG1
#undef CTTS_W4_MFMA1
#undef CTTS_W4_DMA1
#undef CTTS_SB
#undef CTTS_W4_FRAGS
#undef CTTS_W4_MFMA
#undef CTTS_W4_DMA
#undef CTTS_W4_WAIT_CHUNKS
#undef CTTS_UNIFORM64
#undef CTTS_WAIT_VM
    const unsigned long long st2 = (DBG & 4) ? __builtin_readcyclecounter() : 0;
    const unsigned long long rt2 = (DBG & 4) ? __builtin_amdgcn_s_memrealtime() : 0;
    bf16_epilogue<EPI>(a, acc[0], lds, t, mb, wm, 2 * wn, b, n0, l31, lhi);
    bf16_epilogue<EPI>(a, acc[1], lds, t, mb, wm, 2 * wn + 1, b, n0, l31, lhi);
    if ((DBG & 4) && blockIdx.x == 1000 && t == 0) {
        g_w4_stamps[0] = st2 - st1; g_w4_stamps[1] = rt2 - rt1;
        g_w4_stamps[2] = __builtin_readcyclecounter() - st2; g_w4_stamps[3] = st1 - st0; g_w4_stamps[4] = nch;
    }
}

// dst packed [MB][nch][4][256][8]; thread = one 16-byte unit (mb, chunk, g, r)
__global__ __launch_bounds__(256) void pack_a_bf16_kernel(bf16_t* __restrict__ dst, const float* __restrict__ src,
                                                         int nch_total, int k_off, int ksrc, int epi, int C, int M,
                                                         long long src_row_off, long long src_row_stride,
                                                         int src_k_stride, int k_group, int k_member, int part, int f16) {
    const int mb = blockIdx.y;
    const int ug = blockIdx.x;                // 8-wide k group index within [0, ksrc/8)
    const int r = threadIdx.x;
    const int drow = bgemm_dense_row(epi, mb, r, C, M);
    const int k0 = ug * 8;
    // destination k of this group: slab (32 wide) remap for interleaved members
    const int kk = k_group > 1 ? k_off + ((k0 / BGEMM_KC) * k_group + k_member) * BGEMM_KC + k0 % BGEMM_KC : k_off + k0;
    bf16_t* d = dst + ((((size_t)mb * nch_total + kk / BGEMM_KC) * 4 + (kk % BGEMM_KC) / 8) * BGEMM_BM + r) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const float v = (drow >= 0 && k < ksrc) ? src[(src_row_off + drow) * src_row_stride + (long long)k * src_k_stride] : 0.f;
        const bf16_t hi = f16 ? f32_to_f16_rne(v) : f32_to_bf16_rne(v);
        d[j] = part == 0 ? hi : f32_to_bf16_rne(v - bf16_to_f32(hi));          // (part 1: the split-bf16 form only)
    }
}

}  // namespace
}  // namespace ctts
