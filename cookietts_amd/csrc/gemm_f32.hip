// fp32 MFMA conv-GEMM for gfx950 (see gemm_f32.h for the contract).
//
// Tiling (CDNA4-first, 64-wide waves):
//   workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); block tile 256 x 128, K chunk 16.
//   wave tile 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x2_f32 -> 128 accumulator VGPRs,
//   2 waves per SIMD (2 workgroups per CU) so one wave's LDS/global/barrier time is covered
//   by the other wave's MFMAs; the f32 MFMA pipe (64 cycles per 32x32x2) is the bound.
//   LDS: 2 stages x (A 16x256 + B 16x128) fp32 = 48 KiB; both operands k-major so a
//   fragment is one conflict-free ds_read_b32 per lane (lane l: row/col l&31, k = l>>5).
//   Global->LDS staging goes through registers and is issued one chunk ahead of the MFMAs
//   (the f32 MFMA rate leaves >10x headroom on the load path).
#include "gemm_f32.h"

namespace ctts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int A_STAGE = GEMM_KC * GEMM_BM;              // 4096 floats
constexpr int B_STAGE = GEMM_KC * GEMM_BN;              // 2048 floats
constexpr int STAGE = A_STAGE + B_STAGE;                // 6144 floats = 24 KiB

__device__ __forceinline__ float4 load4(const float* p, int aligned) {
    if (aligned) return *reinterpret_cast<const float4*>(p);
    float4 v;
    v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
    return v;
}

// Gate math on the hardware transcendental unit: exp via v_exp_f32 (2^x), reciprocal via
// v_rcp_f32 (1 ulp).  Absolute error of tanh/sigmoid <= ~3e-7, far inside the parity budget.
__device__ __forceinline__ float fast_sigmoid(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
}
__device__ __forceinline__ float fast_tanh(float u) {
    // tanh(u) = 1 - 2 / (1 + e^{2u}); saturates correctly at +-inf
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * 2.8853900817779268f));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void conv_gemm_f32_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // block -> (m-block, n-tile, batch).  Dispatch places block id on XCD id % 8, so with
    // mb = id % MB an XCD keeps re-using the same 1-2 weight slices in its private L2.
    int id = blockIdx.x;
    const int mb = id % a.MB;
    id /= a.MB;
    const int tile = id % a.ntiles;
    const int b = id / a.ntiles;
    const int n0 = tile * GEMM_BN;

    const float* Ablk = a.A + (size_t)mb * a.nch_total * A_STAGE;

    // staging assignments
    const int brow = t >> 4;            // 0..15  (k row of the B chunk)
    const int bcol = (t & 15) * 4;      // 0..60  (+64 for the second load)

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Segment table -> scalar registers (static indices only: a dynamically indexed kernarg
    // struct would be copied to scratch).
    const float* sbase[GEMM_MAX_SEG];
    int snch[GEMM_MAX_SEG], salign[GEMM_MAX_SEG];
#pragma unroll
    for (int s = 0; s < GEMM_MAX_SEG; ++s) {
        const GemmSeg& g = a.seg[s];
        // everything except the k-row of the chunk folded into one per-thread base pointer
        sbase[s] = g.base + (size_t)b * g.bstride + (size_t)(mb * g.mb_rows + brow) * a.ld +
                   (a.pad + n0 + g.shift + bcol);
        snch[s] = s < a.nseg ? g.nch : 0x7fffffff;
        salign[s] = g.aligned;
    }
    const size_t chunk_rows = (size_t)GEMM_KC * a.ld;

    float4 ra0, ra1, ra2, ra3, rb0, rb1;
    int seg = 0, local = 0;
    const float* ap = Ablk + t * 4;

#define CTTS_ISSUE_LOADS()                                                                      \
    do {                                                                                        \
        ra0 = *reinterpret_cast<const float4*>(ap);                                             \
        ra1 = *reinterpret_cast<const float4*>(ap + 1024);                                      \
        ra2 = *reinterpret_cast<const float4*>(ap + 2048);                                      \
        ra3 = *reinterpret_cast<const float4*>(ap + 3072);                                      \
        ap += A_STAGE;                                                                          \
        const float* sb = seg == 0 ? sbase[0] : seg == 1 ? sbase[1] : seg == 2 ? sbase[2] : sbase[3]; \
        const int sn = seg == 0 ? snch[0] : seg == 1 ? snch[1] : seg == 2 ? snch[2] : snch[3];  \
        const int sa = seg == 0 ? salign[0] : seg == 1 ? salign[1] : seg == 2 ? salign[2] : salign[3]; \
        const float* bp = sb + (size_t)local * chunk_rows;                                      \
        rb0 = load4(bp, sa);                                                                    \
        rb1 = load4(bp + 64, sa);                                                               \
        if (++local == sn) { local = 0; ++seg; }                                                \
    } while (0)

#define CTTS_STORE_LDS(buf)                                                                     \
    do {                                                                                        \
        float* As_ = lds + (buf) * STAGE + t * 4;                                               \
        float* Bs_ = lds + (buf) * STAGE + A_STAGE + brow * GEMM_BN + bcol;                     \
        *reinterpret_cast<float4*>(As_) = ra0;                                                  \
        *reinterpret_cast<float4*>(As_ + 1024) = ra1;                                           \
        *reinterpret_cast<float4*>(As_ + 2048) = ra2;                                           \
        *reinterpret_cast<float4*>(As_ + 3072) = ra3;                                           \
        *reinterpret_cast<float4*>(Bs_) = rb0;                                                  \
        *reinterpret_cast<float4*>(Bs_ + 64) = rb1;                                             \
    } while (0)

#define CTTS_LOAD_FRAG(AV, BV, ks)                                                              \
    do {                                                                                        \
        const int krow_ = 2 * (ks) + lhi;                                                       \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) AV[mt] = As[krow_ * GEMM_BM + mt * 32]; \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) BV[nt] = Bs[krow_ * GEMM_BN + nt * 32]; \
    } while (0)

#define CTTS_MFMA(AV, BV)                                                                       \
    do {                                                                                        \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                        \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                    \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[mt], BV[nt], acc[mt][nt], 0, 0, 0); \
    } while (0)

    CTTS_ISSUE_LOADS();
    CTTS_STORE_LDS(0);
    __syncthreads();

    const int nch = a.nch_total;
    for (int ch = 0; ch < nch; ++ch) {
        const int cur = ch & 1;
        const bool more = ch + 1 < nch;
        if (more) CTTS_ISSUE_LOADS();
        const float* As = lds + cur * STAGE + wm * 128 + l31;
        const float* Bs = lds + cur * STAGE + A_STAGE + wn * 64 + l31;
        // k-step ks+1's fragments are read from LDS while ks runs on the MFMA pipe; the
        // sched_group_barrier sequence pins that software pipeline (hipcc otherwise sinks every
        // ds_read to just before its first use and exposes the LDS latency 16x per chunk).
        float av[GEMM_KC / 2][4], bv[GEMM_KC / 2][2];
#pragma unroll
        for (int ks = 0; ks < GEMM_KC / 2; ++ks) CTTS_LOAD_FRAG(av[ks], bv[ks], ks);
#pragma unroll
        for (int ks = 0; ks < GEMM_KC / 2; ++ks) CTTS_MFMA(av[ks], bv[ks]);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);      // 3 x ds_read2_b32: fragments of k-step 0
#pragma unroll
        for (int ks = 0; ks < GEMM_KC / 2 - 1; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // fragments of k-step ks+1
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // 8 MFMAs of k-step ks
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        if (more) CTTS_STORE_LDS(cur ^ 1);
        __syncthreads();
    }
#undef CTTS_ISSUE_LOADS
#undef CTTS_STORE_LDS
#undef CTTS_LOAD_FRAG
#undef CTTS_MFMA

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // bias via LDS: a global bias load between the stores would force vmcnt(0) (which on gfx9
    // also drains the stores) once per element.
    lds[t] = a.bias[mb * GEMM_BM + t];
    __syncthreads();
    const float* bias = lds + wm * 128;
    if constexpr (EPI == GEMM_EPI_GATE || EPI == GEMM_EPI_MAG) {
        float* dst = a.dst0 + (size_t)b * a.dst0_bstride;
        const int cbase = mb * 128 + wm * 64;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (cbase + mt * 32 >= a.pairC) continue;             // uniform: whole tile is channel padding
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = n0 + wn * 64 + nt * 32 + l31;
                if (n < a.L) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const int c = cbase + mt * 32 + row;
                        const float u0 = acc[mt][nt][r] + bias[mt * 32 + row];
                        const float u1 = acc[mt + 2][nt][r] + bias[64 + mt * 32 + row];
                        float v;
                        if constexpr (EPI == GEMM_EPI_GATE) v = fast_tanh(u0) * fast_sigmoid(u1);
                        else v = sqrtf(u0 * u0 + u1 * u1);
                        if (c < a.pairC) dst[(size_t)c * a.dst_ld + a.dst_pad + n] = v;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int rbase = mb * GEMM_BM + wm * 128 + mt * 32;   // uniform per tile
            if (rbase >= a.M) continue;                              // zero-padded rows of a ragged M
            const bool second = rbase >= a.split;                    // split is a multiple of 32
            float* dst = second ? a.dst1 + (size_t)b * a.dst1_bstride : a.dst0 + (size_t)b * a.dst0_bstride;
            const int accum = second ? a.acc1 : a.acc0;
            const int rdst = second ? rbase - a.split : rbase;
            // read-modify-write: issue all 32 loads of this row-tile before the first store so
            // the wave pays one memory latency per tile, not one per element (the compiler must
            // otherwise order every load behind the previous, possibly aliasing, store).
            float old[2][16];
            if (accum) {   // uniform; columns >= L of a padded row are readable, so no per-lane guard
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        old[nt][r] = dst[(size_t)(rdst + row) * a.dst_ld + a.dst_pad + n0 + wn * 64 + nt * 32 + l31];
                    }
            } else {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[nt][r] = 0.0f;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = n0 + wn * 64 + nt * 32 + l31;
                if (n < a.L) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        float v = acc[mt][nt][r] + bias[mt * 32 + row] + old[nt][r];
                        if constexpr (EPI == GEMM_EPI_LOG) v = logf(fmaxf(v, a.clip));
                        if (rbase + row < a.M) dst[(size_t)(rdst + row) * a.dst_ld + a.dst_pad + n] = v;
                    }
                }
            }
        }
    }
}

}  // namespace

int launch_gemm_f32(int epi, const GemmArgs& a, hipStream_t stream) {
    CTTS_CHECK_ARG(a.nseg >= 1 && a.nseg <= GEMM_MAX_SEG, "gemm: nseg=%d", a.nseg);
    int nch = 0;
    for (int s = 0; s < a.nseg; ++s) {
        CTTS_CHECK_ARG(a.seg[s].nch > 0, "gemm: empty segment %d", s);
        CTTS_CHECK_ARG(a.seg[s].shift >= -a.pad && a.seg[s].shift <= a.pad,
                       "gemm: shift %d exceeds halo %d", a.seg[s].shift, a.pad);
        nch += a.seg[s].nch;
    }
    CTTS_CHECK_ARG(nch == a.nch_total, "gemm: chunk count mismatch %d vs %d", nch, a.nch_total);
    CTTS_CHECK_ARG(a.ld % 4 == 0 && a.ntiles * GEMM_BN + 2 * a.pad <= a.ld && a.L <= a.ntiles * GEMM_BN,
                   "gemm: bad geometry ld=%d pad=%d L=%d ntiles=%d", a.ld, a.pad, a.L, a.ntiles);
    CTTS_CHECK_ARG(gemm_epi_is_pair(epi) || a.split % 32 == 0, "gemm: split %d not a multiple of 32", a.split);
    CTTS_CHECK_ARG(gemm_epi_is_pair(epi) ? (a.pairC > (a.MB - 1) * 128 && a.pairC <= a.MB * 128)
                                         : (a.M > (a.MB - 1) * GEMM_BM && a.M <= a.MB * GEMM_BM),
                   "gemm: M=%d pairC=%d MB=%d", a.M, a.pairC, a.MB);
    CTTS_CHECK_ARG(a.dst_ld > 0, "gemm: dst_ld not set");
    const long long blocks = (long long)a.MB * a.ntiles * a.batch;
    CTTS_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "gemm: grid %lld", blocks);
    dim3 grid((unsigned)blocks), block(256);
    switch (epi) {
        case GEMM_EPI_GATE: hipLaunchKernelGGL(conv_gemm_f32_kernel<GEMM_EPI_GATE>, grid, block, 0, stream, a); break;
        case GEMM_EPI_MAG: hipLaunchKernelGGL(conv_gemm_f32_kernel<GEMM_EPI_MAG>, grid, block, 0, stream, a); break;
        case GEMM_EPI_LOG: hipLaunchKernelGGL(conv_gemm_f32_kernel<GEMM_EPI_LOG>, grid, block, 0, stream, a); break;
        case GEMM_EPI_SPLIT: hipLaunchKernelGGL(conv_gemm_f32_kernel<GEMM_EPI_SPLIT>, grid, block, 0, stream, a); break;
        default: set_error("gemm: unknown epilogue %d", epi); return CTTS_E_ARG;
    }
    CTTS_CHECK_LAUNCH("conv_gemm_f32");
    return CTTS_OK;
}

}  // namespace ctts
