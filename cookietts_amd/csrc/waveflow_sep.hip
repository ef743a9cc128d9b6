// The two 1x1 GEMMs of a WaveFlow layer with a separable in-layer (glow_ax.py:525-531, 556-626), fused into ONE
// launch for C = 128 - everything after the depthwise stage:
//
//   depthwise output tile [128 ch][64 cols] -> LDS
//   pointwise 1x1 (C -> 2C) as a 256 x 128 x 64 MFMA GEMM    ->  + bias + upsampled conditioning -> tanh * sigmoid
//   gated tile back into the same LDS region                 ->  res/skip 1x1 (C -> 2C) as a second GEMM
//   x_{i+1} = x_i + res into the next layer's ring slot, skip accumulated into `out`                  (RMW)
//
// The two-launch form (gate GEMM 90 us + res/skip GEMM 84 us per (row, layer) at the author's sizes, against an
// MFMA floor of 80 us for both) writes and re-reads the gated activations through HBM, reads the conditioning with
// scattered 4-byte loads in an epilogue nobody overlaps, and pays a launch boundary 1216 times per call.
// The depthwise stage stays its own launch: 49 gathered taps per output want full occupancy to cover their latency
// (measured: computed inside this kernel, at 2 waves per SIMD, the layer took 325 us instead of 268).
//
// Block = 256 threads = 4 waves, 64 columns.  Wave w owns GEMM rows [64w, 64w+64) = 2 x 2 tiles of 32x32 (64
// accumulator VGPRs).  GEMM 1 rows are packed so that a wave holds 32 channels' tanh rows (tile row 0) and the SAME
// channels' sigmoid rows (tile row 1): the gate is lane-local.  A is streamed in 16-row K chunks (16 KiB) through one
// LDS stage with a register prefetch; B is the LDS tile.  50 KiB of LDS -> three workgroups per CU.
#include <mutex>

#include "waveflow_sep.h"
#include "gemm_bf16.h"   // pack_bf16x2
#include "gemm_f32.h"

namespace ctts {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SC = 128;            // channels
constexpr int SN = 64;             // columns per workgroup
constexpr int SM = 256;            // GEMM rows (2C)
constexpr int SKC = 16;            // K rows per A stage
constexpr int TILE_F = SC * SN;    // 8192 floats: depthwise tile, later the gated tile
constexpr int ASTG_F = SKC * SM;   // 4096 floats
constexpr size_t SEP_LDS_BYTES = (size_t)(TILE_F + 2 * ASTG_F + 2 * SM) * sizeof(float);   // 67 584 B: two workgroups per CU

__device__ __forceinline__ float sep_sigmoid(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
}
__device__ __forceinline__ float sep_tanh(float u) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * 2.8853900817779268f));
}

// acc[mt][nt] += A[64w + 32mt .. +32][0..128) . B[0..128)[32nt .. +32]   with A streamed from its packed image
// ([8 chunks][16][256], contiguous) through `As`, B = the LDS tile `Bs` ([128][64], k-major).
// X3: the split-bf16 form of the same contraction (see conv_gemm_f32_kernel<..., X3> in gemm_f32.hip): the 8 values a
// lane reads per fragment become one bf16x8 hi / lo operand pair, 3 bf16 MFMAs per 32x32 tile and chunk instead of 8 fp32.
typedef __bf16 sep_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int sep_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sep_split8(const float (&v)[8], sep_u32x4& hi, sep_u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int h = pack_bf16x2(v[2 * j], v[2 * j + 1]);
        hi[j] = h;
        lo[j] = pack_bf16x2(v[2 * j] - __builtin_bit_cast(float, h << 16), v[2 * j + 1] - __builtin_bit_cast(float, h & 0xffff0000u));
    }
}

template <bool X3>
__device__ __forceinline__ void tile_gemm(f32x16 (&acc)[2][2], const float* __restrict__ Ag, float* __restrict__ As,
                                          const float* __restrict__ Bs, int t, int w, int l31, int lhi, bool active) {
    // named registers: a loop-carried array here is demoted to scratch by hipcc
    const float4* Ag4 = reinterpret_cast<const float4*>(Ag) + t;
    float4* As4 = reinterpret_cast<float4*>(As) + t;
    float4 p0 = Ag4[0], p1 = Ag4[256], p2 = Ag4[512], p3 = Ag4[768];
    for (int ch = 0; ch < SC / SKC; ++ch) {
        As4[0] = p0; As4[256] = p1; As4[512] = p2; As4[768] = p3;
        __syncthreads();
        if (ch + 1 < SC / SKC) {
            const float4* nx = Ag4 + (size_t)(ch + 1) * (ASTG_F / 4);
            p0 = nx[0]; p1 = nx[256]; p2 = nx[512]; p3 = nx[768];
        }
        if (X3 && active) {
            sep_u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) v[ks] = As[(2 * ks + lhi) * SM + 64 * w + 32 * mt + l31];
                sep_split8(v, ah[mt], al[mt]);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float v[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) v[ks] = Bs[(ch * SKC + 2 * ks + lhi) * SN + 32 * nt + l31];
                sep_split8(v, bh[nt], bl[nt]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sep_bf16x8, al[mt]), __builtin_bit_cast(sep_bf16x8, bh[nt]), acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sep_bf16x8, ah[mt]), __builtin_bit_cast(sep_bf16x8, bl[nt]), acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sep_bf16x8, ah[mt]), __builtin_bit_cast(sep_bf16x8, bh[nt]), acc[mt][nt], 0, 0, 0);
                }
        } else if (active) {
            float av[SKC / 2][2], bv[SKC / 2][2];
#pragma unroll
            for (int ks = 0; ks < SKC / 2; ++ks) {
                const int krow = 2 * ks + lhi;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) av[ks][mt] = As[krow * SM + 64 * w + 32 * mt + l31];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bv[ks][nt] = Bs[(ch * SKC + krow) * SN + 32 * nt + l31];
            }
#pragma unroll
            for (int ks = 0; ks < SKC / 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][mt], bv[ks][nt], acc[mt][nt], 0, 0, 0);
            // pin the LDS -> MFMA software pipeline (k-step ks+1's fragments are read while ks runs on the matrix pipe)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int ks = 0; ks < SKC / 2 - 1; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __syncthreads();     // every wave is done with this A stage (and, after the last chunk, with Bs)
    }
}


// ---- fp32 main loop, DMA-staged and hand-scheduled (second half of round 3) ------------------------------------------
// Same contraction as tile_gemm<false>, but: A chunks go global -> LDS by DMA into TWO stages (no register round trip, no
// ds_write), one barrier per chunk, fragment reads and their waits written by hand (hipcc waits for every LDS result with
// lgkmcnt(0) once LDS-DMA is in the loop; LDS returns in order, so "k-step ks is here" = lgkmcnt(4), the four reads issued
// behind it), and the MFMA stream runs through the chunk boundary: wait-for-landing + barrier of chunk ch + 1 sit in front of
// the last k-step of chunk ch, the DMA of chunk ch + 2 (into the stage just released) and the first fragments of chunk
// ch + 1 are issued right behind them.  Split in two so that the caller can put its own loads between the first DMAs and
// the loop (in-order return: anything issued BEFORE the DMAs would be waited for by the first vmcnt).
typedef const __attribute__((address_space(1))) float* sep_gptr;
typedef __attribute__((address_space(3))) float* sep_lptr;
struct SepFrag { float a0[2], a1[2], b0[2], b1[2]; };

#define SEP_ISSUE(ch_, st_)                                                                                      \
    _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_)                                                             \
        __builtin_amdgcn_global_load_lds((sep_gptr)(Ag + (size_t)(ch_) * ASTG_F + (w + 4 * p_) * 256 + lane * 4), \
                                         (sep_lptr)(As + (st_) * ASTG_F + (w + 4 * p_) * 256), 16, 0, 0);
#define SEP_READ(ks, aaddr, baddr)                                                                               \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a0[(ks) & 1]) : "v"(aaddr), "n"((ks) * 2 * SM * 4));   \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a1[(ks) & 1]) : "v"(aaddr), "n"((ks) * 2 * SM * 4 + 128)); \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.b0[(ks) & 1]) : "v"(baddr), "n"((ks) * 2 * SN * 4));   \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.b1[(ks) & 1]) : "v"(baddr), "n"((ks) * 2 * SN * 4 + 128));
#define SEP_WAIT(n_, ks)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(f.a0[(ks) & 1]), "+v"(f.a1[(ks) & 1]), "+v"(f.b0[(ks) & 1]), "+v"(f.b1[(ks) & 1]));
#define SEP_MFMA(ks)                                                                                             \
    if (active) {                                                                                                \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[(ks) & 1], f.b0[(ks) & 1], acc[0][0], 0, 0, 0);     \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[(ks) & 1], f.b1[(ks) & 1], acc[0][1], 0, 0, 0);     \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[(ks) & 1], f.b0[(ks) & 1], acc[1][0], 0, 0, 0);     \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[(ks) & 1], f.b1[(ks) & 1], acc[1][1], 0, 0, 0);     \
    }

// chunks 0 and 1 on their way (8 DMA instructions per wave outstanding when this returns)
__device__ __forceinline__ void sep_gemm_issue(const float* __restrict__ Ag, float* As, int w, int lane) {
    SEP_ISSUE(0, 0)
    SEP_ISSUE(1, 1)
}

// EXTRA_VM = vector-memory instructions the caller issued AFTER sep_gemm_issue (they return behind chunk 1)
template <int EXTRA_VM>
__device__ __forceinline__ void sep_gemm_run(f32x16 (&acc)[2][2], const float* __restrict__ Ag, float* As, const float* Bs,
                                             int w, int lane, int l31, int lhi, bool active) {
    SepFrag f;
    const unsigned a_lane = (unsigned)(size_t)(sep_lptr)As + (unsigned)((lhi * SM + 64 * w + l31) * 4);
    const unsigned b_lane = (unsigned)(size_t)(sep_lptr)const_cast<float*>(Bs) + (unsigned)((lhi * SN + l31) * 4);
    if constexpr (EXTRA_VM + 4 <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXTRA_VM + 4) : "memory");   // chunk 0 landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SEP_READ(0, a_lane, b_lane)
#pragma unroll 1
    for (int ch = 0; ch < SC / SKC; ++ch) {
        const int st = ch & 1;
        const unsigned aa = a_lane + (unsigned)(st * ASTG_F * 4), ba = b_lane + (unsigned)(ch * SKC * SN * 4);
        const unsigned an = a_lane + (unsigned)((st ^ 1) * ASTG_F * 4), bn = b_lane + (unsigned)((ch + 1) * SKC * SN * 4);
        SEP_READ(1, aa, ba) SEP_WAIT(4, 0) SEP_MFMA(0)
        __builtin_amdgcn_sched_barrier(0);
        SEP_READ(2, aa, ba) SEP_WAIT(4, 1) SEP_MFMA(1)
        __builtin_amdgcn_sched_barrier(0);
        SEP_READ(3, aa, ba) SEP_WAIT(4, 2) SEP_MFMA(2)
        __builtin_amdgcn_sched_barrier(0);
        SEP_READ(4, aa, ba) SEP_WAIT(4, 3) SEP_MFMA(3)
        __builtin_amdgcn_sched_barrier(0);
        SEP_READ(5, aa, ba) SEP_WAIT(4, 4) SEP_MFMA(4)
        __builtin_amdgcn_sched_barrier(0);
        SEP_READ(6, aa, ba) SEP_WAIT(4, 5) SEP_MFMA(5)
        __builtin_amdgcn_sched_barrier(0);
        SEP_READ(7, aa, ba) SEP_WAIT(4, 6) SEP_MFMA(6)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // chunk ch + 1 landed (and whatever the caller slipped in)
        __builtin_amdgcn_s_barrier();                       // ... for everyone; all reads of this stage have been issued
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 2 < SC / SKC) { SEP_ISSUE(ch + 2, st) }
        SEP_READ(0, an, bn)                                 // (past the last chunk: valid LDS, never used)
        SEP_WAIT(4, 7) SEP_MFMA(7)
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a0[0]), "+v"(f.a1[0]), "+v"(f.b0[0]), "+v"(f.b1[0]));
    __syncthreads();                                        // nobody reads Bs or the A stages any more
}
#undef SEP_ISSUE
#undef SEP_READ
#undef SEP_WAIT
#undef SEP_MFMA

template <bool X3>
__global__ __launch_bounds__(256, 2) void wf_sep_layer_kernel(const WfSepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sep_lds[];     // SEP_LDS_BYTES: tile | two A stages | biases
    float* tile = sep_lds;
    float* As = tile + TILE_F;
    float* bias1 = As + 2 * ASTG_F;
    float* bias2 = bias1 + SM;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tileid = blockIdx.x % a.ntiles, b = blockIdx.x / a.ntiles;
    const int n0 = tileid * SN;
    const size_t bofs = (size_t)b * SC * a.ld;

    // ---- phase 1: depthwise output tile -> LDS (16 lanes cover the 64 columns of a channel: 256-byte runs)
    bias1[t] = a.b1[t];
    bias2[t] = a.b2[t];
    {
        const int q = t & 15, c0 = t >> 4;                 // 16 channels per pass
        const float* src = a.dwout + bofs + a.pad + n0 + 4 * q;
        float4 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const float4*>(src + (size_t)(c0 + 16 * it) * a.ld);
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<float4*>(tile + (c0 + 16 * it) * SN + 4 * q) = v[it];
    }

    // ---- phase 2: conditioning addend of this wave's 32 gate pairs, in the accumulator layout (issued now, used
    // after GEMM 1): cnd[h][nt][r] = cond[h*C + 32w + row(r)][n0 + 32nt + l31]
    float cnd[2][2][16];
    {
        const float* cb = a.cond + (size_t)b * 2 * SC * a.ld + a.pad + n0 + l31;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    cnd[h][nt][r] = cb[(size_t)(h * SC + ch) * a.ld + 32 * nt];
                }
    }
    __syncthreads();                                    // tile and biases in LDS

    // ---- phase 3: pointwise GEMM, gate
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if constexpr (X3) {
        tile_gemm<X3>(acc, a.A1, As, tile, t, w, l31, lhi, true);
    } else {
        sep_gemm_issue(a.A1, As, w, lane);
        sep_gemm_run<0>(acc, a.A1, As, tile, w, lane, l31, lhi, true);
    }
    // (tile_gemm ends with a barrier: nobody reads the depthwise tile any more)
#define CTTS_SEP_GATE(EXPR)                                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                        \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * lhi;                                                \
            const float u0 = acc[0][nt][r] + bias1[64 * w + rr] + cnd[0][nt][r];                            \
            const float u1 = acc[1][nt][r] + bias1[64 * w + 32 + rr] + cnd[1][nt][r];                       \
            tile[(32 * w + rr) * SN + 32 * nt + l31] = EXPR;                                                \
        }
    if (a.gate == GATE_GTU) {                               // uniform
        CTTS_SEP_GATE(sep_tanh(u0) * sep_sigmoid(u1))
    } else {                                                // the other gated units (glow_ax.py:45-165), one loop copy each
        switch (a.gate) {
#define CTTS_SEP_CASE(K) case K: CTTS_SEP_GATE(gate_eval<K>(u0, u1)) break;
            CTTS_SEP_CASE(1) CTTS_SEP_CASE(2) CTTS_SEP_CASE(3) CTTS_SEP_CASE(4) CTTS_SEP_CASE(5) CTTS_SEP_CASE(6)
            CTTS_SEP_CASE(7) CTTS_SEP_CASE(8) CTTS_SEP_CASE(9) CTTS_SEP_CASE(10) CTTS_SEP_CASE(11) CTTS_SEP_CASE(12)
            default: CTTS_SEP_GATE(gate_eval<13>(u0, u1)) break;
#undef CTTS_SEP_CASE
        }
    }
#undef CTTS_SEP_GATE
    __syncthreads();

    // ---- phase 4: res/skip GEMM on the gated tile
    const bool active = 64 * w < a.rs_rows;             // last layer: 128 skip rows only
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // rows < split -> x_{i+1} = x_i + res, rows >= split -> out (+)= skip: the old values of this wave's two row tiles
    // (64 registers, the ones the conditioning addend occupied until the gate) are requested together with the first A
    // chunks of the second GEMM instead of behind it
    const int split = a.rs_rows == SM ? SC : 0;
    float old[2][2][16];
    auto load_old = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int rbase = 64 * w + 32 * mt;
            const bool second = rbase >= split;
            const float* src = (second ? a.out : a.xin) + bofs;
            const bool accum = second ? a.acc_out != 0 : true;
            const int rdst = second ? rbase - split : rbase;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    old[mt][nt][r] = (accum && active) ? src[(size_t)(rdst + rr) * a.ld + a.pad + n0 + 32 * nt + l31] : 0.0f;
                }
        }
    };
    if constexpr (X3) {
        tile_gemm<X3>(acc, a.A2, As, tile, t, w, l31, lhi, active);
        if (!active) return;
        load_old();
    } else {
        load_old();                                         // in flight together with the first A chunks (loads return in
        sep_gemm_issue(a.A2, As, w, lane);                  // order: issued behind the DMAs they would need a wait count of
        sep_gemm_run<0>(acc, a.A2, As, tile, w, lane, l31, lhi, active);   // 68, two more than the counter has)
        if (!active) return;
    }

    // ---- phase 5: stores
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int rbase = 64 * w + 32 * mt;
        const bool second = rbase >= split;
        float* dst = (second ? a.out : a.xout) + bofs;
        const int rdst = second ? rbase - split : rbase;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + 32 * nt + l31;
            if (n < a.L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    dst[(size_t)(rdst + rr) * a.ld + a.pad + n] = acc[mt][nt][r] + bias2[rbase + rr] + old[mt][nt][r];
                }
            }
        }
    }
}

// A images: [8][16][256] with A[chunk][kk][m] = W[dense_row(m)][16*chunk + kk]; gate order for the pointwise layer:
// m -> wave m/64, tanh (m%64 < 32) or sigmoid half, pair index m%32 -> dense row half*C + 32*(m/64) + m%32
__global__ __launch_bounds__(256) void wf_sep_pack_kernel(const float* __restrict__ pw_w, const float* __restrict__ pw_b,
                                                          const float* __restrict__ rs_w, const float* __restrict__ rs_b,
                                                          float* __restrict__ A1, float* __restrict__ b1,
                                                          float* __restrict__ A2, float* __restrict__ b2, int rs_rows) {
    const int m = threadIdx.x, k = blockIdx.x;          // k in [0, 128)
    const int d1 = ((m % 64) / 32) * SC + 32 * (m / 64) + m % 32;
    A1[(size_t)(k / SKC) * ASTG_F + (k % SKC) * SM + m] = pw_w[(size_t)d1 * SC + k];
    A2[(size_t)(k / SKC) * ASTG_F + (k % SKC) * SM + m] = m < rs_rows ? rs_w[(size_t)m * SC + k] : 0.0f;
    if (k == 0) {
        b1[m] = pw_b[d1];
        b2[m] = m < rs_rows ? rs_b[m] : 0.0f;
    }
}

}  // namespace

bool wf_sep_supported(int C) { return C == SC; }

int launch_wf_sep_pack(const float* pw_w, const float* pw_b, const float* rs_w, const float* rs_b, float* A1, float* b1,
                       float* A2, float* b2, int rs_rows, hipStream_t s) {
    CTTS_CHECK_ARG(pw_w && pw_b && rs_w && rs_b && A1 && b1 && A2 && b2 && (rs_rows == SM || rs_rows == SC),
                   "wf_sep_pack: bad argument");
    hipLaunchKernelGGL(wf_sep_pack_kernel, dim3(SC), dim3(256), 0, s, pw_w, pw_b, rs_w, rs_b, A1, b1, A2, b2, rs_rows);
    CTTS_CHECK_LAUNCH("wf_sep_pack");
    return CTTS_OK;
}

int launch_wf_sep_layer(const WfSepArgs& a, int batch, hipStream_t s) {
    CTTS_CHECK_ARG(a.L <= a.ntiles * SN && a.ntiles * SN + a.pad <= a.ld && a.pad % 4 == 0 && a.ld % 4 == 0 && a.dwout &&
                   a.cond && a.xin && a.out, "wf_sep_layer: geometry L=%d ld=%d pad=%d", a.L, a.ld, a.pad);
    {   // 67 584 B of dynamic LDS: above the 64 KiB default, opt in once per process (one process per GPU)
        static std::mutex mu;
        static bool attr_set = false;
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set) {
            CTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wf_sep_layer_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEP_LDS_BYTES));
            CTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wf_sep_layer_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEP_LDS_BYTES));
            attr_set = true;
        }
    }
    if (a.split_bf16) hipLaunchKernelGGL(wf_sep_layer_kernel<true>, dim3((unsigned)(a.ntiles * batch)), dim3(256), SEP_LDS_BYTES, s, a);
    else hipLaunchKernelGGL(wf_sep_layer_kernel<false>, dim3((unsigned)(a.ntiles * batch)), dim3(256), SEP_LDS_BYTES, s, a);
    CTTS_CHECK_LAUNCH("wf_sep_layer");
    return CTTS_OK;
}

}  // namespace ctts
