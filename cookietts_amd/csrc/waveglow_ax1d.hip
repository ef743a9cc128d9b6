// "ax" WaveGlow core with waveflow=False: AffineCouplingBlock + 1-D WN + channel mixing by
// InvertibleConv1x1 or PermuteHeight (reference: _4_mtw/waveglow/efficient_model_ax.py:309-346,
// glow_ax.py:375-418, efficient_modules.py:94-105, 269-286, 360-403).  See include/cookietts_hip.h.
//
// Data layout: every tensor is the padded row layout of the conv-GEMM, [B][rows][ld] with the L valid
// time steps at columns [pad, pad + L).  `audio` keeps all n_group rows for the whole call: the latent of
// flow k is rows [ch_off_k, n_group), the early-output chunks sit in the rows above it in the order
// the reference concatenates them back (ax:310-316, 340-341), so "cat" is a change of ch_off.
//
// Per flow, in the reference's inverse order:
//   [un-mix]      (mix_first=False: ax:324-325)              \
//   start 1x1     (glow_ax.py:376)                             } second half of ax_boundary_kernel
//   conditioning  frame-rate rows, linearly interpolated       inside the GATE epilogue (glow_ax.py:362-373, 389-390)
//   n_layers x (dilated conv + cond + gate | res/skip 1x1)    conv_gemm_f32<GATE> with the conditioning as
//                                                             interpolated epilogue addend, conv_gemm_f32<SPLIT>
//   end 1x1       (2*n_half <= 32 rows: not a GEMM-shaped job) \
//   coupling inverse, NaN -> 0, [un-mix if mix_first]          } first half of ax_boundary_kernel (em:100-104, ax:333-337)
// Everything between the last res/skip GEMM of flow k and the first in-layer GEMM of flow k - 1 works on the <= 32
// latent rows of one time step, so it is ONE launch per flow boundary (was: a 256-row GEMM block for <= 32 `end` rows +
// three elementwise launches = 58 us of the 1.1 ms a notebook-config flow takes at batch 1).
// `output` starts from the first layer's skip (glow_ax.py:405-410): 0 + r == r exactly, so the SPLIT epilogue's
// "store on layer 0, accumulate afterwards" is bit-identical.
#include <algorithm>
#include <vector>

#include "gemm_f32.h"
#include "waveglow_kernels.h"

namespace ctts {
namespace {

constexpr size_t ALIGN_F = 64;
inline size_t align_up(size_t v) { return (v + ALIGN_F - 1) / ALIGN_F * ALIGN_F; }
constexpr int A_TILE = GEMM_KC * GEMM_BM;
constexpr int AX_MAX_GROUP = 32;     // LDS of the mixing kernels: (n*n + n*256 + n) floats <= 37 KB

struct AxFlowDims { int n_rem, n_half, ch_off; };

struct AxPlan {
    ctts_wgax_config c;
    int C, nch_in, nch_c, mb_in;
    std::vector<AxFlowDims> fd;
    struct Flow {
        size_t start_w, start_b, end_w, end_b, winv;
        std::vector<size_t> in_A, in_b, rs_A, rs_b;
    };
    std::vector<Flow> fl;
    size_t total;
    int rs_rows(int layer) const { return (layer < c.n_layers - 1 && !c.merge_res_skip) ? 2 * C : C; }
    int rs_mb(int layer) const { return (rs_rows(layer) + GEMM_BM - 1) / GEMM_BM; }
};

int make_ax_plan(const ctts_wgax_config* cfg, AxPlan& p) {
    CTTS_CHECK_ARG(cfg != nullptr, "wgax: config is NULL");
    p.c = *cfg;
    const auto& c = p.c;
    CTTS_CHECK_ARG(c.n_flows >= 1 && c.n_layers >= 1 && c.n_layers <= 12, "wgax: n_flows=%d n_layers=%d", c.n_flows, c.n_layers);
    CTTS_CHECK_ARG(gemm_mode_valid(c.f32_gemm_mode), "f32_gemm_mode=%d (CTTS_GEMM_DEFAULT / _F32 / _BF16X3 / _BF16X6)", c.f32_gemm_mode);
    CTTS_CHECK_ARG(c.n_group >= 2 && c.n_group % 2 == 0 && c.n_group <= AX_MAX_GROUP, "wgax: n_group=%d (even, <= 32)", c.n_group);
    CTTS_CHECK_ARG(c.kernel_size % 2 == 1 && c.kernel_size >= 1 && c.kernel_size <= GEMM_MAX_SEG - 1,
                   "wgax: kernel_size=%d (odd, <= 11)", c.kernel_size);
    // 32: the res / skip split row of the SPLIT epilogue is a multiple of 32; a ragged last M-block (channels that are
    // not a multiple of 128) is zero-padded in the packed weights and masked in the epilogues
    CTTS_CHECK_ARG(c.n_channels >= 32 && c.n_channels % 32 == 0, "wgax: n_channels=%d (multiple of 32)", c.n_channels);
    CTTS_CHECK_ARG(c.n_early_every >= 1 && c.n_early_size >= 0 && c.n_early_size % 2 == 0, "wgax: early outputs");
    CTTS_CHECK_ARG(c.mixing == CTTS_MIX_PERMUTE || c.mixing == CTTS_MIX_CONV1X1, "wgax: mixing=%d", c.mixing);
    CTTS_CHECK_ARG(c.mixing != CTTS_MIX_PERMUTE || c.n_flows % 2 == 0, "wgax: PermuteHeight requires even n_flows");
    CTTS_CHECK_ARG(c.gated_unit >= 0 && c.gated_unit < GATE_KINDS && (c.merge_res_skip == 0 || c.merge_res_skip == 1),
                   "wgax: gated_unit=%d merge_res_skip=%d", c.gated_unit, c.merge_res_skip);
    p.C = c.n_channels;
    p.nch_c = p.C / GEMM_KC;
    p.nch_in = c.kernel_size * p.nch_c;
    p.mb_in = (2 * p.C + GEMM_BM - 1) / GEMM_BM;
    int n_rem = c.n_group;
    p.fd.resize(c.n_flows);
    for (int k = 0; k < c.n_flows; ++k) {                       // ax:170-189
        if (k % c.n_early_every == 0 && k > 0) n_rem -= c.n_early_size;
        CTTS_CHECK_ARG(n_rem >= 2 && n_rem % 2 == 0, "wgax: flow %d has %d remaining channels", k, n_rem);
        p.fd[k] = {n_rem, n_rem / 2, c.n_group - n_rem};
    }
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    p.fl.resize(c.n_flows);
    for (int k = 0; k < c.n_flows; ++k) {
        auto& f = p.fl[k];
        const auto& d = p.fd[k];
        f.start_w = take((size_t)p.C * d.n_half);
        f.start_b = take(p.C);
        f.end_w = take((size_t)2 * d.n_half * p.C);             // dense [2h][C], read by ax_boundary_kernel
        f.end_b = take((size_t)2 * d.n_half);
        f.winv = take((size_t)d.n_rem * d.n_rem);
        for (int i = 0; i < c.n_layers; ++i) {
            f.in_A.push_back(take((size_t)p.mb_in * p.nch_in * A_TILE));
            f.in_b.push_back(take((size_t)p.mb_in * GEMM_BM));
            f.rs_A.push_back(take((size_t)p.rs_mb(i) * p.nch_c * A_TILE));
            f.rs_b.push_back(take((size_t)p.rs_mb(i) * GEMM_BM));
        }
    }
    p.total = o;
    return CTTS_OK;
}

struct AxGeom { int L, ld, pad, ntiles; };

int make_ax_geom(const AxPlan& p, long long samples, AxGeom& g) {
    CTTS_CHECK_ARG(samples >= p.c.n_group && samples % p.c.n_group == 0 && samples / p.c.n_group < (1 << 30),
                   "wgax: samples=%lld not a positive multiple of n_group=%d", samples, p.c.n_group);
    g.L = (int)(samples / p.c.n_group);
    int reach = 0;
    for (int i = 0; i < p.c.n_layers; ++i) reach = std::max(reach, (p.c.kernel_size / 2) * (p.c.dilation_w[i] > 0 ? p.c.dilation_w[i] : 1 << i));
    g.pad = round_up(reach > 128 ? reach : 128, 32);
    g.ntiles = (g.L + GEMM_BN - 1) / GEMM_BN;
    g.ld = round_up(g.L, 256) + 2 * g.pad;
    return CTTS_OK;
}

struct AxWs { float *audio, *x, *act, *out; size_t total; };

void ax_carve(const AxPlan& p, const AxGeom& g, int batch, float* base, AxWs& w) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return base ? base + r : nullptr; };
    const size_t B = batch;
    w.audio = take(B * p.c.n_group * g.ld);
    w.x = take(B * p.C * g.ld);
    w.act = take(B * p.C * g.ld);
    w.out = take(B * p.C * g.ld);
    w.total = o;
}

// PermuteHeight (efficient_modules.py:360-403): out[i] = in[perm[i]]; reverse all rows for k % 4 in {0, 1},
// reverse each half separately for k % 4 in {2, 3}.  Its own inverse.
void ax_permutation(int k, int n, int* perm) {
    if (k % 4 == 2 || k % 4 == 3) {
        const int half = n / 2;
        for (int i = 0; i < half; ++i) perm[i] = half - 1 - i;
        for (int i = half; i < n; ++i) perm[i] = n - 1 - (i - half);
    } else {
        for (int i = 0; i < n; ++i) perm[i] = n - 1 - i;
    }
}

// audio rows [b][g][pad + l] = z[b][G*l + g]   (ax:310); halo and tail columns are left untouched (zero)
__global__ __launch_bounds__(256) void ax_squeeze_kernel(const float* __restrict__ z, float* __restrict__ audio, int G, int L,
                                                         int ld, int pad) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    const float* zb = z + (size_t)b * G * L + (size_t)l * G;
    float* ab = audio + (size_t)b * G * ld + pad + l;
    for (int g = 0; g < G; ++g) ab[(size_t)g * ld] = zb[g];
}

// wave[b][G*l + g] = audio[b][g][pad + l]   (ax:346)
__global__ __launch_bounds__(256) void ax_unsqueeze_kernel(const float* __restrict__ audio, float* __restrict__ wave, int G,
                                                           int L, int ld, int pad) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    const float* ab = audio + (size_t)b * G * ld + pad + l;
    float* wb = wave + (size_t)b * G * L + (size_t)l * G;
    for (int g = 0; g < G; ++g) wb[g] = ab[(size_t)g * ld];
}

// ---- the flow boundary: end 1x1 + coupling inverse (+ un-mix) of flow k, (un-mix +) start 1x1 of flow k - 1 -----------
// A workgroup owns 64 time steps and keeps their G <= 32 latent rows in LDS from the first load to the last store; 8
// waves.  At batch 1 a launch is ~180 workgroups, one per CU: what matters is the number of DEPENDENT memory round trips,
// so both small contractions are laid out for "issue every load, wait once":
//   end      e[r] = be[r] + sum_c We[r][c] * out[c]     (<= 32 rows x C) as v_mfma_f32_32x32x2_f32 over (32 rows) x (2 x
//            32 time steps): the waves split C in chunks of 32 channels, fragments straight from global memory (16 + 32
//            independent loads per chunk), partial tiles meet in LDS
//   couple   (log_s, t) = (e[:h], e[h:]);  a1 = (a1 - t) / exp(log_s)   (efficient_modules.py:100-103; note the order,
//            glow.py has (b, log_s));  NaN -> 0 on the flow's latent rows (ax:13-16, 333-334)
//   mix      PERMUTE: out[i] = in[src[i]] (em:360-373);  CONV1X1: out[i] = sum_j Winv[i][j] * in[j], j ascending with
//            fma (em:283: F.conv1d with W^-1) - after the coupling when mix_first, before `start` otherwise
//   start    x[c] = bs[c] + sum_{j < h'} Ws[c][j] * a[ch_off' + j]   (glow_ax.py:376): C rows x (K = h' <= 16) on the same
//            MFMA, a wave per 32 channels, B fragments from the LDS rows, accumulators initialised with the bias
struct AxPerm { int src[AX_MAX_GROUP]; };

struct AxBoundary {
    float* audio;               // [B][G][ld]
    int G, C, L, ld, pad, mixing, ignore_nan;
    // flow k just finished its layers (do_couple)
    int do_couple, ch_off, h, mix_after;
    const float* out;           // [B][C][ld] skip sum
    const float *end_w, *end_b; // [2h][C], [2h]
    const float* winv;          // [2h][2h] (CONV1X1 and mix_after)
    AxPerm perm;
    // flow k - 1 starts (do_start)
    int do_start, s_ch_off, s_h, mix_before;
    const float* s_winv;
    AxPerm s_perm;
    const float *start_w, *start_b;   // [C][h'], [C]
    float* x;                   // [B][C][ld]
};

typedef float axb_f32x16 __attribute__((ext_vector_type(16)));
constexpr int AXB_COLS = 64;
constexpr int AXB_ROWS = AX_MAX_GROUP;
constexpr int AXB_WAVES = 8;
constexpr int AXB_THREADS = 64 * AXB_WAVES;
// LDS floats: latent rows | 4 partial e tiles | e rows | mix scratch | one mixing matrix | one permutation
constexpr int AXB_TILE = AXB_ROWS * AXB_COLS;
constexpr int AXB_LDS_FLOATS = AXB_TILE + 4 * AXB_TILE + AXB_TILE + AXB_TILE + AXB_ROWS * AXB_ROWS + AXB_ROWS;
static_assert(AXB_LDS_FLOATS * sizeof(float) <= 64 * 1024, "ax_boundary_kernel: static LDS");

// rows [off, off + n) of sa <- mix(rows [off, off + n) of sa), through sb; every thread calls it
__device__ __forceinline__ void axb_mix(float* sa, float* sb, float* sW, int* sP, const float* __restrict__ Winv,
                                        const AxPerm& perm, int mixing, int off, int n, int lane, int wave) {
    const int t = wave * 64 + lane;
    __syncthreads();                                        // previous users of sW / sP / sb are done, sa is complete
    if (mixing == CTTS_MIX_CONV1X1) {
        for (int i = t; i < n * n; i += AXB_THREADS) sW[i] = Winv[i];
    } else if (t < AX_MAX_GROUP) {
        int v = 0;
#pragma unroll
        for (int q = 0; q < AX_MAX_GROUP; ++q) v = (t == q) ? perm.src[q] : v;   // a kernarg array indexed at run time would go to scratch
        sP[t] = v;
    }
    __syncthreads();
    for (int i = wave; i < n; i += AXB_WAVES) {
        float v;
        if (mixing == CTTS_MIX_PERMUTE) {
            v = sa[(off + sP[i]) * AXB_COLS + lane];
        } else {
            v = 0.f;
            for (int j = 0; j < n; ++j) v = fmaf(sW[i * n + j], sa[(off + j) * AXB_COLS + lane], v);
        }
        sb[i * AXB_COLS + lane] = v;
    }
    __syncthreads();
    for (int i = wave; i < n; i += AXB_WAVES) sa[(off + i) * AXB_COLS + lane] = sb[i * AXB_COLS + lane];
}

__global__ __launch_bounds__(AXB_THREADS) void ax_boundary_kernel(const AxBoundary p) {
    __shared__ float smem[AXB_LDS_FLOATS];
    float* sa = smem;                                   // [G][64]  latent rows of this tile
    float* red = sa + AXB_TILE;                         // [4][32][64] partial e tiles (waves w and w + 4 share slot w)
    float* se = red + 4 * AXB_TILE;                     // [32][64] e rows
    float* sb = se + AXB_TILE;                          // [32][64] mix scratch
    float* sW = sb + AXB_TILE;                          // [32 * 32]
    int* sP = reinterpret_cast<int*>(sW + AXB_ROWS * AXB_ROWS);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int l0 = blockIdx.x * AXB_COLS;
    const int b = blockIdx.y;
    const int G = p.G, C = p.C;
    const int l = l0 + lane;
    const bool valid = l < p.L;
    float* ab = p.audio + (size_t)b * G * p.ld + p.pad;
    int first_dirty = G;                                 // rows >= first_dirty go back to HBM

    // loads of a ragged last tile re-read the last valid time step
    for (int g = wave; g < G; g += AXB_WAVES) sa[g * AXB_COLS + lane] = ab[(size_t)g * p.ld + min(l, p.L - 1)];

    if (p.do_couple) {
        const int h = p.h, n = 2 * h;
        axb_f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const float* ob = p.out + (size_t)b * C * p.ld + p.pad;
        const int col0 = min(l0 + l31, p.L - 1), col1 = min(l0 + 32 + l31, p.L - 1);
        const float* wrow = p.end_w + (size_t)min(l31, n - 1) * C + lhi;
        const float wmask = l31 < n ? 1.f : 0.f;         // rows >= n of the 32-row tile: zero weights
        for (int q = wave; q < C / 32; q += AXB_WAVES) {
            float av[16], b0[16], b1[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                av[ks] = wrow[q * 32 + 2 * ks];
                const float* orow = ob + (size_t)(q * 32 + 2 * ks + lhi) * p.ld;
                b0[ks] = orow[col0];
                b1[ks] = orow[col1];
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks] * wmask, b0[ks], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks] * wmask, b1[ks], acc[1], 0, 0, 0);
            }
        }
        // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        float* slot = red + (wave & 3) * AXB_TILE;
        if (wave < 4) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slot[((r & 3) + 8 * (r >> 2) + 4 * lhi) * AXB_COLS + nt * 32 + l31] = acc[nt][r];
        }
        __syncthreads();
        if (wave >= 4) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slot[((r & 3) + 8 * (r >> 2) + 4 * lhi) * AXB_COLS + nt * 32 + l31] += acc[nt][r];
        }
        __syncthreads();
        for (int r = wave; r < n; r += AXB_WAVES) {
            const float* q = red + r * AXB_COLS + lane;
            se[r * AXB_COLS + lane] = p.end_b[r] + ((q[0] + q[AXB_TILE]) + (q[2 * AXB_TILE] + q[3 * AXB_TILE]));
        }
        __syncthreads();
        for (int j = wave; j < h; j += AXB_WAVES) {
            float a0 = sa[(p.ch_off + j) * AXB_COLS + lane];
            float a1 = (sa[(p.ch_off + h + j) * AXB_COLS + lane] - se[(h + j) * AXB_COLS + lane]) / expf(se[j * AXB_COLS + lane]);
            if (p.ignore_nan) { a0 = (a0 != a0) ? 0.f : a0; a1 = (a1 != a1) ? 0.f : a1; }
            sa[(p.ch_off + j) * AXB_COLS + lane] = a0;
            sa[(p.ch_off + h + j) * AXB_COLS + lane] = a1;
        }
        if (p.mix_after) axb_mix(sa, sb, sW, sP, p.winv, p.perm, p.mixing, p.ch_off, n, lane, wave);
        first_dirty = p.ch_off;
    }
    if (p.do_start && p.mix_before) {
        axb_mix(sa, sb, sW, sP, p.s_winv, p.s_perm, p.mixing, p.s_ch_off, 2 * p.s_h, lane, wave);
        first_dirty = min(first_dirty, p.s_ch_off);
    }
    __syncthreads();
    if (valid)
        for (int g = first_dirty + wave; g < G; g += AXB_WAVES) ab[(size_t)g * p.ld + l] = sa[g * AXB_COLS + lane];
    if (p.do_start) {
        const int hs = p.s_h;
        float* xb = p.x + (size_t)b * C * p.ld + p.pad;
        // B fragments of the K = 16 (h' zero-padded) contraction: the same for every 32-channel tile of this wave
        float b0[8], b1[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int j = 2 * ks + lhi;
            const float* arow = sa + (p.s_ch_off + min(j, hs - 1)) * AXB_COLS;
            b0[ks] = j < hs ? arow[l31] : 0.f;
            b1[ks] = j < hs ? arow[32 + l31] : 0.f;
        }
        for (int m = wave; m < C / 32; m += AXB_WAVES) {
            const int c0 = 32 * m;
            float av[8];
            const float* wrow = p.start_w + (size_t)(c0 + l31) * hs;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int j = 2 * ks + lhi;
                av[ks] = j < hs ? wrow[j] : 0.f;
            }
            axb_f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = p.start_b[c0 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], b0[ks], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], b1[ks], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = l0 + nt * 32 + l31;
                if (col < p.L) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        xb[(size_t)(c0 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * p.ld + col] = acc[nt][r];
                }
            }
        }
    }
}

// x[b][c][pad - halo .. pad) = x[b][c][pad];  x[b][c][pad + T .. pad + T + halo) = x[b][c][pad + T - 1]
__global__ __launch_bounds__(64) void replicate_halo_kernel(float* __restrict__ x, int T, int ld, int pad, int halo) {
    float* row = x + (size_t)blockIdx.x * ld + pad;
    const int i = threadIdx.x;
    if (i < halo) { row[-1 - i] = row[0]; row[T + i] = row[T - 1]; }
}

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

size_t ctts_wgax_packed_bytes(const ctts_wgax_config* cfg) {
    AxPlan p;
    if (make_ax_plan(cfg, p)) return 0;
    return p.total * sizeof(float);
}

int ctts_wgax_pack_flow(const ctts_wgax_config* cfg, int32_t k, const ctts_wgax_flow_weights* w, void* packed,
                        void* stream) {
    AxPlan p;
    int rc = make_ax_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(k >= 0 && k < p.c.n_flows && w && packed, "wgax pack_flow: bad argument (flow %d)", k);
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    const auto& f = p.fl[k];
    const auto& d = p.fd[k];
    const int C = p.C, ks = p.c.kernel_size;
    auto d2d = [&](size_t off, const float* src, size_t n) -> int {
        CTTS_CHECK_ARG(src != nullptr, "wgax pack_flow: NULL weight pointer");
        CTTS_CHECK_HIP(hipMemcpyAsync(blob + off, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return CTTS_OK;
    };
    if ((rc = d2d(f.start_w, w->start_w, (size_t)C * d.n_half))) return rc;
    if ((rc = d2d(f.start_b, w->start_b, C))) return rc;
    if (p.c.mixing == CTTS_MIX_CONV1X1) {
        if ((rc = d2d(f.winv, w->w_inverse, (size_t)d.n_rem * d.n_rem))) return rc;
    }
    CTTS_CHECK_ARG(w->end_w && w->end_b && w->in_w && w->in_b && w->rs_w && w->rs_b, "wgax pack_flow: NULL weights");
    if ((rc = d2d(f.end_w, w->end_w, (size_t)2 * d.n_half * C))) return rc;      // dense [2h][C] as it is
    if ((rc = d2d(f.end_b, w->end_b, (size_t)2 * d.n_half))) return rc;
    for (int i = 0; i < p.c.n_layers; ++i) {
        CTTS_CHECK_ARG(w->in_w[i] && w->in_b[i] && w->rs_w[i] && w->rs_b[i], "wgax pack_flow: NULL layer %d weights", i);
        // K = [per 16-channel slab: tap 0 .. tap ks-1];  in_w[i] is [2C][C][ks]
        for (int t = 0; t < ks; ++t)
            if ((rc = launch_pack_a(blob + f.in_A[i], w->in_w[i] + t, GEMM_BM, p.mb_in, p.nch_in, 0, C, GEMM_EPI_GATE, C, 2 * C,
                                    0, (long long)C * ks, ks, s, ks, t))) return rc;
        if ((rc = launch_pack_bias(blob + f.in_b[i], GEMM_BM, p.mb_in, w->in_b[i], 0, nullptr, 0, GEMM_EPI_GATE, C, 2 * C,
                                   s))) return rc;
        const int rows = p.rs_rows(i);
        if ((rc = launch_pack_a(blob + f.rs_A[i], w->rs_w[i], GEMM_BM, p.rs_mb(i), p.nch_c, 0, C, GEMM_EPI_SPLIT, C, rows, 0,
                                C, 1, s))) return rc;
        if ((rc = launch_pack_bias(blob + f.rs_b[i], GEMM_BM, p.rs_mb(i), w->rs_b[i], 0, nullptr, 0, GEMM_EPI_SPLIT, C,
                                   rows, s))) return rc;
    }
    return CTTS_OK;
}

size_t ctts_wgax_workspace_bytes(const ctts_wgax_config* cfg, int32_t batch, int64_t samples) {
    AxPlan p; AxGeom g; AxWs w;
    if (make_ax_plan(cfg, p) || make_ax_geom(p, samples, g) || batch < 1) return 0;
    ax_carve(p, g, batch, nullptr, w);
    return w.total * sizeof(float);
}

int ctts_replicate_halo_f32(float* x, int32_t batch, int32_t C, int32_t T, int32_t ld, int32_t pad, int32_t halo,
                            void* stream) {
    CTTS_CHECK_ARG(x && batch >= 1 && C >= 1 && T >= 1 && halo >= 0 && halo <= 64 && halo <= pad && pad + T + halo <= ld,
                   "replicate_halo: bad argument");
    if (halo == 0) return CTTS_OK;
    hipLaunchKernelGGL(replicate_halo_kernel, dim3((unsigned)((size_t)batch * C)), dim3(64), 0, as_stream(stream), x, T, ld,
                       pad, halo);
    CTTS_CHECK_LAUNCH("replicate_halo");
    return CTTS_OK;
}

int ctts_wgax_inverse_f32(const ctts_wgax_config* cfg, const void* packed, const float* z, const float* cond,
                          int32_t cond_ld, int32_t cond_pad, int32_t frames, float* audio, int32_t batch,
                          int64_t samples, void* workspace, size_t workspace_bytes, void* stream) {
    AxPlan p; AxGeom g; AxWs w;
    int rc = make_ax_plan(cfg, p); if (rc) return rc;
    rc = make_ax_geom(p, samples, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && z && cond && audio && workspace && batch >= 1 && frames >= 1 && cond_ld >= frames + cond_pad,
                   "wgax inverse: bad argument");
    ax_carve(p, g, batch, static_cast<float*>(workspace), w);
    if (w.total * sizeof(float) > workspace_bytes) {
        set_error("wgax inverse: workspace %zu bytes < required %zu", workspace_bytes, w.total * sizeof(float));
        return CTTS_E_WORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    const int G = p.c.n_group, C = p.C, L = g.L, ks = p.c.kernel_size, nl = p.c.n_layers;
    const long long cstride = (long long)C * g.ld;
    const dim3 lgrid((L + 255) / 256, batch);

    hipLaunchKernelGGL(ax_squeeze_kernel, lgrid, dim3(256), 0, s, z, w.audio, G, L, g.ld, g.pad);
    CTTS_CHECK_LAUNCH("ax_squeeze");

    auto base_args = [&]() {
        GemmArgs a{};
        a.ld = g.ld; a.pad = g.pad; a.L = L; a.ntiles = g.ntiles; a.batch = batch;
        a.dst_ld = g.ld; a.dst_pad = g.pad;
        a.gemm_mode = p.c.f32_gemm_mode;
        return a;
    };
    // one launch per flow boundary: end + coupling (+ un-mix) of flow `done`, (un-mix +) start of flow `next`; -1 = none
    auto boundary = [&](int done, int next) -> int {
        AxBoundary q{};
        q.audio = w.audio; q.G = G; q.C = C; q.L = L; q.ld = g.ld; q.pad = g.pad;
        q.mixing = p.c.mixing; q.ignore_nan = p.c.ignore_nan ? 1 : 0;
        if (done >= 0) {
            const auto& f = p.fl[done];
            const auto& d = p.fd[done];
            q.do_couple = 1; q.ch_off = d.ch_off; q.h = d.n_half; q.mix_after = p.c.mix_first ? 1 : 0;
            q.out = w.out; q.end_w = blob + f.end_w; q.end_b = blob + f.end_b; q.winv = blob + f.winv;
            if (p.c.mixing == CTTS_MIX_PERMUTE) ax_permutation(done, d.n_rem, q.perm.src);
        }
        if (next >= 0) {
            const auto& f = p.fl[next];
            const auto& d = p.fd[next];
            q.do_start = 1; q.s_ch_off = d.ch_off; q.s_h = d.n_half; q.mix_before = p.c.mix_first ? 0 : 1;   // ax:324-325
            q.s_winv = blob + f.winv; q.start_w = blob + f.start_w; q.start_b = blob + f.start_b; q.x = w.x;
            if (p.c.mixing == CTTS_MIX_PERMUTE) ax_permutation(next, d.n_rem, q.s_perm.src);
        }
        hipLaunchKernelGGL(ax_boundary_kernel, dim3((L + AXB_COLS - 1) / AXB_COLS, batch), dim3(AXB_THREADS), 0, s, q);
        CTTS_CHECK_LAUNCH("ax_boundary");
        return CTTS_OK;
    };
    if ((rc = boundary(-1, p.c.n_flows - 1))) return rc;
    for (int k = p.c.n_flows - 1; k >= 0; --k) {
        const auto& f = p.fl[k];
        const float* fr = cond + (size_t)k * batch * 2 * C * nl * cond_ld;
        for (int i = 0; i < nl; ++i) {
            const int dil = p.c.dilation_w[i] > 0 ? p.c.dilation_w[i] : 1 << i;
            {
                GemmArgs a = base_args();
                a.A = blob + f.in_A[i]; a.bias = blob + f.in_b[i];
                a.nseg = ks; a.interleave = ks; a.nch_total = p.nch_in; a.MB = p.mb_in;
                for (int t = 0; t < ks; ++t) a.seg[t] = {w.x, cstride, p.nch_c, (t - ks / 2) * dil, 0, 0};
                a.dst0 = w.act; a.dst0_bstride = cstride;
                a.M = 2 * C; a.pairC = C;
                // frame-rate conditioning rows of this layer, interpolated to sample rate inside the GATE epilogue
                // (glow_ax.py:362-373, 389-390); read as they are when the rates already agree
                a.addend = fr + (size_t)i * 2 * C * cond_ld;
                a.addend_bstride = (long long)2 * C * nl * cond_ld;
                a.addend_ld = cond_ld; a.addend_pad = cond_pad;
                a.addend_frames = frames == L ? 0 : frames;
                a.gate = p.c.gated_unit;
                if ((rc = launch_gemm_f32(GEMM_EPI_GATE, a, s))) return rc;
            }
            {
                // merge_res_skip: every layer's C rows are skip rows and x stays the `start` output (glow_ax.py:401-416)
                const bool last = i == nl - 1 || p.c.merge_res_skip;
                GemmArgs a = base_args();
                a.A = blob + f.rs_A[i]; a.bias = blob + f.rs_b[i];
                a.nseg = 1; a.nch_total = p.nch_c; a.MB = p.rs_mb(i);
                a.seg[0] = {w.act, cstride, p.nch_c, 0, 0, 0};
                a.M = p.rs_rows(i);
                a.dst0 = w.x; a.dst0_bstride = cstride; a.acc0 = 1;
                a.dst1 = w.out; a.dst1_bstride = cstride; a.acc1 = i > 0 ? 1 : 0;
                a.split = last ? 0 : C;
                if ((rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s))) return rc;
            }
        }
        if ((rc = boundary(k, k - 1))) return rc;
    }
    hipLaunchKernelGGL(ax_unsqueeze_kernel, lgrid, dim3(256), 0, s, w.audio, audio, G, L, g.ld, g.pad);
    CTTS_CHECK_LAUNCH("ax_unsqueeze");
    return CTTS_OK;
}

}  // extern "C"
