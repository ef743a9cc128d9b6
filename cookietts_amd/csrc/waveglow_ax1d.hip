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
//   [un-mix]      (mix_first=False: ax:324-325)              ax_mix_kernel, in place on the latent rows
//   start 1x1                                                  ax_start_kernel
//   conditioning  frame-rate rows, linearly interpolated       inside the GATE epilogue (glow_ax.py:362-373, 389-390)
//   n_layers x (dilated conv + cond + gate | res/skip 1x1)    conv_gemm_f32<GATE> with the conditioning as
//                                                             interpolated epilogue addend, conv_gemm_f32<SPLIT>
//   end 1x1                                                    conv_gemm_f32<SPLIT>, M = 2*n_half
//   coupling inverse, NaN -> 0, [un-mix if mix_first]          ax_couple_kernel (em:100-104, ax:333-337)
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
    int C, nch_in, nch_c, mb_in, mb_end;
    std::vector<AxFlowDims> fd;
    struct Flow {
        size_t start_w, start_b, end_A, end_b, winv;
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
    CTTS_CHECK_ARG(gemm_mode_valid(c.f32_gemm_mode), "f32_gemm_mode=%d (CTTS_GEMM_DEFAULT / _F32 / _BF16X3)", c.f32_gemm_mode);
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
    p.mb_end = 1;
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
        f.end_A = take((size_t)p.mb_end * p.nch_c * A_TILE);
        f.end_b = take((size_t)p.mb_end * GEMM_BM);
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

struct AxWs { float *audio, *x, *act, *out, *e; size_t total; int e_rows; };

void ax_carve(const AxPlan& p, const AxGeom& g, int batch, float* base, AxWs& w) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return base ? base + r : nullptr; };
    const size_t B = batch;
    w.e_rows = p.c.n_group;                                   // 2*n_half <= n_group
    w.audio = take(B * p.c.n_group * g.ld);
    w.x = take(B * p.C * g.ld);
    w.act = take(B * p.C * g.ld);
    w.out = take(B * p.C * g.ld);
    w.e = take(B * w.e_rows * g.ld);
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

struct AxPerm { int src[AX_MAX_GROUP]; };

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

// One thread owns one time step of the n latent rows (n <= 64), held in LDS column `tid` (conflict-free).
//   PERMUTE: out[i] = in[src[i]]                                     (efficient_modules.py:360-373)
//   CONV1X1: out[i] = sum_j Winv[i][j] * in[j], j ascending with fma  (em:283: F.conv1d with W^-1)
__device__ inline void ax_mix_column(const float* col, int n, int mixing, const int* sP, const float* sW, float* dst,
                                     size_t dst_stride) {
    if (mixing == CTTS_MIX_PERMUTE) {
        for (int i = 0; i < n; ++i) dst[(size_t)i * dst_stride] = col[sP[i] * 256];
    } else {
        for (int i = 0; i < n; ++i) {
            float s = 0.f;
            for (int j = 0; j < n; ++j) s = fmaf(sW[i * n + j], col[j * 256], s);
            dst[(size_t)i * dst_stride] = s;
        }
    }
}

__global__ __launch_bounds__(256) void ax_mix_kernel(float* __restrict__ audio, const float* __restrict__ Winv, AxPerm perm,
                                                     int mixing, int G, int ch_off, int n, int L, int ld, int pad) {
    extern __shared__ float smem[];
    float* sW = smem;                       // [n*n] (CONV1X1)
    float* sa = smem + n * n;               // [n][256]
    int* sP = reinterpret_cast<int*>(sa + n * 256);   // [n]  (a kernarg array indexed at run time would go to scratch)
    if (mixing == CTTS_MIX_CONV1X1)
        for (int i = threadIdx.x; i < n * n; i += 256) sW[i] = Winv[i];
    if (threadIdx.x < AX_MAX_GROUP) {
        int v = 0;
#pragma unroll
        for (int q = 0; q < AX_MAX_GROUP; ++q) v = (threadIdx.x == q) ? perm.src[q] : v;
        if ((int)threadIdx.x < n) sP[threadIdx.x] = v;
    }
    __syncthreads();
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    float* ab = audio + ((size_t)b * G + ch_off) * ld + pad + l;
    float* col = sa + threadIdx.x;
    for (int j = 0; j < n; ++j) col[j * 256] = ab[(size_t)j * ld];
    ax_mix_column(col, n, mixing, sP, sW, ab, (size_t)ld);
}

// x[b][c][pad + l] = bs[c] + sum_{j < h} Ws[c][j] * audio[b][ch_off + j][pad + l]   (glow_ax.py:376 `start`)
// grid (L/256, C/8, B): a thread keeps 8 channels of one time step; the h input rows are re-read per channel group
// (h*4 bytes per step, L2-resident) - the stage is bound by the C*4 bytes per step it writes.
__global__ __launch_bounds__(256) void ax_start_kernel(const float* __restrict__ audio, const float* __restrict__ Ws,
                                                       const float* __restrict__ bs, float* __restrict__ x, int C, int G,
                                                       int ch_off, int h, int L, int ld, int pad) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * 8, b = blockIdx.z;
    if (l >= L) return;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = bs[c0 + q];
    const float* ab = audio + ((size_t)b * G + ch_off) * ld + pad + l;
    for (int j = 0; j < h; ++j) {
        const float a = ab[(size_t)j * ld];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = fmaf(Ws[(c0 + q) * h + j], a, acc[q]);
    }
    float* xb = x + ((size_t)b * C + c0) * ld + pad + l;
#pragma unroll
    for (int q = 0; q < 8; ++q) xb[(size_t)q * ld] = acc[q];
}

// (log_s, t) = (e[:h], e[h:]);  a1 = (a1 - t) / exp(log_s)   (efficient_modules.py:100-103; note the order,
// glow.py has (b, log_s));  NaN -> 0 on the whole latent (ax:13-16, 333-334);  then, for mix_first, the un-mix.
__global__ __launch_bounds__(256) void ax_couple_kernel(float* __restrict__ audio, const float* __restrict__ e,
                                                        const float* __restrict__ Winv, AxPerm perm, int mixing,
                                                        int mix_here, int ignore_nan, int G, int e_rows, int ch_off, int h,
                                                        int L, int ld, int pad) {
    extern __shared__ float smem[];
    const int n = 2 * h;
    float* sW = smem;
    float* sa = smem + n * n;
    int* sP = reinterpret_cast<int*>(sa + n * 256);
    if (mix_here && mixing == CTTS_MIX_CONV1X1)
        for (int i = threadIdx.x; i < n * n; i += 256) sW[i] = Winv[i];
    if (threadIdx.x < AX_MAX_GROUP) {
        int v = 0;
#pragma unroll
        for (int q = 0; q < AX_MAX_GROUP; ++q) v = (threadIdx.x == q) ? perm.src[q] : v;
        if ((int)threadIdx.x < n) sP[threadIdx.x] = v;
    }
    __syncthreads();
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= L) return;
    float* ab = audio + ((size_t)b * G + ch_off) * ld + pad + l;
    const float* eb = e + (size_t)b * e_rows * ld + pad + l;
    float* col = sa + threadIdx.x;
    for (int j = 0; j < h; ++j) {
        float a0 = ab[(size_t)j * ld];
        float a1 = (ab[(size_t)(h + j) * ld] - eb[(size_t)(h + j) * ld]) / expf(eb[(size_t)j * ld]);
        if (ignore_nan) { a0 = (a0 != a0) ? 0.f : a0; a1 = (a1 != a1) ? 0.f : a1; }
        col[j * 256] = a0;
        col[(h + j) * 256] = a1;
    }
    if (mix_here) {
        ax_mix_column(col, n, mixing, sP, sW, ab, (size_t)ld);
    } else {
        for (int j = 0; j < n; ++j) ab[(size_t)j * ld] = col[j * 256];
    }
}

// x[b][c][pad - halo .. pad) = x[b][c][pad];  x[b][c][pad + T .. pad + T + halo) = x[b][c][pad + T - 1]
__global__ __launch_bounds__(64) void replicate_halo_kernel(float* __restrict__ x, int T, int ld, int pad, int halo) {
    float* row = x + (size_t)blockIdx.x * ld + pad;
    const int i = threadIdx.x;
    if (i < halo) { row[-1 - i] = row[0]; row[T + i] = row[T - 1]; }
}

size_t mix_smem(int n) { return ((size_t)n * n + (size_t)n * 256 + (size_t)n) * sizeof(float); }

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
    // end: dense [2h][C] -> one 256-row M-block, rows >= 2h zero
    if ((rc = launch_pack_a(blob + f.end_A, w->end_w, GEMM_BM, p.mb_end, p.nch_c, 0, C, GEMM_EPI_SPLIT, C, 2 * d.n_half, 0,
                            C, 1, s))) return rc;
    if ((rc = launch_pack_bias(blob + f.end_b, GEMM_BM, p.mb_end, w->end_b, 0, nullptr, 0, GEMM_EPI_SPLIT, C,
                               2 * d.n_half, s))) return rc;
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
    for (int k = p.c.n_flows - 1; k >= 0; --k) {
        const auto& f = p.fl[k];
        const auto& d = p.fd[k];
        AxPerm perm{};
        if (p.c.mixing == CTTS_MIX_PERMUTE) ax_permutation(k, d.n_rem, perm.src);
        if (!p.c.mix_first) {                                                   // ax:324-325
            hipLaunchKernelGGL(ax_mix_kernel, lgrid, dim3(256), mix_smem(d.n_rem), s, w.audio, blob + f.winv, perm,
                               p.c.mixing, G, d.ch_off, d.n_rem, L, g.ld, g.pad);
            CTTS_CHECK_LAUNCH("ax_mix");
        }
        hipLaunchKernelGGL(ax_start_kernel, dim3((L + 255) / 256, C / 8, batch), dim3(256), 0, s, w.audio, blob + f.start_w,
                           blob + f.start_b, w.x, C, G, d.ch_off, d.n_half, L, g.ld, g.pad);
        CTTS_CHECK_LAUNCH("ax_start");
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
        {   // end 1x1: out [C] -> e [2h] (rows >= 2h of the M-block are padding, never stored)
            GemmArgs a = base_args();
            a.A = blob + f.end_A; a.bias = blob + f.end_b;
            a.nseg = 1; a.nch_total = p.nch_c; a.MB = p.mb_end;
            a.seg[0] = {w.out, cstride, p.nch_c, 0, 0, 0};
            a.M = 2 * d.n_half;
            a.dst0 = w.e; a.dst0_bstride = (long long)w.e_rows * g.ld;
            a.dst1 = w.e; a.dst1_bstride = (long long)w.e_rows * g.ld;
            a.split = 0;
            if ((rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s))) return rc;
        }
        hipLaunchKernelGGL(ax_couple_kernel, lgrid, dim3(256), mix_smem(d.n_rem), s, w.audio, w.e, blob + f.winv, perm,
                           p.c.mixing, p.c.mix_first ? 1 : 0, p.c.ignore_nan ? 1 : 0, G, w.e_rows, d.ch_off, d.n_half, L,
                           g.ld, g.pad);
        CTTS_CHECK_LAUNCH("ax_couple");
    }
    hipLaunchKernelGGL(ax_unsqueeze_kernel, lgrid, dim3(256), 0, s, w.audio, audio, G, L, g.ld, g.pad);
    CTTS_CHECK_LAUNCH("ax_unsqueeze");
    return CTTS_OK;
}

}  // extern "C"
