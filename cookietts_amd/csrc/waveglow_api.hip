// C ABI of the WaveGlow (glow.py topology) mel->wave path: packed-weight layout, workspace
// layout and the launch sequence.  See include/cookietts_hip.h for the contract.
#include <mutex>
#include <vector>

#include "gemm_bf16.h"
#include "gemm_f32.h"
#include "tuning.h"
#include "waveglow_kernels.h"

namespace ctts {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

constexpr size_t ALIGN_F = 64;  // section alignment in floats (256 B)
inline size_t align_up(size_t v) { return (v + ALIGN_F - 1) / ALIGN_F * ALIGN_F; }
constexpr int A_TILE = GEMM_KC * GEMM_BM;

struct FlowDims { int n_rem, n_half, ch_off; };

struct Plan {
    ctts_waveglow_config c;
    int C, H, K0, n_in;                 // WN channels, cond hidden, n_mel*G, flows
    int S;                              // speaker-embedding rows per flow, padded to a multiple of 32 (0: single speaker)
    int nch0, nch1h, nch_in, nch_rs;    // K chunks: cond0, cond1, in-layer, res/skip
    int mb_in;                          // M-blocks of the in-layer GEMM
    std::vector<FlowDims> fd;
    // packed blob offsets (floats)
    size_t up_w, up_wp, up_b, cond0_A, cond0_b, cond1_A, cond1_b;
    bool up_mfma;
    struct Flow {
        size_t start_w, start_b, end_w, end_b, winv, spk_tab;
        std::vector<size_t> in_A, in_b, rs_A, rs_b;
        // deferred-skip form (round 4, profiles/r4_07_res_skip_experiments.txt): res rows per layer, skip rows of a group of
        // up to F32_SKIP_GROUP layers side by side along K
        std::vector<size_t> res_A, res_b, skip_A, skip_b;
    };
    std::vector<Flow> fl;
    size_t total;

    static constexpr int F32_SKIP_GROUP = 4;
    int mb_c() const { return (C + GEMM_BM - 1) / GEMM_BM; }
    int n_groups() const { return (c.n_layers + F32_SKIP_GROUP - 1) / F32_SKIP_GROUP; }
    int group_layers(int g) const { return g + 1 < n_groups() ? F32_SKIP_GROUP : c.n_layers - g * F32_SKIP_GROUP; }
    int rs_rows(int layer) const { return layer < c.n_layers - 1 ? 2 * C : C; }
    int rs_mb(int layer) const { return (rs_rows(layer) + GEMM_BM - 1) / GEMM_BM; }
};

int make_plan(const ctts_waveglow_config* cfg, Plan& p) {
    CTTS_CHECK_ARG(cfg != nullptr, "config is NULL");
    p.c = *cfg;
    const auto& c = p.c;
    CTTS_CHECK_ARG(c.n_flows >= 1 && c.n_layers >= 1 && c.n_layers <= 12, "n_flows=%d n_layers=%d", c.n_flows, c.n_layers);
    CTTS_CHECK_ARG(gemm_mode_valid(c.f32_gemm_mode), "f32_gemm_mode=%d (CTTS_GEMM_DEFAULT / _F32 / _BF16X3 / _BF16X6)", c.f32_gemm_mode);
    CTTS_CHECK_ARG(c.n_group >= 4 && c.n_group % 4 == 0 && c.n_group <= 16, "n_group=%d (4, 8, 12 or 16)", c.n_group);
    CTTS_CHECK_ARG(c.kernel_size == 3, "kernel_size=%d (only 3 built)", c.kernel_size);
    CTTS_CHECK_ARG(c.n_channels >= 128 && c.n_channels % 128 == 0, "n_channels=%d (multiple of 128)", c.n_channels);
    CTTS_CHECK_ARG(c.cond_hidden == GEMM_BM, "cond_hidden=%d (reference hard-codes 256)", c.cond_hidden);
    CTTS_CHECK_ARG(c.hop_length > 0 && c.win_length % c.hop_length == 0 && c.hop_length % c.n_group == 0,
                   "win=%d hop=%d n_group=%d", c.win_length, c.hop_length, c.n_group);
    CTTS_CHECK_ARG((c.n_mel_channels * c.n_group) % GEMM_KC == 0, "n_mel*n_group=%d not a multiple of 16",
                   c.n_mel_channels * c.n_group);
    CTTS_CHECK_ARG(c.n_early_every >= 1 && c.n_early_size >= 0 && c.n_early_size % 2 == 0, "early outputs");
    CTTS_CHECK_ARG(c.speaker_embed_dim >= 0 && c.speaker_embed_dim <= 1024, "speaker_embed_dim=%d", c.speaker_embed_dim);
    p.C = c.n_channels;
    p.H = c.cond_hidden;
    p.K0 = c.n_mel_channels * c.n_group;
    p.S = round_up(c.speaker_embed_dim, 32);
    CTTS_CHECK_ARG(p.S == 0 || p.K0 % 32 == 0, "multispeaker needs n_mel*n_group %% 32 == 0");
    p.nch0 = (p.K0 + p.S) / GEMM_KC;
    p.nch1h = p.H / GEMM_KC;
    p.nch_in = (c.kernel_size * p.C + p.H) / GEMM_KC;
    p.nch_rs = p.C / GEMM_KC;
    p.mb_in = 2 * p.C / GEMM_BM;
    // per-flow channel counts exactly as glow.py:251-265
    int n_half = c.n_group / 2, n_rem = c.n_group;
    p.fd.resize(c.n_flows);
    for (int k = 0; k < c.n_flows; ++k) {
        if (k % c.n_early_every == 0 && k > 0) { n_half -= c.n_early_size / 2; n_rem -= c.n_early_size; }
        CTTS_CHECK_ARG(n_half >= 1 && n_rem == 2 * n_half, "flow %d: n_half=%d n_remaining=%d", k, n_half, n_rem);
        p.fd[k] = {n_rem, n_half, c.n_group - n_rem};
    }
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    p.up_w = take((size_t)c.n_mel_channels * c.n_mel_channels * c.win_length);
    // the same weights in MFMA fragment order, for the shape the MFMA upsampling kernel is built for (waveglow_kernels.hip)
    p.up_mfma = upsample_mfma_shape(c.n_mel_channels, c.win_length, c.hop_length, c.n_group);
    p.up_wp = p.up_mfma ? take((size_t)c.n_mel_channels * c.n_mel_channels * c.win_length) : 0;
    p.up_b = take(c.n_mel_channels);
    p.cond0_A = take((size_t)c.n_flows * p.nch0 * A_TILE);
    p.cond0_b = take((size_t)c.n_flows * GEMM_BM);
    p.cond1_A = take((size_t)c.n_flows * p.nch1h * A_TILE);
    p.cond1_b = take((size_t)c.n_flows * GEMM_BM);
    p.fl.resize(c.n_flows);
    for (int k = 0; k < c.n_flows; ++k) {
        auto& f = p.fl[k];
        f.start_w = take((size_t)p.C * p.fd[k].n_half);
        f.start_b = take(p.C);
        f.end_w = take((size_t)2 * p.fd[k].n_half * p.C);
        f.end_b = take(2 * p.fd[k].n_half);
        f.winv = take((size_t)p.fd[k].n_rem * p.fd[k].n_rem);
        f.spk_tab = take((size_t)CTTS_N_SPEAKERS * c.speaker_embed_dim);
        for (int i = 0; i < c.n_layers; ++i) {
            f.in_A.push_back(take((size_t)p.mb_in * p.nch_in * A_TILE));
            f.in_b.push_back(take((size_t)p.mb_in * GEMM_BM));
            f.rs_A.push_back(take((size_t)p.rs_mb(i) * p.nch_rs * A_TILE));
            f.rs_b.push_back(take((size_t)p.rs_mb(i) * GEMM_BM));
            f.res_A.push_back(i + 1 < c.n_layers ? take((size_t)p.mb_c() * p.nch_rs * A_TILE) : 0);
            f.res_b.push_back(i + 1 < c.n_layers ? take((size_t)p.mb_c() * GEMM_BM) : 0);
        }
        for (int gi = 0; gi < p.n_groups(); ++gi) {
            f.skip_A.push_back(take((size_t)p.mb_c() * p.group_layers(gi) * p.nch_rs * A_TILE));
            f.skip_b.push_back(take((size_t)p.mb_c() * GEMM_BM));
        }
    }
    p.total = o;
    return CTTS_OK;
}

struct Geom { int L, ld, pad, ntiles; };

int make_geom(const Plan& p, int frames, Geom& g) {
    CTTS_CHECK_ARG(frames >= 1, "frames=%d", frames);
    const long long T = (long long)frames * p.c.hop_length;
    g.L = (int)(T / p.c.n_group);
    int maxd = 1 << (p.c.n_layers - 1);
    g.pad = round_up(maxd > 128 ? maxd : 128, 32);
    g.ntiles = (g.L + GEMM_BN - 1) / GEMM_BN;
    g.ld = round_up(g.L, 256) + 2 * g.pad;          // room for the 256-wide tiles of the bf16 GEMM
    return CTTS_OK;
}

struct Workspace {
    float *audio, *spect, *spk, *h_tmp, *h_all, *x, *act, *out;
    float* act_all;     // deferred-skip form: the gated activations of every layer of a flow ([n_layers] x act)
    size_t total;  // floats
};

void carve(const Plan& p, const Geom& g, int batch, float* base, Workspace& w) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return base ? base + r : nullptr; };
    const size_t B = batch;
    w.audio = take(B * p.c.n_group * g.L);
    w.spect = take(B * p.K0 * g.ld);
    w.spk = take(B * p.c.n_flows * p.S * g.ld);          // speaker-embedding rows (glow.py:193-196), per flow
    w.h_tmp = take(B * p.c.n_flows * p.H * g.ld);
    w.h_all = take(B * p.c.n_flows * p.H * g.ld);
    w.x = take(B * p.C * g.ld);
    w.act = take(B * p.C * g.ld);
    w.out = take(B * p.C * g.ld);
    // gated activations of ONE skip group (its skip GEMM runs at the end of the group; the next group reuses the slots)
    w.act_all = take((size_t)std::min(p.c.n_layers, (int)Plan::F32_SKIP_GROUP) * B * p.C * g.ld);
    w.total = o;
}

// ---- profiling hooks ---------------------------------------------------------------
// A profile is a caller-owned handle (ctts_profile_create); a thread records into the handle it bound
// (ctts_profile_bind), so two threads timing two models at once keep disjoint slots.  No process-wide switch.
struct Prof {
    std::mutex mu;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[CTTS_PROF_N];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};
thread_local Prof* t_prof = nullptr;

struct ProfScope {
    int which; hipStream_t s; hipEvent_t stop = nullptr; bool active = false;
    ProfScope(int which_, hipStream_t s_) : which(which_), s(s_) {
        Prof* pr = t_prof;
        if (!pr) return;
        std::lock_guard<std::mutex> lk(pr->mu);
        std::pair<hipEvent_t, hipEvent_t> e;
        if (!pr->pool.empty()) { e = pr->pool.back(); pr->pool.pop_back(); }
        else { if (hipEventCreate(&e.first) != hipSuccess || hipEventCreate(&e.second) != hipSuccess) return; }
        (void)hipEventRecord(e.first, s);
        stop = e.second;
        pr->ev[which].push_back(e);
        active = true;
    }
    ~ProfScope() { if (active) (void)hipEventRecord(stop, s); }
};

// ---- stage launchers ---------------------------------------------------------------
GemmArgs base_args(const Plan& p, const Geom& g, int batch) {
    GemmArgs a{};
    a.gemm_mode = p.c.f32_gemm_mode;
    a.ld = g.ld; a.pad = g.pad; a.L = g.L; a.ntiles = g.ntiles; a.batch = batch;
    a.dst_ld = g.ld; a.dst_pad = g.pad;
    return a;
}

// spk[b][k*S + e][pad + l] = table_k[ids[b]][e]  (e < sdim), zero rows above: the reference concatenates the speaker
// embedding, repeated over time, under the spectrogram (glow.py:193-196); here it is the second K segment of cond layer 0
__global__ __launch_bounds__(256) void speaker_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                           float* __restrict__ spk, int k, int n_flows, int S, int sdim,
                                                           int L, int ld, int pad) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y, b = blockIdx.z;
    if (n >= ld) return;
    float v = 0.f;
    if (e < sdim && n >= pad && n < pad + L) {
        const long long id = ids[b];      // an id outside the table poisons the utterance (NaN) instead of reading out of bounds
        v = (id >= 0 && id < CTTS_N_SPEAKERS) ? table[(size_t)id * sdim + e] : __builtin_nanf("");
    }
    spk[((size_t)b * n_flows * S + (size_t)k * S + e) * ld + n] = v;
}

int fill_speaker_rows(const Plan& p, const Geom& g, const float* blob, const int64_t* ids, float* spk, int batch,
                      hipStream_t s) {
    if (p.S == 0) return CTTS_OK;
    CTTS_CHECK_ARG(ids != nullptr, "multispeaker model (speaker_embed_dim=%d) needs speaker ids (glow.py:193)",
                   p.c.speaker_embed_dim);
    for (int k = 0; k < p.c.n_flows; ++k)
        hipLaunchKernelGGL(speaker_rows_kernel, dim3((g.ld + 255) / 256, p.S, batch), dim3(256), 0, s,
                           blob + p.fl[k].spk_tab, ids, spk, k, p.c.n_flows, p.S, p.c.speaker_embed_dim, g.L, g.ld, g.pad);
    CTTS_CHECK_LAUNCH("speaker_rows");
    return CTTS_OK;
}

int run_cond(const Plan& p, const Geom& g, const float* blob, const float* spect, const float* spk, float* h_tmp,
             float* h_all, int batch, hipStream_t s) {
    const long long hstride = (long long)p.c.n_flows * p.H * g.ld;
    GemmArgs a = base_args(p, g, batch);
    a.A = blob + p.cond0_A; a.bias = blob + p.cond0_b;
    a.nseg = 1; a.nch_total = p.nch0; a.MB = p.c.n_flows;
    a.seg[0] = {spect, (long long)p.K0 * g.ld, p.K0 / GEMM_KC, 0, 0, 0};
    if (p.S) {   // flow k (= M-block k) reads its own S embedding rows
        CTTS_CHECK_ARG(spk != nullptr, "wn_cond: multispeaker model needs the speaker rows");
        a.nseg = 2;
        a.seg[1] = {spk, (long long)p.c.n_flows * p.S * g.ld, p.S / GEMM_KC, 0, p.S, 0};
    }
    a.dst0 = h_tmp; a.dst0_bstride = hstride; a.acc0 = 0;
    a.dst1 = h_tmp; a.dst1_bstride = hstride; a.acc1 = 0;
    a.split = p.c.n_flows * GEMM_BM;
    a.M = p.c.n_flows * GEMM_BM;
    int rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s);
    if (rc) return rc;
    a.A = blob + p.cond1_A; a.bias = blob + p.cond1_b;
    a.nseg = 1; a.nch_total = p.nch1h;
    a.seg[0] = {h_tmp, hstride, p.nch1h, 0, GEMM_BM, 0};
    a.dst0 = h_all; a.dst1 = h_all;
    return launch_gemm_f32(GEMM_EPI_SPLIT, a, s);
}

int run_wn_stack(const Plan& p, const Geom& g, const float* blob, int k, const float* audio, const float* h_all,
                 float* x, float* act, float* out, int batch, hipStream_t s, float* act_all = nullptr) {
    const auto& f = p.fl[k];
    const auto& d = p.fd[k];
    const long long cstride = (long long)p.C * g.ld;
    const long long hstride = (long long)p.c.n_flows * p.H * g.ld;
    int rc = launch_wn_start(audio, blob + f.start_w, blob + f.start_b, x, batch, p.C, p.c.n_group, d.ch_off,
                             d.n_half, g.L, g.ld, g.pad, s);
    if (rc) return rc;
    const int ncx = p.C / GEMM_KC;
    // Deferred-skip form (default since round 4; CTTS_F32_NO_DEFER_SKIP = the per-layer form; the fp32 analogue of the bf16 path's): every layer's gated activation is
    // kept, the per-layer launch computes the res rows only (x += W_res act), and the skip rows of up to four layers are
    // ONE contraction with K = 4 C at the end of the group.  Same products, different summation order of the skip sum.
    const bool defer = act_all != nullptr && !tuning().f32_no_defer_skip;
    const size_t act_stride = (size_t)batch * p.C * g.ld;
    for (int i = 0; i < p.c.n_layers; ++i) {
        const int dil = 1 << i;
        if (defer) act = act_all + (size_t)(i % Plan::F32_SKIP_GROUP) * act_stride;
        {
            GemmArgs a = base_args(p, g, batch);
            a.A = blob + f.in_A[i]; a.bias = blob + f.in_b[i];
            a.nseg = 4; a.interleave = 3; a.nch_total = p.nch_in; a.MB = p.mb_in;
            a.seg[0] = {x, cstride, ncx, -dil, 0, 0};
            a.seg[1] = {x, cstride, ncx, 0, 0, 0};
            a.seg[2] = {x, cstride, ncx, dil, 0, 0};
            a.seg[3] = {h_all + (size_t)k * p.H * g.ld, hstride, p.nch1h, 0, 0, 0};
            a.dst0 = act; a.dst0_bstride = cstride;
            a.M = 2 * p.C; a.pairC = p.C;
            ProfScope ps(CTTS_PROF_WN_IN, s);
            rc = launch_gemm_f32(GEMM_EPI_GATE, a, s);
            if (rc) return rc;
        }
        if (defer) {
            const bool last = i == p.c.n_layers - 1;
            if (!last) {
                GemmArgs a = base_args(p, g, batch);
                a.A = blob + f.res_A[i]; a.bias = blob + f.res_b[i];
                a.nseg = 1; a.nch_total = p.nch_rs; a.MB = p.mb_c();
                a.seg[0] = {act, cstride, p.nch_rs, 0, 0, 0};
                a.M = p.C; a.split = p.C;
                a.dst0 = x; a.dst0_bstride = cstride; a.acc0 = 1;
                a.dst1 = out; a.dst1_bstride = cstride; a.acc1 = 0;
                ProfScope ps(CTTS_PROF_WN_RS, s);
                rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s);
                if (rc) return rc;
            }
            const int gi = i / Plan::F32_SKIP_GROUP;
            if (last || (i + 1) % Plan::F32_SKIP_GROUP == 0) {
                const int nl = p.group_layers(gi);
                GemmArgs a = base_args(p, g, batch);
                a.A = blob + f.skip_A[gi]; a.bias = blob + f.skip_b[gi];
                a.nseg = nl; a.nch_total = nl * p.nch_rs; a.MB = p.mb_c();
                for (int j = 0; j < nl; ++j)
                    a.seg[j] = {act_all + (size_t)j * act_stride, cstride, p.nch_rs, 0, 0, 0};
                a.M = p.C; a.split = 0;
                a.dst0 = x; a.dst0_bstride = cstride; a.acc0 = 1;
                a.dst1 = out; a.dst1_bstride = cstride; a.acc1 = gi > 0 ? 1 : 0;
                ProfScope ps(CTTS_PROF_WN_SKIP, s);
                rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s);
                if (rc) return rc;
            }
        } else {
            const bool last = i == p.c.n_layers - 1;
            GemmArgs a = base_args(p, g, batch);
            a.A = blob + f.rs_A[i]; a.bias = blob + f.rs_b[i];
            a.nseg = 1; a.nch_total = p.nch_rs; a.MB = p.rs_mb(i);
            a.seg[0] = {act, cstride, p.nch_rs, 0, 0, 0};
            a.M = p.rs_rows(i);
            a.dst0 = x; a.dst0_bstride = cstride; a.acc0 = 1;
            a.dst1 = out; a.dst1_bstride = cstride; a.acc1 = i > 0 ? 1 : 0;
            a.split = last ? 0 : p.C;
            ProfScope ps(CTTS_PROF_WN_RS, s);
            rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s);
            if (rc) return rc;
        }
    }
    return CTTS_OK;
}

int run_flow_tail(const Plan& p, const Geom& g, const float* blob, int k, const float* out, float* audio,
                  float* wave, int batch, hipStream_t s) {
    const auto& f = p.fl[k];
    const auto& d = p.fd[k];
    return launch_flow_tail(out, audio, wave, blob + f.end_w, blob + f.end_b, blob + f.winv, batch, p.C,
                            p.c.n_group, d.ch_off, d.n_half, g.L, g.ld, g.pad, s);
}

// ======================================================================================
// bf16 variant (BASELINE config 3): WN in-layer / res-skip GEMMs on bf16 MFMA with fp32 accumulation,
// WN activations (x, act, skip sum, cond hidden) stored bf16 in the K8-blocked layout; upsampling, the
// two small cond layers, the `end` conv, the coupling and the inverse 1x1 conv stay fp32.
// The res and skip halves of res_skip_layers are separate GEMMs here: the res rows update x after every layer
// (x feeds the next layer), the skip rows are deferred - every layer's gated activation stays in HBM and the skip
// sum is ONE contraction over K = n_layers * C at the end of the stack, BF_SKIP_GROUP layers per launch.  The
// per-layer form re-read and re-wrote the skip sum (2 x B*C*L bf16) in all but one of its n_layers epilogues.
constexpr int BF_SKIP_GROUP = 4;                   // 4 layers x 3 products = BGEMM_MAX_SEG segments in the split form
struct BfPlan {
    std::vector<std::vector<size_t>> in_A, rs_A;   // [flow][layer] offsets in bf16 elements (rs_A: res rows, layers < last)
    std::vector<std::vector<size_t>> skip_A;       // [flow][group]: skip rows of the group's layers side by side along K
    std::vector<size_t> skip_b;                    // [flow]: fp32 [2][mb_c*256] = {sum of the skip biases, zeros}
    size_t cond0_A, cond1_A;                       // flow-batched cond layers 0 / 1 (M-block k = flow k)
    size_t total;
    int nch_in, nch_rs, nch_c0, nch_c1;
    int mb_c, n_groups;                            // M-blocks of a C-row GEMM; skip launches per flow
    int P;                                         // K products per operand pair: 1 = bf16, 3 = split bf16 (hi*hi + lo*hi + hi*lo)
    int f16;                                       // 1: IEEE-half storage and MFMA operands instead of bf16 (P == 1 only)
    int group_layers(int g, int n_layers) const { return std::min(BF_SKIP_GROUP, n_layers - g * BF_SKIP_GROUP); }
};

void make_bf_plan(const Plan& p, BfPlan& q, int P, int f16 = 0) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = (o + n + 127) / 128 * 128; return r; };
    q.P = P;
    q.f16 = f16;
    q.nch_in = P * (p.c.kernel_size * p.C + p.H) / BGEMM_KC;
    q.nch_rs = P * p.C / BGEMM_KC;
    q.nch_c0 = P * (p.K0 + p.S) / BGEMM_KC;
    q.nch_c1 = P * p.H / BGEMM_KC;
    q.cond0_A = take((size_t)p.c.n_flows * q.nch_c0 * BGEMM_KC * BGEMM_BM);
    q.cond1_A = take((size_t)p.c.n_flows * q.nch_c1 * BGEMM_KC * BGEMM_BM);
    q.mb_c = (p.C + BGEMM_BM - 1) / BGEMM_BM;
    q.n_groups = (p.c.n_layers + BF_SKIP_GROUP - 1) / BF_SKIP_GROUP;
    q.in_A.assign(p.c.n_flows, {});
    q.rs_A.assign(p.c.n_flows, {});
    q.skip_A.assign(p.c.n_flows, {});
    for (int k = 0; k < p.c.n_flows; ++k) {
        for (int i = 0; i < p.c.n_layers; ++i) {
            q.in_A[k].push_back(take((size_t)p.mb_in * q.nch_in * BGEMM_KC * BGEMM_BM));
            q.rs_A[k].push_back(i < p.c.n_layers - 1 ? take((size_t)q.mb_c * q.nch_rs * BGEMM_KC * BGEMM_BM) : 0);
        }
        for (int gi = 0; gi < q.n_groups; ++gi)
            q.skip_A[k].push_back(take((size_t)q.mb_c * q.group_layers(gi, p.c.n_layers) * q.nch_rs * BGEMM_KC * BGEMM_BM));
        q.skip_b.push_back(take((size_t)2 * 2 * q.mb_c * BGEMM_BM));        // fp32 pairs of bf16 slots
    }
    q.total = o;
}

// bf16 tensors of the split form are PAIRS of planes: hi at the pointer, lo at pointer + *_lo elements (0 = plain bf16)
struct BfWs {
    float *audio, *spect, *spk;
    bf16_t *spect_bf, *spk_bf, *h_tmp_bf, *h_bf, *x, *act, *out;
    long long spect_lo, spk_lo, h_lo, x_lo, act_lo;
    size_t total_bytes;
};

void carve_bf(const Plan& p, const Geom& g, int batch, char* base, BfWs& w, int P) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = (o + bytes + 255) / 256 * 256; return base ? base + r : nullptr; };
    const size_t B = batch;
    const size_t planes = P == 3 ? 2 : 1;
    w.audio = (float*)take(B * p.c.n_group * g.L * 4);
    w.spect = (float*)take(B * p.K0 * g.ld * 4);
    w.spect_bf = (bf16_t*)take(planes * B * p.K0 * g.ld * 2);
    w.spk = (float*)take(B * p.c.n_flows * p.S * g.ld * 4);
    w.spk_bf = (bf16_t*)take(planes * B * p.c.n_flows * p.S * g.ld * 2);
    w.h_tmp_bf = (bf16_t*)take(planes * B * p.c.n_flows * p.H * g.ld * 2);
    w.h_bf = (bf16_t*)take(planes * B * p.c.n_flows * p.H * g.ld * 2);
    w.x = (bf16_t*)take(planes * B * p.C * g.ld * 2);
    w.act = (bf16_t*)take(planes * B * p.C * g.ld * 2 * std::min(p.c.n_layers, BF_SKIP_GROUP));   // one buffer per layer of a skip group
    w.out = (bf16_t*)take(planes * B * p.C * g.ld * 2);
    const long long on = P == 3 ? 1 : 0;
    w.spect_lo = on * (long long)(B * p.K0 * g.ld);
    w.spk_lo = on * (long long)(B * p.c.n_flows * p.S * g.ld);
    w.h_lo = on * (long long)(B * p.c.n_flows * p.H * g.ld);
    w.x_lo = on * (long long)(B * p.C * g.ld);
    w.act_lo = on * (long long)(B * p.C * g.ld) * std::min(p.c.n_layers, BF_SKIP_GROUP);
    w.total_bytes = o;
}

// K segments of one operand: bf16 = the tensor itself; split bf16 = (hi, lo, hi) against the packed (W_hi, W_hi, W_lo)
int push_segs(BGemmArgs& a, int idx, int P, const bf16_t* base, long long lo_off, long long bstride, int nch, int shift,
              int mb_rows) {
    for (int pi = 0; pi < P; ++pi) a.seg[idx++] = {pi == 1 ? base + lo_off : base, bstride, nch, shift, mb_rows};
    return idx;
}

// fp32 padded [B][rows][ld] -> bf16 K8 [B][rows/8][ld][8]
__device__ __forceinline__ unsigned int pack_bf16x2_residual(float v0, float v1, unsigned int hi) {
    return pack_bf16x2(v0 - __builtin_bit_cast(float, hi << 16), v1 - __builtin_bit_cast(float, hi & 0xffff0000u));
}

__global__ __launch_bounds__(256) void cvt_f32_to_k8_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                            int rows, int ld, long long lo_off, int f16) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int grp = blockIdx.y, b = blockIdx.z;
    if (n >= ld) return;
    const float* s = src + ((size_t)b * rows + (size_t)grp * 8) * ld + n;
    unsigned int pk[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = s[(size_t)(2 * j) * ld], v1 = s[(size_t)(2 * j + 1) * ld];
        pk[j] = f16 ? pack_f16x2(v0, v1) : pack_bf16x2(v0, v1);
        pl[j] = pack_bf16x2_residual(v0, v1, pk[j]);        // (only stored in the split-bf16 form: lo_off != 0)
    }
    bf16_t* d = dst + (((size_t)b * (rows / 8) + grp) * ld + n) * 8;
    *reinterpret_cast<uint4*>(d) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    if (lo_off) *reinterpret_cast<uint4*>(d + lo_off) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}

// x[b][c][n] = bf16(bs[c] + sum_j Ws[c][j] * audio[b][ch_off + j][n]), K8 layout   (glow.py:189)
template <int H>
__global__ __launch_bounds__(256) void wn_start_bf16_kernel(const float* __restrict__ audio, const float* __restrict__ Ws,
                                                            const float* __restrict__ bs, bf16_t* __restrict__ x, int C,
                                                            int G, int ch_off, int L, int ld, int pad, long long lo_off, int f16) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int grp = blockIdx.y, b = blockIdx.z;
    if (n >= L) return;
    float a[H];
#pragma unroll
    for (int j = 0; j < H; ++j) a[j] = audio[((size_t)b * G + ch_off + j) * L + n];
    unsigned int pk[4], pl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = grp * 8 + 2 * q + e;
            float acc = bs[c];
#pragma unroll
            for (int j = 0; j < H; ++j) acc = fmaf(Ws[c * H + j], a[j], acc);
            v[e] = acc;
        }
        pk[q] = f16 ? pack_f16x2(v[0], v[1]) : pack_bf16x2(v[0], v[1]);
        pl[q] = pack_bf16x2_residual(v[0], v[1], pk[q]);
    }
    bf16_t* d = x + (((size_t)b * (C / 8) + grp) * ld + pad + n) * 8;
    *reinterpret_cast<uint4*>(d) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    if (lo_off) *reinterpret_cast<uint4*>(d + lo_off) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}

// end 1x1 conv on the bf16 skip sum + coupling inverse + inverse 1x1 (+ un-squeeze), fp32 math
// FMT: 0 = bf16, 1 = split bf16 (hi + lo planes), 2 = IEEE half - a compile-time choice: with `lo_off` tested at run time the
// compiler put a branch and a full `vmcnt(0)` behind every 16-byte load of the skip sum (0.97 ms per launch at config 3 = 1 TB/s;
// round 5).  Eight groups of loads are in flight before the first is used.
template <int H, int FMT>
__global__ __launch_bounds__(256) void flow_tail_bf16_kernel(const bf16_t* __restrict__ out, float* __restrict__ audio,
                                                             float* __restrict__ wave, const float* __restrict__ Wend,
                                                             const float* __restrict__ bend, const float* __restrict__ Winv,
                                                             int C, int G, int ch_off, int L, int ld, int pad,
                                                             long long lo_off) {
    constexpr int E = 2 * H;
    constexpr int NG = 8;                                    // groups (of 8 channels) per batch of loads
    __shared__ float sW[E * 512 + E * E + E];
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < E * C; i += 256) sW[i] = Wend[i];
    if (threadIdx.x < E * E) sW[E * C + threadIdx.x] = Winv[threadIdx.x];
    if (threadIdx.x < E) sW[E * C + E * E + threadIdx.x] = bend[threadIdx.x];
    __syncthreads();
    if (n >= L) return;
    float e[E];
#pragma unroll
    for (int j = 0; j < E; ++j) e[j] = sW[E * C + E * E + j];
    const uint4* ob = reinterpret_cast<const uint4*>(out) + ((size_t)b * (C / 8)) * ld + pad + n;
    auto accumulate = [&](const uint4& uu, const uint4& ll, int grp) {
        const unsigned int w4[4] = {uu.x, uu.y, uu.z, uu.w}, l4[4] = {ll.x, ll.y, ll.z, ll.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v0, v1;
            if constexpr (FMT == 2) {
                v0 = f16_to_f32((bf16_t)(w4[q] & 0xffff));
                v1 = f16_to_f32((bf16_t)(w4[q] >> 16));
            } else {
                v0 = bf16_to_f32((bf16_t)(w4[q] & 0xffff));
                v1 = bf16_to_f32((bf16_t)(w4[q] >> 16));
                if constexpr (FMT == 1) {
                    v0 += bf16_to_f32((bf16_t)(l4[q] & 0xffff));
                    v1 += bf16_to_f32((bf16_t)(l4[q] >> 16));
                }
            }
            const int c = grp * 8 + 2 * q;
#pragma unroll
            for (int j = 0; j < E; ++j) e[j] = fmaf(sW[j * C + c + 1], v1, fmaf(sW[j * C + c], v0, e[j]));
        }
    };
    auto load_lo = [&](int grp) -> uint4 {
        if constexpr (FMT == 1)
            return *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(ob + (size_t)grp * ld) + lo_off);
        else
            return make_uint4(0u, 0u, 0u, 0u);
    };
    int g0 = 0;
    for (; g0 + NG <= C / 8; g0 += NG) {                     // whole batches: all NG loads issued before the first use
        uint4 u[NG], ul[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            u[k] = ob[(size_t)(g0 + k) * ld];
            ul[k] = load_lo(g0 + k);
        }
#pragma unroll
        for (int k = 0; k < NG; ++k) accumulate(u[k], ul[k], g0 + k);
    }
    for (; g0 < C / 8; ++g0) accumulate(ob[(size_t)g0 * ld], load_lo(g0), g0);   // narrow test widths
    float* ab = audio + ((size_t)b * G + ch_off) * L + n;
    float a[E];
#pragma unroll
    for (int j = 0; j < E; ++j) a[j] = ab[(size_t)j * L];
#pragma unroll
    for (int j = 0; j < H; ++j) a[H + j] = (a[H + j] - e[j]) / expf(e[H + j]);
    float m[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < E; ++j) s = fmaf(sW[E * C + i * E + j], a[j], s);
        m[i] = s;
    }
    if (wave == nullptr) {
#pragma unroll
        for (int i = 0; i < E; ++i) ab[(size_t)i * L] = m[i];
    } else {
        float* wb = wave + (size_t)b * G * L + (size_t)n * G;
#pragma unroll
        for (int i = 0; i < E; ++i) wb[i] = m[i];
    }
}

// dst[0 .. n) (+)= src[0 .. C) for rows < C; first call clears the whole array (incl. the zero half)
__global__ __launch_bounds__(256) void skip_bias_kernel(float* __restrict__ dst, const float* __restrict__ src, int C, int n,
                                                        int first) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = i < C ? src[i] : 0.f;
    dst[i] = first ? v : dst[i] + v;
}

int run_wn_stack_bf16(const Plan& p, const BfPlan& q, const Geom& g, const float* blob, const bf16_t* bblob, int k,
                      const BfWs& w, int batch, hipStream_t s) {
    const auto& f = p.fl[k];
    const auto& d = p.fd[k];
    const long long cstride = (long long)p.C * g.ld;                 // elements per batch item (K8: (C/8)*ld*8)
    const long long hstride = (long long)p.c.n_flows * p.H * g.ld;
    dim3 sgrid((g.L + 255) / 256, p.C / 8, batch);
#define CTTS_BSTART(HH)                                                                                       \
    case HH:                                                                                                  \
        hipLaunchKernelGGL(wn_start_bf16_kernel<HH>, sgrid, dim3(256), 0, s, w.audio, blob + f.start_w,       \
                           blob + f.start_b, w.x, p.C, p.c.n_group, d.ch_off, g.L, g.ld, g.pad, w.x_lo, q.f16); \
        break;
    switch (d.n_half) {
        CTTS_BSTART(1) CTTS_BSTART(2) CTTS_BSTART(3) CTTS_BSTART(4)
        default: set_error("wn_start_bf16: n_half=%d", d.n_half); return CTTS_E_ARG;
    }
#undef CTTS_BSTART
    CTTS_CHECK_LAUNCH("wn_start_bf16");
    const int ncx = p.C / BGEMM_KC, P = q.P, ks = p.c.kernel_size;
    const size_t act_layer = (size_t)batch * cstride;
    int rc;
    const float* skip_b = reinterpret_cast<const float*>(bblob + q.skip_b[k]);
    // skip sum = sum_i W_skip_i act_i + sum_i b_skip_i: K = n_layers * C, fp32 accumulation inside a launch (one launch at the
    // end of each group of BF_SKIP_GROUP layers, over the group's kept activations - the next group reuses their slots), one
    // rounding of the running sum per group of BF_SKIP_GROUP layers
    auto skip_group = [&](int gi) -> int {
        const int nl = q.group_layers(gi, p.c.n_layers);
        BGemmArgs a{};
        a.f16 = q.f16;
        a.ld = g.ld; a.pad = g.pad; a.L = g.L; a.ntiles = g.ntiles; a.batch = batch;
        a.A = bblob + q.skip_A[k][gi]; a.bias = skip_b + (gi == 0 ? 0 : q.mb_c * BGEMM_BM);
        int ns = 0;
        for (int j = 0; j < nl; ++j)
            ns = push_segs(a, ns, P, w.act + (size_t)j * act_layer, w.act_lo, cstride, ncx, 0, 0);
        a.nseg = ns; a.nch_total = nl * q.nch_rs; a.MB = q.mb_c;
        a.M = p.C; a.split = p.C;
        a.dst0 = w.out; a.dst0_bstride = cstride; a.acc0 = gi > 0 ? 1 : 0;
        a.dst1 = w.out; a.dst1_bstride = cstride; a.acc1 = a.acc0;
        a.lo_off = w.x_lo;
        ProfScope ps(CTTS_PROF_WN_SKIP, s);
        return launch_gemm_bf16(BGEMM_EPI_SPLIT, a, s);
    };
    for (int i = 0; i < p.c.n_layers; ++i) {
        const int dil = 1 << i;
        bf16_t* act = w.act + (size_t)(i % BF_SKIP_GROUP) * act_layer;
        {
            BGemmArgs a{};
            a.f16 = q.f16;
            a.ld = g.ld; a.pad = g.pad; a.L = g.L; a.ntiles = g.ntiles; a.batch = batch;
            a.A = bblob + q.in_A[k][i]; a.bias = blob + f.in_b[i];
            int ns = 0;
            for (int t = 0; t < ks; ++t) ns = push_segs(a, ns, P, w.x, w.x_lo, cstride, ncx, (t - ks / 2) * dil, 0);
            a.interleave = ns;                                        // taps (x products) round-robin per 32-channel slab
            ns = push_segs(a, ns, P, w.h_bf + (size_t)k * p.H * g.ld, w.h_lo, hstride, p.H / BGEMM_KC, 0, 0);
            a.nseg = ns; a.nch_total = q.nch_in; a.MB = p.mb_in;
            a.M = 2 * p.C; a.pairC = p.C;
            a.dst0 = act; a.dst0_bstride = cstride; a.lo_off = w.act_lo;
            ProfScope ps(CTTS_PROF_WN_IN, s);
            if ((rc = launch_gemm_bf16(BGEMM_EPI_GATE, a, s))) return rc;
        }
        if (i < p.c.n_layers - 1) {
            // x += W_res act + b_res   (glow.py:213-217; the skip rows wait for the end of the stack)
            BGemmArgs a{};
            a.f16 = q.f16;
            a.ld = g.ld; a.pad = g.pad; a.L = g.L; a.ntiles = g.ntiles; a.batch = batch;
            a.A = bblob + q.rs_A[k][i]; a.bias = blob + f.rs_b[i];
            a.nseg = push_segs(a, 0, P, act, w.act_lo, cstride, ncx, 0, 0); a.nch_total = q.nch_rs; a.MB = q.mb_c;
            a.M = p.C; a.split = p.C;
            a.dst0 = w.x; a.dst0_bstride = cstride; a.acc0 = 1;
            a.dst1 = w.x; a.dst1_bstride = cstride; a.acc1 = 1;
            a.lo_off = w.x_lo;
            ProfScope ps(CTTS_PROF_WN_RS, s);
            if ((rc = launch_gemm_bf16(BGEMM_EPI_SPLIT, a, s))) return rc;
        }
        if (i == p.c.n_layers - 1 || (i + 1) % BF_SKIP_GROUP == 0) {
            if ((rc = skip_group(i / BF_SKIP_GROUP))) return rc;
        }
    }
    return CTTS_OK;
}

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

int ctts_abi_version(void) { return CTTS_ABI_VERSION; }
const char* ctts_last_error(void) { return g_err; }

int ctts_waveglow_geometry_for(const ctts_waveglow_config* cfg, int32_t frames, ctts_waveglow_geometry* out) {
    Plan p; Geom g;
    int rc = make_plan(cfg, p); if (rc) return rc;
    rc = make_geom(p, frames, g); if (rc) return rc;
    CTTS_CHECK_ARG(out != nullptr, "geometry out is NULL");
    out->steps = g.L; out->ld = g.ld; out->pad = g.pad; out->n_remaining = p.fd.back().n_rem;
    return CTTS_OK;
}

int ctts_fold_weightnorm_f32(const float* v, const float* g, float* w, int32_t out_ch, int32_t fan, void* stream) {
    CTTS_CHECK_ARG(v && g && w, "fold_weightnorm: NULL pointer");
    return launch_fold_weightnorm(v, g, w, out_ch, fan, as_stream(stream));
}

size_t ctts_waveglow_packed_bytes(const ctts_waveglow_config* cfg) {
    Plan p;
    if (make_plan(cfg, p)) return 0;
    return p.total * sizeof(float);
}

int ctts_waveglow_pack_upsample(const ctts_waveglow_config* cfg, const float* up_w, const float* up_b, void* packed,
                                void* stream) {
    Plan p;
    int rc = make_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(up_w && up_b && packed, "pack_upsample: NULL pointer");
    float* blob = static_cast<float*>(packed);
    const size_t nw = (size_t)p.c.n_mel_channels * p.c.n_mel_channels * p.c.win_length;
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + p.up_w, up_w, nw * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + p.up_b, up_b, p.c.n_mel_channels * sizeof(float), hipMemcpyDeviceToDevice,
                                  as_stream(stream)));
    if (p.up_mfma) return launch_upsample_pack_mfma(up_w, blob + p.up_wp, as_stream(stream));
    return CTTS_OK;
}

int ctts_waveglow_pack_flow(const ctts_waveglow_config* cfg, int32_t k, const ctts_waveglow_flow_weights* w,
                            void* packed, void* stream) {
    Plan p;
    int rc = make_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(k >= 0 && k < p.c.n_flows, "pack_flow: flow %d", k);
    CTTS_CHECK_ARG(w && packed, "pack_flow: NULL pointer");
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    const auto& f = p.fl[k];
    const auto& d = p.fd[k];
    const int C = p.C, H = p.H, ks = p.c.kernel_size;
    auto d2d = [&](size_t off, const float* src, size_t n) -> int {
        CTTS_CHECK_ARG(src != nullptr, "pack_flow: NULL weight pointer");
        CTTS_CHECK_HIP(hipMemcpyAsync(blob + off, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return CTTS_OK;
    };
    if ((rc = d2d(f.start_w, w->start_w, (size_t)C * d.n_half))) return rc;
    if ((rc = d2d(f.start_b, w->start_b, C))) return rc;
    if ((rc = d2d(f.end_w, w->end_w, (size_t)2 * d.n_half * C))) return rc;
    if ((rc = d2d(f.end_b, w->end_b, 2 * d.n_half))) return rc;
    if ((rc = d2d(f.winv, w->w_inverse, (size_t)d.n_rem * d.n_rem))) return rc;
    // cond layers 0/1: this flow is M-block k of the flow-batched GEMMs
    CTTS_CHECK_ARG(w->cond_w[0] && w->cond_w[1] && w->cond_w[2] && w->cond_b[0] && w->cond_b[1] && w->cond_b[2],
                   "pack_flow: NULL cond weights");
    // cond layer 0 is [H][n_mel*G + speaker_embed_dim] (glow.py:155): spectrogram columns, then the embedding columns
    // (K rows [K0 + sdim, K0 + S) stay zero from the blob's zero fill)
    const int sdim = p.c.speaker_embed_dim;
    if ((rc = launch_pack_a(blob + p.cond0_A + (size_t)k * p.nch0 * A_TILE, w->cond_w[0], GEMM_BM, 1, p.nch0, 0, p.K0,
                            GEMM_EPI_SPLIT, C, H, 0, p.K0 + sdim, 1, s))) return rc;
    if (sdim) {
        if ((rc = launch_pack_a(blob + p.cond0_A + (size_t)k * p.nch0 * A_TILE, w->cond_w[0] + p.K0, GEMM_BM, 1, p.nch0,
                                p.K0, sdim, GEMM_EPI_SPLIT, C, H, 0, p.K0 + sdim, 1, s))) return rc;
        if ((rc = d2d(f.spk_tab, w->speaker_embed, (size_t)CTTS_N_SPEAKERS * sdim))) return rc;
    }
    if ((rc = launch_pack_bias(blob + p.cond0_b + (size_t)k * GEMM_BM, GEMM_BM, 1, w->cond_b[0], 0, nullptr, 0,
                               GEMM_EPI_SPLIT, C, H, s))) return rc;
    if ((rc = launch_pack_a(blob + p.cond1_A + (size_t)k * p.nch1h * A_TILE, w->cond_w[1], GEMM_BM, 1, p.nch1h, 0, H,
                            GEMM_EPI_SPLIT, C, H, 0, H, 1, s))) return rc;
    if ((rc = launch_pack_bias(blob + p.cond1_b + (size_t)k * GEMM_BM, GEMM_BM, 1, w->cond_b[1], 0, nullptr, 0,
                               GEMM_EPI_SPLIT, C, H, s))) return rc;
    for (int i = 0; i < p.c.n_layers; ++i) {
        CTTS_CHECK_ARG(w->in_w[i] && w->in_b[i] && w->rs_w[i] && w->rs_b[i], "pack_flow: NULL layer %d weights", i);
        // in-layer: K = [per 16-channel slab: tap0, tap1, tap2] then [cond];  in_w[i] is [2C][C][ks]
        for (int t = 0; t < ks; ++t)
            if ((rc = launch_pack_a(blob + f.in_A[i], w->in_w[i] + t, GEMM_BM, p.mb_in, p.nch_in, 0, C, GEMM_EPI_GATE, C,
                                    2 * C, 0, (long long)C * ks, ks, s, ks, t))) return rc;
        // cond layer 2 rows [2C*i, 2C*(i+1)) of [2C*n_layers][H]
        if ((rc = launch_pack_a(blob + f.in_A[i], w->cond_w[2], GEMM_BM, p.mb_in, p.nch_in, ks * C, H, GEMM_EPI_GATE, C,
                                2 * C, (long long)2 * C * i, H, 1, s))) return rc;
        if ((rc = launch_pack_bias(blob + f.in_b[i], GEMM_BM, p.mb_in, w->in_b[i], 0, w->cond_b[2], (long long)2 * C * i,
                                   GEMM_EPI_GATE, C, 2 * C, s))) return rc;
        const int rows = p.rs_rows(i);
        if ((rc = launch_pack_a(blob + f.rs_A[i], w->rs_w[i], GEMM_BM, p.rs_mb(i), p.nch_rs, 0, C, GEMM_EPI_SPLIT, C, rows,
                                0, C, 1, s))) return rc;
        if ((rc = launch_pack_bias(blob + f.rs_b[i], GEMM_BM, p.rs_mb(i), w->rs_b[i], 0, nullptr, 0, GEMM_EPI_SPLIT, C, rows,
                                   s))) return rc;
        // deferred-skip form: res rows [0, C) alone; skip rows ([C, 2C), or [0, C) of the last layer) at K offset
        // (member of the group) * C of the group's matrix, skip biases summed per group
        const bool last = i == p.c.n_layers - 1;
        const int gi = i / Plan::F32_SKIP_GROUP, gj = i % Plan::F32_SKIP_GROUP;
        if (!last) {
            if ((rc = launch_pack_a(blob + f.res_A[i], w->rs_w[i], GEMM_BM, p.mb_c(), p.nch_rs, 0, C, GEMM_EPI_SPLIT, C, C,
                                    0, C, 1, s))) return rc;
            if ((rc = launch_pack_bias(blob + f.res_b[i], GEMM_BM, p.mb_c(), w->rs_b[i], 0, nullptr, 0, GEMM_EPI_SPLIT, C, C,
                                       s))) return rc;
        }
        if ((rc = launch_pack_a(blob + f.skip_A[gi], w->rs_w[i], GEMM_BM, p.mb_c(), p.group_layers(gi) * p.nch_rs, gj * C, C,
                                GEMM_EPI_SPLIT, C, C, last ? 0 : C, C, 1, s))) return rc;
        hipLaunchKernelGGL(skip_bias_kernel, dim3((p.mb_c() * GEMM_BM + 255) / 256), dim3(256), 0, s, blob + f.skip_b[gi],
                           w->rs_b[i] + (last ? 0 : C), C, p.mb_c() * GEMM_BM, gj == 0 ? 1 : 0);
        CTTS_CHECK_LAUNCH("skip_bias_f32");
    }
    return CTTS_OK;
}

size_t ctts_waveglow_workspace_bytes(const ctts_waveglow_config* cfg, int32_t batch, int32_t frames) {
    Plan p; Geom g; Workspace w;
    if (make_plan(cfg, p) || make_geom(p, frames, g) || batch < 1) return 0;
    carve(p, g, batch, nullptr, w);
    return w.total * sizeof(float);
}

int ctts_upsample_squeeze_f32(const ctts_waveglow_config* cfg, const void* packed, const float* mel, float* spect,
                              int32_t batch, int32_t frames, void* stream) {
    Plan p; Geom g;
    int rc = make_plan(cfg, p); if (rc) return rc;
    rc = make_geom(p, frames, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && mel && spect && batch >= 1, "upsample_squeeze: bad argument");
    const float* blob = static_cast<const float*>(packed);
    return launch_upsample_squeeze(mel, blob + p.up_w, p.up_mfma ? blob + p.up_wp : nullptr, blob + p.up_b, spect, batch, p.c.n_mel_channels, frames,
                                   p.c.win_length, p.c.hop_length, p.c.n_group, g.ld, g.pad, as_stream(stream));
}

int ctts_wn_cond_f32(const ctts_waveglow_config* cfg, const void* packed, const float* spect, const float* spk_rows,
                     float* h_tmp, float* h_all, int32_t batch, int32_t frames, void* stream) {
    Plan p; Geom g;
    int rc = make_plan(cfg, p); if (rc) return rc;
    rc = make_geom(p, frames, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && spect && h_tmp && h_all && batch >= 1, "wn_cond: bad argument");
    return run_cond(p, g, static_cast<const float*>(packed), spect, spk_rows, h_tmp, h_all, batch, as_stream(stream));
}

int ctts_wn_stack_f32(const ctts_waveglow_config* cfg, const void* packed, int32_t flow, const float* audio,
                      const float* h_all, float* x, float* act, float* out, int32_t batch, int32_t frames,
                      void* stream) {
    Plan p; Geom g;
    int rc = make_plan(cfg, p); if (rc) return rc;
    rc = make_geom(p, frames, g); if (rc) return rc;
    CTTS_CHECK_ARG(flow >= 0 && flow < p.c.n_flows, "wn_stack: flow %d", flow);
    CTTS_CHECK_ARG(packed && audio && h_all && x && act && out && batch >= 1, "wn_stack: bad argument");
    return run_wn_stack(p, g, static_cast<const float*>(packed), flow, audio, h_all, x, act, out, batch,
                        as_stream(stream));
}

int ctts_flow_tail_f32(const ctts_waveglow_config* cfg, const void* packed, int32_t flow, const float* out,
                       float* audio, float* wave, int32_t batch, int32_t frames, void* stream) {
    Plan p; Geom g;
    int rc = make_plan(cfg, p); if (rc) return rc;
    rc = make_geom(p, frames, g); if (rc) return rc;
    CTTS_CHECK_ARG(flow >= 0 && flow < p.c.n_flows, "flow_tail: flow %d", flow);
    CTTS_CHECK_ARG(packed && out && audio && batch >= 1, "flow_tail: bad argument");
    return run_flow_tail(p, g, static_cast<const float*>(packed), flow, out, audio, wave, batch, as_stream(stream));
}

int ctts_waveglow_infer_f32(const ctts_waveglow_config* cfg, const void* packed, const float* mel,
                            const float* z_scaled, float* wave, int32_t batch, int32_t frames, void* workspace,
                            size_t workspace_bytes, void* stream) {
    return ctts_waveglow_infer_spk_f32(cfg, packed, mel, z_scaled, nullptr, wave, batch, frames, workspace,
                                       workspace_bytes, stream);
}

int ctts_waveglow_infer_spk_f32(const ctts_waveglow_config* cfg, const void* packed, const float* mel,
                                const float* z_scaled, const int64_t* speaker_ids, float* wave, int32_t batch,
                                int32_t frames, void* workspace, size_t workspace_bytes, void* stream) {
    Plan p; Geom g; Workspace w;
    int rc = make_plan(cfg, p); if (rc) return rc;
    rc = make_geom(p, frames, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && mel && z_scaled && wave && workspace && batch >= 1, "infer: bad argument");
    carve(p, g, batch, static_cast<float*>(workspace), w);
    if (w.total * sizeof(float) > workspace_bytes) {
        set_error("infer: workspace %zu bytes < required %zu", workspace_bytes, w.total * sizeof(float));
        return CTTS_E_WORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    CTTS_CHECK_HIP(hipMemcpyAsync(w.audio, z_scaled, (size_t)batch * p.c.n_group * g.L * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
    rc = launch_upsample_squeeze(mel, blob + p.up_w, p.up_mfma ? blob + p.up_wp : nullptr, blob + p.up_b, w.spect, batch, p.c.n_mel_channels, frames,
                                 p.c.win_length, p.c.hop_length, p.c.n_group, g.ld, g.pad, s);
    if (rc) return rc;
    rc = fill_speaker_rows(p, g, blob, speaker_ids, w.spk, batch, s);
    if (rc) return rc;
    rc = run_cond(p, g, blob, w.spect, w.spk, w.h_tmp, w.h_all, batch, s);
    if (rc) return rc;
    for (int k = p.c.n_flows - 1; k >= 0; --k) {
        rc = run_wn_stack(p, g, blob, k, w.audio, w.h_all, w.x, w.act, w.out, batch, s, w.act_all);
        if (rc) return rc;
        rc = run_flow_tail(p, g, blob, k, w.out, w.audio, k == 0 ? wave : nullptr, batch, s);
        if (rc) return rc;
    }
    return CTTS_OK;
}

int ctts_last_gemm_loop(void) { return last_gemm_loop(); }
int ctts_tuning_reload(void) { reload_tuning(); return CTTS_OK; }
int ctts_tuning_flags(void) {
    const Tuning t = tuning();
    return (t.f32_no_glds ? 1 : 0) | (t.no_xcd_pair ? 2 : 0) | (t.bf16_no_glds ? 4 : 0) | (t.bf16_no_wide ? 8 : 0) |
           (t.bf16_no_pp ? 16 : 0) | (t.bf16_w4 ? 32 : 0) | (t.bf16_pp_stages == 4 ? 64 : 0) | (t.wf_no_fuse ? 128 : 0) |
           (t.taco_no_fuse ? 256 : 0) | (t.f32_no_small ? 512 : 0) | (t.f32_force_small ? 1024 : 0) | (t.f32_no_splitk ? 2048 : 0) |
           (t.wf_no_vec_interp ? 4096 : 0) | (t.f32_no_defer_skip ? 8192 : 0) | (t.wf_no_region_split ? 16384 : 0) |
           (t.wf_no_row_queue ? 32768 : 0) | (t.wf_row_queue_min >= 0 ? 65536 : 0) | (t.wf_inject_abort ? 131072 : 0) |
           (t.wf_queue_debug ? 262144 : 0) | (t.f32_no_round_split ? 524288 : 0) | (t.bf16_ps ? (1 << 20) : 0) |
           (t.bf16_no_ps ? (1 << 21) : 0) | (t.f32_splitk_w4 ? (1 << 22) : 0) | (t.taco_poll_delay_set ? (1 << 23) : 0) |
           (t.taco_valu ? (1 << 24) : 0) | (t.up_no_mfma ? (1 << 25) : 0);
}

int ctts_profile_create(void** handle) {
    CTTS_CHECK_ARG(handle, "profile_create: NULL");
    *handle = new Prof();
    return CTTS_OK;
}

int ctts_profile_bind(void* handle) {
    t_prof = static_cast<Prof*>(handle);
    return CTTS_OK;
}

int ctts_profile_collect(void* handle, int32_t which, int64_t* launches, double* total_ms) {
    Prof* pr = static_cast<Prof*>(handle);
    CTTS_CHECK_ARG(pr && which >= 0 && which < CTTS_PROF_N && launches && total_ms, "profile_collect: bad argument");
    std::lock_guard<std::mutex> lk(pr->mu);
    double tot = 0.0;
    int64_t n = 0;
    for (auto& e : pr->ev[which]) {
        float ms = 0.f;
        CTTS_CHECK_HIP(hipEventSynchronize(e.second));
        CTTS_CHECK_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        tot += ms; ++n;
        pr->pool.push_back(e);
    }
    pr->ev[which].clear();
    *launches = n; *total_ms = tot;
    return CTTS_OK;
}

int ctts_profile_destroy(void* handle) {
    Prof* pr = static_cast<Prof*>(handle);
    if (!pr) return CTTS_OK;
    if (t_prof == pr) t_prof = nullptr;
    {
        std::lock_guard<std::mutex> lk(pr->mu);
        for (int w = 0; w < CTTS_PROF_N; ++w)
            for (auto& e : pr->ev[w]) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto& e : pr->pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }
    delete pr;
    return CTTS_OK;
}

// ---- bf16 MFMA variants: P = 1 (bf16 storage, config 3) and P = 3 (split bf16: every operand is a hi + lo pair of
// bf16 planes, every contraction three bf16 MFMA products hi*hi + lo*hi + hi*lo with fp32 accumulation - inputs carry
// 16 mantissa bits instead of 8, at a third of the bf16 rate) ------------------------------------------------------
static size_t packed_bf16_bytes_impl(const ctts_waveglow_config* cfg, int P) {   // (the same for f16)
    Plan p; BfPlan q;
    if (make_plan(cfg, p)) return 0;
    if (p.C % BGEMM_KC != 0 || p.H % BGEMM_KC != 0) { set_error("bf16: channels must be multiples of 32"); return 0; }
    make_bf_plan(p, q, P);
    return q.total * sizeof(bf16_t);
}

static int pack_flow_bf16_impl(const ctts_waveglow_config* cfg, int32_t k, const ctts_waveglow_flow_weights* w,
                               void* packed_bf16, void* stream, int P, int f16 = 0) {
    Plan p; BfPlan q;
    int rc = make_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(k >= 0 && k < p.c.n_flows && w && packed_bf16, "pack_flow_bf16: bad argument");
    CTTS_CHECK_ARG(p.C % BGEMM_KC == 0 && p.H % BGEMM_KC == 0, "pack_flow_bf16: channels must be multiples of 32");
    CTTS_CHECK_ARG(!f16 || P == 1, "pack_flow: the split form exists for bf16 only");
    make_bf_plan(p, q, P, f16);
    CTTS_CHECK_ARG(q.nch_in <= BGEMM_MAX_CHUNKS && BF_SKIP_GROUP * q.nch_rs <= BGEMM_MAX_CHUNKS,
                   "pack_flow_bf16: K of %d chunks exceeds the chunk table", q.nch_in);
    hipStream_t s = as_stream(stream);
    bf16_t* bb = static_cast<bf16_t*>(packed_bf16);
    const int C = p.C, H = p.H, ks = p.c.kernel_size;
    CTTS_CHECK_ARG(w->cond_w[0] && w->cond_w[1] && p.K0 % BGEMM_KC == 0, "pack_flow_bf16: cond weights / n_mel*n_group %% 32");
    const int sdim = p.c.speaker_embed_dim;
    bf16_t* c0 = bb + q.cond0_A + (size_t)k * q.nch_c0 * BGEMM_KC * BGEMM_BM;
    bf16_t* c1 = bb + q.cond1_A + (size_t)k * q.nch_c1 * BGEMM_KC * BGEMM_BM;
    // product pi of an operand pair uses W_hi for pi = 0, 1 (against x_hi, x_lo) and W_lo for pi = 2 (against x_hi)
    for (int pi = 0; pi < P; ++pi) {
        const int part = pi == 2;
        if ((rc = launch_pack_a_bf16(c0, w->cond_w[0], 1, q.nch_c0, pi * p.K0, p.K0, BGEMM_EPI_SPLIT, C, H, 0, p.K0 + sdim, 1,
                                     s, 1, 0, part, q.f16))) return rc;
        if (sdim && (rc = launch_pack_a_bf16(c0, w->cond_w[0] + p.K0, 1, q.nch_c0, P * p.K0 + pi * p.S, sdim, BGEMM_EPI_SPLIT,
                                             C, H, 0, p.K0 + sdim, 1, s, 1, 0, part, q.f16))) return rc;
        if ((rc = launch_pack_a_bf16(c1, w->cond_w[1], 1, q.nch_c1, pi * H, H, BGEMM_EPI_SPLIT, C, H, 0, H, 1, s, 1, 0,
                                     part, q.f16))) return rc;
    }
    for (int i = 0; i < p.c.n_layers; ++i) {
        CTTS_CHECK_ARG(w->in_w[i] && w->rs_w[i] && w->cond_w[2] && w->rs_b[i], "pack_flow_bf16: NULL layer %d weights", i);
        // res_skip_layers.i is [2C][C] = {res rows, skip rows} for i < last, [C][C] = skip rows for the last layer
        const bool last = i == p.c.n_layers - 1;
        const int gi = i / BF_SKIP_GROUP, j = i % BF_SKIP_GROUP;
        for (int pi = 0; pi < P; ++pi) {
            const int part = pi == 2;
            // in-layer K = [per 32-channel slab: (tap, product) round-robin] then [cond products]
            for (int t = 0; t < ks; ++t)
                if ((rc = launch_pack_a_bf16(bb + q.in_A[k][i], w->in_w[i] + t, p.mb_in, q.nch_in, 0, C, BGEMM_EPI_GATE, C,
                                             2 * C, 0, (long long)C * ks, ks, s, ks * P, t * P + pi, part, q.f16))) return rc;
            if ((rc = launch_pack_a_bf16(bb + q.in_A[k][i], w->cond_w[2], p.mb_in, q.nch_in, ks * P * C + pi * H, H,
                                         BGEMM_EPI_GATE, C, 2 * C, (long long)2 * C * i, H, 1, s, 1, 0, part, q.f16))) return rc;
            if (!last && (rc = launch_pack_a_bf16(bb + q.rs_A[k][i], w->rs_w[i], q.mb_c, q.nch_rs, pi * C, C, BGEMM_EPI_SPLIT,
                                                  C, C, 0, C, 1, s, 1, 0, part, q.f16))) return rc;
            if ((rc = launch_pack_a_bf16(bb + q.skip_A[k][gi], w->rs_w[i], q.mb_c, q.group_layers(gi, p.c.n_layers) * q.nch_rs,
                                         (j * P + pi) * C, C, BGEMM_EPI_SPLIT, C, C, last ? 0 : C, C, 1, s, 1, 0,
                                         part, q.f16))) return rc;
        }
        hipLaunchKernelGGL(skip_bias_kernel, dim3((2 * q.mb_c * BGEMM_BM + 255) / 256), dim3(256), 0, s,
                           reinterpret_cast<float*>(bb + q.skip_b[k]), w->rs_b[i] + (last ? 0 : C), C,
                           2 * q.mb_c * BGEMM_BM, i == 0 ? 1 : 0);
        CTTS_CHECK_LAUNCH("skip_bias");
    }
    return CTTS_OK;
}

static size_t workspace_bf16_bytes_impl(const ctts_waveglow_config* cfg, int32_t batch, int32_t frames, int P) {
    Plan p; Geom g; BfWs w;
    if (make_plan(cfg, p) || make_geom(p, frames, g) || batch < 1) return 0;
    carve_bf(p, g, batch, nullptr, w, P);
    return w.total_bytes;
}

static int infer_bf16_impl(const ctts_waveglow_config* cfg, const void* packed, const void* packed_bf16, const float* mel,
                           const float* z_scaled, const int64_t* speaker_ids, float* wave, int32_t batch, int32_t frames,
                           void* workspace, size_t workspace_bytes, void* stream, int P, int f16 = 0) {
    Plan p; Geom g; BfWs w; BfPlan q;
    int rc = make_plan(cfg, p); if (rc) return rc;
    rc = make_geom(p, frames, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && packed_bf16 && mel && z_scaled && wave && workspace && batch >= 1, "infer_bf16: bad argument");
    CTTS_CHECK_ARG(p.C % BGEMM_KC == 0 && p.C <= 512, "infer_bf16: n_channels=%d (multiple of 32, <= 512)", p.C);
    CTTS_CHECK_ARG(p.c.n_group <= 8, "infer_bf16: n_group=%d (the reduced-precision flow boundaries hold <= 8 channels; fp32 takes 16)", p.c.n_group);
    make_bf_plan(p, q, P, f16);
    carve_bf(p, g, batch, static_cast<char*>(workspace), w, P);
    if (w.total_bytes > workspace_bytes) {
        set_error("infer_bf16: workspace %zu bytes < required %zu", workspace_bytes, w.total_bytes);
        return CTTS_E_WORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    const bf16_t* bblob = static_cast<const bf16_t*>(packed_bf16);
    CTTS_CHECK_HIP(hipMemcpyAsync(w.audio, z_scaled, (size_t)batch * p.c.n_group * g.L * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
    rc = launch_upsample_squeeze(mel, blob + p.up_w, p.up_mfma ? blob + p.up_wp : nullptr, blob + p.up_b, w.spect, batch, p.c.n_mel_channels, frames,
                                 p.c.win_length, p.c.hop_length, p.c.n_group, g.ld, g.pad, s);
    if (rc) return rc;
    // cond layers 0 / 1 for all flows on bf16 MFMA: spect -> bf16 K8, then two flow-batched GEMMs whose
    // epilogues write the conditioning hidden directly in bf16 K8
    hipLaunchKernelGGL(cvt_f32_to_k8_kernel, dim3((g.ld + 255) / 256, p.K0 / 8, batch), dim3(256), 0, s, w.spect,
                       w.spect_bf, p.K0, g.ld, w.spect_lo, q.f16);
    CTTS_CHECK_LAUNCH("cvt_f32_to_k8");
    if (p.S) {
        if ((rc = fill_speaker_rows(p, g, blob, speaker_ids, w.spk, batch, s))) return rc;
        hipLaunchKernelGGL(cvt_f32_to_k8_kernel, dim3((g.ld + 255) / 256, p.c.n_flows * p.S / 8, batch), dim3(256), 0, s,
                           w.spk, w.spk_bf, p.c.n_flows * p.S, g.ld, w.spk_lo, q.f16);
        CTTS_CHECK_LAUNCH("cvt_f32_to_k8(speaker rows)");
    }
    {
        const long long hstride = (long long)p.c.n_flows * p.H * g.ld;
        BGemmArgs a{};
        a.f16 = q.f16;
        a.ld = g.ld; a.pad = g.pad; a.L = g.L; a.ntiles = g.ntiles; a.batch = batch;
        a.A = bblob + q.cond0_A; a.bias = blob + p.cond0_b;
        a.nch_total = q.nch_c0; a.MB = p.c.n_flows; a.M = p.c.n_flows * BGEMM_BM;
        int ns = push_segs(a, 0, P, w.spect_bf, w.spect_lo, (long long)p.K0 * g.ld, p.K0 / BGEMM_KC, 0, 0);
        if (p.S) ns = push_segs(a, ns, P, w.spk_bf, w.spk_lo, (long long)p.c.n_flows * p.S * g.ld, p.S / BGEMM_KC, 0, p.S);
        a.nseg = ns;
        a.dst0 = w.h_tmp_bf; a.dst0_bstride = hstride; a.acc0 = 0;
        a.dst1 = w.h_tmp_bf; a.dst1_bstride = hstride; a.acc1 = 0;
        a.split = a.M; a.lo_off = w.h_lo;
        if ((rc = launch_gemm_bf16(BGEMM_EPI_SPLIT, a, s))) return rc;
        a.A = bblob + q.cond1_A; a.bias = blob + p.cond1_b;
        a.nch_total = q.nch_c1;
        a.nseg = push_segs(a, 0, P, w.h_tmp_bf, w.h_lo, hstride, p.H / BGEMM_KC, 0, BGEMM_BM);
        a.dst0 = w.h_bf; a.dst1 = w.h_bf;
        if ((rc = launch_gemm_bf16(BGEMM_EPI_SPLIT, a, s))) return rc;
    }
    for (int k = p.c.n_flows - 1; k >= 0; --k) {
        rc = run_wn_stack_bf16(p, q, g, blob, bblob, k, w, batch, s);
        if (rc) return rc;
        const auto& f = p.fl[k];
        const auto& d = p.fd[k];
        float* wv = k == 0 ? wave : nullptr;
        dim3 tgrid((g.L + 255) / 256, batch);
#define CTTS_BTAIL1(HH, FF)                                                                                           \
        hipLaunchKernelGGL((flow_tail_bf16_kernel<HH, FF>), tgrid, dim3(256), 0, s, w.out, w.audio, wv, blob + f.end_w,   \
                           blob + f.end_b, blob + f.winv, p.C, p.c.n_group, d.ch_off, g.L, g.ld, g.pad, w.x_lo)
#define CTTS_BTAIL(HH)                                                                                            \
    case HH:                                                                                                      \
        if (q.f16) CTTS_BTAIL1(HH, 2); else if (w.x_lo) CTTS_BTAIL1(HH, 1); else CTTS_BTAIL1(HH, 0);              \
        break;
        switch (d.n_half) {
            CTTS_BTAIL(1) CTTS_BTAIL(2) CTTS_BTAIL(3) CTTS_BTAIL(4)
            default: set_error("flow_tail_bf16: n_half=%d", d.n_half); return CTTS_E_ARG;
        }
#undef CTTS_BTAIL
#undef CTTS_BTAIL1
        CTTS_CHECK_LAUNCH("flow_tail_bf16");
    }
    return CTTS_OK;
}

size_t ctts_waveglow_packed_bf16_bytes(const ctts_waveglow_config* cfg) { return packed_bf16_bytes_impl(cfg, 1); }
int ctts_waveglow_pack_flow_bf16(const ctts_waveglow_config* cfg, int32_t k, const ctts_waveglow_flow_weights* w,
                                 void* packed_bf16, void* stream) {
    return pack_flow_bf16_impl(cfg, k, w, packed_bf16, stream, 1);
}
size_t ctts_waveglow_workspace_bf16_bytes(const ctts_waveglow_config* cfg, int32_t batch, int32_t frames) {
    return workspace_bf16_bytes_impl(cfg, batch, frames, 1);
}
int ctts_waveglow_infer_bf16(const ctts_waveglow_config* cfg, const void* packed, const void* packed_bf16,
                             const float* mel, const float* z_scaled, float* wave, int32_t batch, int32_t frames,
                             void* workspace, size_t workspace_bytes, void* stream) {
    return infer_bf16_impl(cfg, packed, packed_bf16, mel, z_scaled, nullptr, wave, batch, frames, workspace, workspace_bytes,
                           stream, 1);
}
int ctts_waveglow_infer_spk_bf16(const ctts_waveglow_config* cfg, const void* packed, const void* packed_bf16,
                                 const float* mel, const float* z_scaled, const int64_t* speaker_ids, float* wave,
                                 int32_t batch, int32_t frames, void* workspace, size_t workspace_bytes, void* stream) {
    return infer_bf16_impl(cfg, packed, packed_bf16, mel, z_scaled, speaker_ids, wave, batch, frames, workspace,
                           workspace_bytes, stream, 1);
}

// IEEE-half variant: the layouts, sizes and kernels of the bf16 variant with half storage and v_mfma_f32_32x32x16_f16
int ctts_waveglow_pack_flow_f16(const ctts_waveglow_config* cfg, int32_t k, const ctts_waveglow_flow_weights* w,
                                void* packed_f16, void* stream) {
    return pack_flow_bf16_impl(cfg, k, w, packed_f16, stream, 1, 1);
}
int ctts_waveglow_infer_spk_f16(const ctts_waveglow_config* cfg, const void* packed, const void* packed_f16,
                                const float* mel, const float* z_scaled, const int64_t* speaker_ids, float* wave,
                                int32_t batch, int32_t frames, void* workspace, size_t workspace_bytes, void* stream) {
    return infer_bf16_impl(cfg, packed, packed_f16, mel, z_scaled, speaker_ids, wave, batch, frames, workspace,
                           workspace_bytes, stream, 1, 1);
}

size_t ctts_waveglow_packed_bf16x3_bytes(const ctts_waveglow_config* cfg) { return packed_bf16_bytes_impl(cfg, 3); }
int ctts_waveglow_pack_flow_bf16x3(const ctts_waveglow_config* cfg, int32_t k, const ctts_waveglow_flow_weights* w,
                                   void* packed_bf16x3, void* stream) {
    return pack_flow_bf16_impl(cfg, k, w, packed_bf16x3, stream, 3);
}
size_t ctts_waveglow_workspace_bf16x3_bytes(const ctts_waveglow_config* cfg, int32_t batch, int32_t frames) {
    return workspace_bf16_bytes_impl(cfg, batch, frames, 3);
}
int ctts_waveglow_infer_spk_bf16x3(const ctts_waveglow_config* cfg, const void* packed, const void* packed_bf16x3,
                                   const float* mel, const float* z_scaled, const int64_t* speaker_ids, float* wave,
                                   int32_t batch, int32_t frames, void* workspace, size_t workspace_bytes, void* stream) {
    return infer_bf16_impl(cfg, packed, packed_bf16x3, mel, z_scaled, speaker_ids, wave, batch, frames, workspace,
                           workspace_bytes, stream, 3);
}

}  // extern "C"
