// Tacotron2-TM one-shot stages as operator-level primitives (include/cookietts_hip.h, "Tacotron2-TM one-shot
// stages"): same-padded Conv1d (+ folded eval BatchNorm + LeakyReLU/tanh) on the fp32 MFMA conv-GEMM, embedding
// gather, per-utterance memory columns, padded <-> dense copies.  The packed-sequence LSTM lives in
// tacotron_decoder.hip next to the decoder's LSTM step kernel it shares.
#include "gemm_f32.h"
#include "waveglow_kernels.h"

namespace ctts {
namespace {

constexpr size_t ALIGN_F = 64;
inline size_t align_up(size_t v) { return (v + ALIGN_F - 1) / ALIGN_F * ALIGN_F; }
constexpr int A_TILE = GEMM_KC * GEMM_BM;

struct ConvPlan { int mb, nch; size_t A, bias, wfold, bfold, total; };

int make_conv_plan(const ctts_conv1d_desc* d, ConvPlan& p) {
    CTTS_CHECK_ARG(d != nullptr, "conv1d desc is NULL");
    CTTS_CHECK_ARG(d->c_in >= 16 && d->c_in % GEMM_KC == 0, "conv1d: c_in=%d (multiple of 16)", d->c_in);
    CTTS_CHECK_ARG(d->c_out >= 1, "conv1d: c_out=%d", d->c_out);
    CTTS_CHECK_ARG(d->kernel_size % 2 == 1 && d->kernel_size >= 1 && d->kernel_size <= GEMM_MAX_SEG - 1,
                   "conv1d: kernel_size=%d (odd, <= %d)", d->kernel_size, GEMM_MAX_SEG - 1);
    CTTS_CHECK_ARG(d->act >= 0 && d->act <= 2, "conv1d: act=%d", d->act);
    CTTS_CHECK_ARG(gemm_mode_valid(d->f32_gemm_mode), "conv1d: f32_gemm_mode=%d (CTTS_GEMM_*)", d->f32_gemm_mode);
    p.mb = (d->c_out + GEMM_BM - 1) / GEMM_BM;
    p.nch = d->kernel_size * d->c_in / GEMM_KC;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    p.A = take((size_t)p.mb * p.nch * A_TILE);
    p.bias = take((size_t)p.mb * GEMM_BM);
    p.wfold = take((size_t)d->c_out * d->c_in * d->kernel_size);      // scratch for the BN-folded dense weight
    p.bfold = take(d->c_out);
    p.total = o;
    return CTTS_OK;
}

// w'[o][:] = w[o][:] * s, b'[o] = (b[o] - mean[o]) * s + beta[o], s = gamma[o] / sqrt(var[o] + eps)   (eval BatchNorm1d)
__global__ __launch_bounds__(256) void fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      float eps, float* __restrict__ wf, float* __restrict__ bf, int fan) {
    const int o = blockIdx.x;
    float s = 1.f, sh = 0.f;
    if (gamma) {
        s = gamma[o] / sqrtf(var[o] + eps);
        sh = beta[o] - mean[o] * s;
    }
    for (int i = threadIdx.x; i < fan; i += 256) wf[(size_t)o * fan + i] = w[(size_t)o * fan + i] * s;
    if (threadIdx.x == 0) bf[o] = (b ? b[o] : 0.f) * s + sh;
}

__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ emb, const float* __restrict__ spk,
                                                    const long long* __restrict__ text, const long long* __restrict__ speakers,
                                                    float* __restrict__ x0, int T, int E, int S, int ld, int pad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    float v;
    if (c < E) v = emb[(size_t)text[(size_t)b * T + t] * E + c];
    else v = spk[(size_t)speakers[b] * S + (c - E)];
    x0[((size_t)b * (E + S) + c) * ld + pad + t] = v;
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, long long src_bstride, int src_ld,
                                                       float* __restrict__ dst, int C, int T, int ld, int pad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    dst[((size_t)b * C + c) * ld + pad + t] = src[(size_t)b * src_bstride + (size_t)c * src_ld + t];
}

__global__ __launch_bounds__(256) void unpad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         long long dst_bstride, int dst_ld, int C, int T, int ld, int pad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    dst[(size_t)b * dst_bstride + (size_t)c * dst_ld + t] = src[((size_t)b * C + c) * ld + pad + t];
}

struct MemArgs {
    ctts_taco_memory_weights w;
    const float* hn; const long long* speakers; const float* tm;
    float *memory_in, *pred_sylps;
    const float* gt_sylps;      // model.py:1058 "gt_sylps or pred_sylps": what the SylpsNet reads when the caller gives one, else NULL
    int T, enc_dim, spk_dim, syl_hidden, tm_dim, tm_crushed;
};

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one workgroup per utterance
__global__ __launch_bounds__(256) void memory_kernel(const MemArgs a) {
    __shared__ float red[4];
    __shared__ float vec[1024];          // [speaker embed | sylzu | torchMoji crushed]
    __shared__ float hid[64];
    const int b = blockIdx.x, t = threadIdx.x;
    const int extra = a.spk_dim + 1 + a.tm_crushed;
    // pred_sylps = sylps_layer(hidden_state)
    float acc = 0.f;
    for (int k = t; k < a.enc_dim; k += 256) acc = fmaf(a.w.sylps_w[k], a.hn[(size_t)b * a.enc_dim + k], acc);
    const float pred = block_sum(acc, red) + a.w.sylps_b[0];
    if (t == 0) a.pred_sylps[b] = pred;
    const float sylps = a.gt_sylps ? a.gt_sylps[b] : pred;
    // SylpsNet.infer_auto: cat(sylps, ln sylps) -> Linear -> LeakyReLU(0.05) -> Linear; zu = (cat + res_w * res)[0]
    const float ln = logf(sylps);
    if (t < a.syl_hidden) {
        const float h = a.w.syl_w0[t * 2] * sylps + a.w.syl_w0[t * 2 + 1] * ln + a.w.syl_b0[t];
        hid[t] = h > 0.f ? h : 0.05f * h;
    }
    __syncthreads();
    if (t == 0) {
        float r = a.w.syl_b2[0];
        for (int k = 0; k < a.syl_hidden; ++k) r = fmaf(a.w.syl_w2[k], hid[k], r);
        vec[a.spk_dim] = sylps + a.w.syl_res_weight[0] * r;
    }
    for (int j = t; j < a.spk_dim; j += 256) vec[j] = a.w.speaker_embedding[(size_t)a.speakers[b] * a.spk_dim + j];
    // torchMoji: eval BatchNorm1d then Linear(tm_dim -> tm_crushed); wave per output row
    {
        const int lane = t & 63, wv = t >> 6;
        for (int o = wv; o < a.tm_crushed; o += 4) {
            float s = 0.f;
            for (int k = lane; k < a.tm_dim; k += 64) {
                float x = a.tm[(size_t)b * a.tm_dim + k];
                if (a.w.tm_gamma) x = (x - a.w.tm_mean[k]) / sqrtf(a.w.tm_var[k] + 1e-5f) * a.w.tm_gamma[k] + a.w.tm_beta[k];
                s = fmaf(a.w.tm_w[(size_t)o * a.tm_dim + k], x, s);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) vec[a.spk_dim + 1 + o] = s + a.w.tm_b[o];
        }
    }
    __syncthreads();
    const int row = a.enc_dim + extra;
    for (int i = t; i < a.T * extra; i += 256) {
        const int tt = i / extra, j = i % extra;
        a.memory_in[((size_t)b * a.T + tt) * row + a.enc_dim + j] = vec[j];
    }
}

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

size_t ctts_conv1d_packed_bytes(const ctts_conv1d_desc* d) {
    ConvPlan p;
    if (make_conv_plan(d, p)) return 0;
    return p.total * sizeof(float);
}

int ctts_conv1d_pack_f32(const ctts_conv1d_desc* d, const float* w, const float* b, const float* bn_gamma,
                         const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps, void* packed,
                         void* stream) {
    ConvPlan p;
    int rc = make_conv_plan(d, p); if (rc) return rc;
    CTTS_CHECK_ARG(w && packed, "conv1d_pack: NULL pointer");
    CTTS_CHECK_ARG((bn_gamma != nullptr) == (bn_beta != nullptr) && (bn_gamma != nullptr) == (bn_mean != nullptr) &&
                   (bn_gamma != nullptr) == (bn_var != nullptr), "conv1d_pack: BatchNorm parameters must be all set or all NULL");
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    const int fan = d->c_in * d->kernel_size;
    hipLaunchKernelGGL(fold_bn_kernel, dim3(d->c_out), dim3(256), 0, s, w, b, bn_gamma, bn_beta, bn_mean, bn_var, bn_eps,
                       blob + p.wfold, blob + p.bfold, fan);
    CTTS_CHECK_LAUNCH("fold_bn");
    // K order = [16-channel slab][tap]: member t of a kernel_size-way round-robin group; w is [c_out][c_in][k]
    for (int t = 0; t < d->kernel_size; ++t)
        if ((rc = launch_pack_a(blob + p.A, blob + p.wfold + t, GEMM_BM, p.mb, p.nch, 0, d->c_in, GEMM_EPI_SPLIT, 0,
                                d->c_out, 0, (long long)fan, d->kernel_size, s, d->kernel_size, t))) return rc;
    return launch_pack_bias(blob + p.bias, GEMM_BM, p.mb, blob + p.bfold, 0, nullptr, 0, GEMM_EPI_SPLIT, 0, d->c_out, s);
}

int ctts_conv1d_f32(const ctts_conv1d_desc* d, const void* packed, const float* x, float* y, int32_t accumulate,
                    int32_t batch, int32_t T, int32_t ld, int32_t pad, void* stream) {
    ConvPlan p;
    int rc = make_conv_plan(d, p); if (rc) return rc;
    CTTS_CHECK_ARG(packed && x && y && batch >= 1 && T >= 1, "conv1d: bad argument");
    CTTS_CHECK_ARG(!accumulate || d->act == 0, "conv1d: accumulate with an activation");
    const int ntiles = (T + GEMM_BN - 1) / GEMM_BN;
    CTTS_CHECK_ARG(pad >= d->kernel_size / 2 && ld % 4 == 0 && ntiles * GEMM_BN + 2 * pad <= ld,
                   "conv1d: geometry T=%d ld=%d pad=%d", T, ld, pad);
    const float* blob = static_cast<const float*>(packed);
    GemmArgs a{};
    a.ld = ld; a.pad = pad; a.L = T; a.ntiles = ntiles; a.batch = batch;
    a.dst_ld = ld; a.dst_pad = pad;
    a.gemm_mode = d->f32_gemm_mode;
    a.A = blob + p.A; a.bias = blob + p.bias;
    a.nseg = d->kernel_size; a.interleave = d->kernel_size; a.nch_total = p.nch; a.MB = p.mb; a.M = d->c_out;
    for (int t = 0; t < d->kernel_size; ++t)
        a.seg[t] = {x, (long long)d->c_in * ld, d->c_in / GEMM_KC, t - d->kernel_size / 2, 0, 0};
    a.dst0 = y; a.dst0_bstride = (long long)d->c_out * ld; a.acc0 = accumulate ? 1 : 0;
    a.dst1 = y; a.dst1_bstride = a.dst0_bstride; a.acc1 = 0;
    a.split = p.mb * GEMM_BM;
    a.clip = d->slope;
    const int epi = d->act == 1 ? GEMM_EPI_LRELU : d->act == 2 ? GEMM_EPI_TANH : GEMM_EPI_SPLIT;
    return launch_gemm_f32(epi, a, as_stream(stream));
}

int ctts_taco_embed_f32(const float* embedding, const float* spk_table, const int64_t* text, const int64_t* speakers,
                        float* x0, int32_t batch, int32_t T, int32_t E, int32_t S, int32_t ld, int32_t pad, void* stream) {
    CTTS_CHECK_ARG(embedding && text && x0 && (S == 0 || (spk_table && speakers)) && batch >= 1 && T >= 1, "embed: bad argument");
    hipLaunchKernelGGL(embed_kernel, dim3((T + 255) / 256, E + S, batch), dim3(256), 0, as_stream(stream), embedding,
                       spk_table, reinterpret_cast<const long long*>(text), reinterpret_cast<const long long*>(speakers), x0, T,
                       E, S, ld, pad);
    CTTS_CHECK_LAUNCH("embed");
    return CTTS_OK;
}

int ctts_taco_memory_f32(const ctts_taco_memory_weights* w, const float* hn, const int64_t* speakers,
                         const float* torchmoji, float* memory_in, float* pred_sylps, int32_t batch, int32_t T,
                         int32_t enc_dim, int32_t spk_dim, int32_t syl_hidden, int32_t tm_dim, int32_t tm_crushed,
                         void* stream) {
    return ctts_taco_memory_sylps_f32(w, hn, speakers, torchmoji, nullptr, memory_in, pred_sylps, batch, T, enc_dim, spk_dim,
                                      syl_hidden, tm_dim, tm_crushed, stream);
}

int ctts_taco_memory_sylps_f32(const ctts_taco_memory_weights* w, const float* hn, const int64_t* speakers,
                               const float* torchmoji, const float* gt_sylps, float* memory_in, float* pred_sylps, int32_t batch,
                               int32_t T, int32_t enc_dim, int32_t spk_dim, int32_t syl_hidden, int32_t tm_dim,
                               int32_t tm_crushed, void* stream) {
    CTTS_CHECK_ARG(w && hn && speakers && torchmoji && memory_in && pred_sylps && batch >= 1 && T >= 1, "memory: bad argument");
    CTTS_CHECK_ARG(spk_dim + 1 + tm_crushed <= 1024 && syl_hidden <= 64, "memory: dims too large");
    MemArgs a{};
    a.w = *w; a.hn = hn; a.speakers = reinterpret_cast<const long long*>(speakers); a.tm = torchmoji;
    a.memory_in = memory_in; a.pred_sylps = pred_sylps; a.gt_sylps = gt_sylps;
    a.T = T; a.enc_dim = enc_dim; a.spk_dim = spk_dim; a.syl_hidden = syl_hidden; a.tm_dim = tm_dim; a.tm_crushed = tm_crushed;
    hipLaunchKernelGGL(memory_kernel, dim3(batch), dim3(256), 0, as_stream(stream), a);
    CTTS_CHECK_LAUNCH("memory");
    return CTTS_OK;
}

int ctts_pad_rows_f32(const float* src, int64_t src_bstride, int32_t src_ld, float* dst, int32_t batch, int32_t C,
                      int32_t T, int32_t ld, int32_t pad, void* stream) {
    CTTS_CHECK_ARG(src && dst && batch >= 1 && C >= 1 && T >= 1 && pad + T <= ld, "pad_rows: bad argument");
    hipLaunchKernelGGL(pad_rows_kernel, dim3((T + 255) / 256, C, batch), dim3(256), 0, as_stream(stream), src,
                       (long long)src_bstride, src_ld, dst, C, T, ld, pad);
    CTTS_CHECK_LAUNCH("pad_rows");
    return CTTS_OK;
}

int ctts_unpad_rows_f32(const float* src, float* dst, int64_t dst_bstride, int32_t dst_ld, int32_t batch, int32_t C,
                        int32_t T, int32_t ld, int32_t pad, void* stream) {
    CTTS_CHECK_ARG(src && dst && batch >= 1 && C >= 1 && T >= 1 && pad + T <= ld, "unpad_rows: bad argument");
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((T + 255) / 256, C, batch), dim3(256), 0, as_stream(stream), src, dst,
                       (long long)dst_bstride, dst_ld, C, T, ld, pad);
    CTTS_CHECK_LAUNCH("unpad_rows");
    return CTTS_OK;
}

}  // extern "C"
