// STFT magnitude + mel projection on the fp32 MFMA conv-GEMM (utils/audio/stft.py:79-111,180-207).
//
// The reference computes the STFT as conv1d(x, basis[1026,1,1024], stride=hop): a dense
// contraction of every 1024-sample frame with 1026 windowed DFT rows.  Here the reflect-padded
// frames are laid out as a k-major matrix Xf[k][frame] (one small gather kernel), the basis is
// packed with (Re, Im) rows of a bin in the same wave, and |X| = sqrt(Re^2 + Im^2) is the GEMM
// epilogue; the mel projection is a second GEMM whose epilogue is log(max(., clamp)).
#include "gemm_f32.h"
#include "waveglow_kernels.h"

namespace ctts {
namespace {

constexpr int A_TILE = GEMM_KC * GEMM_BM;
constexpr size_t ALIGN_F = 64;
inline size_t align_up(size_t v) { return (v + ALIGN_F - 1) / ALIGN_F * ALIGN_F; }

struct StftPlan {
    ctts_stft_config c;
    int N, cutoff, kmel, mb_mag, mb_mel;
    size_t basis_A, zero_bias, mel_A, lin_A, inv_A, win_sq, total;
    int mb_lin, mb_inv, kinv;
};

int make_stft_plan(const ctts_stft_config* cfg, StftPlan& p) {
    CTTS_CHECK_ARG(cfg != nullptr, "stft config is NULL");
    p.c = *cfg;
    CTTS_CHECK_ARG(cfg->filter_length >= 32 && cfg->filter_length % GEMM_KC == 0, "filter_length=%d (multiple of 16)",
                   cfg->filter_length);
    CTTS_CHECK_ARG(cfg->hop_length >= 1 && cfg->hop_length <= cfg->filter_length, "hop_length=%d", cfg->hop_length);
    CTTS_CHECK_ARG(cfg->n_mel_channels >= 0 && cfg->n_mel_channels <= 1024, "n_mel_channels=%d", cfg->n_mel_channels);
    p.N = cfg->filter_length;
    p.cutoff = p.N / 2 + 1;
    p.kmel = round_up(p.cutoff, GEMM_KC);
    p.mb_mag = (p.cutoff + 127) / 128;
    p.mb_mel = (cfg->n_mel_channels + GEMM_BM - 1) / GEMM_BM;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    p.basis_A = take((size_t)p.mb_mag * (p.N / GEMM_KC) * A_TILE);
    int mbmax = p.mb_mag > p.mb_mel ? p.mb_mag : p.mb_mel;
    const int mb_lin_ = (2 * p.cutoff + GEMM_BM - 1) / GEMM_BM, mb_inv_ = (p.N + GEMM_BM - 1) / GEMM_BM;
    if (mb_lin_ > mbmax) mbmax = mb_lin_;
    if (mb_inv_ > mbmax) mbmax = mb_inv_;
    p.zero_bias = take((size_t)mbmax * GEMM_BM);
    p.mel_A = take((size_t)p.mb_mel * (p.kmel / GEMM_KC) * A_TILE);
    // phase / inverse path: [Re; Im] rows in dense order, and the transposed inverse basis (N rows x 2*cutoff)
    p.mb_lin = (2 * p.cutoff + GEMM_BM - 1) / GEMM_BM;
    p.lin_A = take((size_t)p.mb_lin * (p.N / GEMM_KC) * A_TILE);
    p.kinv = round_up(2 * p.cutoff, GEMM_KC);
    p.mb_inv = (p.N + GEMM_BM - 1) / GEMM_BM;
    p.inv_A = take((size_t)p.mb_inv * (p.kinv / GEMM_KC) * A_TILE);
    p.win_sq = take(p.N);
    p.total = o;
    return CTTS_OK;
}

struct StftGeom { int frames, ntiles, ld; };

int make_stft_geom(const StftPlan& p, int samples, StftGeom& g) {
    CTTS_CHECK_ARG(samples > p.N / 2, "samples=%d must exceed filter_length/2 (reflect pad)", samples);
    g.frames = samples / p.c.hop_length + 1;
    g.ntiles = (g.frames + GEMM_BN - 1) / GEMM_BN;
    g.ld = g.ntiles * GEMM_BN;
    return CTTS_OK;
}

// Xf[b][k][n] = ypad[b][n*hop + k], ypad = reflect-pad(y, N/2)  (stft.py:91-95)
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ y, float* __restrict__ xf,
                                                          int T, int N, int hop, int frames, int ld) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int k0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
    const int b = blockIdx.z;
    if (n >= ld || k0 >= N) return;
    const float* yb = y + (size_t)b * T;
    float* xb = xf + ((size_t)b * N + k0) * ld + n;
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
        float v = 0.f;
        if (n < frames) {
            int j = n * hop + k0 + kk - N / 2;
            if (j < 0) j = -j;
            if (j >= T) j = 2 * (T - 1) - j;
            v = yb[j];
        }
        xb[(size_t)kk * ld] = v;
    }
}

// (re, im) rows [B][kinv][ld] -> magnitude / phase dense [B][cutoff][frames]   (stft.py:107-110)
__global__ __launch_bounds__(256) void stft_polar_kernel(const float* __restrict__ ri, float* __restrict__ mag,
                                                         float* __restrict__ phase, int cutoff, int kinv, int frames, int ld) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (n >= frames) return;
    const float re = ri[((size_t)b * kinv + c) * ld + n], im = ri[((size_t)b * kinv + cutoff + c) * ld + n];
    const size_t o = ((size_t)b * cutoff + c) * frames + n;
    mag[o] = sqrtf(re * re + im * im);
    if (phase) phase[o] = atan2f(im, re);
}

// R[b][c][n] = m cos(phase), R[b][cutoff + c][n] = m sin(phase), m = bias ? max(mag - bias[c]*strength, 0) : mag
// (stft.py:118-119; denoiser.py:62-67)
__global__ __launch_bounds__(256) void stft_recombine_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                             const float* __restrict__ bias, int bias_bstride,
                                                             float strength, float* __restrict__ R, int cutoff, int kinv,
                                                             int frames, int ld) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (n >= frames) return;
    const size_t i = ((size_t)b * cutoff + c) * frames + n;
    float m = mag[i];
    if (bias) m = fmaxf(m - bias[(size_t)b * bias_bstride + c] * strength, 0.f);
    const float ph = phase[i];
    R[((size_t)b * kinv + c) * ld + n] = m * cosf(ph);
    R[((size_t)b * kinv + cutoff + c) * ld + n] = m * sinf(ph);
}

// overlap-add of Y[b][k][n] (conv_transpose1d, stride hop) + window-sum-square normalisation + N/hop scaling +
// crop N/2 on both sides (stft.py:121-146, audio_processing.py:7-56)
__global__ __launch_bounds__(256) void stft_ola_kernel(const float* __restrict__ Y, const float* __restrict__ win_sq,
                                                       float* __restrict__ out, int N, int hop, int frames, int ld, int T_out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T_out) return;
    const int tt = t + N / 2;                       // position in the un-cropped signal
    float acc = 0.f, wss = 0.f;
    const int n_hi = min(frames - 1, tt / hop);
    for (int n = n_hi; n >= 0 && tt - n * hop < N; --n) {
        const int k = tt - n * hop;
        acc += Y[((size_t)b * N + k) * ld + n];
        wss += win_sq[k];
    }
    if (wss > 1.17549435e-38f) acc /= wss;          // librosa.util.tiny(float32)
    out[(size_t)b * T_out + t] = acc * ((float)N / (float)hop);
}

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

size_t ctts_stft_packed_bytes(const ctts_stft_config* cfg) {
    StftPlan p;
    if (make_stft_plan(cfg, p)) return 0;
    return p.total * sizeof(float);
}

int ctts_stft_pack(const ctts_stft_config* cfg, const float* forward_basis, const float* mel_basis, void* packed,
                   void* stream) {
    StftPlan p;
    int rc = make_stft_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(forward_basis && packed, "stft_pack: NULL pointer");
    CTTS_CHECK_ARG(p.c.n_mel_channels == 0 || mel_basis, "stft_pack: mel_basis is NULL");
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    CTTS_CHECK_HIP(hipMemsetAsync(blob, 0, p.total * sizeof(float), s));
    rc = launch_pack_a(blob + p.basis_A, forward_basis, GEMM_BM, p.mb_mag, p.N / GEMM_KC, 0, p.N, GEMM_EPI_MAG, p.cutoff,
                       2 * p.cutoff, 0, p.N, 1, s);
    if (rc) return rc;
    if (p.c.n_mel_channels > 0)
        rc = launch_pack_a(blob + p.mel_A, mel_basis, GEMM_BM, p.mb_mel, p.kmel / GEMM_KC, 0, p.cutoff, GEMM_EPI_LOG, 0,
                           p.c.n_mel_channels, 0, p.cutoff, 1, s);
    if (rc) return rc;
    return launch_pack_a(blob + p.lin_A, forward_basis, GEMM_BM, p.mb_lin, p.N / GEMM_KC, 0, p.N, GEMM_EPI_SPLIT, 0,
                         2 * p.cutoff, 0, p.N, 1, s);
}

int ctts_stft_pack_inverse(const ctts_stft_config* cfg, const float* inverse_basis, const float* window_sq, void* packed,
                           void* stream) {
    StftPlan p;
    int rc = make_stft_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(inverse_basis && window_sq && packed, "stft_pack_inverse: NULL pointer");
    hipStream_t s = as_stream(stream);
    float* blob = static_cast<float*>(packed);
    // A_inv[k][m] = inverse_basis[m][k]: dense row k (N rows), K index m (2*cutoff): source strides (1, N)
    rc = launch_pack_a(blob + p.inv_A, inverse_basis, GEMM_BM, p.mb_inv, p.kinv / GEMM_KC, 0, 2 * p.cutoff, GEMM_EPI_SPLIT, 0,
                       p.N, 0, 1, p.N, s);
    if (rc) return rc;
    CTTS_CHECK_HIP(hipMemcpyAsync(blob + p.win_sq, window_sq, p.N * sizeof(float), hipMemcpyDeviceToDevice, s));
    return CTTS_OK;
}

size_t ctts_stft_workspace_bytes(const ctts_stft_config* cfg, int32_t batch, int32_t samples) {
    StftPlan p; StftGeom g;
    if (make_stft_plan(cfg, p) || make_stft_geom(p, samples, g) || batch < 1) return 0;
    // frames matrix / overlap-add input [B][N][ld] + (magnitude [B][kmel][ld] | re,im [B][kinv][ld])
    return (align_up((size_t)batch * p.N * g.ld) + align_up((size_t)batch * p.kinv * g.ld)) * sizeof(float);
}

int ctts_stft_mel_f32(const ctts_stft_config* cfg, const void* packed, const float* y, float* mag, float* mel,
                      int32_t batch, int32_t samples, void* workspace, size_t workspace_bytes, void* stream) {
    StftPlan p; StftGeom g;
    int rc = make_stft_plan(cfg, p); if (rc) return rc;
    rc = make_stft_geom(p, samples, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && y && workspace && batch >= 1, "stft_mel: bad argument");
    CTTS_CHECK_ARG(mel == nullptr || p.c.n_mel_channels > 0, "stft_mel: mel requested but n_mel_channels == 0");
    const size_t need = ctts_stft_workspace_bytes(cfg, batch, samples);
    if (need > workspace_bytes) {
        set_error("stft_mel: workspace %zu bytes < required %zu", workspace_bytes, need);
        return CTTS_E_WORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    float* xf = static_cast<float*>(workspace);
    float* wmag = xf + align_up((size_t)batch * p.N * g.ld);
    dim3 grid((g.ld + 63) / 64, (p.N + 63) / 64, batch);
    hipLaunchKernelGGL(stft_frames_kernel, grid, dim3(256), 0, s, y, xf, samples, p.N, p.c.hop_length, g.frames, g.ld);
    CTTS_CHECK_LAUNCH("stft_frames");

    GemmArgs a{};
    a.gemm_mode = CTTS_GEMM_F32;   // Fourier sums cancel: exact fp32 products, whatever the library default is
    a.ld = g.ld; a.pad = 0; a.L = g.frames; a.ntiles = g.ntiles; a.batch = batch;
    a.A = blob + p.basis_A; a.bias = blob + p.zero_bias;
    a.nseg = 1; a.nch_total = p.N / GEMM_KC; a.MB = p.mb_mag;
    a.seg[0] = {xf, (long long)p.N * g.ld, p.N / GEMM_KC, 0, 0, 0};
    a.M = 2 * p.cutoff; a.pairC = p.cutoff;
    a.dst0 = wmag; a.dst0_bstride = (long long)p.kmel * g.ld; a.dst_ld = g.ld; a.dst_pad = 0;
    rc = launch_gemm_f32(GEMM_EPI_MAG, a, s);
    if (rc) return rc;
    if (mag) {
        for (int b = 0; b < batch; ++b)
            CTTS_CHECK_HIP(hipMemcpy2DAsync(mag + (size_t)b * p.cutoff * g.frames, (size_t)g.frames * sizeof(float),
                                            wmag + (size_t)b * p.kmel * g.ld, (size_t)g.ld * sizeof(float),
                                            (size_t)g.frames * sizeof(float), p.cutoff, hipMemcpyDeviceToDevice, s));
    }
    if (mel) {
        GemmArgs m{};
        m.gemm_mode = CTTS_GEMM_F32;
        m.ld = g.ld; m.pad = 0; m.L = g.frames; m.ntiles = g.ntiles; m.batch = batch;
        m.A = blob + p.mel_A; m.bias = blob + p.zero_bias;
        m.nseg = 1; m.nch_total = p.kmel / GEMM_KC; m.MB = p.mb_mel;
        m.seg[0] = {wmag, (long long)p.kmel * g.ld, p.kmel / GEMM_KC, 0, 0, 0};
        m.M = p.c.n_mel_channels; m.split = p.mb_mel * GEMM_BM;
        m.dst0 = mel; m.dst0_bstride = (long long)p.c.n_mel_channels * g.frames; m.acc0 = 0;
        m.dst1 = mel; m.dst1_bstride = m.dst0_bstride; m.acc1 = 0;
        m.dst_ld = g.frames; m.dst_pad = 0; m.clip = p.c.clamp_val;
        rc = launch_gemm_f32(GEMM_EPI_LOG, m, s);
        if (rc) return rc;
    }
    return CTTS_OK;
}

int ctts_stft_transform_f32(const ctts_stft_config* cfg, const void* packed, const float* y, float* mag, float* phase,
                            int32_t batch, int32_t samples, void* workspace, size_t workspace_bytes, void* stream) {
    StftPlan p; StftGeom g;
    int rc = make_stft_plan(cfg, p); if (rc) return rc;
    rc = make_stft_geom(p, samples, g); if (rc) return rc;
    CTTS_CHECK_ARG(packed && y && mag && workspace && batch >= 1, "stft_transform: bad argument");
    const size_t need = ctts_stft_workspace_bytes(cfg, batch, samples);
    if (need > workspace_bytes) { set_error("stft_transform: workspace %zu bytes < required %zu", workspace_bytes, need); return CTTS_E_WORKSPACE; }
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    float* xf = static_cast<float*>(workspace);
    float* ri = xf + align_up((size_t)batch * p.N * g.ld);
    hipLaunchKernelGGL(stft_frames_kernel, dim3((g.ld + 63) / 64, (p.N + 63) / 64, batch), dim3(256), 0, s, y, xf, samples,
                       p.N, p.c.hop_length, g.frames, g.ld);
    CTTS_CHECK_LAUNCH("stft_frames");
    GemmArgs a{};
    a.gemm_mode = CTTS_GEMM_F32;   // Fourier sums cancel: exact fp32 products, whatever the library default is
    a.ld = g.ld; a.pad = 0; a.L = g.frames; a.ntiles = g.ntiles; a.batch = batch; a.dst_ld = g.ld; a.dst_pad = 0;
    a.A = blob + p.lin_A; a.bias = blob + p.zero_bias;
    a.nseg = 1; a.nch_total = p.N / GEMM_KC; a.MB = p.mb_lin; a.M = 2 * p.cutoff;
    a.seg[0] = {xf, (long long)p.N * g.ld, p.N / GEMM_KC, 0, 0, 0};
    a.dst0 = ri; a.dst0_bstride = (long long)p.kinv * g.ld; a.dst1 = ri; a.dst1_bstride = a.dst0_bstride;
    a.split = p.mb_lin * GEMM_BM;
    if ((rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s))) return rc;
    hipLaunchKernelGGL(stft_polar_kernel, dim3((g.frames + 255) / 256, p.cutoff, batch), dim3(256), 0, s, ri, mag, phase,
                       p.cutoff, p.kinv, g.frames, g.ld);
    CTTS_CHECK_LAUNCH("stft_polar");
    return CTTS_OK;
}

int ctts_stft_inverse_f32(const ctts_stft_config* cfg, const void* packed, const float* mag, const float* phase,
                          const float* bias_spec, float strength, float* out, int32_t batch, int32_t frames,
                          void* workspace, size_t workspace_bytes, void* stream) {
    return ctts_stft_inverse_bias_f32(cfg, packed, mag, phase, bias_spec, 0, strength, out, batch, frames, workspace,
                                      workspace_bytes, stream);
}

int ctts_stft_inverse_bias_f32(const ctts_stft_config* cfg, const void* packed, const float* mag, const float* phase,
                               const float* bias_spec, int32_t bias_bstride, float strength, float* out, int32_t batch,
                               int32_t frames, void* workspace, size_t workspace_bytes, void* stream) {
    StftPlan p; StftGeom g;
    int rc = make_stft_plan(cfg, p); if (rc) return rc;
    CTTS_CHECK_ARG(packed && mag && phase && out && workspace && batch >= 1 && frames >= 2, "stft_inverse: bad argument");
    CTTS_CHECK_ARG(bias_bstride == 0 || bias_bstride >= p.cutoff, "stft_inverse: bias stride %d < %d bins", bias_bstride, p.cutoff);
    const int T_out = (frames - 1) * p.c.hop_length;
    rc = make_stft_geom(p, T_out > p.N / 2 ? T_out : p.N, g); if (rc) return rc;
    g.frames = frames; g.ntiles = (frames + GEMM_BN - 1) / GEMM_BN; g.ld = g.ntiles * GEMM_BN;
    const size_t need = (align_up((size_t)batch * p.N * g.ld) + align_up((size_t)batch * p.kinv * g.ld)) * sizeof(float);
    if (need > workspace_bytes) { set_error("stft_inverse: workspace %zu bytes < required %zu", workspace_bytes, need); return CTTS_E_WORKSPACE; }
    hipStream_t s = as_stream(stream);
    const float* blob = static_cast<const float*>(packed);
    float* Y = static_cast<float*>(workspace);
    float* R = Y + align_up((size_t)batch * p.N * g.ld);
    CTTS_CHECK_HIP(hipMemsetAsync(R, 0, (size_t)batch * p.kinv * g.ld * sizeof(float), s));   // K padding rows / tile tail
    hipLaunchKernelGGL(stft_recombine_kernel, dim3((frames + 255) / 256, p.cutoff, batch), dim3(256), 0, s, mag, phase,
                       bias_spec, bias_bstride, strength, R, p.cutoff, p.kinv, frames, g.ld);
    CTTS_CHECK_LAUNCH("stft_recombine");
    GemmArgs a{};
    a.gemm_mode = CTTS_GEMM_F32;   // Fourier sums cancel: exact fp32 products, whatever the library default is
    a.ld = g.ld; a.pad = 0; a.L = frames; a.ntiles = g.ntiles; a.batch = batch; a.dst_ld = g.ld; a.dst_pad = 0;
    a.A = blob + p.inv_A; a.bias = blob + p.zero_bias;
    a.nseg = 1; a.nch_total = p.kinv / GEMM_KC; a.MB = p.mb_inv; a.M = p.N;
    a.seg[0] = {R, (long long)p.kinv * g.ld, p.kinv / GEMM_KC, 0, 0, 0};
    a.dst0 = Y; a.dst0_bstride = (long long)p.N * g.ld; a.dst1 = Y; a.dst1_bstride = a.dst0_bstride;
    a.split = p.mb_inv * GEMM_BM;
    if ((rc = launch_gemm_f32(GEMM_EPI_SPLIT, a, s))) return rc;
    hipLaunchKernelGGL(stft_ola_kernel, dim3((T_out + 255) / 256, batch), dim3(256), 0, s, Y, blob + p.win_sq, out, p.N,
                       p.c.hop_length, frames, g.ld, T_out);
    CTTS_CHECK_LAUNCH("stft_ola");
    return CTTS_OK;
}

}  // extern "C"
