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
    size_t basis_A, zero_bias, mel_A, total;
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
    const int mbmax = p.mb_mag > p.mb_mel ? p.mb_mag : p.mb_mel;
    p.zero_bias = take((size_t)mbmax * GEMM_BM);
    p.mel_A = take((size_t)p.mb_mel * (p.kmel / GEMM_KC) * A_TILE);
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
    return rc;
}

size_t ctts_stft_workspace_bytes(const ctts_stft_config* cfg, int32_t batch, int32_t samples) {
    StftPlan p; StftGeom g;
    if (make_stft_plan(cfg, p) || make_stft_geom(p, samples, g) || batch < 1) return 0;
    return (align_up((size_t)batch * p.N * g.ld) + align_up((size_t)batch * p.kmel * g.ld)) * sizeof(float);
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

}  // extern "C"
