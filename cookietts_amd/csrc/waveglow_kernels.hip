// Non-GEMM kernels of the WaveGlow path (all HBM-bound byte/float shuffles around the
// MFMA conv-GEMM) and the weight-ingest kernels.  gfx950 only.
#include "waveglow_kernels.h"
#include "tuning.h"

namespace ctts {

namespace {

// ------------------------------------------------------------------ weight ingest ----
// w[o][:] = v[o][:] * (g[o] / ||v[o][:]||): one workgroup per output channel.
__global__ __launch_bounds__(256) void fold_weightnorm_kernel(const float* __restrict__ v,
                                                              const float* __restrict__ g,
                                                              float* __restrict__ w, int fan) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const float* vr = v + (size_t)o * fan;
    float s = 0.f;
    for (int i = threadIdx.x; i < fan; i += 256) { const float x = vr[i]; s = fmaf(x, x, s); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float scale = g[o] / sqrtf(tot);
    float* wr = w + (size_t)o * fan;
    for (int i = threadIdx.x; i < fan; i += 256) wr[i] = vr[i] * scale;
}

// dst packed [MB][nch_total][16][bm];  fills K range [k_off, k_off + ksrc):
//   dst(mb, k, r) = src[(src_row_off + dense_row(mb, r)) * src_row_stride + (k - k_off) * src_k_stride]
__global__ __launch_bounds__(256) void pack_a_kernel(float* __restrict__ dst, const float* __restrict__ src, int bm,
                                                     int nch_total, int k_off, int ksrc, int epi, int C, int M,
                                                     long long src_row_off, long long src_row_stride,
                                                     int src_k_stride, int k_group, int k_member) {
    const int mb = blockIdx.y;
    const int k = blockIdx.x;  // 0..ksrc-1
    const int r = threadIdx.x;
    if (r >= bm) return;
    const int drow = gemm_dense_row(epi, bm, mb, r, C, M);
    // k_group > 1: this source is member k_member of a round-robin group (GemmArgs::interleave): its
    // chunk j lands at chunk j*k_group + k_member of the group's K range starting at k_off
    const int kk = k_group > 1 ? k_off + ((k / GEMM_KC) * k_group + k_member) * GEMM_KC + k % GEMM_KC : k_off + k;
    dst[((size_t)mb * nch_total + kk / GEMM_KC) * (GEMM_KC * bm) + (kk % GEMM_KC) * bm + r] =
        drow >= 0 ? src[(src_row_off + drow) * src_row_stride + (long long)k * src_k_stride] : 0.f;
}

// dst[mb*bm + r] = src0[off0 + dense_row] (+ src1[off1 + dense_row])
__global__ __launch_bounds__(256) void pack_bias_kernel(float* __restrict__ dst, int bm, const float* __restrict__ src0,
                                                        long long off0, const float* __restrict__ src1,
                                                        long long off1, int epi, int C, int M) {
    const int mb = blockIdx.x, r = threadIdx.x;
    if (r >= bm) return;
    const int row = gemm_dense_row(epi, bm, mb, r, C, M);
    float v = 0.f;
    if (row >= 0 && src0) {
        v = src0[off0 + row];
        if (src1) v += src1[off1 + row];
    }
    dst[mb * bm + r] = v;
}

// ------------------------------------------------------------ upsample + squeeze ----
// spect[b][o*G+g][pad + l] = bias[o] + sum_{j<taps} sum_i mel[b][i][q-j] * W[i][o][p + hop*j],
//   t = G*l + g = hop*q + p.            (glow.py:318-324)
// Workgroup: one batch item, UQ consecutive frames q, UO output channels; thread <-> phase p
// (hop == 256 threads).  Each W element is loaded once per (i, j, o) and reused for UQ frames
// from registers; mel values are wave-uniform LDS broadcasts.  The [UO][UQ][256] result is
// transposed through LDS so that every spect row is written as contiguous runs of
// UQ*hop/G time steps.
constexpr int UP_UQ = 8;
constexpr int UP_UO = 8;

template <int HOP, int G>
__global__ __launch_bounds__(HOP) void upsample_squeeze_kernel(const float* __restrict__ mel,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ spect, int n_mel, int F,
                                                               int win, int ld, int pad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int taps = win / HOP;
    const int nfr = UP_UQ + taps - 1;            // frames of mel this block touches
    float* smel = smem;                          // [n_mel][nfr]
    float* stile = smem + n_mel * nfr;           // [UO/2][G][UQ*HOP/G + 8] transposed result
    const int p = threadIdx.x;
    const int q0 = blockIdx.x * UP_UQ;
    const int o0 = blockIdx.y * UP_UO;
    const int b = blockIdx.z;
    const float* melb = mel + (size_t)b * n_mel * F;
    for (int idx = p; idx < n_mel * nfr; idx += HOP) {
        const int i = idx / nfr, ff = idx % nfr;
        const int f = q0 - (taps - 1) + ff;      // frame index, may be <0 or >=F
        smel[idx] = (f >= 0 && f < F) ? melb[(size_t)i * F + f] : 0.f;
    }
    __syncthreads();

    float acc[UP_UO][UP_UQ];
#pragma unroll
    for (int o = 0; o < UP_UO; ++o)
#pragma unroll
        for (int q = 0; q < UP_UQ; ++q) acc[o][q] = 0.f;

    for (int i = 0; i < n_mel; ++i) {
        const float* sm = smel + i * nfr;
        for (int j = 0; j < taps; ++j) {
            float w[UP_UO];
#pragma unroll
            for (int o = 0; o < UP_UO; ++o)
                w[o] = (o0 + o < n_mel) ? W[((size_t)i * n_mel + o0 + o) * win + j * HOP + p] : 0.f;
            // frame q0+q uses mel frame q0+q-j  -> smel index (taps-1) + q - j
#pragma unroll
            for (int q = 0; q < UP_UQ; ++q) {
                const float m = sm[taps - 1 + q - j];
#pragma unroll
                for (int o = 0; o < UP_UO; ++o) acc[o][q] = fmaf(m, w[o], acc[o][q]);
            }
        }
    }
    // transpose through LDS (two passes of UO/2 channels): t = HOP*q + p -> (g = p % G, l_local = q*(HOP/G) + p/G)
    constexpr int LPQ = HOP / G;                 // time steps per frame
    constexpr int ROW = UP_UQ * LPQ;             // time steps per block
    constexpr int ROWP = ROW + 8;                // +8 floats: 2-way (free) instead of 8-way write conflicts
    constexpr int HO = UP_UO / 2;
    const int g = p % G, lq = p / G;
    const int L = F * LPQ;
    const int l0 = q0 * LPQ;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int o = 0; o < HO; ++o)
#pragma unroll
            for (int q = 0; q < UP_UQ; ++q) stile[(o * G + g) * ROWP + q * LPQ + lq] = acc[half * HO + o][q];
        __syncthreads();
        for (int idx = p; idx < HO * G * ROW; idx += HOP) {
            const int row = idx / ROW, ll = idx % ROW;
            const int o = o0 + half * HO + row / G;
            if (o < n_mel && l0 + ll < L)
                spect[((size_t)b * n_mel * G + (size_t)(o0 + half * HO) * G + row) * ld + pad + l0 + ll] =
                    stile[row * ROWP + ll] + bias[o];
        }
    }
}

// Any hop / n_group (glow.py:226-265 takes them freely; the kernel above is the benchmark's hop 256 / n_group 8): one thread per
// output sample t of (b, o), 256 consecutive t per workgroup - consecutive t are consecutive phases p of W (coalesced) and
// share the mel frames.  ~10x the tuned kernel's time; the upsampling is < 2 % of an inference either way.
__global__ __launch_bounds__(256) void upsample_squeeze_generic_kernel(const float* __restrict__ mel, const float* __restrict__ W,
                                                                       const float* __restrict__ bias, float* __restrict__ spect,
                                                                       int n_mel, int F, int win, int hop, int G, int ld, int pad) {
    const int b = blockIdx.z, o = blockIdx.y;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int taps = win / hop;
    if (t >= (long long)F * hop) return;
    const int q = (int)(t / hop), p = (int)(t % hop);
    const float* melb = mel + (size_t)b * n_mel * F;
    float acc = bias[o];
    for (int i = 0; i < n_mel; ++i) {
        const float* wr = W + ((size_t)i * n_mel + o) * win + p;
        for (int j = 0; j < taps; ++j)
            if (q - j >= 0) acc = fmaf(melb[(size_t)i * F + q - j], wr[j * hop], acc);
    }
    spect[((size_t)b * n_mel * G + (size_t)o * G + (int)(t % G)) * ld + pad + t / G] = acc;
}

// MFMA form of the benchmark's shape (n_mel 80, hop 256, win 4 hop, n_group 8).  The transposed conv is the GEMM
//   out[(o, p)][(b, q)] = sum_{(i, j)} W[i][o][p + hop j] * mel[b][i][q - j],      M = n_mel hop = 20 480, K = 4 n_mel = 320,
// 11.8 GFLOP per 900-frame utterance - the VALU kernel above runs it at 29 TFLOP/s.  Here W is the STATIONARY operand: a workgroup
// owns one output channel o (256 phases = 16 m-tiles of 16 rows, two per wave) and keeps its 256 x 320 slab in registers as
// `v_mfma_f32_16x16x4_f32` A fragments (160 VGPRs per lane), the mel frames of its chunk sit in LDS ([i][frame], read as B
// fragments: lane (n, j) = frame n - j of channel i, k = 4 i + j), and the chunk's frames stream past 32 at a time.
// Row order of the m-tiles is chosen for the SQUEEZED output: wave g holds the phases p = 8 r + g (r = 0..31), which are the 32
// consecutive time steps of frame q in spect row o G + g - a lane's four accumulators are one aligned float4 of that row and a
// store instruction writes 64-byte runs.  W comes pre-packed in fragment order (upsample_pack_mfma_kernel, at pack time).
constexpr int UM_NMEL = 80, UM_HOP = 256, UM_G = 8, UM_TAPS = 4;
constexpr int UM_KCH = UM_NMEL * UM_TAPS / 4;       // k-chunks of 4 = input channels
constexpr int UM_CHUNK_MAX = 320;                   // frames per workgroup (LDS: n_mel x (320 + 4) floats = 101 KiB)

__global__ __launch_bounds__(256) void upsample_pack_mfma_kernel(const float* __restrict__ W, float* __restrict__ Wp) {
    // Wp[((o G + g) 2 + mt)][q = i / 4][lane][e = i % 4] = W[i][o][hop j + 8 (16 mt + lane % 16) + g],  j = lane / 16
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)UM_NMEL * UM_NMEL * UM_HOP * UM_TAPS) return;
    const int e = idx % 4, lane = (idx / 4) % 64, q = (idx / 256) % (UM_KCH / 4);
    const int mt = (idx / (256 * (UM_KCH / 4))) % 2, g = (idx / (512 * (UM_KCH / 4))) % UM_G;
    const int o = (int)(idx / ((size_t)512 * (UM_KCH / 4) * UM_G));
    const int i = 4 * q + e, j = lane / 16, ph = 8 * (16 * mt + lane % 16) + g;
    Wp[idx] = W[((size_t)i * UM_NMEL + o) * (UM_HOP * UM_TAPS) + UM_HOP * j + ph];
}

__global__ __launch_bounds__(512) void upsample_squeeze_mfma_kernel(const float* __restrict__ mel, const float* __restrict__ Wp,
                                                                    const float* __restrict__ bias, float* __restrict__ spect,
                                                                    int F, int chunk, int nchunks, int srow, int ld, int pad) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smel[];       // [n_mel][srow]: frame (f0 - 3 + x) at x, zeros outside [0, F)
    const int o = blockIdx.x, b = blockIdx.y / nchunks, f0 = (blockIdx.y % nchunks) * chunk;
    const int fend = min(F, f0 + chunk);
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    // A fragments first: their latency hides behind the staging of the mel chunk
    f32x4 a[2][UM_KCH / 4];
    const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((size_t)(o * UM_G + g) * 2) * (UM_KCH / 4) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < UM_KCH / 4; ++q) a[mt][q] = __builtin_nontemporal_load(wp + (mt * (UM_KCH / 4) + q) * 64);
    const float* melb = mel + (size_t)b * UM_NMEL * F;
    // wave g stages channels g, g + 8, ...: <= 6 independent loads per channel in flight (srow <= 324)
    for (int i = g; i < UM_NMEL; i += 8) {
        float v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int f = f0 - (UM_TAPS - 1) + lane + 64 * u;
            v[u] = (lane + 64 * u < srow && f >= 0 && f < F) ? melb[(size_t)i * F + f] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (lane + 64 * u < srow) smel[i * srow + lane + 64 * u] = v[u];
    }
    __syncthreads();
    const float bo = bias[o];
    const int n = lane & 15, jq = lane >> 4;
    const float* sb = smel + (UM_TAPS - 1) + n - jq;                   // B fragment of k-chunk i, frame tile t: sb[i srow + 16 t]
    float* row = spect + ((size_t)b * UM_NMEL * UM_G + (size_t)o * UM_G + g) * ld + pad + 4 * jq;
    for (int t0 = 0; f0 + 16 * t0 < fend; t0 += 2) {
        f32x4 acc[2][2] = {};
        float b0 = sb[16 * t0], b1 = sb[16 * t0 + 16];
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < UM_KCH; ++i) {
            const float c0 = b0, c1 = b1;
            if (i + 1 < UM_KCH) { b0 = sb[(i + 1) * srow + 16 * t0]; b1 = sb[(i + 1) * srow + 16 * t0 + 16]; }   // one k-chunk ahead of its use
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i / 4][i % 4], c0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][i / 4][i % 4], c1, acc[mt][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the next chunk's fragment read (one ds_read2) ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // ... ahead of this chunk's four MFMAs
        }
        // D: lane (n, jq) holds rows 4 jq + v of m-tile mt = time steps 32 f + 16 mt + 4 jq + v of frame f = f0 + 16 t + n
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int f = f0 + 16 * (t0 + tt) + n;
            if (f < fend) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 v = acc[mt][tt];
                    v.x += bo; v.y += bo; v.z += bo; v.w += bo;
                    *reinterpret_cast<f32x4*>(row + (size_t)f * (UM_HOP / UM_G) + 16 * mt) = v;
                }
            }
        }
    }
}

// --------------------------------------------------------------------- WN start ----
// x[b][c][pad + n] = bs[c] + sum_{j<h} Ws[c][j] * audio[b][ch_off + j][n]     (glow.py:189)
template <int H>
__global__ __launch_bounds__(256) void wn_start_kernel(const float* __restrict__ audio, const float* __restrict__ Ws,
                                                       const float* __restrict__ bs, float* __restrict__ x,
                                                       int C, int G, int ch_off, int L, int ld, int pad) {
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int b = blockIdx.z;
    if (n >= L) return;
    float4 a[H];
#pragma unroll
    for (int j = 0; j < H; ++j)
        a[j] = *reinterpret_cast<const float4*>(audio + ((size_t)b * G + ch_off + j) * L + n);
    const int c0 = blockIdx.y * 32;
    float* xb = x + (size_t)b * C * ld + pad + n;
    for (int c = c0; c < c0 + 32 && c < C; ++c) {
        float4 v;
        const float bias = bs[c];
        v.x = v.y = v.z = v.w = bias;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const float w = Ws[c * H + j];
            v.x = fmaf(w, a[j].x, v.x); v.y = fmaf(w, a[j].y, v.y);
            v.z = fmaf(w, a[j].z, v.z); v.w = fmaf(w, a[j].w, v.w);
        }
        *reinterpret_cast<float4*>(xb + (size_t)c * ld) = v;
    }
}

// -------------------------------------------------------------------- flow tail ----
// e = Wend * out + bend; (b, log_s) = (e[:h], e[h:]); a1 = (a1 - b) / exp(log_s);
// audio[ch_off : ch_off+2h] = Winv * [a0; a1]; optional un-squeeze to wave.
// (glow.py:222, 337-340, 349).  Workgroup = 4 waves x 256 time steps; each wave reduces a
// quarter of the C skip channels with 16-byte loads, partial sums meet in LDS.
template <int H>
__global__ __launch_bounds__(256) void flow_tail_kernel(const float* __restrict__ out, float* __restrict__ audio,
                                                        float* __restrict__ wave, const float* __restrict__ Wend,
                                                        const float* __restrict__ bend, const float* __restrict__ Winv,
                                                        int C, int G, int ch_off, int L, int ld, int pad) {
    constexpr int E = 2 * H;
    __shared__ __attribute__((aligned(16))) float part[3][E][256];
    __shared__ float sWinv[E * E];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + lane * 4;
    if (threadIdx.x < E * E) sWinv[threadIdx.x] = Winv[threadIdx.x];
    float4 e[E];
#pragma unroll
    for (int j = 0; j < E; ++j) e[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int cq = C / 4;
    const int cbeg = __builtin_amdgcn_readfirstlane(wv * cq);
    // columns beyond L inside the padded row are readable (zero / never stored)
    const float* ob = out + (size_t)b * C * ld + pad + n;
    for (int c = cbeg; c < cbeg + cq; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(ob + (size_t)c * ld);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const float w = Wend[j * C + c];
            e[j].x = fmaf(w, v.x, e[j].x); e[j].y = fmaf(w, v.y, e[j].y);
            e[j].z = fmaf(w, v.z, e[j].z); e[j].w = fmaf(w, v.w, e[j].w);
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int j = 0; j < E; ++j) *reinterpret_cast<float4*>(&part[wv - 1][j][lane * 4]) = e[j];
    }
    __syncthreads();
    if (wv != 0 || n >= L) return;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const float bj = bend[j];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 pv = *reinterpret_cast<const float4*>(&part[q][j][lane * 4]);
            e[j].x += pv.x; e[j].y += pv.y; e[j].z += pv.z; e[j].w += pv.w;
        }
        e[j].x += bj; e[j].y += bj; e[j].z += bj; e[j].w += bj;
    }
    float* ab = audio + ((size_t)b * G + ch_off) * L + n;
    float4 a[E];
#pragma unroll
    for (int j = 0; j < E; ++j) a[j] = *reinterpret_cast<const float4*>(ab + (size_t)j * L);
#pragma unroll
    for (int j = 0; j < H; ++j) {
        a[H + j].x = (a[H + j].x - e[j].x) / expf(e[H + j].x);
        a[H + j].y = (a[H + j].y - e[j].y) / expf(e[H + j].y);
        a[H + j].z = (a[H + j].z - e[j].z) / expf(e[H + j].z);
        a[H + j].w = (a[H + j].w - e[j].w) / expf(e[H + j].w);
    }
    float4 m[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const float w = sWinv[i * E + j];
            s.x = fmaf(w, a[j].x, s.x); s.y = fmaf(w, a[j].y, s.y);
            s.z = fmaf(w, a[j].z, s.z); s.w = fmaf(w, a[j].w, s.w);
        }
        m[i] = s;
    }
    if (wave == nullptr) {
#pragma unroll
        for (int i = 0; i < E; ++i) *reinterpret_cast<float4*>(ab + (size_t)i * L) = m[i];
    } else if constexpr (E % 4 == 0) {
        // un-squeeze: wave[b][G*l + g] = audio[b][g][l]; on the last flow ch_off == 0 and E == G
        float* wb = wave + (size_t)b * G * L + (size_t)n * G;
        const float* mf = reinterpret_cast<const float*>(m);   // m[i].{x,y,z,w} = channel i, step n+{0..3}
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < E; i += 4) {
                float4 v = make_float4(mf[(i + 0) * 4 + s], mf[(i + 1) * 4 + s], mf[(i + 2) * 4 + s], mf[(i + 3) * 4 + s]);
                *reinterpret_cast<float4*>(wb + s * G + i) = v;
            }
    }
}

}  // namespace

// ----------------------------------------------------------------- host launchers ----

int launch_fold_weightnorm(const float* v, const float* g, float* w, int out_ch, int fan, hipStream_t s) {
    CTTS_CHECK_ARG(out_ch > 0 && fan > 0, "fold_weightnorm: out_ch=%d fan=%d", out_ch, fan);
    hipLaunchKernelGGL(fold_weightnorm_kernel, dim3(out_ch), dim3(256), 0, s, v, g, w, fan);
    CTTS_CHECK_LAUNCH("fold_weightnorm");
    return CTTS_OK;
}

int launch_pack_a(float* dst, const float* src, int bm, int MB, int nch_total, int k_off, int ksrc, int epi, int C, int M,
                  long long src_row_off, long long src_row_stride, int src_k_stride, hipStream_t s, int k_group,
                  int k_member) {
    CTTS_CHECK_ARG(k_off >= 0 && ksrc > 0 && k_off + ksrc * (k_group > 1 ? k_group : 1) <= nch_total * GEMM_KC,
                   "pack_a: k range");
    CTTS_CHECK_ARG(k_group <= 1 || (ksrc % GEMM_KC == 0 && k_member >= 0 && k_member < k_group), "pack_a: k group");
    hipLaunchKernelGGL(pack_a_kernel, dim3(ksrc, MB), dim3(256), 0, s, dst, src, bm, nch_total, k_off, ksrc, epi, C, M,
                       src_row_off, src_row_stride, src_k_stride, k_group, k_member);
    CTTS_CHECK_LAUNCH("pack_a");
    return CTTS_OK;
}

int launch_pack_bias(float* dst, int bm, int MB, const float* src0, long long off0, const float* src1, long long off1,
                     int epi, int C, int M, hipStream_t s) {
    hipLaunchKernelGGL(pack_bias_kernel, dim3(MB), dim3(256), 0, s, dst, bm, src0, off0, src1, off1, epi, C, M);
    CTTS_CHECK_LAUNCH("pack_bias");
    return CTTS_OK;
}

bool upsample_mfma_shape(int n_mel, int win, int hop, int G) {
    return n_mel == UM_NMEL && hop == UM_HOP && win == UM_HOP * UM_TAPS && G == UM_G;
}

int launch_upsample_pack_mfma(const float* W, float* Wp, hipStream_t s) {
    const size_t n = (size_t)UM_NMEL * UM_NMEL * UM_HOP * UM_TAPS;
    hipLaunchKernelGGL(upsample_pack_mfma_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, Wp);
    CTTS_CHECK_LAUNCH("upsample_pack_mfma");
    return CTTS_OK;
}

int launch_upsample_squeeze(const float* mel, const float* W, const float* Wp, const float* bias, float* spect, int batch,
                            int n_mel, int F, int win, int hop, int G, int ld, int pad, hipStream_t s) {
    CTTS_CHECK_ARG(win % hop == 0 && hop % G == 0, "upsample_squeeze: win %d hop %d n_group %d", win, hop, G);
    if (Wp && upsample_mfma_shape(n_mel, win, hop, G) && ld % 4 == 0 && pad % 4 == 0 && (reinterpret_cast<uintptr_t>(spect) & 15) == 0 &&
        !tuning().up_no_mfma) {
        // chunks of <= 320 frames, equal up to a 16-frame tile (900 frames: 304 + 304 + 292): 80 x 3 workgroups per utterance
        const int nchunks = (F + UM_CHUNK_MAX - 1) / UM_CHUNK_MAX;
        const int chunk = ((F + nchunks - 1) / nchunks + 15) / 16 * 16;
        const int srow = (chunk + 31) / 32 * 32 + 4;           // frames f0 - 3 ... f0 + (chunk rounded to a pair of tiles)
        const int lds = UM_NMEL * srow * (int)sizeof(float);
        // once per inference, so set on every call: the attribute belongs to the current device's copy of the kernel
        if (lds > 64 * 1024)
            CTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(upsample_squeeze_mfma_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, UM_NMEL * (UM_CHUNK_MAX + 4) * (int)sizeof(float)));
        hipLaunchKernelGGL(upsample_squeeze_mfma_kernel, dim3(UM_NMEL, (unsigned)(batch * nchunks)), dim3(512), lds, s, mel, Wp, bias,
                           spect, F, chunk, nchunks, srow, ld, pad);
        CTTS_CHECK_LAUNCH("upsample_squeeze_mfma");
        return CTTS_OK;
    }
    if (hop != 256 || G != 8) {
        dim3 ggrid((unsigned)(((long long)F * hop + 255) / 256), n_mel, batch);
        hipLaunchKernelGGL(upsample_squeeze_generic_kernel, ggrid, dim3(256), 0, s, mel, W, bias, spect, n_mel, F, win, hop, G, ld, pad);
        CTTS_CHECK_LAUNCH("upsample_squeeze_generic");
        return CTTS_OK;
    }
    const int taps = win / hop;
    const size_t smem = ((size_t)n_mel * (UP_UQ + taps - 1) + (size_t)(UP_UO / 2) * G * (UP_UQ * (hop / G) + 8)) * sizeof(float);
    CTTS_CHECK_ARG(smem <= 64 * 1024, "upsample_squeeze: LDS %zu", smem);
    dim3 grid((F + UP_UQ - 1) / UP_UQ, (n_mel + UP_UO - 1) / UP_UO, batch);
    hipLaunchKernelGGL((upsample_squeeze_kernel<256, 8>), grid, dim3(256), smem, s, mel, W, bias, spect, n_mel, F,
                       win, ld, pad);
    CTTS_CHECK_LAUNCH("upsample_squeeze");
    return CTTS_OK;
}

int launch_wn_start(const float* audio, const float* Ws, const float* bs, float* x, int batch, int C, int G,
                    int ch_off, int n_half, int L, int ld, int pad, hipStream_t s) {
    CTTS_CHECK_ARG(L % 4 == 0, "wn_start: L=%d not a multiple of 4", L);
    dim3 grid((L / 4 + 255) / 256, (C + 31) / 32, batch);
#define CTTS_START_CASE(H)                                                                                     \
    case H:                                                                                                    \
        hipLaunchKernelGGL(wn_start_kernel<H>, grid, dim3(256), 0, s, audio, Ws, bs, x, C, G, ch_off, L, ld,   \
                           pad);                                                                               \
        break;
    switch (n_half) {
        CTTS_START_CASE(1) CTTS_START_CASE(2) CTTS_START_CASE(3) CTTS_START_CASE(4)
        CTTS_START_CASE(5) CTTS_START_CASE(6) CTTS_START_CASE(7) CTTS_START_CASE(8)
        default:
            set_error("wn_start: n_half=%d unsupported (1..8)", n_half);
            return CTTS_E_ARG;
    }
#undef CTTS_START_CASE
    CTTS_CHECK_LAUNCH("wn_start");
    return CTTS_OK;
}

int launch_flow_tail(const float* out, float* audio, float* wave, const float* Wend, const float* bend,
                     const float* Winv, int batch, int C, int G, int ch_off, int n_half, int L, int ld, int pad,
                     hipStream_t s) {
    CTTS_CHECK_ARG(L % 4 == 0 && C % 4 == 0, "flow_tail: L=%d C=%d", L, C);
    CTTS_CHECK_ARG(wave == nullptr || (ch_off == 0 && 2 * n_half == G && G % 4 == 0),
                   "flow_tail: un-squeeze needs the full group (ch_off=%d n_half=%d G=%d)", ch_off, n_half, G);
    dim3 grid((L + 255) / 256, batch);
#define CTTS_TAIL_CASE(H)                                                                                       \
    case H:                                                                                                     \
        hipLaunchKernelGGL(flow_tail_kernel<H>, grid, dim3(256), 0, s, out, audio, wave, Wend, bend, Winv, C, G, \
                           ch_off, L, ld, pad);                                                                 \
        break;
    switch (n_half) {
        CTTS_TAIL_CASE(1) CTTS_TAIL_CASE(2) CTTS_TAIL_CASE(3) CTTS_TAIL_CASE(4)
        CTTS_TAIL_CASE(5) CTTS_TAIL_CASE(6) CTTS_TAIL_CASE(7) CTTS_TAIL_CASE(8)
        default:
            set_error("flow_tail: n_half=%d unsupported (1..8)", n_half);
            return CTTS_E_ARG;
    }
#undef CTTS_TAIL_CASE
    CTTS_CHECK_LAUNCH("flow_tail");
    return CTTS_OK;
}

}  // namespace ctts
