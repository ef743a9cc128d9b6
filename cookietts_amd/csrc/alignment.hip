// Attention-alignment scoring on the device: the T2S retry loop scores every generated utterance
// from its [dec, enc] attention map and its gate row; the reference does both on the host
// (utils/model/utils.py:47-56 "using CPU because ...", :59-120) after a device->host copy of the whole map.
//
//   stage 1  align_rows_kernel    grid (dec tiles, B): per decoder step the max / first arg-max over encoder
//                                 tokens (one wave per row) and, per tile, the column sums over the valid steps
//   stage 2  align_finish_kernel  grid (B): fixed-order reduction of the tile sums and the six scores
//   first_over_thresh_kernel      grid (B): first step whose gate reaches the threshold (else T-1)
// All reductions have a fixed order: results are run-to-run identical.
#include "common.h"

namespace ctts {
namespace {

constexpr int AL_ROWS = 32;      // decoder steps per stage-1 workgroup (8 per wave)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void align_rows_kernel(const float* __restrict__ al, const float* __restrict__ out_len,
                                                         float* __restrict__ values, float* __restrict__ idx,
                                                         float* __restrict__ partial, int dec, int enc, int tiles) {
    const int b = blockIdx.y, tile = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float olen = out_len ? out_len[b] : (float)(dec - 1);
    const float* A = al + (size_t)b * dec * enc;
    const int d0 = tile * AL_ROWS;
    // max / first arg-max over the encoder axis (torch.max(dim): first maximal index)
    for (int r = wave; r < AL_ROWS; r += 4) {
        const int d = d0 + r;
        if (d >= dec) break;
        const float* row = A + (size_t)d * enc;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int e = lane; e < enc; e += 64) {
            const float v = row[e];
            if (v > best) { best = v; bi = e; }             // ascending e per lane: first occurrence kept
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) {
            values[(size_t)b * dec + d] = best;
            idx[(size_t)b * dec + d] = (float)(bi == 0x7fffffff ? 0 : bi);
        }
    }
    // column sums over this tile's valid decoder steps (utils.py:81-82)
    for (int e = t; e < enc; e += 256) {
        float s = 0.f;
        for (int r = 0; r < AL_ROWS; ++r) {
            const int d = d0 + r;
            if (d < dec && (float)d < olen) s += A[(size_t)d * enc + e];
        }
        partial[((size_t)b * tiles + tile) * enc + e] = s;
    }
}

__global__ __launch_bounds__(256) void align_finish_kernel(const float* __restrict__ values, const float* __restrict__ idx,
                                                           const float* __restrict__ partial,
                                                           const float* __restrict__ in_len,
                                                           const float* __restrict__ out_len, double* __restrict__ out,
                                                           int dec, int enc, int tiles, float thresh) {
    __shared__ float red[6][4];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float ilen = in_len ? in_len[b] : (float)(enc - 1);
    const float olen = out_len ? out_len[b] : (float)(dec - 1);
    // decoder axis: path length of the arg-max track and the mean peak probability
    float dist = 0.f, vsum = 0.f;
    for (int d = t; d < dec; d += 256) {
        if ((float)d < olen) {
            const float cur = idx[(size_t)b * dec + d];
            const float prev = idx[(size_t)b * dec + (d > 0 ? d - 1 : 0)];
            dist += sqrtf((prev - cur) * (prev - cur) + 1.0f);
            vsum += values[(size_t)b * dec + d];
        }
    }
    // encoder axis: total attention per token
    float emax = -INFINITY, emin = INFINITY, esum = 0.f, miss = 0.f;
    for (int e = t; e < enc; e += 256) {
        float tot = 0.f;
        for (int k = 0; k < tiles; ++k) tot += partial[((size_t)b * tiles + k) * enc + e];
        const bool valid = (float)e < ilen;
        const float z = valid ? tot : 0.f;                   // utils.py:85
        emax = fmaxf(emax, z);
        esum += z;
        emin = fminf(emin, valid ? tot : 1.0f);              // :93
        miss += (valid ? tot : 1e3f) < thresh ? 1.f : 0.f;   // :102-103
    }
    dist = wave_sum(dist); vsum = wave_sum(vsum); esum = wave_sum(esum); miss = wave_sum(miss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        emax = fmaxf(emax, __shfl_xor(emax, o, 64));
        emin = fminf(emin, __shfl_xor(emin, o, 64));
    }
    if (lane == 0) {
        red[0][wave] = dist; red[1][wave] = vsum; red[2][wave] = esum; red[3][wave] = miss;
        red[4][wave] = emax; red[5][wave] = emin;
    }
    __syncthreads();
    if (t == 0) {
        const float D = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const float V = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float E = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        const float M = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        const float mx = fmaxf(fmaxf(red[4][0], red[4][1]), fmaxf(red[4][2], red[4][3]));
        const float mn = fminf(fminf(red[5][0], red[5][1]), fminf(red[5][2], red[5][3]));
        const double opt = sqrt((double)ilen * (double)ilen + (double)olen * (double)olen);   // :69
        double* o = out + (size_t)b * 6;
        o[0] = (double)(D + 1.4142135f) / opt;                                  // diagonalitys   (:79)
        o[1] = (double)((V / (float)dec) * ((float)dec / olen));                // avg_prob       (:98-99)
        o[2] = (double)mx;                                                      // encoder_max_focus
        o[3] = (double)mn;                                                      // encoder_min_focus
        o[4] = (double)((E / (float)enc) * ((float)enc / ilen));                // encoder_avg_focus (:89-90)
        o[5] = (double)(M / ilen);                                              // p_missing_enc  (:103)
    }
}

__global__ __launch_bounds__(256) void first_over_thresh_kernel(const float* __restrict__ x, int32_t* __restrict__ out,
                                                                int T, float thr) {
    __shared__ int red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    int first = T - 1;                                       // utils.py:51: the last step always qualifies
    for (int i = t; i < T - 1; i += 256)
        if (x[(size_t)b * T + i] >= thr) { first = i; break; }   // ascending i per thread: its first hit
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    if ((t & 63) == 0) red[t >> 6] = first;
    __syncthreads();
    if (t == 0) out[b] = min(min(red[0], red[1]), min(red[2], red[3]));
}


// Decoder stop rule (model.py:879-904), evaluated on the device for gate logits of steps [step0, step0 + n):
//   for i > 4:  sig_max[b] = max(sig_max[b], sigmoid(gate[b][i]))            (running max per utterance)
//   if min_b sig_max[b] > threshold:  break_point = min(break_point, i + delay)
//   if i >= break_point:  n_total = i + 1, stop
// state = { float sig_max[batch]; int32 break_point; int32 n_total (-1 while running) } carried between calls.
// One workgroup; thread t owns utterances t, t + 256, ...; the per-step minimum meets in LDS.
__global__ __launch_bounds__(256) void stop_rule_kernel(const float* __restrict__ gate, int batch, int gate_ld, int step0,
                                                        int n, float threshold, int delay, float* __restrict__ sig_max,
                                                        int* __restrict__ ints) {
    __shared__ float red[4];
    __shared__ int s_stop;
    const int t = threadIdx.x;
    int break_point = ints[0];
    if (ints[1] >= 0) return;                              // already stopped in an earlier block
    for (int j = 0; j < n; ++j) {
        const int i = step0 + j;
        float m = INFINITY;
        for (int b = t; b < batch; b += 256) {
            float v = sig_max[b];
            if (i > 4) {
                const float sg = 1.0f / (1.0f + expf(-gate[(size_t)b * gate_ld + i]));
                v = fmaxf(sg, v);
                sig_max[b] = v;
            }
            m = fminf(m, v);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
        if ((t & 63) == 0) red[t >> 6] = m;
        __syncthreads();
        if (t == 0) {
            const float mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
            if (mn > threshold) break_point = min(break_point, i + delay);
            s_stop = (i >= break_point) ? i + 1 : -1;
        }
        __syncthreads();
        const int stop = s_stop;
        if (stop >= 0) {
            if (t == 0) { ints[0] = break_point; ints[1] = stop; }
            return;
        }
        // every thread mirrors thread 0's break_point only through the stop decision; keep thread 0's copy authoritative
    }
    if (t == 0) ints[0] = break_point;
}

__global__ void stop_reset_kernel(float* sig_max, int* ints, int batch, int max_steps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < batch) sig_max[t] = 0.f;
    if (t == 0) { ints[0] = max_steps; ints[1] = -1; }
}

}  // namespace
}  // namespace ctts

using namespace ctts;

extern "C" {

size_t ctts_alignment_workspace_bytes(int32_t batch, int32_t dec, int32_t enc) {
    if (batch < 1 || dec < 1 || enc < 1) return 0;
    const size_t tiles = (dec + AL_ROWS - 1) / AL_ROWS;
    return ((size_t)batch * dec * 2 + (size_t)batch * tiles * enc) * sizeof(float);
}

int ctts_alignment_metric_f32(const float* alignments, const float* input_lengths, const float* output_lengths,
                              int32_t batch, int32_t dec, int32_t enc, float enc_min_thresh, double* out,
                              void* workspace, size_t workspace_bytes, void* stream) {
    CTTS_CHECK_ARG(alignments && out && workspace && batch >= 1 && dec >= 1 && enc >= 1, "alignment_metric: bad argument");
    const size_t need = ctts_alignment_workspace_bytes(batch, dec, enc);
    if (workspace_bytes < need) {
        set_error("alignment_metric: workspace %zu bytes < required %zu", workspace_bytes, need);
        return CTTS_E_WORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    const int tiles = (dec + AL_ROWS - 1) / AL_ROWS;
    float* values = static_cast<float*>(workspace);
    float* idx = values + (size_t)batch * dec;
    float* partial = idx + (size_t)batch * dec;
    hipLaunchKernelGGL(align_rows_kernel, dim3(tiles, batch), dim3(256), 0, s, alignments, output_lengths, values, idx,
                       partial, dec, enc, tiles);
    CTTS_CHECK_LAUNCH("align_rows");
    hipLaunchKernelGGL(align_finish_kernel, dim3(batch), dim3(256), 0, s, values, idx, partial, input_lengths,
                       output_lengths, out, dec, enc, tiles, enc_min_thresh);
    CTTS_CHECK_LAUNCH("align_finish");
    return CTTS_OK;
}

int ctts_first_over_thresh_f32(const float* x, int32_t batch, int32_t T, float threshold, int32_t* out, void* stream) {
    CTTS_CHECK_ARG(x && out && batch >= 1 && T >= 1, "first_over_thresh: bad argument");
    hipLaunchKernelGGL(first_over_thresh_kernel, dim3(batch), dim3(256), 0, as_stream(stream), x, out, T, threshold);
    CTTS_CHECK_LAUNCH("first_over_thresh");
    return CTTS_OK;
}

size_t ctts_taco_stop_state_bytes(int32_t batch) {
    if (batch < 1) return 0;
    return ((size_t)batch + 2) * sizeof(float);
}

int ctts_taco_stop_reset(void* state, int32_t batch, int32_t max_decoder_steps, void* stream) {
    CTTS_CHECK_ARG(state && batch >= 1 && max_decoder_steps >= 1, "stop_reset: bad argument");
    float* sig = static_cast<float*>(state);
    hipLaunchKernelGGL(stop_reset_kernel, dim3((batch + 255) / 256), dim3(256), 0, as_stream(stream), sig,
                       reinterpret_cast<int*>(sig + batch), batch, max_decoder_steps);
    CTTS_CHECK_LAUNCH("stop_reset");
    return CTTS_OK;
}

int ctts_taco_stop_rule_f32(const float* gate_logits, int32_t batch, int32_t gate_ld, int32_t step0, int32_t n_steps,
                            float gate_threshold, int32_t gate_delay, void* state, void* stream) {
    CTTS_CHECK_ARG(gate_logits && state && batch >= 1 && step0 >= 0 && n_steps >= 0 && step0 + n_steps <= gate_ld,
                   "stop_rule: bad argument");
    float* sig = static_cast<float*>(state);
    hipLaunchKernelGGL(stop_rule_kernel, dim3(1), dim3(256), 0, as_stream(stream), gate_logits, batch, gate_ld, step0,
                       n_steps, gate_threshold, gate_delay, sig, reinterpret_cast<int*>(sig + batch));
    CTTS_CHECK_LAUNCH("stop_rule");
    return CTTS_OK;
}

}  // extern "C"
