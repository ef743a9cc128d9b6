/*
 * cookietts_hip.h - C ABI of the MI355X (gfx950) mel-to-wave hot path.
 *
 * The reference (CookiePPP/cookietts) has no FFI layer: its hot path sits behind Python
 * nn.Module classes that dispatch stock PyTorch ops.  This header is the
 * boundary those ops are replaced at.  Every entry point cites the reference code it
 * replaces (paths relative to /root/reference/CookieTTS/).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - All data pointers are DEVICE pointers owned by the caller (PyTorch-ROCm in the
 *     Python host).  The library never allocates or frees device memory; scratch is a
 *     caller-provided workspace sized by ctts_*_workspace_bytes().
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All work is
 *     enqueued asynchronously on it; no entry point synchronises.
 *   - Return value: 0 = ok, negative = error (CTTS_E_*); ctts_last_error() returns a
 *     thread-local human-readable message for the last failure on this thread.
 *   - No hidden MODEL state: the immutable pair (config, packed weight blob) is the
 *     "plan" - arithmetic mode, shapes and weights travel in it - and calls are thread-safe
 *     per (workspace, stream).  Process-wide are only the A/B and timing knobs latched from
 *     the environment at the first launch (ctts_tuning_reload / ctts_tuning_flags below:
 *     launch shapes, the persistent decoder's poll delays - never arithmetic except where a
 *     knob's line says "summation order"), and the profiling stamp buffer of
 *     ctts_taco_decoder_persistent_debug.
 *
 * "Padded activation layout": internal activation tensors are [B][rows][ld] fp32 with
 * the L valid time steps of a row at columns [pad, pad+L) and zeros in the halo, so
 * dilated-conv taps read x[l +- d] without bounds checks (ld, pad from
 * ctts_waveglow_geometry()).
 */
#ifndef COOKIETTS_HIP_H
#define COOKIETTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTTS_OK 0
#define CTTS_E_ARG (-1)       /* bad argument / unsupported shape */
#define CTTS_E_LAUNCH (-2)    /* HIP launch or runtime error */
#define CTTS_E_WORKSPACE (-3) /* workspace too small */
#define CTTS_E_ABORT (-4)     /* an earlier call on this workspace gave up a bounded device-side wait (ABI 6; ctts_waveflow_abort_status) */

#define CTTS_ABI_VERSION 7   /* 2: ctts_waveglow_config.speaker_embed_dim, ctts_waveglow_flow_weights.speaker_embed
                              * 3: gated_unit / merge_res_skip in ctts_waveflow_config and ctts_wgax_config
                              * 4: f32_gemm_mode in ctts_waveglow_config, ctts_waveflow_config, ctts_wgax_config and
                              *    ctts_conv1d_desc (the arithmetic mode belongs to the model, not to the process);
                              *    ctts_tuning_reload; the persistent decoder's control words are exactly the last 64 bytes
                              * 5: ctts_set_f32_gemm_mode / ctts_get_f32_gemm_mode speak CTTS_GEMM_* (one encoding for the
                              *    process default and the config structs' field) and are deprecated; ctts_last_gemm_loop
                              * 6: no process-global state behind the ABI: the process-wide GEMM mode is gone (set fails for
                              *    the split modes), profiles are caller-owned handles (ctts_profile_create / _bind /
                              *    _collect(handle, ...) / _destroy replace ctts_profile_enable / _collect(which, ...)), a
                              *    row-queue abort is a status (CTTS_E_ABORT from the next ctts_waveflow_inverse_* on that
                              *    workspace, ctts_waveflow_abort_status) besides the NaN audio; IEEE-half variant of the
                              *    reduced-precision WaveGlow path (ctts_waveglow_pack_flow_f16 / ctts_waveglow_infer_spk_f16)
                              * 7: ctts_set_f32_gemm_mode / ctts_get_f32_gemm_mode are gone (dead since 6); the Tacotron decoder and
                              *    the packed-sequence LSTM take batches up to 256 (ctts_taco_decoder_max_batch, the batched MFMA
                              *    form: larger packed blobs and workspaces - re-query the *_bytes functions); glow.py-class
                              *    WaveGlow: any hop_length, n_group 4 / 8 / 12 / 16 */

/* Main loop of the fp32 conv-GEMM a model's launches use (field f32_gemm_mode of the config structs). */
#define CTTS_GEMM_DEFAULT 0  /* the library default: fp32 MFMA */
#define CTTS_GEMM_F32 1      /* v_mfma_f32_32x32x2_f32: exact fp32 products */
#define CTTS_GEMM_BF16X3 2   /* split bf16: hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulation */
#define CTTS_GEMM_BF16X6 3   /* 3-way split (24 mantissa bits): the six products >= 2^-16 (hh, hm, mh, hl, lh, mm); fp32-grade */
#define CTTS_N_SPEAKERS 512  /* rows of every speaker-embedding table (glow.py:129, efficient_model_ax.py:60) */

int ctts_abi_version(void);
const char* ctts_last_error(void);

/* Constructor arguments of the reference model that shape the path:
 * _4_mtw/waveglow/glow.py:226-265 (WaveGlow.__init__) and :114-186 (WN.__init__). */
typedef struct ctts_waveglow_config {
    int32_t n_mel_channels; /* 80 */
    int32_t n_group;        /* 8 */
    int32_t n_flows;        /* 12 */
    int32_t n_early_every;  /* 4 */
    int32_t n_early_size;   /* 2 */
    int32_t win_length;     /* 1024: ConvTranspose1d kernel */
    int32_t hop_length;     /* 256: ConvTranspose1d stride */
    int32_t n_layers;       /* 8 WN layers, dilation 2^i */
    int32_t n_channels;     /* 512 WN channels (multiple of 128) */
    int32_t kernel_size;    /* 3 */
    int32_t cond_hidden;    /* 256, hard-coded at glow.py:153 */
    int32_t speaker_embed_dim; /* WN_config['speaker_embed_dim'] (glow.py:116,129-133): 0 = single speaker */
    int32_t f32_gemm_mode;  /* CTTS_GEMM_*: main loop of this model's fp32 GEMMs (fp32 entry points only) */
} ctts_waveglow_config;

typedef struct ctts_waveglow_geometry {
    int32_t steps;          /* L = frames*hop/n_group */
    int32_t ld;             /* row stride of padded activation tensors (floats) */
    int32_t pad;            /* left halo (floats) */
    int32_t n_remaining;    /* channels of the initial latent (glow.py:265) */
} ctts_waveglow_geometry;

int ctts_waveglow_geometry_for(const ctts_waveglow_config* cfg, int32_t frames,
                               ctts_waveglow_geometry* out);

/* ---- weight ingest (replaces torch weight_norm hooks + module construction) ---------- */

/* w[o][:] = v[o][:] * (g[o] / ||v[o][:]||_2): torch.nn.utils.weight_norm as applied at
 * glow.py:135-137,155-165,171-173,184 (checkpoint keys *.weight_g / *.weight_v). */
int ctts_fold_weightnorm_f32(const float* v, const float* g, float* w, int32_t out_ch,
                             int32_t fan, void* stream);

/* Dense (already weight-norm-folded) fp32 weights of ONE flow, in the reference's own
 * tensor layouts (state_dict shapes, glow.py:110-186 / :65-83). */
typedef struct ctts_waveglow_flow_weights {
    const float* start_w;      /* [C][n_half]              WN.k.start */
    const float* start_b;      /* [C] */
    const float* cond_w[3];    /* [H][n_mel*G], [H][H], [2*C*n_layers][H]   WN.k.cond_layers.j */
    const float* cond_b[3];
    const float* const* in_w;  /* n_layers x [2C][C][ks]   WN.k.in_layers.i */
    const float* const* in_b;  /* n_layers x [2C] */
    const float* const* rs_w;  /* n_layers x [2C or C][C]  WN.k.res_skip_layers.i */
    const float* const* rs_b;
    const float* end_w;        /* [2*n_half][C]            WN.k.end */
    const float* end_b;        /* [2*n_half] */
    const float* w_inverse;    /* [c][c] fp32 inverse of convinv.k.conv.weight (glow.py:90-99) */
    const float* speaker_embed;/* [CTTS_N_SPEAKERS][speaker_embed_dim]  WN.k.speaker_embed.weight (glow.py:130-133);
                                  NULL when speaker_embed_dim == 0.  cond_w[0] is then [H][n_mel*G + speaker_embed_dim] */
} ctts_waveglow_flow_weights;

/* Bytes of the packed weight blob (device) for this config. */
size_t ctts_waveglow_packed_bytes(const ctts_waveglow_config* cfg);
/* Pack upsample.{weight [n_mel][n_mel][win], bias [n_mel]} (glow.py:238-241). */
int ctts_waveglow_pack_upsample(const ctts_waveglow_config* cfg, const float* up_w,
                                const float* up_b, void* packed, void* stream);
/* Pack one flow's weights into the MFMA-tile order the kernels stream. */
int ctts_waveglow_pack_flow(const ctts_waveglow_config* cfg, int32_t flow,
                            const ctts_waveglow_flow_weights* w, void* packed, void* stream);

/* ---- the hot path -------------------------------------------------------------------- */

size_t ctts_waveglow_workspace_bytes(const ctts_waveglow_config* cfg, int32_t batch,
                                     int32_t frames);

/* WaveGlow.infer (glow.py:314-350) with the noise made an explicit input.
 *   mel      [B][n_mel][F]      fp32 dense
 *   z_scaled [B][n_group][L]    fp32 dense, sigma already applied: the last n_remaining
 *                               rows are the initial latent (glow.py:326), the rows above
 *                               are the early-output noise in prepend order (glow.py:342-347)
 *   wave     [B][F*hop]         fp32 dense (glow.py:349)
 * workspace must be zero-filled once before its FIRST use with a given (batch, frames)
 * geometry (halo columns are never written afterwards). */
int ctts_waveglow_infer_f32(const ctts_waveglow_config* cfg, const void* packed,
                            const float* mel, const float* z_scaled, float* wave,
                            int32_t batch, int32_t frames, void* workspace,
                            size_t workspace_bytes, void* stream);
/* Multispeaker form (WN_config['speaker_embed_dim'] > 0, glow.py:131-133, 193-196): speaker_ids [B] int64 on the
 * device, each in [0, CTTS_N_SPEAKERS).  Every flow's embedding row is appended under the squeezed spectrogram as extra
 * K rows of that flow's cond layer 0.  speaker_ids == NULL is an error for a multispeaker config and ignored otherwise
 * (ctts_waveglow_infer_f32 is this call with NULL). */
int ctts_waveglow_infer_spk_f32(const ctts_waveglow_config* cfg, const void* packed, const float* mel,
                                const float* z_scaled, const int64_t* speaker_ids, float* wave, int32_t batch,
                                int32_t frames, void* workspace, size_t workspace_bytes, void* stream);

/* bf16 variant (BASELINE config 3): the WN in-layer and res/skip contractions run on bf16 MFMA with
 * fp32 accumulation and the WN activations (residual stream, gated activations, skip sum, conditioning
 * hidden) are stored bf16; upsampling, the two small cond layers, `end`, the coupling and the inverse
 * 1x1 conv stay fp32.  Needs BOTH blobs: `packed` (ctts_waveglow_pack_*: fp32 pieces + biases) and
 * `packed_bf16` (ctts_waveglow_pack_flow_bf16, same dense inputs).  I/O tensors are fp32 as above. */
size_t ctts_waveglow_packed_bf16_bytes(const ctts_waveglow_config* cfg);
int ctts_waveglow_pack_flow_bf16(const ctts_waveglow_config* cfg, int32_t flow,
                                 const ctts_waveglow_flow_weights* w, void* packed_bf16, void* stream);
size_t ctts_waveglow_workspace_bf16_bytes(const ctts_waveglow_config* cfg, int32_t batch,
                                          int32_t frames);
int ctts_waveglow_infer_bf16(const ctts_waveglow_config* cfg, const void* packed,
                             const void* packed_bf16, const float* mel, const float* z_scaled,
                             float* wave, int32_t batch, int32_t frames, void* workspace,
                             size_t workspace_bytes, void* stream);
int ctts_waveglow_infer_spk_bf16(const ctts_waveglow_config* cfg, const void* packed, const void* packed_bf16,
                                 const float* mel, const float* z_scaled, const int64_t* speaker_ids, float* wave,
                                 int32_t batch, int32_t frames, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* IEEE-half variant ("f16", ABI 6): the reference's own reduced-precision inference mode (glow.py:343, the notebooks' `.half()`
 * models).  Everything as in the bf16 variant - same layouts, same kernels, same blob and workspace SIZES
 * (ctts_waveglow_packed_bf16_bytes / ctts_waveglow_workspace_bf16_bytes) - but weights and WN activations are rounded to
 * IEEE half (11-bit significands instead of bf16's 8) and the products run on v_mfma_f32_32x32x16_f16, fp32 accumulation.
 * Range: a WN activation beyond 65 504 would become inf (the gated activations are in (-1, 1); the residual stream of the
 * models of this repository stays within a few units).  Same speed as bf16; waveform error against the fp32 reference about
 * 8x smaller (gated in tests/test_waveglow_gpu.py at the fp32 tolerance: RMS rel <= 1e-3). */
int ctts_waveglow_pack_flow_f16(const ctts_waveglow_config* cfg, int32_t flow,
                                const ctts_waveglow_flow_weights* w, void* packed_f16, void* stream);
int ctts_waveglow_infer_spk_f16(const ctts_waveglow_config* cfg, const void* packed, const void* packed_f16,
                                const float* mel, const float* z_scaled, const int64_t* speaker_ids, float* wave,
                                int32_t batch, int32_t frames, void* workspace, size_t workspace_bytes,
                                void* stream);

/* Split-bf16 variant ("bf16x3"): the same bf16 MFMA kernels with every GEMM operand carried as a hi + lo PAIR of bf16
 * planes (hi = bf16(v), lo = bf16(v - hi): 16 mantissa bits) and every contraction computed as the three products
 * hi*hi + lo*hi + hi*lo with fp32 accumulation (the lo*lo term, ~2^-16 of a product, is dropped).  Inputs therefore
 * carry a relative error of ~2^-17 instead of bf16's 2^-9, at a third of the bf16 MFMA rate - about three times the
 * fp32 MFMA rate of gfx950.  Same calling convention as the bf16 entry points; `packed_bf16x3` and the workspace have
 * their own sizes.  Waveform error against the fp32 reference goldens is stated in DESIGN.md and gated in the tests
 * at the fp32 tolerance (RMS rel <= 1e-3). */
size_t ctts_waveglow_packed_bf16x3_bytes(const ctts_waveglow_config* cfg);
int ctts_waveglow_pack_flow_bf16x3(const ctts_waveglow_config* cfg, int32_t flow,
                                   const ctts_waveglow_flow_weights* w, void* packed_bf16x3, void* stream);
size_t ctts_waveglow_workspace_bf16x3_bytes(const ctts_waveglow_config* cfg, int32_t batch,
                                            int32_t frames);
int ctts_waveglow_infer_spk_bf16x3(const ctts_waveglow_config* cfg, const void* packed, const void* packed_bf16x3,
                                   const float* mel, const float* z_scaled, const int64_t* speaker_ids, float* wave,
                                   int32_t batch, int32_t frames, void* workspace, size_t workspace_bytes,
                                   void* stream);

/* Stage entry points (same kernels, exposed for parity tests and profiling). */

/* upsample (ConvTranspose1d) + trim + squeeze: glow.py:318-324.  spect is padded layout
 * [B][n_mel*G][ld].  n_mel 80 / hop 256 / win 1024 / n_group 8 (the benchmark's): a W-stationary fp32 MFMA GEMM over the
 * fragment-ordered copy of the weights that ctts_waveglow_pack_upsample writes beside the plain one; other shapes: VALU kernels. */
int ctts_upsample_squeeze_f32(const ctts_waveglow_config* cfg, const void* packed,
                              const float* mel, float* spect, int32_t batch, int32_t frames,
                              void* stream);
/* One flow's WN stack: glow.py:188-222 up to (not including) `end`.  Reads audio rows
 * [ch_off, ch_off+n_half) of audio [B][n_group][L] dense and h_all (cond hidden for all
 * flows, from ctts_wn_cond_f32); leaves the skip sum in `out` (padded [B][C][ld]). */
int ctts_wn_cond_f32(const ctts_waveglow_config* cfg, const void* packed, const float* spect,
                     const float* spk_rows, float* h_tmp, float* h_all, int32_t batch, int32_t frames,
                     void* stream);   /* spk_rows [B][n_flows*S][ld] (S = speaker_embed_dim rounded up to 32) or NULL */
int ctts_wn_stack_f32(const ctts_waveglow_config* cfg, const void* packed, int32_t flow,
                      const float* audio, const float* h_all, float* x, float* act, float* out,
                      int32_t batch, int32_t frames, void* stream);
/* `end` 1x1 conv + affine coupling inverse + inverse 1x1 conv (+ un-squeeze on the last
 * flow): glow.py:222,337-340,349.  Updates audio in place; if `wave` is non-NULL the
 * mixed channels are written un-squeezed to wave [B][L*n_group] instead. */
int ctts_flow_tail_f32(const ctts_waveglow_config* cfg, const void* packed, int32_t flow,
                       const float* out, float* audio, float* wave, int32_t batch,
                       int32_t frames, void* stream);

/* ---- WaveFlow ("ax" core, waveflow=True): _4_mtw/waveglow/efficient_model_ax.py --------- */

/* Constructor arguments that shape the path (efficient_model_ax.py:19-169, glow_ax.py:427-543).
 * Built: PermuteHeight (folded into row addressing) or InvertibleConv1x1 mixing in both orders (mix_first), early
 * outputs, all fourteen gated units, res_skip=True, merge_res_skip, linear upsampling; in-layers dense (kh*kw <= 11 taps) or separable (depthwise kh x kw
 * + pointwise, glow_ax.py:525-531); conditioning either ONE k=1 linear WN cond layer on the mel, folded into the
 * in-layer GEMM (BASELINE config 4), or an arbitrary per-flow stack evaluated by the caller at frame rate and
 * handed over (cond_precomputed; SURVEY 8f.4: speaker embeddings, model-level and WN-level conv stacks with
 * activations - composed from ctts_conv1d_f32 / ctts_embed_rows_f32 / ctts_scale_add_rows_f32). */
/* channel mixing between flows (efficient_model_ax.py:139-165) */
#define CTTS_MIX_PERMUTE 0
#define CTTS_MIX_CONV1X1 1
/* WN_config['gated_unit'] names in the order of get_gate_func (glow_ax.py:168-198) */
#define CTTS_GATE_GTU 0
#define CTTS_GATE_GTRU 1
#define CTTS_GATE_GTLRU 2
#define CTTS_GATE_GLU 3
#define CTTS_GATE_TTU 4
#define CTTS_GATE_STU 5
#define CTTS_GATE_GTSU 6
#define CTTS_GATE_SPTU 7
#define CTTS_GATE_GSIU 8
#define CTTS_GATE_GSIRU 9
#define CTTS_GATE_GTSRU 10
#define CTTS_GATE_GSIRRU 11
#define CTTS_GATE_GSIRLRU 12
#define CTTS_GATE_GSIRRLRU 13
typedef struct ctts_waveflow_config {
    int32_t n_mel_channels;  /* 80 */
    int32_t n_flows;         /* 8 (even) */
    int32_t n_group;         /* 16: height of the squeezed audio */
    int32_t n_layers;        /* 8, width dilation 2^i */
    int32_t n_channels;      /* 64 (multiple of 64) */
    int32_t kernel_size_w;   /* 3 (odd) */
    int32_t kernel_size_h;   /* 3 */
    int32_t dilation_h;      /* height dilation of every layer unless dilation_h_l overrides it (>= 1; config 4: 1) */
    int32_t seperable_conv;  /* 0 | 1  (WN_config['seperable_conv'], the reference's spelling) */
    int32_t cond_precomputed;/* 0: mel + folded cond layer;  1: ctts_waveflow_inverse_cond_f32 */
    int32_t gated_unit;      /* CTTS_GATE_*: WN_config['gated_unit'] (glow_ax.py:168-198); 0 = 'GTU' */
    int32_t merge_res_skip;  /* 0 | 1: WN_config['merge_res_skip'] (glow_ax.py:612-626: every res_skip layer has C rows,
                                all of them skip; the layer input stays the `start` output) */
    int32_t n_early_every;   /* early outputs (ax:170-176, 340-341); 0 or > n_flows: none */
    int32_t n_early_size;    /* rows leaving the latent at every n_early_every-th flow */
    int32_t mixing;          /* CTTS_MIX_PERMUTE (PermuteHeight, folded into row addressing) | CTTS_MIX_CONV1X1 */
    int32_t mix_first;       /* 0 | 1: un-mix after / before the coupling inverse (ax:324-325, 337-338) */
    int32_t dilation_w[12];  /* WN_config['n_layers_dilations_w'] per layer (glow_ax.py:507-509); 0 = the default 2^i */
    int32_t dilation_h_l[12];/* WN_config['n_layers_dilations_h'] per layer (glow_ax.py:510-512); 0 = `dilation_h` */
    int32_t f32_gemm_mode;   /* CTTS_GEMM_*: main loop of this model's GEMMs (dense and fused separable layers) */
} ctts_waveflow_config;

/* Dense, weight-norm-folded fp32 weights of one flow in checkpoint layouts
 * (keys WN.k.WN.*): */
typedef struct ctts_waveflow_flow_weights {
    const float* start_w;      /* [C]                 WN.start (Conv2d 1->C, 1x1) */
    const float* start_b;      /* [C] */
    const float* cond_w;       /* [2*C*n_layers][n_mel]   WN.cond_layers.0 (k=1); NULL if cond_precomputed */
    const float* cond_b;       /* [2*C*n_layers] */
    const float* const* in_w;  /* n_layers x [2C][C][kh][kw]; separable: the pointwise [2C][C] (in_layers.i.1) */
    const float* const* in_b;  /* n_layers x [2C] */
    const float* const* rs_w;  /* n_layers x [2C or C][C] */
    const float* const* rs_b;
    const float* end_w;        /* [2][C] (row 0 = log_s, row 1 = t; efficient_modules.py:61) */
    const float* end_b;        /* [2] */
    const float* const* dw_w;  /* separable only: n_layers x [C][kh][kw]  depthwise (in_layers.i.0), else NULL */
    const float* const* dw_b;  /* n_layers x [C] */
    const float* w_inverse;    /* [n_rem_k][n_rem_k] fp32 inverse of convinv.k.weight (CTTS_MIX_CONV1X1), else NULL */
} ctts_waveflow_flow_weights;

size_t ctts_waveflow_packed_bytes(const ctts_waveflow_config* cfg);
int ctts_waveflow_pack_flow(const ctts_waveflow_config* cfg, int32_t flow,
                            const ctts_waveflow_flow_weights* w, void* packed, void* stream);
size_t ctts_waveflow_workspace_bytes(const ctts_waveflow_config* cfg, int32_t batch,
                                     int32_t samples);
/* WaveGlow.inverse(z, cond) of the ax core (efficient_model_ax.py:279-357) for waveflow=True:
 *   z    [B][T] fp32, sigma applied (T multiple of n_group)      mel [B][n_mel][frames] (as passed
 *   to inverse(), i.e. already padded by infer())                audio [B][T]
 * Includes the per-flow NaN -> 0 (ax:333-334).  Workspace zero-filled once before first use.
 *
 * How a row of the recurrence is launched (C = 64 models; results of the forms agree bit for bit within one tile shape, see
 * "Which loop a launch really runs" below): one launch per fused layer (small sizes: a layer is one wave's serial chain),
 * or - from 200 column tiles of 128 per layer on (batch 2 at 900 frames) up to 1250 - the ROW QUEUE: ALL rows of a flow (each
 * row = its n_layers fused layers + a tail stage: end conv, affine update of the next latent row, the next row's start conv) as
 * ONE launch whose workgroups take (row, stage, tile) items from an atomic counter in order and wait, per item, for the flags
 * of the neighbouring tiles of the previous stage only.  Items are claimed in order, so the oldest unfinished item can
 * always run: the launch cannot deadlock and needs no co-residency.  Its wait is bounded all the same (two periods of 1 s in a
 * row during which no workgroup of the launch claimed an item - time without progress, not wall time); if it ever expires
 * every workgroup leaves, the call fills `audio` with NaN instead of returning plausible noise (stream-ordered, so THIS
 * call's status code cannot report it) and a sticky status word in the workspace is set: the calling thread's NEXT
 * ctts_waveflow_inverse_* on that workspace - whichever workspaces and streams it used in between: the thread keeps an
 * event per not-yet-checked workspace (16 of them; beyond that the oldest is checked early and reported by the call that
 * displaced it) - waits for that event first, returns CTTS_E_ABORT with ctts_last_error() text and clears the word (the call
 * after that runs normally); ctts_waveflow_abort_status asks at once (a workspace about to be freed: ask before).  NaN is also a legal
 * output of this path (ignore_nan, ax:333-334) - the status, not the NaN, is what says "aborted".
 * CTTS_WF_NO_ROW_QUEUE = always one launch per layer.  The queue's control words and layer
 * descriptors live in the caller's workspace (included in ctts_waveflow_workspace_bytes); the descriptors reach it through a
 * pinned staging buffer of the calling host thread, which the thread's NEXT call rewrites only after an event behind this
 * call's last copy - so a call with the row queue may wait on the host for the thread's previous call and must not be recorded
 * into a HIP graph (use CTTS_WF_NO_ROW_QUEUE there). */
int ctts_waveflow_inverse_f32(const ctts_waveflow_config* cfg, const void* packed, const float* z,
                              const float* mel, float* audio, int32_t batch, int32_t samples,
                              int32_t frames, void* workspace, size_t workspace_bytes,
                              void* stream);

/* Synchronises `stream` and reports (and clears) the workspace's row-queue status: CTTS_OK, or CTTS_E_ABORT if a
 * ctts_waveflow_inverse_* call on this workspace (same cfg / batch / samples, which fix its layout) aborted since the last
 * report.  The Python WaveFlow.inverse / infer call it whenever they synchronise anyway and raise. */
int ctts_waveflow_abort_status(const ctts_waveflow_config* cfg, int32_t batch, int32_t samples, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Same, for cond_precomputed models: cond [n_flows][B][2*C*n_layers][cond_ld] fp32 = the output of each flow's
 * WN conditioning stack at FRAME rate (glow_ax.py:566-577), valid frames at columns [cond_pad, cond_pad+frames);
 * upsampling (gax:545-554) and the per-layer slicing (gax:580-592) happen inside. */
int ctts_waveflow_inverse_cond_f32(const ctts_waveflow_config* cfg, const void* packed, const float* z,
                                   const float* cond, int32_t cond_ld, int32_t cond_pad, float* audio,
                                   int32_t batch, int32_t samples, int32_t frames, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* Small operators the conditioning stacks and the output stage are composed from (padded row layout as for
 * ctts_conv1d_f32: x [B][C][ld], valid columns [pad, pad+T)):
 *   embed_rows:     x[b][row0 + e][pad .. pad+T) = table[ids[b]][e]          (speaker embedding concat, ax:286-291)
 *   scale_add_rows: y = alpha * x + r   (r may be NULL; alpha read from the device: the rezero parameter, ax:299-307)
 *   deemphasis:     y[n] = x[n] + p * y[n-1] per utterance, fp64 recurrence like scipy.signal.lfilter (ax:351-355);
 *                   in place (y == x) allowed
 *   vol_unscale:    x > 0 -> 10^log2(x), x < 0 -> -(10^log2(-x)) in place over n floats (preceived_vol_scaling, ax:342-344)
 *   affine_rows:    x = (x + shift) * scale on rows [0, rows), valid columns only (shift_spect / scale_spect, ax:206-209)
 *   resample_rows:  F.interpolate along time, padded rows in and out: mode 0 'linear' align_corners=True to T_out
 *                   (ax:174, glow_ax.py:365), mode 1 'linear' align_corners=False and mode 2 'nearest' with the given
 *                   scale_factor (TransposedUpsampleNet residual, glow_ax.py:229-231); scale_factor <= 0: T_in / T_out
 *   interleave_phases: ConvTranspose1d(stride, padding) (glow_ax.py:222) = `stride` stride-1 convolutions over the
 *                   input positions, one per output residue r = (n + padding) mod stride, each run with
 *                   ctts_conv1d_f32 into phases[r] ([stride][B][C][ld_in]); this call writes
 *                   y[n] = phases[(n + padding) % stride][(n + padding) / stride] for n in [0, T_out) */
int ctts_vol_unscale_f32(float* x, int64_t n, void* stream);
int ctts_affine_rows_f32(float* x, int32_t batch, int32_t C, int32_t rows, int32_t T, int32_t ld, int32_t pad, float shift,
                         float scale, void* stream);
int ctts_resample_rows_f32(const float* x, float* y, int32_t batch, int32_t C, int32_t T_in, int32_t ld_in, int32_t pad_in,
                           int32_t T_out, int32_t ld_out, int32_t pad_out, int32_t mode, float scale_factor, void* stream);
int ctts_interleave_phases_f32(const float* phases, float* y, int32_t batch, int32_t C, int32_t stride, int32_t padding,
                               int32_t T_in, int32_t ld_in, int32_t pad_in, int32_t T_out, int32_t ld_out, int32_t pad_out,
                               void* stream);
int ctts_embed_rows_f32(const float* table, const int64_t* ids, float* x, int32_t row0, int32_t embed_dim,
                        int32_t batch, int32_t C, int32_t T, int32_t ld, int32_t pad, void* stream);
int ctts_scale_add_rows_f32(const float* x, const float* alpha_dev, const float* r, float* y, int32_t batch,
                            int32_t C, int32_t T, int32_t ld, int32_t pad, void* stream);
int ctts_deemphasis_f32(const float* x, float* y, int32_t batch, int32_t T, double p, void* stream);

/* ---- "ax" WaveGlow core with waveflow=False: AffineCouplingBlock + 1-D WN ------------------- */
/* Replaces, for inference, efficient_model_ax.py:309-346 (the flow loop of WaveGlow.inverse) with
 *   AffineCouplingBlock.inverse   efficient_modules.py:94-105   ((log_s, t) = WN(a0); a1 = (a1 - t) / exp(log_s))
 *   WN.forward (1-D)              glow_ax.py:375-418            (start, dilated in_layers + GTU gate, res/skip, end;
 *                                                               conditioning at frame rate, linearly interpolated
 *                                                               with align_corners=True, glow_ax.py:362-373)
 *   InvertibleConv1x1.inverse     efficient_modules.py:269-286  (W.float().inverse() taken by the caller)
 *   PermuteHeight.inverse         efficient_modules.py:376-403
 *   early outputs                 efficient_model_ax.py:312-316, 340-341;   ignore_nan  :13-16, 333-334
 *   mix_first ordering            efficient_model_ax.py:324-325, 337-338
 * Built: all fourteen gated units, res_skip=True, merge_res_skip, dense in-layers with width dilation 2^i,
 * the per-flow WN conditioning stack is evaluated by the caller (composed from ctts_conv1d_f32 / ctts_embed_rows_f32 /
 * ctts_scale_add_rows_f32 / ctts_replicate_halo_f32) and handed over, like ctts_waveflow_inverse_cond_f32: at FRAME
 * rate (upsample_first=False; `frames` columns, interpolated to the latent's rate inside the gate epilogue) or already
 * at the latent's rate (upsample_first=True, efficient_model_ax.py:116-126, 174-186: model-level TransposedUpsampleNet
 * via ctts_interleave_phases_f32 + ctts_resample_rows_f32; `frames` == samples / n_group, read as it is). */
typedef struct ctts_wgax_config {
    int32_t n_flows;         /* 48 in the reference's timed notebook config */
    int32_t n_group;         /* 24 (even, <= 32) */
    int32_t n_early_every;   /* 16 */
    int32_t n_early_size;    /* 2 (even) */
    int32_t n_layers;        /* 8 */
    int32_t n_channels;      /* 256 (multiple of 128) */
    int32_t kernel_size;     /* 3 (odd, <= 11): WN_config kernel_size_w or kernel_size */
    int32_t mixing;          /* CTTS_MIX_PERMUTE | CTTS_MIX_CONV1X1 */
    int32_t mix_first;       /* 0 | 1 */
    int32_t ignore_nan;      /* 1: NaN -> 0 on the latent after every coupling (the reference's default) */
    int32_t gated_unit;      /* CTTS_GATE_* (0 = 'GTU') */
    int32_t merge_res_skip;  /* 0 | 1 (glow_ax.py:401-416) */
    int32_t dilation_w[12];  /* WN_config['n_layers_dilations_w'] per layer (glow_ax.py:331-333); 0 = the default 2^i */
    int32_t f32_gemm_mode;   /* CTTS_GEMM_*: main loop of this model's GEMMs */
} ctts_wgax_config;

/* Dense, weight-norm-folded fp32 device weights of one flow in checkpoint layouts (keys WN.k.WN.*, convinv.k.weight) */
typedef struct ctts_wgax_flow_weights {
    const float* start_w;      /* [C][n_half_k] */
    const float* start_b;      /* [C] */
    const float* const* in_w;  /* n_layers x [2C][C][ks] */
    const float* const* in_b;  /* n_layers x [2C] */
    const float* const* rs_w;  /* n_layers x [2C or C][C] */
    const float* const* rs_b;
    const float* end_w;        /* [2*n_half_k][C]: rows [0, h) = log_s, [h, 2h) = t (efficient_modules.py:100) */
    const float* end_b;        /* [2*n_half_k] */
    const float* w_inverse;    /* [n_rem_k][n_rem_k] = convinv.k.weight[:, :, 0].float().inverse(); NULL for PERMUTE */
} ctts_wgax_flow_weights;

size_t ctts_wgax_packed_bytes(const ctts_wgax_config* cfg);
int ctts_wgax_pack_flow(const ctts_wgax_config* cfg, int32_t flow, const ctts_wgax_flow_weights* w,
                        void* packed, void* stream);
size_t ctts_wgax_workspace_bytes(const ctts_wgax_config* cfg, int32_t batch, int64_t samples);
/* The flow loop of WaveGlow.inverse(z, cond):
 *   z     [B][samples] fp32, sigma applied; samples a multiple of n_group; early-output noise is part of z (ax:312-316)
 *   cond  [n_flows][B][2*C*n_layers][cond_ld]: each flow's WN conditioning at frame rate, valid frames at columns
 *         [cond_pad, cond_pad + frames)
 *   audio [B][samples]
 * Workspace zero-filled once before first use (halo columns are never written). */
int ctts_wgax_inverse_f32(const ctts_wgax_config* cfg, const void* packed, const float* z, const float* cond,
                          int32_t cond_ld, int32_t cond_pad, int32_t frames, float* audio, int32_t batch,
                          int64_t samples, void* workspace, size_t workspace_bytes, void* stream);
/* padding_mode='replicate' of the conditioning convs (efficient_model_ax.py:90, glow_ax.py:311): fill the `halo`
 * columns either side of the valid range of x [B][C][ld] with the edge values, before a ctts_conv1d_f32 reads them. */
int ctts_replicate_halo_f32(float* x, int32_t batch, int32_t C, int32_t T, int32_t ld, int32_t pad, int32_t halo,
                            void* stream);

/* ---- Tacotron2-TM decoder loop: _2_ttm/tacotron2_tm/model.py:668-767, 851-916 -------------- */

/* Shapes from hparams.py (:201-258).  Built topology = the repo defaults: attention_type 0 with
 * windowed attention, AttRNN_extra_decoder_input=True, two decoder LSTMs with residual, 2-layer prenet. */
typedef struct ctts_taco_decoder_config {
    int32_t n_mel_channels;        /* 80 */
    int32_t memory_in_dim;         /* 1313 = encoder 1024 + speaker 256 + sylzu 1 + torchMoji 32 */
    int32_t memory_dim;            /* 512  memory_bottleneck_dim */
    int32_t attention_dim;         /* 192 */
    int32_t attention_rnn_dim;     /* 1280 */
    int32_t decoder_rnn_dim;       /* 768 */
    int32_t second_decoder_rnn_dim;/* 768 (== decoder_rnn_dim: residual) */
    int32_t prenet_dim;            /* 256 */
    int32_t location_n_filters;    /* 32 */
    int32_t location_kernel_size;  /* 31 */
    int32_t window_range;          /* 16 */
} ctts_taco_decoder_config;

typedef struct ctts_lstm_weights {   /* torch LSTMCell layout, gate order i,f,g,o (layers.py:308-372) */
    const float* w_ih;  /* [4H][I] */
    const float* w_hh;  /* [4H][H] */
    const float* b_ih;  /* [4H] */
    const float* b_hh;  /* [4H] */
} ctts_lstm_weights;

/* Dense fp32 device pointers in checkpoint layouts (state_dict keys decoder.*). */
typedef struct ctts_taco_decoder_weights {
    const float* bottleneck_w;    /* [memory_dim][memory_in_dim]   memory_bottleneck.bottleneck (no bias) */
    const float* memory_layer_w;  /* [A][memory_dim]               attention_layer.memory_layer */
    const float* query_w;         /* [A][Ra]                       attention_layer.query_layer */
    const float* v_w;             /* [A]                           attention_layer.v */
    const float* loc_conv_w;      /* [F][2][K]                     location_layer.location_conv */
    const float* loc_dense_w;     /* [A][F]                        location_layer.location_dense */
    const float* prenet_w1;       /* [P][n_mel]                    prenet.layers.0 (no bias) */
    const float* prenet_w2;       /* [P][P]                        prenet.layers.1 */
    ctts_lstm_weights att_rnn;    /* I = P + memory_dim + Rd, H = Ra */
    ctts_lstm_weights dec_rnn;    /* I = Ra + memory_dim,     H = Rd */
    ctts_lstm_weights dec2_rnn;   /* I = Rd,                  H = Rd2 */
    const float* proj_w;          /* [n_mel][Rd2 + memory_dim]     linear_projection */
    const float* proj_b;          /* [n_mel] */
    const float* gate_w;          /* [Rd2 + memory_dim]            gate_layer */
    const float* gate_b;          /* [1] */
    float windowed_att_pos_offset;/* attention_layer.windowed_att_pos_offset (learned scalar) */
    float exp_smoothing_factor;   /* decoder.exp_smoothing_factor (raw; sigmoid applied inside) */
} ctts_taco_decoder_weights;

size_t ctts_taco_decoder_packed_bytes(const ctts_taco_decoder_config* cfg);
int ctts_taco_decoder_pack(const ctts_taco_decoder_config* cfg, const ctts_taco_decoder_weights* w,
                           void* packed, void* stream);
/* Largest batch one workspace / one ctts_taco_decoder_steps_f32 call takes for this shape: 256 where the batched MFMA form is
 * built (every cell's K = I + H a multiple of 64, every state width a multiple of 16, the windowed attention's limits: window
 * <= +-16, memory <= 512, attention dim <= 256, <= 32 location filters of <= 31 taps - the repo defaults and every checkpoint
 * shape under tests/golden), else 4; 0 for an invalid config.  The reference's server decodes up to
 * simultaneous_texts x batch_size_per_text = 256 rows per call (_5_infer/t2s_server/text2speech.py:418-424, 537, 554). */
int32_t ctts_taco_decoder_max_batch(const ctts_taco_decoder_config* cfg);
size_t ctts_taco_decoder_workspace_bytes(const ctts_taco_decoder_config* cfg, int32_t batch,
                                         int32_t text_len);
/* Decoder.inference prologue (model.py:866-877): memory bottleneck, processed_memory, zero states.
 *   memory_in [B][text_len][memory_in_dim],  lengths [B] int32 (device). */
int ctts_taco_decoder_init_f32(const ctts_taco_decoder_config* cfg, const void* packed,
                               const float* memory_in, const int32_t* lengths, int32_t batch,
                               int32_t text_len, void* workspace, size_t workspace_bytes,
                               void* stream);
/* Run decoder steps [step0, step0 + n_steps) (model.py:879-883 body = prenet + decode()).
 *   keep_masks [max_steps][2][B][P] uint8: the prenet's always-on dropout keep-masks (model.py:189-190)
 *   mel_out [B][n_mel][max_steps], gate_out [B][max_steps] (logits), align_out [B][max_steps][text_len]
 * The stop rule (model.py:898-904) is evaluated by the caller on gate_out between calls.
 * Where ctts_taco_decoder_max_batch(cfg) is 256 every batch takes the batched form: seven launches per step with the cells /
 * query / projection / prenet as MFMA GEMMs over the whole batch (v_mfma_f32_16x16x4_f32, weights streamed once per step
 * whatever the batch: csrc/tacotron_batched.h).  Other shapes (batch <= 4 only), or CTTS_TACO_VALU: six launches per step on
 * the VALU.  Same state layout in the workspace: the forms (and the persistent one below) can be mixed call by call. */
int ctts_taco_decoder_steps_f32(const ctts_taco_decoder_config* cfg, const void* packed,
                                const uint8_t* keep_masks, float* mel_out, float* gate_out,
                                float* align_out, int32_t batch, int32_t text_len, int32_t step0,
                                int32_t n_steps, int32_t max_steps, void* workspace, void* stream);
/* The same, also recording what Decoder.inference(return_hidden_state=True) returns (model.py:762, 888-889):
 *   hidden_out [B][second_decoder_rnn_dim + memory_dim][max_steps] = [dec_h + d2_h | attention context] per step (NULL = off).
 * (The persistent form has no such output: a caller that wants the hidden states uses this entry point.) */
int ctts_taco_decoder_steps_hidden_f32(const ctts_taco_decoder_config* cfg, const void* packed,
                                       const uint8_t* keep_masks, float* mel_out, float* gate_out,
                                       float* align_out, float* hidden_out, int32_t batch, int32_t text_len,
                                       int32_t step0, int32_t n_steps, int32_t max_steps, void* workspace,
                                       void* stream);

/* Persistent form of ctts_taco_decoder_steps_f32: ONE launch of 256 resident workgroups (256 threads each, one wave per
 * SIMD) runs all n_steps steps with EVERY LSTM weight resident on the compute units for the whole launch (registers + LDS;
 * the 108 MB of weights are read once, at entry: ask for blocks of >= 32 steps), products on v_mfma_f32_4x4x1, and vectors
 * move between workgroups as granules that carry their own flag (4-byte self-flagging values, 8-byte {tag, value} for the query; no grid barrier, no per-step launches).  Same state (workspace), same outputs, same keep_masks contract as ctts_taco_decoder_steps_f32, so the
 * two can be mixed call by call.  Built for the repo-default decoder shape (attention RNN 1280, decoder RNNs 768,
 * prenet 256, memory 512, window 16), batch <= 4, text_len <= 1024 on a device with >= 256 CUs:
 * ctts_taco_decoder_persistent_bytes returns 0 otherwise, also when the CURRENT device has fewer CUs (use the per-launch
 * form); with no device at all it answers for the shape alone.
 *   exchange: ctts_taco_decoder_persistent_bytes(...) device bytes, zero-filled ONCE by the caller; the call re-initialises the
 *   granule area (everything but the control words) itself before every launch.  The LAST 64 bytes (bytes - 64 .. bytes) are control words (uint32): word 0 stays 0 on success;
 *   non-zero = a bounded wait gave up (words 1..3: workgroup, phase, step), the outputs of that call are invalid and
 *   every later launch on the same buffer returns immediately (sticky) until the caller zeroes the words. */
size_t ctts_taco_decoder_persistent_bytes(const ctts_taco_decoder_config* cfg, int32_t batch, int32_t text_len);
int ctts_taco_decoder_steps_persistent_f32(const ctts_taco_decoder_config* cfg, const void* packed,
                                           const uint8_t* keep_masks, float* mel_out, float* gate_out,
                                           float* align_out, int32_t batch, int32_t text_len, int32_t step0,
                                           int32_t n_steps, int32_t max_steps, void* workspace, void* exchange,
                                           size_t exchange_bytes, void* stream);

/* Profiling aid for the persistent decoder: `stamps` = device buffer of 256 x 64 x 24 uint64 receiving s_memrealtime
 * (100 MHz) at the phase boundaries (slots 0..12) and at the publish instants (13..17) of the first 64 steps of every
 * later launch; NULL switches it off (scripts/profile_persistent.py). */
int ctts_taco_decoder_persistent_debug(void* stamps);

/* ---- Tacotron2-TM one-shot stages: operator-level primitives ------------------------------- */
/* The encoder (model.py:283-316) and postnet (:218-228) are stacks of "same"-padded Conv1d (+ eval-mode
 * BatchNorm1d, folded into the weights at pack time) + LeakyReLU / tanh, a packed-sequence BiLSTM, and a
 * handful of per-utterance vectors.  They are exposed as three primitives the Python host composes in the
 * reference's own order; all arithmetic is in the library.
 * Padded layout here: x [B][C][ld], valid columns [pad, pad+T), zeros elsewhere, ld % 4 == 0,
 * ld >= roundup(T,128) + 2*pad, pad >= kernel_size/2. */
typedef struct ctts_conv1d_desc {
    int32_t c_in;         /* multiple of 16 */
    int32_t c_out;
    int32_t kernel_size;  /* odd, <= 11 */
    int32_t act;          /* 0 none, 1 LeakyReLU(slope), 2 tanh */
    float slope;
    int32_t f32_gemm_mode; /* CTTS_GEMM_*: main loop of this operator's GEMM */
} ctts_conv1d_desc;
size_t ctts_conv1d_packed_bytes(const ctts_conv1d_desc* d);
/* w [c_out][c_in][k], b [c_out]; bn_* [c_out] or all NULL (eval BatchNorm1d folded: w*s, (b-mean)*s+beta). */
int ctts_conv1d_pack_f32(const ctts_conv1d_desc* d, const float* w, const float* b, const float* bn_gamma,
                         const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                         void* packed, void* stream);
/* y = act(conv1d(x)) (accumulate == 0) or y += conv1d(x) (accumulate != 0, act must be 0). */
int ctts_conv1d_f32(const ctts_conv1d_desc* d, const void* packed, const float* x, float* y,
                    int32_t accumulate, int32_t batch, int32_t T, int32_t ld, int32_t pad, void* stream);

/* Packed-sequence LSTM, one direction (nn.LSTM + pack_padded_sequence semantics, model.py:299-309):
 * item b runs len[b] steps (forward: t = 0..len-1, reverse: t = len-1..0); outputs beyond len are untouched. */
size_t ctts_lstm_seq_packed_bytes(int32_t input_size, int32_t hidden_size);
int ctts_lstm_seq_pack_f32(const ctts_lstm_weights* w, int32_t input_size, int32_t hidden_size, void* packed,
                           void* stream);
size_t ctts_lstm_seq_workspace_bytes(int32_t hidden_size, int32_t batch, int32_t ld);
/*   x [B][I][ld] padded; out[b][t][out_col + u] (row stride out_tstride, batch stride out_bstride) = h_t;
 *   hn[b][hn_col + u] (row stride hn_stride) = final hidden state; lengths int32 device.
 *   batch <= 4: one VALU launch per time step; hidden_size % 64 == 0: batch <= 256, the recurrent product of a time step is an
 *   MFMA GEMM over the batch (csrc/tacotron_batched.h, BG_EPI_SEQ) - ctts_lstm_seq_workspace_bytes answers 0 beyond that. */
int ctts_lstm_seq_f32(const void* packed, const float* x, const int32_t* lengths, int32_t reverse, float* out,
                      int64_t out_bstride, int32_t out_tstride, int32_t out_col, float* hn, int32_t hn_stride,
                      int32_t hn_col, int32_t batch, int32_t T, int32_t input_size, int32_t hidden_size, int32_t ld,
                      int32_t pad, void* workspace, size_t workspace_bytes, void* stream);
/* Both directions of a bidirectional layer (nn.LSTM(bidirectional=True), model.py:299-309) in lockstep: ONE launch per time
 * step covers the forward step t and the reverse step of every item, so the layer costs T dependent launches, not 2 T.
 * Same arithmetic and outputs as two ctts_lstm_seq_f32 calls (reverse = 0 into out_col_fwd / hn_col_fwd, reverse = 1 into
 * out_col_bwd / hn_col_bwd); each direction needs its own workspace of ctts_lstm_seq_workspace_bytes. */
int ctts_lstm_biseq_f32(const void* packed_fwd, const void* packed_bwd, const float* x, const int32_t* lengths, float* out,
                        int64_t out_bstride, int32_t out_tstride, int32_t out_col_fwd, int32_t out_col_bwd, float* hn,
                        int32_t hn_stride, int32_t hn_col_fwd, int32_t hn_col_bwd, int32_t batch, int32_t T, int32_t input_size,
                        int32_t hidden_size, int32_t ld, int32_t pad, void* workspace_fwd, void* workspace_bwd,
                        size_t workspace_bytes, void* stream);

/* x0[b][c][pad+t] = c < E ? embedding[text[b][t]][c] : spk_table[speaker[b]][c - E]   (model.py:1049, 284-288) */
int ctts_taco_embed_f32(const float* embedding, const float* spk_table, const int64_t* text,
                        const int64_t* speakers, float* x0, int32_t batch, int32_t T, int32_t E, int32_t S,
                        int32_t ld, int32_t pad, void* stream);
/* Per-utterance memory columns (model.py:1052-1066 + SylpsNet.infer_auto + tm_bn/tm_linear):
 * pred_sylps[b] = sylps_w . hn[b] + sylps_b;  memory_in[b][t][enc_dim:] = [speaker embed | sylzu | torchMoji crushed]. */
typedef struct ctts_taco_memory_weights {
    const float* sylps_w; const float* sylps_b;              /* encoder.sylps_layer [1][enc_dim], [1] */
    const float* speaker_embedding;                          /* [n_speakers][spk_dim] */
    const float* syl_w0; const float* syl_b0;                /* sylps_net.seq_layers.0 [hid][2], [hid] */
    const float* syl_w2; const float* syl_b2;                /* sylps_net.seq_layers.2 [1][hid], [1] */
    const float* syl_res_weight;                             /* [1] */
    const float* tm_gamma; const float* tm_beta; const float* tm_mean; const float* tm_var;  /* tm_bn (or NULL) */
    const float* tm_w; const float* tm_b;                    /* tm_linear [crushed][tm_dim], [crushed] */
} ctts_taco_memory_weights;
int ctts_taco_memory_f32(const ctts_taco_memory_weights* w, const float* hn, const int64_t* speakers,
                         const float* torchmoji, float* memory_in, float* pred_sylps, int32_t batch, int32_t T,
                         int32_t enc_dim, int32_t spk_dim, int32_t syl_hidden, int32_t tm_dim, int32_t tm_crushed,
                         void* stream);
/* The same with the reference's ``gt_sylps`` override (model.py:1044, 1058 "gt_sylps or pred_sylps"): gt_sylps [B] device floats
 * are what SylpsNet.infer_auto reads (NULL = the predicted value, i.e. ctts_taco_memory_f32); pred_sylps is written either way. */
int ctts_taco_memory_sylps_f32(const ctts_taco_memory_weights* w, const float* hn, const int64_t* speakers,
                               const float* torchmoji, const float* gt_sylps, float* memory_in, float* pred_sylps,
                               int32_t batch, int32_t T, int32_t enc_dim, int32_t spk_dim, int32_t syl_hidden,
                               int32_t tm_dim, int32_t tm_crushed, void* stream);
/* dense [B][C][src_ld] (first T columns) <-> padded [B][C][ld] copies */
int ctts_pad_rows_f32(const float* src, int64_t src_bstride, int32_t src_ld, float* dst, int32_t batch,
                      int32_t C, int32_t T, int32_t ld, int32_t pad, void* stream);
int ctts_unpad_rows_f32(const float* src, float* dst, int64_t dst_bstride, int32_t dst_ld, int32_t batch,
                        int32_t C, int32_t T, int32_t ld, int32_t pad, void* stream);

/* ---- STFT / mel frontend (utils/audio/stft.py) ------------------------------------------- */

/* STFT.__init__ (stft.py:46-77) / TacotronSTFT.__init__ (:155-166) arguments that shape the path. */
typedef struct ctts_stft_config {
    int32_t filter_length;   /* 1024 (multiple of 16) */
    int32_t hop_length;      /* 256 */
    int32_t win_length;      /* 1024 (informational: the window is already folded into the basis) */
    int32_t n_mel_channels;  /* 80; 0 = magnitude only */
    float clamp_val;         /* 1e-5: dynamic_range_compression clip (audio_processing.py:78-84) */
} ctts_stft_config;

size_t ctts_stft_packed_bytes(const ctts_stft_config* cfg);
/* forward_basis [2*(N/2+1)][N] = the module buffer of stft.py:60-76 ([Re;Im] DFT rows x window),
 * mel_basis [n_mel][N/2+1] = the buffer of stft.py:163-166 (may be NULL when n_mel_channels == 0). */
int ctts_stft_pack(const ctts_stft_config* cfg, const float* forward_basis, const float* mel_basis,
                   void* packed, void* stream);
size_t ctts_stft_workspace_bytes(const ctts_stft_config* cfg, int32_t batch, int32_t samples);
/* STFT.transform_jit magnitude (stft.py:79-111) and TacotronSTFT.mel_spectrogram (:180-207):
 *   y   [B][T] fp32 in [-1, 1];   frames = T / hop + 1
 *   mag [B][N/2+1][frames]  (may be NULL)      mel [B][n_mel][frames] = log(max(mel_basis @ mag, clamp))
 * workspace zero-filled once before first use (as for WaveGlow). */
int ctts_stft_mel_f32(const ctts_stft_config* cfg, const void* packed, const float* y, float* mag,
                      float* mel, int32_t batch, int32_t samples, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Phase / inverse path (used by the Denoiser, _4_mtw/waveglow/denoiser.py:55-72):
 *   inverse_basis [2*(N/2+1)][N] = the module buffer of stft.py:62-63,74 (pinv of the scaled DFT basis x window),
 *   window_sq [N] = the zero-centre-padded squared window of audio_processing.py:47-50. */
int ctts_stft_pack_inverse(const ctts_stft_config* cfg, const float* inverse_basis, const float* window_sq,
                           void* packed, void* stream);
/* STFT.transform_jit with phase (stft.py:99-111): mag, phase [B][N/2+1][frames] (phase may be NULL). */
int ctts_stft_transform_f32(const ctts_stft_config* cfg, const void* packed, const float* y, float* mag,
                            float* phase, int32_t batch, int32_t samples, void* workspace,
                            size_t workspace_bytes, void* stream);
/* STFT.inverse (stft.py:117-146): conv_transpose1d with the inverse basis, window-sum-square normalisation,
 * N/hop scaling, crop N/2 each side -> out [B][(frames-1)*hop].  If bias_spec [N/2+1] is non-NULL the magnitude
 * is first replaced by max(mag - bias_spec*strength, 0) (Denoiser.forward, denoiser.py:62-67).
 * workspace: ctts_stft_workspace_bytes(cfg, batch, (frames-1)*hop). */
int ctts_stft_inverse_f32(const ctts_stft_config* cfg, const void* packed, const float* mag, const float* phase,
                          const float* bias_spec, float strength, float* out, int32_t batch, int32_t frames,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Same with one bias spectrum PER UTTERANCE: bias_spec [batch][bias_bstride >= N/2+1] (Denoiser(speaker_dependant=True):
 * bias_spec[speaker_ids], denoiser.py:29-45, 65-66); bias_bstride = 0 shares one spectrum like ctts_stft_inverse_f32. */
int ctts_stft_inverse_bias_f32(const ctts_stft_config* cfg, const void* packed, const float* mag, const float* phase,
                               const float* bias_spec, int32_t bias_bstride, float strength, float* out,
                               int32_t batch, int32_t frames, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- attention-alignment scoring (T2S retry loop) -------------------- */

/* utils/model/utils.py:59-120 `alignment_metric(alignments, input_lengths, output_lengths, enc_min_thresh)`:
 * alignments [batch][dec][enc] fp32 (device); lengths are device fp32 [batch] or NULL (reference defaults:
 * enc-1 / dec-1).  out: device double [batch][6] = diagonalitys, avg_prob, encoder_max_focus,
 * encoder_min_focus, encoder_avg_focus, p_missing_enc.  The input is not modified (the reference zeroes the
 * padded decoder steps of its argument in place, :81). */
size_t ctts_alignment_workspace_bytes(int32_t batch, int32_t dec, int32_t enc);
int ctts_alignment_metric_f32(const float* alignments, const float* input_lengths, const float* output_lengths,
                              int32_t batch, int32_t dec, int32_t enc, float enc_min_thresh, double* out,
                              void* workspace, size_t workspace_bytes, void* stream);
/* utils/model/utils.py:47-56 (= text2speech.py:152-161) `get_first_over_thresh(x, threshold)`:
 * x [batch][T] fp32 -> out int32 [batch], first step with x >= threshold, else T-1. */
int ctts_first_over_thresh_f32(const float* x, int32_t batch, int32_t T, float threshold, int32_t* out, void* stream);

/* Decoder.inference's stop rule (_2_ttm/tacotron2_tm/model.py:879-904) on the device.  The reference copies every
 * step's gate to the host (:885) and evaluates there; here the rule runs over the gate LOGITS the decoder steps wrote,
 * block of steps by block of steps, and only its verdict (4 bytes) ever crosses to the host.
 *   state: ctts_taco_stop_state_bytes(batch) device bytes = { float sig_max[batch]; int32 break_point; int32 n_total };
 *          ctts_taco_stop_reset sets sig_max = 0, break_point = max_decoder_steps, n_total = -1.
 *   ctts_taco_stop_rule_f32 applies steps [step0, step0 + n_steps) of gate_logits [batch][gate_ld]:
 *          i > 4: sig_max = max(sig_max, sigmoid(gate)); min over the batch > gate_threshold -> break_point =
 *          min(break_point, i + gate_delay); i >= break_point -> n_total = i + 1 (sticky; later calls are no-ops).
 *   n_total is the int32 at byte offset 4 * (batch + 1) of state. */
size_t ctts_taco_stop_state_bytes(int32_t batch);
int ctts_taco_stop_reset(void* state, int32_t batch, int32_t max_decoder_steps, void* stream);
int ctts_taco_stop_rule_f32(const float* gate_logits, int32_t batch, int32_t gate_ld, int32_t step0, int32_t n_steps,
                            float gate_threshold, int32_t gate_delay, void* state, void* stream);

/* ---- main loop of the fp32 conv-GEMM ------------------------------------------------------------------------
 * Every fp32 path of the library (WaveGlow fp32, WaveFlow, the ax WaveGlow, conv1d / LSTM input projections, STFT)
 * goes through one conv-GEMM.  mode 0 (default): v_mfma_f32_32x32x2_f32, exact fp32 products.  mode 1 ("split bf16"):
 * the same kernel, tensors, packed weights and epilogues, but each operand value is split in registers into
 * hi = bf16(v), lo = bf16(v - hi) and each product is computed as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16
 * with fp32 accumulation: operands carry 16 mantissa bits (relative product error ~2^-16), the matrix pipe does
 * 3/16 of the cycles.  mode 2 ("split bf16 x6"): hi + mid + lo (24 mantissa bits = the whole fp32 operand) and the six
 * products of weight >= 2^-16 (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi, accumulated smallest first; the three dropped
 * ones are <= 2^-24 relative): fp32-grade results at 6/16 of the fp32 matrix-pipe cycles (tests/test_gemm_mode.py holds
 * its error against the reference goldens within 2x of the fp32 MFMA path's).
 * The mode is part of each model's config struct (f32_gemm_mode = CTTS_GEMM_F32 | CTTS_GEMM_BF16X3 | CTTS_GEMM_BF16X6), so two models in one
 * process can differ and nothing races with in-flight calls.  The STFT entry points always compute in fp32 MFMA (their
 * sums cancel).
 *
 * Which loop a launch really runs also depends on its SHAPE (all of it deterministic in the arguments, none of it in the
 * environment): the fused WaveFlow layer (C = 64) runs the fp32-MFMA split-K shape whenever ntiles * batch <= 128
 * (B <= 2 at 900 frames) under EVERY mode - and, under fp32 MFMA, as the item body of the row queue while a layer has 200-399
 * column tiles of 128 (B = 2-3 at 900 frames) - and the fused separable layer (C = 128) runs fp32 MFMA under CTTS_GEMM_BF16X6;
 * the split-K shape sums in a different order than the other shapes, so the same utterance is bit-identical across
 * batch sizes only within one shape (CTTS_F32_NO_SPLITK: one K order at every size).  ctts_last_gemm_loop() reports what the most recent conv-GEMM launch of the calling
 * thread ran, so that a benchmark row can label itself: bits 0-3 = split level (0 fp32 MFMA, 3, 6), bit 4 = small shape,
 * bit 5 = split-K shape, bit 6 = the WaveFlow row queue, bit 7 = its whole-flow form (one launch per flow, see ctts_waveflow_inverse_f32).
 *
 * REMOVED: the process-wide default of ABI <= 5 (ctts_set_f32_gemm_mode / ctts_get_f32_gemm_mode; no-ops in ABI 6, gone in 7):
 * hidden state shared by every model and thread of a process.  CTTS_GEMM_DEFAULT (0) in a config struct always means fp32
 * MFMA, and so do the two entry points without a config struct (ctts_lstm_seq_f32's input projection,
 * ctts_taco_decoder_init_f32's processed memory). */
int ctts_last_gemm_loop(void);

/* Launch-shape overrides for A/B measurements (CTTS_F32_NO_GLDS, CTTS_F32_NO_SMALL, CTTS_F32_FORCE_SMALL, CTTS_GEMM_NO_XCD_PAIR, CTTS_BF16_NO_GLDS / _NO_WIDE /
 * _NO_PP / _W4 / _PP_STAGES / _PS / _NO_PS / _PS_STAGES / _MAP / _WIDE_MIN, CTTS_WF_NO_FUSE, CTTS_WF_NO_VEC_INTERP, CTTS_WF_NO_REGION_SPLIT, CTTS_WF_NO_ROW_QUEUE, CTTS_F32_NO_ROUND_SPLIT, CTTS_F32_SPLITK_W4, CTTS_TACO_POLL_DELAY, CTTS_TACO_NO_FUSE, CTTS_UP_NO_MFMA) never change results beyond the parity
 * tolerance (CTTS_F32_NO_SPLITK changes the summation order of the fused WaveFlow layer at batch <= 2, see above).  The environment is read once,
 * at the first launch; this re-reads it (tests and profiling scripts that flip a knob in-process). */
int ctts_tuning_reload(void);
/* The knobs as the library currently sees them: bit 0 CTTS_F32_NO_GLDS, 1 CTTS_GEMM_NO_XCD_PAIR, 2 CTTS_BF16_NO_GLDS,
 * 3 CTTS_BF16_NO_WIDE, 4 CTTS_BF16_NO_PP, 5 CTTS_BF16_W4, 6 CTTS_BF16_PP_STAGES=4, 7 CTTS_WF_NO_FUSE, 8 CTTS_TACO_NO_FUSE,
 * 9 CTTS_F32_NO_SMALL, 10 CTTS_F32_FORCE_SMALL, 11 CTTS_F32_NO_SPLITK, 12 CTTS_WF_NO_VEC_INTERP, 13 CTTS_F32_NO_DEFER_SKIP, 14 CTTS_WF_NO_REGION_SPLIT,
 * 15 CTTS_WF_NO_ROW_QUEUE, 16 CTTS_WF_ROW_QUEUE_MIN set, 17 CTTS_WF_INJECT_ABORT, 18 CTTS_WF_QUEUE_DEBUG != 0, 19 CTTS_F32_NO_ROUND_SPLIT,
 * 20 CTTS_BF16_PS (persistent form of the skewed bf16 kernel on every wide launch), 21 CTTS_BF16_NO_PS, 22 CTTS_F32_SPLITK_W4 (per-layer
 * launches of the split-K shape on four waves per tile instead of eight: bit-identical), 23 CTTS_TACO_POLL_DELAY set ("a,c,d,e,h,p": 64-cycle units
 * before the first poll of the persistent decoder's six vector exchanges; timing only), 24 CTTS_TACO_VALU
 * (ctts_taco_decoder_steps_f32 at batch <= 4 on the VALU kernels instead of the batched MFMA form), 25 CTTS_UP_NO_MFMA (the VALU upsampling kernel
 * also for the shape the MFMA one is built for: bit-identical) (tests assert that a knob they set is the one in effect). */
int ctts_tuning_flags(void);

/* ---- in-library kernel timing (bench.py roofline leg) ---------------------------------
 * A profile is an opaque caller-owned handle (ABI 6; until ABI 5 one set of process-global slots).  A thread that has
 * BOUND a handle brackets every launch of the WN GEMMs it issues (ctts_waveglow_infer_* on that thread) with hipEvents on
 * the launch's stream and files them in the handle; ctts_profile_bind(NULL) stops recording on the calling thread.  Two
 * threads with two handles never see each other's launches.  collect synchronises the recorded events of one slot,
 * returns launches + summed milliseconds and empties the slot.  destroy: no thread may still be bound to the handle. */
#define CTTS_PROF_WN_IN 0
#define CTTS_PROF_WN_RS 1   /* fp32: res/skip GEMM; bf16: res GEMM (x += W_res act) */
#define CTTS_PROF_WN_SKIP 2 /* deferred skip GEMM over K = (layers of the group) * C */
#define CTTS_PROF_N 3
int ctts_profile_create(void** handle);
int ctts_profile_bind(void* handle);
int ctts_profile_collect(void* handle, int32_t which, int64_t* launches, double* total_ms);
int ctts_profile_destroy(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* COOKIETTS_HIP_H */
