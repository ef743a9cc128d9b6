// Launch-shape overrides for A/B measurements.  The shapes of one GEMM are tested equal within the parity tolerance, not
// all bit-identical: the split-K shape of the fused WaveFlow layer (CTTS_F32_NO_SPLITK turns it off) sums K in a different
// order AND always computes in fp32 MFMA, also under the split-bf16 modes (include/cookietts_hip.h, "Which loop a launch
// really runs"; ctts_last_gemm_loop reports it).
// The environment is read ONCE, at the first launch that asks; ctts_tuning_reload() re-reads it (tests, profiling
// scripts).  Arithmetic choices (fp32 MFMA vs split-bf16 main loop) are NOT here: they travel in the config structs.
#pragma once

namespace ctts {

struct Tuning {
    bool f32_no_glds;      // CTTS_F32_NO_GLDS: fp32 conv-GEMM without the LDS-DMA staging
    bool f32_no_small;     // CTTS_F32_NO_SMALL: never the 128 x 64 small-problem shape (gemm_f32_small.hip)
    bool f32_force_small;  // CTTS_F32_FORCE_SMALL: the small shape whenever it applies, whatever the grid size (A/B, tests)
    bool f32_no_splitk;    // CTTS_F32_NO_SPLITK: never the split-K shape of the fused WaveFlow layer (batch 1-2)
    bool f32_splitk_w4;    // CTTS_F32_SPLITK_W4: the split-K shape's per-layer launches on four waves per tile (the form before round 5), not eight
    bool f32_no_round_split;  // CTTS_F32_NO_ROUND_SPLIT: never peel the tiles beyond the last whole round of workgroups off a large-shape launch
    bool no_xcd_pair;      // CTTS_GEMM_NO_XCD_PAIR: plain block id -> tile mapping
    bool bf16_no_glds;     // CTTS_BF16_NO_GLDS
    bool bf16_no_wide;     // CTTS_BF16_NO_WIDE: never the 256 x 256 block
    int bf16_wide_min;     // CTTS_BF16_WIDE_MIN: 256 x 256 tiles (m-blocks x column tiles x batch) from which the wide block is taken (default in gemm_f32.hip load_tuning)
    bool bf16_no_pp;       // CTTS_BF16_NO_PP: never the ping-pong kernel
    bool bf16_w4;          // CTTS_BF16_W4: four-wave 128 x 128 wave tiles (opt-in)
    int bf16_pp_stages;    // CTTS_BF16_PP_STAGES: 3 (default) or 4
    bool bf16_ps;          // CTTS_BF16_PS: the persistent form of the skewed 8-wave kernel (one workgroup per CU walks a tile sequence) on every wide launch, not only the short-K ones
    bool bf16_no_ps;       // CTTS_BF16_NO_PS: never the persistent form
    int bf16_ps_stages;    // CTTS_BF16_PS_STAGES: LDS stages of the persistent-stream kernel, 4 (default) or 3
    int bf16_map;          // CTTS_BF16_MAP: (A/B) 1 = MB == 2 launches keep the plain id -> tile map, 2 = MB == 4 launches put all four m-blocks of a tile on one XCD
    bool wf_no_fuse;       // CTTS_WF_NO_FUSE: WaveFlow layer as separate GATE + res/skip launches
    bool taco_poll_delay_set;
    int taco_poll_delay[6];  // CTTS_TACO_POLL_DELAY="a,c,d,e,h,p": persistent decoder, s_sleep(1) units before the first poll of the att_h, ctx, dec_h, d2_h, h1, prenet exchanges (+ 65536: no light phase, straight to the full sweep; default 65572,65632,65548,65556,65544,65544; all 0: the form before round 5)
    bool taco_no_fuse;     // CTTS_TACO_NO_FUSE: per-launch decoder without the fused projection kernel
    int taco_bg_debug;     // CTTS_TACO_BG_DEBUG: timing experiments of attn_post_kernel (wrong results): 1 no tanh, 2 no context, 4 no alignment row, 8 no memory DMA, 16 empty
    bool taco_bg_no_pipe;  // CTTS_TACO_BG_NO_PIPE: batched decoder without the pipelined step (no EARLY cell sums in the small stages' launches)
    int taco_bg_shape;     // CTTS_TACO_BG_SHAPE=100 MTW + S: launch shape of the batched decoder's cell GEMMs (A/B; 0 = by batch)
    bool up_no_mfma;       // CTTS_UP_NO_MFMA: the VALU upsampling kernel also for the shape the MFMA one is built for (A/B)
    bool taco_valu;        // CTTS_TACO_VALU: ctts_taco_decoder_steps_f32 at batch <= 4 on the VALU kernels of rounds 1-3 (six launches per step) instead of the batched MFMA form
    bool f32_no_defer_skip;  // CTTS_F32_NO_DEFER_SKIP: WaveGlow fp32 WN stack with one res/skip GEMM per layer (the form before round 4)
    bool wf_no_region_split; // CTTS_WF_NO_REGION_SPLIT: the fused WaveFlow layer as ONE launch per layer (no A | M | B regions on three streams)
    bool wf_no_row_queue;  // CTTS_WF_NO_ROW_QUEUE: never the one-launch-per-row work queue of the fused WaveFlow layers
    int wf_row_queue_min;  // CTTS_WF_ROW_QUEUE_MIN: take the row queue from this many 128-column items per layer on (A/B; default in waveflow_api.hip)
    int wf_queue_debug;    // CTTS_WF_QUEUE_DEBUG: (diagnosis) 1 no dependency waits, 2 no tile body, 4 no fresh marks, 8 full release fence per item, 16 / 32 force the 128 x 128 / split-K body, 64 two workgroups per CU at every size, 128 one launch per ROW instead of one per flow, 256 no acquire fence per item, 512 split-K items run their prologue after the dependency wait
    int wf_inject_abort;   // CTTS_WF_INJECT_ABORT: (tests) start the call with the queue's abort word set
    bool wf_no_vec_interp; // CTTS_WF_NO_VEC_INTERP: the scalar form of the WaveFlow conditioning interpolation (bit-identical)
    int w4_debug;          // CTTS_BF16_W4_DEBUG (only in builds with -DCTTS_W4_TIMING_EXPERIMENTS)
};

Tuning tuning();           // snapshot (by value)
void reload_tuning();      // ctts_tuning_reload

}  // namespace ctts
