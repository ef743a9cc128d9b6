// Launchers of the bf16 MFMA conv-GEMM for gfx950 (contract: gemm_bf16.h; kernels: gemm_bf16_kernels.h).
#include <cstdlib>

#include "gemm_bf16_kernels.h"
#include "gemm_f32.h"
#include "tuning.h"

namespace ctts {

int launch_pack_a_bf16(bf16_t* dst, const float* src, int MB, int nch_total, int k_off, int ksrc, int epi, int C,
                       int M, long long src_row_off, long long src_row_stride, int src_k_stride, hipStream_t s,
                       int k_group, int k_member, int part, int f16) {
    // a ragged ksrc (e.g. a 20-wide speaker embedding) is zero-filled up to its 32-wide slab boundary
    const int kfill = (ksrc + BGEMM_KC - 1) / BGEMM_KC * BGEMM_KC;
    CTTS_CHECK_ARG(k_off % 8 == 0 && ksrc > 0, "pack_a_bf16: k offset must be 8-aligned");
    CTTS_CHECK_ARG(k_off % BGEMM_KC == 0 || ksrc % 8 == 0, "pack_a_bf16: a ragged k range must start on a slab");
    const int kspan = k_off % BGEMM_KC == 0 ? kfill : ksrc;
    CTTS_CHECK_ARG(k_off + kspan * (k_group > 1 ? k_group : 1) <= nch_total * BGEMM_KC, "pack_a_bf16: k range");
    CTTS_CHECK_ARG(k_group <= 1 || (ksrc % BGEMM_KC == 0 && k_off % BGEMM_KC == 0), "pack_a_bf16: k group");
    hipLaunchKernelGGL(pack_a_bf16_kernel, dim3(kspan / 8, MB), dim3(256), 0, s, dst, src, nch_total, k_off, ksrc, epi, C,
                       M, src_row_off, src_row_stride, src_k_stride, k_group, k_member, part, f16);
    CTTS_CHECK_LAUNCH("pack_a_bf16");
    return CTTS_OK;
}

#ifdef CTTS_W4_TIMING_EXPERIMENTS
extern "C" int ctts_debug_w4_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_stamps), sizeof(g_w4_stamps)) == hipSuccess ? 0 : -1;
}
#endif

int launch_gemm_bf16(int epi, const BGemmArgs& a, hipStream_t stream) {
    CTTS_CHECK_ARG(a.nseg >= 1 && a.nseg <= BGEMM_MAX_SEG, "gemm_bf16: nseg=%d", a.nseg);
    int nch = 0;
    for (int s = 0; s < a.nseg; ++s) {
        CTTS_CHECK_ARG(a.seg[s].nch > 0 && a.seg[s].base, "gemm_bf16: empty segment %d", s);
        CTTS_CHECK_ARG(a.seg[s].shift >= -a.pad && a.seg[s].shift <= a.pad, "gemm_bf16: shift %d exceeds halo %d",
                       a.seg[s].shift, a.pad);
        nch += a.seg[s].nch;
    }
    CTTS_CHECK_ARG(nch == a.nch_total, "gemm_bf16: chunk count mismatch %d vs %d", nch, a.nch_total);
    if (a.interleave > 1) {
        CTTS_CHECK_ARG(a.interleave <= a.nseg, "gemm_bf16: interleave %d > nseg %d", a.interleave, a.nseg);
        for (int s = 1; s < a.interleave; ++s)
            CTTS_CHECK_ARG(a.seg[s].nch == a.seg[0].nch, "gemm_bf16: interleaved segments must have equal length");
    }
    // block-shape overrides for A/B tests (tuning.h: the environment is read once; ctts_tuning_reload re-reads it)
    const Tuning tune = tuning();
    const bool use_glds = !tune.bf16_no_glds;
    const bool no_wide = tune.bf16_no_wide;
    const bool no_pp = tune.bf16_no_pp;
    const int pp_stages = tune.bf16_pp_stages;
    // four-wave 128 x 128 wave tiles: opt-in.  Measured equal to the skewed 8-wave kernel on the in-layer GEMM (both
    // are held by the clock the chip sustains under bf16 MFMA + LDS traffic) and slower on the short-K
    // res GEMM, where its one-wave-per-SIMD epilogue is exposed.
    const bool w4 = tune.bf16_w4;
    // wide (256 x 256, 512 threads) tiles when the problem has enough of them (tune.bf16_wide_min: most of one round of the chip)
    const int ntiles_w = (a.L + 255) / 256;
    const bool pp = !no_pp && a.nch_total + 3 <= BGEMM_PP_MAX_CHUNKS;
    const bool wide = use_glds && !no_wide && (long long)a.MB * ntiles_w * a.batch >= tune.bf16_wide_min && ntiles_w * 256 + 2 * a.pad <= a.ld;
    BGemmArgs b = a;
    if (wide) b.ntiles = ntiles_w;
    const int bn = wide ? 256 : BGEMM_BN;
    CTTS_CHECK_ARG(b.ntiles * bn + 2 * b.pad <= b.ld && b.L <= b.ntiles * bn, "gemm_bf16: geometry");
    CTTS_CHECK_ARG(epi == BGEMM_EPI_GATE ? (b.pairC > (b.MB - 1) * 128 && b.pairC <= b.MB * 128)
                                         : (b.M % 32 == 0 && b.split % 32 == 0 && b.M > (b.MB - 1) * BGEMM_BM &&
                                            b.M <= b.MB * BGEMM_BM),
                   "gemm_bf16: M=%d pairC=%d MB=%d split=%d", b.M, b.pairC, b.MB, b.split);
    long long blocks = (long long)b.MB * b.ntiles * b.batch;
    b.map_mode = 0;
    const long long tiles = (long long)b.ntiles * b.batch;
    // persistent kernel: one workgroup per CU walks its tile sequence (gemm_bf16_kernels.h); K of more than NS chunks.
    // Default for short K (<= 32 chunks: config 3's res GEMM, K = 512, where a tile's prologue is a large share of its
    // time: 0.729 -> 0.675 ms); on the long-K launches it measures equal (in-layer) or behind (skip) the per-tile kernel
    // (profiles/r5_14_bf16_ps_ab.txt).  CTTS_BF16_PS=1: everywhere it applies; CTTS_BF16_NO_PS: nowhere.
    const int ps_stages = tune.bf16_ps_stages;
    const bool ps = wide && pp && !w4 && !tune.bf16_no_ps && (tune.bf16_ps || b.nch_total <= 32) && b.nch_total > ps_stages &&
                    tune.bf16_map != 2;
    const int cus = ps ? wf_row_cus() : 0;
    int ps_grid = 0;
    if (b.MB == 4 && !tune.no_xcd_pair && tune.bf16_map == 2) {
        b.map_mode = 3;
        blocks = 32ll * ((tiles + 7) / 8);
    } else if (b.MB == 4 && !tune.no_xcd_pair) {
        b.map_mode = 1;
        blocks = 16ll * ((tiles + 3) / 4);
        ps_grid = cus / 16 * 16;
    } else if (b.MB == 2 && !tune.no_xcd_pair && tune.bf16_map != 1) {
        b.map_mode = 2;
        blocks = 16ll * ((tiles + 7) / 8);
        ps_grid = cus / 16 * 16;
    } else {
        ps_grid = cus / b.MB * b.MB;
    }
    CTTS_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "gemm_bf16: grid %lld", blocks);
    const dim3 grid((unsigned)blocks);
    if (b.f16) {
        // IEEE-half operands: the default shapes only (the A/B knobs of the bf16 kernels other than PS / NO_PS / NO_WIDE do not apply)
        CTTS_CHECK_ARG(b.lo_off == 0, "gemm_bf16: the split (hi + lo) form exists for bf16 only");
        // ... and a knob without an IEEE-half instantiation is refused, not ignored: an A/B run would compare two identical kernels
        CTTS_CHECK_ARG(use_glds && !w4 && pp_stages != 4 && ps_stages != 3,
                       "gemm_bf16: CTTS_BF16_NO_GLDS / _W4 / _PP_STAGES=4 / _PS_STAGES=3 select bf16-only kernels; unset them for the f16 path");
        const dim3 pg((unsigned)(ps_grid > 0 ? ps_grid : 1));
        if (epi == BGEMM_EPI_GATE) {
            if (ps && ps_grid >= 16 && b.nch_total > 4) hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<BGEMM_EPI_GATE, 4, 0, true>), pg, dim3(512), 0, stream, b);
            else if (wide && pp) hipLaunchKernelGGL((conv_gemm_bf16_pp_kernel<BGEMM_EPI_GATE, 3, true>), grid, dim3(512), 0, stream, b);
            else if (wide) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_GATE, true, 4, true>), grid, dim3(512), 0, stream, b);
            else hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_GATE, true, 2, true>), grid, dim3(256), 0, stream, b);
        } else {
            if (ps && ps_grid >= 16 && b.nch_total > 4) hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<BGEMM_EPI_SPLIT, 4, 0, true>), pg, dim3(512), 0, stream, b);
            else if (wide && pp) hipLaunchKernelGGL((conv_gemm_bf16_pp_kernel<BGEMM_EPI_SPLIT, 3, true>), grid, dim3(512), 0, stream, b);
            else if (wide) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_SPLIT, true, 4, true>), grid, dim3(512), 0, stream, b);
            else hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_SPLIT, true, 2, true>), grid, dim3(256), 0, stream, b);
        }
    } else if (ps && ps_grid >= 16) {
        const dim3 pg((unsigned)ps_grid);
        if (epi == BGEMM_EPI_GATE) {
            if (ps_stages == 3) hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<BGEMM_EPI_GATE, 3>), pg, dim3(512), 0, stream, b);
            else hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<BGEMM_EPI_GATE, 4>), pg, dim3(512), 0, stream, b);
        } else {
            if (ps_stages == 3) hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<BGEMM_EPI_SPLIT, 3>), pg, dim3(512), 0, stream, b);
            else hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<BGEMM_EPI_SPLIT, 4>), pg, dim3(512), 0, stream, b);
        }
    } else if (epi == BGEMM_EPI_GATE) {
#ifdef CTTS_W4_TIMING_EXPERIMENTS     /* stage-removal variants of the four-wave kernel (scripts/w4_stamps.py); not in the product build */
        const int dbg = tune.w4_debug;
#define CTTS_W4_DBG(D) if (wide && pp && w4 && dbg == D) hipLaunchKernelGGL((conv_gemm_bf16_w4_kernel<BGEMM_EPI_GATE, D>), grid, dim3(256), 0, stream, b); else
        CTTS_W4_DBG(1) CTTS_W4_DBG(2) CTTS_W4_DBG(4) CTTS_W4_DBG(5) CTTS_W4_DBG(6)
#undef CTTS_W4_DBG
#endif
        if (wide && pp && w4) hipLaunchKernelGGL((conv_gemm_bf16_w4_kernel<BGEMM_EPI_GATE>), grid, dim3(256), 0, stream, b);
        else if (wide && pp && pp_stages == 4) hipLaunchKernelGGL((conv_gemm_bf16_pp_kernel<BGEMM_EPI_GATE, 4>), grid, dim3(512), 0, stream, b);
        else if (wide && pp) hipLaunchKernelGGL((conv_gemm_bf16_pp_kernel<BGEMM_EPI_GATE, 3>), grid, dim3(512), 0, stream, b);
        else if (wide) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_GATE, true, 4>), grid, dim3(512), 0, stream, b);
        else if (use_glds) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_GATE, true, 2>), grid, dim3(256), 0, stream, b);
        else hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_GATE, false, 2>), grid, dim3(256), 0, stream, b);
    } else {
        if (wide && pp && w4) hipLaunchKernelGGL((conv_gemm_bf16_w4_kernel<BGEMM_EPI_SPLIT>), grid, dim3(256), 0, stream, b);
        else if (wide && pp && pp_stages == 4) hipLaunchKernelGGL((conv_gemm_bf16_pp_kernel<BGEMM_EPI_SPLIT, 4>), grid, dim3(512), 0, stream, b);
        else if (wide && pp) hipLaunchKernelGGL((conv_gemm_bf16_pp_kernel<BGEMM_EPI_SPLIT, 3>), grid, dim3(512), 0, stream, b);
        else if (wide) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_SPLIT, true, 4>), grid, dim3(512), 0, stream, b);
        else if (use_glds) hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_SPLIT, true, 2>), grid, dim3(256), 0, stream, b);
        else hipLaunchKernelGGL((conv_gemm_bf16_kernel<BGEMM_EPI_SPLIT, false, 2>), grid, dim3(256), 0, stream, b);
    }
    CTTS_CHECK_LAUNCH("conv_gemm_bf16");
    return CTTS_OK;
}

}  // namespace ctts
