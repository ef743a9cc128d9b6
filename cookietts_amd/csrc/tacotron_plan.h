// Packed-weight plan and workspace layout of the Tacotron2-TM decoder, shared by the per-launch kernels
// (tacotron_decoder.hip) and the persistent kernel (tacotron_persistent.hip).
#pragma once

#include "gemm_f32.h"
#include "waveglow_kernels.h"

namespace ctts {
namespace taco {

constexpr size_t ALIGN_F = 64;
inline size_t align_up(size_t v) { return (v + ALIGN_F - 1) / ALIGN_F * ALIGN_F; }
constexpr int MAX_NB = 4;          // the VALU per-launch kernels and the persistent kernel: items per launch
constexpr int MAX_BATCH = 256;     // the batched MFMA form (tacotron_batched.h): items per workspace
constexpr int BG_MT = 16, BG_KC = 16;   // batched form: rows per m-tile, K columns per chunk (one v_mfma_f32_16x16x4_f32 x 4)

// One GEMM of the batched form: `tiles` m-tiles of 16 rows x `nchunks` chunks of 16 columns, packed [tile][chunk][lane][4]
struct BgMat { size_t off; int rows, tiles, nchunks; };

struct DecPlan {
    ctts_taco_decoder_config c;
    int I_att, I_dec, I_d2, Dproj;
    size_t bottleneck_wT, memory_wT, query_w, v_w, loc_conv_w, loc_dense_w, loc_fold, prenet_w1, prenet_w2;
    size_t att[4], dec[4], d2[4];
    size_t proj_w, proj_b, scalars;   // proj rows: n_mel mel rows then the gate row; scalars: offset, smoothing
    // persistent decoder (tacotron_persistent.hip): rows [0, n_mel) mel, row n_mel gate, rows [n_mel+1, n_mel+1+P) the
    // first prenet layer folded through the mel projection (W1 . Wp, bias W1 . bp); second prenet layer row-major
    size_t pd_proj_w, pd_proj_b, pd_w2;
    int pd_rows;
    // batched form (batch > 4): the three cells (rows gate-interleaved: tile row 4 j + g = gate g of unit 4 tile + j), the
    // query rows, the folded projection row set,
    // the second prenet layer.  bg_ok = every K a multiple of 64 and every X piece a multiple of 16 wide.
    BgMat bg_att, bg_dec, bg_d2, bg_q, bg_proj, bg_w2;
    bool bg_ok;
    size_t total;
};

inline int make_dec_plan(const ctts_taco_decoder_config* cfg, DecPlan& p) {
    CTTS_CHECK_ARG(cfg != nullptr, "decoder config is NULL");
    p.c = *cfg;
    const auto& c = p.c;
    CTTS_CHECK_ARG(c.n_mel_channels >= 1 && c.n_mel_channels <= 256, "n_mel_channels=%d", c.n_mel_channels);
    CTTS_CHECK_ARG(c.prenet_dim >= 4 && c.prenet_dim <= 256 && c.prenet_dim % 4 == 0 && c.n_mel_channels % 4 == 0,
                   "prenet_dim=%d n_mel=%d (multiples of 4, prenet <= 256)", c.prenet_dim, c.n_mel_channels);
    CTTS_CHECK_ARG(c.memory_dim % 4 == 0 && c.attention_rnn_dim % 4 == 0 && c.decoder_rnn_dim % 4 == 0 &&
                   c.second_decoder_rnn_dim == c.decoder_rnn_dim, "rnn dims must be multiples of 4 and Rd2 == Rd");
    CTTS_CHECK_ARG(c.attention_rnn_dim <= 1536 && c.second_decoder_rnn_dim + c.memory_dim <= 1536, "rnn dims exceed the 1536-wide staging");
    CTTS_CHECK_ARG(c.attention_dim >= 1 && c.attention_dim <= 256 && c.location_n_filters >= 1 &&
                   c.location_n_filters <= 64 && c.location_kernel_size % 2 == 1 && c.location_kernel_size <= 63,
                   "attention shape");
    CTTS_CHECK_ARG(c.window_range >= 1 && c.window_range <= 31, "window_range=%d", c.window_range);
    CTTS_CHECK_ARG(c.memory_in_dim >= 1, "memory_in_dim");
    p.I_att = c.prenet_dim + c.memory_dim + c.decoder_rnn_dim;
    p.I_dec = c.attention_rnn_dim + c.memory_dim;
    p.I_d2 = c.decoder_rnn_dim;
    p.Dproj = c.second_decoder_rnn_dim + c.memory_dim;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    p.bottleneck_wT = take((size_t)c.memory_in_dim * c.memory_dim);
    p.memory_wT = take((size_t)c.memory_dim * c.attention_dim);
    p.query_w = take((size_t)c.attention_dim * c.attention_rnn_dim);
    p.v_w = take(c.attention_dim);
    p.loc_conv_w = take((size_t)c.location_n_filters * 2 * c.location_kernel_size);
    p.loc_dense_w = take((size_t)c.attention_dim * c.location_n_filters);
    p.prenet_w1 = take((size_t)c.n_mel_channels * c.prenet_dim);
    p.prenet_w2 = take((size_t)c.prenet_dim * c.prenet_dim);
    auto lstm = [&](size_t* a, int I, int H) {
        a[0] = take((size_t)4 * H * I); a[1] = take((size_t)4 * H * H); a[2] = take(4 * H); a[3] = take(4 * H);
    };
    lstm(p.att, p.I_att, c.attention_rnn_dim);
    lstm(p.dec, p.I_dec, c.decoder_rnn_dim);
    lstm(p.d2, p.I_d2, c.second_decoder_rnn_dim);
    p.proj_w = take((size_t)(c.n_mel_channels + 1) * p.Dproj);
    p.proj_b = take(c.n_mel_channels + 1);
    p.scalars = take(4);
    p.pd_rows = c.n_mel_channels + 1 + c.prenet_dim;
    p.pd_proj_w = take((size_t)p.pd_rows * p.Dproj);
    p.pd_proj_b = take(p.pd_rows);
    p.pd_w2 = take((size_t)c.prenet_dim * c.prenet_dim);
    auto mat = [&](BgMat& m, int rows, int K) {
        m.rows = rows; m.tiles = (rows + BG_MT - 1) / BG_MT; m.nchunks = K / BG_KC;
        m.off = take((size_t)m.tiles * m.nchunks * 64 * 4);
    };
    const int w16 = c.prenet_dim % 16 | c.memory_dim % 16 | c.decoder_rnn_dim % 16 | c.attention_rnn_dim % 16;
    const int Ks[6] = {p.I_att + c.attention_rnn_dim, p.I_dec + c.decoder_rnn_dim, p.I_d2 + c.second_decoder_rnn_dim,
                       c.attention_rnn_dim, c.second_decoder_rnn_dim + c.memory_dim, c.prenet_dim};
    // ... and the windowed-attention kernel's own limits (attention_window_kernel: AW / ADM / AAD / AF / AK in tacotron_decoder.hip)
    p.bg_ok = w16 == 0 && c.window_range <= 16 && c.memory_dim <= 512 && c.attention_dim <= 256 && c.attention_dim % 4 == 0 &&
              c.location_n_filters <= 32 && c.location_kernel_size <= 31;
    for (int k : Ks) p.bg_ok = p.bg_ok && k % 64 == 0;
    if (p.bg_ok) {
        mat(p.bg_att, 4 * c.attention_rnn_dim, Ks[0]);
        mat(p.bg_dec, 4 * c.decoder_rnn_dim, Ks[1]);
        mat(p.bg_d2, 4 * c.second_decoder_rnn_dim, Ks[2]);
        mat(p.bg_q, c.attention_dim, Ks[3]);
        mat(p.bg_proj, p.pd_rows, Ks[4]);
        mat(p.bg_w2, c.prenet_dim, Ks[5]);
        p.loc_fold = take((size_t)2 * c.location_kernel_size * c.attention_dim);    // location conv folded into the dense layer [2][K][A]
    } else {
        p.loc_fold = 0;
        p.bg_att = p.bg_dec = p.bg_d2 = p.bg_q = p.bg_proj = p.bg_w2 = BgMat{0, 0, 0, 0};
    }
    p.total = o;
    return CTTS_OK;
}

struct DecWs {
    float *memory, *pm, *att_h[2], *att_c, *dec_h[2], *dec_c, *d2_h[2], *d2_c, *w, *cum, *ctx, *pos, *prenet, *qbuf, *gp_att, *gp_dec, *gp_d2, *h1, *apre, *hsum;
    int *lengths, *astart;
    float *part_att, *part_dec, *part_d2;   // batched form, pipelined step: EARLY sums of the cells [slot][m-tiles + 4][NB / 16][64][4] (2, 2, 1 slots)
    size_t total;
};

// batch rows a workspace holds: 1 / 2 / 4 for the small forms; the batched form pads to whole column tiles of its launch
// shape (16 x NT items per workgroup column, NT = 1 / 2 / 4): 16, 32, then multiples of 64
inline int pad_batch(int b) { return b <= 1 ? 1 : b <= 2 ? 2 : b <= 4 ? 4 : b <= 16 ? 16 : b <= 32 ? 32 : (b + 63) / 64 * 64; }

// rows per state array: where the batched form is built the workspace always holds at least one of its column tiles (16 items),
// so that it can also serve batch <= 4 (arrays are dense [item][n]: extra rows change no address of the small forms)
inline int ws_rows(const DecPlan& p, int batch) { return p.bg_ok && pad_batch(batch) < 16 ? 16 : pad_batch(batch); }

inline void dec_carve(const DecPlan& p, int batch, int T, float* base, DecWs& w) {
    const auto& c = p.c;
    const size_t NB = ws_rows(p, batch);
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return base ? base + r : nullptr; };
    w.memory = take(NB * T * c.memory_dim);
    w.pm = take(NB * T * c.attention_dim);
    for (int i = 0; i < 2; ++i) w.att_h[i] = take(NB * c.attention_rnn_dim);
    w.att_c = take(NB * c.attention_rnn_dim);
    for (int i = 0; i < 2; ++i) w.dec_h[i] = take(NB * c.decoder_rnn_dim);
    w.dec_c = take(NB * c.decoder_rnn_dim);
    for (int i = 0; i < 2; ++i) w.d2_h[i] = take(NB * c.second_decoder_rnn_dim);
    w.d2_c = take(NB * c.second_decoder_rnn_dim);
    w.w = take(NB * T); w.cum = take(NB * T);
    w.ctx = take(NB * c.memory_dim);
    w.pos = take(NB);
    w.prenet = take(NB * c.prenet_dim);
    w.qbuf = take(NB * c.attention_dim);
    w.gp_att = take(NB * 4 * c.attention_rnn_dim);          // early partial pre-activations of the three cells
    w.gp_dec = take(NB * 4 * c.decoder_rnn_dim);
    w.gp_d2 = take(NB * 4 * c.second_decoder_rnn_dim);
    w.h1 = take(NB * c.prenet_dim);                          // batched form: first prenet layer of the next step
    w.apre = take(p.bg_ok ? NB * 33 * c.attention_dim : 0);  // batched form: processed-memory window + location term (attn_pre_body)
    w.astart = reinterpret_cast<int*>(take(NB));             // ... and the window start
    w.hsum = take(p.bg_ok ? NB * c.second_decoder_rnn_dim : 0);   // batched form: dec_h + d2_h (the projection's input)
    {
        const size_t nt = NB / 16;
        const bool pipe = p.bg_ok && NB >= 16 && NB <= 64;
        w.part_att = take(pipe ? 2 * (size_t)(c.attention_rnn_dim / 4 + 4) * nt * 256 : 0);
        w.part_dec = take(pipe ? 2 * (size_t)(c.decoder_rnn_dim / 4 + 4) * nt * 256 : 0);
        w.part_d2 = take(pipe ? (size_t)(c.second_decoder_rnn_dim / 4 + 4) * nt * 256 : 0);
    }
    w.lengths = reinterpret_cast<int*>(take(NB));
    w.total = o;
}


}  // namespace taco
}  // namespace ctts
