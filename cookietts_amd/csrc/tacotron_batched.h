// Batched form of the Tacotron2-TM decoder step for the batch sizes the reference's server runs (_5_infer/t2s_server/
// text2speech.py:418-424, 537, 554: up to simultaneous_texts x batch_size_per_text = 256 rows per Decoder.inference call).
//
// With 4 < B <= 256 items the three LSTM cells (layers.py:308-372), the query rows (model.py:126), the projection row set
// (model.py:757-765) and the prenet's second layer (model.py:187-190) are GEMMs  Y[rows x B] = W[rows x K] . X[K x B]  with the
// batch as the N dimension.  The 27 M weights (108 MB) stream once per step whatever B is, so the step costs about the
// same for 5 items as for 16; the products run on v_mfma_f32_16x16x4_f32 (exact fp32 multiply-add, fp32 accumulate).
//
// One kernel body, bg_body<MTW, NT, S, WAVES, EPI>, serves all six GEMMs:
//   * a workgroup = MTW m-tiles of 16 rows x 16 NT items; its WAVES waves split K (a wave = one contiguous slice of the chunks)
//     and are summed through LDS at the end - no barrier inside the K loop;
//   * weights are packed once (bg_pack_kernel) as [m-tile][chunk][lane][4]: chunk = 16 K columns, lane = (row = lane % 16,
//     kq = lane / 16) holds columns 4 kq .. 4 kq + 3 - the A operands of the chunk's four MFMAs (k index = 4 kq + s in MFMA s;
//     any assignment works as long as B uses the same one), so a wave's weight load is ONE contiguous KiB per m-tile and chunk;
//   * X is the concatenation of up to four item-major pieces [item][width] straight from the decoder workspace (prenet | ctx |
//     dec_h | att_h ...): lane (item = lane % 16, kq) reads the 16 bytes x[item][16 c + 4 kq ..] - the B operands of the same
//     four MFMAs - no transposed copy, no concatenation pass;
//   * both streams go through a wave-private LDS ring of S stages filled by global_load_lds (16 bytes per lane, S - 1 chunks
//     ahead, counted vmcnt): no VGPRs held by loads in flight, and no barrier - a wave only ever reads what it loaded itself;
//   * X bytes per weight byte = NT / MTW: every workgroup re-reads the X columns of its items (from L2), so the launch shapes
//     below give a workgroup as many m-tiles as the grid can afford (profiles/r6_*: at MTW = NT = 1 the X stream equals the
//     weight stream and the cells ran at half the weight bandwidth);
//   * D layout of the 16x16x4 MFMA: lane (j = lane / 16, col = lane % 16) holds rows 4 j .. 4 j + 3 of column col.  The cells'
//     rows are packed gate-interleaved (tile row 4 j + g = gate g of unit 4 tile + j), so after the cross-wave sum every lane
//     owns the four pre-activations of ONE (unit, item) pair and updates that cell in place.
// The windowed attention (model.py:93-161) is split where its inputs become available: attn_pre_body needs only the previous
// step's weights and position (window start, location conv, location-dense term added to the processed-memory window) and
// rides along as extra workgroups of the attention-RNN launch; attn_post_kernel (energies, softmax, context) is the part
// between the query and the context.
// Included by tacotron_decoder.hip (inside its anonymous namespace, after AttnArgs).
#pragma once

typedef float bg_f4 __attribute__((ext_vector_type(4)));
typedef unsigned bg_u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) bg_u4* bg_gptr;
typedef __attribute__((address_space(3))) bg_u4* bg_lptr;

enum { BG_EPI_CELL = 0, BG_EPI_LINEAR = 1, BG_EPI_PROJ = 2, BG_EPI_PRENET2 = 3, BG_EPI_SEQ = 4 };

struct BgSeg { const float* ptr; int ld, col0, width; };   // pack time: `width` columns of a row-major matrix from column col0
struct BgPiece { const float* ptr; int width; };            // run time: X piece, dense item-major [item][width]

struct BgArgs {
    const float* W;             // packed tiles of this GEMM
    int nchunks, tiles;         // chunks per tile (K / 16), m-tiles
    BgPiece x[4];
    int cend[4];                // chunk index at which piece i ends (cend[3] = nchunks)
    int rows, batch;            // valid rows (the last tile may be padded), real items (columns beyond are padding)
    // A GEMM may be spread over several LAUNCHES (pipelined step, batched_steps): this role covers chunks [c_begin, c_end) of K;
    // pmode 1 (EARLY): its sums go to slot `pslot` of `part` and nothing else happens; pmode 2 (FINAL): the sums of slots
    // [0, npart) are added (slot order, then its own) before the epilogue.  pmode 0, c_end = 0: the whole K in one launch.
    // part: [slot][m-tile][item tile][lane][4].
    int c_begin, c_end, pmode, pslot, npart, nt_total; float* part;
    // BG_EPI_CELL: LSTMCell, gate order i, f, g, o; bias and state in the checkpoint / workspace layouts.  hsum (second decoder RNN):
    // also hsum[ix] = h' + hres[ix], the residual sum dec_h + d2_h that the gate / mel projection reads (model.py:755-759)
    const float *bih, *bhh; float *c, *h_new; int H; float* hsum; const float* hres;
    // BG_EPI_SEQ: one time step of a packed-sequence LSTM direction (the encoder's BiLSTM): the input projection of every
    // step is precomputed (sq.gadd, biases included), item b is active while sq.step < lengths[b] (see LstmSeq)
    LstmSeq sq; const float* h_old;
    // BG_EPI_LINEAR: y[item * ldy + row] = sum
    float* y; int ldy;
    // BG_EPI_PROJ: rows [0, n_mel) mel, n_mel gate, then the first prenet layer folded through the mel projection;
    // BG_EPI_PRENET2: the second prenet layer.  keep = this step's NEXT-step masks [2][batch][P] (model.py:189-190), or NULL
    const float* bias; const unsigned char* keep; float *mel_out, *gate_out, *act_out; int n_mel, P, step, max_steps;
};

// dst[((tile * nchunks + c) * 64 + lane) * 4 + e] = W'[16 tile + lane % 16][16 c + 4 (lane / 16) + e], W' = the column
// concatenation of `nseg` segments; interleave: W' row 4 u + g = source row g * H + u (the four gates of unit u together)
struct BgPackArgs { BgSeg seg[4]; int nseg, rows, tiles, nchunks, interleave_H; float* dst; };
__global__ __launch_bounds__(256) void bg_pack_kernel(const BgPackArgs a) {
    const size_t total = (size_t)a.tiles * a.nchunks * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e = i & 3, lane = (i >> 2) & 63;
        const size_t tc = i >> 8;
        const int c = (int)(tc % a.nchunks), tile = (int)(tc / a.nchunks);
        const int r = BG_MT * tile + (lane & 15);
        int k = BG_KC * c + 4 * (lane >> 4) + e;
        float v = 0.f;
        if (r < a.rows) {
            const int src_row = a.interleave_H ? (r & 3) * a.interleave_H + (r >> 2) : r;
            for (int s = 0; s < a.nseg; ++s) {
                if (k < a.seg[s].width) { v = a.seg[s].ptr[(size_t)src_row * a.seg[s].ld + a.seg[s].col0 + k]; break; }
                k -= a.seg[s].width;
            }
        }
        a.dst[i] = v;
    }
}

__device__ __forceinline__ float bg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int MTW, int NT, int S, int WAVES>
constexpr int bg_lds_bytes() { return WAVES * S * (MTW + NT) * 1024; }

// blk = which group of MTW m-tiles, ngrp = which group of 16 NT items
// (Fusing dependent stages into one launch through arrival counters - query rows + attention part 2, second decoder RNN +
//  projection + prenet layer, projection + prenet layer alone; fences or fence-free sc1 hand-off - was built and measured: equal
//  at best, up to 1.4x slower where a streaming stage shares the launch: profiles/r6_06.  Seven launches per step it is.)
template <int MTW, int NT, int S, int WAVES, int EPI>
__device__ __forceinline__ void bg_body(const BgArgs& a, bg_u4* lds, int blk, int ngrp) {
    constexpr int UNITS = MTW + NT;                       // 1 KiB units per stage: MTW weight tiles + NT item tiles of one chunk
    static_assert((S - 1) * UNITS <= 60, "vmcnt is a 6-bit counter");
    static_assert(S * UNITS >= MTW * NT, "the ring also holds the waves' partial sums");
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile0 = blk * MTW, n0 = ngrp * (16 * NT);
    const int cbase = a.c_end ? a.c_begin : 0, nch_wg = (a.c_end ? a.c_end : a.nchunks) - cbase;   // this role's chunks [cbase, cbase + nch_wg)
    const int per = (nch_wg + WAVES - 1) / WAVES;
    const int c0 = cbase + min(wave * per, nch_wg), cpw = min(per, cbase + nch_wg - c0);  // this wave's chunks [c0, c0 + cpw)
    bg_u4* ring = lds + wave * (S * UNITS * 64);

    const float* wl[MTW];                                  // weight cursor of the issue side: tile, chunk c0, this lane
#pragma unroll
    for (int m = 0; m < MTW; ++m) wl[m] = a.W + (((size_t)min(tile0 + m, a.tiles - 1) * a.nchunks + c0) * 64 + lane) * 4;
    const int item = n0 + (lane & 15), kq4 = 4 * (lane >> 4);
    // X cursor of the issue side: piece pi, `left` chunks of it to go, this lane's address.  Pieces change a few times per
    // launch; inside the loop nothing is looked up (a table lookup per chunk - scalar loads from the kernarg segment - made
    // every iteration a chain of dependent latencies: profiles/r6_04)
    int pi = (c0 >= a.cend[0]) + (c0 >= a.cend[1]) + (c0 >= a.cend[2]);
    int xwidth = a.x[pi].width;
    int left = a.cend[pi] - c0;
    const float* xp = a.x[pi].ptr + (size_t)item * xwidth + (c0 - (pi ? a.cend[pi - 1] : 0)) * BG_KC + kq4;

    auto issue = [&](int st) {                            // the next chunk of this wave's slice -> stage st
        bg_lptr dst = (bg_lptr)(ring + st * UNITS * 64);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            __builtin_amdgcn_global_load_lds((bg_gptr)wl[m], dst + m * 64, 16, 0, 0);
            wl[m] += 256;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            __builtin_amdgcn_global_load_lds((bg_gptr)(xp + (size_t)(16 * nt) * xwidth), dst + (MTW + nt) * 64, 16, 0, 0);
        xp += BG_KC;
        if (--left == 0 && pi < 3) {                       // wave-uniform, a few times per launch
            ++pi;
            xwidth = a.x[pi].width;
            left = a.cend[pi] - a.cend[pi - 1];
            xp = a.x[pi].ptr + (size_t)item * xwidth + kq4;
        }
    };

    // two accumulator sets by chunk parity: the four MFMAs of a chunk chain on one accumulator, the next chunk's do not wait
    bg_f4 acc[2][MTW][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[h][m][nt] = bg_f4{0.f, 0.f, 0.f, 0.f};

    int ist = 0;                                          // stage the next issue goes to
#pragma unroll
    for (int i = 0; i < S - 1; ++i)
        if (i < cpw) { issue(ist); ist = ist == S - 1 ? 0 : ist + 1; }
    // the epilogue's own operands (bias, cell state, keep bytes) of this wave's FIRST (m-tile, item tile) pair are requested
    // now, behind the ring's first DMAs, so that their latency is the K loop's and not the tail's (they are younger than
    // those DMAs: the counted waits below are merely conservative until they have retired)
    float e_pre[4] = {0.f, 0.f, 0.f, 0.f};
    float e_c = 0.f;
    bool e_keep[4] = {false, false, false, false};
    {
        const int idx = wave;
        const int tile = tile0 + idx / NT, it = n0 + 16 * (idx % NT) + (lane & 15), j = lane >> 4;
        if (idx < MTW * NT && it < a.batch && tile < a.tiles && a.pmode != 1) {
            if constexpr (EPI == BG_EPI_CELL) {
                const int unit = 4 * tile + j;
#pragma unroll
                for (int g = 0; g < 4; ++g) e_pre[g] = a.bih[g * a.H + unit] + a.bhh[g * a.H + unit];
                e_c = a.c[(size_t)it * a.H + unit];
            } else if constexpr (EPI == BG_EPI_PROJ || EPI == BG_EPI_PRENET2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = BG_MT * tile + 4 * j + v;
                    if (r < a.rows) {
                        if constexpr (EPI == BG_EPI_PROJ) {
                            e_pre[v] = a.bias[r];
                            if (r > a.n_mel && a.keep) e_keep[v] = a.keep[(size_t)it * a.P + (r - a.n_mel - 1)] != 0;
                        } else {
                            e_keep[v] = a.keep[((size_t)a.batch + it) * a.P + r] != 0;
                        }
                    }
                }
            }
        }
    }

    // ... and a FINAL role's EARLY sums (<= 2 slots), which are just as old
    constexpr int NKP = (MTW * NT + WAVES - 1) / WAVES;
    bg_f4 ppre[NKP][2];
    if (a.pmode == 2) {
#pragma unroll
        for (int k = 0; k < NKP; ++k)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int idx = min(wave + k * WAVES, MTW * NT - 1);
                ppre[k][sl] = sl < a.npart ? reinterpret_cast<const bg_f4*>(a.part)[(((size_t)sl * (a.tiles + 4) + tile0 + idx / NT) * a.nt_total + ngrp * NT + idx % NT) * 64 + lane]
                                           : bg_f4{0.f, 0.f, 0.f, 0.f};
            }
    }
    // Software pipeline per wave: DMA S - 1 chunks ahead; the operands of chunk i + 1 are read from LDS into the other
    // register set while the MFMAs of chunk i run.
    bg_u4 opA[UNITS], opB[UNITS];
    auto read_ops = [&](bg_u4 (&op)[UNITS], int st) {
        const bg_u4* sp = ring + st * UNITS * 64 + lane;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) op[u] = sp[u * 64];
    };
    auto mfmas = [&](const bg_u4 (&op)[UNITS], bg_f4 (&ac)[MTW][NT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    ac[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(op[m][s]), __uint_as_float(op[MTW + nt][s]), ac[m][nt], 0, 0, 0);
    };
    if (cpw > 0) {
        if (cpw >= S - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * UNITS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        read_ops(opA, 0);
    }
    int rst = 1 % S;                                      // stage of chunk i + 1
    // one pipeline step: chunk i's operands are in `cur`; fetch chunk i + 1's into `nxt`
    auto step = [&](int i, bg_u4 (&cur)[UNITS], bg_u4 (&nxt)[UNITS], bg_f4 (&ac)[MTW][NT]) {
        if (i + S - 1 < cpw) { issue(ist); ist = ist == S - 1 ? 0 : ist + 1; }
        if (i + 1 < cpw) {
            if (i + S - 1 < cpw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * UNITS) : "memory");   // chunk i + 1 landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_ops(nxt, rst);
            rst = rst == S - 1 ? 0 : rst + 1;
        }
        mfmas(cur, ac);
    };
    for (int i = 0; i < cpw; i += 2) {
        step(i, opA, opB, acc[0]);
        if (i + 1 < cpw) step(i + 1, opB, opA, acc[1]);
    }

    // cross-wave sum of the K slices, fixed order (pairwise for four waves: (w0 + w1) + (w2 + w3); eight: two such halves)
    __syncthreads();
    bg_f4* red = reinterpret_cast<bg_f4*>(lds);
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[((wave * MTW + m) * NT + nt) * 64 + lane] = acc[0][m][nt] + acc[1][m][nt];
    __syncthreads();
    constexpr int NK = (MTW * NT + WAVES - 1) / WAVES;     // (m-tile, item tile) pairs per wave
    bg_f4 sums[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int idx = wave + k * WAVES;
        const int m = min(idx, MTW * NT - 1) / NT, nt = min(idx, MTW * NT - 1) % NT;
        bg_f4 pw[WAVES];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) pw[w] = red[((w * MTW + m) * NT + nt) * 64 + lane];
#pragma unroll
        for (int w = 0; w < WAVES; w += 2) pw[w] += pw[w + 1];
#pragma unroll
        for (int w = 0; w < WAVES; w += 4) pw[w] += pw[w + 2];
        sums[k] = pw[0];
        if constexpr (WAVES == 8) sums[k] += pw[4];
    }
    if (a.pmode) {
        auto paddr = [&](int slot, int idx) -> bg_f4* {
            const int m = idx / NT, nt = idx % NT;
            return reinterpret_cast<bg_f4*>(a.part) + (((size_t)slot * (a.tiles + 4) + tile0 + m) * a.nt_total + ngrp * NT + nt) * 64 + lane;
        };
        if (a.pmode == 1) {                                // EARLY: this launch's share of the sums, for the FINAL launch
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (wave + k * WAVES < MTW * NT) *paddr(a.pslot, wave + k * WAVES) = sums[k];
            return;
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) sums[k] = (sums[k] + ppre[k][0]) + ppre[k][1];      // FINAL: own + slot 0 + slot 1 (requested before the K loop)
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int idx = wave + k * WAVES;
        if (idx >= MTW * NT) continue;
        const int m = idx / NT, nt = idx % NT;
        const bg_f4 sum = sums[k];
        const int tile = tile0 + m;
        const int it = n0 + 16 * nt + (lane & 15);
        const int j = lane >> 4;                           // rows 16 tile + 4 j + v, v = 0..3
        if (it >= a.batch || tile >= a.tiles) continue;
        const bool first = idx == wave;                    // operands prefetched above
        if constexpr (EPI == BG_EPI_CELL) {
            const int unit = 4 * tile + j, H = a.H;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = sum[g] + (first ? e_pre[g] : a.bih[g * H + unit] + a.bhh[g * H + unit]);
            const size_t ix = (size_t)it * H + unit;
            const float ig = bg_sigmoid(pre[0]), fg = bg_sigmoid(pre[1]), gg = tanhf(pre[2]), og = bg_sigmoid(pre[3]);
            const float cy = fg * (first ? e_c : a.c[ix]) + ig * gg;
            const float hy = og * tanhf(cy);
            a.c[ix] = cy;
            a.h_new[ix] = hy;
            if (a.hsum) a.hsum[ix] = hy + a.hres[ix];
        } else if constexpr (EPI == BG_EPI_SEQ) {
            const int unit = 4 * tile + j, H = a.H;
            const LstmSeq& sq = a.sq;
            const size_t ix = (size_t)it * H + unit;
            const int len = sq.lengths[it];
            if (sq.step < len) {
                const int tb = sq.reverse ? len - 1 - sq.step : sq.step;
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    pre[g] = sum[g] + sq.gadd[(size_t)it * sq.ga_bstride + (size_t)(g * H + unit) * sq.ga_ld + sq.ga_pad + tb];
                const float ig = bg_sigmoid(pre[0]), fg = bg_sigmoid(pre[1]), gg = tanhf(pre[2]), og = bg_sigmoid(pre[3]);
                const float cy = fg * a.c[ix] + ig * gg;
                const float hy = og * tanhf(cy);
                a.c[ix] = cy;
                a.h_new[ix] = hy;
                sq.out[(size_t)it * sq.out_bstride + (size_t)tb * sq.out_tstride + sq.out_col + unit] = hy;
                if (sq.step == len - 1) sq.hn[(size_t)it * sq.hn_stride + sq.hn_col + unit] = hy;
            } else {
                a.h_new[ix] = a.h_old[ix];
            }
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = BG_MT * tile + 4 * j + v;
                if (r >= a.rows) continue;
                if constexpr (EPI == BG_EPI_LINEAR) {
                    a.y[(size_t)it * a.ldy + r] = sum[v];
                } else if constexpr (EPI == BG_EPI_PROJ) {
                    const float val = sum[v] + (first ? e_pre[v] : a.bias[r]);
                    if (r < a.n_mel) a.mel_out[((size_t)it * a.n_mel + r) * a.max_steps + a.step] = val;
                    else if (r == a.n_mel) a.gate_out[(size_t)it * a.max_steps + a.step] = val;
                    else if (a.keep) {
                        const int jj = r - a.n_mel - 1;
                        const bool kp = first ? e_keep[v] : a.keep[(size_t)it * a.P + jj] != 0;
                        a.act_out[(size_t)it * a.P + jj] = kp ? fmaxf(val, 0.f) * 2.0f : 0.0f;
                    }
                } else {
                    const bool kp = first ? e_keep[v] : a.keep[((size_t)a.batch + it) * a.P + r] != 0;
                    a.act_out[(size_t)it * a.P + r] = kp ? fmaxf(sum[v], 0.f) * 2.0f : 0.0f;
                }
            }
        }
    }
}

// ---- windowed location-sensitive attention, split at the query ----------------------------------------------------------
// Part 1 (attn_pre_body, item b): everything that depends only on the previous step's weights and position - window start
// (model.py:131-140), location conv (:56-60) and location-dense term (:61-62), added to the processed-memory window:
//   apre[b][tt][a] = processed_memory[b][s + tt][a] + sum_(c, j) G[c][j][a] * wcat[c][s + tt - pad + j],  astart[b] = s
// with the two linear maps FOLDED at pack time, G[c][j][a] = sum_f Wd[f][a] * Wloc[f][c][j] (location_fold_kernel): one 62-tap
// filter per attention dim instead of 32 filters, an LDS round trip and a 32-term dense sum.  As conv + dense the stage was ~10 us
// of single-wave LDS latency per workgroup - two thirds of the pipelined step's launch 7 (profiles/r6_10).  Thread = (attention
// dim, run of NPOS consecutive window positions): its filter in 62 registers, the NPOS + 30 weights under the run read once per
// channel (a wave-wide broadcast), NPOS independent accumulators.
constexpr int BGA_W = 33, BGA_F = 32, BGA_K = 31, BGA_RUN = 9;
struct BgAttnLds {
    float wcat[2][BGA_W + BGA_K - 1 + 1];
};
template <int NPOS>
__device__ __forceinline__ void attn_pre_run(const AttnArgs& a, float* apre, const BgAttnLds& L, int b, int s, int W, int ad, int tt0,
                                             const float (&g)[2][BGA_K]) {
    float pmv[NPOS], acc[NPOS];
#pragma unroll
    for (int u = 0; u < NPOS; ++u) {
        const int pos = min(s + min(tt0 + u, W - 1), a.T - 1);
        pmv[u] = a.pm[((size_t)b * a.T + pos) * a.A + ad];
        acc[u] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < NPOS + BGA_K - 1; ++q) {           // weight tt0 + q of each channel feeds position u through tap q - u
        const int qi = min(tt0 + q, W + BGA_K - 2);
        const float w0 = L.wcat[0][qi], w1 = L.wcat[1][qi];
#pragma unroll
        for (int u = 0; u < NPOS; ++u)
            if (q - u >= 0 && q - u < BGA_K) acc[u] = fmaf(g[1][q - u], w1, fmaf(g[0][q - u], w0, acc[u]));
    }
#pragma unroll
    for (int u = 0; u < NPOS; ++u)
        if (tt0 + u < W) apre[((size_t)b * BGA_W + tt0 + u) * a.A + ad] = pmv[u] + acc[u];
}
__device__ __forceinline__ void attn_pre_body(const AttnArgs& a, float* apre, int* astart, BgAttnLds& L, int b) {
    const int t = threadIdx.x, nthr = blockDim.x;
    const int W = 2 * a.R + 1, padk = (a.K - 1) / 2;
    const int len = a.lengths[b];
    float cur = a.pos[b];
    const float off = a.scalars[0];
    // this thread's folded filter [2][K] of attention dim ad: does not depend on the window, requested first
    const int per = a.A <= 64 ? 64 : a.A <= 128 ? 128 : 256, ad = t % per, part = t / per, nparts = nthr / per;
    const bool live = ad < a.A;
    float g[2][BGA_K];
#pragma unroll
    for (int j = 0; j < BGA_K; ++j) {
        g[0][j] = (live && j < a.K) ? a.G[(size_t)j * a.A + ad] : 0.f;
        g[1][j] = (live && j < a.K) ? a.G[((size_t)a.K + j) * a.A + ad] : 0.f;
    }
    if (off != 0.f) cur += off;
    cur = fminf(fmaxf(cur, (float)a.R), (float)(len - 1 - a.R));
    const int s = (int)rintf(fmaxf(cur - (float)a.R, 0.f));
    for (int i = t; i < 2 * (BGA_W + BGA_K - 1); i += nthr) {
        const int c = i / (BGA_W + BGA_K - 1), j = i % (BGA_W + BGA_K - 1);
        const int pos = s - padk + j;
        const float* src = c == 0 ? a.w : a.cum;
        L.wcat[c][j] = (j < W + a.K - 1 && pos >= 0 && pos < a.T) ? src[(size_t)b * a.T + pos] : 0.f;
    }
    __syncthreads();
    if (live)                                              // runs of nine positions: 512 threads x 128 dims -> one run per thread, 256 x 128 -> two
        for (int r = part; BGA_RUN * r < W; r += nparts) attn_pre_run<BGA_RUN>(a, apre, L, b, s, W, ad, BGA_RUN * r, g);
    if (t == 0) astart[b] = s;
}

// Part 2 (one workgroup per item, between the query and the context): energies v . tanh(apre + q) (model.py:107-112), masked
// softmax over the window (:141-146), expected position, context, the step's alignment row.  The memory window goes
// straight into LDS by DMA while the energies are computed.
constexpr int BGA_DM = 512, BGA_A = 256;
constexpr int BGA_POST_LDS_BYTES = ((BGA_W * BGA_DM + 128) * 4 + 1023) / 1024 * 1024;
__device__ __forceinline__ void attn_post_body(const AttnArgs& a, const float* __restrict__ qbuf, const float* __restrict__ apre,
                                               const int* __restrict__ astart, int dbg, float* lds, int b) {
    float* memw = lds;                                     // [BGA_W][Dm] (16-byte aligned)
    float* en = memw + BGA_W * BGA_DM;                     // [64]
    float* wts = en + 64;                                  // [64] (16-byte aligned)
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int W = 2 * a.R + 1;
    const int s = astart[b];
    const int len = a.lengths[b];
    const float pos_old = a.pos[b], sf_raw = a.scalars[1];     // (used by one lane after the softmax: requested here)
    if (!(dbg & 8))
    {   // memory window -> LDS: row tt = 16-byte units [tt * Dm / 4, ...), one wave-instruction = 64 units = 256 floats
        const int upr = a.Dm / 4;                          // units per row (Dm % 256 == 0 is not required: units beyond are masked)
        const int total = W * upr;
        for (int u0 = wv * 64; u0 < total; u0 += 256) {
            const int u = min(u0 + lane, total - 1);
            const int tt = u / upr, d4 = u % upr;
            const int pos = min(s + tt, a.T - 1);
            __builtin_amdgcn_global_load_lds((bg_gptr)(a.memory + ((size_t)b * a.T + pos) * a.Dm + d4 * 4),
                                             (bg_lptr)(reinterpret_cast<bg_u4*>(memw) + u0), 16, 0, 0);
        }
    }
    // energies: wave wv takes window positions wv, wv + 4, ...; a lane sums its attention dims lane, lane + 64, ...
    constexpr int NE = (BGA_W + 3) / 4, NJ = BGA_A / 64;
    float ev[NE];
    {
        float qv[NJ], vv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ad = lane + 64 * j;
            qv[j] = ad < a.A ? qbuf[(size_t)b * a.A + ad] : 0.f;
            vv[j] = ad < a.A ? a.v[ad] : 0.f;
        }
        float x[NE][NJ];
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ad = lane + 64 * j, tt = min(wv + 4 * i, W - 1);
                x[i][j] = ad < a.A ? apre[((size_t)b * BGA_W + tt) * a.A + ad] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            float e = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)      // ~2-ulp tanh on exp2 / rcp (taco_math.h; libm's costs 4 us of this kernel's 12)
                if (lane + 64 * j < a.A) e = fmaf(vv[j], (dbg & 1) ? x[i][j] + qv[j] : tmath::acc_tanh(x[i][j] + qv[j]), e);
            ev[i] = e;
        }
        tmath::wave_totals<NE>(ev);                     // DPP network: no LDS-crossbar round trips
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int tt = wv + 4 * i;
            if (lane == 0 && tt < W) {
                const int pos = s + tt;
                en[tt] = (pos < len && pos < a.T) ? ev[i] : -INFINITY;
            }
        }
    }
    __syncthreads();
    if (wv == 0) {
        const float e = lane < W ? en[lane] : -INFINITY;
        const float m = tmath::wave_max(e);
        const float pexp = lane < W ? expf(e - m) : 0.f;
        float sums[2] = {pexp, 0.f};
        tmath::wave_totals<1>(reinterpret_cast<float (&)[1]>(sums[0]));
        const float wgt = pexp / sums[0];
        if (lane < W) wts[lane] = wgt;
        sums[1] = lane < W ? wgt * (float)(s + lane) : 0.f;
        tmath::wave_totals<1>(reinterpret_cast<float (&)[1]>(sums[1]));
        if (lane == 0) {
            const float sf = sigmoidf_(sf_raw);
            a.pos[b] = pos_old * sf + sums[1] * (1.0f - sf);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's share of the memory window has landed
    __syncthreads();
    if (!(dbg & 2))
    for (int d = 2 * t; d < a.Dm; d += 512) {              // dims d, d + 1: one 8-byte LDS read per window row
        float wreg[BGA_W + 3];
#pragma unroll
        for (int i = 0; i < (BGA_W + 3) / 4; ++i) {
            const float4 w4 = *reinterpret_cast<const float4*>(&wts[4 * i]);
            wreg[4 * i] = w4.x; wreg[4 * i + 1] = w4.y; wreg[4 * i + 2] = w4.z; wreg[4 * i + 3] = w4.w;
        }
        float c0[2] = {0.f, 0.f}, c1[2] = {0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < BGA_W; ++tt) {
            const float wv_ = (tt < W && s + tt < a.T) ? wreg[tt] : 0.f;
            const float2 mv = *reinterpret_cast<const float2*>(&memw[min(tt, W - 1) * a.Dm + d]);
            if (tt & 1) { c1[0] = fmaf(wv_, mv.x, c1[0]); c1[1] = fmaf(wv_, mv.y, c1[1]); }
            else { c0[0] = fmaf(wv_, mv.x, c0[0]); c0[1] = fmaf(wv_, mv.y, c0[1]); }
        }
        *reinterpret_cast<float2*>(&a.ctx[(size_t)b * a.Dm + d]) = make_float2(c0[0] + c1[0], c0[1] + c1[1]);
    }
    if (!(dbg & 4))
    for (int p = t; p < a.T; p += 256) {
        const float wgt = (p >= s && p < s + W) ? wts[p - s] : 0.f;
        a.w[(size_t)b * a.T + p] = wgt;
        a.cum[(size_t)b * a.T + p] += wgt;
        a.align_out[((size_t)b * a.max_steps + a.step) * a.T + p] = wgt;
    }
}

__global__ __launch_bounds__(256) void attn_post_kernel(const AttnArgs a, const float* __restrict__ qbuf, const float* __restrict__ apre,
                                                        const int* __restrict__ astart, int dbg) {
    extern __shared__ __attribute__((aligned(16))) bg_u4 bg_lds[];
    attn_post_body(a, qbuf, apre, astart, dbg, reinterpret_cast<float*>(bg_lds), blockIdx.x);
}

// ---- kernels and launch shapes ---------------------------------------------------------------------------------------------
template <int MTW, int NT, int S, int WAVES, int EPI>
__global__ __launch_bounds__(WAVES * 64) void bg_kernel(const BgArgs a) {
    extern __shared__ __attribute__((aligned(16))) bg_u4 bg_lds[];
    bg_body<MTW, NT, S, WAVES, EPI>(a, bg_lds, blockIdx.x, blockIdx.y);
}

// the attention RNN's launch: workgroups [0, nblk) are the cell's, [nblk, nblk + batch) (of grid row 0) the attention's part 1
template <int MTW, int NT, int S, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bg_cell_attn_kernel(const BgArgs a, int nblk, const AttnArgs at, float* apre, int* astart, int dbg) {
    extern __shared__ __attribute__((aligned(16))) bg_u4 bg_lds[];
    static_assert(sizeof(BgAttnLds) <= bg_lds_bytes<MTW, NT, S, WAVES>(), "the attention scratch shares the ring");
    if ((int)blockIdx.x < nblk) bg_body<MTW, NT, S, WAVES, BG_EPI_CELL>(a, bg_lds, blockIdx.x, blockIdx.y);
    else if (blockIdx.y == 0 && !(dbg & 32)) attn_pre_body(at, apre, astart, *reinterpret_cast<BgAttnLds*>(bg_lds), blockIdx.x - nblk);
}

// more than 64 KiB of dynamic LDS has to be allowed per kernel, once (per instantiation: the flag is a template static)
// both directions of a bidirectional layer advance in ONE launch per time step (blockIdx.z = direction)
template <int MTW, int NT, int S>
__global__ __launch_bounds__(256) void bg_seq2_kernel(const BgArgs a0, const BgArgs a1) {
    extern __shared__ __attribute__((aligned(16))) bg_u4 bg_lds[];
    if (blockIdx.z == 0) bg_body<MTW, NT, S, 4, BG_EPI_SEQ>(a0, bg_lds, blockIdx.x, blockIdx.y);
    else bg_body<MTW, NT, S, 4, BG_EPI_SEQ>(a1, bg_lds, blockIdx.x, blockIdx.y);
}

template <class K, K kernel>
int bg_allow_lds(int bytes) {
    static bool done = false;
    if (!done && bytes > 64 * 1024)
        CTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done = true;
    return CTTS_OK;
}
#define BG_ALLOW_LDS(kernel, bytes) bg_allow_lds<decltype(&kernel), &kernel>(bytes)

// Cells.  Shapes (measured, profiles/r6_03 ... r6_07).  DMA traffic of a cell GEMM = W x (B / 16) x (1 / NT + 1 / MTW): every
// workgroup re-reads the X columns of its items, so more m-tiles per workgroup cut it - but a CU sustains only ~45 GB/s with
// 100 KiB in flight, so a launch needs >= ~150 workgroups to pull at the fabric's 6.3-6.8 TB/s.
//   16 items: attention RNN MTW 2 (160 workgroups, 1.5 W), decoder RNNs MTW 1 (192, 2 W: as 96 they are per-CU bound)
//   32 items: MTW 2, NT 2 (2 W);  64 k items: MTW 2, NT 4 (3 W), grid.y = k
// attn != NULL: the attention's part 1 rides along (the attention RNN's launch).
#define BG_CELL_SHAPES(X) X(4, 1, 6, 4) X(2, 1, 8, 4) X(1, 1, 8, 4) X(2, 1, 6, 8) X(1, 1, 8, 8) \
    X(4, 2, 4, 4) X(2, 2, 5, 4) X(1, 2, 8, 4) X(2, 2, 4, 8) X(2, 4, 4, 4) X(2, 4, 3, 4) X(1, 4, 3, 4) X(2, 4, 3, 8)
inline void bg_cell_shape(int nb_pad, bool att, int shape, int& mtw, int& nt, int& st, int& wvs) {
    mtw = nb_pad <= 16 ? 1 : 2; nt = nb_pad <= 16 ? 1 : nb_pad <= 32 ? 2 : 4; st = nb_pad <= 16 ? 8 : nb_pad <= 32 ? 5 : 4; wvs = 4;
    if (att && nb_pad <= 16) mtw = 2;
    if (shape > 0) { wvs = shape >= 1000 ? 8 : 4; mtw = shape % 1000 / 100; st = shape % 100; }   // A/B knob: CTTS_TACO_BG_SHAPE = (1000: eight waves) + 100 MTW + S
}
inline int bg_launch_cell(const BgArgs& a, int nb_pad, const AttnArgs* attn, float* apre, int* astart, int batch, int shape, hipStream_t s) {
    int mtw, nt, st, wvs;
    bg_cell_shape(nb_pad, attn != nullptr, shape, mtw, nt, st, wvs);
    const int ny = nb_pad <= 32 ? 1 : nb_pad / 64;
    int rc = CTTS_E_ARG;
    bool found = false;
#define BG_TRY(M, N, SS, WV)                                                                                                     \
    if (!found && mtw == M && nt == N && st == SS && wvs == WV) {                                                                 \
        found = true;                                                                                                             \
        constexpr int LDS = bg_lds_bytes<M, N, SS, WV>();                                                                         \
        static_assert(LDS <= 160 * 1024, "LDS of a CU");                                                                          \
        const int nblk = (a.tiles + M - 1) / M;                                                                                   \
        if (attn) {                                                                                                               \
            if ((rc = BG_ALLOW_LDS((bg_cell_attn_kernel<M, N, SS, WV>), LDS))) return rc;                                         \
            hipLaunchKernelGGL((bg_cell_attn_kernel<M, N, SS, WV>), dim3(nblk + batch, ny), dim3(64 * WV), LDS, s, a, nblk, *attn, apre, astart, tuning().taco_bg_debug); \
        } else {                                                                                                                  \
            if ((rc = BG_ALLOW_LDS((bg_kernel<M, N, SS, WV, BG_EPI_CELL>), LDS))) return rc;                                      \
            hipLaunchKernelGGL((bg_kernel<M, N, SS, WV, BG_EPI_CELL>), dim3(nblk, ny), dim3(64 * WV), LDS, s, a);                 \
        }                                                                                                                         \
    }
    BG_CELL_SHAPES(BG_TRY)
#undef BG_TRY
    CTTS_CHECK_ARG(found, "batched decoder: cell shape MTW=%d NT=%d S=%d waves=%d is not instantiated", mtw, nt, st, wvs);
    CTTS_CHECK_LAUNCH("bg_cell");
    return CTTS_OK;
}

// ---- heterogeneous launches of the pipelined step ---------------------------------------------------------------------------
// A latency-bound stage (a small GEMM of 8-22 m-tiles; the attention's part 2) and the EARLY parts of the next cell GEMMs - the
// K columns whose inputs exist already - share one launch: independent roles by block range, no synchronisation between them
// (the cells' FINAL launches add the early sums from their `part` slots).  The streaming work of the step is spread over all
// seven launches instead of three.  What it buys (profiles/r6_08): 81 -> 76 us/step at 32 items, 114 -> 109 at 64, 65 -> 64 at
// 16 - the step is a chain of seven DEPENDENT launches (attention RNN -> query -> attention -> decoder RNN -> second decoder RNN
// -> projection -> prenet -> next step) and a launch of this design costs 5-6 us however little it does (a FINAL cell launch with
// 12 chunks per wave: 6.1 us), so re-distributing the streaming work cannot go below ~45 us; only fewer dependent launches could.
struct BgRoles {
    BgArgs small; int n_small, small_epi;             // blocks [0, n_small): tile = r % tiles, item tile = r / tiles
    BgArgs cell[2]; int n_cell[2], nblk_cell[2];      // then the cell roles: blk = r % nblk, item group = r / nblk
    const AttnArgs* pre; float* apre; int* astart; int n_pre;   // then the NEXT step's attention part 1, one block per item (pre != NULL)
};
// 512 threads: small GEMM (eight waves) + up to two EARLY cell roles of shape <M, N, SS, 8>
// (separate kernel parameters, not one struct holding an array of roles: the kernel reads X pieces by a run-time index, and a
//  dynamically indexed member of an array inside a by-value struct is copied to scratch - 1 KB per lane, 34 us per launch)
template <int M, int N, int SS>
__global__ __launch_bounds__(512) void bg_multi8_kernel(const BgArgs small, int n_small, int small_epi, const BgArgs c0, int n0, int nblk0,
                                                        const BgArgs c1, int n1, int nblk1, const AttnArgs pre, float* apre, int* astart) {
    extern __shared__ __attribute__((aligned(16))) bg_u4 bg_lds[];
    int r = blockIdx.x;
    if (r < n_small) {
        const int tile = r % small.tiles, ng = r / small.tiles;
        if (small_epi == BG_EPI_LINEAR) bg_body<1, 1, 8, 8, BG_EPI_LINEAR>(small, bg_lds, tile, ng);
        else if (small_epi == BG_EPI_PROJ) bg_body<1, 1, 8, 8, BG_EPI_PROJ>(small, bg_lds, tile, ng);
        else bg_body<1, 1, 8, 8, BG_EPI_PRENET2>(small, bg_lds, tile, ng);
        return;
    }
    r -= n_small;
    if (r < n0) { bg_body<M, N, SS, 8, BG_EPI_CELL>(c0, bg_lds, r % nblk0, r / nblk0); return; }
    r -= n0;
    if (r < n1) { bg_body<M, N, SS, 8, BG_EPI_CELL>(c1, bg_lds, r % nblk1, r / nblk1); return; }
    attn_pre_body(pre, apre, astart, *reinterpret_cast<BgAttnLds*>(bg_lds), r - n1);
}
// 256 threads: attention part 2 (one block per item) + up to two EARLY cell roles of shape <M, N, SS, 4>
template <int M, int N, int SS>
__global__ __launch_bounds__(256) void bg_post_multi4_kernel(const AttnArgs at, const float* qbuf, const float* apre, const int* astart, int dbg,
                                                             int n_post, const BgArgs c0, int n0, int nblk0, const BgArgs c1, int n1, int nblk1) {
    extern __shared__ __attribute__((aligned(16))) bg_u4 bg_lds[];
    int r = blockIdx.x;
    if (r < n_post) { attn_post_body(at, qbuf, apre, astart, dbg, reinterpret_cast<float*>(bg_lds), r); return; }
    r -= n_post;
    if (r < n0) { bg_body<M, N, SS, 4, BG_EPI_CELL>(c0, bg_lds, r % nblk0, r / nblk0); return; }
    r -= n0;
    if (r < n1) bg_body<M, N, SS, 4, BG_EPI_CELL>(c1, bg_lds, r % nblk1, r / nblk1);
}

// role shapes by padded batch: eight-wave EARLY roles (M, N, S) = 16: (2, 1, 6), 32: (2, 2, 4), 64: (2, 4, 3); four-wave: 16: (1, 1, 8),
// 32: (2, 2, 5), 64: (2, 4, 3)
inline int bg_launch_multi8(BgRoles& m, int nb_pad, hipStream_t s) {
    const int ny = nb_pad <= 32 ? 1 : nb_pad / 64;
    const int M = 2;
    {   // timing experiments (CTTS_TACO_BG_DEBUG; wrong results): 64 = without the EARLY roles, 128 = without the small stage, 256 = without attention part 1
        const int dbg = tuning().taco_bg_debug;
        if (dbg & 64) m.n_cell[0] = m.n_cell[1] = 0;
        if (dbg & 128) m.n_small = 0;
        if (dbg & 256) m.pre = nullptr;
    }
    for (int i = 0; i < 2; ++i) {
        m.nblk_cell[i] = m.n_cell[i] ? (m.cell[i].tiles + M - 1) / M : 1;
        m.n_cell[i] = m.n_cell[i] ? m.nblk_cell[i] * ny : 0;
    }
    const int blocks = m.n_small + m.n_cell[0] + m.n_cell[1] + (m.pre ? m.n_pre : 0);
    if (blocks == 0) return CTTS_OK;
    const AttnArgs none{};
    int rc;
#define BG_M8(MM, N, SS)                                                                                          \
    {                                                                                                             \
        constexpr int LDS = bg_lds_bytes<MM, N, SS, 8>() > bg_lds_bytes<1, 1, 8, 8>() ? bg_lds_bytes<MM, N, SS, 8>() : bg_lds_bytes<1, 1, 8, 8>(); \
        if ((rc = BG_ALLOW_LDS((bg_multi8_kernel<MM, N, SS>), LDS))) return rc;                                   \
        hipLaunchKernelGGL((bg_multi8_kernel<MM, N, SS>), dim3(blocks), dim3(512), LDS, s, m.small, m.n_small, m.small_epi, m.cell[0], \
                           m.n_cell[0], m.nblk_cell[0], m.cell[1], m.n_cell[1], m.nblk_cell[1], m.pre ? *m.pre : none, m.apre, m.astart); \
    }
    if (nb_pad <= 16) BG_M8(2, 1, 6)
    else if (nb_pad <= 32) BG_M8(2, 2, 4)
    else BG_M8(2, 4, 3)
#undef BG_M8
    CTTS_CHECK_LAUNCH("bg_multi8");
    return CTTS_OK;
}
inline int bg_launch_post_multi4(const AttnArgs& at, const float* qbuf, const float* apre, const int* astart, int batch, const BgArgs& c0,
                                 const BgArgs& c1, int nb_pad, hipStream_t s) {
    const int ny = nb_pad <= 32 ? 1 : nb_pad / 64;
    int rc;
#define BG_P4(MM, N, SS)                                                                                          \
    {                                                                                                             \
        constexpr int LDS = bg_lds_bytes<MM, N, SS, 4>() > BGA_POST_LDS_BYTES ? bg_lds_bytes<MM, N, SS, 4>() : BGA_POST_LDS_BYTES; \
        const int nb0 = (c0.tiles + MM - 1) / MM, nb1 = (c1.tiles + MM - 1) / MM;                                 \
        if ((rc = BG_ALLOW_LDS((bg_post_multi4_kernel<MM, N, SS>), LDS))) return rc;                              \
        hipLaunchKernelGGL((bg_post_multi4_kernel<MM, N, SS>), dim3(batch + (nb0 + nb1) * ny), dim3(256), LDS, s, at, qbuf, apre, astart, \
                           tuning().taco_bg_debug, batch, c0, nb0 * ny, nb0, c1, nb1 * ny, nb1);                  \
    }
    if (nb_pad <= 16) BG_P4(1, 1, 8)
    else if (nb_pad <= 32) BG_P4(2, 2, 5)
    else BG_P4(2, 4, 3)
#undef BG_P4
    CTTS_CHECK_LAUNCH("bg_post_multi4");
    return CTTS_OK;
}

// one time step of one (ndir = 1) or both directions of a packed-sequence LSTM over nb_pad items
inline int bg_launch_seq(const BgArgs& a0, const BgArgs& a1, int ndir, int nb_pad, hipStream_t s) {
    int rc;
#define BG_SEQ(M, N, SS)                                                                                              \
    {                                                                                                                 \
        constexpr int LDS = bg_lds_bytes<M, N, SS, 4>();                                                              \
        if ((rc = BG_ALLOW_LDS((bg_seq2_kernel<M, N, SS>), LDS))) return rc;                                          \
        hipLaunchKernelGGL((bg_seq2_kernel<M, N, SS>), dim3((a0.tiles + M - 1) / M, nb_pad <= 32 ? 1 : nb_pad / 64, ndir), \
                           dim3(256), LDS, s, a0, a1);                                                                \
    }
    if (nb_pad <= 16) BG_SEQ(1, 1, 8)
    else if (nb_pad <= 32) BG_SEQ(2, 2, 5)
    else BG_SEQ(2, 4, 4)
#undef BG_SEQ
    CTTS_CHECK_LAUNCH("bg_seq");
    return CTTS_OK;
}

// The small GEMMs (query rows, projection row set, second prenet layer: 8-22 m-tiles): one m-tile x 16 items per workgroup
// (grid.y = item tiles: their 0.7-2.9 MB of weights come from L2 however often they are read), K split over EIGHT waves,
// 8-deep ring - few workgroups, so latency per chunk is what they are made of.
template <int EPI>
int bg_launch_small(const BgArgs& a, int nb_pad, hipStream_t s) {
    constexpr int LDS = bg_lds_bytes<1, 1, 8, 8>();
    int rc;
    if ((rc = BG_ALLOW_LDS((bg_kernel<1, 1, 8, 8, EPI>), LDS))) return rc;
    hipLaunchKernelGGL((bg_kernel<1, 1, 8, 8, EPI>), dim3(a.tiles, (nb_pad + 15) / 16), dim3(512), LDS, s, a);
    CTTS_CHECK_LAUNCH("bg_small");
    return CTTS_OK;
}

inline void bg_set_x(BgArgs& a, const BgMat& m, const float* p0, int n0, const float* p1, int n1, const float* p2, int n2,
                     const float* p3, int n3) {
    const float* ps[4] = {p0, p1, p2, p3};
    const int ns[4] = {n0, n1, n2, n3};
    int c = 0;
    const float* last = p0;
    for (int i = 0; i < 4; ++i) {
        if (ps[i]) { c += ns[i] / BG_KC; last = ps[i]; }
        a.x[i] = BgPiece{ps[i] ? ps[i] : last, ps[i] ? ns[i] : 16};
        a.cend[i] = c;
    }
    a.cend[3] = m.nchunks;
    a.nchunks = m.nchunks;
    a.tiles = m.tiles;
}
