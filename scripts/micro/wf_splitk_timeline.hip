// Where do the ~40 us of ONE fused WaveFlow layer at batch 1 go?  (config 4: 64 channels, 3 x 3 taps, K = 36 + 5 chunks,
// 226 workgroups of the split-K shape; the matrix pipe accounts for ~17 us: profiles/r3_24.)
// Compiles the product kernel (cookietts_amd/csrc/gemm_f32_small.hip) with CTTS_SMALL_GEMM_STAMPS, replays a chain of such
// layers and prints per-phase times over all blocks of the last launch:
//   0 entry | 1 tables built, first two pairs requested | 2 first pair landed | 3 main loop done | 4 K halves reduced, gated tile
//   published | 5 second GEMM done | 7 stores acknowledged
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCTTS_SMALL_GEMM_STAMPS -I include -I cookietts_amd/csrc \
//       scripts/micro/wf_splitk_timeline.hip -o /tmp/wf_splitk_timeline && /tmp/wf_splitk_timeline
// Variants of the main loop (timing only, results garbage): -DCTTS_EXP_NO_MFMA (no matrix work), -DCTTS_EXP_NO_DMA (stages
// never re-filled), -DCTTS_EXP_NO_LDSREAD (operands stay in registers) - these three act on the FOUR-wave tile (run with W4=1).
// Default: the eight-wave tile of round 5 (two waves per SIMD); W4=1 in the environment: the four-wave tile.
// profiles/r4_19_wf_splitk_timeline.txt, r5_21, r5_31.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../cookietts_amd/csrc/gemm_f32_small.hip"

namespace ctts {
// the symbols the kernel file takes from the rest of the library
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
Tuning tuning() { Tuning t{}; t.f32_splitk_w4 = getenv("W4") != nullptr; return t; }   // W4=1: the four-wave tile (the form before round 5)
void reload_tuning() {}
bool gemm_mode_is_split(int) { return false; }
int gemm_split_level(int) { return 0; }
void note_gemm_loop(int) {}
}  // namespace ctts

using namespace ctts;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    const int C = 64, L = 14400, PADC = 128, layers = 16, kmel = 80;
    const int ld = (L + 255) / 256 * 256 + 2 * PADC;
    const int nch_c = C / GEMM_KC, nch_in = 9 * nch_c + kmel / GEMM_KC;
    const size_t a_tile = (size_t)GEMM_KC * 128;
    float *X[4], *out, *mel, *Ain, *bias, *rsT, *rsb;
    for (auto& p : X) { CK(hipMalloc(&p, (size_t)C * ld * 4)); CK(hipMemset(p, 0, (size_t)C * ld * 4)); }
    CK(hipMalloc(&out, (size_t)C * ld * 4)); CK(hipMemset(out, 0, (size_t)C * ld * 4));
    CK(hipMalloc(&mel, (size_t)kmel * ld * 4)); CK(hipMemset(mel, 0, (size_t)kmel * ld * 4));
    CK(hipMalloc(&Ain, (size_t)layers * nch_in * a_tile * 4));
    CK(hipMalloc(&bias, 128 * 4)); CK(hipMemset(bias, 0, 128 * 4));
    CK(hipMalloc(&rsT, (size_t)64 * 128 * 4)); CK(hipMalloc(&rsb, 128 * 4)); CK(hipMemset(rsb, 0, 128 * 4));
    {
        std::vector<float> h((size_t)layers * nch_in * a_tile);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 2e-5f; }
        CK(hipMemcpy(Ain, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(rsT, h.data(), (size_t)64 * 128 * 4, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)C * ld);
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }
        for (auto& p : X) CK(hipMemcpy(p, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    }
    const int nt = (L + 63) / 64, blocks = nt;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, (size_t)blocks * 8 * 8));
    CK(hipMemset(stamps, 0, (size_t)blocks * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), &stamps, sizeof(stamps)));

    auto layer_args = [&](int layer) {
        GemmArgs a{};
        a.bm = 128; a.ld = ld; a.pad = PADC; a.L = L; a.ntiles = (L + 255) / 256; a.batch = 1; a.dst_ld = ld; a.dst_pad = PADC;
        a.A = Ain + (size_t)layer * nch_in * a_tile; a.bias = bias; a.a_nch_alloc = nch_in;
        a.MB = 1; a.M = 128; a.pairC = C;
        const int dw = 1 << (layer % 8);
        int ns = 0;
        for (int ah = 0; ah < 3; ++ah)
            for (int j = 0; j < 3; ++j) a.seg[ns++] = {X[ah], (long long)C * ld, nch_c, (j - 1) * dw, 0, 0};
        a.seg[ns++] = {mel, (long long)kmel * ld, kmel / GEMM_KC, 0, 0, 0};
        a.nseg = ns; a.nch_total = nch_in;
        a.addend_ld = ld; a.addend_pad = PADC;
        a.rs_wT = rsT; a.rs_bias = rsb; a.rs_rows = 128;
        a.dst0 = X[3]; a.dst0_bstride = (long long)C * ld; a.acc0 = 1; a.src0 = X[2]; a.src0_bstride = (long long)C * ld;
        a.dst1 = out; a.dst1_bstride = (long long)C * ld; a.acc1 = 1; a.split = C;
        return a;
    };
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < layers; ++i)
            if (launch_gemm_f32_small(GEMM_EPI_GATE_RS, layer_args(i), st)) return 1;
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %d fused layers, %.1f us per layer (%d workgroups of %s waves)\n", rep, layers, ms * 1000 / layers, blocks, getenv("W4") ? "four" : "eight");
    }
    std::vector<unsigned long long> h((size_t)blocks * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, tend = 0;
    for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, h[i * 8]); tend = std::max(tend, h[i * 8 + 7]); }
    printf("last launch: first entry -> last store ack %.2f us\n", (tend - t0) / 100.0);
    const int slot[7] = {0, 1, 2, 3, 4, 5, 7};
#ifdef CTTS_PROLOGUE_STAMPS   /* eight-wave tile only: slots 1-5 inside the prologue */
    const char* phase[7] = {"entry skew", "table entry math + A requests", "epilogue operand requests", "table written + barrier", "B requests issued", "first pair lands + barrier", "everything else"};
#else
    const char* phase[7] = {"entry skew", "tables + requests", "first pair lands", "main loop (18 pairs)", "reduce + gate", "second GEMM", "stores + ack"};
#endif
    for (int k = 0; k < 7; ++k) {
        std::vector<double> v;
        for (int i = 0; i < blocks; ++i) {
            const unsigned long long a = k == 0 ? t0 : h[i * 8 + slot[k - 1]], b = h[i * 8 + slot[k]];
            if (b >= a && b != 0) v.push_back((b - a) / 100.0);
        }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        double sum = 0; for (double x : v) sum += x;
        printf("    %-22s mean %6.2f  min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", phase[k], sum / v.size(), v.front(),
               v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    }
#ifdef CTTS_CLOCK_STAMPS
    {
        double cyc = 0, us = 0;
        for (int i = 0; i < blocks; ++i) { cyc += (double)h[i * 8 + 6]; us += (h[i * 8 + 3] - h[i * 8 + 2]) / 100.0; }
        printf("main loop: %.0f shader-clock cycles in %.2f us = %.3f GHz (mean over workgroups)\n", cyc / blocks, us / blocks, cyc / us / 1000.0);
    }
#endif
    return 0;
}
