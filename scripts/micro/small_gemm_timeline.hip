// Where do the microseconds of a SHORT conv-GEMM launch go?  (ax WaveGlow, notebook config, batch 1: in-layer GEMM 90 us,
// res/skip GEMM 36 us per launch, of which the matrix pipe accounts for 61 / 20 us: profiles/r3_07_*.)
// Compiles the product kernel (cookietts_amd/csrc/gemm_f32_small.hip) with CTTS_SMALL_GEMM_STAMPS, replays one WN layer
// of that config - GATE launch (M = 512, K = 3 x 256, interpolated addend) then SPLIT launch (M = 512, K = 256,
// read-modify-write of x and the skip sum) - on cold weights, and prints per-phase times over all blocks of a launch:
//   entry skew (block start - first block start), tables, first chunk, main loop, epilogue operands, stores, and the
//   launch-to-launch period seen by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCTTS_SMALL_GEMM_STAMPS -I include -I cookietts_amd/csrc \
//       scripts/micro/small_gemm_timeline.hip -o /tmp/small_gemm_timeline && /tmp/small_gemm_timeline
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
Tuning tuning() { return Tuning{}; }
void reload_tuning() {}
bool gemm_mode_is_split(int) { return false; }
int gemm_split_level(int) { return 0; }
void note_gemm_loop(int) {}
}  // namespace ctts

using namespace ctts;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void report(const char* name, const std::vector<unsigned long long>& st, int blocks) {
    unsigned long long t0 = ~0ull, t5 = 0;
    for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, st[i * 8]); t5 = std::max(t5, st[i * 8 + 5]); }
    const char* phase[6] = {"entry skew", "tables", "first chunk", "main loop", "epi operands", "compute+stores"};
    printf("%s: %d blocks, first entry -> last store ack %.2f us\n", name, blocks, (t5 - t0) / 100.0);
    for (int k = 0; k < 6; ++k) {
        std::vector<double> v;
        for (int i = 0; i < blocks; ++i) {
            const unsigned long long a = k == 0 ? t0 : st[i * 8 + k - 1], b = st[i * 8 + k];
            if (b >= a && b != 0) v.push_back((b - a) / 100.0);
        }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        double sum = 0; for (double x : v) sum += x;
        printf("    %-16s mean %6.2f  min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", phase[k], sum / v.size(), v.front(),
               v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    }
    {   // where the blocks landed: blocks per CU (XCC, SE, SH, CU of HW_ID)
        std::vector<int> per(8 * 8 * 2 * 16, 0);
        for (int i = 0; i < blocks; ++i) {
            const unsigned long long w = st[i * 8 + 6];
            const unsigned hw = (unsigned)w, xcc = (unsigned)(w >> 32) & 7u;
            per[((xcc * 8 + ((hw >> 13) & 7u)) * 2 + ((hw >> 12) & 1u)) * 16 + ((hw >> 8) & 15u)]++;
        }
        int hist[9] = {0}, used = 0;
        for (int v : per) { if (v > 0) { used++; hist[v > 8 ? 8 : v]++; } }
        printf("    blocks per CU: %d CUs used;", used);
        for (int k = 1; k <= 8; ++k) if (hist[k]) printf("  %d CUs x %d", hist[k], k);
        printf("\n");
    }
    std::vector<double> endt;
    for (int i = 0; i < blocks; ++i) endt.push_back((st[i * 8 + 5] - t0) / 100.0);
    std::sort(endt.begin(), endt.end());
    printf("    block end times: min %.2f  median %.2f  p90 %.2f  max %.2f us\n", endt.front(), endt[blocks / 2], endt[blocks * 9 / 10], endt.back());
}

int main() {
    const int C = 256, L = 11700, F = 469, PADC = 128, layers = 16;
    const int ld = (L + 255) / 256 * 256 + 2 * PADC;
    const int nch_c = C / GEMM_KC, nch_in = 3 * nch_c;
    const size_t a_tile = (size_t)GEMM_KC * 256;
    float *x, *act, *out, *cond, *Ain, *Ars, *bias;
    CK(hipMalloc(&x, (size_t)C * ld * 4)); CK(hipMalloc(&act, (size_t)C * ld * 4)); CK(hipMalloc(&out, (size_t)C * ld * 4));
    CK(hipMalloc(&cond, (size_t)2 * C * 512 * 4));
    CK(hipMalloc(&Ain, (size_t)layers * 2 * nch_in * a_tile * 4)); CK(hipMalloc(&Ars, (size_t)layers * 2 * nch_c * a_tile * 4));
    CK(hipMalloc(&bias, 512 * 4));
    CK(hipMemset(x, 0, (size_t)C * ld * 4)); CK(hipMemset(act, 0, (size_t)C * ld * 4)); CK(hipMemset(out, 0, (size_t)C * ld * 4));
    CK(hipMemset(cond, 0, (size_t)2 * C * 512 * 4)); CK(hipMemset(bias, 0, 512 * 4));
    {   // small random weights so the pipe is not fed zeros
        std::vector<float> h((size_t)layers * 2 * nch_in * a_tile);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 2e-5f; }
        CK(hipMemcpy(Ain, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(Ars, h.data(), (size_t)layers * 2 * nch_c * a_tile * 4, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)C * ld);
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    }
    const int ntiles_s = (L + 63) / 64, blocks = 2 * 2 * ntiles_s;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, (size_t)blocks * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), &stamps, sizeof(stamps)));

    auto gate_args = [&](int layer) {
        GemmArgs a{};
        a.ld = ld; a.pad = PADC; a.L = L; a.ntiles = (L + 127) / 128; a.batch = 1; a.dst_ld = ld; a.dst_pad = PADC; a.bm = 256;
        a.A = Ain + (size_t)layer * 2 * nch_in * a_tile; a.bias = bias;
        a.nseg = 3; a.interleave = 3; a.nch_total = nch_in; a.MB = 2;
        const int dil = 1 << (layer % 8);
        for (int t = 0; t < 3; ++t) a.seg[t] = {x, (long long)C * ld, nch_c, (t - 1) * dil, 0, 0};
        a.dst0 = act; a.dst0_bstride = (long long)C * ld; a.M = 2 * C; a.pairC = C;
        a.addend = cond; a.addend_bstride = 0; a.addend_ld = 512; a.addend_pad = 0; a.addend_frames = F;
        return a;
    };
    auto split_args = [&](int layer) {
        GemmArgs a{};
        a.ld = ld; a.pad = PADC; a.L = L; a.ntiles = (L + 127) / 128; a.batch = 1; a.dst_ld = ld; a.dst_pad = PADC; a.bm = 256;
        a.A = Ars + (size_t)layer * 2 * nch_c * a_tile; a.bias = bias;
        a.nseg = 1; a.nch_total = nch_c; a.MB = 2;
        a.seg[0] = {act, (long long)C * ld, nch_c, 0, 0, 0};
        a.M = 2 * C; a.dst0 = x; a.dst0_bstride = (long long)C * ld; a.acc0 = 1;
        a.dst1 = out; a.dst1_bstride = (long long)C * ld; a.acc1 = 1; a.split = C;
        return a;
    };
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < layers; ++i) {
            if (launch_gemm_f32_small(GEMM_EPI_GATE, gate_args(i), st)) return 1;
            if (launch_gemm_f32_small(GEMM_EPI_SPLIT, split_args(i), st)) return 1;
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %d layers (GATE + SPLIT launch each) %.1f us per layer\n", rep, layers, ms * 1000 / layers);
    }
    std::vector<unsigned long long> h((size_t)blocks * 8);
    // the stamps left in the buffer are those of the last SPLIT launch; replay one GATE launch alone for its own
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    report("SPLIT (res/skip, K = 256, last launch of the chain)", h, blocks);
    CK(hipMemset(stamps, 0, h.size() * 8));
    if (launch_gemm_f32_small(GEMM_EPI_SPLIT, split_args(3), st)) return 1;     // something in front of it, like in the chain
    if (launch_gemm_f32_small(GEMM_EPI_GATE, gate_args(5), st)) return 1;
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    report("GATE (in-layer, K = 768, after a SPLIT launch)", h, blocks);
    return 0;
}
