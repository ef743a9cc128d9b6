// The headline launch alone: WaveGlow config 2 in-layer GEMM (M = 1024, K = 3 x 512 + 256 = 112 chunks, 8 x 28 800 columns,
// GATE epilogue), the product kernel of cookietts_amd/csrc/gemm_f32.hip compiled into this file, timed with HIP events
// over 24 launches on rotating weights - the product loop, without the rest of the model around it.  Round 3 used it with
// temporary patches of the main loop to bound what each ingredient costs (profiles/r3_15_headline_gemm_experiments.txt):
// per launch 6.20 ms as shipped (6.28 with the hand-scheduled variant the other figures build on); LDS-DMA issue removed
// 5.86; fragment reads removed 6.26; barrier removed 6.33; all three removed 5.78 ms (0.93 of the fp32 MFMA peak).  A DMA instruction costs the same with 4-byte lanes, with one M0
// value for all pieces, as buffer_load ... lds, and without the vmcnt wait: it is the ISSUE of the six LDS-DMA
// instructions per wave and chunk (~55 matrix-pipe cycles each) that the loop pays, not their data path.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I cookietts_amd/csrc scripts/micro/headline_gemm.hip -o /tmp/headline_gemm
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../cookietts_amd/csrc/gemm_f32.hip"

namespace ctts {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
bool gemm_f32_small_applies(int, const GemmArgs&) { return false; }
int launch_gemm_f32_small(int, const GemmArgs&, hipStream_t) { return CTTS_E_ARG; }
int wf_row_cus() { return 256; }
}  // namespace ctts
using namespace ctts;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const char* label = argc > 1 ? argv[1] : "product";
    setenv("CTTS_F32_NO_ROUND_SPLIT", "1", 1);           // this file times ONE large-shape launch (the small shape is not linked in)
    const int C = 512, CC = 256, B = 8, PADC = 128, layers = 8;
    const int L = getenv("HG_L") ? atoi(getenv("HG_L")) : 28800;      // (multiples of 128; 28 672 = exactly 14 rounds of 512 workgroups)
    const int ld = L + 2 * PADC;
    const int nch = 3 * C / GEMM_KC + CC / GEMM_KC;      // 112
    const size_t a_tile = (size_t)GEMM_KC * 256;
    float *x, *h, *act, *A, *bias;
    CK(hipMalloc(&x, (size_t)B * C * ld * 4)); CK(hipMalloc(&h, (size_t)B * CC * ld * 4)); CK(hipMalloc(&act, (size_t)B * C * ld * 4));
    CK(hipMalloc(&A, (size_t)layers * 4 * nch * a_tile * 4)); CK(hipMalloc(&bias, 1024 * 4));
    CK(hipMemset(bias, 0, 1024 * 4)); CK(hipMemset(act, 0, (size_t)B * C * ld * 4));
    {
        std::vector<float> w((size_t)layers * 4 * nch * a_tile);
        unsigned s = 12345u;
        for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 2e-5f; }
        CK(hipMemcpy(A, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)B * C * ld);
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(h, hx.data(), (size_t)B * CC * ld * 4, hipMemcpyHostToDevice));
    }
    auto args = [&](int layer) {
        GemmArgs a{};
        a.ld = ld; a.pad = PADC; a.L = L; a.ntiles = L / 128; a.batch = B; a.dst_ld = ld; a.dst_pad = PADC; a.bm = 256;
        a.A = A + (size_t)layer * 4 * nch * a_tile; a.bias = bias;
        a.nseg = 4; a.interleave = 3; a.nch_total = nch; a.MB = 4;
        const int dil = 1 << layer;
        for (int t = 0; t < 3; ++t) a.seg[t] = {x, (long long)C * ld, C / GEMM_KC, (t - 1) * dil, 0, 0};
        a.seg[3] = {h, (long long)CC * ld, CC / GEMM_KC, 0, 0, 0};
        a.dst0 = act; a.dst0_bstride = (long long)C * ld; a.M = 2 * C; a.pairC = C;
        a.gemm_mode = CTTS_GEMM_F32;
        return a;
    };
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) if (launch_gemm_f32(GEMM_EPI_GATE, args(i % layers), st)) return 1;
    CK(hipStreamSynchronize(st));
    const int reps = 24;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) if (launch_gemm_f32(GEMM_EPI_GATE, args(i % layers), st)) return 1;
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 1024 * (double)(nch * GEMM_KC) * (double)B * L;
    printf("L = %d: %d workgroups = %.3f rounds of 512;  ", L, 4 * (L / 128) * B, 4.0 * (L / 128) * B / 512.0);
    printf("%-34s %.4f ms per launch  %.1f TFLOP/s  %.4f of 157.3\n", label, ms / reps, flop / (ms / reps * 1e-3) / 1e12, flop / (ms / reps * 1e-3) / 157.3e12);
    return 0;
}
