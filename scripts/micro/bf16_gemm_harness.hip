// Stand-alone harness of the PRODUCT bf16 conv-GEMM kernels (cookietts_amd/csrc/gemm_bf16_kernels.h, included as is) on
// config 3's three WN launch shapes with random operands: B = 32 utterances x 28 800 columns, C = 512, cond hidden 256.
//   in-layer  GATE   M = 1024, K = 3 taps x 512 (interleaved per 32-channel slab) + 256 = 56 chunks   (MFMA-bound)
//   res       SPLIT  M = 512,  K = 512 = 16 chunks, x += W act (read-modify-write)                     (HBM-bound)
//   skip      SPLIT  M = 512,  K = 4 x 512 = 64 chunks over four layers' activations, = / += into out  (HBM-bound)
// For each shape every kernel form is launched on the same arguments, its destination compared BIT FOR BIT with the skewed
// per-tile kernel's (conv_gemm_bf16_pp_kernel<EPI, 3>, the round-3 product), then timed.  No torch, no python: one
// gpurun minute per iteration.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o bin/bf16_gemm_harness bf16_gemm_harness.hip
#include "../../cookietts_amd/csrc/gemm_bf16_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace ctts;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned long long g_rng = 88172645463325252ull;
static double rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (double)(g_rng >> 11) / 9007199254740992.0; }
static float gauss() { double s = 0; for (int i = 0; i < 12; ++i) s += rnd(); return (float)(s - 6.0); }

// device buffer of n bf16 values ~ scale * N(0,1): a 32 MB random pattern tiled over the tensor
static bf16_t* random_bf16(size_t n, float scale) {
    bf16_t* d;
    CHECK(hipMalloc(&d, n * 2));
    std::vector<bf16_t> h(std::min(n, (size_t)1 << 24));
    const bool zeros = getenv("HARNESS_ZEROS") != nullptr;   // zero operands: no switching power, the clock stays at its maximum: cycles = structure
    for (auto& v : h) v = zeros ? (bf16_t)0 : f32_to_bf16_rne(scale * gauss());
    for (size_t off = 0; off < n; off += h.size())
        CHECK(hipMemcpy(d + off, h.data(), std::min(h.size(), n - off) * 2, hipMemcpyHostToDevice));
    return d;
}

struct Form { const char* name; int kind, ns; };   // kind 0 = pp (per tile), 1 = ps (persistent stream)

template <int EPI>
static void launch(const Form& f, BGemmArgs b, int cus) {
    const long long tiles = (long long)b.ntiles * b.batch;
    if (f.kind == 0) {
        const long long blocks = b.MB == 4 ? 16 * ((tiles + 3) / 4) : b.MB == 2 ? 16 * ((tiles + 7) / 8) : b.MB * tiles;
        hipLaunchKernelGGL((conv_gemm_bf16_pp_kernel<EPI, 3>), dim3((unsigned)blocks), dim3(512), 0, 0, b);
    } else if (f.kind == 2) {
        hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<EPI, 4, 1>), dim3(cus / 16 * 16), dim3(512), 0, 0, b);
    } else if (f.kind == 3) {
        hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<EPI, 3, 1>), dim3(cus / 16 * 16), dim3(512), 0, 0, b);
    } else if (f.ns == 3) {
        hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<EPI, 3>), dim3(cus / 16 * 16), dim3(512), 0, 0, b);
    } else {
        hipLaunchKernelGGL((conv_gemm_bf16_ps_kernel<EPI, 4>), dim3(cus / 16 * 16), dim3(512), 0, 0, b);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int B = 32, L = 28800, C = 512, H = 256, pad = 128;
    const int ntiles = (L + 255) / 256, ld = ntiles * 256 + 2 * pad;
    const long long cstride = (long long)C * ld, hstride = (long long)H * ld;
    const size_t xn = (size_t)B * cstride, hn = (size_t)B * hstride;
    printf("%s, %d CUs; B=%d L=%d ld=%d; tensors %.2f GB each\n", prop.name, cus, B, L, ld, xn * 2 / 1e9);

    bf16_t* x = random_bf16(xn, 1.0f);
    bf16_t* h = random_bf16(hn, 1.0f);
    bf16_t* act[4];
    for (auto& p : act) p = random_bf16(xn, 0.5f);
    bf16_t *dst, *ref, *init;
    CHECK(hipMalloc(&dst, xn * 2)); CHECK(hipMalloc(&ref, xn * 2));
    init = random_bf16(xn, 1.0f);
    const int nch_in = 3 * (C / 32) + H / 32, nch_rs = C / 32, nch_sk = 4 * (C / 32);
    bf16_t* A_in = random_bf16((size_t)4 * nch_in * 4 * 256 * 8, 0.02f);
    bf16_t* A_rs = random_bf16((size_t)2 * nch_rs * 4 * 256 * 8, 0.02f);
    bf16_t* A_sk = random_bf16((size_t)2 * nch_sk * 4 * 256 * 8, 0.02f);
    std::vector<float> hb(1024);
    for (auto& v : hb) v = 0.1f * gauss();
    float* bias;
    CHECK(hipMalloc(&bias, 4096)); CHECK(hipMemcpy(bias, hb.data(), 4096, hipMemcpyHostToDevice));

    auto base = [&]() { BGemmArgs a{}; a.ld = ld; a.pad = pad; a.L = L; a.ntiles = ntiles; a.batch = B; a.bias = bias; return a; };
    struct Case { const char* name; int epi; BGemmArgs a; double flop, bytes; bool rmw; };
    std::vector<Case> cases;
    {   // in-layer, dilation 8
        BGemmArgs a = base();
        const int dil = 8;
        for (int tp = 0; tp < 3; ++tp) a.seg[tp] = BGemmSeg{x, cstride, C / 32, (tp - 1) * dil, 0};
        a.interleave = 3;
        a.seg[3] = BGemmSeg{h, hstride, H / 32, 0, 0};
        a.nseg = 4; a.nch_total = nch_in; a.MB = 4; a.M = 2 * C; a.pairC = C; a.A = A_in;
        a.dst0 = dst; a.dst0_bstride = cstride; a.map_mode = 1;
        cases.push_back({"in-layer GATE (K=1792)", BGEMM_EPI_GATE, a, 2.0 * 1024 * 1792 * (double)B * L, 0, false});
    }
    {   // res: x += W act
        BGemmArgs a = base();
        a.seg[0] = BGemmSeg{act[0], cstride, C / 32, 0, 0};
        a.nseg = 1; a.nch_total = nch_rs; a.MB = 2; a.M = C; a.split = C; a.A = A_rs;
        a.dst0 = dst; a.dst0_bstride = cstride; a.acc0 = 1; a.dst1 = dst; a.dst1_bstride = cstride; a.acc1 = 1; a.map_mode = 2;
        cases.push_back({"res SPLIT += (K=512)", BGEMM_EPI_SPLIT, a, 2.0 * 512 * 512 * (double)B * L, 3.0 * C * 2 * (double)B * L, true});
    }
    for (int accum = 0; accum < 2; ++accum) {   // skip over four layers
        BGemmArgs a = base();
        for (int j = 0; j < 4; ++j) a.seg[j] = BGemmSeg{act[j], cstride, C / 32, 0, 0};
        a.nseg = 4; a.nch_total = nch_sk; a.MB = 2; a.M = C; a.split = C; a.A = A_sk;
        a.dst0 = dst; a.dst0_bstride = cstride; a.acc0 = accum; a.dst1 = dst; a.dst1_bstride = cstride; a.acc1 = accum; a.map_mode = 2;
        cases.push_back({accum ? "skip SPLIT += (K=2048)" : "skip SPLIT = (K=2048)", BGEMM_EPI_SPLIT, a, 2.0 * 512 * 2048 * (double)B * L,
                         (accum ? 6.0 : 5.0) * C * 2 * (double)B * L, accum != 0});
    }
    const Form forms[] = {{"pp per tile, 3 stages (round 3)", 0, 3}, {"ps persistent stream, 4 stages", 1, 4}, {"ps persistent stream, 3 stages", 1, 3},
                          {"ps 4 stages with stamps", 2, 4}, {"ps 3 stages with stamps", 3, 3}};

    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<unsigned int> href(xn / 2), hdst(xn / 2);
    printf("%-26s %-34s %9s %9s %9s  %s\n", "launch", "kernel form", "ms", "TFLOP/s", "GB/s", "vs pp");
    for (const Case& c : cases) {
        bool have_ref = false;
        for (const Form& f : forms) {
            // one checked launch on a fresh destination
            CHECK(hipMemcpy(dst, init, xn * 2, hipMemcpyDeviceToDevice));
            if (c.epi == BGEMM_EPI_GATE) launch<BGEMM_EPI_GATE>(f, c.a, cus); else launch<BGEMM_EPI_SPLIT>(f, c.a, cus);
            CHECK(hipGetLastError()); CHECK(hipDeviceSynchronize());
            const char* verdict = "reference";
            static char msg[128];
            if (!have_ref) { CHECK(hipMemcpy(ref, dst, xn * 2, hipMemcpyDeviceToDevice)); have_ref = true; }
            else {
                CHECK(hipMemcpy(href.data(), ref, xn * 2, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(hdst.data(), dst, xn * 2, hipMemcpyDeviceToHost));
                size_t bad = 0, firstbad = 0;
                for (size_t i = 0; i < href.size(); ++i) if (href[i] != hdst[i]) { if (!bad) firstbad = i; ++bad; }
                if (bad) snprintf(msg, sizeof msg, "DIFFERENT: %zu of %zu dwords, first at %zu", bad, href.size(), firstbad);
                else snprintf(msg, sizeof msg, "bit-identical");
                verdict = msg;
            }
            float best = 1e30f, sum = 0;
            for (int r = 0; r < reps; ++r) {
                if (c.rmw && r % 4 == 0) CHECK(hipMemcpy(dst, init, xn * 2, hipMemcpyDeviceToDevice));   // keep the running sum finite
                CHECK(hipEventRecord(e0));
                if (c.epi == BGEMM_EPI_GATE) launch<BGEMM_EPI_GATE>(f, c.a, cus); else launch<BGEMM_EPI_SPLIT>(f, c.a, cus);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms); if (r >= reps / 2) sum += ms;
                if (getenv("HARNESS_VERBOSE")) printf("      rep %d: %.3f ms\n", r, ms);
            }
            const float mean = sum / (reps - reps / 2);   // second half of the reps: the clock has settled
            if (f.kind >= 2) {
                unsigned long long st[32];
                CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ps_stamps), sizeof st));
                for (int hlf = 0; hlf < 2; ++hlf)
                    printf("    stamps wave %d: tile (loop + epilogue) %llu cycles = %.1f per chunk (%.2f cyc/MFMA/SIMD), of which epilogue %llu cycles, clock %.0f MHz, %llu tiles per workgroup\n",
                           4 * hlf, st[4 * hlf], (double)st[4 * hlf] / c.a.nch_total, (double)st[4 * hlf] / c.a.nch_total / 32.0,
                           st[4 * hlf + 1], (double)st[4 * hlf] / ((double)st[4 * hlf + 2] / 100.0), st[4 * hlf + 3]);
            }
            printf("%-26s %-34s %9.3f %9.1f %9.1f  %s (best %.3f ms)\n", c.name, f.name, mean, c.flop / mean / 1e9,
                   c.bytes / mean / 1e6, verdict, best);
            fflush(stdout);
        }
    }
    return 0;
}
