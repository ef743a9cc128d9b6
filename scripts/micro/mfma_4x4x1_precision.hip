// Is v_mfma_f32_4x4x1_16b_f32 an exact fp32 fma per element?  D = A * B + C for random full-mantissa operands, compared
// with fmaf() and with the double-precision product: prints the worst relative deviations.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_4x4x1_precision.hip -o /tmp/mfma_prec && /tmp/mfma_prec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, const float* c, float* out, int n) {
    const int lane = threadIdx.x;
    for (int it = 0; it < n; ++it) {
        f4 cc = {c[(it * 4 + 0) * 64 + lane], c[(it * 4 + 1) * 64 + lane], c[(it * 4 + 2) * 64 + lane], c[(it * 4 + 3) * 64 + lane]};
        cc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[it * 64 + lane], b[it * 64 + lane], cc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(it * 4 + r) * 64 + lane] = cc[r];
    }
}
int main() {
    const int n = 256;
    float *a, *b, *c, *o;
    hipMallocManaged(&a, n * 64 * 4); hipMallocManaged(&b, n * 64 * 4); hipMallocManaged(&c, n * 256 * 4); hipMallocManaged(&o, n * 256 * 4);
    srand(1);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * 1.2345678f; };
    for (int i = 0; i < n * 64; ++i) { a[i] = rnd(); b[i] = rnd(); }
    for (int i = 0; i < n * 256; ++i) c[i] = rnd() * 0.01f;
    probe<<<1, 64>>>(a, b, c, o, n);
    hipDeviceSynchronize();
    double worst_fma = 0, worst_exact = 0, worst_mul_add = 0;
    for (int it = 0; it < n; ++it)
        for (int r = 0; r < 4; ++r)
            for (int lane = 0; lane < 64; ++lane) {
                const int blk = lane / 4, j = lane % 4;
                const float av = a[it * 64 + 4 * blk + r], bv = b[it * 64 + 4 * blk + j], cv = c[(it * 4 + r) * 64 + lane];
                const float got = o[(it * 4 + r) * 64 + lane];
                const double exact = (double)av * bv + cv;
                const float f = fmaf(av, bv, cv);
                const float ma = av * bv + cv;
                worst_fma = fmax(worst_fma, fabs(got - f) / fabs(exact));
                worst_mul_add = fmax(worst_mul_add, fabs((double)got - (double)ma) / fabs(exact));
                worst_exact = fmax(worst_exact, fabs(got - exact) / fabs(exact));
            }
    printf("v_mfma_f32_4x4x1: worst rel deviation from fmaf %.3e, from (a*b rounded)+c %.3e, from the exact value %.3e\n", worst_fma,
           worst_mul_add, worst_exact);
    return 0;
}
