// Operand layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, found by experiment: A = 1 in ONE lane, B = 1 in ONE lane,
// which D elements become 1?  Prints, for every (lane_a, lane_b) pair that produces a non-zero, the (vgpr, lane) of D.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_4x4x1_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(int la, int lb, float* out) {
    const int lane = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(lane == la ? 1.0f : 0.0f, lane == lb ? 1.0f : 0.0f, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[r * 64 + lane] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * sizeof(float));
    float h[256];
    for (int la = 0; la < 64; la += 1)
        for (int lb = 0; lb < 64; ++lb) {
            probe<<<1, 64>>>(la, lb, d);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; ++i)
                if (h[i] != 0.f && (la < 9 || la == 63)) printf("A lane %2d x B lane %2d -> D vgpr %d lane %2d\n", la, lb, i / 64, i % 64);
        }
    return 0;
}
