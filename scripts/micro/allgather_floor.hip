// What does ONE granule all-gather between 256 co-resident workgroups cost on gfx950, with nothing else going on?
// The persistent Tacotron decoder pays seven of them per step (tacotron_persistent.hip); this measures the floor of the
// protocol itself: every workgroup publishes its share of G values as {tag, value} 8-byte granules (one agent-scope
// relaxed store each), every workgroup polls all G granules (agent-scope relaxed loads, thread t owns t, t + 512, ...),
// workgroup barrier, next exchange on the other parity buffer.  Variants:
//   mode 0: serial polling (issue my NPT loads, wait, check, s_sleep 2, again)         = the product's gather
//   mode 1: two poll rounds in flight (the second issued half a round-trip after the first)
//   mode 2: 16-byte granules {tag, v0, v1, v2} (one third of the requests)
//   mode 3: 8-byte granules, all NPT polling loads of a round ISSUED BACK TO BACK before the first check (the product's
//           loop checks each load before issuing the next: hipcc keeps every relaxed atomic load inside its own
//           predicated block, so a round of NPT loads costs NPT dependent round trips)
// Also prints the time of the same loop with the publish but WITHOUT the gather (compute skeleton only).
//   hipcc --offload-arch=gfx950 -O3 allgather_floor.hip -o allgather_floor && ./allgather_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
constexpr int WG = 256, T = 512;

__device__ __forceinline__ void publish(u64* g, int idx, unsigned epoch, float v) {
    __hip_atomic_store((gu64*)g + idx, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NPT, int MODE>
__device__ __forceinline__ bool gather(const u64* g, int count, float* dst, unsigned epoch, int t) {
    unsigned done = 0;
    for (unsigned spins = 0; spins < 2000000u; ++spins) {
        bool ok = true;
        if (MODE == 3) {
            u64 x[NPT];
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int i = min(t + T * k, count - 1);
                asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(x[k]) : "v"(g + i) : "memory");
            }
#pragma unroll
            for (int k = 0; k < NPT; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[k])::"memory");
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int i = t + T * k;
                if (i < count && !((done >> k) & 1u)) {
                    if ((unsigned)(x[k] >> 32) == epoch) { dst[i] = __uint_as_float((unsigned)x[k]); done |= 1u << k; }
                    else ok = false;
                }
            }
        } else if (MODE == 1) {
            u64 xa[NPT], xb[NPT];
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int i = t + T * k;
                xa[k] = i < count ? __hip_atomic_load((const gu64*)g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
            __builtin_amdgcn_s_sleep(6);
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int i = t + T * k;
                xb[k] = i < count ? __hip_atomic_load((const gu64*)g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int i = t + T * k;
                if (i < count && !((done >> k) & 1u)) {
                    if ((unsigned)(xa[k] >> 32) == epoch) { dst[i] = __uint_as_float((unsigned)xa[k]); done |= 1u << k; }
                    else if ((unsigned)(xb[k] >> 32) == epoch) { dst[i] = __uint_as_float((unsigned)xb[k]); done |= 1u << k; }
                    else ok = false;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int i = t + T * k;
                if (i < count && !((done >> k) & 1u)) {
                    const u64 x = __hip_atomic_load((const gu64*)g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(x >> 32) == epoch) { dst[i] = __uint_as_float((unsigned)x); done |= 1u << k; }
                    else ok = false;
                }
            }
        }
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// 16-byte granules: {tag, v0, v1, v2}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NPT>
__device__ __forceinline__ bool gather16(const u32x4* g, int count, float* dst, unsigned epoch, int t) {
    unsigned done = 0;
    for (unsigned spins = 0; spins < 2000000u; ++spins) {
        bool ok = true;
        u32x4 x[NPT];
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int i = min(t + T * k, count - 1);
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(x[k]) : "v"(g + i) : "memory");
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[k])::"memory");
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int i = t + T * k;
            if (i < count && !((done >> k) & 1u)) {
                if (x[k][0] == epoch) {
                    dst[3 * i] = __uint_as_float(x[k][1]); dst[3 * i + 1] = __uint_as_float(x[k][2]); dst[3 * i + 2] = __uint_as_float(x[k][3]);
                    done |= 1u << k;
                } else ok = false;
            }
        }
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

template <int NPT, int MODE>
__global__ __launch_bounds__(T, 2) void loop_kernel(u64* xb, int G, int steps, int do_gather, u64* stamps, float* sink) {
    __shared__ float X[3 * 8192];
    const int t = threadIdx.x, wg = blockIdx.x;
    const int per = (G + WG - 1) / WG;          // values this workgroup publishes per exchange
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        const unsigned epoch = (unsigned)s + 1u;
        const int gper = (per + 2) / 3;         // 16-byte granules of this workgroup (mode 2)
        u64* buf = xb + (size_t)(s & 1) * (MODE == 2 ? 2 * WG * gper : G);
        if (t == 0 && wg == 0) stamps[s] = __builtin_amdgcn_s_memrealtime();
        if (MODE == 2) {
            if (t < gper) {
                const int gi = wg * gper + t;
                u32x4 v; v[0] = epoch; v[1] = __float_as_uint(acc + t); v[2] = v[1]; v[3] = v[1];
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(reinterpret_cast<u32x4*>(buf) + gi), "v"(v) : "memory");
            }
            if (do_gather) {
                if (!gather16<(NPT + 2) / 3>(reinterpret_cast<const u32x4*>(buf), WG * gper, X, epoch, t)) return;
            }
        } else {
            if (t < per && wg * per + t < G) publish(buf, wg * per + t, epoch, acc + t);
            if (do_gather) {
                if (!gather<NPT, MODE>(buf, G, X, epoch, t)) return;
            }
        }
        __syncthreads();
        acc += X[(t * 7 + s) % 1024];
    }
    if (t == 0 && wg == 0) stamps[steps] = __builtin_amdgcn_s_memrealtime();
    sink[wg * T + t] = acc;
}

template <int NPT, int MODE>
static void run(const char* name, int G, u64* xb, u64* stamps, float* sink) {
    const int steps = 2000;
    for (int g = 1; g >= 0; --g) {
        hipMemset(xb, 0, 4 * 8192 * sizeof(u64));
        hipLaunchKernelGGL((loop_kernel<NPT, MODE>), dim3(WG), dim3(T), 0, 0, xb, G, steps, g, stamps, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
        std::vector<u64> h(steps + 1);
        hipMemcpy(h.data(), stamps, (steps + 1) * sizeof(u64), hipMemcpyDeviceToHost);
        const double us = (double)(h[steps] - h[200]) / 100.0 / (steps - 200);      // 100 MHz counter, skip the warm-up
        printf("%-44s G=%5d  %s: %.2f us per exchange\n", name, G, g ? "publish + gather" : "publish only    ", us);
    }
}

int main() {
    u64 *xb, *stamps; float* sink;
    hipMalloc(&xb, 4 * 8192 * sizeof(u64)); hipMalloc(&stamps, 4096 * sizeof(u64)); hipMalloc(&sink, WG * T * 4);
    run<10, 0>("serial polling, 8-byte granules", 5120, xb, stamps, sink);      // att_h
    run<4, 0>("serial polling, 8-byte granules", 2048, xb, stamps, sink);       // ctx
    run<2, 0>("serial polling, 8-byte granules", 1024, xb, stamps, sink);       // prenet
    run<10, 1>("two rounds in flight, 8-byte granules", 5120, xb, stamps, sink);
    run<4, 1>("two rounds in flight, 8-byte granules", 2048, xb, stamps, sink);
    run<2, 1>("two rounds in flight, 8-byte granules", 1024, xb, stamps, sink);
    run<10, 3>("round issued back to back, 8-byte granules", 5120, xb, stamps, sink);
    run<6, 3>("round issued back to back, 8-byte granules", 3072, xb, stamps, sink);
    run<4, 3>("round issued back to back, 8-byte granules", 2048, xb, stamps, sink);
    run<2, 3>("round issued back to back, 8-byte granules", 1024, xb, stamps, sink);
    run<6, 0>("serial polling, 8-byte granules", 3072, xb, stamps, sink);
    run<10, 2>("serial polling, 16-byte granules", 5120, xb, stamps, sink);
    run<4, 2>("serial polling, 16-byte granules", 2048, xb, stamps, sink);
    return 0;
}
