// pc_probe3.hip -- round 5: VALU instruction forms under a saturated MFMA stream of the other wave of the SIMD (continuation of pc_probe.hip).
// Waves 0-3: v_mfma_f32_16x16x32_bf16 back to back.  Waves 4-7: 64 FMAs per iteration over 16 independent sums with THREE distinct register operands
// (the probe of pc_probe.hip used one register twice), as v_pk_fma_f32 / v_fma_f32 / v_pk_fma + op_sel forms, alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int VMODE>
__global__ __launch_bounds__(512, 1) void probe(long long* out, int iters, int roles, float seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (!(roles & 1)) return;
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
        const unsigned u = __float_as_uint(seed) + lane;
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 au = {u, u + 1, u + 2, u + 3}, bu = {u + 4, u + 5, u + 6, u + 7};
        bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            asm volatile("" ::: "memory");
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.f) out[4000] = 1;
    } else {
        if (!(roles & 2)) return;
        f2 s[16], x[8], w[8];
        for (int i = 0; i < 16; ++i) s[i] = (f2){seed + i, seed};
        for (int i = 0; i < 8; ++i) { x[i] = (f2){seed * i, seed + lane}; w[i] = (f2){seed * 0.5f + i, seed * 0.25f}; }
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (VMODE == 0) s[i] = __builtin_elementwise_fma(x[(i + rep) & 7], w[(i * 3 + rep) & 7], s[i]);
                    else { s[i][0] = __builtin_fmaf(x[(i + rep) & 7][0], w[(i * 3 + rep) & 7][0], s[i][0]); s[i][1] = __builtin_fmaf(x[(i + rep) & 7][1], w[(i * 3 + rep) & 7][1], s[i][1]); }
                }
            asm volatile("" : "+v"(s[0]), "+v"(s[5]), "+v"(x[0]), "+v"(w[0]), "+v"(x[3]), "+v"(w[5])::"memory");
        }
        t1 = __builtin_readcyclecounter();
        float r = 0;
        for (int i = 0; i < 16; ++i) r += s[i][0] + s[i][1];
        if (r == 12345.f) out[4000] = 1;
    }
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}
template <int VMODE> void run(const char* name, int fma_lanes_per_iter) {
    long long* d; (void)hipMalloc(&d, 256 * 8 * 2 * 8 + 65536);
    const int iters = 2000;
    for (int roles = 1; roles <= 3; ++roles) {
        (void)hipMemset(d, 0, 256 * 8 * 2 * 8);
        probe<VMODE><<<256, 512>>>(d, iters, roles, 1.0f);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(256 * 8 * 2);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0; int nm = 0, nv = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) {
            const double dt = (double)(h[(b * 8 + w) * 2 + 1] - h[(b * 8 + w) * 2]);
            if (dt <= 0) continue;
            if (w < 4) { m += dt; ++nm; } else { v += dt; ++nv; }
        }
        printf("%-24s roles=%d  clk/mfma=%6.2f   clk per f32 FMA (64 lanes)=%6.2f\n", name, roles, nm ? m / nm / iters / 48 : 0.0, nv ? v / nv / iters / fma_lanes_per_iter : 0.0);
    }
    (void)hipFree(d);
}
int main() {
    run<0>("v_pk_fma_f32 3 operands", 128);
    run<1>("v_fma_f32 3 operands", 128);
    return 0;
}
