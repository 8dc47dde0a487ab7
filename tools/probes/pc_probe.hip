// pc_probe.hip -- round 5: what two waves of ONE SIMD cost each other when one issues bf16 16x16x32 MFMAs back to back (the pointwise
// phase of dsblock_cs) and the other VALU / LDS work (its depthwise phase).  8 waves per workgroup, one workgroup per CU: waves 0-3 take
// the matrix role, waves 4-7 the vector role (wave w lands on SIMD w % 4).  Prints clocks per instruction for each role alone and together.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pc_probe tools/probes/pc_probe.hip && /tmp/pc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// VMODE: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_fma_f32 + one ds_read_b128 per 10, 3 v_pk_fma + ds_read_b128 per 5, 4 ds_read_b128 only
template <int VMODE>
__global__ __launch_bounds__(512, 1) void probe(long long* out, int iters, int roles, float seed) {
    __shared__ float4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = make_float4(seed, 2.f, 3.f, 4.f);
    __syncthreads();
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
        if (s == 12345.f) out[1000] = 1;
    } else {
        if (!(roles & 2)) return;
        f2 s[16];
        for (int i = 0; i < 16; ++i) s[i] = (f2){seed + i, seed};
        const f2 w = {seed * 0.5f, seed * 0.25f};
        float4 x = make_float4(seed, 1.f, 2.f, 3.f);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            if constexpr (VMODE == 2 || VMODE == 3 || VMODE == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = lds[(lane + q * 64 + it) & 2047];
                    x.x += v.x; 
                    if constexpr (VMODE != 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (VMODE == 2) { s[q * 4 + i][0] = __builtin_fmaf(s[q * 4 + i][0], w[0], x.x); s[q * 4 + i][1] = __builtin_fmaf(s[q * 4 + i][1], w[1], x.y); }
                        else s[q * 4 + i] = __builtin_elementwise_fma(s[q * 4 + i], w, (f2){x.x, x.y});
                    }
                    }
                }
            }
#pragma unroll
            for (int rep = 0; rep < (VMODE >= 2 ? 0 : 3); ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (VMODE == 0) { s[i][0] = __builtin_fmaf(s[i][0], w[0], w[1]); }
                    else s[i] = __builtin_elementwise_fma(s[i], w, w);
                }
            asm volatile("" ::: "memory");
        }
        t1 = __builtin_readcyclecounter();
        float r = x.x;
        for (int i = 0; i < 16; ++i) r += s[i][0] + s[i][1];
        if (r == 12345.f) out[1000] = 1;
    }
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

template <int VMODE>
void run(const char* name, int per_iter_v) {
    long long* d; hipMalloc(&d, 256 * 8 * 2 * 8 + 16384);
    const int iters = 2000;
    for (int roles = 1; roles <= 3; ++roles) {
        hipMemset(d, 0, 256 * 8 * 2 * 8);
        probe<VMODE><<<256, 512>>>(d, iters, roles, 1.0f);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 8 * 2);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0; int nm = 0, nv = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) {
            const double dt = (double)(h[(b * 8 + w) * 2 + 1] - h[(b * 8 + w) * 2]);
            if (dt <= 0) continue;
            if (w < 4) { m += dt; ++nm; } else { v += dt; ++nv; }
        }
        printf("%-28s roles=%d  mfma clk/instr=%6.2f   vector clk/instr=%6.2f (per lds-read group when VMODE>=2)\n", name, roles,
               nm ? m / nm / iters / 48 : 0.0, nv ? v / nv / iters / per_iter_v : 0.0);
    }
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 48);
    run<1>("v_pk_fma_f32", 48);
    run<2>("4x(ds_read_b128+8 v_fma)", 4);
    run<3>("4x(ds_read_b128+4 v_pk_fma)", 4);
    run<4>("4x ds_read_b128", 4);
    return 0;
}
