// MFMA + memory mix probe: how much f32 MFMA throughput survives (a) dependent accumulator chains, (b) concurrent
// global_load_dwordx4 streams (L2-resident or HBM), (c) ds_read_b128 streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NACC independent accumulators; LOADS_PER_32: global loads per 32 MFMAs; LDS_PER_32: ds_reads per 32 MFMAs
template <int NACC, int LOADS, int LDSR>
__global__ __launch_bounds__(256) void k_mix(const float4* src, long span_mask, float* dst, int iters) {
    __shared__ float4 lds[1024];
    const int t = blockIdx.x * 256 + threadIdx.x;
    lds[threadIdx.x] = src[t & 1023]; lds[threadIdx.x + 256] = src[(t + 256) & 1023]; lds[threadIdx.x + 512] = src[(t + 512) & 1023]; lds[threadIdx.x + 768] = src[(t + 768) & 1023];
    __syncthreads();
    float4 a = src[t & 0xFFFF], b = src[(t + 77) & 0xFFFF];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float4 ld[LOADS > 0 ? LOADS : 1], ls[LDSR > 0 ? LDSR : 1];
    long pos = (long)t;
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < LOADS; ++l) { ld[l] = src[(pos + (long)l * 65536 * 4) & span_mask]; }
#pragma unroll
        for (int l = 0; l < LDSR; ++l) { ls[l] = lds[(threadIdx.x + l * 64 + it) & 1023]; }
        pos += 262144 * 5;
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m & 3], bv[(m >> 2) & 3], acc[m % NACC], 0, 0, 0);
#pragma unroll
        for (int l = 0; l < LOADS; ++l) sink += ld[l].x;
#pragma unroll
        for (int l = 0; l < LDSR; ++l) sink += ls[l].y;
    }
    float s = sink;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    dst[t] = s;
}

template <int NACC, int LOADS, int LDSR>
void run(const char* name, const float4* src, long span_mask, float* dst, int wpc) {
    const int blocks = 256 * wpc / 4, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<NACC, LOADS, LDSR>), dim3(blocks), dim3(256), 0, 0, src, span_mask, dst, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = (double)blocks * 4 * iters * 32 * 2048.0;
    double bytes = (double)blocks * 4 * iters * LOADS * 1024.0;
    printf("%-44s waves/CU %2d: %.3f ms  %6.1f TFLOP/s  load %6.2f TB/s\n", name, wpc, best, flops / best / 1e9, bytes / best / 1e9);
}
int main() {
    float4* src; float* dst;
    const size_t n = (size_t)1 << 28;   // 4 GiB of float4? no: 2^28 float4 = 4 GiB
    hipMalloc(&src, n * 16 / 4);        // 1 GiB
    hipMemset(src, 0, n * 4);
    hipMalloc(&dst, 256 * 32 * 256 * 4);
    const long hbm_mask = (long)(n / 4) - 1;     // 1 GiB span
    const long l2_mask = (1L << 17) - 1;         // 2 MiB span
    for (int wpc : {16, 20}) {
        run<8, 0, 0>("8 acc, no mem", src, hbm_mask, dst, wpc);
        run<2, 0, 0>("2 acc (dependent every other)", src, hbm_mask, dst, wpc);
        run<1, 0, 0>("1 acc (back-to-back dependent)", src, hbm_mask, dst, wpc);
        run<8, 2, 0>("8 acc + 2 loads/32 MFMA, L2 span", src, l2_mask, dst, wpc);
        run<8, 6, 0>("8 acc + 6 loads/32 MFMA, L2 span", src, l2_mask, dst, wpc);
        run<8, 2, 0>("8 acc + 2 loads/32 MFMA, HBM span", src, hbm_mask, dst, wpc);
        run<8, 6, 0>("8 acc + 6 loads/32 MFMA, HBM span", src, hbm_mask, dst, wpc);
        run<8, 0, 8>("8 acc + 8 ds_read/32 MFMA", src, hbm_mask, dst, wpc);
        run<8, 1, 8>("8 acc + 1 load HBM + 8 ds_read", src, hbm_mask, dst, wpc);
    }
    return 0;
}
