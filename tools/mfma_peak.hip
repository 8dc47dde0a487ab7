// MFMA peak probe: f32 16x16x4 and bf16 16x16x32 issue rate with zero vs random operands, plus core clock estimate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_f32(const float* src, float* dst, int iters, long long* clk) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xFFFFF]; b[i] = src[(t * 8 + 4 + i) & 0xFFFFF]; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[(i + j) & 3], acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    dst[t] = s;
    if (t == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
__global__ __launch_bounds__(256) void k_bf16(const float* src, float* dst, int iters, long long* clk) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[2], b[2];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 8; ++i) { a[j][i] = (__bf16)src[(t * 32 + j * 8 + i) & 0xFFFFF]; b[j][i] = (__bf16)src[(t * 32 + 16 + j * 8 + i) & 0xFFFFF]; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 1], b[(i + j) & 1], acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    dst[t] = s;
    if (t == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
    const int blocks = 256 * 4, iters = 4000;
    std::vector<float> h(1 << 20);
    float *src, *dst; long long* clk;
    hipMalloc(&src, 4 << 20); hipMalloc(&dst, blocks * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& v : h) v = mode ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f;
        hipMemcpy(src, h.data(), 4 << 20, hipMemcpyHostToDevice);
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(256), 0, 0, src, dst, iters, clk);
                else hipLaunchKernelGGL(k_bf16, dim3(blocks), dim3(256), 0, 0, src, dst, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
                double mfmas = (double)blocks * 4 * iters * 32;
                double flops = mfmas * (kind == 0 ? 2048.0 : 16384.0);
                if (rep == 2) printf("%s %s: %.3f ms  %.1f TFLOP/s   clock64 ticks %lld wall ticks %lld (ratio %.3f)\n", kind == 0 ? "f32 16x16x4 " : "bf16 16x16x32", mode ? "random" : "zeros ", ms, flops / ms / 1e9, c[0], c[1], (double)c[0] / (double)c[1]);
            }
        }
    }
    return 0;
}
