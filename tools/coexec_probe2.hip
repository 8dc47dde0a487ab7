// coexec_probe2.hip -- ONE wave per SIMD: does a wave's own stream overlap an MFMA with the independent VALU / LDS instructions issued right behind it?
// stream = N x { 1 MFMA ; K fillers }.  Compared with the MFMA-only and the filler-only stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MK, int FK, int K, int MODE>   // MK: 0 f32 mfma, 1 bf16 mfma 16x16x32; FK: 0 v_fma_f32, 1 ds_read_b128; MODE: 1 mfma only, 2 fillers only, 3 both
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    const float seed = 1.0f + threadIdx.x * 1e-9f;
    f32x4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 x; for (int e = 0; e < 8; ++e) x[e] = (__bf16)seed;
    float v[8]; for (int e = 0; e < 8; ++e) v[e] = seed + e;
    float4 s = make_float4(0, 0, 0, 0);
    const float4* q = lds + (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE & 1) {
                f32x4& a = (u & 3) == 0 ? a0 : (u & 3) == 1 ? a1 : (u & 3) == 2 ? a2 : a3;
                if (MK == 0) a = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a, 0, 0, 0);
                else a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 2) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (FK == 0) v[(u * K + k) & 7] = __builtin_fmaf(v[(u * K + k) & 7], 0.999f, seed);
                    else { const float4 t = q[((u * K + k) & 15) * 64]; s.x += t.x; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(q));
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3] + s.x;
    for (int e = 0; e < 8; ++e) r += v[e];
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int MK, int FK, int K>
void run(const char* name, int iters) {
    float* out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float t[4] = {0, 0, 0, 0};
    auto launch = [&](int mode) {
        if (mode == 1) hipLaunchKernelGGL((probe<MK, FK, K, 1>), dim3(256), dim3(256), 0, 0, out, iters);
        else if (mode == 2) hipLaunchKernelGGL((probe<MK, FK, K, 2>), dim3(256), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL((probe<MK, FK, K, 3>), dim3(256), dim3(256), 0, 0, out, iters);
    };
    for (int mode = 1; mode <= 3; ++mode) {
        launch(mode);
        (void)hipEventRecord(e0); launch(mode); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&t[mode], e0, e1);
    }
    printf("%-44s mfma %7.1f us  fillers %7.1f us  interleaved %7.1f us  (max %7.1f, sum %7.1f)\n", name, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[1] > t[2] ? t[1] * 1e3 : t[2] * 1e3, (t[1] + t[2]) * 1e3);
    (void)hipFree(out);
}
int main() {
    run<0, 0, 2>("f32 mfma + 2 v_fma each", 4000);
    run<0, 0, 6>("f32 mfma + 6 v_fma each", 4000);
    run<1, 0, 2>("bf16 mfma 16x16x32 + 2 v_fma each", 4000);
    run<1, 0, 3>("bf16 mfma 16x16x32 + 3 v_fma each", 4000);
    run<1, 0, 6>("bf16 mfma 16x16x32 + 6 v_fma each", 4000);
    run<0, 1, 2>("f32 mfma + 2 ds_read_b128 each", 4000);
    run<1, 1, 2>("bf16 mfma 16x16x32 + 2 ds_read_b128 each", 4000);
    return 0;
}
