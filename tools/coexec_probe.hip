// coexec_probe.hip -- does a SIMD overlap one wave's MFMAs with another wave's VALU / LDS instructions?  (hipcc --offload-arch=gfx950 -O3)
// 8 waves per workgroup = 2 per SIMD (waves w and w + 4 share a SIMD); role A = waves 0..3, role B = waves 4..7.
// Each configuration is timed with only A active, only B active, and both: both ~ max(A, B) means the pipes overlap, both ~ A + B means they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>   // 0 f32 mfma 16x16x4, 1 bf16 mfma 16x16x32, 2 v_fma_f32, 3 v_pk_fma_f32, 4 ds_read_b128, 5 v_fma_f32 dependent chain x4 interleaved
__device__ __forceinline__ float work(int iters, float seed, const float4* lds) {
    float out = 0.f;
    if constexpr (KIND == 0) {
        f32x4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a3, 0, 0, 0);
            }
        }
        out = a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (KIND == 1) {
        f32x4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        bf16x8 x; for (int e = 0; e < 8; ++e) x[e] = (__bf16)seed;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a3, 0, 0, 0);
            }
        }
        out = a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (KIND == 2) {
        float a[8]; for (int e = 0; e < 8; ++e) a[e] = seed + e;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = __builtin_fmaf(a[e], 0.999f, seed);
        }
        for (int e = 0; e < 8; ++e) out += a[e];
    } else if constexpr (KIND == 3) {
        f2 a[8]; for (int e = 0; e < 8; ++e) a[e] = (f2){seed + e, seed};
        const f2 m = {0.999f, 0.998f}, c = {seed, seed};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = __builtin_elementwise_fma(a[e], m, c);
        }
        for (int e = 0; e < 8; ++e) out += a[e][0] + a[e][1];
    } else if constexpr (KIND == 4) {
        float4 s = make_float4(0, 0, 0, 0);
        const float4* q = lds + (threadIdx.x & 63);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { const float4 v = q[u * 64]; s.x += v.x; }
            asm volatile("" : "+v"(q));
        }
        out = s.x;
    }
    return out;
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void probe(float* out, int ia, int ib, int mode) {
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) { if (mode & 1) r = work<KA>(ia, 1.0f + threadIdx.x * 1e-9f, lds); }
    else { if (mode & 2) r = work<KB>(ib, 1.0f + threadIdx.x * 1e-9f, lds); }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KA, int KB>
void run(const char* name, int ia, int ib) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL((probe<KA, KB>), dim3(256), dim3(512), 0, 0, out, ia, ib, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KA, KB>), dim3(256), dim3(512), 0, 0, out, ia, ib, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t[mode], e0, e1);
    }
    printf("%-34s A alone %7.1f us  B alone %7.1f us  both %7.1f us  (max %7.1f, sum %7.1f)\n", name, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[1] > t[2] ? t[1] * 1e3 : t[2] * 1e3, (t[1] + t[2]) * 1e3);
    hipFree(out);
}
int main() {
    run<0, 2>("f32 mfma 16x16x4   | v_fma_f32", 2000, 8000);
    run<0, 3>("f32 mfma 16x16x4   | v_pk_fma_f32", 2000, 4000);
    run<1, 2>("bf16 mfma 16x16x32 | v_fma_f32", 4000, 8000);
    run<1, 3>("bf16 mfma 16x16x32 | v_pk_fma_f32", 4000, 4000);
    run<0, 4>("f32 mfma 16x16x4   | ds_read_b128", 2000, 2000);
    run<1, 4>("bf16 mfma 16x16x32 | ds_read_b128", 4000, 2000);
    run<2, 4>("v_fma_f32          | ds_read_b128", 8000, 2000);
    run<2, 2>("v_fma_f32          | v_fma_f32", 8000, 8000);
    run<0, 0>("f32 mfma           | f32 mfma", 2000, 2000);
    run<1, 1>("bf16 mfma          | bf16 mfma", 4000, 4000);
    run<0, 1>("f32 mfma           | bf16 mfma", 2000, 4000);
    return 0;
}
