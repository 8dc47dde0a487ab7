// lds_bw_probe.hip -- round 5: LDS read rate of one CU for ds_read_b128 / ds_read_b64 with W waves per workgroup (one workgroup per CU), conflict-free
// lane addresses (lane * 16) and the 16-lane broadcast pattern of a tap table (g * 16).    hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bw tools/probes/lds_bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
template <int MODE>   // 0: b128 lane*16, 1: b128 broadcast (lane>>4)*16, 2: b64 lane*8
__global__ __launch_bounds__(1024, 1) void probe(long long* out, int iters) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = MODE == 0 ? lane * 16 : MODE == 1 ? (lane >> 4) * 16 : lane * 8;
    addr += (wave & 3) * 4096;
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2) {
            u2 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[q]) : "v"(addr), "n"(0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 16; ++q) acc ^= v[q][0];
        } else {
            u4 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(v[q]) : "v"(addr + (unsigned)(q * 1024 % 4096)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 16; ++q) acc ^= v[q][0];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345) out[5000] = 1;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}
template <int MODE> void run(const char* name, int bytes_per_lane) {
    long long* d; (void)hipMalloc(&d, 1 << 20);
    for (int waves : {4, 8, 16}) {
        const int iters = 2000;
        (void)hipMemset(d, 0, 1 << 20);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(waves * 64), 65536, 0, d, iters);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(256 * 16);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) mx += (double)h[b * 16 + w];
        mx /= 256.0 * waves;
        printf("%-34s waves/CU=%2d  %.1f bytes / clock / CU\n", name, waves, (double)waves * 64 * bytes_per_lane * 16 * iters / mx);
    }
    (void)hipFree(d);
}
int main() {
    run<0>("ds_read_b128 lane*16", 16);
    run<1>("ds_read_b128 16-lane broadcast", 16);
    run<2>("ds_read_b64 lane*8", 8);
    return 0;
}
