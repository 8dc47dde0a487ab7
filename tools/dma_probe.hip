// dma_probe.hip -- hardware facts dsblock_rs.inc relies on, checked on the device (hipcc --offload-arch=gfx950 tools/dma_probe.hip -o tools/dma_probe):
//   1. buffer_load_dwordx4 ... lds reaches every LDS byte offset up to 160 KB through M0 (not only the first 64 KB);
//   2. lanes whose offset is beyond num_records get ZEROS written to their LDS slot (not skipped);
//   3. vmcnt(0) covers the DMA pieces.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, unsigned nbytes, float* out, const unsigned* dsts, int nd) {
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 16; i += 64) lds[i] = make_float4(-7.f, -7.f, -7.f, -7.f);
    __syncthreads();
    const unsigned long base = (unsigned long)src;
    u32x4 rsrc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32)) & 0xFFFFu, nbytes, 0x00020000u};
    for (int d = 0; d < nd; ++d) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dsts[d]);
        // lanes 0..47: offset lane * 16 (in range); lanes 48..55: 0x40000000 + lane * 16 (out of range); lanes 56..63: 0x40000000 + 0x40000000
        unsigned off = lane < 48 ? lane * 16u + d * 1024u : lane < 56 ? 0x40000000u + lane * 16u : 0x80000000u + lane * 16u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(rsrc), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int d = 0; d < nd; ++d) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + dsts[d] + lane * 16);
        reinterpret_cast<float4*>(out)[d * 64 + lane] = v;
    }
}
int main() {
    const int nd = 6;
    std::vector<unsigned> dsts = {0u, 60u * 1024, 70u * 1024, 100u * 1024, 130u * 1024, 159u * 1024};
    std::vector<float> h(nd * 256 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0f + (float)i;
    float *src, *out; unsigned* dd;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, nd * 64 * 16); hipMalloc(&dd, nd * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dd, dsts.data(), nd * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, src, (unsigned)(nd * 1024), out, dd, nd);
    hipError_t e = hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(e));
    std::vector<float> o(nd * 256);
    hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int d = 0; d < nd; ++d) {
        int ok_in = 0, ok_zero = 0;
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 4; ++k) {
                const float v = o[(d * 64 + l) * 4 + k];
                if (l < 48) ok_in += v == h[d * 256 + l * 4 + k];
                else ok_zero += v == 0.0f;
            }
        printf("dst %6u: in-range %d/192 correct, out-of-range %d/64 zero (first oob value %g)\n", dsts[d], ok_in, ok_zero, o[(d * 64 + 48) * 4]);
        bad += ok_in != 192 || ok_zero != 64;
    }
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad ? 1 : 0;
}
