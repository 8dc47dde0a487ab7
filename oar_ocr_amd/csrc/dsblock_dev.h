// dsblock_dev.h -- kernel parameter block + instantiation hooks of the fused depthwise-separable block.
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>

#include "igemm_dev.h"
#include "dsblock.h"

namespace oar {
namespace k {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct DsP {
    const float* x; float* y;
    const float* wd; const float* bd;
    const uint4* wp; const float* bp;
    const float* res; const float* se;
    int N, H, W, C, Ho, Wo, Cout;
    int sh, sw, pt, pl;
    int act1, act2;
    float a1, b1, a2, b2;
    int TR, TC, IR, IC;
    int tiles_x, tiles_y;
    long tiles;
    int KC, NF, y_ld;
    int dbg;   // dsblock_wa timing ablations (OAR_DSB_DBG; wrong results, for tools/dsblock_bench.py only)
};

// one translation unit per (kernel size, column stride): each instantiates the eight (NFW, PFW) wave layouts
void dsblock_launch_k3s1(hipStream_t s, const DsP& p, int nfw, int pfw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_launch_k3s2(hipStream_t s, const DsP& p, int nfw, int pfw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_launch_k5s1(hipStream_t s, const DsP& p, int nfw, int pfw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_launch_k5s2(hipStream_t s, const DsP& p, int nfw, int pfw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);

// wave-autonomous variant (dsblock_wa.inc; 3x3, stride 1): one translation unit per group of cout-fragment counts
constexpr int kWaRing = 2;       // tile buffers in LDS (the host sizes the allocation)
constexpr int kWaThreads = 256;
void dsblock_wa_launch_a(hipStream_t s, const DsP& p, int nf, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);   // NF 1..4 (P = 2)
void dsblock_wa_launch_b(hipStream_t s, const DsP& p, int nf, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);   // NF 5, 6, 8 (P = 2), 12 (P = 1)
template <int NF, int P, typename K>
static void dsblock_wa_one(K kernel, hipStream_t s, const DsP& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    static const bool once = [kernel] { OAR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); return true; }();
    (void)once;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kWaThreads), lds, s, e0, e1, 0, p);
}

#define OAR_DSBLOCK_INSTANTIATE(NAME, KS, SW)                                                                                                  \
    template <int NFW, int PFW>                                                                                                                \
    static void NAME##_one(hipStream_t s, const DsP& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {                                  \
        static const bool once = [] {                                                                                                          \
            OAR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dsblock_kernel<KS, SW, NFW, PFW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            return true;                                                                                                                       \
        }();                                                                                                                                   \
        (void)once;                                                                                                                            \
        hipExtLaunchKernelGGL((dsblock_kernel<KS, SW, NFW, PFW>), dim3(grid), dim3(512), lds, s, e0, e1, 0, p);                                \
    }                                                                                                                                          \
    void NAME(hipStream_t s, const DsP& p, int nfw, int pfw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {                             \
        const int key = nfw * 10 + pfw;                                                                                                        \
        switch (key) {                                                                                                                         \
            case 11: NAME##_one<1, 1>(s, p, grid, lds, e0, e1); break;                                                                         \
            case 21: NAME##_one<2, 1>(s, p, grid, lds, e0, e1); break;                                                                         \
            case 31: NAME##_one<3, 1>(s, p, grid, lds, e0, e1); break;                                                                         \
            case 41: NAME##_one<4, 1>(s, p, grid, lds, e0, e1); break;                                                                         \
            case 32: NAME##_one<3, 2>(s, p, grid, lds, e0, e1); break;                                                                         \
            case 42: NAME##_one<4, 2>(s, p, grid, lds, e0, e1); break;                                                                         \
            case 34: NAME##_one<3, 4>(s, p, grid, lds, e0, e1); break;                                                                         \
            case 44: NAME##_one<4, 4>(s, p, grid, lds, e0, e1); break;                                                                         \
            default: ::oar::fail(OAR_INTERNAL, "dsblock: no kernel for this wave layout");                                                     \
        }                                                                                                                                      \
    }

}  // namespace k
}  // namespace oar
