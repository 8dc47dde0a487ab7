// dsblock_rs.h -- host interface of the row-streaming fused depthwise-separable block (dsblock_rs.inc); included by dsblock.hip
// and by the instantiation units dsblock_rs_*.hip.
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>

#include "igemm_dev.h"
#include "dsblock.h"

namespace oar {
namespace k {

#include "dsblock_rs_p.inc"

// one translation unit per stride combination (3x3): NCH = channel chunks of 16, wpw = waves per workgroup (8 or 12)
void dsblock_rs_launch_k3s11(hipStream_t s, const DsRsP& p, int nch, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s21(hipStream_t s, const DsRsP& p, int nch, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s12(hipStream_t s, const DsRsP& p, int nch, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s22(hipStream_t s, const DsRsP& p, int nch, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
bool dsblock_rs_has(int nch, int wpw);

template <typename K>
static void dsblock_rs_one(K kernel, int wpw, hipStream_t s, const DsRsP& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    static const bool once = [kernel] { OAR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); return true; }();
    (void)once;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(wpw * 64), lds, s, e0, e1, 0, p);
}

#define OAR_DSBLOCK_RS_INSTANTIATE(NAME, KS, SH, SW)                                                                                           \
    void NAME(hipStream_t s, const DsRsP& p, int nch, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {                           \
        switch (nch * 100 + wpw) {                                                                                                             \
            case 108: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 1, 8>, 8, s, p, grid, lds, e0, e1); break;                                  \
            case 112: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 1, 12>, 12, s, p, grid, lds, e0, e1); break;                                \
            case 208: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 2, 8>, 8, s, p, grid, lds, e0, e1); break;                                  \
            case 212: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 2, 12>, 12, s, p, grid, lds, e0, e1); break;                                \
            case 308: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 3, 8>, 8, s, p, grid, lds, e0, e1); break;                                  \
            case 312: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 3, 12>, 12, s, p, grid, lds, e0, e1); break;                                \
            case 408: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 4, 8>, 8, s, p, grid, lds, e0, e1); break;                                  \
            case 608: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, 6, 8>, 8, s, p, grid, lds, e0, e1); break;                                  \
            default: ::oar::fail(OAR_INTERNAL, "dsblock_rs: no kernel for this shape");                                                        \
        }                                                                                                                                      \
    }

}  // namespace k
}  // namespace oar
