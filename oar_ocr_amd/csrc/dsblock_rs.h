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
// one translation unit per stride combination (3x3): nch = channel chunks of 16, nf = cout fragments of 16 (0: the generic kernel that
// loops over fragment pairs at run time), wpw = waves per workgroup
void dsblock_rs_launch_k3s11(hipStream_t s, const DsRsP& p, int nch, int nft, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s21(hipStream_t s, const DsRsP& p, int nch, int nft, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s12(hipStream_t s, const DsRsP& p, int nch, int nft, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s22(hipStream_t s, const DsRsP& p, int nch, int nft, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
// which (nch, nft, wpw) the unit of stride (sh, sw) instantiates
bool dsblock_rs_has(int sh, int sw, int nch, int nft, int wpw);

template <typename K>
static void dsblock_rs_one(K kernel, int wpw, hipStream_t s, const DsRsP& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    static const bool once = [kernel] { OAR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); return true; }();
    (void)once;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(wpw * 64), lds, s, e0, e1, 0, p);
}

#define OAR_RS_CASE(KS, SH, SW, NCH, NFT, WPW) \
    case (NCH) * 10000 + (NFT) * 100 + (WPW): dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, NCH, NFT, WPW>, WPW, s, p, grid, lds, e0, e1); break;

}  // namespace k
}  // namespace oar
