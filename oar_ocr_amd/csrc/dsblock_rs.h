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

// one translation unit per stride combination (3x3): nch = channel chunks of 16, nft = cout fragments of 16, acts = 1 when both
// activations are hard swish (specialised), else 0 (none / relu / hswish read at run time); the unit fixes waves per workgroup
void dsblock_rs_launch_k3s11(hipStream_t s, const DsRsP& p, int nch, int nft, int x6, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s21(hipStream_t s, const DsRsP& p, int nch, int nft, int x6, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s12(hipStream_t s, const DsRsP& p, int nch, int nft, int x6, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_k3s22(hipStream_t s, const DsRsP& p, int nch, int nft, int x6, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);
void dsblock_rs_launch_dbg(hipStream_t s, const DsRsP& p, int dbg, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);   // dsblock_rs_dbg.hip
// waves per workgroup of the (nch, nft, x6) instantiation of stride (sh, sw); 0 = not instantiated
int dsblock_rs_wpw(int sh, int sw, int nch, int nft, int x6);

template <typename K>
static void dsblock_rs_one(K kernel, int wpw, hipStream_t s, const DsRsP& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    static const bool once = [kernel] { OAR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); return true; }();
    (void)once;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(wpw * 64), lds, s, e0, e1, 0, p);
}

// key = nch * 10000 + nft * 100 + x6 * 10 + acts
#define OAR_RS_CASE(KS, SH, SW, NCH, NFT, WPW, X6) \
    case (NCH) * 10000 + (NFT) * 100 + (X6) * 10 + 1: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, NCH, NFT, WPW, 1, X6>, WPW, s, p, grid, lds, e0, e1); break; \
    case (NCH) * 10000 + (NFT) * 100 + (X6) * 10 + 0: dsblock_rs_one(dsblock_rs_kernel<KS, SH, SW, NCH, NFT, WPW, 0, X6>, WPW, s, p, grid, lds, e0, e1); break;

}  // namespace k
}  // namespace oar
