// dsblock_rs2.h -- host interface of the two-block row-streaming kernel (dsblock_rs2.inc); included by dsblock.hip and dsblock_rs2.hip
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>

#include "igemm_dev.h"
#include "dsblock.h"

namespace oar {
namespace k {

#include "dsblock_rs2_p.inc"

// waves per workgroup of the (nch1, nf1, nf2) instantiation; 0 = not instantiated
int dsblock_rs2_wpw(int nch1, int nf1, int nf2);
// 1: the instantiation keeps two slots in the second ring and an item takes R + 5 iterations; 0: one slot, R + 4
int dsblock_rs2_lag(int nch1, int nf1, int nf2);
void dsblock_rs2_launch(hipStream_t s, const DsRs2P& p, int nch1, int nf1, int nf2, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);

}  // namespace k
}  // namespace oar
