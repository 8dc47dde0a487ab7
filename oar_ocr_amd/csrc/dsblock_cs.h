// dsblock_cs.h -- host interface of the chunk-streamed fused depthwise-separable block (dsblock_cs.inc); included by dsblock.hip and dsblock_cs.hip
#pragma once
#include <hip/hip_ext.h>

#include <type_traits>

#include "igemm_dev.h"
#include "dsblock.h"

namespace oar {
namespace k {

#include "dsblock_cs_p.inc"

// bytes of one weight block / of the kernel's LDS for the (ks, sh, sw, nch, nft) instantiation; 0 = not instantiated
size_t dsblock_cs_block_bytes(int ks, int nft);
size_t dsblock_cs_lds(int ks, int sh, int sw, int nch, int nft);
int dsblock_cs_rows(int ks, int sh, int sw, int nch, int nft);   // output rows of a tile (R) of that instantiation
void dsblock_cs_launch(hipStream_t s, const DsCsP& p, int ks, int sh, int sw, int nch, int nft, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1);

// dsblock_pc.hip: the producer / consumer kernel for the same work items and weight blocks; false = no instantiation for the shape
bool dsblock_pc_launch(hipStream_t s, const DsCsP& p, int ks, int sh, int sw, int nch, int nft, int acts, int grid, hipEvent_t e0, hipEvent_t e1);

}  // namespace k
}  // namespace oar
