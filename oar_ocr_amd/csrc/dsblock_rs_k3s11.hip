// dsblock_rs_k3s11.hip -- row-streaming fused depthwise-separable block, 3x3, stride (1, 1) (see dsblock_rs.inc)
#include "dsblock_rs.h"
namespace oar {
namespace k {
#include "dsblock_rs.inc"
void dsblock_rs_launch_k3s11(hipStream_t s, const DsRsP& p, int nch, int nft, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (nch * 1000 + nft * 10 + acts) {
        OAR_RS_CASE(3, 1, 1, 1, 1, 12)
        OAR_RS_CASE(3, 1, 1, 1, 2, 12)
        OAR_RS_CASE(3, 1, 1, 2, 2, 12)
        OAR_RS_CASE(3, 1, 1, 2, 3, 12)
        OAR_RS_CASE(3, 1, 1, 2, 4, 12)
        OAR_RS_CASE(3, 1, 1, 3, 3, 12)
        OAR_RS_CASE(3, 1, 1, 3, 6, 8)
        OAR_RS_CASE(3, 1, 1, 4, 4, 8)
        OAR_RS_CASE(3, 1, 1, 4, 8, 8)
        OAR_RS_CASE(3, 1, 1, 5, 5, 8)
        OAR_RS_CASE(3, 1, 1, 6, 6, 8)
        OAR_RS_CASE(3, 1, 1, 2, 8, 8)
        default: ::oar::fail(OAR_INTERNAL, "dsblock_rs: no kernel for this shape");
    }
}
}  // namespace k
}  // namespace oar
