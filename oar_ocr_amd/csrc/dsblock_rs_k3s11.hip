// dsblock_rs_k3s11.hip -- row-streaming fused depthwise-separable block, 3x3, stride (1, 1) (see dsblock_rs.inc)
#include "dsblock_rs.h"
namespace oar {
namespace k {
#include "dsblock_rs.inc"
void dsblock_rs_launch_k3s11(hipStream_t s, const DsRsP& p, int nch, int nft, int wpw, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (nch * 10000 + nft * 100 + wpw) {
        OAR_RS_CASE(3, 1, 1, 1, 2, 8)
        OAR_RS_CASE(3, 1, 1, 1, 2, 12)
        OAR_RS_CASE(3, 1, 1, 1, 2, 16)
        OAR_RS_CASE(3, 1, 1, 1, 1, 8)
        OAR_RS_CASE(3, 1, 1, 1, 1, 12)
        OAR_RS_CASE(3, 1, 1, 1, 1, 16)
        OAR_RS_CASE(3, 1, 1, 2, 2, 8)
        OAR_RS_CASE(3, 1, 1, 2, 2, 12)
        OAR_RS_CASE(3, 1, 1, 2, 2, 16)
        OAR_RS_CASE(3, 1, 1, 2, 3, 8)
        OAR_RS_CASE(3, 1, 1, 2, 3, 12)
        OAR_RS_CASE(3, 1, 1, 2, 3, 16)
        OAR_RS_CASE(3, 1, 1, 3, 3, 8)
        OAR_RS_CASE(3, 1, 1, 3, 3, 12)
        OAR_RS_CASE(3, 1, 1, 3, 3, 16)
        OAR_RS_CASE(3, 1, 1, 4, 4, 8)
        OAR_RS_CASE(3, 1, 1, 4, 4, 12)
        OAR_RS_CASE(3, 1, 1, 6, 6, 4)
        OAR_RS_CASE(3, 1, 1, 6, 6, 8)
        OAR_RS_CASE(3, 1, 1, 1, 0, 8)
        OAR_RS_CASE(3, 1, 1, 1, 0, 12)
        OAR_RS_CASE(3, 1, 1, 1, 0, 16)
        OAR_RS_CASE(3, 1, 1, 2, 0, 8)
        OAR_RS_CASE(3, 1, 1, 2, 0, 12)
        OAR_RS_CASE(3, 1, 1, 2, 0, 16)
        OAR_RS_CASE(3, 1, 1, 3, 0, 8)
        OAR_RS_CASE(3, 1, 1, 3, 0, 12)
        OAR_RS_CASE(3, 1, 1, 3, 0, 16)
        OAR_RS_CASE(3, 1, 1, 4, 0, 8)
        OAR_RS_CASE(3, 1, 1, 4, 0, 12)
        OAR_RS_CASE(3, 1, 1, 6, 0, 4)
        OAR_RS_CASE(3, 1, 1, 6, 0, 8)
        default: ::oar::fail(OAR_INTERNAL, "dsblock_rs: no kernel for this shape");
    }
}
}  // namespace k
}  // namespace oar
