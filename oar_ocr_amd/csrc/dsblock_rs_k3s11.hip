// dsblock_rs_k3s11.hip -- row-streaming fused depthwise-separable block, 3x3, stride (1, 1) (see dsblock_rs.inc)
#include "dsblock_rs.h"
namespace oar {
namespace k {
#include "dsblock_rs.inc"
void dsblock_rs_launch_k3s11(hipStream_t s, const DsRsP& p, int nch, int nft, int x6, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (nch * 10000 + nft * 100 + x6 * 10 + acts) {
        OAR_RS_CASE(3, 1, 1, 1, 1, 12, 0)
        OAR_RS_CASE(3, 1, 1, 1, 2, 16, 0)
        OAR_RS_CASE(3, 1, 1, 2, 2, 12, 0)
        OAR_RS_CASE(3, 1, 1, 2, 3, 12, 0)
        OAR_RS_CASE(3, 1, 1, 2, 3, 16, 1)
        OAR_RS_CASE(3, 1, 1, 2, 4, 12, 0)
        OAR_RS_CASE(3, 1, 1, 2, 4, 12, 1)
        OAR_RS_CASE(3, 1, 1, 3, 3, 12, 0)
        OAR_RS_CASE(3, 1, 1, 3, 3, 12, 1)
        OAR_RS_CASE(3, 1, 1, 3, 6, 12, 0)
        OAR_RS_CASE(3, 1, 1, 3, 6, 12, 1)
        OAR_RS_CASE(3, 1, 1, 4, 4, 8, 0)
        OAR_RS_CASE(3, 1, 1, 4, 4, 8, 1)
        OAR_RS_CASE(3, 1, 1, 4, 8, 8, 0)
        OAR_RS_CASE(3, 1, 1, 4, 8, 8, 1)
        OAR_RS_CASE(3, 1, 1, 5, 5, 8, 0)
        OAR_RS_CASE(3, 1, 1, 5, 5, 8, 1)
        OAR_RS_CASE(3, 1, 1, 6, 6, 8, 0)
        OAR_RS_CASE(3, 1, 1, 6, 6, 6, 1)
        OAR_RS_CASE(3, 1, 1, 2, 8, 12, 0)
        default: ::oar::fail(OAR_INTERNAL, "dsblock_rs: no kernel for this shape");
    }
}
}  // namespace k
}  // namespace oar
