// dsblock_rs_k3s12.hip -- row-streaming fused depthwise-separable block, 3x3, stride (1, 2) (see dsblock_rs.inc)
#include "dsblock_rs.h"
namespace oar {
namespace k {
#include "dsblock_rs.inc"
OAR_DSBLOCK_RS_INSTANTIATE(dsblock_rs_launch_k3s12, 3, 1, 2)
}  // namespace k
}  // namespace oar
