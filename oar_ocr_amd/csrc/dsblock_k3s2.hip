// dsblock_k3s2.hip -- fused depthwise-separable block, 3x3 depthwise, column stride 2 (see dsblock.inc)
#include "dsblock_dev.h"
namespace oar {
namespace k {
#include "dsblock.inc"
OAR_DSBLOCK_INSTANTIATE(dsblock_launch_k3s2, 3, 2)
}  // namespace k
}  // namespace oar
