// dsblock_k3s1.hip -- fused depthwise-separable block, 3x3 depthwise, column stride 1 (see dsblock.inc)
#include "dsblock_dev.h"
namespace oar {
namespace k {
#include "dsblock.inc"
OAR_DSBLOCK_INSTANTIATE(dsblock_launch_k3s1, 3, 1)
}  // namespace k
}  // namespace oar
