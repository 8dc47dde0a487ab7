// dsblock_k5s2.hip -- fused depthwise-separable block, 5x5 depthwise, column stride 2 (see dsblock.inc)
#include "dsblock_dev.h"
namespace oar {
namespace k {
#include "dsblock.inc"
OAR_DSBLOCK_INSTANTIATE(dsblock_launch_k5s2, 5, 2)
}  // namespace k
}  // namespace oar
