// dsblock_k5s1.hip -- fused depthwise-separable block, 5x5 depthwise, column stride 1 (see dsblock.inc)
#include "dsblock_dev.h"
namespace oar {
namespace k {
#include "dsblock.inc"
OAR_DSBLOCK_INSTANTIATE(dsblock_launch_k5s1, 5, 1)
}  // namespace k
}  // namespace oar
