// dsblock_rs_dbg.hip -- timing ablations of the row-streaming block at its flagship shape (48 -> 48, 3x3 stride 1, hard swish): OAR_DSB_DBG=1..5
// selects a variant that leaves one ingredient out (WRONG results; tools/dsblock_bench.py only).  See DBG in dsblock_rs.inc.
#include "dsblock_rs.h"
namespace oar {
namespace k {
#include "dsblock_rs.inc"
void dsblock_rs_launch_dbg(hipStream_t s, const DsRsP& p, int dbg, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (dbg) {
        case 1: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 1>, 12, s, p, grid, lds, e0, e1); break;
        case 2: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 2>, 12, s, p, grid, lds, e0, e1); break;
        case 3: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 3>, 12, s, p, grid, lds, e0, e1); break;
        case 4: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 4>, 12, s, p, grid, lds, e0, e1); break;
        default: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 5>, 12, s, p, grid, lds, e0, e1); break;
    }
}
}  // namespace k
}  // namespace oar
