// dsblock_rs_dbg.hip -- timing ablations of the row-streaming block at its flagship shape (48 -> 48, 3x3 stride 1, hard swish): OAR_DSB_DBG=<mask>
// selects a variant that leaves ingredients out (WRONG results; tools/dsblock_bench.py only).  See DBG in dsblock_rs.inc.
// Built only with OAR_DSB_ABLATIONS=1 in the environment of oar_ocr_amd/build.py (wrong-result kernels are a build option, not product code);
// the product build gets a stub that fails loudly.
#include "dsblock_rs.h"
namespace oar {
namespace k {
#ifndef OAR_DSB_ABLATIONS
void dsblock_rs_launch_dbg(hipStream_t, const DsRsP&, int, int, size_t, hipEvent_t, hipEvent_t) {
    ::oar::fail(OAR_INTERNAL, "OAR_DSB_DBG needs a library built with OAR_DSB_ABLATIONS=1");
}
#else
#include "dsblock_rs.inc"
void dsblock_rs_launch_dbg(hipStream_t s, const DsRsP& p, int dbg, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (dbg) {
        case 1: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 1>, 12, s, p, grid, lds, e0, e1); break;
        case 2: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 2>, 12, s, p, grid, lds, e0, e1); break;
        case 3: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 3>, 12, s, p, grid, lds, e0, e1); break;
        case 7: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 7>, 12, s, p, grid, lds, e0, e1); break;
        case 8: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 8>, 12, s, p, grid, lds, e0, e1); break;
        case 9: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 9>, 12, s, p, grid, lds, e0, e1); break;
        case 10: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 10>, 12, s, p, grid, lds, e0, e1); break;
        case 11: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 11>, 12, s, p, grid, lds, e0, e1); break;
        case 15: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 15>, 12, s, p, grid, lds, e0, e1); break;
        case 16: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 16>, 12, s, p, grid, lds, e0, e1); break;
        case 19: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 19>, 12, s, p, grid, lds, e0, e1); break;
        case 23: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 23>, 12, s, p, grid, lds, e0, e1); break;
        case 32: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 32>, 12, s, p, grid, lds, e0, e1); break;
        case 64: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 64>, 12, s, p, grid, lds, e0, e1); break;
        case 67: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 67>, 12, s, p, grid, lds, e0, e1); break;
        case 71: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 71>, 12, s, p, grid, lds, e0, e1); break;
        case 72: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 72>, 12, s, p, grid, lds, e0, e1); break;
        case 79: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 79>, 12, s, p, grid, lds, e0, e1); break;
        case 135: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 135>, 12, s, p, grid, lds, e0, e1); break;
        case 143: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 143>, 12, s, p, grid, lds, e0, e1); break;
        case 87: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 87>, 12, s, p, grid, lds, e0, e1); break;
        case 207: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 207>, 12, s, p, grid, lds, e0, e1); break;
        case 256: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 256>, 12, s, p, grid, lds, e0, e1); break;
        case 512: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 512>, 12, s, p, grid, lds, e0, e1); break;
        case 768: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 768>, 12, s, p, grid, lds, e0, e1); break;
        case 328: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 328>, 12, s, p, grid, lds, e0, e1); break;
        case 584: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 584>, 12, s, p, grid, lds, e0, e1); break;
        case 840: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 840>, 12, s, p, grid, lds, e0, e1); break;
        case 264: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 264>, 12, s, p, grid, lds, e0, e1); break;
        case 520: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 520>, 12, s, p, grid, lds, e0, e1); break;
        case 1024: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 1024>, 12, s, p, grid, lds, e0, e1); break;
        case 1032: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 1032>, 12, s, p, grid, lds, e0, e1); break;
        case 1096: dsblock_rs_one(dsblock_rs_kernel<3, 1, 1, 3, 3, 12, 1, 0, 1096>, 12, s, p, grid, lds, e0, e1); break;
        default: ::oar::fail(OAR_INTERNAL, "dsblock_rs_dbg: this mask is not instantiated");
    }
}
#endif
}  // namespace k
}  // namespace oar
