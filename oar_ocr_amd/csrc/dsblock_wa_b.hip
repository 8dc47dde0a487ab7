// dsblock_wa_b.hip -- wave-autonomous fused depthwise-separable block, 5 / 6 / 8 / 12 cout fragments (see dsblock_wa.inc)
#include "dsblock_dev.h"
namespace oar {
namespace k {
#include "dsblock_wa.inc"
void dsblock_wa_launch_b(hipStream_t s, const DsP& p, int nf, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (nf) {
        case 5: dsblock_wa_one<5, 2>(dsblock_wa_kernel<5, 2>, s, p, grid, lds, e0, e1); break;
        case 6: dsblock_wa_one<6, 2>(dsblock_wa_kernel<6, 2>, s, p, grid, lds, e0, e1); break;
        case 8: dsblock_wa_one<8, 2>(dsblock_wa_kernel<8, 2>, s, p, grid, lds, e0, e1); break;
        case 12: dsblock_wa_one<12, 1>(dsblock_wa_kernel<12, 1>, s, p, grid, lds, e0, e1); break;
        default: ::oar::fail(OAR_INTERNAL, "dsblock_wa: no kernel for this fragment count");
    }
}
}  // namespace k
}  // namespace oar
