// dsblock_wa_a.hip -- wave-autonomous fused depthwise-separable block, 1..4 cout fragments (see dsblock_wa.inc)
#include "dsblock_dev.h"
namespace oar {
namespace k {
#include "dsblock_wa.inc"
void dsblock_wa_launch_a(hipStream_t s, const DsP& p, int nf, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    switch (nf) {
        case 1: dsblock_wa_one<1, 2>(dsblock_wa_kernel<1, 2>, s, p, grid, lds, e0, e1); break;
        case 2: dsblock_wa_one<2, 2>(dsblock_wa_kernel<2, 2>, s, p, grid, lds, e0, e1); break;
        case 3: dsblock_wa_one<3, 2>(dsblock_wa_kernel<3, 2>, s, p, grid, lds, e0, e1); break;
        case 4: dsblock_wa_one<4, 2>(dsblock_wa_kernel<4, 2>, s, p, grid, lds, e0, e1); break;
        default: ::oar::fail(OAR_INTERNAL, "dsblock_wa: no kernel for this fragment count");
    }
}
}  // namespace k
}  // namespace oar
