// dsblock_rs2.hip -- instantiations of the two-block row-streaming kernel (dsblock_rs2.inc, round 5)
#include "dsblock_rs2.h"

namespace oar {
namespace k {
#include "dsblock_rs2.inc"

namespace {
template <typename K>
void launch_rs2(K kernel, int wpw, hipStream_t s, const DsRs2P& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    OAR_MAX_LDS_ONCE(kernel, 160 * 1024);
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(wpw * 64), lds, s, e0, e1, 0, p);
}
struct Rs2Inst { int nch1, nf1, nf2, wpw; };
constexpr Rs2Inst kRs2[] = {{2, 3, 3, 8}, {1, 2, 3, 16}};
int rs2_variant() { static const int v = [] { const char* e = getenv("OAR_DSB_RS2_VARIANT"); return e ? atoi(e) : 0; }(); return v; }   // 1: 12 waves, one ring-2 slot, taps in LDS (24 -> 48 -> 48)   // 24 -> 48 -> 48, 16 -> 24 -> 48
}  // namespace

int dsblock_rs2_lag(int nch1, int nf1, int nf2) { return (rs2_variant() == 1 && nch1 == 2 && nf1 == 3 && nf2 == 3) ? 0 : 1; }
int dsblock_rs2_wpw(int nch1, int nf1, int nf2) {
    if (!dsblock_rs2_lag(nch1, nf1, nf2)) return 12;
    for (const auto& t : kRs2) if (t.nch1 == nch1 && t.nf1 == nf1 && t.nf2 == nf2) return t.wpw;
    return 0;
}

#define OAR_RS2_CASE(NCH1, NF1, NF2, WPW, RW1, LAG) \
    if (nch1 == NCH1 && nf1 == NF1 && nf2 == NF2) { \
        if (acts) launch_rs2(dsblock_rs2_kernel<NCH1, NF1, NF2, WPW, 1, RW1, LAG>, WPW, s, p, grid, lds, e0, e1); \
        else launch_rs2(dsblock_rs2_kernel<NCH1, NF1, NF2, WPW, 0, RW1, LAG>, WPW, s, p, grid, lds, e0, e1); \
        return; \
    }
void dsblock_rs2_launch(hipStream_t s, const DsRs2P& p, int nch1, int nf1, int nf2, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    static const bool rw = [] { const char* e = getenv("OAR_DSB_RS2_REGW"); return !e || atoi(e) != 0; }();   // stage 1's taps in registers (A/B knob)
    if (!dsblock_rs2_lag(nch1, nf1, nf2)) { OAR_RS2_CASE(2, 3, 3, 12, false, false) }
    if (rw) { OAR_RS2_CASE(2, 3, 3, 8, true, true) }
    OAR_RS2_CASE(2, 3, 3, 8, false, true)
    OAR_RS2_CASE(1, 2, 3, 16, false, true)
    ::oar::fail(OAR_INTERNAL, "dsblock_rs2: no kernel for this shape");
}
}  // namespace k
}  // namespace oar
