// dsblock_pc.hip -- instantiations of the producer / consumer chunk-streamed separable block (dsblock_pc.inc, round 5)
#include "dsblock_cs.h"
namespace oar {
namespace k {
#include "dsblock_pc.inc"

namespace {
constexpr size_t pc_lds_bytes(int ks, int sh, int sw, int nft, int rows) {
    const int iw = 15 * sw + ks, ir = (rows - 1) * sh + ks, nj = (ir * iw * 4 + 63) / 64;
    const size_t slot = (size_t)nj * 1024, dwb = (size_t)(((ks * ks + 1) * 64 + 1023) / 1024 * 1024), pwb = (size_t)((nft * 1536 + 1023) / 1024 * 1024);
    return 8 * slot + 8 * dwb + 2 * pwb + 4 * 3 * (size_t)rows * 512 + 64 + 64 * (size_t)nft;
}
template <typename K>
void launch_pc(K kernel, hipStream_t s, const DsCsP& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    OAR_MAX_LDS_ONCE(kernel, 160 * 1024);
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, s, e0, e1, 0, p);
}
}  // namespace

#define OAR_PC_CASE(KS, SH, SW, NCH, NFT, ROWS) \
    if (ks == KS && sh == SH && sw == SW && nch == NCH && nft == NFT) { \
        constexpr size_t lds = pc_lds_bytes(KS, SH, SW, NFT, ROWS); \
        static_assert(lds <= 160 * 1024, "LDS"); \
        if (acts) launch_pc(dsblock_pc_kernel<KS, SH, SW, NCH, NFT, ROWS, 1>, s, p, grid, lds, e0, e1); \
        else launch_pc(dsblock_pc_kernel<KS, SH, SW, NCH, NFT, ROWS, 0>, s, p, grid, lds, e0, e1); \
        return true; \
    }
#define OAR_PC_DBG(D) case D: launch_pc(dsblock_pc_kernel<5, 1, 1, 12, 12, 4, 1, D>, s, p, grid, pc_lds_bytes(5, 1, 1, 12, 4), e0, e1); return true;
// false: no producer / consumer instantiation for this shape (the caller runs dsblock_cs_kernel).  The tile rows must be dsblock_cs's (kInst).
bool dsblock_pc_launch(hipStream_t s, const DsCsP& p, int ks, int sh, int sw, int nch, int nft, int acts, int grid, hipEvent_t e0, hipEvent_t e1) {
#ifdef OAR_DSB_ABLATIONS
    static const int dbg = [] { const char* e = getenv("OAR_DSB_PC_DBG"); return e ? atoi(e) : 0; }();   // timing ablations of the 192 -> 192 5x5 instantiation (wrong results)
    if (dbg && ks == 5 && nch == 12 && nft == 12 && acts) {
        switch (dbg) {
            OAR_PC_DBG(1) OAR_PC_DBG(2) OAR_PC_DBG(4) OAR_PC_DBG(8) OAR_PC_DBG(12) OAR_PC_DBG(32) OAR_PC_DBG(64) OAR_PC_DBG(128) OAR_PC_DBG(256) OAR_PC_DBG(28) OAR_PC_DBG(29) OAR_PC_DBG(30) OAR_PC_DBG(512) OAR_PC_DBG(640) OAR_PC_DBG(768) OAR_PC_DBG(16384) OAR_PC_DBG(16896)
            default: break;
        }
    }
#endif
    OAR_PC_CASE(5, 1, 1, 12, 12, 4)
    return false;
}
}  // namespace k
}  // namespace oar
