// dsblock_cs.hip -- instantiations of the chunk-streamed fused depthwise-separable block (dsblock_cs.inc)
#include "dsblock_cs.h"
namespace oar {
namespace k {
#include "dsblock_cs.inc"

namespace {
struct CsInst { int ks, sh, sw, nch, nft, rows; };   // rows: output rows of a tile (R): 4, or 2 where 16 cout fragments leave room for no more accumulators
constexpr CsInst kInst[] = {{5, 1, 1, 12, 12, 4}, {3, 1, 2, 6, 12, 4}, {3, 1, 1, 6, 6, 4}, {5, 1, 1, 8, 8, 4}, {3, 1, 1, 8, 8, 4}, {3, 1, 1, 16, 16, 2}, {3, 2, 2, 8, 16, 2}};
constexpr size_t block_bytes(int ks, int nft) { return (size_t)(((ks * ks + 1) * 64 + 1023) / 1024 * 1024 + (nft * 1536 + 1023) / 1024 * 1024); }
constexpr size_t lds_bytes(int ks, int sh, int sw, int nft, int rows) {
    const int iw = 15 * sw + ks, ir = (rows - 1) * sh + ks, nj = (ir * iw * 4 + 63) / 64;
    const size_t slot = (size_t)nj * 1024, three = 12 * slot + 2 * block_bytes(ks, nft);   // three chunk slots per wave when they fit (the kernel's NRG)
    return three <= 160 * 1024 ? three : 8 * slot + 2 * block_bytes(ks, nft);
}
template <typename K>
void launch_one(K kernel, hipStream_t s, const DsCsP& p, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    OAR_MAX_LDS_ONCE(kernel, 160 * 1024);
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, s, e0, e1, 0, p);
}
}  // namespace

size_t dsblock_cs_block_bytes(int ks, int nft) { return block_bytes(ks, nft); }
size_t dsblock_cs_lds(int ks, int sh, int sw, int nch, int nft) {
    for (const auto& t : kInst) if (t.ks == ks && t.sh == sh && t.sw == sw && t.nch == nch && t.nft == nft) return lds_bytes(ks, sh, sw, nft, t.rows);
    return 0;
}
int dsblock_cs_rows(int ks, int sh, int sw, int nch, int nft) {
    for (const auto& t : kInst) if (t.ks == ks && t.sh == sh && t.sw == sw && t.nch == nch && t.nft == nft) return t.rows;
    return 0;
}

#define OAR_CS_CASE(KS, SH, SW, NCH, NFT, ROWS) \
    if (ks == KS && sh == SH && sw == SW && nch == NCH && nft == NFT) { \
        if (acts) launch_one(dsblock_cs_kernel<KS, SH, SW, NCH, NFT, ROWS, 1>, s, p, grid, lds, e0, e1); \
        else launch_one(dsblock_cs_kernel<KS, SH, SW, NCH, NFT, ROWS, 0>, s, p, grid, lds, e0, e1); \
        return; \
    }
#define OAR_CS_DBG(D) case D: launch_one(dsblock_cs_kernel<5, 1, 1, 12, 12, 4, 1, D>, s, p, grid, lds, e0, e1); return;
void dsblock_cs_launch(hipStream_t s, const DsCsP& p, int ks, int sh, int sw, int nch, int nft, int acts, int grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    // producer / consumer form (dsblock_pc.inc, round 5) wherever it is instantiated; OAR_DSB_PC=0 keeps the one-wave-per-SIMD kernel (A/B runs)
    static const bool pc_on = [] { const char* e = getenv("OAR_DSB_PC"); return !e || atoi(e) != 0; }();
    if (pc_on && dsblock_pc_launch(s, p, ks, sh, sw, nch, nft, acts, grid, e0, e1)) return;
#ifdef OAR_DSB_ABLATIONS
    static const int dbg = [] { const char* e = getenv("OAR_DSB_CS_DBG"); return e ? atoi(e) : 0; }();   // timing ablations of the 192 -> 192 5x5 instantiation (wrong results)
    if (dbg && ks == 5 && nch == 12 && nft == 12 && acts) {
        switch (dbg) {
            OAR_CS_DBG(1) OAR_CS_DBG(4) OAR_CS_DBG(8) OAR_CS_DBG(12) OAR_CS_DBG(28) OAR_CS_DBG(29) OAR_CS_DBG(30) OAR_CS_DBG(31) OAR_CS_DBG(60) OAR_CS_DBG(92) OAR_CS_DBG(156) OAR_CS_DBG(220) OAR_CS_DBG(255)
            default: break;
        }
    }
#endif
    OAR_CS_CASE(5, 1, 1, 12, 12, 4)
    OAR_CS_CASE(3, 1, 2, 6, 12, 4)
    OAR_CS_CASE(3, 1, 1, 6, 6, 4)
    OAR_CS_CASE(5, 1, 1, 8, 8, 4)
    OAR_CS_CASE(3, 1, 1, 8, 8, 4)
    OAR_CS_CASE(3, 1, 1, 16, 16, 2)
    OAR_CS_CASE(3, 2, 2, 8, 16, 2)
    ::oar::fail(OAR_INTERNAL, "dsblock_cs: no kernel for this shape");
}
}  // namespace k
}  // namespace oar
