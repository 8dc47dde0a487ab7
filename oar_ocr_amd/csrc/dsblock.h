// dsblock.h -- fused depthwise-separable block (dsblock.inc); launcher + eligibility.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace oar {
namespace k {

struct DsBlockP {
    int N, H, W, C;            // input NHWC
    int Ho, Wo, Cout;          // output NHWC
    int ks, sh, sw, pt, pl;    // depthwise: square kernel 3 / 5, strides 1 / 2
    Act act1, act2;
    const float* x;
    const float* wd;           // depthwise weights [ks*ks][C]
    const float* bd;           // may be null
    const float* wp;           // pointwise weights in the order dsblock_wp_format() names (IGEMM_W_X6CS: whole weight blocks, wd / bd unused)
    const float* bp;           // may be null
    const float* residual;     // may be null, shape of y
    bool has_res;              // plan time (pointers not known yet): the block ends in + residual
    const float* se;           // may be null: [N][C] gate applied to the depthwise output after act1
    float* y;
    int y_ld;
};
// true when dsblock() can run the block (shape / stride / channel limits); the planner keeps the two convolutions otherwise
bool dsblock_eligible(const DsBlockP& p);
void dsblock(hipStream_t s, const DsBlockP& p);
// fragment layout dsblock() expects p.wp in for this block (IGEMM_W_X6 for the bf16x6 kernels, IGEMM_W_K16 for the f32 row-streaming kernel);
// depends on the shape only (not on the pointers)
int dsblock_wp_format(const DsBlockP& p);
// Two consecutive blocks (a: x -> t, b: t -> y; both 3x3 / stride 1 / pad 1, no residual, no gate) as ONE launch whose intermediate tensor never reaches HBM
// (dsblock_rs2.inc).  Both pointwise tables must be in IGEMM_W_X6RS order; a.y and b.x are unused.  OAR_DSBLOCK_RS2=0 turns it off.
bool dsblock2_eligible(const DsBlockP& a, const DsBlockP& b);
void dsblock2(hipStream_t s, const DsBlockP& a, const DsBlockP& b);
// IGEMM_W_X6CS: bytes of one channel chunk's weight block (dsblock_cs.inc: [taps | bias] rounded to 1 KB, then 1536 B per cout fragment, rounded to 1 KB)
size_t dsblock_cs_block_bytes(int ks, int nft);

}  // namespace k
}  // namespace oar
