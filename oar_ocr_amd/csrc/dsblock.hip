// dsblock.hip -- host side of the fused depthwise-separable block: tile shape, wave layout, launch (kernel: dsblock.inc).
#include "dsblock_dev.h"
#include "dsblock_rs.h"
#include "dsblock_cs.h"
#include "dsblock_rs2.h"

namespace oar {
namespace k {

namespace {
struct DsPlanShape { int TR, TC, IR, IC, nfw, pfw; long tiles; int tiles_x, tiles_y; size_t lds; bool ok; };

DsPlanShape ds_shape(const DsBlockP& p) {
    DsPlanShape r{};
    r.ok = false;
    if (!(p.ks == 3 || p.ks == 5) || !(p.sh == 1 || p.sh == 2) || !(p.sw == 1 || p.sw == 2)) return r;
    if (p.C <= 0 || (p.C & 3) || p.Cout <= 0 || (p.Cout & 3) || p.Cout > 256 || (p.y_ld & 3)) return r;
    if (p.N <= 0 || p.Ho <= 0 || p.Wo <= 0) return r;
    // Measured per layer on the bench graphs (tools/dsblock_bench.py vs the conv_dw + conv_igemm pair, profiles/r2): the fused
    // kernel wins 10-45 % on every 3x3 block and loses on the wide 5x5 ones (192 -> 192: 271 us vs 117 + 130 us; 128 -> 128:
    // 41 vs 37 us), where one workgroup per CU (185+ VGPRs) is left and nothing hides its barrier / LDS latencies.
    static const bool force = [] { const char* e = getenv("OAR_FUSE_DSBLOCK"); return e && atoi(e) == 2; }();
    if (!force && p.ks == 5 && p.C >= 128) return r;
    // wave layout: WN cout groups x WP pixel groups (WN * WP = 4 waves), NFW fragments of 16 couts per wave
    const int nf = (p.Cout + 15) / 16;
    if (nf <= 4) { r.nfw = nf; r.pfw = 1; }                        // 8 waves = 1 cout group x 8 pixel groups
    else if (nf <= 8) { r.nfw = (nf + 1) / 2; r.pfw = 2; }         //           2 x 4
    else { r.nfw = (nf + 3) / 4; r.pfw = 4; }                      //           4 x 2
    if (r.pfw != 1 && r.nfw < 3) r.nfw = 3;
    // tile of 128 output pixels: the fewest tiles (least padded MFMA work), then the smallest input tile
    const int np_cap = (p.sw == 1 ? 6 : 11) * 64;   // pixels the prefetch registers can carry
    long best_tiles = -1; int best_in = 0;
    for (int tc : {16, 32, 64}) {
        const int tr = 128 / tc;
        const int ir = (tr - 1) * p.sh + p.ks, ic = (tc - 1) * p.sw + p.ks;
        if (ir * ic > np_cap) continue;
        const size_t lds = (size_t)ir * ic * 128 + (size_t)(p.ks * p.ks + 1) * 128 + 8 * 3 * 64 * 16 + (size_t)ir * ic * 4 + 16;
        if (lds > 150 * 1024) continue;
        const int tx = (p.Wo + tc - 1) / tc, ty = (p.Ho + tr - 1) / tr;
        const long tiles = (long)p.N * tx * ty;
        if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && ir * ic < best_in)) {
            best_tiles = tiles; best_in = ir * ic;
            r.TR = tr; r.TC = tc; r.IR = ir; r.IC = ic; r.tiles = tiles; r.tiles_x = tx; r.tiles_y = ty; r.lds = lds;
        }
    }
    r.ok = best_tiles > 0;
    return r;
}

// The wave-autonomous kernel (dsblock_wa.inc, round 3): 3x3, stride 1, any C / Cout whose fragment count has an instantiation.
struct WaShape { int nf, P; long tiles; int tiles_x, tiles_y; size_t lds; bool ok; };
WaShape wa_shape(const DsBlockP& p) {
    WaShape r{};
    const char* e = getenv("OAR_DSBLOCK_WA");   // read per call (plan time + launch): tests A/B the two kernels inside one process
    if ((e && atoi(e) == 0) || p.ks != 3 || p.sh != 1 || p.sw != 1) return r;
    if (p.C <= 0 || (p.C & 3) || p.Cout <= 0 || (p.Cout & 3) || (p.y_ld & 3) || p.N <= 0 || p.Ho <= 0 || p.Wo <= 0) return r;
    if (p.pt < 0 || p.pl < 0 || p.pt > 2 || p.pl > 2) return r;
    auto plain = [](const Act& a) { return a.kind == ACT_NONE || a.kind == ACT_RELU || a.kind == ACT_HSWISH; };
    if (!plain(p.act1) || !plain(p.act2)) return r;   // the rarely used activations stay with dsblock.inc (their exp / div paths cost this kernel registers)
    if ((long)p.H * p.W * p.C * 4 >= (1L << 31)) return r;   // 32-bit tile-relative source offsets
    r.nf = (p.Cout + 15) / 16;
    if (!(r.nf <= 6 || r.nf == 8 || r.nf == 12)) return r;
    // Measured per layer at the bench shapes (tools/dsblock_bench.py, this kernel vs dsblock.inc): 16 -> 24 120 vs 149 us, 48 -> 48 135 vs
    // 147, 24 -> 48 92 vs 91, 48 -> 48 @ 240^2 69 vs 80, 32 -> 32 33 vs 41, 64 -> 64 32 vs 33; 96 -> 96 212 vs 165: with 6+ cout fragments
    // per wave the weight fragments (L1) and 176+ registers (2 waves / SIMD) cost more than the LDS operand round trip they replace.
    // OAR_DSBLOCK_WA=2 takes every instantiated shape (A/B runs).
    if (r.nf > 4 && !(e && atoi(e) == 2)) return r;
    r.P = r.nf == 12 ? 1 : 2;
    const int TR = 4 * r.P, in_px = (TR + 2) * 18, nj = (in_px * 8 + 63) / 64;
    r.tiles_x = (p.Wo + 15) / 16; r.tiles_y = (p.Ho + TR - 1) / TR;
    r.tiles = (long)p.N * r.tiles_x * r.tiles_y;
    r.lds = (size_t)kWaRing * nj * 1024 + (size_t)((p.C + 31) / 32) * 1280 + (size_t)r.nf * 64;
    r.ok = true;
    return r;
}

void dsblock_wa(hipStream_t s, const DsBlockP& b, const WaShape& sh) {
    DsP p{};
    p.x = b.x; p.y = b.y; p.wd = b.wd; p.bd = b.bd; p.wp = reinterpret_cast<const uint4*>(b.wp); p.bp = b.bp; p.res = b.residual; p.se = b.se;
    p.N = b.N; p.H = b.H; p.W = b.W; p.C = b.C; p.Ho = b.Ho; p.Wo = b.Wo; p.Cout = b.Cout;
    p.sh = b.sh; p.sw = b.sw; p.pt = b.pt; p.pl = b.pl;
    p.act1 = b.act1.kind; p.a1 = b.act1.alpha; p.b1 = b.act1.beta;
    p.act2 = b.act2.kind; p.a2 = b.act2.alpha; p.b2 = b.act2.beta;
    p.TR = 4 * sh.P; p.TC = 16; p.IR = p.TR + 2; p.IC = 18; p.tiles_x = sh.tiles_x; p.tiles_y = sh.tiles_y; p.tiles = sh.tiles;
    p.KC = (b.C + 31) / 32; p.NF = sh.nf; p.y_ld = b.y_ld;
    { const char* e = getenv("OAR_DSB_DBG"); p.dbg = e ? atoi(e) : 0; }
    const int per_cu = (int)std::min<size_t>(4, (160 * 1024) / sh.lds);
    long grid = std::min<long>(sh.tiles, 256L * per_cu);
    grid = std::max<long>(8, (grid + 7) / 8 * 8);
    const double px_in = (double)b.N * b.H * b.W, px_out = (double)b.N * b.Ho * b.Wo;
    const double bytes = 4.0 * (px_in * b.C + px_out * b.Cout * (b.residual ? 2 : 1)) + 4.0 * b.ks * b.ks * b.C + 6.0 * b.C * b.Cout;
    const double flops = 2.0 * px_out * b.C * (b.ks * b.ks + (double)b.Cout);
    char pname[96];
    const char* cls = "dsblock_wa";   // one profiler class per kernel FAMILY (row-streaming: HBM-bound; chunk-streamed: instruction-bound; bench.py roofline.by_family)
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "dsblock px=%ld C=%d N=%d k3 s1x1 wa%d", (long)px_out, b.C, b.Cout, sh.P); cls = pname; }
    ProfScope ps(s, cls, bytes, flops, true);
    if (sh.nf <= 4) dsblock_wa_launch_a(s, p, sh.nf, (int)grid, sh.lds, ps.start(), ps.stop());
    else dsblock_wa_launch_b(s, p, sh.nf, (int)grid, sh.lds, ps.start(), ps.stop());
}

// The row-streaming kernel (dsblock_rs.inc, round 4): 3x3, strides 1 / 2, pointwise on the f32 matrix pipe, per-wave LDS-DMA rings.
struct RsShape { int nch, nf, acts, x6, wpw, NR, R, segs, tiles_x, items; unsigned lds_dw, lds_pw; size_t lds; bool ok; };
RsShape rs_shape(const DsBlockP& p) {
    RsShape r{};
    const char* e = getenv("OAR_DSBLOCK_RS");   // 0: never (A/B runs against dsblock / dsblock_wa)
    if ((e && atoi(e) == 0) || p.ks != 3 || !(p.sh == 1 || p.sh == 2) || !(p.sw == 1 || p.sw == 2)) return r;
    if (p.C < 4 || (p.C & 3) || p.Cout <= 0 || (p.Cout & 3) || (p.y_ld & 3) || p.N <= 0 || p.Ho <= 0 || p.Wo <= 0) return r;
    if (p.pt < 0 || p.pl < 0 || p.pt > 2 || p.pl > 2 || p.has_res || p.residual || p.se) return r;
    auto plain = [](const Act& a) { return a.kind == ACT_NONE || a.kind == ACT_RELU || a.kind == ACT_HSWISH; };
    if (!plain(p.act1) || !plain(p.act2)) return r;
    if ((long)p.H * p.W * p.C * 4 >= (1L << 29)) return r;                       // 32-bit image-relative DMA offsets, out-of-range marker 2^30
    if ((long)p.N * p.Ho * p.Wo * p.y_ld * 4 >= (1L << 31)) return r;            // 32-bit store offsets, out-of-range marker 2^31
    r.nch = (p.C + 15) / 16; r.nf = (p.Cout + 15) / 16;
    // pointwise on the f32 matrix instruction or as bf16x6 (dsblock_rs.inc, PWX6).  Measured per layer (tools/dsblock_bench.py, OAR_DSB_RS_X6=0|1):
    // the f32 instruction blocks its SIMD for 32 cycles each, so it only pays where the product is tiny
    { const char* f = getenv("OAR_DSB_RS_X6"); r.x6 = f ? (atoi(f) != 0) : (r.nf * r.nch >= 6); }
    r.wpw = dsblock_rs_wpw(p.sh, p.sw, r.nch, r.nf, r.x6);
    if (r.wpw == 0) { r.x6 = !r.x6; r.wpw = dsblock_rs_wpw(p.sh, p.sw, r.nch, r.nf, r.x6); }
    if (r.wpw == 0) return r;
    r.acts = (p.act1.kind == ACT_HSWISH && p.act2.kind == ACT_HSWISH) ? 1 : 0;
    const int IW = 15 * p.sw + 3, SLOT = r.nch * IW * 64, NSET = 2 / p.sh + 1, ncp = (r.nch + 1) / 2;   // (ring slots are packed: one row each)
    const bool pwreg = !r.x6 && r.nf * r.nch <= 12;
    const size_t tables = (size_t)10 * r.nch * 64 + (r.x6 ? (size_t)r.nf * ncp * 3 * 1024 : pwreg ? 0 : (size_t)r.nf * r.nch * 1024);
    if (tables + (size_t)r.wpw * SLOT * (p.sh + 1) > 160 * 1024) return r;
    r.NR = tables + (size_t)r.wpw * SLOT * (p.sh + 2) <= 160 * 1024 ? p.sh + 2 : p.sh + 1;   // the kernel computes the same (constexpr NR): one or two rows beyond the SH being consumed (deeper rings measured equal)
    r.tiles_x = (p.Wo + 15) / 16;
    // rows per item: the fewest wave rounds, then the least warm-up overhead (each item re-reads NSET - 1 rows of halo)
    const long waves = 256L * r.wpw;
    double best = 1e30;
    for (int R = std::min(p.Ho, 4); R <= p.Ho; ++R) {
        const int segs = (p.Ho + R - 1) / R;
        const long items = (long)p.N * segs * r.tiles_x;
        const long rounds = (items + waves - 1) / waves;
        const double cost = (double)rounds * (R + (NSET - 1) * 0.7 + 0.5);
        if (cost < best - 1e-9) { best = cost; r.R = R; r.segs = segs; r.items = (int)items; }
        if (R >= 64 && rounds == 1) break;
    }
    { const char* f = getenv("OAR_DSB_RS_R"); if (f && atoi(f) > 0) { r.R = std::min(p.Ho, atoi(f)); r.segs = (p.Ho + r.R - 1) / r.R; r.items = p.N * r.segs * r.tiles_x; } }
    const size_t rings = (size_t)r.wpw * r.NR * SLOT;
    r.lds_dw = (unsigned)rings; r.lds_pw = r.lds_dw + 10u * r.nch * 64u;
    r.lds = rings + tables;
    r.ok = r.lds <= 160 * 1024;
    return r;
}

void dsblock_rs(hipStream_t s, const DsBlockP& b, const RsShape& sh) {
    DsRsP p{};
    p.x = b.x; p.y = b.y; p.wd = b.wd; p.bd = b.bd; p.wp = reinterpret_cast<const float4*>(b.wp); p.bp = b.bp;
    p.N = b.N; p.H = b.H; p.W = b.W; p.C = b.C; p.Ho = b.Ho; p.Wo = b.Wo; p.Cout = b.Cout; p.y_ld = b.y_ld;
    p.pt = b.pt; p.pl = b.pl; p.act1 = b.act1.kind; p.act2 = b.act2.kind;
    p.R = sh.R; p.segs = sh.segs; p.tiles_x = sh.tiles_x; p.items = sh.items; p.per_xcd = (sh.items + 7) / 8;
    p.NF = sh.nf; p.NR = sh.NR; p.lds_dw = sh.lds_dw; p.lds_pw = sh.lds_pw; p.lds_pb = 0;
    p.img_bytes = (unsigned)((long)b.H * b.W * b.C * 4);
    p.y_bytes = (unsigned)((long)b.N * b.Ho * b.Wo * b.y_ld * 4);
    const int grid = 256;   // one persistent workgroup per CU (a multiple of the 8 XCDs: workgroup i runs on XCD i % 8 and walks that XCD's band)
    const double px_in = (double)b.N * b.H * b.W, px_out = (double)b.N * b.Ho * b.Wo;
    const double bytes = 4.0 * (px_in * b.C + px_out * b.Cout) + 4.0 * b.ks * b.ks * b.C + 4.0 * b.C * b.Cout;
    const double flops = 2.0 * px_out * b.C * (b.ks * b.ks + (double)b.Cout);
    char pname[96];
    const char* cls = "dsblock_rs";   // one profiler class per kernel FAMILY (row-streaming: HBM-bound; chunk-streamed: instruction-bound; bench.py roofline.by_family)
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "dsblock px=%ld C=%d N=%d k3 s%dx%d rs R%d", (long)px_out, b.C, b.Cout, b.sh, b.sw, sh.R); cls = pname; }
    ProfScope ps(s, cls, bytes, flops, true);
    const char* de = getenv("OAR_DSB_DBG");
    if (de && atoi(de) > 0 && b.sh == 1 && b.sw == 1 && sh.nch == 3 && sh.nf == 3 && sh.acts == 1 && !sh.x6) dsblock_rs_launch_dbg(s, p, atoi(de), grid, sh.lds, ps.start(), ps.stop());
    else if (b.sh == 1 && b.sw == 1) dsblock_rs_launch_k3s11(s, p, sh.nch, sh.nf, sh.x6, sh.acts, grid, sh.lds, ps.start(), ps.stop());
    else if (b.sh == 2 && b.sw == 1) dsblock_rs_launch_k3s21(s, p, sh.nch, sh.nf, sh.x6, sh.acts, grid, sh.lds, ps.start(), ps.stop());
    else if (b.sh == 1) dsblock_rs_launch_k3s12(s, p, sh.nch, sh.nf, sh.x6, sh.acts, grid, sh.lds, ps.start(), ps.stop());
    else dsblock_rs_launch_k3s22(s, p, sh.nch, sh.nf, sh.x6, sh.acts, grid, sh.lds, ps.start(), ps.stop());
}

// The chunk-streamed kernel (dsblock_cs.inc, round 4): the wide blocks (C / Cout up to 192, 3x3 and 5x5) whose tables do not fit dsblock_rs's LDS plan.
struct CsShape { int nch, nf, acts, segs, tiles_x, items, iters, grid; size_t lds; bool ok, forced; };
CsShape cs_shape(const DsBlockP& p) {
    CsShape r{};
    const char* e = getenv("OAR_DSBLOCK_CS");   // 0: never; 2: wherever an instantiation exists (A/B runs against dsblock_rs / dsblock)
    if (e && atoi(e) == 0) return r;
    r.forced = e && atoi(e) == 2;
    if (!(p.ks == 3 || p.ks == 5) || p.C <= 0 || (p.C & 15) || p.Cout <= 0 || (p.Cout & 15) || (p.y_ld & 3) || p.N <= 0 || p.Ho <= 0 || p.Wo <= 0) return r;
    if (p.pt < 0 || p.pl < 0 || p.pt >= p.ks || p.pl >= p.ks || p.has_res || p.residual || p.se) return r;
    auto plain = [](const Act& a) { return a.kind == ACT_NONE || a.kind == ACT_RELU || a.kind == ACT_HSWISH; };
    if (!plain(p.act1) || !plain(p.act2)) return r;
    if ((long)p.H * p.W * p.C * 4 >= (1L << 29) || (long)p.N * p.Ho * p.Wo * p.y_ld * 4 >= (1L << 31)) return r;
    r.nch = p.C / 16; r.nf = p.Cout / 16;
    r.lds = dsblock_cs_lds(p.ks, p.sh, p.sw, r.nch, r.nf);
    if (r.lds == 0) return r;
    r.acts = (p.act1.kind == ACT_HSWISH && p.act2.kind == ACT_HSWISH) ? 1 : 0;
    const int rows = dsblock_cs_rows(p.ks, p.sh, p.sw, r.nch, r.nf);
    r.segs = (p.Ho + rows - 1) / rows; r.tiles_x = (p.Wo + 15) / 16;
    const long items = (long)p.N * r.segs * r.tiles_x;
    if (items >= (1L << 30)) return r;
    r.items = (int)items;
    r.grid = (int)std::min<long>(256, ((items + 3) / 4 + 7) / 8 * 8);      // four waves = four items per workgroup and step; a multiple of the 8 XCDs
    r.iters = (int)((items + 4L * r.grid - 1) / (4L * r.grid));
    r.ok = true;
    return r;
}
void dsblock_cs(hipStream_t s, const DsBlockP& b, const CsShape& sh) {
    DsCsP p{};
    p.x = b.x; p.y = b.y; p.wb = reinterpret_cast<const float4*>(b.wp);
    p.bp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(b.wp) + (size_t)sh.nch * dsblock_cs_block_bytes(b.ks, sh.nf));   // the pointwise bias follows the blocks
    p.N = b.N; p.H = b.H; p.W = b.W; p.C = b.C; p.Ho = b.Ho; p.Wo = b.Wo; p.Cout = b.Cout; p.y_ld = b.y_ld;
    p.pt = b.pt; p.pl = b.pl; p.act1 = b.act1.kind; p.act2 = b.act2.kind;
    p.segs = sh.segs; p.tiles_x = sh.tiles_x; p.items = sh.items; p.iters = sh.iters;
    p.img_bytes = (unsigned)((long)b.H * b.W * b.C * 4);
    p.y_bytes = (unsigned)((long)b.N * b.Ho * b.Wo * b.y_ld * 4);
    const double px_in = (double)b.N * b.H * b.W, px_out = (double)b.N * b.Ho * b.Wo;
    const double bytes = 4.0 * (px_in * b.C + px_out * b.Cout) + 4.0 * b.ks * b.ks * b.C + 6.0 * b.C * b.Cout;
    const double flops = 2.0 * px_out * b.C * (b.ks * b.ks + (double)b.Cout);
    char pname[96];
    const char* cls = "dsblock_cs";   // one profiler class per kernel FAMILY (row-streaming: HBM-bound; chunk-streamed: instruction-bound; bench.py roofline.by_family)
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "dsblock px=%ld C=%d N=%d k%d s%dx%d cs", (long)px_out, b.C, b.Cout, b.ks, b.sh, b.sw); cls = pname; }
    ProfScope ps(s, cls, bytes, flops, true);
    dsblock_cs_launch(s, p, b.ks, b.sh, b.sw, sh.nch, sh.nf, sh.acts, sh.grid, sh.lds, ps.start(), ps.stop());
}
// The two-block row-streaming kernel (dsblock_rs2.inc, round 5): block a's output feeds block b through a per-wave LDS ring.
struct Rs2Shape { int nch1, nf1, nf2, acts, wpw, R, segs, tiles_x, items; unsigned lds_ring2, lds_dw1, lds_pw1, lds_dw2, lds_pw2; size_t lds; bool ok; };
Rs2Shape rs2_shape(const DsBlockP& a, const DsBlockP& b) {
    Rs2Shape r{};
    const char* e = getenv("OAR_DSBLOCK_RS2");
    if (e && atoi(e) == 0) return r;
    auto plain = [](const Act& x) { return x.kind == ACT_NONE || x.kind == ACT_RELU || x.kind == ACT_HSWISH; };
    auto simple = [&](const DsBlockP& p) {
        return p.ks == 3 && p.sh == 1 && p.sw == 1 && p.pt == 1 && p.pl == 1 && p.Ho == p.H && p.Wo == p.W && !p.has_res && !p.residual && !p.se && plain(p.act1) && plain(p.act2) &&
               p.C >= 4 && (p.C & 3) == 0 && p.Cout >= 4 && (p.Cout & 3) == 0;
    };
    if (!simple(a) || !simple(b) || a.Cout != b.C || a.N != b.N || a.H != b.H || a.W != b.W || a.N <= 0 || a.H <= 0 || a.W <= 0 || (b.y_ld & 3)) return r;
    if ((long)a.H * a.W * a.C * 4 >= (1L << 29) || (long)b.N * b.H * b.W * b.y_ld * 4 >= (1L << 31)) return r;
    r.nch1 = (a.C + 15) / 16; r.nf1 = (a.Cout + 15) / 16; r.nf2 = (b.Cout + 15) / 16;
    r.wpw = dsblock_rs2_wpw(r.nch1, r.nf1, r.nf2);
    if (r.wpw == 0) return r;
    const bool hs = a.act1.kind == ACT_HSWISH && a.act2.kind == ACT_HSWISH && b.act1.kind == ACT_HSWISH && b.act2.kind == ACT_HSWISH;
    r.acts = hs ? 1 : 0;
    const int nch2 = r.nf1, ncp1 = (r.nch1 + 1) / 2, ncp2 = (nch2 + 1) / 2;
    const size_t slot1 = (size_t)r.nch1 * 18 * 64, slot2 = (size_t)nch2 * 18 * 64;
    const int lag = dsblock_rs2_lag(r.nch1, r.nf1, r.nf2);
    const size_t rings1 = (size_t)r.wpw * 3 * slot1, rings2 = (size_t)r.wpw * (lag ? 2 : 1) * slot2;
    r.lds_ring2 = (unsigned)rings1;
    r.lds_dw1 = (unsigned)(rings1 + rings2);
    r.lds_pw1 = r.lds_dw1 + 10u * r.nch1 * 64u;
    r.lds_dw2 = r.lds_pw1 + (unsigned)(r.nf1 * ncp1 * 3 * 1024);
    r.lds_pw2 = r.lds_dw2 + 10u * nch2 * 64u;
    r.lds = (size_t)r.lds_pw2 + (size_t)r.nf2 * ncp2 * 3 * 1024;
    if (r.lds > 160 * 1024) return r;
    r.tiles_x = (a.W + 13) / 14;
    // rows per item: the fewest wave rounds, then the least pipeline fill (an item of R rows takes R + 5 iterations)
    const long waves = 256L * r.wpw;
    double best = 1e30;
    for (int R = std::min(a.H, 4); R <= a.H; ++R) {
        const int segs = (a.H + R - 1) / R;
        const long items = (long)a.N * segs * r.tiles_x;
        if (items >= (1L << 30)) continue;
        const long rounds = (items + waves - 1) / waves;
        const double cost = (double)rounds * (R + (lag ? 5 : 4));
        if (cost < best - 1e-9) { best = cost; r.R = R; r.segs = segs; r.items = (int)items; }
        if (R >= 64 && rounds == 1) break;
    }
    r.ok = r.R > 0;
    return r;
}
}  // namespace
bool dsblock2_eligible(const DsBlockP& a, const DsBlockP& b) {
    const char* e = getenv("OAR_FUSE_DSBLOCK");
    const bool on = !e || atoi(e) != 0;
    return on && rs2_shape(a, b).ok;
}
void dsblock2(hipStream_t s, const DsBlockP& a, const DsBlockP& b) {
    const Rs2Shape sh = rs2_shape(a, b);
    OAR_CHECK(sh.ok, OAR_INTERNAL, "dsblock2: called on an ineligible pair of blocks");
    DsRs2P p{};
    p.x = a.x; p.y = b.y;
    p.wd1 = a.wd; p.bd1 = a.bd; p.wp1 = reinterpret_cast<const float4*>(a.wp); p.bp1 = a.bp;
    p.wd2 = b.wd; p.bd2 = b.bd; p.wp2 = reinterpret_cast<const float4*>(b.wp); p.bp2 = b.bp;
    p.N = a.N; p.H = a.H; p.W = a.W; p.C1 = a.C; p.C2 = a.Cout; p.C3 = b.Cout; p.y_ld = b.y_ld;
    p.act11 = a.act1.kind; p.act12 = a.act2.kind; p.act21 = b.act1.kind; p.act22 = b.act2.kind;
    p.R = sh.R; p.segs = sh.segs; p.tiles_x = sh.tiles_x; p.items = sh.items; p.per_xcd = (sh.items + 7) / 8;
    p.lds_ring2 = sh.lds_ring2; p.lds_dw1 = sh.lds_dw1; p.lds_pw1 = sh.lds_pw1; p.lds_dw2 = sh.lds_dw2; p.lds_pw2 = sh.lds_pw2;
    p.img_bytes = (unsigned)((long)a.H * a.W * a.C * 4);
    p.y_bytes = (unsigned)((long)b.N * b.H * b.W * b.y_ld * 4);
    const double px = (double)a.N * a.H * a.W;
    const double bytes = 4.0 * px * (a.C + b.Cout) + 4.0 * 9 * (a.C + b.C) + 4.0 * ((double)a.C * a.Cout + (double)b.C * b.Cout);   // the intermediate tensor moves no bytes
    const double flops = 2.0 * px * (a.C * (9.0 + a.Cout) + b.C * (9.0 + b.Cout));
    char pname[96];
    const char* cls = "dsblock_rs";   // the row-streaming family (one launch, two blocks: its algorithmic bytes are the first block's input + the second's output)
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "dsblock2 px=%ld C=%d-%d-%d rs2 R%d", (long)px, a.C, a.Cout, b.Cout, sh.R); cls = pname; }
    ProfScope ps(s, cls, bytes, flops, true);
    dsblock_rs2_launch(s, p, sh.nch1, sh.nf1, sh.nf2, sh.acts, 256, sh.lds, ps.start(), ps.stop());
}
namespace {
// which kernel runs the block: 0 row-streaming, 1 wave-autonomous, 2 chunk-streamed, 3 dsblock.inc, -1 none
int ds_pick(const DsBlockP& p) {
    const CsShape cs = cs_shape(p);
    if (cs.ok && cs.forced) return 2;
    if (rs_shape(p).ok) return 0;
    if (wa_shape(p).ok) return 1;
    if (cs.ok) return 2;
    return ds_shape(p).ok ? 3 : -1;
}
}  // namespace

int dsblock_rs_wpw(int sh, int sw, int nch, int nft, int x6) {
    static const int T[][6] = {{1,1,1,1,0,12},{1,1,1,2,0,16},{1,1,2,2,0,12},{1,1,2,3,0,12},{1,1,2,3,1,16},{1,1,2,4,0,12},{1,1,2,4,1,12},{1,1,3,3,0,12},{1,1,3,3,1,12},{1,1,3,6,0,12},{1,1,3,6,1,12},{1,1,4,4,0,8},{1,1,4,4,1,8},{1,1,4,8,0,8},{1,1,4,8,1,8},{1,1,5,5,0,8},{1,1,5,5,1,8},{1,1,6,6,0,8},{1,1,6,6,1,6},{1,1,2,8,0,12},{2,1,3,6,0,12},{2,1,3,6,1,12},{2,1,2,4,0,12},{2,1,2,4,1,12},{2,1,1,2,0,12},{2,1,4,8,0,8},{2,1,4,8,1,8},{1,2,1,2,0,12},{1,2,2,4,0,12},{1,2,2,4,1,12},{1,2,3,6,0,8},{1,2,3,6,1,8},{1,2,4,8,0,6},{1,2,4,8,1,6},{2,2,2,2,0,12},{2,2,2,4,0,12},{2,2,2,4,1,8},{2,2,4,8,0,4},{2,2,4,8,1,4},{2,2,1,2,0,12},{2,2,3,6,0,6},{2,2,3,6,1,6}};   // generated with the instantiation units (dsblock_rs_k3s*.hip): sh, sw, nch, nft, x6, waves
    for (const auto& t : T) if (t[0] == sh && t[1] == sw && t[2] == nch && t[3] == nft && t[4] == x6) return t[5];
    return 0;
}

int dsblock_wp_format(const DsBlockP& p) {
    const int k = ds_pick(p);
    if (k == 2) return IGEMM_W_X6CS;
    if (k != 0) return IGEMM_W_X6;
    return rs_shape(p).x6 ? IGEMM_W_X6RS : IGEMM_W_K16;
}

bool dsblock_eligible(const DsBlockP& p) {
    const char* e = getenv("OAR_FUSE_DSBLOCK");
    const bool on = !e || atoi(e) != 0;
    return on && ds_pick(p) >= 0;
}

void dsblock(hipStream_t s, const DsBlockP& b) {
    const int pick = ds_pick(b);
    if (pick == 0) { dsblock_rs(s, b, rs_shape(b)); return; }
    if (pick == 1) { dsblock_wa(s, b, wa_shape(b)); return; }
    if (pick == 2) { dsblock_cs(s, b, cs_shape(b)); return; }
    const DsPlanShape sh = ds_shape(b);
    OAR_CHECK(sh.ok, OAR_INTERNAL, "dsblock: called on an ineligible block");
    DsP p{};
    p.x = b.x; p.y = b.y; p.wd = b.wd; p.bd = b.bd; p.wp = reinterpret_cast<const uint4*>(b.wp); p.bp = b.bp; p.res = b.residual; p.se = b.se;
    p.N = b.N; p.H = b.H; p.W = b.W; p.C = b.C; p.Ho = b.Ho; p.Wo = b.Wo; p.Cout = b.Cout;
    p.sh = b.sh; p.sw = b.sw; p.pt = b.pt; p.pl = b.pl;
    p.act1 = b.act1.kind; p.a1 = b.act1.alpha; p.b1 = b.act1.beta;
    p.act2 = b.act2.kind; p.a2 = b.act2.alpha; p.b2 = b.act2.beta;
    p.TR = sh.TR; p.TC = sh.TC; p.IR = sh.IR; p.IC = sh.IC; p.tiles_x = sh.tiles_x; p.tiles_y = sh.tiles_y; p.tiles = sh.tiles;
    p.KC = (b.C + 31) / 32; p.NF = (b.Cout + 15) / 16; p.y_ld = b.y_ld;
    // two workgroups per CU when LDS allows (and the variant's registers: the narrow layouts fit 128 VGPRs)
    const int per_cu = sh.lds * 2 <= 160 * 1024 ? 2 : 1;
    long grid = std::min<long>(sh.tiles, 256L * per_cu);
    grid = std::max<long>(8, (grid + 7) / 8 * 8);   // a multiple of the 8 XCDs: workgroup i runs on XCD i % 8 and walks that XCD's band of tiles
    const double px_in = (double)b.N * b.H * b.W, px_out = (double)b.N * b.Ho * b.Wo;
    const double bytes = 4.0 * (px_in * b.C + px_out * b.Cout * (b.residual ? 2 : 1)) + 4.0 * b.ks * b.ks * b.C + 6.0 * b.C * b.Cout;
    const double flops = 2.0 * px_out * b.C * (b.ks * b.ks + (double)b.Cout);
    char pname[96];
    const char* cls = "dsblock";   // one profiler class per kernel FAMILY (row-streaming: HBM-bound; chunk-streamed: instruction-bound; bench.py roofline.by_family)
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "dsblock px=%ld C=%d N=%d k%d s%dx%d t%dx%d", (long)px_out, b.C, b.Cout, b.ks, b.sh, b.sw, sh.TR, sh.TC); cls = pname; }
    ProfScope ps(s, cls, bytes, flops, true);
    if (b.ks == 3 && b.sw == 1) dsblock_launch_k3s1(s, p, sh.nfw, sh.pfw, (int)grid, sh.lds, ps.start(), ps.stop());
    else if (b.ks == 3) dsblock_launch_k3s2(s, p, sh.nfw, sh.pfw, (int)grid, sh.lds, ps.start(), ps.stop());
    else if (b.sw == 1) dsblock_launch_k5s1(s, p, sh.nfw, sh.pfw, (int)grid, sh.lds, ps.start(), ps.stop());
    else dsblock_launch_k5s2(s, p, sh.nfw, sh.pfw, (int)grid, sh.lds, ps.start(), ps.stop());
}

}  // namespace k
}  // namespace oar
