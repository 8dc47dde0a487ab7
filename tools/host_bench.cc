// Host-side DB post-processing microbenchmark on real masks (gpurun_out/masks8.bin: 8 x 960 x 960 bytes, the oracle's
// thresholded probability maps of bench pages 0..7; made by `python tools/make_mask.py --oracle 8`).  One thread, per-page figures.
//   /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -x c++ -Ioar_ocr_amd/csrc tools/host_bench.cc oar_ocr_amd/csrc/db_host.cc -o /tmp/host_bench
#include "db_host.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <array>
#include <vector>
using namespace oar::host;
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main(int argc, char** argv) {
    const int W = 960, H = 960, N = 8, REP = 20;
    std::vector<uint8_t> m((size_t)N * W * H);
    FILE* f = fopen(argc > 1 ? argv[1] : "gpurun_out/masks8.bin", "rb");
    if (!f || fread(m.data(), 1, m.size(), f) != m.size()) { fprintf(stderr, "no masks\n"); return 1; }
    fclose(f);
    const int rb = (W + 7) / 8;
    std::vector<uint8_t> bits((size_t)N * H * rb, 0);
    for (int k = 0; k < N; ++k)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                if (m[((size_t)k * H + y) * W + x]) bits[((size_t)k * H + y) * rb + (x >> 3)] |= 1u << (x & 7);
    double t_find = 0, t_mb = 0, t_fin = 0;
    size_t n_un = 0, n_fin = 0;
    size_t n_c = 0, n_pts = 0, n_box = 0;
    for (int r = 0; r < REP; ++r) {
        for (int k = 0; k < N; ++k) {
            auto t0 = clk::now();
            std::vector<Contour> cs = find_contours_band_bits(bits.data() + (size_t)k * H * rb, rb, W, 0, H, 1000, true);
            auto t1 = clk::now();
            std::vector<std::array<Pt, 4>> boxes;
            for (auto& c : cs) {
                Pt mb[4]; float msd;
                bool ok = contour_mini_box(c, mb, msd);
                if (r == 0) { n_pts += c.pts.size(); if (ok && msd >= 3) n_box++; }
                if (ok && msd >= 3) boxes.push_back({mb[0], mb[1], mb[2], mb[3]});
            }
            auto t2 = clk::now();
            for (auto& bx : boxes) {   // a11-a12 for every candidate (the pipeline only does it for those that pass the score gate)
                std::vector<Pt> un = unclip(bx.data(), 1.5f);
                Pt bp[4]; float ss;
                bool ok2 = !un.empty() && mini_box(un, bp, ss);
                if (r == 0) { n_un += un.size(); n_fin += ok2; }
            }
            auto t3 = clk::now();
            t_fin += ms(t2, t3);
            if (r == 0) n_c += cs.size();
            t_find += ms(t0, t1); t_mb += ms(t1, t2);
        }
    }
    printf("per page: contours=%.1f kept_points=%.0f boxes=%.1f | find=%.3f ms minibox=%.3f ms | unclip + second mini box=%.3f ms (%.0f vertices per box)\n", (double)n_c / N, (double)n_pts / N,
           (double)n_box / N, t_find / (REP * N), t_mb / (REP * N), t_fin / (REP * N), (double)n_un / std::max<size_t>(n_fin, 1));
}
