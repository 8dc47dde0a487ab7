// Mutation fuzzer for the host image decoders (image_decode.cc: PNG; jpeg_decode.cc: JPEG entropy decode + host renderer;
// image_misc_decode.cc: BMP / PNM / TIFF / GIF) -- the code that parses untrusted bytes behind oar_image_decode.  Built with
// -fsanitize=address,undefined (tests/test_image_fuzz_cpu.py runs a short batch; run it longer by hand:
//   python tools/fuzz/make_seeds.py /tmp/seeds && g++ ... && ./fuzz 1000000 1 /tmp/seeds/*).
// A decoder may reject (oar::Error) or decode; anything else -- a sanitizer report, bad_alloc, a size mismatch -- aborts.
// PNG chunk CRCs are repaired after mutation (4 times in 5) so that mutations reach the inflate / unfilter / palette code.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>

#include "common.h"
#include "jpeg_decode.h"
namespace oar { namespace img {
bool is_png(const uint8_t* b, size_t n);
void decode_png(const uint8_t* b, size_t n, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height);
bool decode_misc(const uint8_t* b, size_t n, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height);
} }

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {});
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: fuzz <iterations> <seed> <files...>\n"); return 2; }
    const long iters = atol(argv[1]);
    std::mt19937 rng((unsigned)atoi(argv[2]));
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 3; i < argc; ++i) seeds.push_back(slurp(argv[i]));
    long ok = 0, rejected = 0, unknown = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> d = seeds[rng() % seeds.size()];
        const int nmut = 1 + rng() % 6;
        for (int m = 0; m < nmut && !d.empty(); ++m) {
            switch (rng() % 6) {
                case 0: d[rng() % d.size()] = (uint8_t)rng(); break;
                case 1: d[rng() % std::min<size_t>(d.size(), 64)] = (uint8_t)rng(); break;   // header bias
                case 2: d.resize(rng() % (d.size() + 1)); break;                             // truncation
                case 3: {                                                                    // a 32-bit field: extremes or noise
                    const size_t p = rng() % d.size();
                    const uint32_t v = (rng() % 4 == 0) ? 0xffffffffu : (rng() % 3 == 0 ? 0x7fffffffu : (uint32_t)rng());
                    for (int k = 0; k < 4 && p + k < d.size(); ++k) d[p + k] = (uint8_t)(v >> (8 * k));
                    break;
                }
                case 4: d[rng() % d.size()] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 5: {                                                                    // splice
                    const size_t p = rng() % d.size(), q = rng() % d.size();
                    const size_t len = std::min((size_t)(rng() % 32), std::min(d.size() - p, d.size() - q));
                    memmove(&d[p], &d[q], len);
                    break;
                }
            }
        }
        if (d.size() > 8 && oar::img::is_png(d.data(), d.size()) && rng() % 5) {
            size_t p = 8;
            while (p + 12 <= d.size()) {
                const uint32_t len = (uint32_t)d[p] << 24 | d[p + 1] << 16 | d[p + 2] << 8 | d[p + 3];
                if (len > d.size() || p + 12 + len > d.size()) break;
                const uint32_t c = (uint32_t)crc32(0, &d[p + 4], 4 + len);
                d[p + 8 + len] = (uint8_t)(c >> 24); d[p + 9 + len] = (uint8_t)(c >> 16); d[p + 10 + len] = (uint8_t)(c >> 8); d[p + 11 + len] = (uint8_t)c;
                p += 12 + len;
            }
        }
        std::vector<uint8_t> rgb;
        uint32_t w = 0, h = 0;
        try {
            bool mine = true;
            if (oar::img::is_png(d.data(), d.size())) oar::img::decode_png(d.data(), d.size(), rgb, w, h);
            else if (oar::img::is_jpeg(d.data(), d.size())) {
                oar::img::JpegImage im;
                oar::img::jpeg_entropy_decode(d.data(), d.size(), im);
                oar::img::jpeg_render_host(im, rgb);
                w = im.w; h = im.h;
            } else mine = oar::img::decode_misc(d.data(), d.size(), rgb, w, h);
            if (!mine) { ++unknown; continue; }
            if (rgb.size() != (size_t)w * h * 3) { printf("decoded size mismatch: %zu bytes for %u x %u\n", rgb.size(), w, h); abort(); }
            ++ok;
        } catch (const oar::Error&) {
            ++rejected;
        } catch (const std::bad_alloc&) {
            printf("bad_alloc: an allocation escaped the decoding budget\n");
            abort();
        }
    }
    printf("decoded %ld rejected %ld unknown-format %ld\n", ok, rejected, unknown);
    return 0;
}
