// jpeg_decode.cc -- SURVEY 8f-3: JPEG (baseline + progressive Huffman, 8-bit, 1 or 3 components) -> RgbImage.
//
// Reference: load_image_from_memory (oar-ocr-core/src/utils/image.rs:65-68) = image 0.25.6 `load_from_memory` (zune-jpeg underneath)
// + `DynamicImage::to_rgb8`.  JPEG decoding is NOT bit-specified: the IDCT, the chroma upsampling filter and the YCbCr -> RGB
// arithmetic are implementation choices.  zune-jpeg's cannot be pinned here (no Rust toolchain, no vectors in the reference), so this
// decoder restates the de-facto standard instead -- libjpeg / libjpeg-turbo's default decompression path, operation for operation:
//   * entropy decoding: ITU T.81 Annex F / G (sequential and progressive Huffman, restart intervals, EOB runs, successive approximation);
//   * IDCT: jidctint.c `jpeg_idct_islow` (13-bit constants, PASS1_BITS 2; its zero-AC shortcuts are exact, so they are not reproduced);
//   * upsampling: jdsample.c "fancy" triangle filters h2v1 / h2v2 / h1v2 (with jdmainct.c's edge rules: neighbours clamp to the
//     component's REAL rows / columns), pixel replication for the other integral ratios and for components no wider than 2 samples;
//   * colour: jdcolor.c `ycc_rgb_convert` (16-bit fixed point tables), grey replicated, Adobe transform 0 / 'R','G','B' ids = RGB as is.
// PIL decodes through libjpeg-turbo with exactly these defaults, so the pin is EXACT equality with PIL on generated files
// (tests/test_image_decode_cpu.py) -- and, like Triangle resize, "unpinned against the crate the reference links" (expected agreement
// with zune-jpeg: +-1..2 grey levels at chroma edges).
// Two halves: jpeg_entropy_decode (host only: a Huffman stream is serial) yields quantised coefficient planes; the pixel half
// (dequantise + IDCT + upsample + colour) exists twice -- jpeg_render_host below and the HIP kernels of jpeg.hip, bit-identical.
// Not decoded (OAR_UNSUPPORTED_OP, named): arithmetic coding, lossless / hierarchical processes, 12-bit samples, 4-component (CMYK / YCCK).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "jpeg_decode.h"

namespace oar {
namespace img {

namespace {
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool defined = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int32_t maxcode[18];    // largest code of length l (-1: none)
    int32_t valptr[17];
    int32_t mincode[17];
    uint8_t look_len[512];  // 9-bit lookahead: code length (0 = longer than 9 bits)
    uint8_t look_val[512];
    void build() {
        int code = 0, k = 0;
        int huffcode[257];
        uint8_t huffsize[257];
        for (int l = 1; l <= 16; ++l)
            for (int i = 0; i < bits[l]; ++i) huffsize[k++] = (uint8_t)l;
        const int n = k;
        huffsize[n] = 0;
        k = 0;
        int si = n ? huffsize[0] : 0;
        while (k < n) {
            while (k < n && huffsize[k] == si) huffcode[k++] = code++;
            OAR_CHECK(code <= (1 << si), OAR_INVALID_INPUT, "image load: bad JPEG Huffman table");
            code <<= 1; ++si;
        }
        int p = 0;
        for (int l = 1; l <= 16; ++l) {
            if (bits[l]) { valptr[l] = p; mincode[l] = huffcode[p]; p += bits[l]; maxcode[l] = huffcode[p - 1]; }
            else maxcode[l] = -1;
        }
        maxcode[17] = 0x7fffffff;
        std::memset(look_len, 0, sizeof look_len);
        p = 0;
        for (int l = 1; l <= 9; ++l)
            for (int i = 0; i < bits[l]; ++i, ++p) {
                const int first = huffcode[p] << (9 - l);
                for (int c = 0; c < (1 << (9 - l)); ++c) { look_len[first + c] = (uint8_t)l; look_val[first + c] = vals[p]; }
            }
        defined = true;
    }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t buf = 0; int cnt = 0;
    int marker = 0;          // a marker met inside the entropy-coded segment (feeding stops, zeros follow)
    long pad = 0;            // zero bits fed behind a marker / the end of the data (a scan that CONSUMES them has run out of data)
    void fill() {
        while (cnt <= 56) {
            int byte = 0;
            if (!marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    while (p < end && *p == 0xFF) ++p;          // fill bytes
                    if (p < end && *p == 0x00) ++p;             // stuffed zero: a data 0xFF
                    else { marker = p < end ? *p++ : 0xD9; byte = 0; }
                }
            } else if (!marker) {
                marker = 0xD9;   // ran off the data: behave as at EOI
            }
            if (marker) pad += 8;
            buf |= (uint64_t)byte << (56 - cnt);
            cnt += 8;
        }
    }
    inline int peek(int n) { if (cnt < n) fill(); return (int)(buf >> (64 - n)); }
    inline void skip(int n) { buf <<= n; cnt -= n; }
    inline int get(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
    inline int bit() { return get(1); }
    void align_reset() { buf = 0; cnt = 0; pad = 0; }
    // padding bits already consumed: the bits still in the buffer are the most recently fed ones
    long starved() const { return pad > cnt ? pad - cnt : 0; }
    int decode(const Huff& h) {
        if (cnt < 16) fill();
        const int look = (int)(buf >> 55);   // 9 bits
        const int l = h.look_len[look];
        if (l) { skip(l); return h.look_val[look]; }
        int code = (int)(buf >> 54), len = 10;   // 10 bits and up
        while (len <= 16 && code > h.maxcode[len]) { ++len; code = (int)(buf >> (64 - len)); }
        OAR_CHECK(len <= 16, OAR_INVALID_INPUT, "image load: corrupt JPEG data (bad Huffman code)");
        skip(len);
        return h.vals[(h.valptr[len] + code - h.mincode[len]) & 255];
    }
};
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }   // T.81 F.2.2.1 EXTEND

inline uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }
}  // namespace

bool is_jpeg(const uint8_t* b, size_t n) { return n >= 3 && b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF; }

void jpeg_entropy_decode(const uint8_t* b, size_t n, JpegImage& im) {
    OAR_CHECK(is_jpeg(b, n), OAR_INVALID_INPUT, "image load: not a JPEG (no SOI)");
    Huff dc[4], ac[4];
    uint16_t qt[4][64];
    bool qdef[4] = {false, false, false, false};
    bool have_sof = false, progressive = false, jfif = false, seen_scan = false;
    // untrusted input: a progressive file may carry any number of scans, each walking every block.  Legitimate encoders emit ~10 (libjpeg's
    // default script) to a few dozen; the total work is capped as libjpeg-turbo / zune-jpeg cap it (scan count, block visits)
    int n_scans = 0;
    long long block_visits = 0, block_budget = 0;
    int adobe_transform = -1, restart_interval = 0;
    uint8_t comp_id[3] = {0, 0, 0};
    int comp_tq[3] = {0, 0, 0};
    im = JpegImage();
    size_t pos = 2;
    for (;;) {
        // next marker
        while (pos < n && b[pos] != 0xFF) ++pos;            // (garbage between segments is skipped, as libjpeg does with a warning)
        while (pos < n && b[pos] == 0xFF) ++pos;
        OAR_CHECK(pos < n, OAR_INVALID_INPUT, "image load: truncated JPEG (no EOI)");
        const int m = b[pos++];
        if (m == 0xD9) break;                               // EOI
        if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;  // a stuffed FF 00 left over from a scan / TEM / stray RSTn: no payload
        OAR_CHECK(pos + 2 <= n, OAR_INVALID_INPUT, "image load: truncated JPEG segment");
        const size_t len = be16(b + pos);
        OAR_CHECK(len >= 2 && pos + len <= n, OAR_INVALID_INPUT, "image load: truncated JPEG segment");
        const uint8_t* d = b + pos + 2;
        const size_t dl = len - 2;
        if (m == 0xC0 || m == 0xC1 || m == 0xC2) {          // SOF0 baseline / SOF1 extended sequential / SOF2 progressive (Huffman)
            OAR_CHECK(!have_sof, OAR_INVALID_INPUT, "image load: JPEG with more than one frame header");
            OAR_CHECK(dl >= 6, OAR_INVALID_INPUT, "image load: bad JPEG SOF");
            if (d[0] != 8) fail(OAR_UNSUPPORTED_OP, "image load: " + std::to_string((int)d[0]) + "-bit JPEG is not decoded by this library (8-bit is)");
            im.h = be16(d + 1); im.w = be16(d + 3); im.ncomp = d[5];
            OAR_CHECK(im.w > 0 && im.h > 0, OAR_INVALID_INPUT, "image load: JPEG with zero dimensions");
            if (im.ncomp == 4) fail(OAR_UNSUPPORTED_OP, "image load: 4-component (CMYK / YCCK) JPEG is not decoded by this library");
            OAR_CHECK((im.ncomp == 1 || im.ncomp == 3) && dl >= 6 + (size_t)im.ncomp * 3, OAR_INVALID_INPUT, "image load: bad JPEG component count");
            im.hmax = im.vmax = 1;
            for (int c = 0; c < im.ncomp; ++c) {
                comp_id[c] = d[6 + c * 3];
                im.comp[c].h = d[7 + c * 3] >> 4; im.comp[c].v = d[7 + c * 3] & 15; comp_tq[c] = d[8 + c * 3] & 3;
                OAR_CHECK(im.comp[c].h >= 1 && im.comp[c].h <= 4 && im.comp[c].v >= 1 && im.comp[c].v <= 4, OAR_INVALID_INPUT, "image load: bad JPEG sampling factors");
                im.hmax = std::max(im.hmax, im.comp[c].h); im.vmax = std::max(im.vmax, im.comp[c].v);
            }
            if (im.ncomp == 1) { im.comp[0].h = im.comp[0].v = 1; im.hmax = im.vmax = 1; }   // a single component is never subsampled
            for (int c = 0; c < im.ncomp; ++c)
                OAR_CHECK(im.hmax % im.comp[c].h == 0 && im.vmax % im.comp[c].v == 0, OAR_UNSUPPORTED_OP, "image load: JPEG with fractional sampling ratios is not decoded by this library");
            im.mcux = (im.w + 8 * im.hmax - 1) / (8 * im.hmax); im.mcuy = (im.h + 8 * im.vmax - 1) / (8 * im.vmax);
            uint64_t total = 0;
            for (int c = 0; c < im.ncomp; ++c) {
                JpegComp& k = im.comp[c];
                k.bw = im.mcux * k.h; k.bh = im.mcuy * k.v;
                k.dw = (im.w * k.h + im.hmax - 1) / im.hmax; k.dh = (im.h * k.v + im.vmax - 1) / im.vmax;
                total += (uint64_t)k.bw * k.bh * 64 * 2;
            }
            // allocation budget of image::Limits::default() (512 MiB), as for PNG: coefficient planes + the RGB image
            OAR_CHECK(total + (uint64_t)im.w * im.h * 3 <= (512ull << 20), OAR_INVALID_INPUT, "image load: JPEG needs more than the 512 MiB allocation limit (image::Limits::default)");
            for (int c = 0; c < im.ncomp; ++c) im.comp[c].coef.assign((size_t)im.comp[c].bw * im.comp[c].bh * 64, 0);
            progressive = m == 0xC2;
            have_sof = true;
        } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            fail(OAR_UNSUPPORTED_OP, std::string("image load: JPEG process SOF") + std::to_string(m - 0xC0) + " (" + (m >= 0xC9 ? "arithmetic coding" : "lossless / hierarchical") +
                                         ") is not decoded by this library");
        } else if (m == 0xC4) {                             // DHT
            size_t q = 0;
            while (q < dl) {
                OAR_CHECK(q + 17 <= dl, OAR_INVALID_INPUT, "image load: bad JPEG DHT");
                const int tc = d[q] >> 4, th = d[q] & 15;
                OAR_CHECK(tc <= 1 && th <= 3, OAR_INVALID_INPUT, "image load: bad JPEG DHT");
                Huff& h = tc ? ac[th] : dc[th];
                int cnt = 0;
                h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = d[q + l]; cnt += h.bits[l]; }
                OAR_CHECK(cnt <= 256 && q + 17 + cnt <= dl, OAR_INVALID_INPUT, "image load: bad JPEG DHT");
                std::memset(h.vals, 0, sizeof h.vals);
                std::memcpy(h.vals, d + q + 17, cnt);
                h.build();
                q += 17 + cnt;
            }
        } else if (m == 0xDB) {                             // DQT (stored in zigzag order; kept in natural order)
            size_t q = 0;
            while (q < dl) {
                const int pq = d[q] >> 4, tq = d[q] & 15;
                OAR_CHECK(pq <= 1 && tq <= 3 && q + 1 + 64 * (pq + 1) <= dl, OAR_INVALID_INPUT, "image load: bad JPEG DQT");
                for (int i = 0; i < 64; ++i) qt[tq][kZigzag[i]] = pq ? be16(d + q + 1 + i * 2) : d[q + 1 + i];
                qdef[tq] = true;
                q += 1 + 64 * (pq + 1);
            }
        } else if (m == 0xDD) {                             // DRI
            OAR_CHECK(dl >= 2, OAR_INVALID_INPUT, "image load: bad JPEG DRI");
            restart_interval = be16(d);
        } else if (m == 0xE0) {
            if (dl >= 5 && !std::memcmp(d, "JFIF\0", 5)) jfif = true;
        } else if (m == 0xEE) {
            if (dl >= 12 && !std::memcmp(d, "Adobe", 5)) adobe_transform = d[11];
        } else if (m == 0xDA) {                             // SOS + entropy-coded segment
            OAR_CHECK(have_sof && dl >= 1, OAR_INVALID_INPUT, "image load: JPEG scan before the frame header");
            const int ns = d[0];
            OAR_CHECK(ns >= 1 && ns <= im.ncomp && dl >= 1 + (size_t)ns * 2 + 3, OAR_INVALID_INPUT, "image load: bad JPEG SOS");
            // KNOWN LIMITATION: an interleaved scan over a SUBSET of the components has its own MCU geometry (T.81 A.2.3), which this decoder does not
            // implement.  libjpeg-turbo / PIL never write one, but mozjpeg -dc-scan-opt 2 and jpgcrush scripts emit a Cb + Cr interleaved DC scan: such a
            // file is refused here (OAR_UNSUPPORTED_OP, never mis-decoded) where the reference's image crate decodes it.  README "Image formats".
            OAR_CHECK(ns == 1 || ns == im.ncomp, OAR_UNSUPPORTED_OP, "image load: JPEG scan interleaves a subset of the components (unsupported)");
            OAR_CHECK(++n_scans <= 256, OAR_INVALID_INPUT, "image load: JPEG with more than 256 scans");
            if (block_budget == 0) {
                for (int k = 0; k < im.ncomp; ++k) block_budget += (long long)im.comp[k].bw * im.comp[k].bh;
                block_budget = block_budget * 48 + 4096;   // every block visited 48 times: DC + 13 refinements of it, the AC bands and theirs, with room
            }
            int sc[3], td[3], ta[3];
            for (int i = 0; i < ns; ++i) {
                int c = -1;
                for (int k = 0; k < im.ncomp; ++k) if (comp_id[k] == d[1 + i * 2]) c = k;
                OAR_CHECK(c >= 0, OAR_INVALID_INPUT, "image load: JPEG scan names an unknown component");
                for (int j = 0; j < i; ++j) OAR_CHECK(sc[j] != c, OAR_INVALID_INPUT, "image load: JPEG scan repeats a component");
                sc[i] = c; td[i] = d[2 + i * 2] >> 4; ta[i] = d[2 + i * 2] & 15;
                OAR_CHECK(td[i] <= 3 && ta[i] <= 3, OAR_INVALID_INPUT, "image load: bad JPEG SOS table selector");
            }
            const int Ss = d[1 + ns * 2], Se = d[2 + ns * 2], Ah = d[3 + ns * 2] >> 4, Al = d[3 + ns * 2] & 15;
            if (progressive) {
                OAR_CHECK(Ss <= Se && Se <= 63 && Al <= 13 && (Ss == 0 ? Se == 0 : ns == 1) && (Ah == 0 || Ah == Al + 1), OAR_INVALID_INPUT, "image load: bad progressive JPEG scan parameters");
            } else {
                OAR_CHECK(Ss == 0 && Se == 63 && Ah == 0 && Al == 0, OAR_INVALID_INPUT, "image load: bad sequential JPEG scan parameters");
            }
            for (int i = 0; i < ns; ++i) {
                if (!progressive || Ss == 0) { if (!progressive || Ah == 0) OAR_CHECK(dc[td[i]].defined, OAR_INVALID_INPUT, "image load: JPEG scan uses an undefined DC Huffman table"); }
                if (!progressive || Ss > 0) OAR_CHECK(ac[ta[i]].defined, OAR_INVALID_INPUT, "image load: JPEG scan uses an undefined AC Huffman table");
            }
            BitReader br{b + pos + len, b + n};
            int pred[3] = {0, 0, 0};
            int eobrun = 0;
            // one block of one component; (bx, by) in blocks
            auto block = [&](int i, int bx, int by) {
                OAR_CHECK(++block_visits <= block_budget, OAR_INVALID_INPUT, "image load: JPEG scans exceed the decoding budget (too many passes over the image)");
                OAR_CHECK(br.starved() <= 64, OAR_INVALID_INPUT, "image load: truncated JPEG (the entropy-coded data ends before the scan does)");
                JpegComp& k = im.comp[sc[i]];
                int16_t* cf = k.coef.data() + ((size_t)by * k.bw + bx) * 64;
                if (!progressive) {
                    const int s = br.decode(dc[td[i]]);
                    OAR_CHECK(s <= 11, OAR_INVALID_INPUT, "image load: corrupt JPEG data (DC category)");
                    pred[i] += s ? extend(br.get(s), s) : 0;
                    cf[0] = (int16_t)pred[i];
                    for (int kk = 1; kk < 64;) {
                        const int rs = br.decode(ac[ta[i]]), r = rs >> 4, s2 = rs & 15;
                        if (s2 == 0) { if (r != 15) break; kk += 16; continue; }
                        kk += r;
                        OAR_CHECK(kk < 64, OAR_INVALID_INPUT, "image load: corrupt JPEG data (AC run past the block)");
                        cf[kZigzag[kk++]] = (int16_t)extend(br.get(s2), s2);
                    }
                } else if (Ss == 0) {
                    if (Ah == 0) {
                        const int s = br.decode(dc[td[i]]);
                        OAR_CHECK(s <= 11, OAR_INVALID_INPUT, "image load: corrupt JPEG data (DC category)");
                        pred[i] += s ? extend(br.get(s), s) : 0;
                        cf[0] = (int16_t)(pred[i] * (1 << Al));
                    } else if (br.bit()) {
                        cf[0] = (int16_t)(cf[0] | (1 << Al));
                    }
                } else if (Ah == 0) {                       // AC first pass (T.81 G.1.2.2)
                    if (eobrun > 0) { --eobrun; return; }
                    for (int kk = Ss; kk <= Se;) {
                        const int rs = br.decode(ac[ta[i]]), r = rs >> 4, s2 = rs & 15;
                        if (s2 == 0) {
                            if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.get(r); break; }
                            kk += 16;
                        } else {
                            kk += r;
                            OAR_CHECK(kk <= Se, OAR_INVALID_INPUT, "image load: corrupt JPEG data (AC run past the band)");
                            cf[kZigzag[kk++]] = (int16_t)(extend(br.get(s2), s2) * (1 << Al));
                        }
                    }
                } else {                                    // AC refinement (T.81 G.1.2.3; libjpeg jdphuff.c decode_mcu_AC_refine)
                    const int p1 = 1 << Al, m1 = -(1 << Al);
                    int kk = Ss;
                    auto refine = [&](int16_t& c) {
                        if (br.bit() && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
                    };
                    if (eobrun == 0) {
                        while (kk <= Se) {
                            const int rs = br.decode(ac[ta[i]]);
                            int r = rs >> 4, s2 = rs & 15;
                            if (s2) {
                                OAR_CHECK(s2 == 1, OAR_INVALID_INPUT, "image load: corrupt JPEG data (refinement magnitude)");
                                s2 = br.bit() ? p1 : m1;
                            } else if (r != 15) {
                                eobrun = 1 << r;
                                if (r) eobrun += br.get(r);
                                break;
                            }
                            while (kk <= Se) {
                                int16_t& c = cf[kZigzag[kk]];
                                if (c != 0) refine(c);
                                else if (--r < 0) break;
                                ++kk;
                            }
                            if (s2 && kk <= Se) cf[kZigzag[kk]] = (int16_t)s2;
                            ++kk;
                        }
                    }
                    if (eobrun > 0) {
                        for (; kk <= Se; ++kk) {
                            int16_t& c = cf[kZigzag[kk]];
                            if (c != 0) refine(c);
                        }
                        --eobrun;
                    }
                }
            };
            auto restart = [&]() {
                br.align_reset();
                if (!br.marker) {   // the marker has not been pulled into the bit buffer yet: find it
                    while (br.p < br.end && *br.p != 0xFF) ++br.p;
                    while (br.p < br.end && *br.p == 0xFF) ++br.p;
                    if (br.p < br.end) br.marker = *br.p++;
                }
                OAR_CHECK(br.marker >= 0xD0 && br.marker <= 0xD7, OAR_INVALID_INPUT, "image load: corrupt JPEG data (restart marker missing)");
                br.marker = 0;
                pred[0] = pred[1] = pred[2] = 0;
                eobrun = 0;
            };
            long since = 0;
            if (ns == 1) {   // non-interleaved: the component's own blocks, real extent only (T.81 A.2.2)
                const JpegComp& k = im.comp[sc[0]];
                const int rbw = (k.dw + 7) / 8, rbh = (k.dh + 7) / 8;
                for (int by = 0; by < rbh; ++by)
                    for (int bx = 0; bx < rbw; ++bx) {
                        if (restart_interval && since == restart_interval) { restart(); since = 0; }
                        block(0, bx, by);
                        ++since;
                    }
            } else {
                for (int my = 0; my < im.mcuy; ++my)
                    for (int mx = 0; mx < im.mcux; ++mx) {
                        if (restart_interval && since == restart_interval) { restart(); since = 0; }
                        for (int i = 0; i < ns; ++i) {
                            const JpegComp& k = im.comp[sc[i]];
                            for (int vy = 0; vy < k.v; ++vy)
                                for (int hx = 0; hx < k.h; ++hx) block(i, mx * k.h + hx, my * k.v + vy);
                        }
                        ++since;
                    }
            }
            seen_scan = true;
            // continue after the entropy-coded data: at the marker the reader stopped on, or by searching for the next one
            if (br.marker) { pos = (size_t)(br.p - b) - 2; }
            else { pos = (size_t)(br.p - b); }
            continue;
        }
        pos += len;
    }
    OAR_CHECK(have_sof && seen_scan, OAR_INVALID_INPUT, "image load: JPEG without image data");
    for (int c = 0; c < im.ncomp; ++c) {
        OAR_CHECK(qdef[comp_tq[c]], OAR_INVALID_INPUT, "image load: JPEG component uses an undefined quantisation table");
        std::memcpy(im.comp[c].q, qt[comp_tq[c]], sizeof im.comp[c].q);
    }
    // colour space as libjpeg's default_decompress_parms decides it (jdapimin.c)
    if (im.ncomp == 1) im.color = 0;
    else if (jfif) im.color = 1;
    else if (adobe_transform == 0) im.color = 2;
    else if (adobe_transform == 1) im.color = 1;
    else if (comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') im.color = 2;
    else im.color = 1;
}

// ------------------------------------------------------------------------------------------------ pixel half (host)
void jpeg_idct_block(const int16_t* cf, const uint16_t* q, uint8_t* out, int stride) {
    // jidctint.c jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2)
    constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819,
                  F2562 = 20995, F3072 = 25172;
    W32 ws[64];
    for (int x = 0; x < 8; ++x) {
        const W32 i0 = cf[x] * q[x], i1 = cf[8 + x] * q[8 + x], i2 = cf[16 + x] * q[16 + x], i3 = cf[24 + x] * q[24 + x], i4 = cf[32 + x] * q[32 + x], i5 = cf[40 + x] * q[40 + x],
                  i6 = cf[48 + x] * q[48 + x], i7 = cf[56 + x] * q[56 + x];   // |int16 x uint16| < 2^31
        W32 z1 = (i2 + i6) * F0541;
        const W32 t2 = z1 + i6 * (-F1847), t3 = z1 + i2 * F0765;
        const W32 t0 = (i0 + i4).shl(13), t1 = (i0 - i4).shl(13);
        const W32 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        W32 o0 = i7, o1 = i5, o2 = i3, o3 = i1;
        z1 = o0 + o3; W32 z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
        const W32 z5 = (z3 + z4) * F1175;
        o0 *= F0298; o1 *= F2053; o2 *= F3072; o3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
        auto ds = [](W32 v) { return W32((v + (1 << 10)).sra(11)); };
        ws[x] = ds(t10 + o3); ws[56 + x] = ds(t10 - o3); ws[8 + x] = ds(t11 + o2); ws[48 + x] = ds(t11 - o2);
        ws[16 + x] = ds(t12 + o1); ws[40 + x] = ds(t12 - o1); ws[24 + x] = ds(t13 + o0); ws[32 + x] = ds(t13 - o0);
    }
    for (int y = 0; y < 8; ++y) {
        const W32* w = ws + y * 8;
        W32 z1 = (w[2] + w[6]) * F0541;
        const W32 t2 = z1 + w[6] * (-F1847), t3 = z1 + w[2] * F0765;
        const W32 t0 = (w[0] + w[4]).shl(13), t1 = (w[0] - w[4]).shl(13);
        const W32 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        W32 o0 = w[7], o1 = w[5], o2 = w[3], o3 = w[1];
        z1 = o0 + o3; W32 z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
        const W32 z5 = (z3 + z4) * F1175;
        o0 *= F0298; o1 *= F2053; o2 *= F3072; o3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
        auto rl = [](W32 x) { const int v = (x + (1 << 17)).sra(18) + 128; return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        uint8_t* o = out + (size_t)y * stride;
        o[0] = rl(t10 + o3); o[7] = rl(t10 - o3); o[1] = rl(t11 + o2); o[6] = rl(t11 - o2);
        o[2] = rl(t12 + o1); o[5] = rl(t12 - o1); o[3] = rl(t13 + o0); o[4] = rl(t13 - o0);
    }
}

// one output sample of component c at full-resolution pixel (x, y): jdsample.c
uint8_t jpeg_upsampled(const uint8_t* plane, int stride, int dw, int dh, int hs, int vs, int x, int y) {
    const int cx = x / hs, cy = y / vs;
    const bool fancy_h = hs == 2 && dw > 2, fancy_v = vs == 2;
    if (hs == 1 && vs == 1) return plane[(size_t)cy * stride + cx];
    if (hs == 2 && vs == 1 && fancy_h) {           // h2v1_fancy_upsample
        const uint8_t* r = plane + (size_t)cy * stride;
        const int v = r[cx] * 3;
        if ((x & 1) == 0) return cx == 0 ? r[0] : (uint8_t)((v + r[cx - 1] + 1) >> 2);
        return cx == dw - 1 ? r[cx] : (uint8_t)((v + r[cx + 1] + 2) >> 2);
    }
    if (hs == 1 && vs == 2) {                      // h1v2_fancy_upsample (libjpeg-turbo)
        const int ny = (y & 1) == 0 ? std::max(cy - 1, 0) : std::min(cy + 1, dh - 1);
        const int bias = (y & 1) == 0 ? 1 : 2;
        return (uint8_t)((plane[(size_t)cy * stride + cx] * 3 + plane[(size_t)ny * stride + cx] + bias) >> 2);
    }
    if (hs == 2 && vs == 2 && fancy_h && fancy_v) {   // h2v2_fancy_upsample
        const int ny = (y & 1) == 0 ? std::max(cy - 1, 0) : std::min(cy + 1, dh - 1);
        const uint8_t* r0 = plane + (size_t)cy * stride;
        const uint8_t* r1 = plane + (size_t)ny * stride;
        const int cur = r0[cx] * 3 + r1[cx];
        if ((x & 1) == 0) {
            if (cx == 0) return (uint8_t)((cur * 4 + 8) >> 4);
            return (uint8_t)((cur * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4);
        }
        if (cx == dw - 1) return (uint8_t)((cur * 4 + 7) >> 4);
        return (uint8_t)((cur * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4);
    }
    return plane[(size_t)std::min(cy, dh - 1) * stride + std::min(cx, dw - 1)];   // int_upsample / h2v1_upsample / h2v2_upsample: replication
}

void jpeg_ycc_to_rgb(int y, int cb, int cr, uint8_t* o) {
    // jdcolor.c build_ycc_rgb_table + ycc_rgb_convert (SCALEBITS 16)
    const int xb = cb - 128, xr = cr - 128;
    const int r = y + ((91881 * xr + 32768) >> 16);
    const int g = y + ((-22554 * xb + 32768 + (-46802) * xr) >> 16);
    const int bl = y + ((116130 * xb + 32768) >> 16);
    o[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); o[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g); o[2] = (uint8_t)(bl < 0 ? 0 : bl > 255 ? 255 : bl);
}

void jpeg_render_host(const JpegImage& im, std::vector<uint8_t>& rgb) {
    std::vector<uint8_t> plane[3];
    for (int c = 0; c < im.ncomp; ++c) {
        const JpegComp& k = im.comp[c];
        plane[c].assign((size_t)k.bw * 8 * k.bh * 8, 0);
        for (int by = 0; by < k.bh; ++by)
            for (int bx = 0; bx < k.bw; ++bx)
                jpeg_idct_block(k.coef.data() + ((size_t)by * k.bw + bx) * 64, k.q, plane[c].data() + ((size_t)by * 8 * k.bw + bx) * 8, k.bw * 8);
    }
    rgb.assign((size_t)im.w * im.h * 3, 0);
    for (uint32_t y = 0; y < im.h; ++y)
        for (uint32_t x = 0; x < im.w; ++x) {
            uint8_t s[3] = {0, 0, 0};
            for (int c = 0; c < im.ncomp; ++c) {
                const JpegComp& k = im.comp[c];
                s[c] = jpeg_upsampled(plane[c].data(), k.bw * 8, k.dw, k.dh, im.hmax / k.h, im.vmax / k.v, (int)x, (int)y);
            }
            uint8_t* o = rgb.data() + ((size_t)y * im.w + x) * 3;
            if (im.color == 0) o[0] = o[1] = o[2] = s[0];
            else if (im.color == 2) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
            else jpeg_ycc_to_rgb(s[0], s[1], s[2], o);
        }
}

}  // namespace img
}  // namespace oar
