// ctc_host.cc -- CTC collapse / string assembly on the host, inside the library (rows a19, a20 of SURVEY 8a).
// The reference does this in Rust right after the argmax (processors/decode.rs:505-614, called from
// models/recognition/crnn.rs:263-293) and the recognition adapter applies the score filter
// (domain/adapters/text_recognition_adapter.rs:60-102).  It is a few microseconds of integer work per region; the
// Python mirror (api.CTCLabelDecode) needed ~10 ms per 1000 regions, which is why it lives here as well.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

struct oar_ctc_dict {
    std::vector<std::string> chars;   // class i -> UTF-8 bytes of its character; [0] = blank ('\0')
};

namespace {
using oar::fail;

// length in bytes of the UTF-8 sequence that starts at s[0] (1 for a malformed lead byte: taken as is)
size_t utf8_len(const unsigned char* s, size_t avail) {
    size_t n = s[0] < 0x80 ? 1 : (s[0] >> 5) == 0x6 ? 2 : (s[0] >> 4) == 0xE ? 3 : (s[0] >> 3) == 0x1E ? 4 : 1;
    return n <= avail ? n : avail;
}
template <typename T>
T* cm(size_t n) {
    T* p = (T*)std::malloc((n ? n : 1) * sizeof(T));
    if (!p) throw std::bad_alloc();
    return p;
}

struct Decoded {   // one sequence
    std::string text;
    float score = 0.0f;
    std::vector<uint32_t> cols;
};

// CTCLabelDecode::decode_argmax_with_positions (decode.rs:549-614) for one sequence
void decode_one(const oar_ctc_dict& d, const int64_t* idx, const float* prob, uint32_t T, Decoded& out) {
    out.text.clear(); out.cols.clear();
    out.text.reserve((size_t)T * 3); out.cols.reserve(T);   // (one allocation each instead of a growth series per text line)
    const int64_t n_chars = (int64_t)d.chars.size();
    int64_t prev = 0;               // blank_index
    float sum = 0.0f;               // filtered_prob.iter().sum::<f32>(): sequential f32
    for (uint32_t t = 0; t < T; ++t) {
        const int64_t i = idx[t];
        if (i != 0 && i != prev && i >= 0 && i < n_chars) {   // character.get(idx) is None past the table
            out.text += d.chars[(size_t)i];
            sum += prob[t];
            out.cols.push_back(t);
        }
        prev = i;                   // unconditional (decode.rs:517-526): an out-of-table index still separates repeats
    }
    out.score = out.cols.empty() ? 0.0f : sum / (float)out.cols.size();
}

void pack(const std::vector<Decoded>& seqs, const std::vector<uint32_t>& Ts, float threshold, oar_text_result* out) {
    std::memset(out, 0, sizeof *out);
    const size_t n = seqs.size();
    size_t bytes = 0, nchar = 0;
    std::vector<uint8_t> keep(n, 1);
    for (size_t i = 0; i < n; ++i) {
        keep[i] = seqs[i].score >= threshold ? 1 : 0;   // the adapter keeps the slot and the score, drops text / positions
        if (keep[i]) { bytes += seqs[i].text.size(); nchar += seqs[i].cols.size(); }
    }
    out->n = (uint32_t)n;
    out->text_offsets = cm<uint64_t>(n + 1);
    out->utf8 = cm<char>(bytes + 1);
    out->scores = cm<float>(n);
    out->char_offsets = cm<uint64_t>(n + 1);
    out->char_cols = cm<uint32_t>(nchar);
    out->char_positions = cm<float>(nchar);
    out->seq_len = cm<uint32_t>(n);
    out->kept = cm<uint8_t>(n);
    size_t b = 0, c = 0;
    for (size_t i = 0; i < n; ++i) {
        out->text_offsets[i] = b; out->char_offsets[i] = c;
        out->scores[i] = seqs[i].score; out->seq_len[i] = Ts[i]; out->kept[i] = keep[i];
        if (!keep[i]) continue;
        std::memcpy(out->utf8 + b, seqs[i].text.data(), seqs[i].text.size());
        b += seqs[i].text.size();
        const float fT = (float)Ts[i];
        for (uint32_t t : seqs[i].cols) { out->char_cols[c] = t; out->char_positions[c] = (float)t / fT; ++c; }
    }
    out->text_offsets[n] = b; out->char_offsets[n] = c;
    out->utf8[b] = 0;
}

// decodes one UTF-8 scalar value starting at s (Rust `char`); malformed input yields the lead byte itself
uint32_t utf8_scalar(const unsigned char* s, size_t avail, size_t& used) {
    used = utf8_len(s, avail);
    if (used == 1) return s[0];
    uint32_t c = used == 2 ? (s[0] & 0x1Fu) : used == 3 ? (s[0] & 0x0Fu) : (s[0] & 0x07u);
    for (size_t i = 1; i < used; ++i) c = (c << 6) | (s[i] & 0x3Fu);
    return c;
}
// OAROCR::is_cjk (src/oarocr/ocr.rs:1075-1082)
bool is_cjk(uint32_t u) {
    return (u >= 0x4E00 && u <= 0x9FFF) || (u >= 0x3400 && u <= 0x4DBF) || (u >= 0x20000 && u <= 0x2A6DF) || (u >= 0x2A700 && u <= 0x2B73F) ||
           (u >= 0x2B740 && u <= 0x2B81F);
}
struct Extent { float x_min, y_min, x_max, y_max; };
// BoundingBox::x_min / y_min / x_max / y_max: f32 fold over the points
Extent extent_of(const float* pts_xy, uint32_t n_points) {
    Extent e{pts_xy[0], pts_xy[1], pts_xy[0], pts_xy[1]};
    for (uint32_t i = 1; i < n_points; ++i) {
        const float x = pts_xy[2 * i], y = pts_xy[2 * i + 1];
        e.x_min = x < e.x_min ? x : e.x_min; e.x_max = x > e.x_max ? x : e.x_max;
        e.y_min = y < e.y_min ? y : e.y_min; e.y_max = y > e.y_max ? y : e.y_max;
    }
    return e;
}
void push_box(std::vector<float>& out, float x1, float y1, float x2, float y2) {   // BoundingBox::from_coords (geometry.rs:98-106)
    const float b[8] = {x1, y1, x2, y1, x2, y2, x1, y2};
    out.insert(out.end(), b, b + 8);
}
// OAROCR::ctc_word_boxes (src/oarocr/ocr.rs:949-1020), f32 arithmetic in the reference's operation order
void ctc_word_boxes(const float* pts_xy, uint32_t n_points, const char* text, size_t text_len, const uint32_t* cols, uint32_t n_cols, uint32_t seq_len,
                    float wh_ratio, float max_wh_ratio, std::vector<float>& out) {
    if (n_cols == 0 || seq_len == 0 || text_len == 0 || n_points == 0) return;
    const float EPS = 1.1920929e-7f;
    const float eff = (float)seq_len * (wh_ratio / max_wh_ratio);
    if (eff <= EPS) return;
    const Extent e = extent_of(pts_xy, n_points);
    const float width = e.x_max - e.x_min;
    const float cell = width / (eff > EPS ? eff : EPS);
    std::vector<uint32_t> chars;
    for (size_t b = 0; b < text_len;) { size_t used; chars.push_back(utf8_scalar((const unsigned char*)text + b, text_len - b, used)); b += used; }
    const float avg = width / (float)(chars.size() ? chars.size() : 1);
    std::vector<float> centers(n_cols);
    for (uint32_t i = 0; i < n_cols; ++i) centers[i] = e.x_min + ((float)cols[i] + 0.5f) * cell;
    for (uint32_t i = 0; i < n_cols; ++i) {
        const uint32_t ch = i < chars.size() ? chars[i] : (uint32_t)'?';
        const float c = centers[i];
        float l, r;
        if (is_cjk(ch)) {
            const float half = avg / 2.0f;
            l = c - half; l = l > e.x_min ? l : e.x_min;
            r = c + half; r = r < e.x_max ? r : e.x_max;
        } else {
            l = i == 0 ? e.x_min : (centers[i - 1] + c) / 2.0f; l = l > e.x_min ? l : e.x_min;
            r = i == n_cols - 1 ? e.x_max : (c + centers[i + 1]) / 2.0f; r = r < e.x_max ? r : e.x_max;
        }
        push_box(out, l, e.y_min, r, e.y_max);
    }
}
// OAROCR::char_positions_to_word_boxes (src/oarocr/ocr.rs:1036-1072): the fallback when no column indices exist
void positions_word_boxes(const float* pts_xy, uint32_t n_points, const float* pos, uint32_t n_pos, uint32_t char_count, std::vector<float>& out) {
    if (n_pos == 0 || char_count == 0 || n_points == 0) return;
    const Extent e = extent_of(pts_xy, n_points);
    const float width = e.x_max - e.x_min;
    const float cw = width / (float)char_count;
    for (uint32_t i = 0; i < n_pos; ++i) {
        const float c = e.x_min + (pos[i] * width);
        float l = c - cw / 2.0f; l = l > e.x_min ? l : e.x_min;
        float r = c + cw / 2.0f; r = r < e.x_max ? r : e.x_max;
        push_box(out, l, e.y_min, r, e.y_max);
    }
}

template <typename F>
oar_status guarded(F&& f) {
    try { f(); return OAR_OK; }
    catch (const oar::Error& e) { oar::set_last_error(e.what()); return e.code; }
    catch (const std::bad_alloc&) { oar::set_last_error("host allocation failed"); return OAR_OOM; }
    catch (const std::exception& e) { oar::set_last_error(e.what()); return OAR_INTERNAL; }
    catch (...) { oar::set_last_error("unknown error"); return OAR_INTERNAL; }
}
}  // namespace

extern "C" {

oar_status oar_ctc_dict_create(const char* dict_utf8, size_t len, int32_t use_space_char, oar_ctc_dict** out) {
    return guarded([&] {
        OAR_CHECK(out && (dict_utf8 || len == 0), OAR_INVALID_INPUT, "oar_ctc_dict_create: bad arguments");
        *out = nullptr;
        std::unique_ptr<oar_ctc_dict> d(new oar_ctc_dict());
        d->chars.push_back(std::string(1, '\0'));   // blank at index 0 (decode.rs:407-408)
        // str::lines(): split at '\n', a trailing '\r' of a line is not part of it; no final empty line
        size_t a = 0;
        while (a < len) {
            size_t e = a;
            while (e < len && dict_utf8[e] != '\n') ++e;
            size_t l = e - a;
            if (l && e < len && dict_utf8[a + l - 1] == '\r') --l;            // "\r\n"
            if (l) {                                                           // filter_map(|s| s.chars().next())
                const size_t cl = utf8_len((const unsigned char*)dict_utf8 + a, l);
                d->chars.push_back(std::string(dict_utf8 + a, cl));
            }
            a = e + 1;
        }
        if (use_space_char) d->chars.push_back(" ");   // after the dictionary characters (decode.rs:125-127); class order: blank, dict..., ' '
        *out = d.release();
    });
}
void oar_ctc_dict_destroy(oar_ctc_dict* d) { delete d; }
uint32_t oar_ctc_dict_classes(const oar_ctc_dict* d) { return d ? (uint32_t)d->chars.size() : 0; }

oar_status oar_ctc_decode(const oar_ctc_dict* dict, const int64_t* indices, const float* probs, uint32_t batch, uint32_t seq_len,
                          float score_threshold, oar_text_result* out) {
    return guarded([&] {
        OAR_CHECK(dict && out && (batch == 0 || seq_len == 0 || (indices && probs)), OAR_INVALID_INPUT, "oar_ctc_decode: bad arguments");
        if (seq_len == 0) batch = 0;   // an empty time axis collapses the batch (decode.rs:465-472, test :747-757)
        std::vector<Decoded> seqs(batch);
        std::vector<uint32_t> Ts(batch, seq_len);
        for (uint32_t b = 0; b < batch; ++b) decode_one(*dict, indices + (size_t)b * seq_len, probs + (size_t)b * seq_len, seq_len, seqs[b]);
        pack(seqs, Ts, score_threshold, out);
    });
}

oar_status oar_ocr_decode(const oar_ctc_dict* dict, const oar_ocr_result* res, float score_threshold, oar_text_result* out) {
    return guarded([&] {
        OAR_CHECK(dict && res && out, OAR_INVALID_INPUT, "oar_ocr_decode: bad arguments");
        // A call's thousand text lines decoded into flat buffers sized from the totals (decode_one + pack gave every line a string and a
        // vector of its own: ~2 200 allocations per call, a third of the 0.3 ms this took); same statements per time step, same outputs.
        const uint32_t n = res->n_regions;
        const oar_ctc_dict& d = *dict;
        const int64_t n_chars = (int64_t)d.chars.size();
        size_t total_T = 0, max_char = 1;
        for (uint32_t k = 0; k < n; ++k) {
            const uint64_t a = res->ctc_offsets[k], b = res->ctc_offsets[k + 1];
            OAR_CHECK(b - a == 0 || b - a == res->seq_len[k], OAR_INTERNAL, "oar_ocr_decode: region CTC length differs from its seq_len");
            total_T += (size_t)(b - a);
        }
        for (const std::string& c : d.chars) max_char = std::max(max_char, c.size());
        std::memset(out, 0, sizeof *out);
        out->n = n;
        out->text_offsets = cm<uint64_t>((size_t)n + 1);
        out->scores = cm<float>(n);
        out->char_offsets = cm<uint64_t>((size_t)n + 1);
        out->seq_len = cm<uint32_t>(n);
        out->kept = cm<uint8_t>(n);
        out->utf8 = cm<char>(total_T * max_char + 1);            // upper bounds: every time step a character
        out->char_cols = cm<uint32_t>(total_T);
        out->char_positions = cm<float>(total_T);
        size_t bpos = 0, cpos = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint64_t a = res->ctc_offsets[k], b = res->ctc_offsets[k + 1];
            const uint32_t T = (uint32_t)(b - a), Tk = res->seq_len[k];
            const int64_t* idx = res->ctc_indices + a;
            const float* prob = res->ctc_probs + a;
            const size_t b0 = bpos, c0 = cpos;
            int64_t prev = 0;               // blank_index
            float sum = 0.0f;               // filtered_prob.iter().sum::<f32>(): sequential f32 (decode.rs:549-614, as decode_one)
            for (uint32_t t = 0; t < T; ++t) {
                const int64_t i = idx[t];
                if (i != 0 && i != prev && i >= 0 && i < n_chars) {
                    const std::string& ch = d.chars[(size_t)i];
                    std::memcpy(out->utf8 + bpos, ch.data(), ch.size());
                    bpos += ch.size();
                    sum += prob[t];
                    out->char_cols[cpos++] = t;
                }
                prev = i;
            }
            const size_t nc = cpos - c0;
            const float score = nc == 0 ? 0.0f : sum / (float)nc;
            const bool keep = score >= score_threshold;   // the adapter keeps the slot and the score, drops text / positions
            out->text_offsets[k] = b0; out->char_offsets[k] = c0;
            out->scores[k] = score; out->seq_len[k] = Tk; out->kept[k] = keep ? 1 : 0;
            if (!keep) { bpos = b0; cpos = c0; continue; }
            const float fT = (float)Tk;
            for (size_t c = c0; c < cpos; ++c) out->char_positions[c] = (float)out->char_cols[c] / fT;
        }
        out->text_offsets[n] = bpos; out->char_offsets[n] = cpos;
        out->utf8[bpos] = 0;
    });
}

oar_status oar_ctc_word_boxes(const float* line_pts_xy, uint32_t n_points, const char* text_utf8, size_t text_len, const uint32_t* col_indices,
                              uint32_t n_cols, uint32_t seq_len, float wh_ratio, float max_wh_ratio, float* boxes, uint32_t cap_boxes, uint32_t* n_boxes) {
    return guarded([&] {
        OAR_CHECK(n_boxes && (n_points == 0 || line_pts_xy) && (text_len == 0 || text_utf8) && (n_cols == 0 || col_indices), OAR_INVALID_INPUT,
                  "oar_ctc_word_boxes: bad arguments");
        std::vector<float> out;
        ctc_word_boxes(line_pts_xy, n_points, text_utf8, text_len, col_indices, n_cols, seq_len, wh_ratio, max_wh_ratio, out);
        *n_boxes = (uint32_t)(out.size() / 8);
        if (boxes) {
            OAR_CHECK(cap_boxes >= *n_boxes, OAR_INVALID_INPUT, "oar_ctc_word_boxes: output buffer too small");
            std::memcpy(boxes, out.data(), out.size() * sizeof(float));
        }
    });
}

oar_status oar_ocr_word_boxes(const oar_ocr_result* res, const oar_text_result* txt, oar_word_boxes* out) {
    return guarded([&] {
        OAR_CHECK(res && txt && out && res->n_regions == txt->n, OAR_INVALID_INPUT, "oar_ocr_word_boxes: result / text mismatch");
        std::memset(out, 0, sizeof *out);
        const uint32_t n = res->n_regions;
        std::vector<float> all;
        std::vector<uint64_t> offs(n + 1, 0);
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t p0 = res->point_offsets ? res->point_offsets[k] : 4 * k, p1 = res->point_offsets ? res->point_offsets[k + 1] : 4 * k + 4;
            const float* pts = res->points + (size_t)p0 * 2;
            const uint64_t c0 = txt->char_offsets[k], c1 = txt->char_offsets[k + 1];
            const uint64_t t0 = txt->text_offsets[k], t1 = txt->text_offsets[k + 1];
            const uint32_t cw = res->crop_wh[2 * k], ch = res->crop_wh[2 * k + 1];
            const float wh = (float)cw / (float)(ch > 1 ? ch : 1);   // PooledRegion::wh_ratio (ocr.rs:739)
            // ocr.rs:860-877: column indices when there are any (and T > 0), else the normalised positions, else none
            if (c1 > c0 && txt->seq_len[k] > 0)
                ctc_word_boxes(pts, p1 - p0, txt->utf8 + t0, (size_t)(t1 - t0), txt->char_cols + c0, (uint32_t)(c1 - c0), txt->seq_len[k], wh, res->max_wh_ratio[k], all);
            offs[k + 1] = all.size() / 8;
        }
        out->n_regions = n;
        out->box_offsets = cm<uint64_t>(n + 1);
        std::memcpy(out->box_offsets, offs.data(), (n + 1) * sizeof(uint64_t));
        out->boxes = cm<float>(all.size());
        std::memcpy(out->boxes, all.data(), all.size() * sizeof(float));
    });
}

oar_status oar_char_positions_to_word_boxes(const float* line_pts_xy, uint32_t n_points, const float* char_positions, uint32_t n_positions,
                                            uint32_t char_count, float* boxes, uint32_t cap_boxes, uint32_t* n_boxes) {
    return guarded([&] {
        OAR_CHECK(n_boxes && (n_points == 0 || line_pts_xy) && (n_positions == 0 || char_positions), OAR_INVALID_INPUT, "oar_char_positions_to_word_boxes: bad arguments");
        std::vector<float> out;
        positions_word_boxes(line_pts_xy, n_points, char_positions, n_positions, char_count, out);
        *n_boxes = (uint32_t)(out.size() / 8);
        if (boxes) {
            OAR_CHECK(cap_boxes >= *n_boxes, OAR_INVALID_INPUT, "oar_char_positions_to_word_boxes: output buffer too small");
            std::memcpy(boxes, out.data(), out.size() * sizeof(float));
        }
    });
}

// ---- multi-process hosts (header: "multi-process hosts"): block partition + the wire format of a rank's final results
oar_status oar_shard_range(uint64_t n_items, uint32_t world_size, uint32_t rank, uint64_t* begin, uint64_t* end) {
    return guarded([&] {
        OAR_CHECK(begin && end && world_size > 0 && rank < world_size, OAR_INVALID_INPUT, "oar_shard_range: bad world_size / rank");
        const uint64_t base = n_items / world_size, rem = n_items % world_size;
        *begin = (uint64_t)rank * base + std::min<uint64_t>(rank, rem);
        *end = *begin + base + (rank < rem ? 1 : 0);
    });
}

oar_status oar_ocr_pack(const oar_ocr_result* res, const oar_text_result* txt, uint8_t** blob, size_t* len) {
    return guarded([&] {
        OAR_CHECK(res && txt && blob && len, OAR_INVALID_INPUT, "oar_ocr_pack: bad arguments");
        OAR_CHECK(txt->n == res->n_regions, OAR_SHAPE_MISMATCH, "oar_ocr_pack: the texts do not belong to this result");
        OAR_CHECK(!res->point_offsets, OAR_INVALID_INPUT, "oar_ocr_pack carries quad boxes only");
        const uint64_t n = res->n_images, nr = res->n_regions, nb = nr ? txt->text_offsets[nr] : 0;
        const size_t total = 24 + 4 * (n + 1) + 32 * nr + 4 * nr + 8 * (nr + 1) + nb;
        uint8_t* b = static_cast<uint8_t*>(std::malloc(total ? total : 1));
        OAR_CHECK(b, OAR_DEVICE, "oar_ocr_pack: out of memory");
        uint8_t* w = b;
        auto put = [&](const void* src, size_t bytes) { if (bytes) std::memcpy(w, src, bytes); w += bytes; };
        const int64_t head[3] = {(int64_t)n, (int64_t)nr, (int64_t)nb};
        put(head, 24);
        if (res->region_offsets) put(res->region_offsets, 4 * (n + 1));
        else { const uint32_t z = 0; for (uint64_t i = 0; i <= n; ++i) put(&z, 4); }
        put(res->points, 32 * nr);
        put(txt->scores, 4 * nr);
        if (txt->text_offsets) put(txt->text_offsets, 8 * (nr + 1));
        else { const uint64_t z = 0; put(&z, 8); }
        put(txt->utf8, nb);
        *blob = b; *len = total;
    });
}

void oar_blob_free(uint8_t* blob) { std::free(blob); }

oar_status oar_packed_merge(const uint8_t* const* blobs, const size_t* lens, uint32_t n_blobs, oar_packed_pages* out) {
    return guarded([&] {
        OAR_CHECK(out && (n_blobs == 0 || (blobs && lens)), OAR_INVALID_INPUT, "oar_packed_merge: bad arguments");
        std::memset(out, 0, sizeof *out);
        uint64_t n = 0, nr = 0, nb = 0;
        for (uint32_t i = 0; i < n_blobs; ++i) {   // validate every header against its blob's length before anything is copied
            OAR_CHECK(blobs[i] && lens[i] >= 24, OAR_INVALID_INPUT, "oar_packed_merge: blob " + std::to_string(i) + " is too short");
            int64_t h[3];
            std::memcpy(h, blobs[i], 24);
            OAR_CHECK(h[0] >= 0 && h[1] >= 0 && h[2] >= 0 && h[0] < (1ll << 31) && h[1] < (1ll << 31) && h[2] < (1ll << 40), OAR_INVALID_INPUT, "oar_packed_merge: corrupt header in blob " + std::to_string(i));
            const uint64_t need = 24 + 4 * ((uint64_t)h[0] + 1) + 36 * (uint64_t)h[1] + 8 * ((uint64_t)h[1] + 1) + (uint64_t)h[2];
            OAR_CHECK(need == lens[i], OAR_INVALID_INPUT, "oar_packed_merge: blob " + std::to_string(i) + " has " + std::to_string(lens[i]) + " bytes, its header says " + std::to_string(need));
            n += (uint64_t)h[0]; nr += (uint64_t)h[1]; nb += (uint64_t)h[2];
        }
        OAR_CHECK(n < (1ull << 31) && nr < (1ull << 31), OAR_INVALID_INPUT, "oar_packed_merge: too many pages / regions");
        out->n_images = (uint32_t)n; out->n_regions = (uint32_t)nr;
        out->region_offsets = cm<uint32_t>(n + 1);
        out->points = cm<float>(nr * 8 + 1);
        out->scores = cm<float>(nr + 1);
        out->text_offsets = cm<uint64_t>(nr + 1);
        out->utf8 = cm<char>(nb + 1);
        uint64_t pi = 0, ri = 0, bi = 0;
        out->region_offsets[0] = 0; out->text_offsets[0] = 0;
        try {
        for (uint32_t i = 0; i < n_blobs; ++i) {
            int64_t h[3];
            std::memcpy(h, blobs[i], 24);
            const uint64_t bn = (uint64_t)h[0], bnr = (uint64_t)h[1], bnb = (uint64_t)h[2];
            const uint8_t* r = blobs[i] + 24;
            for (uint64_t k = 0; k <= bn; ++k) {
                uint32_t v;
                std::memcpy(&v, r + 4 * k, 4);
                OAR_CHECK(v <= bnr && (k == 0 ? v == 0 : true), OAR_INVALID_INPUT, "oar_packed_merge: region offsets out of range in blob " + std::to_string(i));
                if (k) out->region_offsets[pi + k] = (uint32_t)(ri + v);
            }
            r += 4 * (bn + 1);
            std::memcpy(out->points + ri * 8, r, 32 * bnr); r += 32 * bnr;
            std::memcpy(out->scores + ri, r, 4 * bnr); r += 4 * bnr;
            for (uint64_t k = 0; k <= bnr; ++k) {
                uint64_t v;
                std::memcpy(&v, r + 8 * k, 8);
                OAR_CHECK(v <= bnb, OAR_INVALID_INPUT, "oar_packed_merge: text offsets out of range in blob " + std::to_string(i));
                if (k) out->text_offsets[ri + k] = bi + v;
            }
            r += 8 * (bnr + 1);
            std::memcpy(out->utf8 + bi, r, bnb);
            pi += bn; ri += bnr; bi += bnb;
        }
        } catch (...) { oar_packed_pages_free(out); throw; }
        out->utf8[nb] = 0;
    });
}

void oar_packed_pages_free(oar_packed_pages* p) {
    if (!p) return;
    std::free(p->region_offsets); std::free(p->points); std::free(p->scores); std::free(p->text_offsets); std::free(p->utf8);
    std::memset(p, 0, sizeof *p);
}

void oar_word_boxes_free(oar_word_boxes* w) {
    if (!w) return;
    std::free(w->box_offsets); std::free(w->boxes);
    std::memset(w, 0, sizeof *w);
}

void oar_text_result_free(oar_text_result* r) {
    if (!r) return;
    std::free(r->text_offsets); std::free(r->utf8); std::free(r->scores); std::free(r->char_offsets); std::free(r->char_cols);
    std::free(r->char_positions); std::free(r->seq_len); std::free(r->kept);
    std::memset(r, 0, sizeof *r);
}

}  // extern "C"
