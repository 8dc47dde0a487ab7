// ctc_host.cc -- CTC collapse / string assembly on the host, inside the library (rows a19, a20 of SURVEY 8a).
// The reference does this in Rust right after the argmax (processors/decode.rs:505-614, called from
// models/recognition/crnn.rs:263-293) and the recognition adapter applies the score filter
// (domain/adapters/text_recognition_adapter.rs:60-102).  It is a few microseconds of integer work per region; the
// Python mirror (api.CTCLabelDecode) needed ~10 ms per 1000 regions, which is why it lives here as well.
#include <cstdlib>
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
        const uint32_t n = res->n_regions;
        std::vector<Decoded> seqs(n);
        std::vector<uint32_t> Ts(n, 0);
        for (uint32_t k = 0; k < n; ++k) {
            const uint64_t a = res->ctc_offsets[k], b = res->ctc_offsets[k + 1];
            Ts[k] = res->seq_len[k];
            OAR_CHECK(b - a == 0 || b - a == Ts[k], OAR_INTERNAL, "oar_ocr_decode: region CTC length differs from its seq_len");
            if (b > a) decode_one(*dict, res->ctc_indices + a, res->ctc_probs + a, (uint32_t)(b - a), seqs[k]);
        }
        pack(seqs, Ts, score_threshold, out);
    });
}

void oar_text_result_free(oar_text_result* r) {
    if (!r) return;
    std::free(r->text_offsets); std::free(r->utf8); std::free(r->scores); std::free(r->char_offsets); std::free(r->char_cols);
    std::free(r->char_positions); std::free(r->seq_len); std::free(r->kept);
    std::memset(r, 0, sizeof *r);
}

}  // extern "C"
