// coder.cpp — host-side entropy coder behind the C ABI of include/onedc_coder.h.
//
// Byte-wise rANS (32-bit state, lower bound 2^23, 16-bit probabilities) with a
// 2-bit bypass escape for out-of-table symbols.  The wire format and the
// symbol/escape semantics are those of the reference coder
// (src/cpp/rans/rans.cpp:101-187 encode, :303-362 decode; rans_byte.h), so
// streams are byte-identical for identical inputs.  The implementation is our
// own: flat tables, a bucketed cumulative-frequency lookup instead of the
// reference's linear scan (rans.cpp:326-329), bounds-checked stream reads, and
// a batch entry point that drives independent streams on a thread pool.
#include "../../include/onedc_coder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t kProbBits = 16;
constexpr uint32_t kProbMask = (1u << kProbBits) - 1;
constexpr uint32_t kStateLow = 1u << 23;  // renormalisation lower bound
constexpr uint32_t kBypassBits = 2;
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1;
constexpr int kLutShift = 8;  // cumulative-frequency buckets of 256
constexpr int kLutSize = 1 << (kProbBits - kLutShift);

// One group of CDF tables as handed to add_cdf: a dense [n_tables][row_len] block.
struct TableGroup {
    int n_tables = 0;
    int row_len = 0;
    std::vector<int32_t> cdf;      // [n_tables][row_len]
    std::vector<int32_t> size;     // entries used per table (symbols = size - 1, escape = size - 2)
    std::vector<int32_t> offset;   // value of symbol 0
    std::vector<uint8_t> lut;      // decoder only: [n_tables][kLutSize] first candidate symbol
    // encoder only: [n_tables][row_len] (start, frequency, reciprocal) of every symbol, so that coding a symbol is one
    // 12-byte copy when it is pushed and a multiply-high instead of a 32-bit division when the stream is written
    struct EncSym {
        uint16_t start;
        uint16_t range;   // frequency; 0 marks a raw bypass digit held in `start`
        uint32_t rcp;     // ceil(2^(31 + shift) / range), shift = ceil(log2 range)   (range >= 2)
        uint32_t shift;   // x / range == mulhi(x, rcp) >> (shift - 1) for every x < 2^31 (Alverson; rans_byte.h uses the same)
    };
    std::vector<EncSym> enc_sym;

    static EncSym make_enc_sym(uint32_t start, uint32_t range) {
        EncSym e{(uint16_t)start, (uint16_t)range, 0u, 0u};
        if (range >= 2) {
            uint32_t sh = 0;
            while (range > (1u << sh)) ++sh;
            e.rcp = (uint32_t)(((1ull << (sh + 31)) + range - 1) / range);
            e.shift = sh - 1;
        }
        return e;
    }

    int load(const int32_t *cdfs, int n, int len, const int32_t *sizes, const int32_t *offsets,
             bool with_lut) {
        if (!cdfs || !sizes || !offsets || n <= 0 || len < 2) return ONEDC_ERR_ARG;
        for (int t = 0; t < n; ++t) {
            if (sizes[t] < 2 || sizes[t] > len) return ONEDC_ERR_ARG;
            // a usable row starts at 0, rises strictly (every symbol has a non-zero frequency: the encoder divides by
            // it, the decoder's search stops on it) and ends at 2^16 -- otherwise `while (c[s + 1] <= cum) ++s` below
            // could walk past the row on input nobody validated
            const int32_t *c = cdfs + (size_t)t * len;
            if (c[0] != 0 || c[sizes[t] - 1] != (int32_t)(1u << kProbBits)) return ONEDC_ERR_ARG;
            for (int k = 0; k + 1 < sizes[t]; ++k)
                if (c[k] >= c[k + 1]) return ONEDC_ERR_ARG;
        }
        n_tables = n;
        row_len = len;
        cdf.assign(cdfs, cdfs + (size_t)n * len);
        size.assign(sizes, sizes + n);
        offset.assign(offsets, offsets + n);
        if (!with_lut) {
            enc_sym.assign((size_t)n * len, EncSym{0, 0, 0, 0});
            for (int t = 0; t < n; ++t) {
                const int32_t *c = &cdf[(size_t)t * len];
                for (int k = 0; k + 1 < size[t]; ++k)
                    enc_sym[(size_t)t * len + k] = make_enc_sym((uint32_t)c[k], (uint32_t)(c[k + 1] - c[k]));
            }
        }
        if (with_lut) {
            if (len > 257) return ONEDC_ERR_ARG;  // lut entries are uint8
            lut.resize((size_t)n * kLutSize);
            for (int t = 0; t < n; ++t) {
                const int32_t *c = &cdf[(size_t)t * len];
                const int nsym = size[t] - 1;
                int s = 0;
                for (int b = 0; b < kLutSize; ++b) {
                    const int32_t lo = b << kLutShift;
                    while (s + 1 < nsym && c[s + 1] <= lo) ++s;
                    lut[(size_t)t * kLutSize + b] = (uint8_t)s;
                }
            }
        }
        return ONEDC_OK;
    }
};

typedef TableGroup::EncSym PendingSym;

// ---- single-stream encoder --------------------------------------------------
// The pending symbols of a stream: a plain growable array that is never value-initialised (std::vector::resize would
// zero 12 bytes per symbol before the loop below overwrites them)
struct PendingBuf {
    PendingSym *d = nullptr;
    size_t n = 0, cap = 0;
    PendingBuf() = default;
    PendingBuf(const PendingBuf &) = delete;
    PendingBuf &operator=(const PendingBuf &) = delete;
    PendingBuf(PendingBuf &&o) noexcept : d(o.d), n(o.n), cap(o.cap) { o.d = nullptr; o.n = o.cap = 0; }
    ~PendingBuf() { free(d); }
    bool room(size_t total) {                      // capacity for `total` entries, contents kept
        if (total <= cap) return true;
        size_t c = std::max(total, cap + cap / 2 + 1024);
        auto *nd = (PendingSym *)realloc(d, c * sizeof(PendingSym));
        if (!nd) return false;
        d = nd;
        cap = c;
        return true;
    }
    size_t size() const { return n; }
    void clear() { n = 0; }
};

struct EncPart {
    PendingBuf pending;
    std::vector<uint8_t> stream;
    bool oom = false;

    void push(const TableGroup &g, const int16_t *symbols, const int16_t *indexes, int64_t n) {
        // one slot per symbol up front (the common case writes through a raw pointer); an escape appends its bypass
        // digits through the slow path, which makes room first
        size_t w = pending.size();
        if (!pending.room(w + (size_t)n)) {
            oom = true;
            return;
        }
        PendingSym *out = pending.d;
        const PendingSym *table = g.enc_sym.data();
        const int row = g.row_len;
        for (int64_t i = 0; i < n; ++i) {
            const int t = indexes[i];
            if (t < 0) continue;
            const int32_t escape = g.size[t] - 2;
            int32_t v = (int32_t)symbols[i] - g.offset[t];
            if ((uint32_t)v < (uint32_t)escape) {               // 0 <= v < escape
                out[w++] = table[(size_t)t * row + v];
                continue;
            }
            uint32_t raw;
            if (v < 0) raw = (uint32_t)(-2 * v - 1);
            else raw = (uint32_t)(2 * (v - escape));
            int digits = 0;
            while ((raw >> (digits * kBypassBits)) != 0) ++digits;
            const size_t need = 2 + (size_t)digits / kBypassMax + (size_t)digits;          // escape + count run + digits
            if (w + need + (size_t)(n - i) > pending.cap) {
                pending.n = w;
                if (!pending.room(w + need + (size_t)(n - i) + 64)) {
                    oom = true;
                    return;
                }
                out = pending.d;
            }
            out[w++] = table[(size_t)t * row + escape];
            int rem = digits;  // unary-ish count: runs of kBypassMax then the remainder
            while (rem >= (int)kBypassMax) {
                out[w++] = PendingSym{(uint16_t)kBypassMax, 0, 0, 0};
                rem -= kBypassMax;
            }
            out[w++] = PendingSym{(uint16_t)rem, 0, 0, 0};
            for (int d = 0; d < digits; ++d)
                out[w++] = PendingSym{(uint16_t)((raw >> (d * kBypassBits)) & kBypassMax), 0, 0, 0};
        }
        pending.n = w;
    }

    void flush() {
        // every pending entry emits at most two bytes, plus four for the final state
        std::vector<uint8_t> buf(pending.size() * 2 + 8);
        uint8_t *end = buf.data() + buf.size();
        uint8_t *p = end;
        uint32_t x = kStateLow;
        const PendingSym *sym = pending.d;
        for (size_t k = pending.size(); k-- > 0;) {
            const PendingSym s = sym[k];
            if (s.range >= 2) {
                const uint32_t freq = s.range;
                const uint32_t x_max = freq << 15;  // ((L >> 16) << 8) * freq
                while (x >= x_max) {
                    *--p = (uint8_t)(x & 0xff);
                    x >>= 8;
                }
                const uint32_t q = (uint32_t)(((uint64_t)x * s.rcp) >> 32) >> s.shift;      // x / freq, exact for x < 2^31
                x = (q << kProbBits) + (x - q * freq) + s.start;
            } else if (s.range == 1) {
                while (x >= (1u << 15)) {
                    *--p = (uint8_t)(x & 0xff);
                    x >>= 8;
                }
                x = (x << kProbBits) + s.start;
            } else {
                const uint32_t x_max = (1u << (kProbBits - kBypassBits)) << 15;
                while (x >= x_max) {
                    *--p = (uint8_t)(x & 0xff);
                    x >>= 8;
                }
                x = (x << kBypassBits) | s.start;
            }
        }
        p -= 4;
        p[0] = (uint8_t)(x >> 0);
        p[1] = (uint8_t)(x >> 8);
        p[2] = (uint8_t)(x >> 16);
        p[3] = (uint8_t)(x >> 24);
        stream.assign(p, end);
    }
};

// ---- single-stream decoder --------------------------------------------------
struct DecPart {
    std::vector<uint8_t> bytes;
    size_t pos = 0;
    uint32_t x = 0;
    bool ready = false;
    bool overrun = false;
    bool malformed = false;   // an escape announced more bypass digits than an int16 symbol can need

    inline uint32_t next_byte() {
        if (pos < bytes.size()) return bytes[pos++];
        overrun = true;
        ++pos;
        return 0;
    }
    void begin(const uint8_t *p, size_t n) {
        bytes.assign(p, p + n);
        pos = 0;
        overrun = false;
        malformed = false;
        x = next_byte();
        x |= next_byte() << 8;
        x |= next_byte() << 16;
        x |= next_byte() << 24;
        ready = true;
    }
    inline uint32_t raw_bits() {
        const uint32_t v = x & kBypassMax;
        x >>= kBypassBits;
        if (x < kStateLow) x = (x << 8) | next_byte();
        return v;
    }
    void pull(const TableGroup &g, const int16_t *indexes, int64_t n, int16_t *out) {
        for (int64_t i = 0; i < n; ++i) {
            const int t = indexes[i];
            if (t < 0) {
                out[i] = 0;
                continue;
            }
            const int32_t *c = &g.cdf[(size_t)t * g.row_len];
            const int32_t escape = g.size[t] - 2;
            const uint32_t cum = x & kProbMask;
            int s = g.lut[(size_t)t * kLutSize + (cum >> kLutShift)];
            while ((uint32_t)c[s + 1] <= cum) ++s;
            const uint32_t start = (uint32_t)c[s];
            const uint32_t freq = (uint32_t)c[s + 1] - start;
            x = freq * (x >> kProbBits) + cum - start;
            while (x < kStateLow) x = (x << 8) | next_byte();
            int32_t v = s;
            if (v == escape) {
                uint32_t d = raw_bits();
                int32_t digits = (int32_t)d;
                while (d == kBypassMax && digits <= 16) {
                    d = raw_bits();
                    digits += (int32_t)d;
                }
                // int16 symbols need at most 9 two-bit digits; 16 digits fill the 32-bit word (a larger count would
                // shift by >= 32: undefined) -- anything beyond is a corrupt or desynchronised stream
                if (digits > 16) {
                    malformed = true;
                    for (int64_t r = i; r < n; ++r) out[r] = 0;
                    return;
                }
                uint32_t raw = 0;
                for (int32_t k = 0; k < digits; ++k) raw |= raw_bits() << (k * kBypassBits);
                v = (int32_t)(raw >> 1);
                if (raw & 1)
                    v = -v - 1;
                else
                    v += escape;
            }
            out[i] = (int16_t)(v + g.offset[t]);
        }
    }
};

template <class F>
void parallel_for(int count, int n_threads, F &&body) {
    n_threads = std::max(1, std::min(n_threads, count));
    if (n_threads == 1) {
        for (int i = 0; i < count; ++i) body(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    pool.reserve(n_threads - 1);
    auto run = [&] {
        for (int i; (i = next.fetch_add(1)) < count;) body(i);
    };
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(run);
    run();
    for (auto &th : pool) th.join();
}

}  // namespace

struct onedc_rans_enc {
    std::vector<EncPart> parts;
    std::vector<std::shared_ptr<TableGroup>> groups;
    std::vector<uint8_t> framed;
    bool flushed = false;
};

struct onedc_rans_dec {
    std::vector<DecPart> parts;
    std::vector<std::shared_ptr<TableGroup>> groups;
};

extern "C" {

const char *onedc_coder_version(void) { return "onedc-coder 0.2"; }
#ifndef ONEDC_BUILD_ID
#define ONEDC_BUILD_ID "unidentified"
#endif
static const char onedc_build_id_marker[] = "@(#)onedc-build-id:" ONEDC_BUILD_ID;      // (found in the file by build.py)
const char *onedc_coder_build_id(void) { return onedc_build_id_marker + 19; }

onedc_rans_enc *onedc_rans_enc_create(int /*multi_thread*/, int stream_part) {
    if (stream_part < 1 || stream_part > 16) return nullptr;
    auto *e = new (std::nothrow) onedc_rans_enc;
    if (e) e->parts.resize(stream_part);
    return e;
}
void onedc_rans_enc_destroy(onedc_rans_enc *enc) { delete enc; }

int onedc_rans_enc_add_cdf(onedc_rans_enc *enc, const int32_t *cdfs, int n_tables, int row_len,
                           const int32_t *sizes, const int32_t *offsets) {
    if (!enc) return ONEDC_ERR_ARG;
    auto g = std::make_shared<TableGroup>();
    const int rc = g->load(cdfs, n_tables, row_len, sizes, offsets, false);
    if (rc != ONEDC_OK) return rc;
    enc->groups.push_back(g);
    return (int)enc->groups.size() - 1;
}
int onedc_rans_enc_empty_cdf_buffer(onedc_rans_enc *enc) {
    if (!enc) return ONEDC_ERR_ARG;
    enc->groups.clear();
    return ONEDC_OK;
}
int onedc_rans_enc_reset(onedc_rans_enc *enc) {
    if (!enc) return ONEDC_ERR_ARG;
    for (auto &p : enc->parts) {
        p.pending.clear();
        p.oom = false;                                      // a failed allocation does not outlive the symbols it was for
    }
    enc->flushed = false;
    return ONEDC_OK;
}
int onedc_rans_enc_encode_with_indexes(onedc_rans_enc *enc, const int16_t *symbols,
                                       const int16_t *indexes, int64_t n, int group) {
    if (!enc || n < 0 || (n > 0 && (!symbols || !indexes))) return ONEDC_ERR_ARG;
    if (group < 0 || group >= (int)enc->groups.size()) return ONEDC_ERR_ARG;
    const TableGroup &g = *enc->groups[group];
    for (int64_t i = 0; i < n; ++i)
        if (indexes[i] >= g.n_tables) return ONEDC_ERR_ARG;
    // contiguous split over the parts, the last part takes the remainder (py_rans.cpp:31-36)
    const int np = (int)enc->parts.size();
    const int64_t each = n / np;
    size_t at_entry[16];                                     // (stream parts: a 4-bit field of the flag byte)
    for (int p = 0; p < np && p < 16; ++p) at_entry[p] = enc->parts[p].pending.n;
    for (int p = 0; p < np; ++p) {
        const int64_t cnt = p < np - 1 ? each : n - each * (np - 1);
        enc->parts[p].push(g, symbols + p * each, indexes + p * each, cnt);
        if (enc->parts[p].oom) {                             // out of memory for the pending symbols: the call did not happen
            for (int q = 0; q < np && q < 16; ++q) {         // -- every part back to where it stood, the flag cleared
                enc->parts[q].pending.n = at_entry[q];
                enc->parts[q].oom = false;
            }
            return ONEDC_ERR_CAPACITY;
        }
    }
    return ONEDC_OK;
}
int onedc_rans_enc_flush(onedc_rans_enc *enc) {
    if (!enc) return ONEDC_ERR_ARG;
    for (auto &p : enc->parts) p.flush();
    // framing: flag byte, sizes of all but the last part, then the part streams
    const int np = (int)enc->parts.size();
    size_t total = 0, widest = 0;
    for (int p = 0; p < np; ++p) {
        total += enc->parts[p].stream.size();
        if (p < np - 1) widest = std::max(widest, enc->parts[p].stream.size());
    }
    const int size_bytes = widest > 65535 ? 4 : 2;
    const size_t head = 1 + (size_t)(np - 1) * size_bytes;
    enc->framed.resize(head + total);
    uint8_t *o = enc->framed.data();
    o[0] = (uint8_t)(((np - 1) << 4) + (size_bytes == 2 ? 1 : 0));
    for (int p = 0; p < np - 1; ++p) {
        const uint32_t sz = (uint32_t)enc->parts[p].stream.size();
        if (size_bytes == 2) {
            const uint16_t s16 = (uint16_t)sz;
            memcpy(o + 1 + 2 * p, &s16, 2);
        } else {
            memcpy(o + 1 + 4 * p, &sz, 4);
        }
    }
    size_t at = head;
    for (int p = 0; p < np; ++p) {
        const auto &s = enc->parts[p].stream;
        if (!s.empty()) memcpy(o + at, s.data(), s.size());
        at += s.size();
    }
    enc->flushed = true;
    return ONEDC_OK;
}
int64_t onedc_rans_enc_stream_size(onedc_rans_enc *enc) {
    if (!enc) return ONEDC_ERR_ARG;
    if (!enc->flushed) return ONEDC_ERR_STATE;
    return (int64_t)enc->framed.size();
}
int onedc_rans_enc_get_stream(onedc_rans_enc *enc, uint8_t *out, int64_t capacity) {
    if (!enc || !out) return ONEDC_ERR_ARG;
    if (!enc->flushed) return ONEDC_ERR_STATE;
    if (capacity < (int64_t)enc->framed.size()) return ONEDC_ERR_CAPACITY;
    memcpy(out, enc->framed.data(), enc->framed.size());
    return ONEDC_OK;
}

onedc_rans_dec *onedc_rans_dec_create(int stream_part) {
    if (stream_part < 1 || stream_part > 16) return nullptr;
    auto *d = new (std::nothrow) onedc_rans_dec;
    if (d) d->parts.resize(stream_part);
    return d;
}
void onedc_rans_dec_destroy(onedc_rans_dec *dec) { delete dec; }

int onedc_rans_dec_add_cdf(onedc_rans_dec *dec, const int32_t *cdfs, int n_tables, int row_len,
                           const int32_t *sizes, const int32_t *offsets) {
    if (!dec) return ONEDC_ERR_ARG;
    auto g = std::make_shared<TableGroup>();
    const int rc = g->load(cdfs, n_tables, row_len, sizes, offsets, true);
    if (rc != ONEDC_OK) return rc;
    dec->groups.push_back(g);
    return (int)dec->groups.size() - 1;
}
int onedc_rans_dec_empty_cdf_buffer(onedc_rans_dec *dec) {
    if (!dec) return ONEDC_ERR_ARG;
    dec->groups.clear();
    return ONEDC_OK;
}
int onedc_rans_dec_set_stream(onedc_rans_dec *dec, const uint8_t *stream, int64_t n) {
    if (!dec || !stream || n < 1) return ONEDC_ERR_ARG;
    const int np = (stream[0] >> 4) + 1;
    if (np != (int)dec->parts.size()) return ONEDC_ERR_STREAM;
    const int size_bytes = (stream[0] & 0x0f) == 1 ? 2 : 4;
    int64_t at = 1;
    std::vector<int64_t> sizes;
    int64_t known = 0;
    for (int p = 0; p < np - 1; ++p) {
        if (at + size_bytes > n) return ONEDC_ERR_STREAM;
        uint32_t sz = 0;
        if (size_bytes == 2) {
            uint16_t s16;
            memcpy(&s16, stream + at, 2);
            sz = s16;
        } else {
            memcpy(&sz, stream + at, 4);
        }
        at += size_bytes;
        sizes.push_back(sz);
        known += sz;
    }
    if (n - at - known < 0) return ONEDC_ERR_STREAM;
    sizes.push_back(n - at - known);
    for (int p = 0; p < np; ++p) {
        dec->parts[p].begin(stream + at, (size_t)sizes[p]);
        at += sizes[p];
    }
    return ONEDC_OK;
}
int onedc_rans_dec_decode_stream(onedc_rans_dec *dec, const int16_t *indexes, int64_t n, int group,
                                 int16_t *out) {
    if (!dec || n < 0 || (n > 0 && (!indexes || !out))) return ONEDC_ERR_ARG;
    if (group < 0 || group >= (int)dec->groups.size()) return ONEDC_ERR_ARG;
    const TableGroup &g = *dec->groups[group];
    for (int64_t i = 0; i < n; ++i)
        if (indexes[i] >= g.n_tables) return ONEDC_ERR_ARG;
    const int np = (int)dec->parts.size();
    for (int p = 0; p < np; ++p)
        if (!dec->parts[p].ready) return ONEDC_ERR_STATE;
    const int64_t each = n / np;
    bool overrun = false;
    for (int p = 0; p < np; ++p) {
        const int64_t cnt = p < np - 1 ? each : n - each * (np - 1);
        dec->parts[p].pull(g, indexes + p * each, cnt, out + p * each);
        overrun |= dec->parts[p].overrun | dec->parts[p].malformed;
    }
    return overrun ? ONEDC_ERR_STREAM : ONEDC_OK;
}

int onedc_rans_dec_finished(onedc_rans_dec *dec) {
    if (!dec) return ONEDC_ERR_ARG;
    for (const DecPart &part : dec->parts) {
        if (!part.ready) return ONEDC_ERR_STATE;
        // the decoder walks the encoder's states backwards: after the last symbol it stands on the encoder's initial
        // state with every byte consumed.  Anything else means some (index, symbol) decision differed from the encoder's.
        if (part.overrun || part.malformed || part.pos != part.bytes.size() || part.x != kStateLow)
            return ONEDC_ERR_STREAM;
    }
    return ONEDC_OK;
}

int onedc_rans_dec_decode_batch(onedc_rans_dec *const *decs, int batch, const int16_t *indexes,
                                int64_t n, int group, int16_t *out, int n_threads, int *status) {
    if (!decs || batch < 0 || n < 0) return ONEDC_ERR_ARG;
    std::vector<int> rc(batch, ONEDC_OK);
    parallel_for(batch, n_threads, [&](int b) {
        rc[b] = onedc_rans_dec_decode_stream(decs[b], indexes + (size_t)b * n, n, group,
                                             out + (size_t)b * n);
    });
    int worst = ONEDC_OK;
    for (int b = 0; b < batch; ++b) {
        if (status) status[b] = rc[b];
        if (rc[b] != ONEDC_OK && worst == ONEDC_OK) worst = rc[b];
    }
    return worst;
}
int onedc_rans_enc_encode_batch(onedc_rans_enc *const *encs, int batch, const int16_t *symbols,
                                const int16_t *indexes, int64_t n, int group, int n_threads) {
    if (!encs || batch < 0 || n < 0) return ONEDC_ERR_ARG;
    std::vector<int> rc(batch, ONEDC_OK);
    parallel_for(batch, n_threads, [&](int b) {
        rc[b] = onedc_rans_enc_encode_with_indexes(encs[b], symbols + (size_t)b * n,
                                                   indexes + (size_t)b * n, n, group);
    });
    for (int b = 0; b < batch; ++b)
        if (rc[b] != ONEDC_OK) return rc[b];
    return ONEDC_OK;
}

int onedc_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    if (!pmf || !cdf || n < 1 || precision < 1 || precision > 31) return ONEDC_ERR_ARG;
    const int m = n + 1;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i)
        cdf[i + 1] = (uint32_t)(std::round(pmf[i] * (float)(1 << precision)) + 0.5f);
    uint32_t total = 0;
    for (int i = 0; i < m; ++i) total += cdf[i];
    if (total == 0) return ONEDC_ERR_ARG;
    for (int i = 0; i < m; ++i) cdf[i] = (uint32_t)(((1ull << precision) * cdf[i]) / total);
    for (int i = 1; i < m; ++i) cdf[i] += cdf[i - 1];
    cdf[m - 1] = 1u << precision;
    for (int i = 0; i < m - 1; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        // zero-width symbol: take one count from the narrowest symbol wider than 1
        uint32_t best_freq = ~0u;
        int donor = -1;
        for (int j = 0; j < m - 1; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best_freq) {
                best_freq = f;
                donor = j;
            }
        }
        if (donor < 0) return ONEDC_ERR_ARG;
        if (donor < i)
            for (int j = donor + 1; j <= i; ++j) --cdf[j];
        else
            for (int j = i + 1; j <= donor; ++j) ++cdf[j];
    }
    return ONEDC_OK;
}

static inline uint32_t be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
static inline void put_be32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

int onedc_container_parse(const uint8_t *data, int64_t n, int index_unit_length, int ds,
                          onedc_container *out) {
    if (!data || !out || index_unit_length < 1 || ds < 1) return ONEDC_ERR_ARG;
    if (n < 16) return ONEDC_ERR_STREAM;
    out->height = be32(data);
    out->width = be32(data + 4);
    out->len_y = be32(data + 8);
    out->len_caption = be32(data + 12);
    out->pad_height = (uint32_t)(((uint64_t)out->height + ds - 1) / ds * ds);
    out->pad_width = (uint32_t)(((uint64_t)out->width + ds - 1) / ds * ds);
    const uint64_t cells = (uint64_t)(out->pad_height / ds) * (out->pad_width / ds);
    out->len_z = (uint32_t)((cells * (uint64_t)index_unit_length + 7) / 8);
    out->off_y = 16;
    out->off_z = out->off_y + out->len_y;
    out->off_caption = out->off_z + out->len_z;
    if (out->off_caption + (int64_t)out->len_caption > n) return ONEDC_ERR_STREAM;
    return ONEDC_OK;
}

int64_t onedc_container_write(uint32_t height, uint32_t width, const uint8_t *y, uint32_t len_y,
                              const uint8_t *z, uint32_t len_z, const uint8_t *caption,
                              uint32_t len_caption, uint8_t *out, int64_t capacity) {
    const int64_t need = 16 + (int64_t)len_y + len_z + len_caption;
    if (!out) return need;
    if (capacity < need) return ONEDC_ERR_CAPACITY;
    if ((len_y && !y) || (len_z && !z) || (len_caption && !caption)) return ONEDC_ERR_ARG;
    put_be32(out, height);
    put_be32(out + 4, width);
    put_be32(out + 8, len_y);
    put_be32(out + 12, len_caption);
    uint8_t *p = out + 16;
    if (len_y) memcpy(p, y, len_y);
    p += len_y;
    if (len_z) memcpy(p, z, len_z);
    p += len_z;
    if (len_caption) memcpy(p, caption, len_caption);
    return need;
}

int64_t onedc_pack_indices(const int32_t *indices, int64_t count, int unit_bits, uint8_t *out,
                           int64_t capacity) {
    if (count < 0 || unit_bits < 1 || unit_bits > 31) return ONEDC_ERR_ARG;
    const int64_t total_bits = count * unit_bits;
    const int64_t nbytes = (total_bits + 7) / 8;
    if (!out) return nbytes;
    if (!indices && count) return ONEDC_ERR_ARG;
    if (capacity < nbytes) return ONEDC_ERR_CAPACITY;
    memset(out, 0, (size_t)nbytes);
    // right-aligned big-endian: bit 0 of the string is the MSB of index 0, the
    // string's last bit is the LSB of the last byte
    const int64_t lead = nbytes * 8 - total_bits;
    for (int64_t i = 0; i < count; ++i) {
        const uint32_t v = (uint32_t)indices[i];
        if (v >> unit_bits) return ONEDC_ERR_ARG;
        for (int b = 0; b < unit_bits; ++b) {
            if (!((v >> (unit_bits - 1 - b)) & 1)) continue;
            const int64_t pos = lead + i * unit_bits + b;
            out[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7));
        }
    }
    return nbytes;
}

int onedc_unpack_indices(const uint8_t *bytes, int64_t nbytes, int64_t count, int unit_bits,
                         int32_t *out) {
    if (!bytes || !out || count < 0 || unit_bits < 1 || unit_bits > 31) return ONEDC_ERR_ARG;
    const int64_t total_bits = count * unit_bits;
    if (nbytes * 8 < total_bits) return ONEDC_ERR_STREAM;
    const int64_t lead = nbytes * 8 - total_bits;
    for (int64_t i = 0; i < count; ++i) {
        uint32_t v = 0;
        for (int b = 0; b < unit_bits; ++b) {
            const int64_t pos = lead + i * unit_bits + b;
            v = (v << 1) | ((bytes[pos >> 3] >> (7 - (pos & 7))) & 1u);
        }
        out[i] = (int32_t)v;
    }
    return ONEDC_OK;
}

}  // extern "C"
