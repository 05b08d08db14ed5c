/* oracle/rans_port.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C scalar restatement of the reference's host entropy coder, kept as
 * close to the reference control flow as C allows (linear CDF scan, one symbol
 * at a time, single stream part).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product coder is
 * onedc_amd/csrc/coder.cpp.
 *
 * Parity pin: checked byte-for-byte against the reference's own C++ sources
 * compiled into oracle/_ref (see oracle/Makefile) and against the golden
 * streams under tests/golden/ that were produced by that build.
 *
 * Follows (paths relative to the reference root):
 *   encode  src/cpp/rans/rans.cpp:101-159 (symbol -> pending list), :161-187 (flush)
 *   decode  src/cpp/rans/rans.cpp:303-362, RansDecGetBits :53-68
 *   state   src/cpp/rans/rans_byte.h:62-155 (init / put / flush / get / advance)
 *   framing src/cpp/py_rans/py_rans.cpp:91-142 (flag byte), :156-187
 *   cdf     src/cpp/ops/ops.cpp:24-82 (pmf_to_quantized_cdf)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION 16
#define BYPASS_PRECISION 2
#define MAX_BYPASS_VAL 3
#define RANS_L (1u << 23)

typedef struct {
    uint16_t start, range;
} sym_t;

typedef struct {
    const int32_t *cdf; /* [n_tables][row_len] */
    const int32_t *sizes;
    const int32_t *offsets;
    int row_len;
} tables_t;

/* rans_byte.h:90-97 RansEncPut with RansEncRenorm :70-81 */
static void enc_put(uint32_t *r, uint8_t **pp, uint32_t start, uint32_t freq) {
    uint32_t x = *r;
    uint32_t x_max = freq << 15;
    while (x >= x_max) {
        *(--(*pp)) = (uint8_t)(x & 0xff);
        x >>= 8;
    }
    *r = ((x / freq) << PRECISION) + (x % freq) + start;
}

/* rans.cpp:34-51 RansEncPutBits */
static void enc_put_bits(uint32_t *r, uint8_t **pp, uint32_t val, uint32_t nbits) {
    uint32_t x = *r;
    uint32_t freq = 1u << (PRECISION - nbits);
    uint32_t x_max = freq << 15;
    while (x >= x_max) {
        *(--(*pp)) = (uint8_t)(x & 0xff);
        x >>= 8;
    }
    *r = (x << nbits) | val;
}

/* Encode n symbols in `steps` consecutive calls' worth of data (already
 * concatenated by the caller) and return the framed stream: flag byte 0x01 +
 * rANS bytes.  Returns the stream length or -1 when `cap` is too small. */
long oracle_rans_encode(const int16_t *symbols, const int16_t *indexes, long n, const int32_t *cdf,
                        int row_len, const int32_t *sizes, const int32_t *offsets, uint8_t *out,
                        long cap) {
    sym_t *syms = (sym_t *)malloc(sizeof(sym_t) * (size_t)(n * 24 + 8));
    long ns = 0;
    for (long i = 0; i < n; ++i) { /* rans.cpp:116-158 */
        int32_t cdf_idx = indexes[i];
        if (cdf_idx < 0) continue;
        const int32_t *c = cdf + (long)cdf_idx * row_len;
        int32_t max_value = sizes[cdf_idx] - 2;
        int32_t value = symbols[i] - offsets[cdf_idx];
        uint32_t raw_val = 0;
        if (value < 0) {
            raw_val = (uint32_t)(-2 * value - 1);
            value = max_value;
        } else if (value >= max_value) {
            raw_val = (uint32_t)(2 * (value - max_value));
            value = max_value;
        }
        syms[ns].start = (uint16_t)c[value];
        syms[ns].range = (uint16_t)(c[value + 1] - c[value]);
        ++ns;
        if (value == max_value) {
            int32_t n_bypass = 0;
            while ((raw_val >> (n_bypass * BYPASS_PRECISION)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= MAX_BYPASS_VAL) {
                syms[ns].start = MAX_BYPASS_VAL;
                syms[ns].range = 0;
                ++ns;
                val -= MAX_BYPASS_VAL;
            }
            syms[ns].start = (uint16_t)val;
            syms[ns].range = 0;
            ++ns;
            for (int32_t j = 0; j < n_bypass; ++j) {
                syms[ns].start = (uint16_t)((raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL);
                syms[ns].range = 0;
                ++ns;
            }
        }
    }
    /* rans.cpp:161-187 flush: code the pending list last-to-first, bytes grow downwards */
    long bufsz = ns * 2 + 8;
    uint8_t *buf = (uint8_t *)malloc((size_t)bufsz);
    uint8_t *end = buf + bufsz, *p = end;
    uint32_t r = RANS_L;
    for (long k = ns - 1; k >= 0; --k) {
        if (syms[k].range != 0)
            enc_put(&r, &p, syms[k].start, syms[k].range);
        else
            enc_put_bits(&r, &p, syms[k].start, BYPASS_PRECISION);
    }
    p -= 4; /* rans_byte.h:100-112 RansEncFlush */
    p[0] = (uint8_t)(r >> 0);
    p[1] = (uint8_t)(r >> 8);
    p[2] = (uint8_t)(r >> 16);
    p[3] = (uint8_t)(r >> 24);
    long nbytes = (long)(end - p);
    long total = nbytes + 1;
    if (total <= cap) {
        out[0] = 0x01; /* py_rans.cpp:116-118: one stream, 2-byte size fields */
        memcpy(out + 1, p, (size_t)nbytes);
    } else {
        total = -1;
    }
    free(buf);
    free(syms);
    return total;
}

typedef struct {
    const uint8_t *ptr;
    uint32_t state;
} oracle_dec_t;

/* py_rans.cpp:156-187 + rans.cpp:280-285 + rans_byte.h:116-128 */
void oracle_rans_dec_init(oracle_dec_t *d, const uint8_t *stream) {
    const uint8_t *p = stream + 1; /* skip the flag byte */
    d->state = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    d->ptr = p + 4;
}

static uint32_t dec_get_bits(oracle_dec_t *d, uint32_t n_bits) { /* rans.cpp:53-68 */
    uint32_t x = d->state;
    uint32_t val = x & ((1u << n_bits) - 1);
    x >>= n_bits;
    if (x < RANS_L) {
        x = (x << 8) | *d->ptr;
        d->ptr += 1;
    }
    d->state = x;
    return val;
}

/* rans.cpp:303-362 */
void oracle_rans_decode(oracle_dec_t *d, const int16_t *indexes, long n, const int32_t *cdf,
                        int row_len, const int32_t *sizes, const int32_t *offsets, int16_t *out) {
    for (long i = 0; i < n; ++i) {
        int32_t cdf_idx = indexes[i];
        if (cdf_idx < 0) {
            out[i] = 0;
            continue;
        }
        const int32_t *c = cdf + (long)cdf_idx * row_len;
        int32_t max_value = sizes[cdf_idx] - 2;
        uint32_t cum_freq = d->state & ((1u << PRECISION) - 1);
        int s = 0; /* linear scan: first entry greater than cum_freq, minus one */
        while (s < sizes[cdf_idx] && !((uint32_t)c[s] > cum_freq)) ++s;
        s -= 1;
        uint32_t start = (uint32_t)c[s], freq = (uint32_t)(c[s + 1] - c[s]);
        uint32_t x = d->state; /* rans_byte.h:137-155 RansDecAdvance */
        x = freq * (x >> PRECISION) + (x & ((1u << PRECISION) - 1)) - start;
        while (x < RANS_L) x = (x << 8) | *d->ptr++;
        d->state = x;
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(d, BYPASS_PRECISION);
            int32_t n_bypass = val;
            while (val == MAX_BYPASS_VAL) {
                val = (int32_t)dec_get_bits(d, BYPASS_PRECISION);
                n_bypass += val;
            }
            int32_t raw_val = 0;
            for (int j = 0; j < n_bypass; ++j) {
                val = (int32_t)dec_get_bits(d, BYPASS_PRECISION);
                raw_val |= val << (j * BYPASS_PRECISION);
            }
            value = raw_val >> 1;
            if (raw_val & 1)
                value = -value - 1;
            else
                value += max_value;
        }
        out[i] = (int16_t)(value + offsets[cdf_idx]);
    }
}

/* ops.cpp:24-82 */
int oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    int m = n + 1;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)(roundf(pmf[i] * (1 << precision)) + 0.5);
    int total = 0; /* the reference accumulates into an int (std::accumulate(..., 0)) */
    for (int i = 0; i < m; ++i) total += (int)cdf[i];
    for (int i = 0; i < m; ++i)
        cdf[i] = (uint32_t)((((unsigned long long)1 << precision) * cdf[i]) / (uint32_t)total);
    for (int i = 1; i < m; ++i) cdf[i] += cdf[i - 1];
    cdf[m - 1] = 1u << precision;
    for (int i = 0; i < m - 1; ++i) {
        if (cdf[i] == cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (int j = 0; j < m - 1; ++j) {
                uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) {
                    best_freq = freq;
                    best_steal = j;
                }
            }
            if (best_steal == -1) return -1;
            if (best_steal < i) {
                for (int j = best_steal + 1; j <= i; ++j) cdf[j]--;
            } else {
                for (int j = i + 1; j <= best_steal; ++j) cdf[j]++;
            }
        }
    }
    return 0;
}
