/* onedc_coder.h — C ABI of the host-side entropy coder boundary.
 *
 * This is the drop-in boundary for the reference's native code on the decode
 * hot path.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repository root).  The reference exposes
 * these through pybind11 modules `MLCodec_rans` / `MLCodec_CXX`
 * (src/cpp/py_rans/py_rans.cpp:261-281, src/cpp/ops/ops.cpp:84-91) that are
 * called from src/modules/entropy/entropy_models.py:32-94.
 *
 * Conventions: plain pointers + sizes, caller-allocated outputs, `int` status
 * codes (0 = ok, <0 = error), no exceptions cross the ABI.  Coder instances
 * are not thread-safe (same as the reference: rans.h:86-94,144-151) but
 * distinct instances are independent, so a thread pool may drive one
 * instance per bitstream.
 *
 * Bytes produced by the encoder are identical to the reference for identical
 * (symbols, indexes, tables); the decoder returns identical symbols.
 */
#ifndef ONEDC_CODER_H
#define ONEDC_CODER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ONEDC_OK 0
#define ONEDC_ERR_ARG (-1)          /* null pointer / bad size / bad group index */
#define ONEDC_ERR_STATE (-2)        /* call order violated (e.g. decode before set_stream) */
#define ONEDC_ERR_STREAM (-3)       /* bitstream too short / malformed framing */
#define ONEDC_ERR_CAPACITY (-4)     /* caller buffer too small */

typedef struct onedc_rans_enc onedc_rans_enc;
typedef struct onedc_rans_dec onedc_rans_dec;

/* ---- encoder: replaces class RansEncoder (py_rans.h:13-36) over
 *      RansEncoderLib / RansEncoderLibMultiThread (rans.h:64-127) ---------- */

/* RansEncoder(bool multiThread, int streamPart)  py_rans.cpp:11-20.
 * `multi_thread` is accepted for signature parity; this implementation encodes
 * synchronously (results are identical, the reference thread only defers work). */
onedc_rans_enc *onedc_rans_enc_create(int multi_thread, int stream_part);
void onedc_rans_enc_destroy(onedc_rans_enc *enc);

/* add_cdf(int32[N,L] cdfs, int32[N] sizes, int32[N] offsets) -> group index
 * py_rans.cpp:47-76, rans.cpp:71-93.  Returns group index >= 0 or error < 0. */
int onedc_rans_enc_add_cdf(onedc_rans_enc *enc, const int32_t *cdfs, int n_tables,
                           int row_len, const int32_t *sizes, const int32_t *offsets);
/* empty_cdf_buffer()  py_rans.cpp:78-83, rans.cpp:95-99 */
int onedc_rans_enc_empty_cdf_buffer(onedc_rans_enc *enc);
/* reset()  py_rans.cpp:144-148, rans.cpp:191 */
int onedc_rans_enc_reset(onedc_rans_enc *enc);
/* encode_with_indexes(int16[] symbols, int16[] indexes, int group)
 * py_rans.cpp:22-45, rans.cpp:101-159.  indexes[i] < 0 => symbol skipped. */
int onedc_rans_enc_encode_with_indexes(onedc_rans_enc *enc, const int16_t *symbols,
                                       const int16_t *indexes, int64_t n, int group);
/* flush()  py_rans.cpp:85-89, rans.cpp:161-187 */
int onedc_rans_enc_flush(onedc_rans_enc *enc);
/* get_encoded_stream() -> uint8[]  py_rans.cpp:91-142 (1 flag byte, optional
 * per-part sizes, then the part streams).  Two calls: size query, then fill. */
int64_t onedc_rans_enc_stream_size(onedc_rans_enc *enc);
int onedc_rans_enc_get_stream(onedc_rans_enc *enc, uint8_t *out, int64_t capacity);

/* ---- decoder: replaces class RansDecoder (py_rans.h:38-57) over
 *      RansDecoderLib (rans.h:129-160) --------------------------------------- */

/* RansDecoder(int streamPart)  py_rans.cpp:150-154 */
onedc_rans_dec *onedc_rans_dec_create(int stream_part);
void onedc_rans_dec_destroy(onedc_rans_dec *dec);
/* add_cdf / empty_cdf_buffer  py_rans.cpp:223-259, rans.cpp:287-301 */
int onedc_rans_dec_add_cdf(onedc_rans_dec *dec, const int32_t *cdfs, int n_tables,
                           int row_len, const int32_t *sizes, const int32_t *offsets);
int onedc_rans_dec_empty_cdf_buffer(onedc_rans_dec *dec);
/* set_stream(uint8[])  py_rans.cpp:156-187, rans.cpp:280-285.  The bytes are
 * copied (as the reference does), the caller may free them afterwards. */
int onedc_rans_dec_set_stream(onedc_rans_dec *dec, const uint8_t *stream, int64_t n);
/* decode_stream(int16[] indexes, int group) -> int16[]
 * py_rans.cpp:189-221, rans.cpp:303-362.  `out` has room for n symbols.
 * Returns ONEDC_ERR_STREAM when the decoder ran past the end of the stream
 * (missing bytes read as 0; the reference reads out of bounds there). */
int onedc_rans_dec_decode_stream(onedc_rans_dec *dec, const int16_t *indexes, int64_t n,
                                 int group, int16_t *out);

/* Extension (no reference counterpart): has the decoder consumed its stream EXACTLY -- every byte read, none
 * missing, state back at the encoder's initial state (1 << 23, rans.cpp:161-187 writes the final state of a coder
 * that started there)?  ONEDC_OK, or ONEDC_ERR_STREAM when the symbols pulled so far do not account for the stream:
 * a scale index or symbol decision differed from the encoder's somewhere (or the stream is corrupt).  The reference
 * has no such check: a desynchronised decode returns a wrong picture silently.  Call after the last decode_stream. */
int onedc_rans_dec_finished(onedc_rans_dec *dec);

/* Extension (no reference counterpart; the reference is batch=1): decode the
 * next `n` symbols of `batch` independent decoders concurrently on
 * `n_threads` host threads.  indexes/out are [batch][n] contiguous.
 * status[i] receives the per-stream status. */
int onedc_rans_dec_decode_batch(onedc_rans_dec *const *decs, int batch, const int16_t *indexes,
                                int64_t n, int group, int16_t *out, int n_threads, int *status);
/* Extension: the encode-side twin (push symbols for `batch` encoders). */
int onedc_rans_enc_encode_batch(onedc_rans_enc *const *encs, int batch, const int16_t *symbols,
                                const int16_t *indexes, int64_t n, int group, int n_threads);

/* ---- MLCodec_CXX.pmf_to_quantized_cdf(list[float], int) -> list[uint32]
 *      src/cpp/ops/ops.cpp:24-82.  cdf_out has n + 1 entries. ---------------- */
int onedc_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf_out);

/* ---- container format: src/modules/entropy/utils.py:95-132 (encode_i /
 *      decode_i) and get_padding_size :7-16 ---------------------------------- */
typedef struct onedc_container {
    uint32_t height, width;          /* original picture size */
    uint32_t pad_height, pad_width;  /* rounded up to a multiple of ds, pad right/bottom */
    uint32_t len_y, len_z, len_caption;
    int64_t off_y, off_z, off_caption; /* byte offsets into the container */
} onedc_container;

/* decode_i(data, index_unit_length, ds)  utils.py:108-132 */
int onedc_container_parse(const uint8_t *data, int64_t n, int index_unit_length, int ds,
                          onedc_container *out);
/* encode_i(...)  utils.py:95-105.  Size query with out == NULL. Returns bytes written. */
int64_t onedc_container_write(uint32_t height, uint32_t width, const uint8_t *y, uint32_t len_y,
                              const uint8_t *z, uint32_t len_z, const uint8_t *caption,
                              uint32_t len_caption, uint8_t *out, int64_t capacity);

/* FSQ index (de)serialisation: right-aligned big-endian bit string,
 * src/models/sd15_onedc_codec_stage1/codec_module.py:404-409 (pack), :425-429 (unpack). */
int64_t onedc_pack_indices(const int32_t *indices, int64_t count, int unit_bits, uint8_t *out,
                           int64_t capacity);
int onedc_unpack_indices(const uint8_t *bytes, int64_t nbytes, int64_t count, int unit_bits,
                         int32_t *out);

const char *onedc_coder_version(void);
/* identity (sha256 prefix) of coder.cpp + this header as compiled into the library (onedc_amd/build.py: coder_source_id) */
const char *onedc_coder_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* ONEDC_CODER_H */
