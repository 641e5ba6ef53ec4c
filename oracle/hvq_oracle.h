/*
 * oracle/hvq_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single-threaded) of the HVQM4 1.3/1.5 picture
 * reconstruction algorithm of the reference decoder (Tilka/hvqm4,
 * h4m_audio_decode.c).  It exists to CHECK the HIP path; nothing under
 * hvqm4_amd/ may include, link or call it.
 *
 * Parity pin: the reference ships no golden vectors, so this
 * oracle is pinned against outputs of the reference itself run in the build
 * container (oracle/_ref, tests/test_oracle.py) and against the committed
 * fixtures under tests/golden/ that were generated from it (tests/golden/make_golden.py).
 */
#ifndef HVQ_ORACLE_H
#define HVQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HvqOracle HvqOracle;

HvqOracle *hvqo_create(int width, int height, int h_samp, int v_samp, int is15);
void hvqo_destroy(HvqOracle *o);
uint32_t hvqo_picsize(const HvqOracle *o);

/* `pic` points at the picture data (4 bytes past the record start, after disp_id); planes are
 * tightly packed Y|U|V exactly like the reference's `present` buffers (h4m:2340-2350). */
void hvqo_decode_ipic(HvqOracle *o, const uint8_t *pic, uint8_t *present);
void hvqo_decode_bpic(HvqOracle *o, const uint8_t *pic, uint8_t *present,
                      const uint8_t *past, const uint8_t *future);
void hvqo_decode_ppic(HvqOracle *o, const uint8_t *pic, uint8_t *present, const uint8_t *past);

/* whole .h4m in memory -> pictures in decode order (picture rotation of h4m:2087-2137) */
int hvqo_decode_clip(const uint8_t *file, size_t n, uint8_t *out, size_t out_cap, int max_pics);
/* decode-only timing (CPU baseline "port"); returns seconds, *pixels = luma pixels decoded */
double hvqo_time_clip(const uint8_t *file, size_t n, int reps, uint64_t *pixels);

/* display epilogue of the reference player: YUV 4:2:0 -> RGB24, float, exactly dumpRGB (h4m:897-926) */
void hvqo_yuv420_to_rgb(const uint8_t *yuv, int w, int h, uint8_t *rgb);

/* white-box pieces for known-answer tests */
void hvqo_weight_block(uint8_t dst16[16], uint8_t v, uint8_t t, uint8_t b, uint8_t l, uint8_t r);
void hvqo_motion_comp(uint8_t dst16[16], const uint8_t *src, uint32_t stride, int hx, int hy);
void hvqo_tables(int32_t div16[16], int32_t mcdiv512[512]);
const uint8_t *hvqo_nest(const HvqOracle *o);           /* 70*38 bytes */
const uint8_t *hvqo_map(const HvqOracle *o, int plane, uint32_t *stride, uint32_t *rows); /* {value,type} pairs incl. border */

#ifdef __cplusplus
}
#endif
#endif
