/*
 * hvq_parse.h -- host entropy parse: HVQM4 picture bitstream -> descriptor blob (hvq_desc.h).
 *
 * This is the serial half of the reference's HVQM4DecodeIpic / HVQM4DecodeBpic
 * (h4m_audio_decode.c:1970-2056): bit reader, prefix trees, overflow symbols, run lengths,
 * DC prediction, nest construction, P/B descriptor map and motion-vector chains
 * (SURVEY.md rows a22-a29).  It never touches pixels, so it can run
 * arbitrarily far ahead of the GPU and one parser per stream can run on its own thread.
 */
#ifndef HVQ_PARSE_H
#define HVQ_PARSE_H

#include <stddef.h>
#include <stdint.h>

#include "hvq_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HVQ_OK            0
#define HVQ_E_ARG        -1
#define HVQ_E_OVERFLOW   -2   /* blob capacity exceeded (absurd basis counts) */
#define HVQ_E_GEOMETRY   -3   /* unsupported size / sampling */
#define HVQ_E_NOGPU      -4
#define HVQ_E_HIP        -5
#define HVQ_E_STATE      -6
#define HVQ_E_CONTAINER  -7

typedef struct HvqParser HvqParser;

/* width/height multiples of 8 (h4m:1749-1752), <= 8192; h_samp/v_samp: 2,2 (4:2:0) or 1,1 */
HvqParser *hvq_parser_create(int width, int height, int h_samp, int v_samp, int is15);
void hvq_parser_destroy(HvqParser *p);

/* capacity that holds any picture whose blocks carry <= 15 bases each */
size_t hvq_parser_blob_bound(const HvqParser *p);
uint32_t hvq_parser_pic_bytes(const HvqParser *p);

/* geometry part of the blob header (everything but the per-picture fields): the layout any blob of this parser has */
void hvq_parser_layout(const HvqParser *p, HvqPicHeader *out);

/*
 * Parse one picture.  `frame_type` is the container id 0x10 I / 0x20 P / 0x30 B
 * (h4m:2065-2070); `pic` points at the picture data (after the 4-byte disp_id); `len` is the
 * number of readable bytes at `pic` (0 = unknown, trust the stream like the reference does).
 * Writes the blob to `blob` (16-byte aligned) and its size to *blob_len.
 */
int hvq_parse_picture(HvqParser *p, int frame_type, const uint8_t *pic, size_t len,
                      uint8_t *blob, size_t cap, size_t *blob_len);

/* Threads that parse ONE picture of this parser (the caller's included; 1 = the caller alone, the default; at most 8): a picture's
 * sections are independent bit buffers (h4m:1981-1993, 2030-2044), so block kinds, DC values, vectors and the planes' payloads are
 * decoded side by side by a small persistent pool.  Same blob byte for byte.  Returns the count in effect. */
int hvq_parser_set_threads(HvqParser *p, int threads);

/* every HVQ_F_* flag the last hvq_parse_picture raised -- the blob header's, plus what a P/B picture's second pass raised behind it */
uint32_t hvq_parser_last_flags(const HvqParser *p);

/* the nest of the last I picture this parser has parsed (h4m:1132-1164), nibble-packed as the blobs carry it: `out` = ALIGN16(HVQ_NESTP_BYTES)
 * bytes.  (An I picture's blob contains its nest only when one of its own blocks needs it; later P/B pictures may need it regardless.) */
void hvq_parser_packed_nest(const HvqParser *p, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
