/*
 * hvq_container.c -- .h4m (HVQM4 1.3/1.5) container demux, in memory.
 *
 * Restates the file-level part of the reference player: header parse + checks of load_header
 * (h4m_audio_decode.c:2175-2247) and the GOP-block / frame-record walk of main (h4m:2427-2537), without file I/O
 * and without exit(): every inconsistency the reference aborts on is an error code here.  Wire format:
 * SURVEY.md Appendix B.  Audio records are skipped like the reference does (h4m:2486, 2506).
 */
#include <string.h>

#include "../../include/hvqm4_amd.h"

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

int hvq_h4m_header(const uint8_t *data, size_t n, HvqH4mInfo *out)
{
    static const char m13[16] = "HVQM4 1.3", m15[16] = "HVQM4 1.5";
    if (!data || !out || n < 0x44) return HVQ_E_ARG;
    memset(out, 0, sizeof *out);
    if (!memcmp(data, m15, 16)) out->is_1_5 = 1;
    else if (memcmp(data, m13, 16)) return HVQ_E_CONTAINER;            /* "does not appear to be a HVQM4 file" */
    out->header_size = be32(data + 0x10);
    out->body_size = be32(data + 0x14);
    out->blocks = be32(data + 0x18);
    out->video_frames = be32(data + 0x1C);
    out->audio_frames = be32(data + 0x20);
    out->usec_per_frame = be32(data + 0x24);
    out->max_frame_size = be32(data + 0x28);
    out->width = (uint16_t)be16(data + 0x34);
    out->height = (uint16_t)be16(data + 0x36);
    out->h_samp = data[0x38];
    out->v_samp = data[0x39];
    out->video_mode = data[0x3A];
    if (out->header_size != 0x44) return HVQ_E_CONTAINER;              /* h4m:2213 */
    if (out->blocks == 0) return HVQ_E_CONTAINER;                      /* h4m:2215-2219 */
    if (be32(data + 0x2C) != 0 || data[0x3B] != 0) return HVQ_E_CONTAINER;   /* h4m:2237, 2244 */
    if (out->video_mode != 0 && out->video_mode != 0x12) return HVQ_E_CONTAINER;   /* h4m:2240-2243 */
    uint32_t ss = (uint32_t)out->h_samp * out->v_samp;
    out->pic_bytes = ss ? (uint32_t)out->width * out->height * (ss + 2) / ss : 0;   /* h4m:2343-2345 */
    return HVQ_OK;
}

void hvq_h4m_begin(HvqH4mIter *it)
{
    memset(it, 0, sizeof *it);
    it->pos = 0x44;
}

/* 1 = a video picture was produced, 0 = end of file (all counts consistent), < 0 = malformed */
int hvq_h4m_next(const uint8_t *data, size_t n, HvqH4mIter *it, int *frame_type, uint32_t *disp_id,
                 const uint8_t **pic, size_t *len)
{
    HvqH4mInfo info;
    int rc = hvq_h4m_header(data, n, &info);
    if (rc) return rc;
    for (;;) {
        if (it->v_left == 0 && it->a_left == 0) {
            if (it->in_block) {                                        /* block size check, h4m:2531-2535 */
                if (it->pos != it->block_end) return HVQ_E_CONTAINER;
                it->in_block = 0;
            }
            if (it->block == info.blocks)
                return it->video_seen == info.video_frames ? 0 : HVQ_E_CONTAINER;   /* h4m:2544-2549 */
            if (it->pos + 20 > n) return HVQ_E_CONTAINER;
            uint32_t bsize = be32(data + it->pos + 4);
            it->v_left = be32(data + it->pos + 8);
            it->a_left = be32(data + it->pos + 12);
            if (be32(data + it->pos + 16) != 0x01000000u) return HVQ_E_CONTAINER;  /* h4m:2436 */
            it->pos += 20;
            it->block_end = it->pos + bsize;
            it->gop_start = it->video_seen;
            it->in_block = 1;
            it->block++;
            continue;
        }
        if (it->pos + 8 > n) return HVQ_E_CONTAINER;
        uint32_t id1 = be16(data + it->pos), id2 = be16(data + it->pos + 2), size = be32(data + it->pos + 4);
        it->pos += 8;
        if (it->pos + size > n) return HVQ_E_CONTAINER;
        if (id1 == 1) {
            if (id2 != 0x10 && id2 != 0x20 && id2 != 0x30) return HVQ_E_CONTAINER;  /* h4m:2113-2116 */
            if (size < 4 || it->v_left == 0) return HVQ_E_CONTAINER;
            *frame_type = (int)id2;
            *disp_id = it->gop_start + be32(data + it->pos);           /* h4m:2085, 2122 */
            *pic = data + it->pos + 4;
            *len = size - 4;
            it->pos += size;
            it->v_left--;
            it->video_seen++;
            return 1;
        }
        if (id1 != 0 || it->a_left == 0) return HVQ_E_CONTAINER;       /* h4m:2509-2513 */
        it->pos += size;                                               /* audio: skipped */
        it->a_left--;
    }
}
