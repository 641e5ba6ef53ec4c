/*
 * oracle/ref_wrap.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin wrapper translation unit around the *unmodified* reference decoder.
 * The reference source is NOT copied into this repository: it is #include-d
 * from where it lies (-I/root/reference) and compiled into
 * oracle/_ref/libh4mref.so by oracle/Makefile.  Every function of the
 * reference is `static`, so the only way to reach them is from inside the
 * same TU.
 *
 * What this file adds (own code): an in-memory .h4m demuxer that replays the
 * reference player's picture-buffer rotation (h4m_audio_decode.c:2087-2137)
 * and hands every decoded picture back as raw Y|U|V, a section-consumption
 * probe used to validate the synthetic stream writer, white-box entry points
 * for known-answer tests, and a decode-only timer for the CPU baseline.
 */
#define NATIVE 1
#define HVQM4_FFMPEG 1
#include "h4m_audio_decode.c"

#include <time.h>
#ifndef REF_SLACK
#define REF_SLACK 64
#endif

#define REF_API __attribute__((visibility("default")))

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
static uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

typedef struct {
    Player   player;
    uint32_t picsize;
    int      is15;
} RefCtx;

static int ref_open(RefCtx *c, const uint8_t *file, size_t n)
{
    if (n < 0x44) return -1;
    int is13 = !memcmp(file, HVQM4_13_magic, 16);
    int is15 = !memcmp(file, HVQM4_15_magic, 16);
    if (!is13 && !is15) return -2;
    c->is15 = is15;
    HVQM4InitDecoder();
    VideoInfo vi;
    vi.hres = be16(file + 0x34);
    vi.vres = be16(file + 0x36);
    vi.h_samp = file[0x38];
    vi.v_samp = file[0x39];
    vi.video_mode = file[0x3A];
    HVQM4InitSeqObj(&c->player.seqobj, &vi);
    uint32_t sz = HVQM4BuffSize(&c->player.seqobj);
    VideoState *st = calloc(1, sz);
    st->padding[0] = (uint8_t)is15;              /* h4m:2414-2417 */
    HVQM4SetBuffer(&c->player.seqobj, st);
    uint32_t res = (uint32_t)vi.hres * vi.vres;
    uint32_t ss = (uint32_t)vi.h_samp * vi.v_samp;
    c->picsize = res * (ss + 2) / ss;            /* h4m:2343-2345 */
    /* calloc + slack so that deterministic bytes are read by any in-buffer run-off */
    c->player.past = calloc(1, c->picsize + REF_SLACK);
    c->player.present = calloc(1, c->picsize + REF_SLACK);
    c->player.future = calloc(1, c->picsize + REF_SLACK);
    return 0;
}

static void ref_close(RefCtx *c)
{
    free(c->player.seqobj.state);
    free(c->player.past);
    free(c->player.present);
    free(c->player.future);
}

/* one video record: rotation exactly as decode_video (h4m:2087-2137) minus file I/O and dumps */
static void ref_one(RefCtx *c, uint16_t ftype, const uint8_t *payload, uint32_t size, uint8_t *scratch)
{
    Player *pl = &c->player;
    memcpy(scratch, payload, size);
    memset(scratch + size, 0, 8);                /* reader may touch 3 bytes past the end */
    if (ftype != B_FRAME) { void *t = pl->past; pl->past = pl->future; pl->future = t; }
    switch (ftype) {
    case I_FRAME: HVQM4DecodeIpic(&pl->seqobj, scratch + 4, pl->present); break;
    case P_FRAME: HVQM4DecodePpic(&pl->seqobj, scratch + 4, pl->present, pl->past); break;
    default:      HVQM4DecodeBpic(&pl->seqobj, scratch + 4, pl->present, pl->past, pl->future); break;
    }
}

static void ref_rotate_after(RefCtx *c, uint16_t ftype)
{
    Player *pl = &c->player;
    if (ftype != B_FRAME) { void *t = pl->present; pl->present = pl->future; pl->future = t; }
}

/*
 * Decode a whole .h4m held in memory.  Pictures are written to `out` in
 * decode order, each `picsize` bytes (Y|U|V).  `probe`, when non-NULL,
 * receives 20 int32 per picture: for each of the 19 bit buffers (order:
 * basis_num[2], basis_num_run[2], dc_values[3], bufTree0[3], fixvl[3],
 * dc_rle[3], mv_h, mv_v, mcb_type) the byte offset of its read cursor
 * relative to the picture data start after the decode call (-1 when the
 * section is empty), then mcb_proc.  Returns pictures decoded or <0.
 */
REF_API int ref_decode_clip(const uint8_t *file, size_t n, uint8_t *out, size_t out_cap,
                            int32_t *probe, int max_pics)
{
    RefCtx c;
    int rc = ref_open(&c, file, n);
    if (rc) return rc;
    uint32_t blocks = be32(file + 0x18);
    uint32_t maxf = be32(file + 0x28);
    uint8_t *scratch = malloc((size_t)maxf + n + 64);
    size_t pos = 0x44;
    int pics = 0;
    for (uint32_t b = 0; b < blocks && pos + 20 <= n; ++b) {
        uint32_t vcount = be32(file + pos + 8), acount = be32(file + pos + 12);
        pos += 20;
        uint32_t v = 0, a = 0;
        while ((v < vcount || a < acount) && pos + 8 <= n) {
            uint16_t id1 = be16(file + pos), id2 = be16(file + pos + 2);
            uint32_t size = be32(file + pos + 4);
            pos += 8;
            if (id1 == 1) {
                if (max_pics >= 0 && pics >= max_pics) goto done;
                ref_one(&c, id2, file + pos, size, scratch);
                if ((size_t)(pics + 1) * c.picsize <= out_cap)
                    memcpy(out + (size_t)pics * c.picsize, c.player.present, c.picsize);
                if (probe) {
                    VideoState *s = c.player.seqobj.state;
                    const uint8_t *base = scratch + 4;
                    BitBuffer *bb[20] = {
                        &s->basis_num[0].buf, &s->basis_num[1].buf,
                        &s->basis_num_run[0].buf, &s->basis_num_run[1].buf,
                        &s->dc_values[0].buf, &s->dc_values[1].buf, &s->dc_values[2].buf,
                        &s->bufTree0[0].buf, &s->bufTree0[1].buf, &s->bufTree0[2].buf,
                        &s->fixvl[0], &s->fixvl[1], &s->fixvl[2],
                        &s->dc_rle[0].buf, &s->dc_rle[1].buf, &s->dc_rle[2].buf,
                        &s->mv_h.buf, &s->mv_v.buf, &s->mcb_type.buf, &s->mcb_proc.buf };
                    for (int i = 0; i < 20; ++i)
                        probe[pics * 20 + i] = bb[i]->ptr ? (int32_t)((const uint8_t *)bb[i]->ptr - base) : -1;
                }
                ref_rotate_after(&c, id2);
                ++pics; ++v;
            } else {
                ++a;
            }
            pos += size;
        }
    }
done:
    free(scratch);
    ref_close(&c);
    return pics;
}

/* picture size in bytes for a clip header, 0 on error */
REF_API uint32_t ref_picsize(const uint8_t *file, size_t n)
{
    if (n < 0x44) return 0;
    uint32_t res = (uint32_t)be16(file + 0x34) * be16(file + 0x36);
    uint32_t ss = (uint32_t)file[0x38] * file[0x39];
    return ss ? res * (ss + 2) / ss : 0;
}

/*
 * CPU baseline timer: decode the clip `reps` times, timing ONLY the
 * HVQM4Decode* calls (no demux, no copies, no dumps).  Returns seconds;
 * *pixels receives luma pixels decoded in the timed calls.
 */
REF_API double ref_time_clip(const uint8_t *file, size_t n, int reps, uint64_t *pixels)
{
    double total = 0.0;
    uint64_t px = 0;
    for (int r = 0; r < reps; ++r) {
        RefCtx c;
        if (ref_open(&c, file, n)) return -1.0;
        uint32_t blocks = be32(file + 0x18);
        uint8_t *scratch = malloc(n + 64);
        size_t pos = 0x44;
        for (uint32_t b = 0; b < blocks && pos + 20 <= n; ++b) {
            uint32_t vcount = be32(file + pos + 8), acount = be32(file + pos + 12);
            pos += 20;
            uint32_t v = 0, a = 0;
            while ((v < vcount || a < acount) && pos + 8 <= n) {
                uint16_t id1 = be16(file + pos), id2 = be16(file + pos + 2);
                uint32_t size = be32(file + pos + 4);
                pos += 8;
                if (id1 == 1) {
                    Player *pl = &c.player;
                    memcpy(scratch, file + pos, size);
                    memset(scratch + size, 0, 8);
                    if (id2 != B_FRAME) { void *t = pl->past; pl->past = pl->future; pl->future = t; }
                    struct timespec t0, t1;
                    clock_gettime(CLOCK_MONOTONIC, &t0);
                    switch (id2) {
                    case I_FRAME: HVQM4DecodeIpic(&pl->seqobj, scratch + 4, pl->present); break;
                    case P_FRAME: HVQM4DecodePpic(&pl->seqobj, scratch + 4, pl->present, pl->past); break;
                    default:      HVQM4DecodeBpic(&pl->seqobj, scratch + 4, pl->present, pl->past, pl->future); break;
                    }
                    clock_gettime(CLOCK_MONOTONIC, &t1);
                    total += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
                    px += (uint64_t)pl->seqobj.width * pl->seqobj.height;
                    ref_rotate_after(&c, id2);
                    ++v;
                } else ++a;
                pos += size;
            }
        }
        free(scratch);
        ref_close(&c);
    }
    if (pixels) *pixels = px;
    return total;
}

/* ---- white-box entry points for known-answer tests ---- */

REF_API void ref_weight_block(uint8_t *dst16, uint8_t v, uint8_t t, uint8_t b, uint8_t l, uint8_t r)
{
    WeightImBlock(dst16, 4, v, t, b, l, r);
}

REF_API void ref_motion_comp(uint8_t *dst16, const uint8_t *src, uint32_t src_stride, uint32_t hx, uint32_t hy)
{
    _MotionComp(dst16, 4, src, src_stride, hx, hy);
}

REF_API void ref_tables(int32_t *div16, int32_t *mcdiv512)
{
    init_global_constants();
    memcpy(div16, divTable, sizeof divTable);
    memcpy(mcdiv512, mcdivTable, sizeof mcdivTable);
}

REF_API void ref_layout(uint32_t *out)
{
    out[0] = sizeof(VideoState);
    out[1] = offsetof(VideoState, padding);
    out[2] = sizeof(SeqObj);
    out[3] = sizeof(VideoInfo);
    out[4] = offsetof(VideoState, nest_data);
}

REF_API uint32_t ref_buffsize(uint16_t w, uint16_t h, uint8_t hs, uint8_t vs)
{
    SeqObj s; VideoInfo vi = { w, h, hs, vs, 0 };
    HVQM4InitSeqObj(&s, &vi);
    return HVQM4BuffSize(&s);
}

/* YUV 4:2:0 picture -> RGB24 through the reference's own dumpRGB (h4m:901-926): it writes a binary PPM,
 * which is read back.  Returns 0 on success. */
REF_API int ref_rgb(const uint8_t *yuv, uint16_t w, uint16_t h, uint8_t *rgb_out, const char *tmp_path)
{
    Player pl;
    memset(&pl, 0, sizeof pl);
    pl.seqobj.width = w; pl.seqobj.height = h; pl.seqobj.h_samp = 2; pl.seqobj.v_samp = 2;
    pl.present = (void *)yuv;
    dumpRGB(&pl, tmp_path);
    FILE *f = fopen(tmp_path, "rb");
    if (!f) return -1;
    unsigned fw = 0, fh = 0, maxv = 0;
    if (fscanf(f, "P6\n%u %u\n%u", &fw, &fh, &maxv) != 3 || fw != w || fh != h) { fclose(f); return -2; }
    fgetc(f);
    size_t n = fread(rgb_out, 1, (size_t)w * h * 3, f);
    fclose(f);
    remove(tmp_path);
    return n == (size_t)w * h * 3 ? 0 : -3;
}
