/*
 * examples/h4m_player.c -- a player written against the seven HVQM4 SDK entry points only, with the call
 * sequence and the picture-buffer rotation of the reference's player (Tilka/hvqm4 h4m_audio_decode.c: decv_init
 * h4m:2340-2350, main h4m:2409-2419, decode_video h4m:2078-2138).  Linked against libhvqm4_amd.so it decodes on
 * the MI355X; the same source compiles against the reference's own definitions.  The only non-SDK calls are the
 * container walk (hvq_h4m_*, the library's restatement of load_header / the record loop) and HVQM4ReleaseBuffer.
 *
 *   cc -O2 -Iinclude examples/h4m_player.c -Lhvqm4_amd -lhvqm4_amd -Wl,-rpath,$PWD/hvqm4_amd -o h4m_player
 *   ./h4m_player clip.h4m [out.yuv]      prints one line per picture: decode ordinal, type, display id, FNV-1a 64
 *                                        of the Y|U|V bytes; writes the pictures (decode order) to out.yuv
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hvqm4.h"
#include "hvqm4_amd.h"

typedef struct {                /* the reference's Player (h4m:2060-2076), minus audio */
    SeqObj seqobj;
    void *past, *present, *future;
} Player;

static uint64_t fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

/* decode_video of the reference (h4m:2078-2138): rotate, decode, rotate */
static int decode_video(Player *pl, int frame_type, const uint8_t *frame /* after disp_id */)
{
    if (frame_type != HVQ_FRAME_B) { void *t = pl->past; pl->past = pl->future; pl->future = t; }      /* h4m:2087-2093 */
    switch (frame_type) {
    case HVQ_FRAME_I: HVQM4DecodeIpic(&pl->seqobj, frame, pl->present); break;
    case HVQ_FRAME_P: HVQM4DecodePpic(&pl->seqobj, frame, pl->present, pl->past); break;
    case HVQ_FRAME_B: HVQM4DecodeBpic(&pl->seqobj, frame, pl->present, pl->past, pl->future); break;
    default: return -1;
    }
    return HVQM4GetLastError();
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s clip.h4m [out.yuv]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *file = malloc((size_t)n + 16);
    if (!file || fread(file, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "read failed\n"); return 2; }
    memset(file + n, 0, 16);
    fclose(f);
    FILE *out = argc > 2 ? fopen(argv[2], "wb") : NULL;

    HvqH4mInfo info;
    if (hvq_h4m_header(file, (size_t)n, &info)) { fprintf(stderr, "not an HVQM4 file\n"); return 1; }

    /* main (h4m:2409-2419) */
    Player pl;
    VideoInfo vi = { info.width, info.height, info.h_samp, info.v_samp, info.video_mode };
    HVQM4InitDecoder();
    if (HVQM4GetLastError()) { fprintf(stderr, "%s\n", HVQM4GetLastErrorString()); return 1; }
    HVQM4InitSeqObj(&pl.seqobj, &vi);
    VideoState *state = malloc(HVQM4BuffSize(&pl.seqobj));
    HVQM4SetBuffer(&pl.seqobj, state);
    state->padding[0] = info.is_1_5;                       /* the reference's 1.3 / 1.5 switch byte (h4m:2414-2417) */
    /* decv_init (h4m:2340-2350) */
    pl.past = calloc(1, info.pic_bytes); pl.present = calloc(1, info.pic_bytes); pl.future = calloc(1, info.pic_bytes);

    HvqH4mIter it;
    hvq_h4m_begin(&it);
    int type, rc, ordinal = 0;
    uint32_t disp;
    const uint8_t *pic;
    size_t len;
    while ((rc = hvq_h4m_next(file, (size_t)n, &it, &type, &disp, &pic, &len)) == 1) {
        int err = decode_video(&pl, type, pic);
        if (err) { fprintf(stderr, "picture %d: error %d: %s\n", ordinal, err, HVQM4GetLastErrorString()); return 1; }
        printf("%d %02x %u %016llx\n", ordinal, type, disp, (unsigned long long)fnv1a(pl.present, info.pic_bytes));
        if (out) fwrite(pl.present, 1, info.pic_bytes, out);
        if (type != HVQ_FRAME_B) { void *t = pl.present; pl.present = pl.future; pl.future = t; }      /* h4m:2131-2137 */
        ++ordinal;
    }
    if (rc < 0) { fprintf(stderr, "container error\n"); return 1; }
    if (out) fclose(out);
    HVQM4ReleaseBuffer(&pl.seqobj);
    free(state); free(pl.past); free(pl.present); free(pl.future); free(file);
    return 0;
}
