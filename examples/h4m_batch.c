/*
 * examples/h4m_batch.c -- the throughput path from C: N concurrent streams of one .h4m file, entropy parse on the GPU,
 * batches of one GOP per stream, streamed, and EVERY picture brought back to the host in DISPLAY order:
 *
 *     hvq_submit_many_device_async(batch 0); hvq_flush_begin();
 *     loop: hvq_submit_many_device_async(batch k + 1);     copied and uploaded while batch k is parsed
 *           hvq_flush_end(batch k);                   reconstruction launched
 *           hvq_flush_begin(batch k + 1);             parse queued
 *           hvq_read_pictures(batch k);               one synchronisation; the copies run beside the parse of batch k + 1
 *
 * Display order (h4m:2085, 2122): a picture's display index is gop_start + disp_id, which hvq_h4m_next returns.
 *
 *   cc -O2 -Iinclude examples/h4m_batch.c -Lhvqm4_amd -lhvqm4_amd -Wl,-rpath,$PWD/hvqm4_amd -o h4m_batch
 *   ./h4m_batch clip.h4m [streams=16] [host|gpu] [out.yuv]
 * prints, per display index, the decode ordinal and the FNV-1a 64 of the picture (stream 0; all streams are checked to be
 * equal -- they decode the same clip), writes stream 0's pictures in display order to out.yuv, and reports the rate.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "hvqm4_amd.h"

static uint64_t fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s: %d: %s\n", #call, rc_, hvq_last_error_string()); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s clip.h4m [streams] [host|gpu] [out.yuv]\n", argv[0]); return 2; }
    const int nstreams = argc > 2 ? atoi(argv[2]) : 16;
    const int gpu_parse = argc > 3 ? strcmp(argv[3], "host") != 0 : 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *file = malloc((size_t)n + 16);
    if (!file || fread(file, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "read failed\n"); return 2; }
    memset(file + n, 0, 16);
    fclose(f);

    HvqH4mInfo info;
    CHECK(hvq_h4m_header(file, (size_t)n, &info));
    /* the clip's pictures, once, with their display indices */
    enum { MAXPIC = 4096 };
    static int types[MAXPIC];
    static uint32_t disp[MAXPIC];
    static const uint8_t *pics[MAXPIC];
    static size_t lens[MAXPIC];
    int npic = 0, type, rc;
    uint32_t d;
    HvqH4mIter it;
    hvq_h4m_begin(&it);
    while (npic < MAXPIC && (rc = hvq_h4m_next(file, (size_t)n, &it, &type, &d, &pics[npic], &lens[npic])) == 1) { types[npic] = type; disp[npic++] = d; }
    if (npic == 0) { fprintf(stderr, "no pictures\n"); return 1; }
    /* decode ordinal of every display index */
    int *by_disp = malloc(sizeof(int) * (size_t)npic);
    for (int k = 0; k < npic; ++k) by_disp[k] = -1;
    for (int k = 0; k < npic; ++k) if (disp[k] < (uint32_t)npic) by_disp[disp[k]] = k;
    for (int k = 0; k < npic; ++k) if (by_disp[k] < 0) { fprintf(stderr, "display index %d is missing\n", k); return 1; }

    const int per = 16;                                       /* pictures of a stream per batch */
    HvqContext *ctx;
    CHECK(hvq_context_create(0, &ctx));
    int *sid = malloc(sizeof(int) * (size_t)nstreams);
    /* two batches of a stream are resident at a time (the one being read and the one queued behind it), plus the anchors */
    for (int s = 0; s < nstreams; ++s) CHECK(sid[s] = hvq_stream_open(ctx, info.width, info.height, info.h_samp, info.v_samp, info.is_1_5, 2 * per + 4));

    const size_t cap = (size_t)nstreams * per;
    int *b_sid = malloc(sizeof(int) * cap), *b_ft = malloc(sizeof(int) * cap), *r_sid = malloc(sizeof(int) * cap), *r_ord = malloc(sizeof(int) * cap);
    const uint8_t **b_pic = malloc(sizeof(*b_pic) * cap);
    size_t *b_len = malloc(sizeof(size_t) * cap);
    void **r_dst = malloc(sizeof(void *) * cap);
    /* all pictures of all streams, decode order, in pinned memory: the copies are asynchronous DMA */
    uint8_t *all = hvq_pinned_alloc((size_t)nstreams * npic * info.pic_bytes);
    if (!all) { fprintf(stderr, "%s\n", hvq_last_error_string()); return 1; }
    const double t0 = now();
    const int nbatch = (npic + per - 1) / per;
    for (int b = 0; b <= nbatch; ++b) {                          /* round b: submit batch b, launch and read batch b - 1 */
        const int at = b * per;
        if (b < nbatch) {
            int m = 0;
            for (int k = at; k < at + per && k < npic; ++k)          /* picture-major, like a player would submit them */
                for (int s = 0; s < nstreams; ++s) { b_sid[m] = sid[s]; b_ft[m] = types[k]; b_pic[m] = pics[k]; b_len[m] = lens[k]; ++m; }
            /* the pictures point into the clip, which outlives the loop: the deferred copy (a worker thread of the library) is safe */
            if (gpu_parse) CHECK(hvq_submit_many_device_async(ctx, m, b_sid, b_ft, b_pic, b_len, NULL));
            else CHECK(hvq_submit_many(ctx, m, b_sid, b_ft, b_pic, b_len, 8, NULL));
        }
        if (b > 0) CHECK(hvq_flush_end(ctx));                     /* batch b - 1: reconstruction launched */
        if (b < nbatch) CHECK(hvq_flush_begin(ctx));              /* batch b: parse queued */
        if (b > 0) {                                              /* ... and batch b - 1 read, beside that parse */
            int r = 0;
            for (int s = 0; s < nstreams; ++s)                    /* stream-major: consecutive destinations, few large copies */
                for (int k = at - per; k < at && k < npic; ++k) { r_sid[r] = sid[s]; r_ord[r] = k; r_dst[r] = all + ((size_t)s * npic + k) * info.pic_bytes; ++r; }
            CHECK(hvq_read_pictures(ctx, r, r_sid, r_ord, r_dst));
        }
    }
    const double dt = now() - t0;

    FILE *out = argc > 4 ? fopen(argv[4], "wb") : NULL;
    for (int k = 0; k < npic; ++k) {                              /* display order */
        const int o = by_disp[k];
        const uint8_t *p0 = all + (size_t)o * info.pic_bytes;
        for (int s = 1; s < nstreams; ++s)
            if (memcmp(p0, all + ((size_t)s * npic + o) * info.pic_bytes, info.pic_bytes)) { fprintf(stderr, "stream %d differs at picture %d\n", s, o); return 1; }
        printf("display %d ordinal %d type %02x %016llx\n", k, o, types[o], (unsigned long long)fnv1a(p0, info.pic_bytes));
        if (out) fwrite(p0, 1, info.pic_bytes, out);
    }
    if (out) fclose(out);
    fprintf(stderr, "%d streams x %d pictures %ux%u, %s parse, every picture read back: %.1f Mpixel/s\n", nstreams, npic, info.width, info.height,
            gpu_parse ? "GPU" : "host", (double)nstreams * npic * info.width * info.height / dt / 1e6);
    hvq_pinned_free(all);
    hvq_context_destroy(ctx);
    return 0;
}
