/*
 * examples/h4m_batch.c -- the throughput path from C: N concurrent streams of one .h4m file, entropy parse on the GPU,
 * batches of one GOP per stream, streamed (the next batch is copied and uploaded while the batch in flight is parsed):
 *
 *     hvq_submit_many_device(batch 0); hvq_flush_begin();
 *     loop: hvq_submit_many_device(next batch); hvq_flush_end(); hvq_flush_begin();
 *
 *   cc -O2 -Iinclude examples/h4m_batch.c -Lhvqm4_amd -lhvqm4_amd -Wl,-rpath,$PWD/hvqm4_amd -o h4m_batch
 *   ./h4m_batch clip.h4m [streams=16] [host|gpu]
 * prints the FNV-1a 64 of the last picture of every stream (all equal: the streams decode the same clip) and the rate.
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
    if (argc < 2) { fprintf(stderr, "usage: %s clip.h4m [streams] [host|gpu]\n", argv[0]); return 2; }
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
    /* the clip's pictures, once */
    enum { MAXPIC = 4096 };
    static int types[MAXPIC];
    static const uint8_t *pics[MAXPIC];
    static size_t lens[MAXPIC];
    int npic = 0, type, rc;
    uint32_t disp;
    HvqH4mIter it;
    hvq_h4m_begin(&it);
    while (npic < MAXPIC && (rc = hvq_h4m_next(file, (size_t)n, &it, &type, &disp, &pics[npic], &lens[npic])) == 1) types[npic++] = type;
    if (npic == 0) { fprintf(stderr, "no pictures\n"); return 1; }

    HvqContext *ctx;
    CHECK(hvq_context_create(0, &ctx));
    int *sid = malloc(sizeof(int) * (size_t)nstreams);
    for (int s = 0; s < nstreams; ++s) CHECK(sid[s] = hvq_stream_open(ctx, info.width, info.height, info.h_samp, info.v_samp, info.is_1_5, 6));

    /* batch = up to 16 consecutive pictures of every stream, picture-major like a player would submit them */
    const int per = 16;
    int *b_sid = malloc(sizeof(int) * (size_t)(nstreams * per)), *b_ft = malloc(sizeof(int) * (size_t)(nstreams * per));
    const uint8_t **b_pic = malloc(sizeof(*b_pic) * (size_t)(nstreams * per));
    size_t *b_len = malloc(sizeof(size_t) * (size_t)(nstreams * per));
    const double t0 = now();
    int in_flight = 0;
    for (int at = 0; at < npic; at += per) {
        int m = 0;
        for (int k = at; k < at + per && k < npic; ++k)
            for (int s = 0; s < nstreams; ++s) { b_sid[m] = sid[s]; b_ft[m] = types[k]; b_pic[m] = pics[k]; b_len[m] = lens[k]; ++m; }
        if (gpu_parse) CHECK(hvq_submit_many_device(ctx, m, b_sid, b_ft, b_pic, b_len, NULL));
        else CHECK(hvq_submit_many(ctx, m, b_sid, b_ft, b_pic, b_len, 8, NULL));
        if (in_flight) CHECK(hvq_flush_end(ctx));             /* the batch submitted one round earlier */
        CHECK(hvq_flush_begin(ctx));
        in_flight = 1;
    }
    CHECK(hvq_flush_end(ctx));
    CHECK(hvq_sync(ctx));
    const double dt = now() - t0;

    uint8_t *yuv = malloc(info.pic_bytes);
    for (int s = 0; s < nstreams; ++s) {
        CHECK(hvq_read_picture(ctx, sid[s], npic - 1, yuv, info.pic_bytes));
        printf("stream %d last picture %016llx\n", s, (unsigned long long)fnv1a(yuv, info.pic_bytes));
    }
    fprintf(stderr, "%d streams x %d pictures %ux%u, %s parse: %.1f Mpixel/s\n", nstreams, npic, info.width, info.height,
            gpu_parse ? "GPU" : "host", (double)nstreams * npic * info.width * info.height / dt / 1e6);
    hvq_context_destroy(ctx);
    return 0;
}
