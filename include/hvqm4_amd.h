/*
 * hvqm4_amd.h -- batched, device-resident extension of the HVQM4 decode path.
 *
 * The SDK signatures (hvqm4.h) hand host pointers in and out, so every call pays PCIe for
 * whole pictures.  This API is what a player/transcoder binds for throughput: pictures stay
 * in HBM, any number of streams are decoded per launch, and the host entropy parse
 * (h4m_audio_decode.c:1970-2056, serial per stream) overlaps GPU work.  It replaces the
 * reference's `decode_video` loop (h4m:2078-2138): frame-buffer rotation becomes slot
 * assignment, and pictures with no mutual dependency (across streams, and B pictures /
 * the next anchor inside a stream) are reconstructed by ONE kernel launch.
 *
 * Plain C ABI: pointers and sizes only.  All functions return HVQ_OK (0) / a non-negative
 * result, or a negative HVQ_E_* code; hvq_last_error_string() describes the last failure.
 */
#ifndef HVQM4_AMD_H
#define HVQM4_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVQ_OK            0
#define HVQ_E_ARG        -1
#define HVQ_E_OVERFLOW   -2
#define HVQ_E_GEOMETRY   -3
#define HVQ_E_NOGPU      -4
#define HVQ_E_HIP        -5
#define HVQ_E_STATE      -6
#define HVQ_E_CONTAINER  -7   /* malformed .h4m file (every case the reference exits on) */
#define HVQ_E_UNSUPPORTED -8  /* a picture this back end refuses rather than decode differently from the reference: malformed input
                                 that had to be clamped (HVQM4_AMD_ALLOW_CLAMPED=1 decodes it), an overflow-symbol run that does not
                                 end inside its section, or a P picture with future-referencing macroblocks (h4m:2058-2061, decoded
                                 like the reference since round 3) whose previous buffer content has left the stream's slot ring.
                                 The picture is not decoded, `present` is untouched, the stream resumes at its next I picture; other
                                 streams of the batch are unaffected. */

#define HVQ_FRAME_I 0x10   /* container frame ids, h4m:2065-2070 */
#define HVQ_FRAME_P 0x20
#define HVQ_FRAME_B 0x30

typedef struct HvqContext HvqContext;

typedef struct HvqStats {
    uint64_t pictures;          /* pictures in the last flushed batch */
    uint64_t luma_pixels;       /* sum of w*h over them */
    uint64_t algorithmic_bytes; /* 1.5 B/px written + 1.5 B/px read for P/B */
    uint64_t descriptor_bytes;  /* blob bytes uploaded (not credited in the roofline) */
    uint32_t launches;          /* kernel launches per pass: dependency levels x launch queues */
    uint32_t workgroups;        /* total workgroups per pass */
    double   parse_seconds;     /* host entropy-parse time accumulated by hvq_stream_submit */
    uint32_t flags_or;          /* OR of all blob header flags (HVQ_F_*) */
    uint32_t gpu_parsed;        /* pictures of the batch whose bitstream was parsed on the GPU */
    double   gpu_parse_ms;      /* device time of that parse launch (HIP events) */
    uint32_t gpu_parse_retried; /* of those, pictures the flat parse path handed to the chain decoder (unusual section layout,
                                   capacities, overflow groups at the caps) -- same result, slower */
    uint32_t dropped;           /* pictures of the batch that were not reconstructed: rejected, or behind a rejected picture of their stream */
    uint32_t launch_queues;     /* 1, or 2: a batch of 16 streams or more of one picture size deals its streams -- by work -- to two chains of
                                   launches on two HIP streams (a hardware queue each) that run side by side; HVQM4_AMD_QUEUES=1 / 2 forces either */
    uint64_t queue_bytes;       /* always 0 since round 6 (rounds 3-5: bytes of the two-pass variant's tile queues); kept for the layout */
    uint64_t copy_bytes;        /* hvq_submit_many_device / _async: bitstream bytes the library copied into its pinned arena, summed since
                                   the context was created (hvq_submit_many_arena copies nothing) */
    double   copy_seconds;      /* ... and the wall time its copy threads took over them: host_copy GB/s = copy_bytes / copy_seconds */
} HvqStats;

int  hvq_context_create(int device, HvqContext **out);
/* Launch queues a batch of this context may use: 2 (default: a batch of 16 streams or more of one picture size deals its dependency levels
 * to two HIP streams, HvqStats.launch_queues) or 1.  A process that runs TWO contexts on a GPU side by side (INTEGRATION.md "Two contexts per
 * GPU") sets 1 on both: the contexts are each other's second queue, four queues get in each other's way (157 against 171 Gpixel/s). */
int  hvq_context_set_launch_queues(HvqContext *ctx, int n);
void hvq_context_destroy(HvqContext *ctx);

/* A stream = one clip.  `nslots` >= 3 picture buffers stay resident in HBM per stream (the
 * reference rotates exactly 3, h4m:2340-2350; more slots let later pictures start earlier). */
int  hvq_stream_open(HvqContext *ctx, int width, int height, int h_samp, int v_samp, int is_1_5, int nslots);
int  hvq_stream_close(HvqContext *ctx, int stream);
/* Host threads that share the entropy parse of ONE picture of this stream (its sections are independent bit buffers, h4m:1981-1993,
 * 2030-2044; same blob byte for byte): 1 (default) .. 8.  For callers with few streams -- hvq_stream_submit parses a stream's pictures one
 * after the other, hvq_submit_many parses STREAMS side by side.  The SDK entry points set 4 (HVQM4_AMD_SDK_PARSE_THREADS).  Returns the
 * count in effect. */
int  hvq_stream_set_parse_threads(HvqContext *ctx, int stream, int threads);
/* Host only: bytes of the picture ring hvq_stream_open would allocate ((nslots + 1) slots; 0: geometry refused).  The kernels
 * address reference pictures as ring base + 32-bit offset, so hvq_stream_open fails with HVQ_E_OVERFLOW from 4 GiB on. */
uint64_t hvq_stream_ring_bytes(int width, int height, int h_samp, int v_samp, int nslots);

/* Parse one picture (host) and queue it.  `pic` = picture data after the 4-byte disp_id,
 * `len` its length.  Returns the picture's ordinal in the stream (decode order). */
int  hvq_stream_submit(HvqContext *ctx, int stream, int frame_type, const uint8_t *pic, size_t len);

/* Same as hvq_stream_submit for `n` pictures, with the host entropy parse spread over `threads` worker
 * threads (pictures of one stream are parsed in order by one worker -- the parser carries the nest of the
 * last I picture; different streams run concurrently).  Pictures are queued in array order.  ordinals[i]
 * (may be NULL) receives picture i's ordinal.  Returns HVQ_OK or the first error. */
int  hvq_submit_many(HvqContext *ctx, int n, const int *streams, const int *frame_types,
                     const uint8_t *const *pics, const size_t *lens, int threads, int *ordinals);

/* GPU entropy parse: queue RAW picture bitstreams; hvq_flush uploads them and parses them on
 * the device (one workgroup per picture, hvq_gparse.hip), so no host core touches a bit of the stream.  Same
 * queueing semantics as hvq_submit_many.  A stream uses either the host parser or the GPU parser for its whole
 * lifetime (the host parser keeps the nest of the last I picture); lens[] must be the real picture lengths.
 * Errors of the device parse (HVQ_E_OVERFLOW, HVQ_E_ARG) are reported by hvq_flush.
 * The bitstreams are copied into the library's pinned arena before the call returns: `pics[i]` are the caller's again on return. */
int  hvq_submit_many_device(HvqContext *ctx, int n, const int *streams, const int *frame_types,
                            const uint8_t *const *pics, const size_t *lens, int *ordinals);
/* The same with the copy DEFERRED to a worker thread of the library: the call returns at once, so a streaming caller reaches
 * hvq_flush_end (and the batch in flight its reconstruction launches) without waiting for 160 MB of host memcpy.  `pics[i]` (and
 * the arrays) must stay readable and unchanged until the next hvq_flush_begin or hvq_sync, which joins the worker; an upload that
 * fails on the worker is reported there and drops the whole queued batch (its streams resume at their next I picture).
 * (Round 4 did this inside hvq_submit_many_device whenever a batch was in flight; since round 5 it is opt-in by name.
 * HVQM4_AMD_ASYNC_SUBMIT=1 restores the old behaviour of the plain call.) */
int  hvq_submit_many_device_async(HvqContext *ctx, int n, const int *streams, const int *frame_types,
                                  const uint8_t *const *pics, const size_t *lens, int *ordinals);
/* Zero-copy submit.  hvq_arena_reserve hands out `bytes` of the pinned (DMA-able) arena the library uploads from; the caller
 * writes its pictures there itself -- a container reader read()s file bytes straight into it -- at 256-byte aligned, ascending
 * offsets, picture i occupying hvq_arena_stride(lens[i]) bytes (its length plus the zero padding the device reader needs, which
 * the library writes).  hvq_submit_many_arena then queues them without touching a byte: no memcpy, no second pass over the
 * host's memory.  One reservation at a time; the pointer is valid until its submit (or the next hvq_flush_begin, which drops an
 * unsubmitted reservation).  Same queueing semantics and errors as hvq_submit_many_device. */
int  hvq_arena_reserve(HvqContext *ctx, size_t bytes, void **ptr);
size_t hvq_arena_stride(size_t len);
int  hvq_submit_many_arena(HvqContext *ctx, int n, const int *streams, const int *frame_types,
                           const size_t *offsets, const size_t *lens, int *ordinals);

/* Upload queued descriptors, group queued pictures into dependency levels, launch. Async. */
int  hvq_flush(HvqContext *ctx);

/* The same in two halves, for streaming.  hvq_flush_begin queues what needs no answer from the GPU (uploads, the
 * entropy-parse kernel) and returns; the queued batch is now "in flight" and the caller may already submit the NEXT
 * batch -- its bitstreams are copied and uploaded (second arena, copy stream) while this one is parsed.
 * hvq_flush_end takes the parse results, builds the launch tables and launches the reconstruction.  One batch can be
 * in flight; every call that needs its pictures (sync, read, replay, stats, close) ends it implicitly. */
int  hvq_flush_begin(HvqContext *ctx);
int  hvq_flush_end(HvqContext *ctx);
/* The streaming step in one call: end the batch in flight and begin the queued one -- hvq_flush_end + hvq_flush_begin in effect,
 * errors and state included (the first error of either half is returned) -- but with the queued batch's parse kernel launched
 * BEFORE the host takes the results of the batch in flight, so that the GPU parses batch k + 1 while the host builds the tables
 * of batch k, whose reconstruction launches line up behind that parse.  Taken when both batches are GPU-parsed throughout and the
 * queued batch's bitstreams are in the arena before the parse results of the batch in flight arrive; in every other case the two
 * halves run in the plain order.  A streaming loop is: submit(k + 1) [deferred or zero-copy], hvq_flush_next, use batch k.
 * HVQM4_AMD_FLUSH_NEXT=0: always the plain order. */
int  hvq_flush_next(HvqContext *ctx);
int  hvq_sync(HvqContext *ctx);

/* Re-run the launches of the last flush `reps` times (descriptors already resident in HBM), the launch queues running free.
 * *gpu_ms = elapsed time between HIP events recorded on the launch stream around all reps. */
int  hvq_replay(HvqContext *ctx, int reps, float *gpu_ms);
/* What a NEW batch costs behind its parse, repeated `reps` times: what = 1 -- every repetition forks and joins the launch queues
 * exactly as a flush does (the reconstruction stage of the product; bench.py's timed step); what = 0 -- hvq_replay: the launch queues
 * fork once and join once around all repetitions (a queue that is ahead runs into the next pass); what = 2 -- nothing, 0 ms (it timed
 * the queue build of the two-pass variant, deleted in round 6). */
int  hvq_replay_stage(HvqContext *ctx, int reps, int what, float *gpu_ms);

/* Copy a still-resident picture (Y|U|V, pic_bytes) to host memory; synchronises. */
int  hvq_read_picture(HvqContext *ctx, int stream, int ordinal, void *dst, size_t cap);
uint32_t hvq_stream_pic_bytes(HvqContext *ctx, int stream);

/* Bulk readback for the throughput path: `n` resident pictures to host memory, all copies queued on a stream of their own behind
 * the reconstruction, ONE synchronisation.  dst[i] receives hvq_stream_pic_bytes() bytes; pinned destinations
 * (hvq_pinned_alloc) make the copies asynchronous DMA. */
int  hvq_read_pictures(HvqContext *ctx, int n, const int *streams, const int *ordinals, void *const *dst);
void *hvq_pinned_alloc(size_t bytes);
void hvq_pinned_free(void *p);
/* Device address of a resident picture for consumers on the GPU (valid until the stream's ring reuses the slot, `nslots`
 * pictures later at the earliest); order the consumer after hvq_sync(). */
int  hvq_picture_device_ptr(HvqContext *ctx, int stream, int ordinal, const void **ptr);

/* Display epilogue of the reference player (dumpRGB, h4m:897-926) on the GPU: converts a resident 4:2:0
 * picture to interleaved RGB24 (w*h*3 bytes, float math identical to the reference) and copies it to host. */
int  hvq_read_picture_rgb(HvqContext *ctx, int stream, int ordinal, void *dst, size_t cap);
/* The same conversion for a picture in host memory (Y|U|V 4:2:0, width a multiple of 4, height even): upload, convert, download. */
int  hvq_convert_yuv420_rgb(HvqContext *ctx, const void *yuv, int width, int height, void *rgb);
/* Measurement helper: converts the newest resident picture of every open 4:2:0 stream in ONE launch, `reps`
 * times, timed with HIP events on the launch stream.  bytes_per_rep = 1.5 B/px read + 3 B/px written. */
int  hvq_rgb_bench(HvqContext *ctx, int reps, float *gpu_ms, uint64_t *bytes_per_rep, uint32_t *pictures);

/* Measurement helper: `reps` copies of `bytes` from pinned host memory to the device on the context's copy stream, HIP-event timed:
 * the PCIe rate the upload of a batch's bitstreams can reach on this box (GB/s, 1e9). */
int  hvq_h2d_probe(HvqContext *ctx, size_t bytes, int reps, double *gb_per_s);

int  hvq_get_stats(HvqContext *ctx, HvqStats *out);
/* self-test of the kernels' replacement for the reference's division tables (h4m:265-273): out[0..15] = 256 / d, out[16..271] = 4096 / d
 * as the device computes them (0 for d = 0) */
int  hvq_debug_table_divisions(HvqContext *ctx, uint32_t *out);
const char *hvq_last_error_string(void);

/* .h4m container demux in memory (header checks of load_header h4m:2175-2247, block/record walk of h4m:2427-2537).
 * Host only.  Typical use:
 *     HvqH4mInfo info; hvq_h4m_header(file, n, &info);
 *     int sid = hvq_stream_open(ctx, info.width, info.height, info.h_samp, info.v_samp, info.is_1_5, 6);
 *     HvqH4mIter it; hvq_h4m_begin(&it);
 *     while (hvq_h4m_next(file, n, &it, &type, &disp, &pic, &len) == 1) hvq_stream_submit(ctx, sid, type, pic, len);
 *     hvq_flush(ctx);                                                                                         */
typedef struct HvqH4mInfo {
    uint32_t header_size, body_size, blocks, video_frames, audio_frames, usec_per_frame, max_frame_size;
    uint32_t pic_bytes;            /* w*h*(hs*vs+2)/(hs*vs), h4m:2343-2345 */
    uint16_t width, height;
    uint8_t  h_samp, v_samp, video_mode, is_1_5;
} HvqH4mInfo;
typedef struct HvqH4mIter {
    size_t   pos, block_end;
    uint32_t block, v_left, a_left, video_seen, gop_start, in_block;
} HvqH4mIter;
int  hvq_h4m_header(const uint8_t *data, size_t n, HvqH4mInfo *out);
void hvq_h4m_begin(HvqH4mIter *it);
/* 1: produced a video picture (`pic` = data after the disp_id word, `disp_id` = gop_start + record disp_id);
 * 0: clean end of file; < 0: HVQ_E_CONTAINER */
int  hvq_h4m_next(const uint8_t *data, size_t n, HvqH4mIter *it, int *frame_type, uint32_t *disp_id,
                  const uint8_t **pic, size_t *len);

/* Host-only pieces, usable without a GPU (parse is pixel-independent). */
typedef struct HvqParser HvqParser;
HvqParser *hvq_parser_create(int width, int height, int h_samp, int v_samp, int is_1_5);
void hvq_parser_destroy(HvqParser *p);
size_t hvq_parser_blob_bound(const HvqParser *p);
uint32_t hvq_parser_pic_bytes(const HvqParser *p);
int  hvq_parse_picture(HvqParser *p, int frame_type, const uint8_t *pic, size_t len,
                       uint8_t *blob, size_t cap, size_t *blob_len);
/* Length of a picture from its own section table (what the SDK entry points, whose signatures carry no length, parse with).
 * `limit` = readable bytes at `pic` (0 = unknown: the table's words are trusted like the reference trusts them). */
int  hvq_picture_length(const uint8_t *pic, int frame_type, uint32_t limit, size_t *len);

#ifdef __cplusplus
}
#endif
#endif
