/*
 * hvq_runtime.cpp -- host side of the MI355X HVQM4 back end: the C ABI of include/hvqm4.h
 * (the reference's SDK entry points, h4m_audio_decode.c:275, 819, 828, 957, 1970, 2018, 2058)
 * and include/hvqm4_amd.h (batched device-resident path).
 *
 * Picture-buffer rotation of the reference player (h4m:2087-2093, 2131-2137) is restated as
 * slot assignment: every decoded picture gets a slot in an HBM-resident ring of its stream,
 * anchors (I/P) stay pinned while they are the past/future reference.  Queued pictures are
 * grouped into dependency LEVELS (RAW on reference slots, WAR/WAW on the destination slot);
 * one kernel launch reconstructs one level across all streams.  Workgroups are dealt to the
 * tile table so that all tiles of a picture land on one XCD (blockIdx % 8), keeping its map,
 * nest and reference-picture reads in that XCD's L2.
 *
 * There is NO CPU reconstruction path: without a usable HIP device every entry point that
 * would produce pixels fails with HVQ_E_NOGPU.
 */
#include <hip/hip_runtime.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/hvqm4.h"
#include "../../include/hvqm4_amd.h"
#include "hvq_desc.h"
#include "hvq_parse.h"
#include "hvq_gparse_core.h"

#ifdef GP_PROBE
extern "C" hipError_t hvq_launch_parse_probe(const HvqParseJob *jobs_dev, HvqParseResult *results_dev, uint32_t n, uint32_t rowbuf_stride,
                                             uint32_t exit_at, hipStream_t stream);
#endif
extern "C" hipError_t hvq_launch_parse(const HvqParseJob *jobs_dev, HvqParseResult *results_dev, uint32_t n,
                                       uint32_t rowbuf_stride, uint32_t use_flat, const uint32_t *redo_dev, uint64_t *timing_dev,
                                       hipStream_t stream);
extern "C" hipError_t hvq_launch_nest_commit(const uint64_t *pairs_dev, uint32_t n, hipStream_t stream);
extern "C" uint32_t hvq_gparse_scratch_bytes(uint32_t total_blocks, uint32_t total_runs, uint32_t nmb);
extern "C" int hvq_parse_occupancy(uint32_t rowbuf_stride);

extern "C" hipError_t hvq_launch_recon_inline(const HvqJob *jobs_dev, uint32_t nslots, uint32_t max_wgs, uint32_t tiles_per_wg,
                                              uint32_t items_cap, uint32_t pair_cap, uint32_t pool_cap, hipStream_t stream);
extern "C" uint32_t hvq_recon_inline_static_lds(uint32_t tiles_per_wg, uint32_t items_cap);
extern "C" uint32_t hvq_recon_inline_dyn_lds(uint32_t pair_cap, uint32_t pool_cap);
extern "C" hipError_t hvq_launch_gather(const uint64_t *src_dev, uint8_t *dst_dev, uint32_t n, uint32_t pic_bytes, hipStream_t stream);
extern "C" hipError_t hvq_launch_upload(const void *src_pinned, void *dst_dev, size_t bytes, hipStream_t stream);
extern "C" hipError_t hvq_launch_selfref(const HvqJob *job_dev, const uint8_t *side, uint8_t *dst, hipStream_t stream);
extern "C" hipError_t hvq_launch_table_div(uint32_t *out_dev, hipStream_t stream);

#ifdef HVQ_STAMPS
extern "C" void hvq_set_stamps(unsigned long long *p);
#endif
extern "C" hipError_t hvq_launch_rgb(const void *jobs_dev, int njobs, int max_lanes, int wide, hipStream_t stream);
struct RgbJob { const uint8_t *yuv; uint8_t *rgb; int w, h; };

#define HVQ_EXPORT extern "C" __attribute__((visibility("default")))

static thread_local std::string g_err;
static thread_local int g_sdk_err = 0;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(HVQ_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

HVQ_EXPORT const char *hvq_last_error_string(void) { return g_err.c_str(); }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* workgroups of a picture: one per pair of consecutive tiles of a plane (the kernel's mapping) */
static uint32_t picture_workgroups(const uint32_t tile_first[4])
{
    uint32_t n = 0;
    for (int k = 0; k < 3; ++k) n += (tile_first[k + 1] - tile_first[k] + 1) / 2;
    return n;
}

/* Pictures this back end refuses instead of decoding them differently from the reference:
 *   HVQ_F_CAPPED    an overflow-symbol run ended on the parsers' cap: the reference would have gone on summing (h4m:654-677);
 *   HVQ_F_CLAMPED   a nest origin or vector target outside what the reference's arithmetic keeps in bounds was clamped
 *                   (malformed input); HVQM4_AMD_ALLOW_CLAMPED=1 decodes such pictures with the clamped values. */
static const char *unsupported_reason(uint32_t flags)
{
    if (flags & HVQ_F_CAPPED) return "an overflow-symbol run is longer than this back end follows (the reference sums for as long as the stream says, h4m:654-677)";
    if (flags & HVQ_F_CLAMPED) {
        const char *e = getenv("HVQM4_AMD_ALLOW_CLAMPED");
        if (!(e && atoi(e) > 0)) return "malformed picture: a nest origin or vector target had to be clamped (HVQM4_AMD_ALLOW_CLAMPED=1 decodes it anyway)";
    }
    return nullptr;
}

/* ------------------------------------------------------------------ context */
struct Slot {
    int w_level = -1;   /* level (in the pending batch) of the launch that writes the current content */
    int r_level = -1;   /* highest level that reads it */
    int pic = -1;       /* ordinal of the occupant */
};

struct Stream {
    bool open = false;
    HvqParser *parser = nullptr;
    int w = 0, h = 0;
    uint32_t pic_bytes = 0, slot_bytes = 0;
    uint8_t *dev = nullptr;                  /* (nslots + 1) slots; the last one stays zero */
    std::vector<Slot> slots;
    int anchor_old = -1, anchor_new = -1;    /* "past" / "future" of the reference player */
    int ring = 0;
    int npics = 0;
    std::vector<int> pic_slot;
    /* GPU entropy parse (hvq_submit_many_device): the stream's pictures never meet the host parser */
    int parse_mode = 0;                      /* 0 undecided, 1 host parser, 2 device parser */
    HvqPicHeader layout{};                   /* geometry part of every blob of this stream */
    uint32_t blob_cap = 0, scratch_bytes = 0;
    uint8_t *nest_keep = nullptr;            /* two slots: [nest_cur] = packed nest of the last I picture of earlier
                                                batches (zero before any); a flush commits into the other slot */
    int nest_cur = 0;
    uint8_t *nest_keep_ptr(int which) const { return nest_keep + (size_t)which * GP_ALIGN16(HVQ_NESTP_BYTES); }
    int nest_src = -1;                       /* pending index of the last I picture queued in this batch, -1: nest_keep */
    bool need_I = false;                     /* a picture of this stream was rejected: P/B pictures are refused until the next I picture */
    /* the reference player's three buffers (h4m:2087-2093, 2131-2137), as ordinals of the pictures they hold (-1: never written):
     * a P picture with future-referencing macroblocks reads what `present` held before (h4m:2058-2061) */
    int rp_past = -1, rp_future = -1, rp_present = -1;
    const void *sdk_present = nullptr;       /* SDK path: the caller's `present` buffer of the picture being submitted */
    int inflight_from = 0x7FFFFFFF;          /* ordinals >= this belong to the batch between hvq_flush_begin and hvq_flush_end */
    uint8_t *slot_ptr(int s) const { return dev + (size_t)(s < 0 ? (int)slots.size() : s) * slot_bytes; }
};

struct Pending {
    uint32_t max_items, max_pairs;
    int stream, ordinal, level;
    size_t blob_off, blob_len;
    int dst, ref0, ref1;
    uint32_t ntiles, kind;
    uint32_t nwg;                      /* workgroups: pairs of consecutive tiles, plane by plane */
    uint32_t w, h;
    /* device-parsed pictures: raw bitstream in the arena instead of a blob */
    bool dev = false;
    int nest_ref = -1;                 /* pending index of the governing I picture, -1: the stream's nest_keep */
    uint64_t dev_blob = 0, dev_nest = 0, nest_ptr = 0;
    uint32_t flags = 0, unk_shift = 0, pool_dwords = 0;
    int status = 0;                    /* device parse status bits (GP_ST_*) */
    bool redo = false;                 /* GPU-parsed picture whose overflow run outlasted the device parser's cap: parsed again on the host,
                                          its blob lies at rp_dev + redo_off, its header in redo_hd */
    size_t redo_off = 0;
    HvqPicHeader redo_hd{};
    bool dropped = false;              /* rejected at flush time (or follows a rejected picture of its stream): not reconstructed */
    int old_slot = -1;                 /* P pictures: slot that holds what the reference's `present` buffer held before this picture
                                          (-1: never written -> the zero slot; -2: that picture's slot has been reused since) */
    const void *host_old = nullptr;    /* SDK path: the caller's `present` buffer itself */
};

struct Launch {
    int level;
    int queue;                         /* launch queue (0: the main stream, 1: stream2) */
    uint32_t first_tile, ntiles;       /* its picture slots in the launch table: first entry, count */
    uint32_t max_tiles, workgroups;    /* grid = (8, max_tiles, slots / 8) at the chosen tiles per workgroup; workgroups that do work */
    uint32_t max_wg[2], wgs[2];        /* the same for one / two tiles per workgroup (chosen at flush_end, when the queues are known) */
    uint32_t tpw;
    uint32_t items_cap;                /* LDS sizing of the launch: max over its pictures */
    uint32_t pair_cap, pool_cap;       /* its dynamic LDS: pair list entries, staged pool dwords */
};

/* a P picture with future-referencing macroblocks, behind the launch of its level: previous content into the destination slot
 * (unless it is there already), then the raster-order walk (hvq_selfref_kernel) from the side buffer */
struct SelfRef {
    int level;
    int queue;                         /* the launch queue of its picture's level launch */
    uint32_t job;                      /* launch slot */
    const uint8_t *old_dev;            /* device source of the previous content, nullptr: in place already */
    const void *old_host;              /* SDK path: host source */
    uint8_t *dst;
    size_t side_off;                   /* side buffer inside selfref_dev */
    uint32_t pic_bytes;
};

struct HvqContext {
    int device = 0;
    hipStream_t stream = nullptr;      /* every launch of a batch: its dependency levels in order */
    hipStream_t qstream[4] = { nullptr, nullptr, nullptr, nullptr };   /* launch queues 1..3 (created on first use; queue 0 is `stream`), see build_tiles */
    hipEvent_t ev_fork = nullptr, ev_join[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<Stream> streams;
    /* staging: pinned host arena mirrored by a device arena */
    uint8_t *host_arena = nullptr, *dev_arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    size_t arena_uploaded = 0;         /* [0, arena_uploaded) is already on its way to dev_arena (early H2D of bitstreams) */
    /* Two arenas: host_arena/dev_arena is the one being FILLED; the other belongs to the batch in flight (between
     * hvq_flush_begin and hvq_flush_end) or is idle.  Uploads run on their own stream so that the next batch's
     * bitstreams travel while this batch is parsed and reconstructed. */
    uint8_t *host_arena_alt = nullptr, *dev_arena_alt = nullptr;
    size_t arena_cap_alt = 0;
    int arena_id = 0;                  /* which of the two the current one is */
    bool arena_waited = false;         /* copy_stream already waits for the last batch that used the current arena */
    bool resv_active = false;          /* hvq_arena_reserve: [resv_base, resv_base + resv_bytes) of the arena being filled is the caller's to write */
    size_t resv_base = 0, resv_bytes = 0;
    hipStream_t copy_stream = nullptr, read_stream = nullptr;
    hipEvent_t ev_read = nullptr;
    hipEvent_t ev_copy = nullptr, ev_arena_free[2] = { nullptr, nullptr };
    std::vector<Pending> pending;
    /* batch in flight */
    bool fl_active = false;
    std::vector<Pending> fl_pending;
    std::vector<uint64_t> fl_nest_pairs;
    std::vector<int> fl_nest_streams;  /* stream of each pair */
    uint8_t *fl_host = nullptr, *fl_dev = nullptr;
    int fl_arena_id = 0;
    /* pinned staging of the table uploads, one set per arena id: an H2D from pageable memory would block the caller
     * until the stream has drained (the parse kernel!), which is exactly what hvq_flush_begin must not do */
    struct Pinned { uint8_t *p = nullptr; size_t cap = 0; } pin[2][5];   /* [arena id][parse jobs, tiles, jobs, nest pairs, re-parsed blobs] */
    std::vector<HvqJob> jobs_host;     /* the tables are built here, then copied into the pinned staging */
    std::vector<HvqTileRef> tiles_host;
    /* last flushed batch (kept resident for hvq_replay) */
    HvqJob *jobs_dev = nullptr;        /* one job per launch slot, in launch order (padding slots: total_tiles = 0) */
    size_t jobs_cap = 0;
    uint8_t *tq_dev = nullptr;         /* self-referencing P pictures: the pool offset of every block (written by the level's launch, read by hvq_selfref_kernel) */
    size_t tq_cap = 0;
    uint8_t *rb_dev = nullptr;         /* bulk readback: pictures gathered into one buffer, then few large copies */
    size_t rb_cap = 0;
    uint64_t *rb_tab_dev = nullptr, *rb_tab_host = nullptr;   /* their slot addresses (device table, pinned staging) */
    size_t rb_tab_cap = 0;
    uint8_t *selfref_dev = nullptr;    /* side buffers of the batch's self-referencing P pictures */
    size_t selfref_cap = 0;
    std::vector<SelfRef> selfrefs;
    std::vector<Launch> launches;
    std::vector<Launch> fl_launches;   /* of the batch in flight: tile ranges known at begin, LDS sizes at end */
    int max_queues = 2;                /* hvq_context_set_launch_queues */
    std::vector<uint8_t> fl_qof;       /* launch queue of every stream in the batch in flight (two queues: dealt by work, build_tiles) */
    HvqStats stats{};
    double parse_seconds = 0;
    std::atomic<uint64_t> copy_bytes{ 0 };     /* bitstream bytes copied into the pinned arena (copy threads), and their wall time in ns */
    std::atomic<uint64_t> copy_ns{ 0 };
    uint8_t *rgb_dev = nullptr;        /* scratch of the display epilogue */
    size_t rgb_cap = 0;
    RgbJob *rgb_jobs_dev = nullptr;
    size_t rgb_jobs_cap = 0;
    /* GPU entropy parse: blobs + scratch + nests of a batch, its job and result tables, the events around its parse kernel.  Two
     * sets: hvq_flush_next queues the parse of batch k + 1 BEFORE it takes the results of batch k, so the reconstruction of batch k
     * reads one set while the parse of batch k + 1 fills the other.  ps_live is the set the code below means by PS(c). */
    struct ParseSet {
        std::vector<size_t> fl_idx;        /* the batch's GPU-parsed pictures (indices into fl_pending) */
        std::vector<HvqParseJob> pjobs_host;
        HvqParseResult *pr_host = nullptr; /* pinned: the parse workgroups write their result records here */
        size_t pr_host_cap = 0;
        uint8_t *gp_dev = nullptr;
        size_t gp_cap = 0;
        HvqParseJob *pj_dev = nullptr;
        uint64_t *np_dev = nullptr;
        size_t pj_cap = 0;
        hipEvent_t ev_parse = nullptr, evp0 = nullptr, evp1 = nullptr;   /* results complete; timing of the parse kernel */
        uint32_t fl_rowbuf = 0;
        uint64_t *timing_dev = nullptr;
    } ps[2];
    int ps_live = 0;
    bool abandoned = false;                    /* flush_abandon ran since this was cleared */
    bool arena_idle[2] = { false, false };     /* nothing on the GPU reads this arena or its staging any more (hvq_flush_next) */
    double gpu_parse_ms = 0;           /* device time of the parse kernel of the last flush */
    uint32_t gpu_parse_retried = 0;    /* pictures of the last flush the flat parse path handed to the chains */
    uint32_t *redo_dev = nullptr;      /* their indices, for the chains kernel */
    uint8_t *rp_dev = nullptr;         /* blobs of the pictures parsed again on the host (overflow runs beyond the device parser's cap) */
    size_t rp_cap = 0;
    std::vector<uint8_t> rp_host;
    size_t redo_cap = 0;
    /* streaming: the bitstream copy of the batch being queued runs on a worker while the batch in flight is finished */
    std::thread copy_worker;
    bool copy_active = false;
    std::atomic<bool> copy_done{ false };      /* the worker has finished (hvq_flush_next polls it beside the parse event) */
    int copy_rc = 0;
    std::string copy_err;
};

static inline HvqContext::ParseSet &PS(HvqContext *c) { return c->ps[c->ps_live]; }
static inline const HvqContext::ParseSet &PS(const HvqContext *c) { return c->ps[c->ps_live]; }

static int arena_reserve(HvqContext *c, size_t need)
{
    if (c->arena_used + need <= c->arena_cap) return HVQ_OK;
    if (c->resv_active) return fail(HVQ_E_STATE, "the arena would have to grow while a reservation of it is outstanding (hvq_submit_many_arena first)");
    size_t ncap = c->arena_cap ? c->arena_cap : (size_t)64 << 20;
    while (ncap < c->arena_used + need) ncap *= 2;
    uint8_t *nh = nullptr, *nd = nullptr;
    HIPCHK(hipHostMalloc((void **)&nh, ncap, hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&nd, ncap));
    if (c->arena_used) memcpy(nh, c->host_arena, c->arena_used);
    if (c->host_arena) {
        HIPCHK(hipStreamSynchronize(c->copy_stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipHostFree(c->host_arena));
        HIPCHK(hipFree(c->dev_arena));
    }
    c->host_arena = nh; c->dev_arena = nd; c->arena_cap = ncap;
    c->arena_uploaded = 0;             /* the new device arena holds nothing yet */
    return HVQ_OK;
}

/* queue the H2D of [arena_uploaded, upto) of the arena being filled on the copy stream */
static int arena_upload(HvqContext *c, size_t upto)
{
    if (upto <= c->arena_uploaded) return HVQ_OK;
    if (!c->arena_waited) {            /* the batch that used this arena two flushes ago may still be reconstructing */
        if (!c->arena_idle[c->arena_id]) HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_arena_free[c->arena_id], 0));
        c->arena_waited = true;
    }
    HIPCHK(hipMemcpyAsync(c->dev_arena + c->arena_uploaded, c->host_arena + c->arena_uploaded, upto - c->arena_uploaded,
                          hipMemcpyHostToDevice, c->copy_stream));
    c->arena_uploaded = upto;
    return HVQ_OK;
}

static int flush_end(HvqContext *c);
static int flush_abandon(HvqContext *c, int rc);

/* wait for the bitstream copy of the batch being queued (hvq_submit_many_device in streaming).  A failed copy (a HIP error in one of
 * its uploads) takes the queued batch with it: its pictures were never uploaded. */
static int copy_join(HvqContext *c)
{
    if (!c->copy_active) return HVQ_OK;
    if (c->copy_worker.joinable()) c->copy_worker.join();
    c->copy_active = false;
    if (!c->copy_rc) return HVQ_OK;
    for (auto &p : c->pending) {
        Stream &s = c->streams[(size_t)p.stream];
        if ((size_t)p.ordinal < s.pic_slot.size() && s.pic_slot[(size_t)p.ordinal] == p.dst) {
            s.pic_slot[(size_t)p.ordinal] = -1;
            if (p.dst >= 0 && s.slots[(size_t)p.dst].pic == p.ordinal) s.slots[(size_t)p.dst].pic = -1;
        }
        s.anchor_old = s.anchor_new = -1; s.need_I = true; s.nest_src = -1;
    }
    c->pending.clear();
    c->arena_used = 0; c->arena_uploaded = 0;
    return fail(c->copy_rc, "bitstream upload of the queued batch failed (%s); the batch was dropped", c->copy_err.c_str());
}

/* HVQM4_AMD_FLUSH_TIMING=1: host-side timeline of the flush halves on stderr (development aid) */
static bool flush_timing() { static const bool on = getenv("HVQM4_AMD_FLUSH_TIMING") != nullptr; return on; }
static double now_ms()
{
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

/* A bitstream into the pinned arena with non-temporal stores: the arena is written once and read by the DMA engine, so fetching
 * its lines into the cache first (what a plain store does) doubles the memory traffic of a copy whose 160 MB per batch have to
 * fit into the 4.7 ms the GPU takes for the batch before (HVQM4_AMD_NT_COPY=0: memcpy). */
static inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

static inline void copy_to_arena(uint8_t *dst, const uint8_t *src, size_t len)
{
    static const bool nt = !(getenv("HVQM4_AMD_NT_COPY") && atoi(getenv("HVQM4_AMD_NT_COPY")) == 0);
    size_t i = 0;
#if defined(__x86_64__)
    if (nt && ((uintptr_t)dst & 15u) == 0 && len >= 4096) {
        for (; i + 64 <= len; i += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16)),
                          c = _mm_loadu_si128((const __m128i *)(src + i + 32)), d = _mm_loadu_si128((const __m128i *)(src + i + 48));
            _mm_stream_si128((__m128i *)(dst + i), a); _mm_stream_si128((__m128i *)(dst + i + 16), b);
            _mm_stream_si128((__m128i *)(dst + i + 32), c); _mm_stream_si128((__m128i *)(dst + i + 48), d);
        }
    }
#else
    (void)nt;
#endif
    memcpy(dst + i, src + i, len - i);
}

/* copy `bytes` into pinned staging buffer [id][which] and queue its upload to `dst` on the compute stream */
static int staged_upload(HvqContext *c, int id, int which, void *dst, const void *src, size_t bytes)
{
    HvqContext::Pinned &b = c->pin[id][which];
    if (bytes > b.cap) {
        if (b.p) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
        const size_t ncap = align_up(bytes * 2, 4096);
        HIPCHK(hipHostMalloc((void **)&b.p, ncap, hipHostMallocDefault));
        b.cap = ncap;
    }
    memcpy(b.p, src, bytes);
    /* Tables of a few hundred KB go up by a kernel of the compute queue that reads the pinned buffer over PCIe.  A copy command goes
     * to a DMA engine, and the runtime deals engines to streams as it likes: the job table of batch k, queued at flush_end, ended up
     * behind the 160 MB of batch k + 1's bitstreams on the copy stream's engine once in four or five batches and held the
     * reconstruction -- and the parse queued behind it -- back by 1-2 ms (HVQM4_AMD_KERNEL_UPLOAD=0: copy commands).  Buffers are
     * 16-byte multiples (pinned capacity is rounded to 4 KB, device tables are allocated with slack). */
    static const bool by_kernel = !(getenv("HVQM4_AMD_KERNEL_UPLOAD") && atoi(getenv("HVQM4_AMD_KERNEL_UPLOAD")) == 0);
    if (by_kernel && bytes <= ((size_t)8 << 20) && ((uintptr_t)dst & 15u) == 0) {
        const size_t pad = align_up(bytes, 16);
        if (pad > bytes) memset(b.p + bytes, 0, pad - bytes);
        HIPCHK(hvq_launch_upload(b.p, dst, bytes, c->stream));
    } else HIPCHK(hipMemcpyAsync(dst, b.p, bytes, hipMemcpyHostToDevice, c->stream));
    return HVQ_OK;
}

HVQ_EXPORT int hvq_context_create(int device, HvqContext **out)
{
    if (!out) return fail(HVQ_E_ARG, "null out");
#if defined(__x86_64__)
    /* the host entropy parse (hvq_parse.c) is built for x86-64-v3 (csrc/Makefile) */
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("bmi2"))
        return fail(HVQ_E_ARG, "this build's host entropy parse needs an x86-64-v3 CPU (AVX2, BMI2); rebuild with HOST_ARCH= for older hosts");
#endif
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(HVQ_E_NOGPU, "no HIP device available (%s): the HVQM4 reconstruction path is GPU-only",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(HVQ_E_ARG, "device %d out of range (%d devices)", device, n);
    HIPCHK(hipSetDevice(device));
    HvqContext *c = new HvqContext();
    c->device = device;
    struct Guard { HvqContext *c; ~Guard() { if (c) hvq_context_destroy(c); } } guard{ c };     /* a failing step below frees what exists */
    {   /* The launch stream at the highest stream priority: the runtime deals a process's streams to four hardware queues in the order of
         * their creation, per priority level -- with a copy and a read-back stream per context, the launch streams of two contexts (two
         * players of one process, bench.py's two-context streaming leg) landed on ONE hardware queue and their kernels ran one after the
         * other (4.6 ms per half batch each instead of 3.6: profiles/r05_flush_next.txt 6).  At a level of their own the launch streams of
         * up to four contexts get a queue each.  HVQM4_AMD_STREAM_PRIORITY=0: plain streams as before. */
        int lo = 0, hi = 0;
        static const bool prio = !(getenv("HVQM4_AMD_STREAM_PRIORITY") && atoi(getenv("HVQM4_AMD_STREAM_PRIORITY")) == 0);
        if (prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
            HIPCHK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
        else HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->read_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_read, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    for (auto &S : c->ps) {
        HIPCHK(hipEventCreateWithFlags(&S.ev_parse, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&S.evp0));
        HIPCHK(hipEventCreate(&S.evp1));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_arena_free[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_arena_free[1], hipEventDisableTiming));
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    guard.c = nullptr;
    *out = c;
    return HVQ_OK;
}

HVQ_EXPORT int hvq_context_set_launch_queues(HvqContext *c, int n)
{
    if (!c || n < 1 || n > 2) return fail(HVQ_E_ARG, "launch queues: 1 or 2");
    c->max_queues = n;
    return HVQ_OK;
}

HVQ_EXPORT void hvq_context_destroy(HvqContext *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)copy_join(c);
    (void)flush_end(c);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->streams) {
        if (s.parser) hvq_parser_destroy(s.parser);
        if (s.dev) (void)hipFree(s.dev);
        if (s.nest_keep) (void)hipFree(s.nest_keep);
    }
    for (auto &S : c->ps) {
        if (S.gp_dev) (void)hipFree(S.gp_dev);
        if (S.pj_dev) (void)hipFree(S.pj_dev);
        if (S.np_dev) (void)hipFree(S.np_dev);
        if (S.pr_host) (void)hipHostFree(S.pr_host);
        if (S.timing_dev) (void)hipFree(S.timing_dev);
        if (S.ev_parse) (void)hipEventDestroy(S.ev_parse);
        if (S.evp0) (void)hipEventDestroy(S.evp0);
        if (S.evp1) (void)hipEventDestroy(S.evp1);
    }
    if (c->host_arena) (void)hipHostFree(c->host_arena);
    if (c->redo_dev) (void)hipFree(c->redo_dev);
    if (c->rp_dev) (void)hipFree(c->rp_dev);
    if (c->dev_arena) (void)hipFree(c->dev_arena);
    if (c->host_arena_alt) (void)hipHostFree(c->host_arena_alt);
    if (c->dev_arena_alt) (void)hipFree(c->dev_arena_alt);
    for (auto &set : c->pin) for (auto &b : set) if (b.p) (void)hipHostFree(b.p);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    for (auto e : c->ev_arena_free) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->read_stream) { (void)hipStreamSynchronize(c->read_stream); (void)hipStreamDestroy(c->read_stream); }
    for (int q = 1; q < 4; ++q) {
        if (c->qstream[q]) { (void)hipStreamSynchronize(c->qstream[q]); (void)hipStreamDestroy(c->qstream[q]); }
        if (c->ev_join[q]) (void)hipEventDestroy(c->ev_join[q]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_read) (void)hipEventDestroy(c->ev_read);
    if (c->jobs_dev) (void)hipFree(c->jobs_dev);
    if (c->tq_dev) (void)hipFree(c->tq_dev);
    if (c->selfref_dev) (void)hipFree(c->selfref_dev);
    if (c->rb_dev) (void)hipFree(c->rb_dev);
    if (c->rb_tab_dev) (void)hipFree(c->rb_tab_dev);
    if (c->rb_tab_host) (void)hipHostFree(c->rb_tab_host);
    if (c->rgb_dev) (void)hipFree(c->rgb_dev);
    if (c->rgb_jobs_dev) (void)hipFree(c->rgb_jobs_dev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

/* bytes of a stream's picture ring ((nslots + 1) slots: the last one stays zero), 0 for a geometry the parser refuses.  Host only.
 * Every reference read of the kernels is ring base + 32-bit offset (HvqJob::ref0_off, block records, pair entries): the ring
 * must stay below 4 GiB, which hvq_stream_open enforces with this number. */
HVQ_EXPORT uint64_t hvq_stream_ring_bytes(int width, int height, int h_samp, int v_samp, int nslots)
{
    HvqParser *p = hvq_parser_create(width, height, h_samp, v_samp, 1);
    if (!p || nslots < 0) { if (p) hvq_parser_destroy(p); return 0; }
    const uint64_t slot = align_up((size_t)hvq_parser_pic_bytes(p) + 64, 256);
    hvq_parser_destroy(p);
    return ((uint64_t)nslots + 1u) * slot;
}

HVQ_EXPORT int hvq_stream_open(HvqContext *c, int width, int height, int h_samp, int v_samp, int is15, int nslots)
{
    if (!c) return fail(HVQ_E_ARG, "null context");
    if (nslots < 3) return fail(HVQ_E_ARG, "nslots must be >= 3 (past, present, future)");
    { int rcj = copy_join(c); if (rcj) return rcj; }
    HIPCHK(hipSetDevice(c->device));
    HvqParser *p = hvq_parser_create(width, height, h_samp, v_samp, is15);
    if (!p) return fail(HVQ_E_GEOMETRY, "unsupported geometry %dx%d sampling %dx%d (need multiples of 8, <= 8192; samplings 2x2, 1x1 and 2x1 -- the reference's own tables cannot decode 1x2)",
                        width, height, h_samp, v_samp);
    Stream s;
    s.open = true; s.parser = p; s.w = width; s.h = height;
    s.pic_bytes = hvq_parser_pic_bytes(p);
    s.slot_bytes = (uint32_t)align_up((size_t)s.pic_bytes + 64, 256);
    s.slots.resize((size_t)nslots);
    size_t bytes = (size_t)(nslots + 1) * s.slot_bytes;
    if (bytes >= ((size_t)1 << 32)) {
        hvq_parser_destroy(p);
        return fail(HVQ_E_OVERFLOW, "picture ring of %d + 1 slots x %u bytes exceeds 4 GiB (reference reads are ring base + 32-bit offset): open the stream with fewer slots",
                    nslots, s.slot_bytes);
    }
    hipError_t e = hipMalloc((void **)&s.dev, bytes);
    if (e != hipSuccess) { hvq_parser_destroy(p); return fail(HVQ_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    e = hipMemsetAsync(s.dev, 0, bytes, c->stream);
    if (e != hipSuccess) { hvq_parser_destroy(p); (void)hipFree(s.dev); return fail(HVQ_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e)); }
    c->streams.push_back(s);
    return (int)c->streams.size() - 1;
}

/* Host threads that share the parse of ONE picture of this stream (hvq_parser_set_threads: the picture's sections side by side; 1 = the
 * submitting thread alone, the default).  For callers that decode few streams (a lone clip: hvq_stream_submit parses its pictures one after
 * the other; hvq_submit_many parses streams side by side and leaves a stream's pictures to one thread).  Returns the count in effect. */
HVQ_EXPORT int hvq_stream_set_parse_threads(HvqContext *c, int sid, int threads)
{
    if (!c || sid < 0 || sid >= (int)c->streams.size() || !c->streams[sid].open) return fail(HVQ_E_ARG, "bad stream %d", sid);
    return hvq_parser_set_threads(c->streams[(size_t)sid].parser, threads);
}

HVQ_EXPORT int hvq_stream_close(HvqContext *c, int sid)
{
    if (!c || sid < 0 || sid >= (int)c->streams.size() || !c->streams[sid].open) return fail(HVQ_E_ARG, "bad stream %d", sid);
    for (auto &p : c->pending)
        if (p.stream == sid) return fail(HVQ_E_STATE, "stream %d has queued pictures; flush first", sid);
    { int rcj = copy_join(c); if (rcj) return rcj; }
    { int rc = flush_end(c); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    Stream &s = c->streams[sid];
    hvq_parser_destroy(s.parser); s.parser = nullptr;
    HIPCHK(hipFree(s.dev)); s.dev = nullptr;
    if (s.nest_keep) { HIPCHK(hipFree(s.nest_keep)); s.nest_keep = nullptr; }
    c->launches.clear();               /* the resident batch may reference the freed slots: no replay after a close */
    s.open = false;
    return HVQ_OK;
}

HVQ_EXPORT uint32_t hvq_stream_pic_bytes(HvqContext *c, int sid)
{
    if (!c || sid < 0 || sid >= (int)c->streams.size() || !c->streams[sid].open) return 0;
    return c->streams[sid].pic_bytes;
}

static int alloc_slot(Stream &s)
{
    int n = (int)s.slots.size();
    for (int i = 0; i < n; ++i) {
        int cand = (s.ring + i) % n;
        if (cand != s.anchor_old && cand != s.anchor_new) { s.ring = (cand + 1) % n; return cand; }
    }
    return -1;
}

/* queue one picture: slot assignment == picture rotation of h4m:2087-2093 / 2131-2137, then the dependency level */
static int enqueue_common(HvqContext *c, int sid, int frame_type, Pending q)
{
    Stream &s = c->streams[(size_t)sid];
    q.stream = sid; q.ordinal = s.npics;
    if (frame_type != HVQ_FRAME_B) { std::swap(s.anchor_old, s.anchor_new); std::swap(s.rp_past, s.rp_future); }   /* past <-> future */
    /* where what the reference's `present` buffer holds right now lives here (looked up before the ring hands out a slot) */
    const int old_ord = s.rp_present;
    q.old_slot = old_ord < 0 ? -1 : (s.pic_slot[(size_t)old_ord] >= 0 ? s.pic_slot[(size_t)old_ord] : -2);
    q.host_old = s.sdk_present;
    s.rp_present = q.ordinal;
    if (frame_type != HVQ_FRAME_B) std::swap(s.rp_present, s.rp_future);
    q.dst = alloc_slot(s);
    if (frame_type == HVQ_FRAME_I) { q.ref0 = -1; q.ref1 = -1; }
    else if (frame_type == HVQ_FRAME_P) { q.ref0 = s.anchor_old; q.ref1 = q.dst; }  /* future aliases present, h4m:2060 */
    else { q.ref0 = s.anchor_old; q.ref1 = s.anchor_new; }
    int lvl = 0;
    auto dep_w = [&](int slot) { if (slot >= 0 && slot != q.dst) lvl = std::max(lvl, s.slots[slot].w_level + 1); };
    dep_w(q.ref0); dep_w(q.ref1);
    lvl = std::max(lvl, std::max(s.slots[q.dst].w_level, s.slots[q.dst].r_level) + 1);
    if (frame_type == HVQ_FRAME_P && q.old_slot >= 0 && q.old_slot != q.dst) {
        /* should the picture turn out to reference itself (known after its parse), that slot is read behind this level's launch:
         * complete by then, and not handed to a later picture before */
        lvl = std::max(lvl, s.slots[(size_t)q.old_slot].w_level);
        s.slots[(size_t)q.old_slot].r_level = std::max(s.slots[(size_t)q.old_slot].r_level, lvl);
    }
    q.level = lvl;
    if (q.ref0 >= 0) s.slots[q.ref0].r_level = std::max(s.slots[q.ref0].r_level, lvl);
    if (q.ref1 >= 0 && q.ref1 != q.dst) s.slots[q.ref1].r_level = std::max(s.slots[q.ref1].r_level, lvl);
    if (s.slots[q.dst].pic >= 0) s.pic_slot[(size_t)s.slots[q.dst].pic] = -1;
    s.slots[q.dst].w_level = lvl; s.slots[q.dst].r_level = -1; s.slots[q.dst].pic = q.ordinal;
    s.pic_slot.push_back(q.dst);
    if (frame_type != HVQ_FRAME_B) s.anchor_new = q.dst;     /* present <-> future: newest anchor becomes "future" */
    c->pending.push_back(q);
    return s.npics++;
}

/* host-parsed picture: blob already in the host arena at `off` */
static int enqueue_picture(HvqContext *c, int sid, int frame_type, size_t off, size_t blen)
{
    Pending q{};
    q.blob_off = off; q.blob_len = blen;
    const HvqPicHeader *hd = (const HvqPicHeader *)(c->host_arena + off);
    q.ntiles = hd->tile_first[3]; q.nwg = picture_workgroups(hd->tile_first);
    q.kind = hd->pic_kind; q.w = hd->width; q.h = hd->height;
    q.max_items = hd->max_items; q.max_pairs = hd->max_pairs;
    return enqueue_common(c, sid, frame_type, q);
}

static int check_submit_args(HvqContext *c, int sid, int frame_type, const uint8_t *pic, size_t len, bool check_resume = true)
{
    if (!c || sid < 0 || sid >= (int)c->streams.size() || !c->streams[sid].open) return fail(HVQ_E_ARG, "bad stream %d", sid);
    if (!pic || len < 8 + 0x44 + 4) return fail(HVQ_E_ARG, "picture too short (%zu bytes)", len);
    if (frame_type != HVQ_FRAME_I && frame_type != HVQ_FRAME_P && frame_type != HVQ_FRAME_B)
        return fail(HVQ_E_ARG, "unknown frame type 0x%x", frame_type);
    if (check_resume && c->streams[sid].need_I && frame_type != HVQ_FRAME_I)
        return fail(HVQ_E_STATE, "stream %d: a picture was rejected, decoding resumes at the next I picture", sid);
    return HVQ_OK;
}

/* the same rule over a list of pictures: an I picture earlier in the list re-opens its stream for the ones after it */
static int check_resume_order(HvqContext *c, int n, const int *streams, const int *frame_types)
{
    std::vector<char> need(c->streams.size());
    for (size_t i = 0; i < need.size(); ++i) need[i] = c->streams[i].need_I;
    for (int i = 0; i < n; ++i) {
        if (frame_types[i] == HVQ_FRAME_I) need[(size_t)streams[i]] = 0;
        else if (need[(size_t)streams[i]])
            return fail(HVQ_E_STATE, "stream %d: a picture was rejected, decoding resumes at the next I picture", streams[i]);
    }
    return HVQ_OK;
}

HVQ_EXPORT int hvq_stream_submit(HvqContext *c, int sid, int frame_type, const uint8_t *pic, size_t len)
{
    int rc = check_submit_args(c, sid, frame_type, pic, len);
    if (rc) return rc;
    rc = copy_join(c);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    Stream &s = c->streams[sid];
    if (s.parse_mode == 2) return fail(HVQ_E_STATE, "stream %d is parsed on the GPU; use hvq_submit_many_device", sid);
    s.parse_mode = 1;
    size_t bound = align_up(hvq_parser_blob_bound(s.parser), 256);
    rc = arena_reserve(c, bound);
    if (rc) return rc;
    size_t off = c->arena_used, blen = 0;
    auto t0 = std::chrono::steady_clock::now();
    rc = hvq_parse_picture(s.parser, frame_type, pic, len, c->host_arena + off, bound, &blen);
    c->parse_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return fail(rc, "parse failed (%d) for stream %d picture %d", rc, sid, s.npics);
    /* the header's flags plus what a P/B picture's second pass raised behind it (an endless overflow run in a block's scalars, a
     * clamped vector target): refused, never decoded differently */
    if (const char *why = unsupported_reason(((const HvqPicHeader *)(c->host_arena + off))->flags | hvq_parser_last_flags(s.parser))) {
        s.need_I = true;
        return fail(HVQ_E_UNSUPPORTED, "stream %d picture %d: %s", sid, s.npics, why);
    }
    if (frame_type == HVQ_FRAME_I) s.need_I = false;
    c->arena_used = off + align_up(blen, 256);
    return enqueue_picture(c, sid, frame_type, off, blen);
}

HVQ_EXPORT int hvq_submit_many(HvqContext *c, int n, const int *streams, const int *frame_types,
                               const uint8_t *const *pics, const size_t *lens, int threads, int *ordinals)
{
    if (!c || n < 0 || !streams || !frame_types || !pics || !lens) return fail(HVQ_E_ARG, "bad arguments");
    for (int i = 0; i < n; ++i) {
        int rc = check_submit_args(c, streams[i], frame_types[i], pics[i], lens[i], false);
        if (rc) return rc;
    }
    if (n == 0) return HVQ_OK;
    { int rcj = copy_join(c); if (rcj) return rcj; }
    { int rc = check_resume_order(c, n, streams, frame_types); if (rc) return rc; }
    for (int i = 0; i < n; ++i) {
        Stream &s = c->streams[(size_t)streams[i]];
        if (s.parse_mode == 2) return fail(HVQ_E_STATE, "stream %d is parsed on the GPU; use hvq_submit_many_device", streams[i]);
        s.parse_mode = 1;
    }
    HIPCHK(hipSetDevice(c->device));
    threads = std::max(1, std::min(threads, 256));
    /* work units = streams (a parser is stateful); unit u gets its pictures in array order */
    std::vector<std::vector<int>> per_stream(c->streams.size());
    std::vector<int> units;
    for (int i = 0; i < n; ++i) {
        if (per_stream[(size_t)streams[i]].empty()) units.push_back(streams[i]);
        per_stream[(size_t)streams[i]].push_back(i);
    }
    /* workers parse into private growing buffers (no per-picture allocation), then copy their blobs into the
     * pinned arena in parallel once the layout is known */
    struct Piece { int worker; size_t off, len; uint32_t late; };     /* late: flags raised behind the blob header (hvq_parser_last_flags) */
    std::vector<Piece> pieces((size_t)n, Piece{ -1, 0, 0, 0 });
    std::vector<int> rcs((size_t)n, 0);
    struct RawBuf {                       /* growing byte buffer without value-initialisation */
        uint8_t *p = nullptr; size_t size = 0, cap = 0;
        ~RawBuf() { free(p); }
        bool ensure(size_t n) { if (n <= cap) return true; size_t nc = std::max(cap * 2, n); uint8_t *q = (uint8_t *)realloc(p, nc); if (!q) return false; p = q; cap = nc; return true; }
    };
    std::vector<RawBuf> wbuf((size_t)threads);
    std::atomic<size_t> next{ 0 };
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int wid) {
        RawBuf &buf = wbuf[(size_t)wid];
        for (;;) {
            size_t u = next.fetch_add(1);
            if (u >= units.size()) break;
            Stream &s = c->streams[(size_t)units[u]];
            const size_t bound = hvq_parser_blob_bound(s.parser);
            for (int i : per_stream[(size_t)units[u]]) {
                if (!buf.ensure(buf.size + bound + 32)) { rcs[(size_t)i] = HVQ_E_OVERFLOW; break; }
                const size_t at = (size_t)((((uintptr_t)buf.p + buf.size + 15) & ~(uintptr_t)15) - (uintptr_t)buf.p);
                size_t blen = 0;
                rcs[(size_t)i] = hvq_parse_picture(s.parser, frame_types[i], pics[i], lens[i], buf.p + at, bound, &blen);
                if (rcs[(size_t)i]) break;                 /* later pictures of the stream would see a wrong parser state */
                pieces[(size_t)i] = Piece{ wid, at, blen, hvq_parser_last_flags(s.parser) };
                buf.size = at + blen;
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(worker, t);
        worker(0);
        for (auto &t : pool) t.join();
    }
    for (int i = 0; i < n; ++i)
        if (rcs[(size_t)i] || pieces[(size_t)i].worker < 0)
            return fail(rcs[(size_t)i] ? rcs[(size_t)i] : HVQ_E_STATE, "parse failed for picture %d (stream %d)", i, streams[i]);
    for (int i = 0; i < n; ++i) {
        const Piece &pc = pieces[(size_t)i];
        if (const char *why = unsupported_reason(((const HvqPicHeader *)(wbuf[(size_t)pc.worker].p + pc.off))->flags | pc.late)) {
            c->streams[(size_t)streams[i]].need_I = true;
            return fail(HVQ_E_UNSUPPORTED, "picture %d (stream %d): %s; nothing of this call was queued", i, streams[i], why);
        }
    }
    for (int i = 0; i < n; ++i) if (frame_types[i] == HVQ_FRAME_I) c->streams[(size_t)streams[i]].need_I = false;
    size_t need = 0;
    std::vector<size_t> offs((size_t)n);
    for (int i = 0; i < n; ++i) { offs[(size_t)i] = need; need += align_up(pieces[(size_t)i].len, 256); }
    int rc = arena_reserve(c, need);
    if (rc) return rc;
    const size_t base = c->arena_used;
    {
        std::atomic<int> nexti{ 0 };
        auto copier = [&]() {
            for (;;) {
                int i = nexti.fetch_add(1);
                if (i >= n) break;
                const Piece &pc = pieces[(size_t)i];
                memcpy(c->host_arena + base + offs[(size_t)i], wbuf[(size_t)pc.worker].p + pc.off, pc.len);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(copier);
        copier();
        for (auto &t : pool) t.join();
    }
    c->parse_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n; ++i) {
        int ord = enqueue_picture(c, streams[i], frame_types[i], base + offs[(size_t)i], pieces[(size_t)i].len);
        if (ordinals) ordinals[i] = ord;
    }
    c->arena_used = base + need;
    return HVQ_OK;
}

/* GPU entropy parse: queue the raw bitstreams; hvq_flush parses them on the device (hvq_gparse.hip).  Three ways for the bytes to
 * reach the pinned arena:
 *   SUBMIT_COPY        copied before the call returns (hvq_submit_many_device: the caller's buffers are free again on return)
 *   SUBMIT_COPY_ASYNC  copied by a worker thread (hvq_submit_many_device_async: the buffers stay the caller's until the next
 *                      hvq_flush_begin / hvq_sync joins the worker)
 *   SUBMIT_ARENA       already there: the caller filled a reservation of the arena itself (hvq_arena_reserve) -- no copy at all */
enum SubmitMode { SUBMIT_COPY, SUBMIT_COPY_ASYNC, SUBMIT_ARENA };
static int submit_device(HvqContext *c, int n, const int *streams, const int *frame_types, const uint8_t *const *pics_in,
                         const size_t *arena_offs, const size_t *lens, int *ordinals, SubmitMode mode)
{
    if (!c || n < 0 || !streams || !frame_types || !lens || (mode == SUBMIT_ARENA ? !arena_offs : !pics_in)) return fail(HVQ_E_ARG, "bad arguments");
    size_t need = 0;
    { int rcj = copy_join(c); if (rcj) return rcj; }
    std::vector<const uint8_t *> arena_pics;
    if (mode == SUBMIT_ARENA) {
        if (!c->resv_active) return fail(HVQ_E_STATE, "no arena reservation outstanding (hvq_arena_reserve)");
        arena_pics.resize((size_t)n);
        size_t prev_end = 0;
        for (int i = 0; i < n; ++i) {
            if (lens[i] > c->resv_bytes)               /* before the span is computed: lens[i] + 32 must not wrap */
                return fail(HVQ_E_ARG, "picture %d: %zu bytes exceed the reservation (%zu)", i, lens[i], c->resv_bytes);
            const size_t span = align_up(lens[i] + 32, 256);
            if ((arena_offs[i] & 255u) || arena_offs[i] < prev_end || arena_offs[i] > c->resv_bytes || span > c->resv_bytes - arena_offs[i])
                return fail(HVQ_E_ARG, "picture %d: offset %zu (+ %zu bytes with its padding) is not a 256-byte aligned, ascending range of the reservation", i, arena_offs[i], span);
            prev_end = arena_offs[i] + span;
            arena_pics[(size_t)i] = c->host_arena + c->resv_base + arena_offs[i];
        }
    }
    const uint8_t *const *pics = mode == SUBMIT_ARENA ? arena_pics.data() : pics_in;
    for (int i = 0; i < n; ++i) {
        int rc = check_submit_args(c, streams[i], frame_types[i], pics[i], lens[i], false);
        if (rc) return rc;
        if (lens[i] > 0x7FFFFFF0u) return fail(HVQ_E_ARG, "picture %d: the GPU parser needs the real picture length", i);
        if (c->streams[(size_t)streams[i]].parse_mode == 1)
            return fail(HVQ_E_STATE, "stream %d is parsed on the host; a stream keeps one parser for its lifetime", streams[i]);
        need += align_up(lens[i] + 32, 256);
    }
    if (n == 0) { if (mode == SUBMIT_ARENA) c->resv_active = false; return HVQ_OK; }
    { int rcr = check_resume_order(c, n, streams, frame_types); if (rcr) return rcr; }
    HIPCHK(hipSetDevice(c->device));
    if (mode != SUBMIT_ARENA) {
        int rc = arena_reserve(c, need);
        if (rc) return rc;
    }
    std::vector<size_t> offs((size_t)n);
    /* Everything below changes stream and queue state before the last thing that can fail (allocation, upload): a failed
     * call must leave the context as it found it, so the touched streams are saved and put back. */
    std::vector<std::pair<int, Stream>> saved;
    const size_t pend0 = c->pending.size(), used0 = c->arena_used;
    auto rollback = [&]() {
        for (auto &kv : saved) {
            Stream &s = c->streams[(size_t)kv.first];
            if (s.nest_keep && !kv.second.nest_keep) (void)hipFree(s.nest_keep);
            s = kv.second;
        }
        c->pending.resize(pend0);
        c->arena_used = used0;
        if (c->arena_uploaded > used0) c->arena_uploaded = used0;
    };
#define HIPCHK_RB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rollback(); return fail(HVQ_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
    for (int i = 0; i < n; ++i) {
        Stream &s = c->streams[(size_t)streams[i]];
        {
            bool have = false;
            for (auto &kv : saved) have |= kv.first == streams[i];
            if (!have) saved.emplace_back(streams[i], s);
        }
        if (s.parse_mode == 0) {
            s.parse_mode = 2;
            hvq_parser_layout(s.parser, &s.layout);
            s.blob_cap = (uint32_t)align_up(hvq_parser_blob_bound(s.parser), 256);
            uint32_t blocks = 0;
            for (int k = 0; k < 3; ++k) blocks += (uint32_t)s.layout.hb[k] * s.layout.vb[k];
            s.scratch_bytes = (uint32_t)align_up(hvq_gparse_scratch_bytes(blocks, s.layout.tile_first[3] * (HVQ_TILE_BLOCKS / 64),
                                                                          s.layout.mcb_w * s.layout.mcb_h), 256);
            HIPCHK_RB(hipMalloc((void **)&s.nest_keep, 2 * GP_ALIGN16(HVQ_NESTP_BYTES)));
            HIPCHK_RB(hipMemsetAsync(s.nest_keep, 0, 2 * GP_ALIGN16(HVQ_NESTP_BYTES), c->stream));
        }
        if (mode == SUBMIT_ARENA) offs[(size_t)i] = c->resv_base + arena_offs[i];      /* arena_used already covers the reservation */
        else { offs[(size_t)i] = c->arena_used; c->arena_used += align_up(lens[i] + 32, 256); }
        Pending q{};
        q.dev = true;
        q.blob_off = offs[(size_t)i]; q.blob_len = lens[i];
        q.ntiles = s.layout.tile_first[3]; q.nwg = picture_workgroups(s.layout.tile_first);
        q.kind = frame_types[i] == HVQ_FRAME_I ? HVQ_PIC_I : (frame_types[i] == HVQ_FRAME_P ? HVQ_PIC_P : HVQ_PIC_B);
        q.w = s.layout.width; q.h = s.layout.height;
        q.unk_shift = pics[i][1];
        q.nest_ref = s.nest_src;
        const int ord = enqueue_common(c, streams[i], frame_types[i], q);
        if (frame_types[i] == HVQ_FRAME_I) {
            s.need_I = false;
            s.nest_src = (int)c->pending.size() - 1;
            c->pending.back().nest_ref = s.nest_src;
        }
        if (ordinals) ordinals[i] = ord;
    }
    /* bitstreams -> pinned arena -> HBM, pipelined: a few threads copy a chunk of pictures (zero padded: the device
     * reader sees zeros past the end), its H2D is queued at once and runs while the next chunk is being copied.
     * While a batch is in flight (streaming: hvq_flush_begin ... hvq_flush_end) the copy runs on a worker thread and this call
     * returns at once, so that the caller reaches hvq_flush_end -- and the in-flight batch its reconstruction launches -- as soon as
     * the parse results arrive, however long the host takes over 160 MB of memcpy (4-7.5 ms on a shared host, measured); the next
     * hvq_flush_begin joins the worker.  `pics[i]` must stay readable until then. */
    struct CopyJob { std::vector<const uint8_t *> pics; std::vector<size_t> lens, offs; size_t end_used; bool early; uint8_t *host; };
    CopyJob job;
    job.pics.assign(pics, pics + n); job.lens.assign(lens, lens + n); job.offs = offs;
    job.end_used = c->arena_used; job.early = c->arena_uploaded == offs[0];          /* nothing older is waiting for the flush-time upload */
    job.host = c->host_arena;
    auto run_copy = [c](const CopyJob &j) -> int {
        const int n = (int)j.pics.size();
        const auto tc0 = std::chrono::steady_clock::now();
        /* chunks of ~16 MB: a chunk's H2D is queued as soon as its last picture is in the arena.  The copy threads are started ONCE
         * per batch and take pictures off a shared counter (round 3 started and joined a set of threads per chunk: 70 thread
         * starts per 160 MB batch); the calling thread copies too and queues the uploads in chunk order. */
        const size_t chunk_bytes = (size_t)16 << 20;
        std::vector<int> chunk_end;                         /* picture index behind each chunk */
        {
            size_t bytes = 0;
            for (int i = 0; i < n; ++i) {
                bytes += align_up(j.lens[(size_t)i] + 32, 256);
                if (bytes >= chunk_bytes || i == n - 1) { chunk_end.push_back(i + 1); bytes = 0; }
            }
        }
        std::vector<std::atomic<int>> left(chunk_end.size());
        for (size_t k = 0; k < chunk_end.size(); ++k) left[k].store(chunk_end[k] - (k ? chunk_end[k - 1] : 0));
        std::vector<int> chunk_of((size_t)n);
        for (size_t k = 0, i = 0; k < chunk_end.size(); ++k) for (; (int)i < chunk_end[k]; ++i) chunk_of[i] = (int)k;
        std::atomic<int> next{ 0 };
        auto copy_some = [&](int budget) {               /* up to `budget` pictures; returns false when none was left */
            bool any = false;
            while (budget-- > 0) {
                const int i = next.fetch_add(1);
                if (i >= n) return any;
                uint8_t *dst = j.host + j.offs[(size_t)i];
                const size_t span = align_up(j.lens[(size_t)i] + 32, 256);
                copy_to_arena(dst, j.pics[(size_t)i], j.lens[(size_t)i]);
                memset(dst + j.lens[(size_t)i], 0, span - j.lens[(size_t)i]);
#if defined(__x86_64__)
                _mm_sfence();                            /* the streamed lines are in memory before the chunk's upload is queued */
#else
                std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
                left[(size_t)chunk_of[(size_t)i]].fetch_sub(1, std::memory_order_release);
                any = true;
            }
            return true;
        };
        size_t total = 0;
        for (int i = 0; i < n; ++i) total += j.lens[(size_t)i];
        /* 8 threads when the host has them: with 4 the copy of a dense batch (160 MB) takes about as long as the GPU leaves for it */
        static const int copy_threads = getenv("HVQM4_AMD_COPY_THREADS") ? std::max(1, atoi(getenv("HVQM4_AMD_COPY_THREADS")))
                                                                         : (std::thread::hardware_concurrency() >= 16 ? 8 : 4);
        const int nt = total >= ((size_t)4 << 20) ? copy_threads : 1;
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back([&]() { while (copy_some(1 << 30)) {} });
        int rc = HVQ_OK;
        for (size_t k = 0; k < chunk_end.size(); ++k) {
            while (left[k].load(std::memory_order_acquire) > 0)
                if (!copy_some(4)) std::this_thread::yield();      /* help; when nothing is left to take, wait for the others */
            if (j.early && !rc) rc = arena_upload(c, k + 1 < chunk_end.size() ? j.offs[(size_t)chunk_end[k]] : j.end_used);
        }
        for (auto &t : pool) t.join();
        c->copy_bytes += total;
        c->copy_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tc0).count();
        /* test hook (tests/test_gpu_batch.py): the upload of this batch "fails" -- exercises the rollback of both copy modes */
        if (!rc && getenv("HVQM4_AMD_TEST_FAIL_COPY")) rc = fail(HVQ_E_HIP, "bitstream upload failed (HVQM4_AMD_TEST_FAIL_COPY)");
        return rc;
    };
    if (mode == SUBMIT_ARENA) {
        /* the bytes are in place: zero the padding behind every picture (the device reader sees zeros past the end) and queue the
         * upload of the whole reservation at once */
        for (int i = 0; i < n; ++i) {
            uint8_t *dst = c->host_arena + offs[(size_t)i];
            memset(dst + lens[i], 0, align_up(lens[i] + 32, 256) - lens[i]);
        }
        c->resv_active = false;
        if (c->arena_uploaded == c->resv_base) {                    /* nothing older is waiting for the flush-time upload */
            int rcu = arena_upload(c, c->resv_base + c->resv_bytes);
            if (rcu) { rollback(); return rcu; }
        }
        return HVQ_OK;
    }
    /* only the _async entry point defers its copy (rounds 4-5 had an environment switch that made the plain call defer too: a caller
     * of the plain call may free its buffers on return, so the switch was a use-after-free waiting to happen; removed in round 6) */
    if (mode == SUBMIT_COPY_ASYNC) {
        c->copy_rc = HVQ_OK;
        c->copy_err.clear();
        c->copy_active = true;
        c->copy_done.store(false, std::memory_order_relaxed);
        const int dev = c->device;
        c->copy_worker = std::thread([c, dev, run_copy, job = std::move(job)]() {
            (void)hipSetDevice(dev);
            c->copy_rc = run_copy(job);
            if (c->copy_rc) c->copy_err = g_err;          /* the worker's thread-local error text */
            c->copy_done.store(true, std::memory_order_release);
        });
    } else {
        int rcc = run_copy(job);
        if (rcc) { rollback(); return rcc; }
    }
#undef HIPCHK_RB
    return HVQ_OK;
}

HVQ_EXPORT int hvq_submit_many_device(HvqContext *c, int n, const int *streams, const int *frame_types,
                                      const uint8_t *const *pics, const size_t *lens, int *ordinals)
{
    return submit_device(c, n, streams, frame_types, pics, nullptr, lens, ordinals, SUBMIT_COPY);
}

HVQ_EXPORT int hvq_submit_many_device_async(HvqContext *c, int n, const int *streams, const int *frame_types,
                                            const uint8_t *const *pics, const size_t *lens, int *ordinals)
{
    return submit_device(c, n, streams, frame_types, pics, nullptr, lens, ordinals, SUBMIT_COPY_ASYNC);
}

/* Zero-copy submit: the caller fills a reservation of the pinned arena itself (a container reader read()s file bytes straight into
 * it), so the 160 MB per batch the copying calls move through the host's caches and DRAM once more never move at all. */
HVQ_EXPORT int hvq_arena_reserve(HvqContext *c, size_t bytes, void **ptr)
{
    if (!c || !ptr || bytes == 0) return fail(HVQ_E_ARG, "bad arguments");
    { int rcj = copy_join(c); if (rcj) return rcj; }
    if (c->resv_active) return fail(HVQ_E_STATE, "an arena reservation is outstanding already: submit it (hvq_submit_many_arena) first");
    HIPCHK(hipSetDevice(c->device));
    const size_t base = align_up(c->arena_used, 256);
    bytes = align_up(bytes, 256);
    { int rc = arena_reserve(c, base - c->arena_used + bytes); if (rc) return rc; }
    if (c->arena_uploaded == c->arena_used) c->arena_uploaded = base;      /* the alignment gap carries nothing */
    c->resv_base = base; c->resv_bytes = bytes; c->resv_active = true;
    c->arena_used = base + bytes;
    *ptr = c->host_arena + base;
    return HVQ_OK;
}

HVQ_EXPORT size_t hvq_arena_stride(size_t len) { return align_up(len + 32, 256); }

HVQ_EXPORT int hvq_submit_many_arena(HvqContext *c, int n, const int *streams, const int *frame_types,
                                     const size_t *offsets, const size_t *lens, int *ordinals)
{
    return submit_device(c, n, streams, frame_types, nullptr, offsets, lens, ordinals, SUBMIT_ARENA);
}

/* GPU-parsed pictures of the pending batch: lay their blobs out, queue the parse kernel and the read-back of its
 * results on the compute stream (which already waits for the bitstreams' H2D).  Nothing here waits for the GPU. */
static int device_parse_launch(HvqContext *c)
{
    std::vector<size_t> &idx = PS(c).fl_idx;
    idx.clear();
    size_t need = 0;
    uint32_t rowbuf = 0;
    for (size_t i = 0; i < c->pending.size(); ++i) {
        Pending &p = c->pending[i];
        if (!p.dev) continue;
        const Stream &s = c->streams[(size_t)p.stream];
        idx.push_back(i);
        need += (size_t)s.blob_cap + s.scratch_bytes + align_up(GP_ALIGN16(HVQ_NESTP_BYTES), 256);
        rowbuf = std::max(rowbuf, (uint32_t)s.layout.hb[0] + 2u);
    }
    if (idx.empty()) return HVQ_OK;
    if (need > PS(c).gp_cap) {
        if (PS(c).gp_dev) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(PS(c).gp_dev)); PS(c).gp_dev = nullptr; PS(c).gp_cap = 0; }
        HIPCHK(hipMalloc((void **)&PS(c).gp_dev, need));
        PS(c).gp_cap = need;
    }
    if (idx.size() > PS(c).pj_cap) {
        if (PS(c).pj_dev) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(PS(c).pj_dev)); HIPCHK(hipFree(PS(c).np_dev)); }
        PS(c).pj_cap = idx.size() * 2;
        HIPCHK(hipMalloc((void **)&PS(c).pj_dev, PS(c).pj_cap * sizeof(HvqParseJob)));
        HIPCHK(hipMalloc((void **)&PS(c).np_dev, PS(c).pj_cap * 2 * sizeof(uint64_t)));
    }
    if (idx.size() > PS(c).pr_host_cap) {
        if (PS(c).pr_host) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipHostFree(PS(c).pr_host)); }
        PS(c).pr_host_cap = idx.size() * 2;
        HIPCHK(hipHostMalloc((void **)&PS(c).pr_host, PS(c).pr_host_cap * sizeof(HvqParseResult), hipHostMallocDefault));
    }
    std::vector<HvqParseJob> &jobs = PS(c).pjobs_host;
    jobs.assign(idx.size(), HvqParseJob{});
    size_t off = 0;
    for (size_t k = 0; k < idx.size(); ++k) {
        Pending &p = c->pending[idx[k]];
        const Stream &s = c->streams[(size_t)p.stream];
        HvqParseJob &j = jobs[k];
        p.dev_blob = (uint64_t)(uintptr_t)(PS(c).gp_dev + off);             off += s.blob_cap;
        j.scratch = (uint64_t)(uintptr_t)(PS(c).gp_dev + off);              off += s.scratch_bytes;
        p.dev_nest = (uint64_t)(uintptr_t)(PS(c).gp_dev + off);             off += align_up(GP_ALIGN16(HVQ_NESTP_BYTES), 256);
        j.pic = (uint64_t)(uintptr_t)(c->dev_arena + p.blob_off);
        j.blob = p.dev_blob;
        j.nest_out = p.dev_nest;
        j.len = (uint32_t)p.blob_len;
        j.pic_dwords = (uint32_t)((p.blob_len + 32) / 4);     /* zero padding the flat parse path may look into (gf_setup_lanes) */
        j.cap = s.blob_cap;
        j.width = (uint16_t)p.w; j.height = (uint16_t)p.h;
        j.frame_type = (uint8_t)(p.kind == HVQ_PIC_I ? HVQ_FRAME_I : (p.kind == HVQ_PIC_P ? HVQ_FRAME_P : HVQ_FRAME_B));
        j.h_samp = s.layout.wshift ? 2 : 1; j.v_samp = s.layout.hshift ? 2 : 1;
        j.is15 = (s.layout.flags & HVQ_F_IS15) ? 1 : 0;
    }
    { int rcu = staged_upload(c, c->arena_id, 0, PS(c).pj_dev, jobs.data(), jobs.size() * sizeof(HvqParseJob)); if (rcu) return rcu; }
    HIPCHK(hipEventRecord(PS(c).evp0, c->stream));
    /* HVQM4_AMD_PARSE_TIMING=1: per-phase times of the parse kernel (development aid, prints to stderr) */
    static const bool want_timing = getenv("HVQM4_AMD_PARSE_TIMING") != nullptr;
    PS(c).timing_dev = nullptr;
    if (want_timing) {
        HIPCHK(hipMalloc((void **)&PS(c).timing_dev, jobs.size() * 16 * sizeof(uint64_t)));
        HIPCHK(hipMemsetAsync(PS(c).timing_dev, 0, jobs.size() * 16 * sizeof(uint64_t), c->stream));
    }
    /* HVQM4_AMD_PARSE_FLAT=0: round 1's chains only (the flat path falls back to them by itself where it has to) */
    static const bool use_flat = !(getenv("HVQM4_AMD_PARSE_FLAT") && atoi(getenv("HVQM4_AMD_PARSE_FLAT")) == 0);
    PS(c).fl_rowbuf = rowbuf;
    /* The parse workgroups write their result records straight into pinned host memory (one 48-byte record per picture).  A
     * read-back copy queued behind the parse kernel would sit on a DMA engine for the whole parse -- and every second batch the
     * NEXT batch's bitstream uploads (copy stream) were dealt to that same engine and started only when the parse ended: periods
     * of 6.2 and 8.4 ms alternating (kernel and copy trace, profiles/r04g_streaming_timeline.txt). */
#ifdef GP_PROBE
    /* probe build: HVQM4_AMD_PARSE_EXIT=<stamp> runs the flat kernel up to that stamp first (its own events, time on stderr) */
    if (const char *ex = getenv("HVQM4_AMD_PARSE_EXIT")) {
        static hipEvent_t pe0 = nullptr, pe1 = nullptr;
        static float last_ms = -1.f;
        if (!pe0) { HIPCHK(hipEventCreate(&pe0)); HIPCHK(hipEventCreate(&pe1)); }
        else if (hipEventQuery(pe1) == hipSuccess) { float pm = 0; if (hipEventElapsedTime(&pm, pe0, pe1) == hipSuccess) last_ms = pm; }
        if (last_ms >= 0) fprintf(stderr, "hvqm4_amd parse probe: exit at stamp %d: %.3f ms (previous batch)\n", atoi(ex), last_ms);
        HIPCHK(hipEventRecord(pe0, c->stream));
        HIPCHK(hvq_launch_parse_probe(PS(c).pj_dev, PS(c).pr_host, (uint32_t)jobs.size(), rowbuf, (uint32_t)atoi(ex), c->stream));
        HIPCHK(hipEventRecord(pe1, c->stream));
        HIPCHK(hipEventRecord(PS(c).evp0, c->stream));
    }
#endif
    HIPCHK(hvq_launch_parse(PS(c).pj_dev, PS(c).pr_host, (uint32_t)jobs.size(), rowbuf, use_flat ? 1u : 0u, nullptr, PS(c).timing_dev, c->stream));
    HIPCHK(hipEventRecord(PS(c).evp1, c->stream));
    HIPCHK(hipEventRecord(PS(c).ev_parse, c->stream));
    return HVQ_OK;
}

static size_t jobs_len_of(const HvqContext *c, size_t k) { return k < PS(c).pjobs_host.size() ? (size_t)PS(c).pjobs_host[k].len : 0; }

/* wait for the parse results of the batch in flight and take them over */
static int device_parse_finish(HvqContext *c)
{
    const std::vector<size_t> &idx = PS(c).fl_idx;
    if (idx.empty()) return HVQ_OK;
    {   /* The reconstruction launches wait for this thread to have seen the parse results: the GPU idles for as long as the wake-up
         * takes.  Polling the event costs this thread a core for the length of the parse kernel and takes the results some tens of
         * microseconds earlier than the driver's wait (HVQM4_AMD_SPIN_WAIT=0: the driver's wait). */
        static const bool spin = !(getenv("HVQM4_AMD_SPIN_WAIT") && atoi(getenv("HVQM4_AMD_SPIN_WAIT")) == 0);
        if (spin) {
            hipError_t q;
            while ((q = hipEventQuery(PS(c).ev_parse)) == hipErrorNotReady) { for (int k = 0; k < 32; ++k) cpu_relax(); }
            HIPCHK(q);
        } else HIPCHK(hipEventSynchronize(PS(c).ev_parse));
    }
    const HvqParseResult *res = PS(c).pr_host;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, PS(c).evp0, PS(c).evp1));
    c->gpu_parse_ms = ms;
    /* Pictures the flat parse path could not serve (sections in an unusual order, array capacities, overflow groups at
     * the chains' caps) come back marked: the chains kernel parses those, same blobs as they would have been. */
    uint32_t n_redo = 0;
    {
        std::vector<uint32_t> redo;
        for (size_t k = 0; k < idx.size(); ++k) if (res[k].pad[0] == 2u) redo.push_back((uint32_t)k);
        n_redo = (uint32_t)redo.size();
        if (n_redo) {
            if (redo.size() > c->redo_cap) {
                if (c->redo_dev) { HIPCHK(hipFree(c->redo_dev)); c->redo_dev = nullptr; c->redo_cap = 0; }
                HIPCHK(hipMalloc((void **)&c->redo_dev, redo.size() * 2 * sizeof(uint32_t)));
                c->redo_cap = redo.size() * 2;
            }
            HIPCHK(hipMemcpyAsync(c->redo_dev, redo.data(), redo.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipEventRecord(c->ev0, c->stream));
            HIPCHK(hvq_launch_parse(PS(c).pj_dev, PS(c).pr_host, n_redo, PS(c).fl_rowbuf, 0u, c->redo_dev, nullptr, c->stream));
            HIPCHK(hipEventRecord(c->ev1, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));       /* also keeps `redo` alive until its upload is done */
            float ms2 = 0;
            HIPCHK(hipEventElapsedTime(&ms2, c->ev0, c->ev1));
            c->gpu_parse_ms += ms2;                        /* the parse time of the batch includes the second launch */
        }
    }
    const bool want_timing_print = PS(c).timing_dev != nullptr;
    if (PS(c).timing_dev) {
        std::vector<uint64_t> tm(idx.size() * 16);
        HIPCHK(hipMemcpy(tm.data(), PS(c).timing_dev, tm.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIPCHK(hipFree(PS(c).timing_dev)); PS(c).timing_dev = nullptr;
        /* stamp k of hvq_parse_kernel (GP_STAMP): average time since the picture's start, in the order they happened */
        static const char *label[16] = { "start", "trees", "mb types", "DC placed", "run sums", "chains ready or decoder", "coefficients/scans",
                                         "merge", "run scan", "entries", "emit count", "emit scan or MV x", "tags+lists", "kinds/DC or lanes",
                                         "expansions", "MV y" };
        double sum[3][16] = {}; size_t cnt[3][16] = {}, pics[3] = {}; uint64_t t_min = ~0ull, t_max = 0;
        uint64_t s_max = 0, d_min = ~0ull, d_max = 0, k_max[3] = { 0, 0, 0 }, k_end[3] = { 0, 0, 0 };
        if (getenv("HVQM4_AMD_PARSE_PROF"))
            for (size_t k = 0; k < idx.size() && k < 16; ++k)
                fprintf(stderr, "prof picture %zu kind %d: wait part %llu  decode part %llu  rounds %llu (shader clocks)\n", k, (int)c->fl_pending[idx[k]].kind,
                        (unsigned long long)tm[16 * k + 9], (unsigned long long)tm[16 * k + 10], (unsigned long long)tm[16 * k + 6]);
        for (size_t k = 0; k < idx.size(); ++k) {
            const uint64_t *t = &tm[16 * k];
            const int kind = (int)c->fl_pending[idx[k]].kind;
            for (int ph = 1; ph < 16; ++ph) if (t[ph]) { sum[kind][ph] += (double)(t[ph] - t[0]) * 0.01; cnt[kind][ph]++; }
            pics[kind]++; t_min = std::min(t_min, t[0]); t_max = std::max(t_max, t[7]);
        }
        for (size_t k = 0; k < idx.size(); ++k) {
            const uint64_t *t = &tm[16 * k];
            const int kind = (int)c->fl_pending[idx[k]].kind;
            s_max = std::max(s_max, t[0]); d_min = std::min(d_min, t[7] - t[0]); d_max = std::max(d_max, t[7] - t[0]);
            k_max[kind] = std::max(k_max[kind], t[7] - t[0]); k_end[kind] = std::max(k_end[kind], t[7] - t_min);
        }
        fprintf(stderr, "hvqm4_amd parse: slowest picture / last end per kind: I %.3f / %.3f  P %.3f / %.3f  B %.3f / %.3f ms\n",
                (double)k_max[0] * 1e-5, (double)k_end[0] * 1e-5, (double)k_max[1] * 1e-5, (double)k_end[1] * 1e-5,
                (double)k_max[2] * 1e-5, (double)k_end[2] * 1e-5);
        fprintf(stderr, "hvqm4_amd parse occupancy: %d workgroups per CU (runtime query)\n", hvq_parse_occupancy(256));
        fprintf(stderr, "hvqm4_amd parse timing: %zu pictures, kernel %.3f ms, first start -> last end %.3f ms, last start +%.3f ms, "
                "per picture %.3f .. %.3f ms\n",
                idx.size(), ms, (double)(t_max - t_min) * 1e-5, (double)(s_max - t_min) * 1e-5, (double)d_min * 1e-5, (double)d_max * 1e-5);
        if (const char *dump = getenv("HVQM4_AMD_PARSE_TIMING_DUMP")) {       /* raw rows of the latest batch: kind, bytes, 16 stamps (10 ns) */
            static int seq = 0;
            const std::string path = std::string(dump) + "." + std::to_string(seq++);
            if (FILE *f = seq <= 12 ? fopen(path.c_str(), "w") : nullptr) {
                for (size_t k = 0; k < idx.size(); ++k) {
                    fprintf(f, "%d %zu", (int)c->fl_pending[idx[k]].kind, (size_t)jobs_len_of(c, k));
                    for (int ph = 0; ph < 16; ++ph) fprintf(f, " %llu", (unsigned long long)(tm[16 * k + ph] ? tm[16 * k + ph] - t_min : 0ull));
                    fprintf(f, "\n");
                }
                fclose(f);
            }
        }
        for (int kind = 0; kind < 3; ++kind) {
            if (!pics[kind]) continue;
            std::vector<std::pair<double, int>> ord;
            for (int ph = 1; ph < 16; ++ph) if (cnt[kind][ph]) ord.push_back({ sum[kind][ph] / (double)cnt[kind][ph], ph });
            std::sort(ord.begin(), ord.end());
            fprintf(stderr, "  %c pictures (%zu), us since start:", "IPB"[kind], pics[kind]);
            double prev = 0;
            for (auto &o : ord) { fprintf(stderr, " %s %.0f (+%.0f) |", label[o.second], o.first, o.first - prev); prev = o.first; }
            fprintf(stderr, "\n");
        }
    }
    {
        const uint32_t retried = n_redo;
        uint64_t spins = 0;
        for (size_t k = 0; k < idx.size(); ++k) spins += res[k].pad[1];
        c->gpu_parse_retried = retried;
        if (want_timing_print) fprintf(stderr, "hvqm4_amd parse: %u of %zu pictures handed to the chains; decode wave waited %.1f rounds per picture\n",
                                       retried, idx.size(), (double)spins / (double)idx.size());
    }
    /* Overflow-symbol runs that outlast the device parser's cap (GP_SOVF_CAP symbols; the picture comes back HVQ_F_CAPPED): the
     * reference sums for as long as the stream says (h4m:654-677), and so does the HOST parser since round 5 -- bounded by the bits
     * the picture has left.  Such a picture (nothing an encoder writes; rounds 3-4 refused it) is parsed once more, here, from its
     * bitstream in the pinned arena; its blob replaces the device parser's, an I picture's nest is put where the stream's later
     * pictures look for it.  What even the host parser cannot end (a one-leaf tree outside the window) stays refused.
     * Bounded per batch (HVQ_REDO_MAX pictures, HVQ_REDO_BYTES of blobs): the re-parse runs serially on the caller's thread between a
     * batch's parse results and its first launch, and a crafted input whose pictures are ALL capped must not turn a 4 ms flush into
     * seconds for every stream of the batch -- beyond the bound a capped picture is refused as rounds 3-4 refused all of them. */
    constexpr uint32_t HVQ_REDO_MAX = 32;
    constexpr size_t HVQ_REDO_BYTES = (size_t)32 << 20;
    uint32_t n_reparsed = 0;
    c->rp_host.clear();
    for (size_t k = 0; k < idx.size(); ++k) {
        Pending &p = c->fl_pending[idx[k]];
        const size_t raw_len = jobs_len_of(c, k);
        p.status = (int)res[k].status;                 /* judged per picture by flush_end: one bad clip must not poison the batch */
        p.max_items = res[k].max_items; p.max_pairs = res[k].max_pairs;
        p.flags = res[k].flags;
        p.pool_dwords = res[k].pool_dwords;
        p.redo = false;
        if (!p.status && (p.flags & HVQ_F_CAPPED) && !p.dropped && raw_len && n_reparsed < HVQ_REDO_MAX && c->rp_host.size() < HVQ_REDO_BYTES) {
            ++n_reparsed;
            Stream &s = c->streams[(size_t)p.stream];
            const size_t bound = hvq_parser_blob_bound(s.parser), off = align_up(c->rp_host.size(), 256);
            c->rp_host.resize(off + bound + 2048);
            size_t out_len = 0;
            const int ft = p.kind == HVQ_PIC_I ? HVQ_FRAME_I : p.kind == HVQ_PIC_P ? HVQ_FRAME_P : HVQ_FRAME_B;
            const int rcp = hvq_parse_picture(s.parser, ft, c->fl_host + p.blob_off, raw_len, c->rp_host.data() + off, bound, &out_len);
            const HvqPicHeader *hd = (const HvqPicHeader *)(c->rp_host.data() + off);
            const uint32_t all_flags = hd->flags | hvq_parser_last_flags(s.parser);     /* incl. what pass 2 raised behind the header */
            if (rcp == HVQ_OK && !(all_flags & HVQ_F_CAPPED)) {
                p.redo = true; p.redo_off = off; p.redo_hd = *hd; p.redo_hd.flags = all_flags;
                out_len = align_up(out_len, 256);
                if (p.kind == HVQ_PIC_I) hvq_parser_packed_nest(s.parser, c->rp_host.data() + off + out_len);   /* behind the blob: always, not only when the blob has one */
                c->rp_host.resize(off + out_len + (p.kind == HVQ_PIC_I ? align_up(GP_ALIGN16(HVQ_NESTP_BYTES), 256) : 0));
                p.flags = all_flags; p.pool_dwords = hd->pool_dwords; p.max_items = hd->max_items; p.max_pairs = hd->max_pairs;
                p.blob_len = hd->total_bytes;
                continue;
            }
            c->rp_host.resize(off);                      /* still capped (or unparsable): refused by flush_end */
        }
        p.blob_len = res[k].total_bytes;
    }
    if (!c->rp_host.empty()) {
        if (c->rp_host.size() > c->rp_cap) {
            if (c->rp_dev) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->rp_dev)); c->rp_dev = nullptr; c->rp_cap = 0; }
            HIPCHK(hipMalloc((void **)&c->rp_dev, c->rp_host.size() * 2));
            c->rp_cap = c->rp_host.size() * 2;
        }
        { int rcu = staged_upload(c, c->fl_arena_id, 4, c->rp_dev, c->rp_host.data(), c->rp_host.size()); if (rcu) return rcu; }
        for (size_t k = 0; k < idx.size(); ++k) {
            const Pending &p = c->fl_pending[idx[k]];
            if (p.redo && p.kind == HVQ_PIC_I && p.dev_nest)
                HIPCHK(hipMemcpyAsync((void *)(uintptr_t)p.dev_nest, c->rp_dev + p.redo_off + align_up(p.redo_hd.total_bytes, 256),
                                      GP_ALIGN16(HVQ_NESTP_BYTES), hipMemcpyDeviceToDevice, c->stream));
        }
    }
    return HVQ_OK;
}

/* The launches of the resident batch go out dependency level by dependency level, every launch on the queue build_tiles dealt its
 * streams to (one queue, or two for large uniform batches: see there).
 * The second launch queue forks from the main stream (everything queued there so far is done before its first launch) ... */
static int queues_fork(HvqContext *c, bool *two)
{
    int nq = 1;
    for (auto &L : c->launches) nq = std::max(nq, L.queue + 1);
    *two = nq > 1;
    if (!*two) return HVQ_OK;
    if (!c->ev_fork) HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    for (int q = 1; q < nq; ++q) {
        if (!c->qstream[q]) {
            /* at the launch streams' priority level (hvq_context_create): a hardware queue of its own */
            int lo = 0, hi = 0;
            static const bool prio = !(getenv("HVQM4_AMD_STREAM_PRIORITY") && atoi(getenv("HVQM4_AMD_STREAM_PRIORITY")) == 0);
            if (prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
                HIPCHK(hipStreamCreateWithPriority(&c->qstream[q], hipStreamNonBlocking, hi));
            else HIPCHK(hipStreamCreateWithFlags(&c->qstream[q], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&c->ev_join[q], hipEventDisableTiming));
        }
        HIPCHK(hipStreamWaitEvent(c->qstream[q], c->ev_fork, 0));
    }
    return HVQ_OK;
}

/* ... and joins it again: whatever is queued on the main stream next runs behind both queues */
static int queues_join(HvqContext *c, bool two)
{
    if (!two) return HVQ_OK;
    int nq = 1;
    for (auto &L : c->launches) nq = std::max(nq, L.queue + 1);
    for (int q = 1; q < nq; ++q) {
        HIPCHK(hipEventRecord(c->ev_join[q], c->qstream[q]));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join[q], 0));
    }
    return HVQ_OK;
}

/* one pass over the launches of the resident batch, every launch on its queue (between queues_fork and queues_join) */
static int run_launches(HvqContext *c)
{
    for (auto &L : c->launches) {
        hipStream_t st = L.queue ? c->qstream[L.queue] : c->stream;
        HIPCHK(hvq_launch_recon_inline(c->jobs_dev + L.first_tile, L.ntiles, L.max_tiles, L.tpw, L.items_cap, L.pair_cap, L.pool_cap, st));
        for (const SelfRef &sr : c->selfrefs) {
            if (sr.level != L.level || sr.queue != L.queue) continue;
            if (sr.old_host) HIPCHK(hipMemcpyAsync(sr.dst, sr.old_host, sr.pic_bytes, hipMemcpyHostToDevice, st));
            else if (sr.old_dev) HIPCHK(hipMemcpyAsync(sr.dst, sr.old_dev, sr.pic_bytes, hipMemcpyDeviceToDevice, st));
            HIPCHK(hvq_launch_selfref(c->jobs_dev + sr.job, c->selfref_dev + sr.side_off, sr.dst, st));
        }
    }
    return HVQ_OK;
}

/* launch table of the batch in flight: one launch per dependency level (and queue), one slot {job, tiles} per picture; the
 * slot count is padded to a multiple of 8 so that grid column x runs on XCD x % 8 and a picture's tiles share one L2.
 * Needs nothing from the GPU, so it is built and uploaded while the parse kernel runs. */
static int build_tiles(HvqContext *c)
{
    std::vector<HvqTileRef> &tiles = c->tiles_host;
    tiles.clear();
    c->fl_launches.clear();
    int max_level = 0;
    for (auto &p : c->fl_pending) max_level = std::max(max_level, p.level);
    /* Two launch queues: clips are independent, so the dependency levels of two halves of the streams form two chains of launches
     * that run on two HIP streams with a hardware queue each -- while one chain drains a level or waits at the head of the next, the
     * other keeps the CUs busy (a launch is some twenty generations of workgroups; the first and the last of them leave CUs idle).
     * Rounds 1-4 had this behind HVQM4_AMD_QUEUES and dropped it (-7 ... -9 %): the second stream was a plain one, and the runtime deals
     * plain streams to four hardware queues in creation order -- with the context's copy and read-back streams in between, the two
     * launch streams could share one.  At the launch streams' own priority level: dense 965 -> 900 us per step (tools/replay_pair_probe.py,
     * profiles/r05_flush_next.txt 7); 64 / 32 / 16 streams gain 7 / 14 / 19 %.  Batches of fewer than 16 streams keep one queue (each queue's
     * launches should hold the 8 picture slots that spread a launch over the XCDs); HVQM4_AMD_QUEUES=1 / 2 forces either. */
    int nq = 1;
    {
        static const int qenv = getenv("HVQM4_AMD_QUEUES") ? atoi(getenv("HVQM4_AMD_QUEUES")) : 0;
        std::vector<char> seen(c->streams.size(), 0);
        size_t nstreams = 0;
        bool uniform = true;                               /* every picture of the batch has the same number of tiles */
        for (auto &p : c->fl_pending) {
            if (!seen[(size_t)p.stream]) { seen[(size_t)p.stream] = 1; ++nstreams; }
            uniform &= p.ntiles == c->fl_pending[0].ntiles;
        }
        /* mixed picture sizes keep one queue: BASELINE config 4 (320x240 and 640x480 clips alternating, 25 levels) took 1762 us per step
         * with two queues against 1303 with one, however the streams were dealt (a launch's grid is as tall as its largest picture) */
        nq = qenv >= 2 ? std::min(qenv, 4) : (qenv == 1 ? 1 : (c->max_queues >= 2 && nstreams >= 16 && uniform ? 2 : 1));
        /* streams to queues by WORK (tiles of their pictures in this batch), heaviest first to the lighter queue: clips of mixed sizes
         * (BASELINE config 4 alternates 320x240 and 640x480) dealt by parity put every large clip on one queue, and that chain then ran
         * alone for most of the step (1766 against 1295 us) */
        c->fl_qof.assign(c->streams.size(), 0);
        if (nq >= 2) {
            std::vector<uint64_t> work(c->streams.size(), 0);
            for (auto &p : c->fl_pending) work[(size_t)p.stream] += p.ntiles;
            std::vector<uint32_t> order;
            for (size_t sidx = 0; sidx < work.size(); ++sidx) if (seen[sidx]) order.push_back((uint32_t)sidx);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return work[a] > work[b]; });
            uint64_t load[4] = { 0, 0, 0, 0 };
            for (uint32_t sidx : order) {
                int q = 0;
                for (int k = 1; k < nq; ++k) if (load[k] < load[q]) q = k;
                c->fl_qof[sidx] = (uint8_t)q; load[q] += work[sidx];
            }
        }
    }
    for (int lvl = 0; lvl <= max_level; ++lvl)
      for (int qi = 0; qi < nq; ++qi) {
        Launch L{};
        L.queue = qi;
        L.level = lvl; L.first_tile = (uint32_t)tiles.size();
        /* submission order (sorting same-stream pictures into one grid column of consecutive groups was tried: +1 % dense, -3 % flat, dropped) */
        /* (walking odd levels in reverse order, so that a level reads first the anchors its predecessor wrote last, was measured in
         * round 5: dense -1.5 %, flat -3 %, profiles/r05_recon_steps.txt) */
        for (size_t i = 0; i < c->fl_pending.size(); ++i) {
            const Pending &p = c->fl_pending[i];
            if (p.level != lvl || (nq >= 2 && c->fl_qof[(size_t)p.stream] != qi)) continue;
            tiles.push_back(HvqTileRef{ (uint32_t)i, p.ntiles });
            L.max_wg[0] = std::max(L.max_wg[0], p.ntiles); L.wgs[0] += p.ntiles;
            L.max_wg[1] = std::max(L.max_wg[1], p.nwg); L.wgs[1] += p.nwg;
        }
        L.ntiles = (uint32_t)tiles.size() - L.first_tile;
        if (!L.ntiles) continue;
        if (L.ntiles >= 8)
            while (L.ntiles & 7u) { tiles.push_back(HvqTileRef{ 0xFFFFFFFFu, 0u }); ++L.ntiles; }   /* padding slots exit at once */
        c->fl_launches.push_back(L);
    }
    return HVQ_OK;                     /* the slots become the ORDER of the job table (flush_end): a workgroup finds its job by its grid position */
}

/* First half of a flush: everything that can be queued without waiting for the GPU.  The pending batch becomes the
 * batch in flight; the caller may queue the NEXT batch (hvq_submit_*) before hvq_flush_end -- its bitstreams are copied
 * and uploaded (other arena, copy stream) while this batch is being parsed. */
/* begin, part A: the queued batch's bitstreams go up (what the early uploads left); its parse kernel is queued behind them
 * (device_parse_launch) */
static int begin_upload(HvqContext *c)
{
    /* the pinned staging of this arena id is free again (hvq_flush_next knows that without asking: arena_idle) */
    if (!c->arena_idle[c->arena_id]) HIPCHK(hipEventSynchronize(c->ev_arena_free[c->arena_id]));
    /* 1. descriptors / bitstreams -> HBM; the compute stream waits for the copy stream */
    { int rc = arena_upload(c, c->arena_used); if (rc) return rc; }
    HIPCHK(hipEventRecord(c->ev_copy, c->copy_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_copy, 0));
    return HVQ_OK;
}

/* begin, part B: the queued batch becomes the batch in flight */
static int begin_rest(HvqContext *c, int rc)
{
    /* the nest a GPU-parsed P/B picture uses: its batch's governing I picture, else the stream's kept one; the last I
     * picture of every stream is committed to the other kept slot at the end of the batch */
    c->fl_nest_pairs.clear();
    c->fl_nest_streams.clear();
    for (auto &p : c->pending)
        if (p.dev) {
            const Stream &s = c->streams[(size_t)p.stream];
            p.nest_ptr = p.nest_ref >= 0 ? c->pending[(size_t)p.nest_ref].dev_nest : (uint64_t)(uintptr_t)s.nest_keep_ptr(s.nest_cur);
        }
    for (auto &s : c->streams)
        if (s.open && s.nest_src >= 0) {
            c->fl_nest_pairs.push_back(c->pending[(size_t)s.nest_src].dev_nest);
            c->fl_nest_pairs.push_back((uint64_t)(uintptr_t)s.nest_keep_ptr(s.nest_cur ^ 1));   /* replays of this batch keep reading [nest_cur] */
            c->fl_nest_streams.push_back((int)(&s - c->streams.data()));
            s.nest_cur ^= 1;
            s.nest_src = -1;
        }
    /* the batch is in flight: levels restart from zero for whatever is queued next, which goes to the other arena */
    c->fl_pending.swap(c->pending);
    c->pending.clear();
    for (auto &p : c->fl_pending) {
        Stream &s = c->streams[(size_t)p.stream];
        s.inflight_from = std::min(s.inflight_from, p.ordinal);
    }
    for (auto &s : c->streams)
        for (auto &sl : s.slots) { sl.w_level = -1; sl.r_level = -1; }
    c->fl_host = c->host_arena; c->fl_dev = c->dev_arena; c->fl_arena_id = c->arena_id;
    c->arena_idle[c->arena_id] = false;
    std::swap(c->host_arena, c->host_arena_alt);
    std::swap(c->dev_arena, c->dev_arena_alt);
    std::swap(c->arena_cap, c->arena_cap_alt);
    c->arena_id ^= 1;
    c->arena_used = 0; c->arena_uploaded = 0; c->arena_waited = false;
    c->resv_active = false;            /* a reservation that was never submitted goes with its arena */
    c->fl_active = true;
    if (!rc) rc = build_tiles(c);
    if (rc) return flush_abandon(c, rc);
    return HVQ_OK;
}

/* First half of a flush: everything that can be queued without waiting for the GPU.  The pending batch becomes the
 * batch in flight; the caller may queue the NEXT batch (hvq_submit_*) before hvq_flush_end -- its bitstreams are copied
 * and uploaded (other arena, copy stream) while this batch is being parsed. */
HVQ_EXPORT int hvq_flush_begin(HvqContext *c)
{
    if (!c) return fail(HVQ_E_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    { int rc = flush_end(c); if (rc) return rc; }            /* at most one batch in flight */
    { int rcj = copy_join(c); if (rcj) return rcj; }         /* the queued batch's bitstreams are in the arena (streaming: copied by a worker) */
    if (c->pending.empty()) return HVQ_OK;
    const double tb0 = now_ms();
    { int rc = begin_upload(c); if (rc) return rc; }
    /* 1b. streams parsed on the GPU: bitstreams -> blobs, one launch */
    int rc = device_parse_launch(c);
    rc = begin_rest(c, rc);
    if (flush_timing()) fprintf(stderr, "flush_begin %.3f -> %.3f ms\n", tb0, now_ms());
    return rc;
}

/* Streaming step: end the batch in flight (k) and begin the queued one (k + 1) -- in the order that keeps the GPU busy.  The plain
 * pair hvq_flush_end / hvq_flush_begin leaves the GPU idle between the parse kernel of batch k and its first reconstruction launch
 * for as long as the host takes over the parse results and the job table (0.13-0.3 ms of a 4 ms period, measured); here the parse
 * kernel of batch k + 1 is queued FIRST, on its own set of buffers, so the host's part of batch k runs beside it and the launches of
 * batch k line up behind it.  Taken only when it is safe and useful: both batches parsed on the GPU throughout (a host-parsed blob
 * lives in the arena until its reconstruction ends) and the queued batch's bitstreams in the arena before batch k's parse results
 * arrive; otherwise this IS hvq_flush_end followed by hvq_flush_begin.  Returns the first error of either half; the state
 * afterwards is the one the plain pair leaves. */
HVQ_EXPORT int hvq_flush_next(HvqContext *c)
{
    if (!c) return fail(HVQ_E_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    static const bool allow = !(getenv("HVQM4_AMD_FLUSH_NEXT") && atoi(getenv("HVQM4_AMD_FLUSH_NEXT")) == 0);
    bool ahead = allow && c->fl_active && !c->pending.empty() && !PS(c).fl_idx.empty();
    if (ahead) {
        for (auto &p : c->fl_pending) if (!p.dev) { ahead = false; break; }
        for (auto &p : c->pending) if (!p.dev) { ahead = false; break; }
    }
    if (ahead && c->copy_active) {
        /* whichever comes first: the queued batch's bitstreams are in the arena (go ahead), or batch k's parse results are there
         * (its launches must not wait for a slow copy: plain order) */
        for (;;) {
            if (c->copy_done.load(std::memory_order_acquire)) break;
            if (hipEventQuery(PS(c).ev_parse) != hipErrorNotReady) { ahead = false; break; }
            for (int k = 0; k < 32; ++k) cpu_relax();
        }
    }
    if (!ahead) {
        if (flush_timing()) fprintf(stderr, "flush_next  %.3f: plain order\n", now_ms());
        const int rc_end = flush_end(c);
        const int rc_begin = hvq_flush_begin(c);
        return rc_end ? rc_end : rc_begin;
    }
    { int rcj = copy_join(c); if (rcj) { const int rc_end = flush_end(c); return rc_end ? rc_end : rcj; } }
    const double tb0 = now_ms();
    const int fl_arena = c->fl_arena_id;
    /* Everything stays on the one launch stream: parse k + 1, then the launches of batch k, in that order.  The parse kernel on a
     * stream of its own, so that the reconstruction of batch k runs BESIDE it, was built and measured (profiles/r05_flush_next.txt):
     * both kernels are bound by instruction issue, whichever reaches the CUs first keeps the other out until its workgroups thin
     * out, periods alternate 3.45 / 4.9 ms and their mean (4.17) is worse than this order's 3.94. */
    { int rcu = begin_upload(c); if (rcu) { const int rc_end = flush_end(c); return rc_end ? rc_end : rcu; } }   /* the queued batch stays queued */
    c->ps_live ^= 1;                   /* batch k + 1 parses into the other set ... */
    const int rc_parse = device_parse_launch(c);
    c->ps_live ^= 1;                   /* ... while batch k is finished from its own */
    c->abandoned = false;
    const int rc_end = flush_end(c);
    /* every GPU reader of batch k's arena is done (its parse results were taken, a second parse launch was waited for), and what
     * its staging buffers still feed is consumed before the host writes them again (behind the parse of the batch after next) */
    if (!c->abandoned) c->arena_idle[fl_arena] = true;
    c->ps_live ^= 1;
    const int rc_begin = begin_rest(c, rc_parse);
    if (flush_timing()) fprintf(stderr, "flush_next  %.3f -> %.3f ms\n", tb0, now_ms());
    return rc_end ? rc_end : rc_begin;
}

/* a flush that fails half way: the pictures of the batch in flight were never reconstructed -- they must not read as resident
 * (hvq_read_pictures / hvq_picture_device_ptr would otherwise hand out whatever their slots held), and nothing is in flight */
static int flush_abandon(HvqContext *c, int rc)
{
    for (auto &p : c->fl_pending) {
        Stream &s = c->streams[(size_t)p.stream];
        if ((size_t)p.ordinal < s.pic_slot.size() && s.pic_slot[(size_t)p.ordinal] == p.dst) {
            s.pic_slot[(size_t)p.ordinal] = -1;
            if (p.dst >= 0 && s.slots[(size_t)p.dst].pic == p.ordinal) s.slots[(size_t)p.dst].pic = -1;
        }
        s.anchor_old = s.anchor_new = -1;          /* the stream's references are gone with them: it resumes at its next I picture */
        s.need_I = true;
    }
    for (auto &s : c->streams) s.inflight_from = 0x7FFFFFFF;
    c->fl_active = false;
    c->abandoned = true;
    c->fl_pending.clear();
    PS(c).fl_idx.clear();
    c->launches.clear();                           /* nothing coherent to replay */
    return rc;
}
#define HIPCHK_FL(expr)                                                                                              \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) return flush_abandon(c, fail(HVQ_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)));      \
    } while (0)

/* Second half: take the parse results, build the job and tile tables, launch the reconstruction. */
static int flush_end(HvqContext *c)
{
    if (!c->fl_active) return HVQ_OK;
    c->fl_active = false;
    const double te0 = now_ms();
    { int rc = device_parse_finish(c); if (rc) return flush_abandon(c, rc); }
    const double te1 = now_ms();
    /* Judge the GPU-parsed pictures one by one.  A picture the parser could not take (status) or that this back end refuses
     * (unsupported_reason) is dropped together with every later picture of ITS stream in the batch; all other streams are
     * reconstructed as if it had not been there.  The dropped pictures read as "not resident", the stream waits for its next
     * I picture, and the flush reports the first such error after everything else has been launched. */
    int first_rc = HVQ_OK;
    uint32_t n_dropped = 0;
    {
        std::vector<char> broken(c->streams.size(), 0);
        for (size_t i = 0; i < c->fl_pending.size(); ++i) {
            Pending &p = c->fl_pending[i];
            Stream &s = c->streams[(size_t)p.stream];
            const char *why = nullptr;
            int code = HVQ_OK;
            if ((!broken[(size_t)p.stream] || p.kind == HVQ_PIC_I) && !p.dropped && p.dev) {
                if (p.status) { code = (p.status & GP_ST_OVERFLOW) ? HVQ_E_OVERFLOW : HVQ_E_ARG; why = "the GPU parser rejected the bitstream"; }
                else if ((why = unsupported_reason(p.flags)) != nullptr) code = HVQ_E_UNSUPPORTED;
            }
            if (!broken[(size_t)p.stream] && !p.dropped && !code && p.kind == HVQ_PIC_P && p.old_slot == -2 && !p.host_old) {
                const uint32_t fl = p.dev ? p.flags : ((const HvqPicHeader *)(c->fl_host + p.blob_off))->flags;
                if (fl & HVQ_F_SELF_REF) {
                    code = HVQ_E_UNSUPPORTED;
                    why = "P picture with future-referencing macroblocks (h4m:2058-2061) whose `present` buffer content -- the picture "
                          "the reference player's third buffer holds at this point -- is no longer resident: open the stream with more slots";
                }
            }
            if (code) {
                broken[(size_t)p.stream] = 1;
                if (!first_rc) first_rc = fail(code, "stream %d picture %d: %s (status %d); dropped with the later pictures of this stream, "
                                               "the other streams of the batch were decoded", p.stream, p.ordinal, why, p.status);
            }
            if (p.kind == HVQ_PIC_I && !code) broken[(size_t)p.stream] = 0;     /* an I picture restarts its stream inside the batch */
            /* dropped: follows a rejected picture of its stream in this batch, or was marked when the batch before this one was
             * judged (streaming: it was queued already).  Either way it is not reconstructed and must not read as resident. */
            if (!broken[(size_t)p.stream] && !p.dropped) continue;
            p.dropped = true;
            ++n_dropped;
            if ((size_t)p.ordinal < s.pic_slot.size() && s.pic_slot[(size_t)p.ordinal] == p.dst) {
                s.pic_slot[(size_t)p.ordinal] = -1;
                if (s.slots[(size_t)p.dst].pic == p.ordinal) s.slots[(size_t)p.dst].pic = -1;
            }
        }
        for (size_t sid = 0; sid < broken.size(); ++sid)
            if (broken[sid]) {
                Stream &s = c->streams[sid];
                /* what is already queued for the next batch follows the rejected picture: P/B pictures up to the first queued
                 * I picture are dropped; with an I picture queued the stream has restarted there and keeps its state */
                bool restarted = false;
                for (auto &q : c->pending) {
                    if (q.stream != (int)sid) continue;
                    if (q.kind == HVQ_PIC_I) { restarted = true; break; }
                    q.dropped = true;
                }
                if (!restarted) { s.anchor_old = s.anchor_new = -1; s.need_I = true; }
            }
        /* nests of broken streams are not committed (their last I picture may be among the dropped) */
        std::vector<uint64_t> keep;
        for (size_t k = 0; k < c->fl_nest_streams.size(); ++k)
            if (!broken[(size_t)c->fl_nest_streams[k]]) { keep.push_back(c->fl_nest_pairs[2 * k]); keep.push_back(c->fl_nest_pairs[2 * k + 1]); }
        c->fl_nest_pairs.swap(keep);
    }
    /* 2. job table (the tile table went up at begin) */
    std::vector<HvqJob> &jobs = c->jobs_host;
    const std::vector<HvqTileRef> &slots = c->tiles_host;          /* launch slots in launch order (build_tiles) */
    jobs.assign(slots.size(), HvqJob{});
    std::vector<size_t> tq_off(slots.size(), 0);
    std::vector<char> has_tq(slots.size(), 0);
    std::vector<uint32_t> slot_of(c->fl_pending.size(), 0);
    size_t tq_bytes = 0, side_bytes = 0;
    c->selfrefs.clear();
    HvqStats st{};
    for (size_t k = 0; k < slots.size(); ++k) {
        HvqJob &j = jobs[k];                                        /* zeroed by the assign above */
        if (slots[k].job == 0xFFFFFFFFu) continue;                  /* padding slot: total_tiles = 0, its workgroups exit at once */
        const size_t i = slots[k].job;
        slot_of[i] = (uint32_t)k;
        const Pending &p = c->fl_pending[i];
        const Stream &s = c->streams[(size_t)p.stream];
        HvqPicHeader hdev;
        if (p.dev && !p.redo) {                        /* geometry from the stream, per-picture fields from the parse result */
            hdev = s.layout;
            hdev.pic_kind = (uint8_t)p.kind; hdev.unk_shift = (uint8_t)p.unk_shift; hdev.flags = p.flags;
            if (p.kind == HVQ_PIC_I) hdev.mv_off = 0;
            hdev.nest_off = 1;
        }
        /* a GPU-parsed picture that was parsed again on the host (device_parse_finish): the host parser's blob and header */
        const HvqPicHeader *hd = p.redo ? &p.redo_hd : p.dev ? &hdev : (const HvqPicHeader *)(c->fl_host + p.blob_off);
        const uint64_t blob = p.redo ? (uint64_t)(uintptr_t)(c->rp_dev + p.redo_off) : p.dev ? p.dev_blob : (uint64_t)(uintptr_t)(c->fl_dev + p.blob_off);
        const uint64_t dst = (uint64_t)(uintptr_t)s.slot_ptr(p.dst);
        j.ring = (uint64_t)(uintptr_t)s.dev;                 /* every reference read is ring + a 32-bit offset */
        j.ref0_off = (uint32_t)(s.slot_ptr(p.ref0) - s.dev);
        j.ref1_off = (uint32_t)(s.slot_ptr(p.ref1) - s.dev);
        j.pool = blob + hd->pool_off;
        j.mv = blob + hd->mv_off;
        j.wave_base = blob + hd->wave_base_off;
        j.nest = hd->nest_off ? blob + hd->nest_off : 0;
        if (p.dev) j.nest = p.nest_ptr;
        j.slot_bytes = s.slot_bytes;
        j.flags = (hd->flags & 0xFFFFu) | ((uint32_t)hd->pic_kind << HVQ_JOB_KIND_SHIFT) | ((uint32_t)hd->unk_shift << HVQ_JOB_UNK_SHIFT);
        j.width = hd->width;
        j.mcb_w = hd->mcb_w;
        j.pool_dwords = (p.dev && !p.redo) ? p.pool_dwords : hd->pool_dwords;
        j.total_tiles = p.dropped ? 0u : hd->tile_first[3];          /* 0: the tile records of this picture become padding entries */
        if (p.dropped) continue;
        if (p.kind == HVQ_PIC_P && (hd->flags & HVQ_F_SELF_REF)) {
            /* a self-referencing P picture: the data-parallel pass writes a side buffer and leaves every block's pool offset in a
             * section of `tq_dev`; the walk behind this level's launch (hvq_selfref_kernel) merges the side buffer into the slot */
            const uint32_t nt = hd->tile_first[3];
            tq_bytes = align_up(tq_bytes, 256);
            tq_off[k] = tq_bytes;
            has_tq[k] = 1;
            j.q_offs_off = 16;
            tq_bytes += 16 + (size_t)nt * HVQ_TILE_BLOCKS * 4;
            SelfRef sr{};
            sr.level = p.level; sr.job = (uint32_t)k;
            sr.old_host = p.host_old;
            sr.old_dev = (p.host_old || p.old_slot == p.dst) ? nullptr : s.slot_ptr(p.old_slot);     /* -1: the zero slot */
            sr.dst = s.slot_ptr(p.dst);
            side_bytes = align_up(side_bytes, 256);
            sr.side_off = side_bytes;
            sr.pic_bytes = s.pic_bytes;
            side_bytes += s.slot_bytes;
            c->selfrefs.push_back(sr);
        }
        for (int k = 0; k < 3; ++k) {
            HvqPlaneRec &r = j.plane[k];
            r.map = blob + hd->map_off[k];
            r.dst = dst + hd->plane_off[k];
            r.plane_off = hd->plane_off[k];
            r.tile_first = hd->tile_first[k];
            const uint32_t ws = k ? hd->wshift : 0, hs = k ? hd->hshift : 0;
            r.hbvb = (uint32_t)hd->hb[k] | ((uint32_t)hd->vb[k] << 16);
            r.pw_sub = (uint32_t)(hd->width >> ws) | (ws << 16) | (hs << 24);
            if (k) j.tile_first12[k - 1] = hd->tile_first[k];
            j.hb_magic[k] = hd->hb[k] > 1 ? (uint32_t)(0x100000000ull / hd->hb[k]) + 1u : 0u;
            j.hb_magic16[k] = hd->hb[k] ? (65536u + hd->hb[k] - 1u) / hd->hb[k] : 0u;
        }
        st.pictures++;
        st.luma_pixels += (uint64_t)p.w * p.h;
        st.algorithmic_bytes += (uint64_t)s.pic_bytes * (p.kind == HVQ_PIC_I ? 1u : 2u);
        st.descriptor_bytes += p.blob_len;
        st.flags_or |= hd->flags;
        st.gpu_parsed += p.dev ? 1u : 0u;
    }
    /* LDS sizes of the launches (their tile ranges were dealt at begin): accumulators are 16 dwords per queued block,
     * rows padded to 32 entries (LDS banks); pairs above the cap take the kernel's serial fallback */
    /* fullest tile per launch: one pass over the pictures (this runs between the parse results and the first launch: the GPU waits) */
    std::vector<uint32_t> lmi(c->fl_launches.size(), 0), lmp(c->fl_launches.size(), 0);
    bool twoq = false;
    int nqs = 1;
    for (auto &L : c->fl_launches) { twoq |= L.queue >= 1; nqs = std::max(nqs, L.queue + 1); }
    for (auto &sr : c->selfrefs) sr.queue = twoq ? (int)c->fl_qof[(size_t)c->fl_pending[slots[sr.job].job].stream] : 0;
    for (size_t i = 0; i < c->fl_pending.size(); ++i) {
        const Pending &p = c->fl_pending[i];
        if (p.dropped) continue;
        for (size_t l = 0; l < c->fl_launches.size(); ++l) {
            const Launch &L = c->fl_launches[l];
            if (p.level != L.level || (twoq && (int)c->fl_qof[(size_t)p.stream] != L.queue)) continue;
            lmi[l] = std::max(lmi[l], p.max_items); lmp[l] = std::max(lmp[l], p.max_pairs);
            break;
        }
    }
    for (size_t li = 0; li < c->fl_launches.size(); ++li) {
        Launch &L = c->fl_launches[li];
        const uint32_t mi = lmi[li], mp = lmp[li];
        static const int force_tpw = getenv("HVQM4_AMD_TILES_PER_WG") ? atoi(getenv("HVQM4_AMD_TILES_PER_WG")) : 0;
        {
            /* hvq_recon_inline_kernel keeps its item queue, pair list and the tile range of the pool in LDS: accumulator rows
             * for the fullest tile (x tiles per workgroup), a pair list for its pairs, the staged pool for its bases, scalars and a
             * few literal blocks (what does not fit is read from HBM; more pairs than the list holds: the items walk their bases).
             * Two tiles per workgroup when that keeps more tiles resident on a CU (8 workgroups by waves, 160 KB of LDS). */
            static const uint32_t pair_lim = getenv("HVQM4_AMD_PAIR_CAP") ? (uint32_t)std::max(1, atoi(getenv("HVQM4_AMD_PAIR_CAP"))) : 4096u;
            auto sized = [&](uint32_t t, uint32_t *cap, uint32_t *pairs, uint32_t *pool) -> uint32_t {
                static const uint32_t steps2[] = { 32, 64, 96, 128, 192, 256, 384, 512 }, steps1[] = { 32, 64, 96, 128, 192, 256 };
                const uint32_t want = std::min(256u * t, std::max(32u, t * mi));
                uint32_t ic = t == 2 ? 512u : 256u;
                for (uint32_t v : steps2) if (t == 2 && v >= want) { ic = v; break; }
                for (uint32_t v : steps1) if (t == 1 && v >= want) { ic = v; break; }
                *cap = ic;
                *pairs = std::max(1u, std::min(pair_lim, t * mp));
                /* HVQM4_AMD_POOL_CAP (tests): a smaller staging area, so that ordinary clips reach the read-from-HBM path of tiles
                 * whose payload exceeds it */
                static const uint32_t pool_lim = getenv("HVQM4_AMD_POOL_CAP") ? (uint32_t)std::max(0, atoi(getenv("HVQM4_AMD_POOL_CAP"))) : 1536u;
                *pool = std::min(pool_lim, t * (mp + 2u * mi + 128u));
                if (*pool) *pool = (*pool + 4u + 3u) & ~3u;      /* + 4: the staged range starts at the 16-byte boundary below the tile's first dword */
                return (hvq_recon_inline_static_lds(t, ic) + hvq_recon_inline_dyn_lds(*pairs, *pool) + 511u) & ~511u;
            };
            uint32_t cap1, pr1, po1, cap2, pr2, po2;
            const uint32_t lds1 = sized(1, &cap1, &pr1, &po1), lds2 = sized(2, &cap2, &pr2, &po2);
            const uint32_t res1 = std::min(8u, 163840u / lds1), res2 = 2u * std::min(8u, 163840u / lds2);
            /* measured (profiles/r04b_*): two tiles pay when they double the resident tiles (natural, flat), not for +25 % (dense) */
            const bool two = lds2 <= 65536u && (force_tpw ? force_tpw >= 2 : 2u * res2 >= 3u * res1);
            L.tpw = two ? 2u : 1u;
            L.items_cap = two ? cap2 : cap1; L.pair_cap = two ? pr2 : pr1; L.pool_cap = two ? po2 : po1;
        }
        (void)mp;
        L.max_tiles = L.max_wg[L.tpw - 1]; L.workgroups = L.wgs[L.tpw - 1];
        st.workgroups += L.workgroups;
    }
    c->launches = c->fl_launches;
    st.launches = (uint32_t)c->launches.size();
    st.launch_queues = (uint32_t)nqs;
    st.parse_seconds = c->parse_seconds;
    st.gpu_parse_ms = st.gpu_parsed ? c->gpu_parse_ms : 0.0;
    st.gpu_parse_retried = st.gpu_parsed ? c->gpu_parse_retried : 0u;
    st.dropped = n_dropped;
    if (jobs.size() > c->jobs_cap) {
        if (c->jobs_dev) { HIPCHK_FL(hipStreamSynchronize(c->stream)); HIPCHK_FL(hipFree(c->jobs_dev)); }
        c->jobs_cap = jobs.size() * 2;
        HIPCHK_FL(hipMalloc((void **)&c->jobs_dev, c->jobs_cap * sizeof(HvqJob)));
    }
    if (tq_bytes > c->tq_cap) {
        if (c->tq_dev) { HIPCHK_FL(hipStreamSynchronize(c->stream)); HIPCHK_FL(hipFree(c->tq_dev)); c->tq_dev = nullptr; c->tq_cap = 0; }
        const size_t ncap = align_up(tq_bytes + tq_bytes / 4, 4096);
        HIPCHK_FL(hipMalloc((void **)&c->tq_dev, ncap));
        c->tq_cap = ncap;
    }
    for (size_t k = 0; k < jobs.size(); ++k)
        if (jobs[k].total_tiles && has_tq[k]) jobs[k].tq = (uint64_t)(uintptr_t)(c->tq_dev + tq_off[k]);
    if (side_bytes > c->selfref_cap) {
        if (c->selfref_dev) { HIPCHK_FL(hipStreamSynchronize(c->stream)); HIPCHK_FL(hipFree(c->selfref_dev)); c->selfref_dev = nullptr; c->selfref_cap = 0; }
        HIPCHK_FL(hipMalloc((void **)&c->selfref_dev, side_bytes));
        c->selfref_cap = side_bytes;
    }
    for (const SelfRef &sr : c->selfrefs)
        for (int k = 0; k < 3; ++k) {
            HvqPlaneRec &r = jobs[sr.job].plane[k];
            r.dst = (uint64_t)(uintptr_t)(c->selfref_dev + sr.side_off) + r.plane_off;
        }
    /* stream-ordered after whatever still reads the previous table */
    { int rcu = staged_upload(c, c->fl_arena_id, 2, c->jobs_dev, jobs.data(), jobs.size() * sizeof(HvqJob)); if (rcu) return flush_abandon(c, rcu); }
    /* 3. one launch per level */
    {
        bool two = false;
        int rc = queues_fork(c, &two);
        if (!rc) rc = run_launches(c);
        /* joined also when a launch failed: whatever the other queues hold is ordered in front of everything the main stream gets next */
        { const int rcj = queues_join(c, two); if (!rc) rc = rcj; }
        if (rc) return flush_abandon(c, rc);
    }
    if (!c->fl_nest_pairs.empty()) {   /* the last I picture's nest of every GPU-parsed stream must outlive this batch's buffers */
        { int rcu = staged_upload(c, c->fl_arena_id, 3, PS(c).np_dev, c->fl_nest_pairs.data(), c->fl_nest_pairs.size() * sizeof(uint64_t)); if (rcu) return flush_abandon(c, rcu); }
        HIPCHK_FL(hvq_launch_nest_commit(PS(c).np_dev, (uint32_t)(c->fl_nest_pairs.size() / 2), c->stream));
    }
    HIPCHK_FL(hipEventRecord(c->ev_arena_free[c->fl_arena_id], c->stream));    /* this batch's arena may be refilled after this */
    HIPCHK_FL(hipEventRecord(c->ev_read, c->stream));                           /* every picture flushed so far is complete behind this */
    for (auto &s : c->streams) s.inflight_from = 0x7FFFFFFF;
    c->stats = st;
    c->fl_pending.clear();
    PS(c).fl_idx.clear();
    if (flush_timing()) fprintf(stderr, "flush_end   %.3f: parse results at %.3f, launches queued %.3f ms (parse kernel %.3f ms)\n", te0, te1, now_ms(), c->gpu_parse_ms);
    return first_rc;
}

HVQ_EXPORT int hvq_flush_end(HvqContext *c)
{
    if (!c) return fail(HVQ_E_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    return flush_end(c);
}

HVQ_EXPORT int hvq_flush(HvqContext *c)
{
    int rc = hvq_flush_begin(c);
    return rc ? rc : hvq_flush_end(c);
}

HVQ_EXPORT int hvq_sync(HvqContext *c)
{
    if (!c) return fail(HVQ_E_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    { int rcj = copy_join(c); if (rcj) return rcj; }
    { int rc = flush_end(c); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    return HVQ_OK;
}

/* `reps` passes over the resident batch.  joined = true: every pass forks and joins the launch queues exactly as a flush does
 * (flush_end: queues_fork, run_launches, queues_join) -- the product's step, pass r + 1 starts when BOTH queues have finished pass r.
 * joined = false: the queues fork once and join once around all passes (inside a queue pass r + 1 follows pass r in stream order, and
 * the queues hold different streams' pictures): a queue that is ahead runs into the next pass, which hides the queues' imbalance and
 * one join per step -- measured beside the joined form, never the headline.
 * (Round 5 measured a HIP graph of the pass for small batches -- one GPU's share of BASELINE config 4, 128 pictures in 7 launches --
 * 92.4 us per step against 87.2 with plain launches, profiles/r05_recon_steps.txt.  The launches are a dependency chain, each as long as
 * a workgroup's lifetime; there is no launch overhead for a graph to remove.  Not kept.) */
static int replay_passes(HvqContext *c, bool joined, int reps)
{
    bool two = false;
    if (!joined) { int rc = queues_fork(c, &two); if (rc) return rc; }
    for (int r = 0; r < reps; ++r) {
        if (joined) { int rc = queues_fork(c, &two); if (rc) return rc; }
        { int rc = run_launches(c); if (rc) return rc; }
        if (joined) { int rc = queues_join(c, two); if (rc) return rc; }
    }
    return joined ? HVQ_OK : queues_join(c, two);
}

HVQ_EXPORT int hvq_replay(HvqContext *c, int reps, float *gpu_ms)
{
    if (!c || reps < 0) return fail(HVQ_E_ARG, "bad arguments");
    { int rc = flush_end(c); if (rc) return rc; }
    if (c->launches.empty()) return fail(HVQ_E_STATE, "nothing flushed yet");
    if (!c->selfrefs.empty())
        return fail(HVQ_E_STATE, "the resident batch holds self-referencing P pictures: their previous buffer content is gone after the first pass");
    if (!c->pending.empty()) return fail(HVQ_E_STATE, "pictures queued since the last flush");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    { int rc = replay_passes(c, false, reps); if (rc) return rc; }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (gpu_ms) *gpu_ms = ms;
#ifdef HVQ_STAMPS
    if (getenv("HVQM4_AMD_STAMPS")) {
        /* diagnostic build: one more pass with phase stamps, per-launch mean segment lengths on stderr */
        static const char *seg_i[14] = { "job record (+ early barrier)", "trip 2 issue", "trip 2 lands", "classes, rows requested, scans", "slots, lists",
                                         "rows land", "phase A", "barrier 1", "pair phase (B1)", "barrier 2", "item phase (B2)", "barrier 3", "store issue", "stores land" };
        static const int from_i[14] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13 }, to_i[14] = { 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14 };
        for (auto &L : c->launches) {
            const char **seg = seg_i;
            const int *from = from_i, *to = to_i;
            const int NS = 14, LAST = 14;
            unsigned long long *d = nullptr;
            const size_t n = (size_t)L.ntiles * L.max_tiles * 64;
            HIPCHK(hipMalloc((void **)&d, n * 8));
            HIPCHK(hipMemsetAsync(d, 0, n * 8, c->stream));
            hvq_set_stamps(d);
            HIPCHK(hvq_launch_recon_inline(c->jobs_dev + L.first_tile, L.ntiles, L.max_tiles, L.tpw, L.items_cap, L.pair_cap, L.pool_cap, c->stream));
            hvq_set_stamps(nullptr);
            std::vector<unsigned long long> h(n);
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipFree(d));
            double sum[14] = {}, life = 0; size_t cnt[14] = {}, nw = 0;
            for (size_t t = 0; t < (size_t)L.ntiles * L.max_tiles; ++t)
                for (int w = 0; w < 4; ++w) {
                    const unsigned long long *q = &h[t * 64 + w * 16];
                    if (!q[LAST]) continue;
                    for (int k = 0; k < NS; ++k)
                        if (q[from[k]] && q[to[k]]) { sum[k] += (double)(q[to[k]] - q[from[k]]); cnt[k]++; }
                    life += (double)(q[LAST] - q[0]); nw++;
                }
            fprintf(stderr, "stamps L%d: %u workgroups, %zu waves, mean wave lifetime %.0f cycles;", L.level, L.workgroups, nw, nw ? life / nw : 0.0);
            for (int k = 0; k < NS; ++k) fprintf(stderr, " %s %.0f |", seg[k], cnt[k] ? sum[k] / cnt[k] : 0.0);
            fprintf(stderr, "\n");
        }
    }
#endif
    return HVQ_OK;
}

/* what = 1: `reps` passes, each forking and joining the launch queues as a flush does -- the reconstruction stage of the product,
 * everything a batch of new pictures costs behind its parse (bench.py's timed step).  what = 0: hvq_replay (the queues run free over
 * all passes).  what = 2: nothing (it timed the queue build of the two-pass variant, deleted in round 6; kept so that callers of
 * rounds 3-5 still link): 0 ms. */
HVQ_EXPORT int hvq_replay_stage(HvqContext *c, int reps, int what, float *gpu_ms)
{
    if (!c || reps < 0 || what < 0 || what > 2) return fail(HVQ_E_ARG, "bad arguments");
    if (what == 0) return hvq_replay(c, reps, gpu_ms);
    { int rc = flush_end(c); if (rc) return rc; }
    if (c->launches.empty()) return fail(HVQ_E_STATE, "nothing flushed yet");
    if (!c->selfrefs.empty())
        return fail(HVQ_E_STATE, "the resident batch holds self-referencing P pictures: their previous buffer content is gone after the first pass");
    if (!c->pending.empty()) return fail(HVQ_E_STATE, "pictures queued since the last flush");
    if (what == 2) { if (gpu_ms) *gpu_ms = 0.f; return HVQ_OK; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    { int rc = replay_passes(c, true, reps); if (rc) return rc; }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (gpu_ms) *gpu_ms = ms;
    return HVQ_OK;
}

HVQ_EXPORT int hvq_read_picture(HvqContext *c, int sid, int ordinal, void *dst, size_t cap)
{
    if (!c || sid < 0 || sid >= (int)c->streams.size() || !c->streams[sid].open) return fail(HVQ_E_ARG, "bad stream %d", sid);
    Stream &s = c->streams[sid];
    if (ordinal < 0 || ordinal >= s.npics) return fail(HVQ_E_ARG, "bad picture ordinal %d", ordinal);
    if (cap < s.pic_bytes) return fail(HVQ_E_ARG, "destination too small");
    { int rc = flush_end(c); if (rc) return rc; }
    for (auto &p : c->pending)
        if (p.stream == sid && p.ordinal == ordinal) return fail(HVQ_E_STATE, "picture %d is queued but not flushed", ordinal);
    int slot = s.pic_slot[(size_t)ordinal];
    if (slot < 0) return fail(HVQ_E_STATE, "picture %d is no longer resident (slot reused)", ordinal);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(dst, s.slot_ptr(slot), s.pic_bytes, hipMemcpyDeviceToHost));
    return HVQ_OK;
}

/* convert `n` resident pictures in one launch into consecutive regions of the RGB scratch; optional timing */
static int rgb_run(HvqContext *c, RgbJob *jobs, int n, int reps, float *gpu_ms)
{
    size_t need = 0;
    int max_lanes = 0, wide = 1;
    for (int i = 0; i < n; ++i) if (jobs[i].w % 16) wide = 0;
    for (int i = 0; i < n; ++i) { need += (size_t)jobs[i].w * jobs[i].h * 3; max_lanes = std::max(max_lanes, (jobs[i].w >> 2) * jobs[i].h); }
    if (need > c->rgb_cap) {
        if (c->rgb_dev) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->rgb_dev)); }
        HIPCHK(hipMalloc((void **)&c->rgb_dev, need));
        c->rgb_cap = need;
    }
    if ((size_t)n > c->rgb_jobs_cap) {
        if (c->rgb_jobs_dev) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->rgb_jobs_dev)); }
        HIPCHK(hipMalloc((void **)&c->rgb_jobs_dev, (size_t)n * sizeof(RgbJob)));
        c->rgb_jobs_cap = (size_t)n;
    }
    size_t off = 0;
    for (int i = 0; i < n; ++i) { jobs[i].rgb = c->rgb_dev + off; off += (size_t)jobs[i].w * jobs[i].h * 3; }
    HIPCHK(hipMemcpyAsync(c->rgb_jobs_dev, jobs, (size_t)n * sizeof(RgbJob), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (gpu_ms) HIPCHK(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; ++r) HIPCHK(hvq_launch_rgb(c->rgb_jobs_dev, n, max_lanes, wide, c->stream));
    if (gpu_ms) {
        HIPCHK(hipEventRecord(c->ev1, c->stream));
        HIPCHK(hipEventSynchronize(c->ev1));
        HIPCHK(hipEventElapsedTime(gpu_ms, c->ev0, c->ev1));
    }
    return HVQ_OK;
}

HVQ_EXPORT int hvq_read_picture_rgb(HvqContext *c, int sid, int ordinal, void *dst, size_t cap)
{
    if (!c || sid < 0 || sid >= (int)c->streams.size() || !c->streams[sid].open) return fail(HVQ_E_ARG, "bad stream %d", sid);
    Stream &s = c->streams[sid];
    if (ordinal < 0 || ordinal >= s.npics) return fail(HVQ_E_ARG, "bad picture ordinal %d", ordinal);
    const size_t need = (size_t)s.w * s.h * 3;
    if (cap < need) return fail(HVQ_E_ARG, "destination too small");
    if (s.pic_bytes != (uint32_t)(s.w * s.h * 3 / 2)) return fail(HVQ_E_GEOMETRY, "RGB epilogue needs 4:2:0");
    { int rc = flush_end(c); if (rc) return rc; }
    for (auto &p : c->pending)
        if (p.stream == sid && p.ordinal == ordinal) return fail(HVQ_E_STATE, "picture %d is queued but not flushed", ordinal);
    int slot = s.pic_slot[(size_t)ordinal];
    if (slot < 0) return fail(HVQ_E_STATE, "picture %d is no longer resident (slot reused)", ordinal);
    HIPCHK(hipSetDevice(c->device));
    RgbJob job{ s.slot_ptr(slot), nullptr, s.w, s.h };
    int rc = rgb_run(c, &job, 1, 1, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dst, c->rgb_dev, need, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return HVQ_OK;
}

/* the reference player's dumpRGB (h4m:897-926) on a host picture: Y|U|V 4:2:0 in, RGB24 out */
HVQ_EXPORT int hvq_convert_yuv420_rgb(HvqContext *c, const void *yuv, int width, int height, void *rgb)
{
    if (!c || !yuv || !rgb || width <= 0 || height <= 0 || (width & 3) || (height & 1) || width > 16384 || height > 16384)
        return fail(HVQ_E_ARG, "bad arguments (width a multiple of 4, height even)");
    HIPCHK(hipSetDevice(c->device));
    { int rc = flush_end(c); if (rc) return rc; }
    const size_t nyuv = (size_t)width * height * 3 / 2, nrgb = (size_t)width * height * 3;
    uint8_t *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, nyuv));
    hipError_t e = hipMemcpyAsync(d, yuv, nyuv, hipMemcpyHostToDevice, c->stream);
    int rc = HVQ_OK;
    if (e != hipSuccess) rc = fail(HVQ_E_HIP, "upload: %s", hipGetErrorString(e));
    RgbJob job{ d, nullptr, width, height };
    if (!rc) rc = rgb_run(c, &job, 1, 1, nullptr);
    if (!rc) {
        e = hipMemcpyAsync(rgb, c->rgb_dev, nrgb, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(HVQ_E_HIP, "download: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    return rc;
}

/* ---- getting pictures out at the rate they are made (the per-picture hvq_read_picture synchronises every time) ---- */
HVQ_EXPORT void *hvq_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(HVQ_E_HIP, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}

HVQ_EXPORT void hvq_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

/* resident picture -> its slot, after ending the batch in flight; nullptr + error when it is queued, unknown or gone */
static const uint8_t *resident_picture(HvqContext *c, int sid, int ordinal, int *rc)
{
    *rc = HVQ_OK;
    if (sid < 0 || sid >= (int)c->streams.size() || !c->streams[(size_t)sid].open) { *rc = fail(HVQ_E_ARG, "bad stream %d", sid); return nullptr; }
    Stream &s = c->streams[(size_t)sid];
    if (ordinal < 0 || ordinal >= s.npics) { *rc = fail(HVQ_E_ARG, "stream %d: bad picture ordinal %d", sid, ordinal); return nullptr; }
    for (auto &p : c->pending)
        if (p.stream == sid && p.ordinal == ordinal) { *rc = fail(HVQ_E_STATE, "stream %d picture %d is queued but not flushed", sid, ordinal); return nullptr; }
    const int slot = s.pic_slot[(size_t)ordinal];
    if (slot < 0) { *rc = fail(HVQ_E_STATE, "stream %d picture %d is no longer resident (slot reused, or the picture was dropped)", sid, ordinal); return nullptr; }
    return s.slot_ptr(slot);
}

HVQ_EXPORT int hvq_read_pictures(HvqContext *c, int n, const int *streams, const int *ordinals, void *const *dst)
{
    if (!c || n < 0 || (n && (!streams || !ordinals || !dst))) return fail(HVQ_E_ARG, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    /* Pictures of batches that hvq_flush_end has launched are read WITHOUT ending the batch in flight: a streaming player calls
     * flush_end(k), flush_begin(k + 1), read(k) -- the copies of batch k then run beside the parse of batch k + 1.  Only when a
     * requested picture belongs to the batch in flight is that batch ended first. */
    bool need_end = false;
    for (int i = 0; i < n && !need_end; ++i)
        if (streams[i] >= 0 && streams[i] < (int)c->streams.size() && ordinals[i] >= c->streams[(size_t)streams[i]].inflight_from) need_end = true;
    if (need_end) { int rc = flush_end(c); if (rc) return rc; }
    /* all copies are queued on the read stream behind the launches of the last ended batch (the event flush_end recorded); ONE
     * wait at the end */
    HIPCHK(hipStreamWaitEvent(c->read_stream, c->ev_read, 0));
    std::vector<const uint8_t *> src((size_t)n);
    bool same = n > 0;
    for (int i = 0; i < n; ++i) {
        int rc = HVQ_OK;
        src[(size_t)i] = resident_picture(c, streams[i], ordinals[i], &rc);
        if (!src[(size_t)i]) return rc;
        if (!dst[i]) return fail(HVQ_E_ARG, "null destination %d", i);
        same = same && c->streams[(size_t)streams[i]].pic_bytes == c->streams[(size_t)streams[0]].pic_bytes;
    }
    if (same && n >= 4) {
        /* every picture lives in its own slot: a kernel gathers them into one buffer (HBM speed), and what crosses PCIe is one
         * transfer per run of consecutive destinations instead of one per picture (2048 x 460 KB: 20 GB/s apiece, ~50 together) */
        const uint32_t pb = c->streams[(size_t)streams[0]].pic_bytes;
        const size_t need = (size_t)n * pb;
        if (need > c->rb_cap) {
            if (c->rb_dev) { HIPCHK(hipStreamSynchronize(c->read_stream)); HIPCHK(hipFree(c->rb_dev)); c->rb_dev = nullptr; c->rb_cap = 0; }
            HIPCHK(hipMalloc((void **)&c->rb_dev, need));
            c->rb_cap = need;
        }
        if ((size_t)n > c->rb_tab_cap) {
            HIPCHK(hipStreamSynchronize(c->read_stream));
            if (c->rb_tab_dev) HIPCHK(hipFree(c->rb_tab_dev));
            if (c->rb_tab_host) HIPCHK(hipHostFree(c->rb_tab_host));
            c->rb_tab_dev = nullptr; c->rb_tab_host = nullptr;
            c->rb_tab_cap = (size_t)n * 2;
            HIPCHK(hipMalloc((void **)&c->rb_tab_dev, c->rb_tab_cap * sizeof(uint64_t)));
            HIPCHK(hipHostMalloc((void **)&c->rb_tab_host, c->rb_tab_cap * sizeof(uint64_t), hipHostMallocDefault));
        }
        for (int i = 0; i < n; ++i) c->rb_tab_host[i] = (uint64_t)(uintptr_t)src[(size_t)i];
        HIPCHK(hipMemcpyAsync(c->rb_tab_dev, c->rb_tab_host, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, c->read_stream));
        HIPCHK(hvq_launch_gather(c->rb_tab_dev, c->rb_dev, (uint32_t)n, pb, c->read_stream));
        for (int i = 0; i < n;) {
            int j = i + 1;
            while (j < n && (uint8_t *)dst[j] == (uint8_t *)dst[j - 1] + pb) ++j;
            HIPCHK(hipMemcpyAsync(dst[i], c->rb_dev + (size_t)i * pb, (size_t)(j - i) * pb, hipMemcpyDeviceToHost, c->read_stream));
            i = j;
        }
    } else {
        for (int i = 0; i < n; ++i)
            HIPCHK(hipMemcpyAsync(dst[i], src[(size_t)i], c->streams[(size_t)streams[i]].pic_bytes, hipMemcpyDeviceToHost, c->read_stream));
    }
    HIPCHK(hipStreamSynchronize(c->read_stream));
    return HVQ_OK;
}

/* device address of a resident picture (Y|U|V, hvq_stream_pic_bytes) for consumers on the GPU; valid until the stream's ring
 * hands the slot to a later picture (nslots pictures later at the earliest).  Work queued by the caller must be ordered after
 * hvq_sync(), or after the event the caller records behind it. */
HVQ_EXPORT int hvq_picture_device_ptr(HvqContext *c, int sid, int ordinal, const void **ptr)
{
    if (!c || !ptr) return fail(HVQ_E_ARG, "bad arguments");
    { int rc = flush_end(c); if (rc) return rc; }
    int rc = HVQ_OK;
    *ptr = resident_picture(c, sid, ordinal, &rc);
    return rc;
}

HVQ_EXPORT int hvq_rgb_bench(HvqContext *c, int reps, float *gpu_ms, uint64_t *bytes_per_rep, uint32_t *pictures)
{
    if (!c || reps < 1) return fail(HVQ_E_ARG, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    { int rc = flush_end(c); if (rc) return rc; }
    std::vector<RgbJob> jobs;
    uint64_t bytes = 0;
    for (auto &s : c->streams) {
        if (!s.open || s.npics == 0 || s.pic_bytes != (uint32_t)(s.w * s.h * 3 / 2)) continue;
        int slot = s.pic_slot[(size_t)s.npics - 1];
        if (slot < 0) continue;
        jobs.push_back(RgbJob{ s.slot_ptr(slot), nullptr, s.w, s.h });
        bytes += (uint64_t)s.w * s.h * 9 / 2;              /* 1.5 B/px read + 3 B/px written */
    }
    if (jobs.empty()) return fail(HVQ_E_STATE, "no resident 4:2:0 picture");
    int rc = rgb_run(c, jobs.data(), (int)jobs.size(), reps, gpu_ms);
    if (rc) return rc;
    if (bytes_per_rep) *bytes_per_rep = bytes;
    if (pictures) *pictures = (uint32_t)jobs.size();
    return HVQ_OK;
}

/* Measurement helper: `reps` pinned-host -> device copies of `bytes` on the context's copy stream, timed with HIP events: the PCIe
 * rate a batch's bitstream upload can reach on this box (the bound of streaming from host memory, whatever the kernels do). */
HVQ_EXPORT int hvq_h2d_probe(HvqContext *c, size_t bytes, int reps, double *gb_per_s)
{
    if (!c || !bytes || reps < 1 || !gb_per_s) return fail(HVQ_E_ARG, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    void *h = nullptr, *d = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&d, bytes);
    if (e == hipSuccess) { memset(h, 0x5a, bytes); e = hipEventCreate(&e0); }
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->copy_stream);   /* warm-up: mappings, clocks */
    if (e == hipSuccess) e = hipEventRecord(e0, c->copy_stream);
    for (int r = 0; r < reps && e == hipSuccess; ++r) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(e1, c->copy_stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    HIPCHK(e);
    *gb_per_s = ms > 0 ? (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
    return HVQ_OK;
}

/* self-test: out[0..15] = the kernels' divTable quotients 256 / d, out[16..271] = their mcdivTable quotients 4096 / d (h4m:265-273) */
HVQ_EXPORT int hvq_debug_table_divisions(HvqContext *c, uint32_t *out)
{
    if (!c || !out) return fail(HVQ_E_ARG, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    uint32_t *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, 272 * sizeof(uint32_t)));
    hipError_t e = hvq_launch_table_div(d, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d, 272 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPCHK(e);
    return HVQ_OK;
}

HVQ_EXPORT int hvq_get_stats(HvqContext *c, HvqStats *out)
{
    if (!c || !out) return fail(HVQ_E_ARG, "bad arguments");
    { int rc = flush_end(c); if (rc) return rc; }
    *out = c->stats;
    out->parse_seconds = c->parse_seconds;
    out->copy_bytes = c->copy_bytes.load(); out->copy_seconds = (double)c->copy_ns.load() * 1e-9;
    return HVQ_OK;
}

/* ------------------------------------------------------------------ SDK entry points */
namespace {

constexpr uint64_t SDK_MAGIC = 0x4856514d34414d44ull;   /* "HVQM4AMD" */

struct SdkBinding {
    std::mutex mu;                     /* one decode at a time per SeqObj; different SeqObjs decode concurrently (each has its own context) */
    HvqContext *ctx = nullptr;
    int stream = -1;
    int w = 0, h = 0, hs = 0, vs = 0, is15 = -1;
    uint32_t pic_bytes = 0;
    uint32_t max_frame = 0;            /* HVQM4SetMaxFrameSize: readable bytes at `frame` (0 = unknown) */
    /* what the three device slots hold: the host buffer whose content was last written there by this library */
    const void *host[3] = { nullptr, nullptr, nullptr };
    bool valid[3] = { false, false, false };
    /* pinned staging, one picture each: [0], [1] reference pictures on their way up (their DMA runs beside the host parse),
     * [2] the decoded picture on its way down (one DMA, then one memcpy into the caller's pageable buffer) */
    uint8_t *stage[3] = { nullptr, nullptr, nullptr };
};

struct SdkHeader {          /* lives at the start of the caller's work buffer */
    uint64_t magic;
    SdkBinding *binding;
};

std::mutex g_sdk_mu;                    /* the registry of bindings */
std::mutex g_sdk_ctx_mu;                /* the probe context (its own lock: it is taken by a decode that holds its binding's lock) */
HvqContext *g_sdk_ctx = nullptr;
std::set<SdkBinding *> g_bindings;

void sdk_fail(int code)
{
    g_sdk_err = code;
    fprintf(stderr, "hvqm4_amd: %s\n", g_err.c_str());
}

int sdk_device()
{
    const char *dev = getenv("HVQM4_AMD_DEVICE");
    return dev ? atoi(dev) : 0;
}

HvqContext *sdk_context()              /* HVQM4InitDecoder: a missing GPU is reported at init time */
{
    if (g_sdk_ctx) return g_sdk_ctx;
    int rc = hvq_context_create(sdk_device(), &g_sdk_ctx);
    if (rc) { sdk_fail(rc); g_sdk_ctx = nullptr; }
    return g_sdk_ctx;
}

void sdk_free_binding(SdkBinding *b)
{
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        (void)hvq_sync(b->ctx);
        for (auto &p : b->stage) if (p) { (void)hipHostFree(p); p = nullptr; }
        hvq_context_destroy(b->ctx);            /* closes its stream */
    }
    delete b;
}

void sdk_release_locked(SdkHeader *hd)
{
    if (hd->magic == SDK_MAGIC && g_bindings.count(hd->binding)) {
        SdkBinding *b = hd->binding;
        g_bindings.erase(b);
        { std::lock_guard<std::mutex> lk(b->mu); }      /* a decode in flight on another thread ends first */
        sdk_free_binding(b);
    }
    hd->magic = 0; hd->binding = nullptr;
}

/* The SeqObj's binding with ITS lock held, or an unlocked lock + error.  The binding's lock is taken while the registry lock is still
 * held (hand-over): a HVQM4SetBuffer / HVQM4ReleaseBuffer of the same SeqObj on another thread can then no longer free the binding
 * between the look-up and the decode's own lock (advisor finding of round 4: it could).  try_lock + retry, so that a second thread
 * decoding on the SAME SeqObj (it waits) never blocks the registry for the players of other SeqObjs; nothing that holds a binding's
 * lock waits for the registry lock (the probe context has a lock of its own), so the order registry -> binding cannot deadlock. */
std::unique_lock<std::mutex> sdk_lookup(SeqObj *seq, SdkBinding **out)
{
    *out = nullptr;
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(g_sdk_mu);
            if (!seq || !seq->state) { fail(HVQ_E_ARG, "SeqObj has no work buffer (call HVQM4SetBuffer)"); sdk_fail(HVQ_E_ARG); return {}; }
            SdkHeader *hd = (SdkHeader *)seq->state;
            if (hd->magic != SDK_MAGIC || !g_bindings.count(hd->binding)) { fail(HVQ_E_STATE, "work buffer was not initialised by HVQM4SetBuffer"); sdk_fail(HVQ_E_STATE); return {}; }
            std::unique_lock<std::mutex> bl(hd->binding->mu, std::try_to_lock);
            if (bl.owns_lock()) { *out = hd->binding; return bl; }
        }
        std::this_thread::yield();
    }
}

/* (re)open the device stream when the 1.3/1.5 switch byte changed (h4m:2414-2417); called with the binding's lock held */
bool sdk_open(SdkBinding *b, SeqObj *seq)
{
    const int is15 = seq->state->padding[0] != 0;
    if (b->stream >= 0 && b->is15 != is15) { hvq_stream_close(b->ctx, b->stream); b->stream = -1; }
    if (b->stream < 0) {
        if (!b->ctx) {
            {   /* the context HVQM4InitDecoder probed the device with serves the first SeqObj */
                std::lock_guard<std::mutex> lk(g_sdk_ctx_mu);
                b->ctx = g_sdk_ctx; g_sdk_ctx = nullptr;
            }
            if (!b->ctx) {
                int rc = hvq_context_create(sdk_device(), &b->ctx);
                if (rc) { b->ctx = nullptr; sdk_fail(rc); return false; }
            }
        }
        int sid = hvq_stream_open(b->ctx, b->w, b->h, b->hs, b->vs, is15, 3);
        if (sid < 0) { sdk_fail(sid); return false; }
        b->stream = sid; b->is15 = is15;
        b->pic_bytes = hvq_stream_pic_bytes(b->ctx, sid);
        {   /* One synchronous picture at a time: its sections are parsed side by side by a small pool of the SeqObj's parser (78 % of a
             * call was the parse on ONE core).  HVQM4_AMD_SDK_PARSE_THREADS (default 4, capped by the cores this process may run on;
             * 1 = the calling thread alone). */
            static const int want = getenv("HVQM4_AMD_SDK_PARSE_THREADS") ? atoi(getenv("HVQM4_AMD_SDK_PARSE_THREADS")) : 4;
            const int cores = (int)std::max(1u, std::thread::hardware_concurrency());
            (void)hvq_parser_set_threads(b->ctx->streams[(size_t)sid].parser, std::max(1, std::min(want, cores)));
        }
        for (auto &p : b->stage)
            if (!p && hipHostMalloc((void **)&p, b->pic_bytes, hipHostMallocDefault) != hipSuccess) p = nullptr;   /* without staging: plain copies */
        for (int i = 0; i < 3; ++i) b->valid[i] = false;
    }
    return true;
}

/* One synchronous picture: upload the caller's reference pictures, reconstruct, read back.  ONE wait on the GPU per call.
 * HVQM4_AMD_TRUST_PICTURES=1: a reference picture is not uploaded again when `past` / `future` is a host buffer this
 * library itself filled last (as `present` of an earlier call on the same SeqObj) and its device copy is still in place --
 * valid for players that do not touch decoded pictures, which the SDK contract does not promise (hence opt-in). */
void sdk_decode(SeqObj *seq, int ftype, const uint8_t *frame, void *present, const void *past, const void *future)
{
    SdkBinding *b = nullptr;
    std::unique_lock<std::mutex> lk = sdk_lookup(seq, &b);
    if (!b) return;
    /* HVQM4_AMD_SDK_TIMING=1: where a call's time goes, on stderr (development aid) */
    static const bool timing = getenv("HVQM4_AMD_SDK_TIMING") != nullptr;
    double tm[6] = { 0, 0, 0, 0, 0, 0 };
    if (timing) tm[0] = now_ms();
    if (!sdk_open(b, seq)) return;
    HvqContext *c = b->ctx;
    Stream &s = c->streams[(size_t)b->stream];
    /* the SDK signatures carry no length: it comes from the picture's own section table (hvq_picture_length), so that the
     * host parser bounds every later read */
    size_t len = 0;
    { int rc = hvq_picture_length(frame, ftype, b->max_frame, &len);
      if (rc) { fail(rc, "malformed picture: a section lies outside the frame"); sdk_fail(rc); return; } }
    static const bool trust = getenv("HVQM4_AMD_TRUST_PICTURES") && atoi(getenv("HVQM4_AMD_TRUST_PICTURES")) > 0;
    int used[2] = { -1, -1 };
    /* a call that fails half way leaves uploads in flight and slots half written: nothing on the device is trusted afterwards */
    struct Undo { SdkBinding *b; HvqContext *c; bool ok; ~Undo() { if (!ok) { (void)hipStreamSynchronize(c->stream); for (auto &v : b->valid) v = false; } } } undo{ b, c, false };
    /* device slot that holds `src`: the resident copy if trusted, else a slot not already taken, refreshed from the host.  The
     * upload goes through pinned staging and is asynchronous: it runs while the host parses the picture */
    auto bring = [&](const void *src, int k) -> int {
        int slot = -1;
        if (trust)
            for (int i = 0; i < 3; ++i) if (b->valid[i] && b->host[i] == src && i != used[0]) slot = i;
        if (slot < 0) {
            for (int i = 0; i < 3 && slot < 0; ++i) if (i != used[0] && !(b->valid[i] && trust && (b->host[i] == past || b->host[i] == future))) slot = i;
            if (slot < 0) slot = used[0] == 0 ? 1 : 0;
            const void *from = src;
            if (b->stage[k]) { memcpy(b->stage[k], src, s.pic_bytes); from = b->stage[k]; }
            hipError_t e = hipMemcpyAsync(s.slot_ptr(slot), from, s.pic_bytes, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) { fail(HVQ_E_HIP, "upload of reference picture: %s", hipGetErrorString(e)); sdk_fail(HVQ_E_HIP); return -1; }
            b->host[slot] = src; b->valid[slot] = true;
        }
        used[k] = slot;
        return slot;
    };
    s.anchor_old = -1; s.anchor_new = -1;
    if (ftype == HVQ_FRAME_P) {
        const int ps = bring(past, 0);
        if (ps < 0) return;
        s.anchor_new = ps;                               /* swapped to "past" by the submit's rotation */
    } else if (ftype == HVQ_FRAME_B) {
        const int ps = bring(past, 0);
        if (ps < 0) return;
        const int fs = bring(future, 1);
        if (fs < 0) return;
        s.anchor_old = ps; s.anchor_new = fs;
    }
    int dst = -1;
    for (int i = 0; i < 3; ++i) if (i != used[0] && i != used[1] && (dst < 0 || b->host[i] == present)) dst = i;
    s.ring = dst;                                        /* alloc_slot takes the first slot from here that is no anchor */
    b->valid[dst] = false;
    s.sdk_present = ftype == HVQ_FRAME_P ? present : nullptr;   /* a self-referencing P picture reads the caller's buffer (h4m:2058-2061) */
    if (timing) tm[1] = now_ms();
    int ord = hvq_stream_submit(c, b->stream, ftype, frame, len);
    c->streams[(size_t)b->stream].sdk_present = nullptr;
    if (ord < 0) { sdk_fail(ord); return; }
    if (timing) tm[2] = now_ms();
    int rc = hvq_flush(c);
    if (rc) { sdk_fail(rc); return; }
    if (timing) tm[3] = now_ms();
    if (b->stage[2]) {
        /* the picture's DMA is queued behind its launches: one wait for both, then one memcpy into the caller's buffer */
        const int slot = s.pic_slot[(size_t)ord];
        hipError_t e = slot < 0 ? hipErrorInvalidValue : hipMemcpyAsync(b->stage[2], s.slot_ptr(slot), s.pic_bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { fail(HVQ_E_HIP, "download of the decoded picture: %s", hipGetErrorString(e)); sdk_fail(HVQ_E_HIP); return; }
        memcpy(present, b->stage[2], s.pic_bytes);
    } else {
        rc = hvq_read_picture(c, b->stream, ord, present, s.pic_bytes);
        if (rc) { sdk_fail(rc); return; }
    }
    b->host[dst] = present; b->valid[dst] = true;        /* host and device copies are the same now */
    undo.ok = true;
    if (timing) {
        tm[4] = now_ms();
        fprintf(stderr, "sdk_decode 0x%x: bind + reference uploads %.3f | host parse %.3f | flush (uploads, launches) %.3f | wait + download %.3f | total %.3f ms\n",
                ftype, tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[4] - tm[0]);
    }
}

}  // namespace

HVQ_EXPORT void HVQM4InitDecoder(void)
{
    /* the reference fills divTable/mcdivTable here (h4m:265-278); ours are compile-time constants
     * in the kernel.  Probe the device early so a missing GPU is reported at init time. */
    std::lock_guard<std::mutex> lk(g_sdk_ctx_mu);
    (void)sdk_context();
}

HVQ_EXPORT void HVQM4InitSeqObj(SeqObj *seqobj, VideoInfo *videoinfo)
{
    seqobj->width = videoinfo->hres;
    seqobj->height = videoinfo->vres;
    seqobj->h_samp = videoinfo->h_samp;
    seqobj->v_samp = videoinfo->v_samp;
}

HVQ_EXPORT uint32_t HVQM4BuffSize(SeqObj *seqobj)
{
    /* same arithmetic as h4m:828-840 so callers allocate what they always did */
    uint32_t hb = seqobj->width / 4, vb = seqobj->height / 4;
    uint32_t yb = (hb + 2) * (vb + 2);
    uint32_t uhb = seqobj->h_samp == 2 ? hb / 2 : hb, uvb = seqobj->v_samp == 2 ? vb / 2 : vb;
    uint32_t uvblocks = (uhb + 2) * (uvb + 2);
    return (uint32_t)sizeof(VideoState) + (yb + uvblocks * 2) * (uint32_t)sizeof(uint16_t);
}

HVQ_EXPORT void HVQM4SetBuffer(SeqObj *seqobj, void *workbuff)
{
    std::lock_guard<std::mutex> lk(g_sdk_mu);
    SdkHeader *hd = (SdkHeader *)workbuff;
    sdk_release_locked(hd);                 /* re-binding an already bound buffer must not leak */
    seqobj->state = (VideoState *)workbuff;
    SdkBinding *b = new SdkBinding();
    b->w = seqobj->width; b->h = seqobj->height; b->hs = seqobj->h_samp; b->vs = seqobj->v_samp;
    g_bindings.insert(b);
    hd->magic = SDK_MAGIC;
    hd->binding = b;
}

HVQ_EXPORT void HVQM4ReleaseBuffer(SeqObj *seqobj)
{
    std::lock_guard<std::mutex> lk(g_sdk_mu);
    if (seqobj && seqobj->state) sdk_release_locked((SdkHeader *)seqobj->state);
}

HVQ_EXPORT void HVQM4SetMaxFrameSize(SeqObj *seqobj, uint32_t bytes)
{
    std::lock_guard<std::mutex> lk(g_sdk_mu);
    if (!seqobj || !seqobj->state) return;
    SdkHeader *hd = (SdkHeader *)seqobj->state;
    if (hd->magic == SDK_MAGIC && g_bindings.count(hd->binding)) hd->binding->max_frame = bytes;
}

HVQ_EXPORT void HVQM4SetVersion15(SeqObj *seqobj, int is15)
{
    if (seqobj && seqobj->state) seqobj->state->padding[0] = is15 ? 1 : 0;
}

HVQ_EXPORT void HVQM4DecodeIpic(SeqObj *seqobj, uint8_t const *frame, void *present)
{
    sdk_decode(seqobj, HVQ_FRAME_I, frame, present, nullptr, nullptr);
}

HVQ_EXPORT void HVQM4DecodePpic(SeqObj *seqobj, uint8_t const *frame, void *present, void *past)
{
    sdk_decode(seqobj, HVQ_FRAME_P, frame, present, past, nullptr);
}

HVQ_EXPORT void HVQM4DecodeBpic(SeqObj *seqobj, uint8_t const *frame, void *present, void *past, void *future)
{
    sdk_decode(seqobj, HVQ_FRAME_B, frame, present, past, future);
}

HVQ_EXPORT int HVQM4GetLastError(void)
{
    int e = g_sdk_err;
    g_sdk_err = 0;
    return e;
}

HVQ_EXPORT const char *HVQM4GetLastErrorString(void) { return g_err.c_str(); }
