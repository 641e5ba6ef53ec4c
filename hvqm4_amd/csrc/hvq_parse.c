/*
 * hvq_parse.c -- host entropy parse (see hvq_parse.h).  From-scratch; each stage cites the
 * reference lines (h4m: = h4m_audio_decode.c) whose bitstream semantics it follows.
 *
 * Differences from the reference that matter for speed, none for results:
 *   - 64-bit bit reservoir and a 10-bit first-level lookup table per prefix tree instead of a
 *     bit-at-a-time tree walk (the walk is ~35 % of the reference's CPU time);
 *   - pixel-independent: AOT coefficient sums, word offsets, the two MC-residual scalars and
 *     absolute motion-vector targets are resolved here, everything pixel-dependent
 *     (divTable/mcdivTable lookups, min/max, means) is left to the GPU;
 *   - reentrant: all state lives in the HvqParser (the reference keeps readTree_signed /
 *     readTree_scale and its tables in globals, h4m:262-263, 604-605).
 */
#include "hvq_parse.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdatomic.h>
#include <time.h>

#define LUT_BITS 10
/* caps of the overflow-symbol loops, identical in hvq_gparse_core.h (see there) */
#define SOVF_CAP 4096
#define UOVF_CAP(nmb) ((int)((nmb) / 255u) + 16)

/* ------------------------------------------------------------------ bit reader */
typedef struct {
    const uint8_t *p;
    const uint8_t *end;
    uint64_t acc;       /* left-aligned */
    int cnt;
    int live;           /* section present (size != 0, h4m:1061-1071) */
} BitRd;

static inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static inline uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

/* Top up the reservoir to >= 56 valid bits.  Fast path: one unaligned big-endian 64-bit load OR-ed below the valid
 * bits; only whole bytes are accounted, the few extra bits that come along are the stream's own next bits and are
 * OR-ed again (idempotent) by the next refill.  Within 8 bytes of the end: byte-wise with zero fill. */
static inline void br_refill(BitRd *b)
{
    if (b->cnt > 32) return;
    if (__builtin_expect(b->p + 8 <= b->end, 1)) {
        uint64_t w;
        memcpy(&w, b->p, 8);
        w = __builtin_bswap64(w);
        b->acc |= w >> b->cnt;
        const int adv = (63 - b->cnt) >> 3;
        b->p += adv;
        b->cnt += adv * 8;
    } else {
        while (b->cnt <= 56) {
            const uint64_t byte = b->p < b->end ? *b->p : 0u;
            b->acc |= byte << (56 - b->cnt);
            b->p++;
            b->cnt += 8;
        }
    }
}

static inline uint32_t br_take(BitRd *b, int n)      /* n <= 25 */
{
    if (n == 0) return 0;
    br_refill(b);
    uint32_t v = (uint32_t)(b->acc >> (64 - n));
    b->acc <<= n;
    b->cnt -= n;
    return v;
}

/* ------------------------------------------------------------------ prefix trees (h4m:385-394, 604-651) */
typedef struct {
    int root, next;
    int16_t kid[2][512];
    int32_t leaf[256];
    uint32_t lut[1 << LUT_BITS];        /* [31:26] bits consumed; bit 25 set: [15:0] the leaf's VALUE (one look-up per symbol);
                                           bit 25 clear: [24:16] inner node the walk goes on from (codes longer than the table) */
} Code;
#define LUT_LEAF (1u << 25)

static int code_node(Code *c, BitRd *b, int is_signed, int scale, int depth)
{
    if (depth > 600) return 0;                       /* malformed: unbounded recursion guard */
    if (br_take(b, 1) == 0) {
        int byte = (int)br_take(b, 8);
        int v = (is_signed && byte > 0x7F) ? byte - 256 : byte;
        c->leaf[byte] = (int16_t)((uint32_t)v << scale);          /* int16 truncation: h4m:613-617 */
        return byte;
    }
    /* more than 255 inner nodes cannot come from 256 leaf bytes: malformed; stop before node 511 can become its
     * own child (an endless walk in sym) */
    if (c->next >= 511) return 0;
    int id = c->next++;
    c->kid[0][id] = (int16_t)code_node(c, b, is_signed, scale, depth + 1);
    c->kid[1][id] = (int16_t)code_node(c, b, is_signed, scale, depth + 1);
    return id;
}

static void code_lut(Code *c, int node, int depth, uint32_t prefix)
{
    if (node < 256 || depth == LUT_BITS) {
        uint32_t lo = prefix << (LUT_BITS - depth), n = 1u << (LUT_BITS - depth);
        const uint32_t e = ((uint32_t)depth << 26) | (node < 256 ? LUT_LEAF | (uint32_t)(uint16_t)c->leaf[node] : (uint32_t)node << 16);
        for (uint32_t i = 0; i < n; ++i) c->lut[lo + i] = e;
        return;
    }
    code_lut(c, c->kid[0][node], depth + 1, prefix << 1);
    code_lut(c, c->kid[1][node], depth + 1, (prefix << 1) | 1);
}

static void code_read(Code *c, BitRd *carrier, int is_signed, int scale)     /* h4m:632-642 */
{
    c->next = 0x100;
    c->root = carrier->live ? code_node(c, carrier, is_signed, scale, 0) : 0;
    code_lut(c, c->root, 0, 0);
}

static inline int32_t sym(const Code *c, BitRd *b)                            /* h4m:644-651 */
{
    br_refill(b);
    const uint32_t e = c->lut[b->acc >> (64 - LUT_BITS)];
    const int len = (int)(e >> 26);
    b->acc <<= len;
    b->cnt -= len;
    if (__builtin_expect(e & LUT_LEAF, 1)) return (int32_t)(int16_t)e;     /* leaf values are int16 (h4m:613-617) */
    int id = (int)((e >> 16) & 511u);
    while (id >= 256) {
        if (b->cnt == 0) br_refill(b);
        id = c->kid[b->acc >> 63][id];
        b->acc <<= 1;
        b->cnt--;
    }
    return c->leaf[id];
}

/* The reference sums overflow symbols for as long as the stream says (h4m:654-664).  The fast loop stops at SOVF_CAP symbols -- the
 * cap the GPU parser shares; a run that long is not something an encoder writes -- and a run that is STILL open then is followed to
 * its end by the slow loop below (round 5; rounds 3-4 refused such a picture): every symbol of a tree with two leaves or more
 * consumes a bit, so the loop is bounded by the bits left in the picture, behind which the reader delivers zeros -- the reference
 * would be reading foreign memory there.  `*flags` gets HVQ_F_CAPPED only when even that does not end the run (a one-leaf tree whose
 * value lies outside the window: the reference never returns); the picture is refused then, and the rest of its parse returns at
 * once (the values no longer matter, and a full-size picture of such reads must not take seconds). */
static int32_t sym_sovf(const Code *c, BitRd *b, int32_t lo, int32_t hi, uint32_t *flags)       /* h4m:654-664 */
{
    uint32_t total = 0;
    int32_t v;
    int guard = 0;
    if (__builtin_expect(*flags & HVQ_F_CAPPED, 0)) return 0;
    do { v = sym(c, b); total += (uint32_t)v; } while ((v <= lo || v >= hi) && ++guard < SOVF_CAP);
    if (__builtin_expect(v <= lo || v >= hi, 0)) {
        if (b->end != (const uint8_t *)UINTPTR_MAX) {            /* picture length known: the walk is bounded */
            const uint8_t *at = b->p - (b->cnt >> 3);
            size_t left = at < b->end ? (size_t)(b->end - at) * 8u + 64u : 64u;
            while ((v <= lo || v >= hi) && left--) { v = sym(c, b); total += (uint32_t)v; }
        }
        if (v <= lo || v >= hi) *flags |= HVQ_F_CAPPED;
    }
    return (int32_t)total;
}

/* Run lengths of the macroblock type / proc runs.  A loop that ends on its cap is EXACT here and raises no flag (round 4; it did,
 * and refused a picture the reference decodes: one clip in 14 000 of the randomized sweep writes "to the end of the picture" as a
 * chain of seventeen 0xFF symbols): the capped total, cap x 255 with cap = UOVF_CAP(macroblocks) = macroblocks / 255 + 16, already
 * exceeds the picture's macroblocks, so the run covers everything that is left exactly as the longer sum would, and nothing reads
 * the section again behind a run that ends the picture. */
static int32_t sym_uovf(const Code *c, BitRd *b, int cap, uint32_t *flags)     /* h4m:667-677 */
{
    int32_t total = 0, v;
    int guard = 0;
    (void)flags;
    do { v = sym(c, b); total += v; } while (v >= 0xFF && ++guard < cap);
    return total;
}

/* ------------------------------------------------------------------ parser state */
typedef struct {
    int hb, vb, stride;          /* blocks; map row length incl. border */
    int bx_per, by_per, nblk;    /* blocks of this plane per macroblock */
    int moff[4];                 /* block offsets inside a macroblock, order TL, BL, BR, TR (h4m:447-455, 862-865) */
    int dx[4], dy[4];            /* the same as block coordinates */
    uint32_t nblocks, ntiles;
} PPlane;

struct HvqParser {
    int w, h, is15, landscape, nest_w, nest_h, wshift, hshift;
    PPlane pl[3];
    uint8_t nest[HVQ_NEST_BYTES];
    Code c_dc, c_run, c_bt, c_bn, c_mv, c_mcb;
    BitRd bn[2], bnr[2], dc[3], bt[3], rle[3], mvh, mvv, mtype, mproc;
    const uint8_t *fx[3];
    const uint8_t *end;
    int unk_shift, dc_shift;
    int32_t dc_lo, dc_hi;
    uint32_t *blk_off[3];        /* per-block pool offsets (P/B pass 2 writes out of raster order) */
    uint32_t map_off[3], mv_off, wave_base_off, fixed_bytes, pic_bytes, plane_off[3];
    uint32_t total_tiles;
    uint32_t flags;
    uint32_t max_items, max_pairs;
    size_t bound;
    uint8_t *dcv[3];             /* DC values of the picture being parsed, bordered like the maps (dcv_ent) */
    /* pass 2 in two steps: the planes' coefficient symbols and MC-residual scalars are decoded front to back into these arrays (the
     * serial part: one task per section), then the payload dwords are assembled from them by tasks that each take a range of
     * macroblock rows -- every cursor at a row boundary follows from counts the pool layout has made (row_cum) */
    int16_t *bt_sym[3]; size_t bt_cap[3];
    int32_t *sc_val[3]; size_t sc_cap[3];
    struct RowCum { uint32_t bases, resid, off; } *row_cum[3];       /* [macroblock row]: AOT bases, MC-residual blocks, payload dwords of the plane before it */
    uint32_t n_bases[3], n_resid[3];
    uint8_t *mb_tag;             /* P/B: per macroblock type << 5 | proc << 4 (0: intra), from the type / proc runs */
    int is_P;
    uint8_t res[4];              /* P/B: residual bits of the vectors, h0 h1 v0 v1 (h4m:2023-2026) */
    uint32_t tflags[8];          /* flags raised by the tasks of a phase (one word each: tasks run side by side) */
    uint32_t capped;             /* a task met an overflow run that does not end: the other tasks stop decoding values too */
    struct ParsePool *tp;        /* hvq_parser_set_threads: workers that run a phase's tasks beside the calling thread */
    /* HVQM4_AMD_PARSE_TIMING=1 (development aid): wall time per stage, summed over the parser's pictures, on stderr at destroy */
    int timing;
    uint64_t t_mark, t_stage[2][5], t_pics[2];
};

#define ALIGN16(x) (((x) + 15u) & ~15u)

static pthread_once_t g_type_info_once = PTHREAD_ONCE_INIT;    /* two threads may create their first parser at once */
static void build_type_info(void);

HvqParser *hvq_parser_create(int width, int height, int h_samp, int v_samp, int is15)
{
#if defined(__x86_64__) && defined(__AVX2__)
    /* this file is built for x86-64-v3 (Makefile HOST_ARCH: BMI2 shifts in the bit reader): every entry point that parses goes through a
     * parser object, so an older CPU is refused HERE with a NULL parser instead of dying on an illegal instruction in the first symbol */
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("bmi2")) return NULL;
#endif
    if (width < 8 || height < 8 || (width & 7) || (height & 7) || width > 8192 || height > 8192) return NULL;
    /* 4:2:0, 4:4:4, and 4:2:2 as h_samp 2 / v_samp 1 (two chroma blocks per macroblock, one above the other).  The fourth
     * combination the reference's setHVQPlaneDesc accepts, h_samp 1 / v_samp 2, is not decodable by the reference itself: with
     * two blocks per macroblock its tables (mcb_offset[1] = stride, pb_offset[1] = 4 rows down, h4m:863-870) address the block
     * BELOW, i.e. the next macroblock row's, and never the one to the right. */
    if (!((h_samp == 2 || h_samp == 1) && (v_samp == 2 || v_samp == 1)) || (h_samp == 1 && v_samp == 2)) return NULL;
    pthread_once(&g_type_info_once, build_type_info);
    HvqParser *p = calloc(1, sizeof *p);
    if (!p) return NULL;
    p->w = width; p->h = height; p->is15 = is15 != 0;
    p->landscape = width >= height;                              /* h4m:965-975 */
    p->nest_w = p->landscape ? 70 : 38;
    p->nest_h = p->landscape ? 38 : 70;
    p->wshift = h_samp == 2; p->hshift = v_samp == 2;
    uint32_t off = sizeof(HvqPicHeader), poff = 0, blocks = 0;
    for (int i = 0; i < 3; ++i) {
        PPlane *q = &p->pl[i];
        int ws = i ? p->wshift : 0, hs = i ? p->hshift : 0;
        q->hb = (width >> ws) / 4; q->vb = (height >> hs) / 4;    /* h4m:856-857 */
        q->stride = q->hb + 2;
        q->bx_per = 2 >> ws; q->by_per = 2 >> hs; q->nblk = q->bx_per * q->by_per;
        q->moff[0] = 0; q->moff[1] = q->stride; q->moff[2] = q->stride + 1; q->moff[3] = 1;
        q->dx[0] = 0; q->dy[0] = 0; q->dx[1] = 0; q->dy[1] = 1; q->dx[2] = 1; q->dy[2] = 1; q->dx[3] = 1; q->dy[3] = 0;
        q->nblocks = (uint32_t)q->hb * q->vb;
        q->ntiles = (q->nblocks + HVQ_TILE_BLOCKS - 1) / HVQ_TILE_BLOCKS;
        p->total_tiles += q->ntiles;
        p->map_off[i] = off;
        off = ALIGN16(off + 2u * q->stride * (q->vb + 2));
        p->plane_off[i] = poff;
        poff += (uint32_t)(width >> ws) * (height >> hs);
        blocks += q->nblocks;
        p->blk_off[i] = malloc(sizeof(uint32_t) * (q->nblocks + 1));
        p->dcv[i] = malloc((size_t)q->stride * (q->vb + 2) + 16);
        p->row_cum[i] = malloc(sizeof(struct RowCum) * ((size_t)height / 8 + 2));
    }
    p->pic_bytes = poff;
    p->timing = getenv("HVQM4_AMD_PARSE_TIMING") != NULL;
    p->mb_tag = malloc((size_t)(width / 8) * (size_t)(height / 8) + 1);
    p->mv_off = off;
    off = ALIGN16(off + 4u * (width / 8) * (height / 8));
    p->wave_base_off = off;
    off = ALIGN16(off + 4u * p->total_tiles * (HVQ_TILE_BLOCKS / 64));
    p->fixed_bytes = off;
    p->bound = (size_t)off + 64u * blocks + ALIGN16(HVQ_NESTP_BYTES) + 64;
    return p;
}

static void pool_destroy(struct ParsePool *tp);

void hvq_parser_destroy(HvqParser *p)
{
    if (!p) return;
    if (p->timing)
        for (int k = 0; k < 2; ++k)
            if (p->t_pics[k])
                fprintf(stderr, "hvq_parse %s pictures (%llu): trees+tags %.1f | phase 1 %.1f | merge+layout %.1f | phase 2 %.1f | rest %.1f us per picture\n",
                        k ? "P/B" : "I", (unsigned long long)p->t_pics[k], p->t_stage[k][0] * 1e-3 / p->t_pics[k], p->t_stage[k][1] * 1e-3 / p->t_pics[k],
                        p->t_stage[k][2] * 1e-3 / p->t_pics[k], p->t_stage[k][3] * 1e-3 / p->t_pics[k], p->t_stage[k][4] * 1e-3 / p->t_pics[k]);
    pool_destroy(p->tp);
    for (int i = 0; i < 3; ++i) { free(p->blk_off[i]); free(p->dcv[i]); free(p->row_cum[i]); free(p->bt_sym[i]); free(p->sc_val[i]); }
    free(p->mb_tag);
    free(p);
}

size_t hvq_parser_blob_bound(const HvqParser *p) { return p->bound; }
uint32_t hvq_parser_pic_bytes(const HvqParser *p) { return p->pic_bytes; }

/* ------------------------------------------------------------------ sections (h4m:1061-1071, 1979-1993, 2030-2044) */
static const uint8_t *section(const HvqParser *p, const uint8_t *data, const uint8_t *tab, int i, int *live)
{
    const uint8_t *s = data + be32(tab + 4 * i);
    if (s + 4 > p->end) { *live = 0; return p->end; }
    *live = be32(s) != 0;
    return s + 4;
}

static BitRd section_bits(const HvqParser *p, const uint8_t *data, const uint8_t *tab, int i)
{
    BitRd b = { 0 };
    b.p = section(p, data, tab, i, &b.live);
    b.end = p->end;
    return b;
}

static void common_sections(HvqParser *p, const uint8_t *data, const uint8_t *tab)
{
    for (int i = 0; i < 2; ++i) {
        p->bn[i] = section_bits(p, data, tab, 2 * i);
        p->bnr[i] = section_bits(p, data, tab, 2 * i + 1);
    }
    for (int k = 0; k < 3; ++k) {
        int live;
        p->dc[k] = section_bits(p, data, tab, 4 + 3 * k);
        p->bt[k] = section_bits(p, data, tab, 5 + 3 * k);
        p->fx[k] = section(p, data, tab, 6 + 3 * k, &live);
    }
}

/* ------------------------------------------------------------------ blob helpers */
static inline uint8_t *map_ent(const HvqParser *p, uint8_t *blob, int plane, int by, int bx)
{
    return blob + p->map_off[plane] + 2u * ((uint32_t)(by + 1) * p->pl[plane].stride + (uint32_t)(bx + 1));
}

/* DC values are decoded into a side array per plane (bordered like the map, one byte per entry) and merged into the map's value
 * bytes afterwards: the tasks that decode a plane's block kinds and its DC values run side by side, and as writers of the odd and
 * of the even bytes of the SAME map rows they would pass every cache line back and forth between their cores */
static inline uint8_t *dcv_ent(const HvqParser *p, int plane, int by, int bx)
{
    return p->dcv[plane] + (size_t)(by + 1) * p->pl[plane].stride + (size_t)(bx + 1);
}

static void dcv_init(const HvqParser *p, int fill)
{
    for (int i = 0; i < 3; ++i) {
        const PPlane *q = &p->pl[i];
        uint8_t *d = p->dcv[i];
        const size_t n = (size_t)q->stride * (q->vb + 2);
        memset(d, fill, n);
        memset(d, 0x7F, (size_t)q->stride);
        memset(d + (size_t)(q->vb + 1) * q->stride, 0x7F, (size_t)q->stride);
        for (int r = 1; r <= q->vb; ++r) { d[(size_t)r * q->stride] = 0x7F; d[(size_t)r * q->stride + q->stride - 1] = 0x7F; }
    }
}

static void dcv_merge(const HvqParser *p, uint8_t *blob)
{
    for (int i = 0; i < 3; ++i) {
        const PPlane *q = &p->pl[i];
        uint8_t *m = blob + p->map_off[i];
        const uint8_t *d = p->dcv[i];
        const size_t n = (size_t)q->stride * (q->vb + 2);
        for (size_t k = 0; k < n; ++k) m[2 * k] = d[k];
    }
}

static void init_maps(const HvqParser *p, uint8_t *blob)                         /* h4m:951-955, 1001-1040 */
{
    for (int i = 0; i < 3; ++i) {
        const PPlane *q = &p->pl[i];
        uint8_t *m = blob + p->map_off[i];
        size_t n = (size_t)q->stride * (q->vb + 2);
        memset(m, 0, 2 * n);
        for (int c = 0; c < q->stride; ++c) {
            m[2 * c] = 0x7F; m[2 * c + 1] = 0xFF;
            m[2 * ((size_t)(q->vb + 1) * q->stride + c)] = 0x7F; m[2 * ((size_t)(q->vb + 1) * q->stride + c) + 1] = 0xFF;
        }
        for (int r = 1; r <= q->vb; ++r) {
            uint8_t *l = m + 2 * (size_t)r * q->stride, *rr = l + 2 * (q->stride - 1);
            l[0] = 0x7F; l[1] = 0xFF; rr[0] = 0x7F; rr[1] = 0xFF;
        }
    }
}

/* per type byte: payload dwords (hvq_payload_dwords), whether the kernel queues the block, its basis count and the
 * header flags it raises -- one table per context: [0] I luma, [1] I chroma, [2] P/B */
typedef struct { uint8_t n, item, pairs, flags, resid; } TypeInfo;
static TypeInfo g_type_info[3][256];

static void build_type_info(void)
{
    for (int ctx = 0; ctx < 3; ++ctx)
        for (uint32_t t = 0; t < 256; ++t) {
            const int is_pb = ctx == 2, il = ctx == 0;
            const uint32_t n = hvq_payload_dwords(t, is_pb, il);
            const int inter = is_pb && (t & 0x60u);
            const uint32_t kind = il ? t : (t & 0xFu);
            TypeInfo ti = { (uint8_t)n, 0, 0, 0, 0 };
            if (n && kind != 6) {                         /* queued by the kernel: intra AOT or MC residual */
                ti.item = 1;
                ti.pairs = (uint8_t)(inter ? kind - 1 : kind);
                ti.resid = inter ? 1 : 0;                 /* two scalars in front of its bases (h4m:1405-1406) */
                if (!inter) ti.flags = (uint8_t)(HVQ_F_HAS_NEST | (kind > 15 ? HVQ_F_BIG_AOT : 0));
            }
            g_type_info[ctx][t] = ti;
        }
}

/* raster scan of the type maps: per-block pool offsets, per-64-block bases, flags; returns pool dwords */
static uint32_t layout_pool(HvqParser *p, uint8_t *blob, int is_pb)
{
    uint32_t *wave_base = (uint32_t *)(blob + p->wave_base_off);
    uint32_t off = 0, wv = 0, flags = 0;
    p->max_items = 0; p->max_pairs = 0;
    for (int i = 0; i < 3; ++i) {
        const PPlane *q = &p->pl[i];
        const TypeInfo *info = g_type_info[is_pb ? 2 : (i == 0 ? 0 : 1)];
        uint32_t *blk_off = p->blk_off[i];
        uint32_t b = 0, items = 0, pairs = 0, cb = 0, cr = 0;
        const uint32_t off0 = off;
        struct RowCum *rc = p->row_cum[i];
        for (int by = 0; by < q->vb; ++by) {
            const uint8_t *row = map_ent(p, blob, i, by, 0);
            if (by % q->by_per == 0) { rc->bases = cb; rc->resid = cr; rc->off = off - off0; ++rc; }
            for (int bx = 0; bx < q->hb; ++bx, ++b) {
                if ((b & 63u) == 0) {
                    wave_base[wv++] = off;
                    if ((b % HVQ_TILE_BLOCKS) == 0) {
                        if (items > p->max_items) p->max_items = items;
                        if (pairs > p->max_pairs) p->max_pairs = pairs;
                        items = pairs = 0;
                    }
                }
                const TypeInfo ti = info[row[2 * bx + 1]];
                blk_off[b] = off;
                off += ti.n;
                items += ti.item;
                pairs += ti.pairs;
                flags |= ti.flags;
                cb += ti.pairs;
                cr += ti.resid;
            }
        }
        rc->bases = cb; rc->resid = cr; rc->off = off - off0;       /* behind the last macroblock row */
        p->n_bases[i] = cb; p->n_resid[i] = cr;
        if (items > p->max_items) p->max_items = items;
        if (pairs > p->max_pairs) p->max_pairs = pairs;
        /* the last tile of a plane may be ragged: its unused runs point at the plane's end */
        while (wv % (HVQ_TILE_BLOCKS / 64)) wave_base[wv++] = off;
    }
    p->flags |= flags;
    return off;
}

static void fill_header(const HvqParser *p, uint8_t *blob, int kind, uint32_t pool_dwords, uint32_t total)
{
    HvqPicHeader *h = (HvqPicHeader *)blob;
    memset(h, 0, sizeof *h);
    h->magic = HVQ_MAGIC;
    h->total_bytes = total;
    h->width = (uint16_t)p->w; h->height = (uint16_t)p->h;
    h->pic_kind = (uint8_t)kind;
    h->unk_shift = (uint8_t)p->unk_shift;
    h->dc_shift = (uint8_t)p->dc_shift;
    h->wshift = (uint8_t)p->wshift; h->hshift = (uint8_t)p->hshift;
    h->flags = p->flags | (p->is15 ? HVQ_F_IS15 : 0) | (p->landscape ? HVQ_F_LANDSCAPE : 0);
    uint32_t t = 0;
    for (int i = 0; i < 3; ++i) {
        h->hb[i] = (uint16_t)p->pl[i].hb; h->vb[i] = (uint16_t)p->pl[i].vb;
        h->plane_off[i] = p->plane_off[i];
        h->map_off[i] = p->map_off[i];
        h->tile_first[i] = t;
        t += p->pl[i].ntiles;
    }
    h->tile_first[3] = t;
    h->pic_bytes = p->pic_bytes;
    h->mv_off = kind == HVQ_PIC_I ? 0 : p->mv_off;
    h->wave_base_off = p->wave_base_off;
    h->pool_off = p->fixed_bytes;
    h->pool_dwords = pool_dwords;
    h->nest_off = (p->flags & HVQ_F_HAS_NEST) ? ALIGN16(p->fixed_bytes + 4u * pool_dwords) : 0;
    h->mcb_w = (uint32_t)p->w / 8; h->mcb_h = (uint32_t)p->h / 8;
    h->max_items = (uint16_t)p->max_items; h->max_pairs = p->max_pairs;
}

void hvq_parser_layout(const HvqParser *p, HvqPicHeader *out)
{
    HvqParser q = *p;
    q.flags = 0; q.unk_shift = 0; q.dc_shift = 0; q.max_items = 0; q.max_pairs = 0;
    fill_header(&q, (uint8_t *)out, HVQ_PIC_P, 0, 0);
}

/* nest values are 4 bits (h4m:1211): two per byte, value n in nibble n -- what the kernel stages in LDS */
static void pack_nest(uint8_t *dst, const uint8_t *nest)
{
    memset(dst, 0, ALIGN16(HVQ_NESTP_BYTES));
    for (int i = 0; i < HVQ_NEST_BYTES; i += 2) dst[i >> 1] = (uint8_t)((nest[i] & 0xF) | ((nest[i + 1] & 0xF) << 4));
}

void hvq_parser_packed_nest(const HvqParser *p, uint8_t *out) { pack_nest(out, p->nest); }

/* ------------------------------------------------------------------ tasks
 * A picture's sections are independent bit buffers (h4m:1981-1993, 2030-2044): which section a loop of the reference reads from never
 * depends on what another section said, except through the macroblock tags (type / proc runs) of a P/B picture.  The parse is
 * therefore cut into TASKS that own the sections they read -- block kinds of luma, block kinds of chroma, DC values per plane,
 * vectors, payloads per plane -- in two phases with the pool layout between them.  One thread runs the tasks in order (the batched
 * paths: one parser per thread, pictures in parallel); hvq_parser_set_threads gives a parser a small persistent pool that runs a
 * phase's tasks side by side (the SDK boundary: one synchronous picture at a time, 78 % of a call was this parse on one core).
 * Same tasks, same code, same blob either way. */
static inline uint32_t *task_flags(HvqParser *p, int task) { return &p->tflags[task]; }

/* a task that meets an endless overflow run tells the others: their values no longer matter either (sym_sovf returns at once) */
static inline int32_t t_sovf(HvqParser *p, const Code *c, BitRd *b, uint32_t *flags)
{
    if (__builtin_expect(__atomic_load_n(&p->capped, __ATOMIC_RELAXED), 0)) *flags |= HVQ_F_CAPPED;
    const int32_t v = sym_sovf(c, b, p->dc_lo, p->dc_hi, flags);
    if (__builtin_expect(*flags & HVQ_F_CAPPED, 0)) __atomic_store_n(&p->capped, 1u, __ATOMIC_RELAXED);
    return v;
}

/* ------------------------------------------------------------------ I pictures */
static void ipic_kinds_luma(HvqParser *p, uint8_t *blob)                         /* h4m:1073-1100 */
{
    const PPlane *Y = &p->pl[0];
    BitRd bn = p->bn[0], bnr = p->bnr[0];
    uint32_t run = 0;
    for (int by = 0; by < Y->vb; ++by) {
        uint8_t *row = map_ent(p, blob, 0, by, 0);
        for (int bx = 0; bx < Y->hb; ++bx) {
            if (run) { --run; continue; }                        /* type already 0 */
            int32_t k = sym(&p->c_bn, &bn) & 0xFFFF;
            if ((int16_t)k == 0) run = (uint32_t)sym(&p->c_run, &bnr);
            row[2 * bx + 1] = (uint8_t)k;
        }
    }
}

static void ipic_kinds_chroma(HvqParser *p, uint8_t *blob)                       /* h4m:1102-1130 */
{
    const PPlane *C = &p->pl[1];
    BitRd bn = p->bn[1], bnr = p->bnr[1];
    uint32_t run = 0;
    for (int by = 0; by < C->vb; ++by) {
        uint8_t *ru = map_ent(p, blob, 1, by, 0), *rv = map_ent(p, blob, 2, by, 0);
        for (int bx = 0; bx < C->hb; ++bx) {
            if (run) { --run; continue; }
            int32_t k = sym(&p->c_bn, &bn) & 0xFFFF;
            if ((int16_t)k == 0) run = (uint32_t)sym(&p->c_run, &bnr);
            ru[2 * bx + 1] = k & 0xF;
            rv[2 * bx + 1] = (k >> 4) & 0xF;
        }
    }
}

static void ipic_dc_plane(HvqParser *p, uint8_t *blob, int i, uint32_t *flags)   /* h4m:1043-1058, 1132-1164 */
{
    const PPlane *q = &p->pl[i];
    BitRd dc = p->dc[i], rle = p->rle[i];
    uint32_t run = 0;
    (void)blob;
    for (int by = 0; by < q->vb; ++by) {
        uint8_t *row = dcv_ent(p, i, by, 0);
        const uint8_t *up = dcv_ent(p, i, by - 1, 0);
        uint8_t pred = up[0];
        for (int bx = 0; bx < q->hb; ++bx) {
            uint32_t delta = 0;
            if (run) --run;
            else {
                delta = (uint32_t)t_sovf(p, &p->c_dc, &dc, flags);
                if (delta == 0) run = (uint32_t)sym(&p->c_run, &rle);
            }
            uint8_t v = (uint8_t)(pred + delta);               /* uint8 wrap: h4m:1145-1149 */
            row[bx] = v;
            pred = (uint8_t)((v + up[bx + 1] + 1) / 2);
        }
    }
}

/* ---- pass 2, step a (I, P and B pictures): a plane's coefficient symbols (bufTree0, h4m:726) and MC-residual scalars (h4m:1405-1406)
 * decoded front to back into arrays -- the serial part of the payload walk, one task per section */
static int payload_arrays(HvqParser *p)
{
    for (int i = 0; i < 3; ++i) {
        if (p->n_bases[i] > p->bt_cap[i]) {
            free(p->bt_sym[i]);
            p->bt_cap[i] = (size_t)p->n_bases[i] + p->n_bases[i] / 4 + 64;
            p->bt_sym[i] = malloc(sizeof(int16_t) * p->bt_cap[i]);
            if (!p->bt_sym[i]) { p->bt_cap[i] = 0; return HVQ_E_OVERFLOW; }
        }
        if (2u * (size_t)p->n_resid[i] > p->sc_cap[i]) {
            free(p->sc_val[i]);
            p->sc_cap[i] = 2u * (size_t)p->n_resid[i] + p->n_resid[i] / 2 + 64;
            p->sc_val[i] = malloc(sizeof(int32_t) * p->sc_cap[i]);
            if (!p->sc_val[i]) { p->sc_cap[i] = 0; return HVQ_E_OVERFLOW; }
        }
    }
    return HVQ_OK;
}

static void decode_coefficients(HvqParser *p, int i)
{
    BitRd b = p->bt[i];
    int16_t *o = p->bt_sym[i];
    const uint32_t n = p->n_bases[i];
    for (uint32_t k = 0; k < n; ++k) o[k] = (int16_t)sym(&p->c_bt, &b);     /* leaf values are int16 (h4m:613-617) */
}

static void decode_scalars(HvqParser *p, int i, uint32_t *flags)
{
    BitRd b = p->dc[i];
    int32_t *o = p->sc_val[i];
    const uint32_t n = 2u * p->n_resid[i];
    for (uint32_t k = 0; k < n; ++k) o[k] = t_sovf(p, &p->c_dc, &b, flags);
}

/* heaviest first: 0 luma coefficients, 1 luma scalars, 2 / 3 chroma coefficients, 4 / 5 chroma scalars */
static void payload_phase_a(HvqParser *p, uint8_t *blob, int task)
{
    (void)blob;
    switch (task) {
    case 0: decode_coefficients(p, 0); break;
    case 1: decode_scalars(p, 0, task_flags(p, task)); break;
    case 2: case 3: decode_coefficients(p, task - 1); break;
    default: decode_scalars(p, task - 3, task_flags(p, task)); break;
    }
}

/* ---- pass 2, step b: the payload dwords of a range of macroblock rows of one plane, from the fixed-length section (byte addressed,
 * h4m:545-548) and the arrays of step a.  Where the range starts in each of them follows from row_cum. */
typedef struct { const uint8_t *fx, *end; const int16_t *bt; const int32_t *sc; } AsmCur;

static AsmCur asm_cursor(const HvqParser *p, int i, int mrow)
{
    const struct RowCum *rc = &p->row_cum[i][mrow];
    const uint32_t lits = (rc->off - rc->bases - 2u * rc->resid) / 4u;
    const size_t at = 16u * (size_t)lits + 2u * (size_t)rc->bases;
    AsmCur c;
    c.end = p->end;
    c.fx = (p->fx[i] <= p->end && at <= (size_t)(p->end - p->fx[i])) ? p->fx[i] + at : p->end;     /* beyond the picture: zeros, like the serial walk */
    c.bt = p->bt_sym[i] + rc->bases;
    c.sc = p->sc_val[i] + 2u * (size_t)rc->resid;
    return c;
}

static inline void asm_literal(AsmCur *c, uint32_t *dst)                         /* h4m:543-549 */
{
    const uint8_t *s = c->fx;
    if (s + 16 <= c->end) { memcpy(dst, s, 16); c->fx = s + 16; }
    else { memset(dst, 0, 16); c->fx = c->end; }
}

/* `n` bases of one block: word from fixvl, coefficient from bufTree0 (h4m:691-692, 726-731 / 738-739, 767-772) */
static inline void asm_bases(AsmCur *c, uint32_t n, uint32_t *dst)
{
    uint32_t run = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint8_t *s = c->fx;
        uint32_t word = 0;
        if (s + 2 <= c->end) { word = be16(s); c->fx = s + 2; } else c->fx = c->end;
        run += (uint32_t)(int32_t)*c->bt++;
        dst[k] = HVQ_BASIS(word, (run + ((word >> 13) & 3u)) & 0x3FFFFu);
    }
}

/* I picture: plane raster order == reference consumption order (h4m:2011-2015) */
static void ipic_assemble(HvqParser *p, uint8_t *blob, int i, int r0, int r1)
{
    const PPlane *q = &p->pl[i];
    uint32_t *pool = (uint32_t *)(blob + p->fixed_bytes);
    int by0 = r0 * q->by_per, by1 = r1 * q->by_per;
    if (by1 > q->vb) by1 = q->vb;
    if (by0 >= by1) return;
    AsmCur c = asm_cursor(p, i, r0);
    uint32_t b = (uint32_t)by0 * (uint32_t)q->hb;
    for (int by = by0; by < by1; ++by) {
        const uint8_t *row = map_ent(p, blob, i, by, 0);
        for (int bx = 0; bx < q->hb; ++bx, ++b) {
            uint32_t k = row[2 * bx + 1];
            if (k == 0 || k == 8) continue;
            uint32_t *dst = pool + p->blk_off[i][b];
            if (k == 6) asm_literal(&c, dst);
            else asm_bases(&c, k, dst);
        }
    }
}

/* the macroblock-row range of assembly task `task`: luma in four parts, the chroma planes whole */
static inline void asm_range(const HvqParser *p, int task, int *plane, int *r0, int *r1)
{
    const int mh = p->h / 8;
    if (task < 4) { *plane = 0; *r0 = mh * task / 4; *r1 = mh * (task + 1) / 4; }
    else { *plane = task - 3; *r0 = 0; *r1 = mh; }
}

static void ipic_phase2b(HvqParser *p, uint8_t *blob, int task)
{
    int i, r0, r1;
    asm_range(p, task, &i, &r0, &r1);
    ipic_assemble(p, blob, i, r0, r1);
}

/* a prefix tree read from the head of its carrier section (h4m:632-642); the section's cursor goes through a copy on this thread's
 * stack (the parser's cursors of neighbouring sections share cache lines, and the trees of a picture are read side by side) */
static void tree_from(Code *c, BitRd *carrier, int is_signed, int scale)
{
    BitRd b = *carrier;
    code_read(c, &b, is_signed, scale);
    *carrier = b;
}

/* phase 0 of an I picture (h4m:1996-1999): the four trees, and the maps' borders */
static void ipic_phase0(HvqParser *p, uint8_t *blob, int task)
{
    switch (task) {
    case 0: tree_from(&p->c_bn, &p->bn[0], 0, 0); tree_from(&p->c_run, &p->bnr[0], 0, 0); break;
    case 1: tree_from(&p->c_dc, &p->dc[0], 1, p->dc_shift & 31); tree_from(&p->c_bt, &p->bt[0], 0, 2); break;
    default: init_maps(p, blob); dcv_init(p, 0); break;
    }
}

/* phase 1 of an I picture, heaviest first: 0 luma kinds, 1 luma DC, 2 chroma kinds, 3 / 4 chroma DC */
static void ipic_phase1(HvqParser *p, uint8_t *blob, int task)
{
    switch (task) {
    case 0: ipic_kinds_luma(p, blob); break;
    case 1: ipic_dc_plane(p, blob, 0, task_flags(p, task)); break;
    case 2: ipic_kinds_chroma(p, blob); break;
    default: ipic_dc_plane(p, blob, task - 2, task_flags(p, task)); break;
    }
}

static void make_nest(HvqParser *p, const uint8_t *blob, int nx, int ny)         /* h4m:1166-1239 */
{
    const PPlane *Y = &p->pl[0];
    int cols = Y->hb < p->nest_w ? Y->hb : p->nest_w;
    int rows = Y->vb < p->nest_h ? Y->vb : p->nest_h;
    int mcols = p->nest_w - cols; if (mcols > cols) mcols = cols;
    int mrows = p->nest_h - rows; if (mrows > rows) mrows = rows;
    /* The reference indexes the bordered map FLAT (payload + stride * nest_y + nest_x, h4m:1169): a window that overlaps the
     * border column / row reads border entries (value 0x7F) and the next row's first entries -- in bounds and deterministic as
     * long as its last element lies inside the (hb+2) x (vb+2) array, and reproduced here by the same flat indexing.  Only an
     * origin whose window would leave the array (the reference then reads foreign memory) is clamped and flagged. */
    if ((ny + rows - 1) * Y->stride + nx + cols - 1 > Y->stride * (Y->vb + 1) - 2) {
        if (nx + cols > Y->hb) nx = Y->hb - cols;
        if (ny + rows > Y->vb) ny = Y->vb - rows;
        p->flags |= HVQ_F_CLAMPED;
    }
    memset(p->nest, 0, sizeof p->nest);
    for (int r = 0; r < rows; ++r) {
        uint8_t *row = p->nest + r * p->nest_w;
        const uint8_t *src = map_ent(p, (uint8_t *)blob, 0, ny + r, nx);
        for (int c = 0; c < cols; ++c) row[c] = (src[2 * c] >> 4) & 0xF;
        for (int c = 0; c < mcols; ++c) row[cols + c] = row[cols - 1 - c];
    }
    for (int r = 0; r < mrows; ++r)
        memcpy(p->nest + (rows + r) * p->nest_w, p->nest + (rows - 1 - r) * p->nest_w, (size_t)p->nest_w);
}

static void run_phase(HvqParser *p, void (*fn)(HvqParser *, uint8_t *, int), uint8_t *blob, int ntasks);
static uint64_t now_ns(void);
#define TMARK(k, i) do { if (p->timing) { const uint64_t t_ = now_ns(); p->t_stage[k][i] += t_ - p->t_mark; p->t_mark = t_; } } while (0)

static int parse_ipic(HvqParser *p, const uint8_t *pic, uint8_t *blob, size_t cap, size_t *blob_len)
{
    p->dc_shift = pic[0];
    p->unk_shift = pic[1];
    int nx = (int)be16(pic + 4), ny = (int)be16(pic + 6);
    const uint8_t *tab = pic + 8, *data = pic + 8 + 0x40;
    common_sections(p, data, tab);
    for (int k = 0; k < 3; ++k) p->rle[k] = section_bits(p, data, tab, 13 + k);
    p->dc_hi = (int32_t)((uint32_t)0x7F << (p->dc_shift & 31));
    p->dc_lo = (int32_t)((uint32_t)-0x80 << (p->dc_shift & 31));
    run_phase(p, ipic_phase0, blob, 3);
    TMARK(0, 0);
    run_phase(p, ipic_phase1, blob, 5);
    TMARK(0, 1);
    for (int t = 0; t < 5; ++t) p->flags |= p->tflags[t];
    dcv_merge(p, blob);
    make_nest(p, blob, nx, ny);
    uint32_t pool_dwords = layout_pool(p, blob, 0);
    size_t total = (size_t)p->fixed_bytes + 4u * (size_t)pool_dwords;
    if (p->flags & HVQ_F_HAS_NEST) total = ALIGN16(total) + ALIGN16(HVQ_NESTP_BYTES);
    total = ALIGN16(total);
    if (total > cap || pool_dwords >= (1u << 22)) return HVQ_E_OVERFLOW;   /* kernel packs offsets in 22 bits */
    fill_header(p, blob, HVQ_PIC_I, pool_dwords, (uint32_t)total);
    if (payload_arrays(p)) return HVQ_E_OVERFLOW;
    TMARK(0, 2);
    run_phase(p, payload_phase_a, blob, 4);                       /* an I picture has no MC-residual scalars: tasks 0, 2, 3 decode, 1 is empty */
    run_phase(p, ipic_phase2b, blob, 6);
    TMARK(0, 3);
    if (p->flags & HVQ_F_HAS_NEST)
        pack_nest(blob + ((HvqPicHeader *)blob)->nest_off, p->nest);
    TMARK(0, 4);
    p->t_pics[0]++;
    *blob_len = total;
    return HVQ_OK;
}

/* ------------------------------------------------------------------ P/B pictures */
typedef struct { uint32_t value, count; } RunLen;

/* the macroblock tags -- type << 5 | proc << 4, 0 for an intra macroblock -- from the type and proc runs (h4m:1545-1622, 1742-1776):
 * the one thing every other section's loop depends on, decoded first and alone (two short run-length sections) */
static void pb_tags(HvqParser *p, int is_P)
{
    static const uint32_t step[2][4] = { { 1, 2, 0, 2 }, { 2, 0, 1, 0 } };
    const int cap = UOVF_CAP((uint32_t)(p->w / 8) * (uint32_t)(p->h / 8));
    RunLen type = { 0, 0 }, proc = { 0, 0 };
    if (p->mproc.live) { proc.value = br_take(&p->mproc, 1); proc.count = (uint32_t)sym_uovf(&p->c_mcb, &p->mproc, cap, &p->flags); }
    if (p->mtype.live) { type.value = br_take(&p->mtype, 2); type.count = (uint32_t)sym_uovf(&p->c_mcb, &p->mtype, cap, &p->flags); }
    const int nmb = (p->w / 8) * (p->h / 8);
    uint8_t *tags = p->mb_tag;
    for (int m = 0; m < nmb; ++m) {
        if (type.count == 0) {
            type.value = step[br_take(&p->mtype, 1)][type.value & 3];
            type.count = (uint32_t)sym_uovf(&p->c_mcb, &p->mtype, cap, &p->flags);
        }
        --type.count;
        if (type.value == 0) { tags[m] = 0; continue; }
        if (is_P && type.value >= 2) p->flags |= HVQ_F_SELF_REF;
        if (proc.count == 0) { proc.value ^= 1; proc.count = (uint32_t)sym_uovf(&p->c_mcb, &p->mproc, cap, &p->flags); }
        --proc.count;
        tags[m] = (uint8_t)((type.value << 5) | (proc.value << 4));
    }
}

/* block kinds of one plane group (g = 0 luma, 1 chroma: U in the low nibble, V in the high one) over all macroblocks (h4m:1670-1740):
 * every block's type byte = its macroblock's tag | its kind; plain-MC macroblocks (proc 1) carry no kinds */
static void pb_kinds_group(HvqParser *p, uint8_t *blob, int g)
{
    const int mw = p->w / 8, mh = p->h / 8;
    const uint8_t *tags = p->mb_tag;
    uint32_t rl = 0;
    BitRd bn = p->bn[g], bnr = p->bnr[g];
    if (g == 0) {
        const PPlane *Y = &p->pl[0];
        for (int my = 0, m = 0; my < mh; ++my)
            for (int mx = 0; mx < mw; ++mx, ++m) {
                const uint8_t tag = tags[m];
                uint8_t *e = map_ent(p, blob, 0, my * Y->by_per, mx * Y->bx_per);
                if (tag & 0x10) { for (int j = 0; j < Y->nblk; ++j) e[2 * Y->moff[j] + 1] = tag; continue; }
                for (int j = 0; j < Y->nblk; ++j) {
                    uint8_t *t = &e[2 * Y->moff[j] + 1];
                    if (rl) { *t = tag; --rl; continue; }
                    int16_t k = (int16_t)sym(&p->c_bn, &bn);
                    if (k) *t = (uint8_t)(tag | k);
                    else { *t = tag; rl = (uint32_t)sym(&p->c_run, &bnr); }
                }
            }
        return;
    }
    const PPlane *C = &p->pl[1];
    for (int my = 0, m = 0; my < mh; ++my)
        for (int mx = 0; mx < mw; ++mx, ++m) {
            const uint8_t tag = tags[m];
            uint8_t *eu = map_ent(p, blob, 1, my * C->by_per, mx * C->bx_per);
            uint8_t *ev = map_ent(p, blob, 2, my * C->by_per, mx * C->bx_per);
            if (tag & 0x10) { for (int j = 0; j < C->nblk; ++j) eu[2 * C->moff[j] + 1] = ev[2 * C->moff[j] + 1] = tag; continue; }
            for (int j = 0; j < C->nblk; ++j) {
                uint8_t *tu = &eu[2 * C->moff[j] + 1], *tv = &ev[2 * C->moff[j] + 1];
                if (rl) { *tu = *tv = tag; --rl; continue; }
                int16_t k = (int16_t)sym(&p->c_bn, &bn);
                if (k) { *tu = (uint8_t)(tag | (k & 0xF)); *tv = (uint8_t)(tag | ((k >> 4) & 0xF)); }
                else { *tu = *tv = tag; rl = (uint32_t)sym(&p->c_run, &bnr); }
            }
        }
}

/* DC values of the intra macroblocks of one plane (h4m:1649-1668): a running sum, back to 0x7F at every non-intra macroblock */
static void pb_dc_plane(HvqParser *p, uint8_t *blob, int i, uint32_t *flags)
{
    const int mw = p->w / 8, mh = p->h / 8;
    const uint8_t *tags = p->mb_tag;
    const PPlane *q = &p->pl[i];
    BitRd dc = p->dc[i];
    uint32_t pbdc = 0x7F;
    (void)blob;
    for (int my = 0, m = 0; my < mh; ++my)
        for (int mx = 0; mx < mw; ++mx, ++m) {
            if (tags[m]) { pbdc = 0x7F; continue; }
            uint8_t *e = dcv_ent(p, i, my * q->by_per, mx * q->bx_per);
            for (int j = 0; j < q->nblk; ++j) {
                pbdc += (uint32_t)t_sovf(p, &p->c_dc, &dc, flags);
                e[q->moff[j]] = (uint8_t)pbdc;
            }
        }
    p->dc[i] = dc;                                                  /* the plane's scalars of pass 2 follow in the same section */
}

static void mvec(HvqParser *p, int32_t *acc, BitRd *b, int rbits)               /* h4m:1846-1860 */
{
    rbits &= 15;
    int32_t lim = (int32_t)(1u << (rbits + 5));
    int32_t v = (int32_t)((uint32_t)sym(&p->c_mv, b) << rbits);
    v += (int32_t)br_take(b, rbits);
    *acc += v;
    if (*acc >= lim) *acc -= lim << 1;
    else if (*acc < -lim) *acc += lim << 1;
}

static inline int16_t clamp16(uint32_t *flags, int32_t v)
{
    if (v > 32767) { *flags |= HVQ_F_CLAMPED; return 32767; }
    if (v < -32768) { *flags |= HVQ_F_CLAMPED; return -32768; }
    return (int16_t)v;
}

/* the vector chains (h4m:1943-1955): absolute half-sample targets of the inter macroblocks */
static void pb_vectors(HvqParser *p, uint8_t *blob, uint32_t *flags)
{
    const int mw = p->w / 8, mh = p->h / 8;
    const uint8_t *tags = p->mb_tag;
    int16_t *mvs = (int16_t *)(blob + p->mv_off);
    int cur_ref = -1;
    int32_t mvx = 0, mvy = 0;
    BitRd mvh = p->mvh, mvv = p->mvv;
    for (int my = 0, m = 0; my < mh; ++my)
        for (int mx = 0; mx < mw; ++mx, ++m) {
            int16_t *mvo = mvs + 2 * m;
            const int t = tags[m] >> 5;
            if (t == 0) { mvo[0] = mvo[1] = 0; continue; }
            const int r = t - 1;
            if (r != cur_ref) { cur_ref = r; mvx = mvy = 0; }       /* h4m:1943-1949 */
            mvec(p, &mvx, &mvh, p->res[r]);
            mvec(p, &mvy, &mvv, p->res[2 + r]);
            mvo[0] = clamp16(flags, mx * 16 + mvx);                 /* h4m:1954-1955 */
            mvo[1] = clamp16(flags, my * 16 + mvy);
        }
}

/* pass 2 for a range of macroblock rows of one plane (h4m:1919-1967, 1789-1827, 1862-1910): macroblock raster order */
static void pb_assemble(HvqParser *p, uint8_t *blob, int i, int r0, int r1)
{
    const int mw = p->w / 8;
    const uint8_t *tags = p->mb_tag;
    const PPlane *q = &p->pl[i];
    uint32_t *pool = (uint32_t *)(blob + p->fixed_bytes);
    const int sh_dc = p->dc_shift & 31, sh_unk = p->unk_shift & 31;
    if (r0 >= r1) return;
    AsmCur c = asm_cursor(p, i, r0);
    for (int my = r0, m = r0 * mw; my < r1; ++my)
        for (int mx = 0; mx < mw; ++mx, ++m) {
            const uint8_t tag = tags[m];
            if (tag & 0x10) continue;                               /* proc 1: plain MC, no payload (h4m:1327-1355) */
            const int by0 = my * q->by_per, bx0 = mx * q->bx_per;
            const uint8_t *e = map_ent(p, blob, i, by0, bx0);
            if (tag == 0) {                                         /* intra, h4m:1789-1827 */
                for (int j = 0; j < q->nblk; ++j) {
                    uint32_t k = e[2 * q->moff[j] + 1] & 0xFu;
                    if (k == 0 || k == 8) continue;
                    uint32_t b = (uint32_t)(by0 + q->dy[j]) * q->hb + (uint32_t)(bx0 + q->dx[j]);
                    uint32_t *dst = pool + p->blk_off[i][b];
                    if (k == 6) asm_literal(&c, dst);
                    else asm_bases(&c, k, dst);
                }
                continue;
            }
            for (int j = 0; j < q->nblk; ++j) {                     /* h4m:1862-1910 */
                uint32_t k = e[2 * q->moff[j] + 1] & 0xFu;
                if (k == 0) continue;
                uint32_t b = (uint32_t)(by0 + q->dy[j]) * q->hb + (uint32_t)(bx0 + q->dx[j]);
                uint32_t *dst = pool + p->blk_off[i][b];
                if (k == 6) { asm_literal(&c, dst); continue; }
                asm_bases(&c, k - 1, dst + 2);
                const int32_t s1 = *c.sc++, s2 = *c.sc++;           /* h4m:1405-1406 */
                dst[0] = (uint32_t)(s1 >> sh_dc) << sh_unk;
                dst[1] = (uint32_t)(s2 >> sh_dc);
            }
        }
}

static void pb_tags(HvqParser *p, int is_P);

/* phase 0 of a P/B picture (h4m:2045-2050): the six trees, the macroblock tags behind theirs, and the maps' borders */
static void pb_phase0(HvqParser *p, uint8_t *blob, int task)
{
    switch (task) {
    case 0: tree_from(&p->c_mcb, &p->mtype, 0, 0); pb_tags(p, p->is_P); break;
    case 1: tree_from(&p->c_bn, &p->bn[0], 0, 0); tree_from(&p->c_run, &p->bnr[0], 0, 0); break;
    case 2: tree_from(&p->c_dc, &p->dc[0], 1, p->dc_shift & 31); tree_from(&p->c_bt, &p->bt[0], 0, 2); tree_from(&p->c_mv, &p->mvh, 1, 0); break;
    default: init_maps(p, blob); dcv_init(p, 0); break;
    }
}

/* phase 1 of a P/B picture, heaviest first: 0 luma kinds, 1 luma DC, 2 vectors, 3 chroma kinds, 4 / 5 chroma DC */
static void pb_phase1(HvqParser *p, uint8_t *blob, int task)
{
    switch (task) {
    case 0: pb_kinds_group(p, blob, 0); break;
    case 1: pb_dc_plane(p, blob, 0, task_flags(p, task)); break;
    case 2: pb_vectors(p, blob, task_flags(p, task)); break;
    case 3: pb_kinds_group(p, blob, 1); break;
    default: pb_dc_plane(p, blob, task - 3, task_flags(p, task)); break;
    }
}
static void pb_phase2b(HvqParser *p, uint8_t *blob, int task)
{
    int i, r0, r1;
    asm_range(p, task, &i, &r0, &r1);
    pb_assemble(p, blob, i, r0, r1);
}

static int parse_pbpic(HvqParser *p, int is_P, const uint8_t *pic, uint8_t *blob, size_t cap, size_t *blob_len)
{
    p->dc_shift = pic[0];
    p->unk_shift = pic[1];
    p->res[0] = pic[2]; p->res[1] = pic[4]; p->res[2] = pic[3]; p->res[3] = pic[5];     /* h0 h1 v0 v1 (h4m:2023-2026) */
    const uint8_t *tab = pic + 8, *data = pic + 8 + 0x44;
    common_sections(p, data, tab);
    p->mvh = section_bits(p, data, tab, 13);
    p->mvv = section_bits(p, data, tab, 14);
    p->mtype = section_bits(p, data, tab, 15);
    p->mproc = section_bits(p, data, tab, 16);
    p->dc_hi = (int32_t)((uint32_t)0x7F << (p->dc_shift & 31));
    p->dc_lo = (int32_t)((uint32_t)-0x80 << (p->dc_shift & 31));
    p->is_P = is_P;
    run_phase(p, pb_phase0, blob, 4);
    TMARK(1, 0);
    run_phase(p, pb_phase1, blob, 6);
    TMARK(1, 1);
    uint32_t late = p->tflags[2];                                   /* a clamped vector target: raised behind the header, as the reference's order has it */
    p->tflags[2] = 0;
    for (int t = 0; t < 6; ++t) { p->flags |= p->tflags[t]; p->tflags[t] = 0; }
    dcv_merge(p, blob);
    uint32_t pool_dwords = layout_pool(p, blob, 1);
    size_t total = (size_t)p->fixed_bytes + 4u * (size_t)pool_dwords;
    if (p->flags & HVQ_F_HAS_NEST) total = ALIGN16(total) + ALIGN16(HVQ_NESTP_BYTES);
    total = ALIGN16(total);
    if (total > cap || pool_dwords >= (1u << 22)) return HVQ_E_OVERFLOW;
    /* the header carries the flags known HERE, like the device parser's (the blobs are compared byte for byte); what pass 2 still
     * raises (a clamped vector target cannot come later, an endless overflow run in a block's scalars can) is in hvq_parser_last_flags */
    fill_header(p, blob, is_P ? HVQ_PIC_P : HVQ_PIC_B, pool_dwords, (uint32_t)total);
    if (payload_arrays(p)) return HVQ_E_OVERFLOW;
    TMARK(1, 2);
    run_phase(p, payload_phase_a, blob, 6);
    run_phase(p, pb_phase2b, blob, 6);
    TMARK(1, 3);
    p->flags |= late;
    for (int t = 0; t < 6; ++t) p->flags |= p->tflags[t];
    if (p->flags & HVQ_F_HAS_NEST)
        pack_nest(blob + ((HvqPicHeader *)blob)->nest_off, p->nest);
    TMARK(1, 4);
    p->t_pics[1]++;
    *blob_len = total;
    return HVQ_OK;
}

/* ------------------------------------------------------------------ task pool
 * A few persistent workers per parser (hvq_parser_set_threads).  A phase is published as ONE 64-bit word -- epoch << 32 | tasks << 8 |
 * next task -- and a task is claimed by a compare-and-swap on that word, so a worker that is late for a phase can never claim a task
 * of the next one with the previous phase's function.  The calling thread claims tasks like a worker and then waits for the phase's
 * count of finished tasks.  Idle workers spin for a short while (the next phase, or the next picture of a synchronous decode loop, is
 * microseconds away) and then sleep on a condition variable. */
typedef void (*PhaseFn)(HvqParser *, uint8_t *, int);
struct ParsePool {
    int nworkers;
    pthread_t th[7];
    pthread_mutex_t mu;
    pthread_cond_t cv;
    _Atomic uint64_t state;
    _Atomic uint32_t done;
    _Atomic int sleepers, stop;
    HvqParser *p;                /* the phase: written before its epoch is published */
    PhaseFn fn;
    uint8_t *blob;
};

static inline void cpu_pause(void)
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

static uint64_t now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

/* claim and run tasks of the phase `st` belongs to until none is left; returns the last state seen */
static uint64_t pool_drain(struct ParsePool *tp, uint64_t st)
{
    for (;;) {
        const uint32_t nt = (uint32_t)(st >> 8) & 0xFFu, nx = (uint32_t)st & 0xFFu;
        if (nx >= nt) return st;
        if (atomic_compare_exchange_weak(&tp->state, &st, st + 1)) {
            tp->fn(tp->p, tp->blob, (int)nx);
            atomic_fetch_add_explicit(&tp->done, 1u, memory_order_release);
            st = atomic_load(&tp->state);
        }
    }
}

static void *pool_worker(void *arg)
{
    struct ParsePool *tp = (struct ParsePool *)arg;
    uint64_t st = atomic_load(&tp->state);
    for (;;) {
        st = pool_drain(tp, st);
        /* idle: spin for ~150 us, then sleep until the state word changes */
        const uint64_t t0 = now_ns();
        uint64_t cur;
        uint32_t spins = 0;
        for (;;) {
            if (atomic_load(&tp->stop)) return NULL;
            cur = atomic_load(&tp->state);
            if (cur != st) break;
            cpu_pause();
            if ((++spins & 255u) == 0 && now_ns() - t0 > 150000u) {
                pthread_mutex_lock(&tp->mu);
                atomic_fetch_add(&tp->sleepers, 1);
                while ((cur = atomic_load(&tp->state)) == st && !atomic_load(&tp->stop)) pthread_cond_wait(&tp->cv, &tp->mu);
                atomic_fetch_sub(&tp->sleepers, 1);
                pthread_mutex_unlock(&tp->mu);
                if (cur != st) break;
            }
        }
        st = cur;
    }
}

static void run_phase(HvqParser *p, PhaseFn fn, uint8_t *blob, int ntasks)
{
    struct ParsePool *tp = p->tp;
    if (!tp || ntasks < 2) {
        for (int t = 0; t < ntasks; ++t) fn(p, blob, t);
        return;
    }
    tp->p = p; tp->fn = fn; tp->blob = blob;
    atomic_store_explicit(&tp->done, 0u, memory_order_relaxed);
    const uint64_t ep = (atomic_load(&tp->state) >> 32) + 1u;
    uint64_t st = (ep << 32) | ((uint64_t)ntasks << 8);
    atomic_store(&tp->state, st);                                   /* publishes fn / blob / done with it */
    if (atomic_load(&tp->sleepers) > 0) {
        pthread_mutex_lock(&tp->mu);
        pthread_cond_broadcast(&tp->cv);
        pthread_mutex_unlock(&tp->mu);
    }
    (void)pool_drain(tp, st);
    while (atomic_load_explicit(&tp->done, memory_order_acquire) < (uint32_t)ntasks) cpu_pause();
}

static void pool_destroy(struct ParsePool *tp)
{
    if (!tp) return;
    atomic_store(&tp->stop, 1);
    pthread_mutex_lock(&tp->mu);
    pthread_cond_broadcast(&tp->cv);
    pthread_mutex_unlock(&tp->mu);
    for (int i = 0; i < tp->nworkers; ++i) pthread_join(tp->th[i], NULL);
    pthread_mutex_destroy(&tp->mu);
    pthread_cond_destroy(&tp->cv);
    free(tp);
}

/* `threads` threads parse a picture of this parser (the caller's included; 1 = the caller alone, the default; at most 8).  For callers
 * that decode one picture at a time (the SDK entry points); the batched paths parse pictures side by side instead.  Returns the
 * thread count in effect. */
int hvq_parser_set_threads(HvqParser *p, int threads)
{
    if (!p) return 0;
    if (threads < 1) threads = 1;
    if (threads > 8) threads = 8;
    if (p->tp && p->tp->nworkers == threads - 1) return threads;
    pool_destroy(p->tp);
    p->tp = NULL;
    if (threads == 1) return 1;
    struct ParsePool *tp = (struct ParsePool *)calloc(1, sizeof *tp);
    if (!tp) return 1;
    pthread_mutex_init(&tp->mu, NULL);
    pthread_cond_init(&tp->cv, NULL);
    atomic_init(&tp->state, 0);
    for (int i = 0; i < threads - 1; ++i) {
        if (pthread_create(&tp->th[tp->nworkers], NULL, pool_worker, tp) != 0) break;
        tp->nworkers++;
    }
    if (tp->nworkers == 0) { pthread_mutex_destroy(&tp->mu); pthread_cond_destroy(&tp->cv); free(tp); return 1; }
    p->tp = tp;
    return tp->nworkers + 1;
}

/* every HVQ_F_* flag the last hvq_parse_picture raised, including what a P/B picture's pass 2 raised behind the blob header */
uint32_t hvq_parser_last_flags(const HvqParser *p) { return p ? p->flags : 0u; }

int hvq_parse_picture(HvqParser *p, int frame_type, const uint8_t *pic, size_t len,
                      uint8_t *blob, size_t cap, size_t *blob_len)
{
    if (!p || !pic || !blob || !blob_len) return HVQ_E_ARG;
    if (cap < p->fixed_bytes) return HVQ_E_OVERFLOW;
    if (len && len < 8 + 0x44 + 4) return HVQ_E_ARG;
    p->end = len ? pic + len : (const uint8_t *)UINTPTR_MAX;
    p->flags = 0;
    p->capped = 0;
    memset(p->tflags, 0, sizeof p->tflags);
    if (p->timing) p->t_mark = now_ns();
    switch (frame_type) {
    case 0x10: return parse_ipic(p, pic, blob, cap, blob_len);
    case 0x20: return parse_pbpic(p, 1, pic, blob, cap, blob_len);
    case 0x30: return parse_pbpic(p, 0, pic, blob, cap, blob_len);
    default: return HVQ_E_ARG;
    }
}


/* Length of the picture at `frame`, from its own section table: 8 header bytes, 16 (I) or 17 (P/B) big-endian section
 * offsets, every section = 32-bit size + payload (h4m:1978-1993, 2029-2044).  The SDK signatures carry no length, and the
 * reference itself reads exactly these words unchecked; with the length the parser bounds every later read (zeros beyond).
 * `limit` (readable bytes at `frame`, 0 = unknown) additionally bounds the walk over the table itself. */
int hvq_picture_length(const uint8_t *frame, int frame_type, uint32_t limit, size_t *out)
{
    const uint32_t nsec = frame_type == 0x10 ? 16u : 17u, base = 8u + 4u * nsec;
    uint64_t end = base;
    if (!frame || !out) return HVQ_E_ARG;
    if (limit && limit < base) return HVQ_E_ARG;
    for (uint32_t i = 0; i < nsec; ++i) {
        const uint8_t *q = frame + 8 + 4 * i;
        const uint64_t at = (uint64_t)base + (((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]);
        if (at + 4 > (limit ? (uint64_t)limit : (uint64_t)0x7FFFFFF0u)) return HVQ_E_ARG;
        const uint8_t *z = frame + at;
        const uint64_t size = ((uint32_t)z[0] << 24) | ((uint32_t)z[1] << 16) | ((uint32_t)z[2] << 8) | z[3];
        if (at + 4 + size > end) end = at + 4 + size;
    }
    if (limit && end > limit) end = limit;
    if (end > 0x7FFFFFF0u) return HVQ_E_ARG;
    *out = (size_t)end;
    return HVQ_OK;
}
