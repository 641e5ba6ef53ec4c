/*
 * hvq_gparse_core.h -- GPU-side entropy parse of one HVQM4 picture into the descriptor blob of hvq_desc.h
 * ("GPU-side entropy decode ... sections are independent bit buffers, h4m:1981-1993").
 *
 * Same bitstream semantics as the host parser (hvq_parse.c, which cites the reference lines), re-cut for a
 * workgroup: the reference's -- and the host parser's -- macroblock loops interleave reads from up to 17 bit
 * buffers; because every buffer has its own cursor, the loops split exactly into CHAINS that each own the cursors
 * they read (block kinds per plane group, DC values per plane, payload per plane, motion vectors per component,
 * macroblock type and proc runs), separated by workgroup barriers where one chain needs another's output.  A chain
 * is serial and runs on one lane; chains of one phase run concurrently on different waves/lanes; everything that
 * is per-block and independent (map initialisation, macroblock tags, pool layout, nest) runs on all threads.
 * One workgroup parses one picture; thousands of pictures are in flight per launch.
 *
 * The functions are plain C so that tests/native/gparse_emul.c can run the very same code on the CPU, phase by
 * phase, and compare its blobs with hvq_parse.c byte for byte; hvq_gparse.hip wraps them in the kernel.
 *
 * Deliberate differences from hvq_parse.c (documented in DESIGN.md):
 *   - stateless per picture: the packed nest of an I picture is written to `nest_out`, not into the blob, and P/B
 *     blobs carry no nest copy (the runtime points the job at the governing I picture's nest);
 *   - a prefix tree taken from an EMPTY section yields leaf value 0 (the host parser, like the reference, returns
 *     whatever leaf 0 held from an earlier picture; legal streams never read symbols from such a tree);
 *   - the picture length must be known (the host parser can trust the stream when told len = 0).
 */
#ifndef HVQ_GPARSE_CORE_H
#define HVQ_GPARSE_CORE_H

#include <stdint.h>

#include "hvq_desc.h"

#if defined(__HIPCC__)
#define GP_FN __host__ __device__ static inline
#else
#define GP_FN static inline
#endif

#define GP_LUT_BITS 9
#define GP_MAX_OVF_ITER 65536
#define GP_ALIGN16(x) (((x) + 15u) & ~15u)

/* status bits of HvqParseResult.status */
#define GP_ST_OVERFLOW 1u      /* blob capacity / 22-bit pool offsets exceeded (HVQ_E_OVERFLOW) */
#define GP_ST_BADTREE  2u      /* prefix tree nests deeper than any 256-leaf tree can */
#define GP_ST_BADARG   4u

/* one parse job = one picture (device-visible, 64 bytes) */
typedef struct HvqParseJob {
    uint64_t pic;              /* picture data (after the 4-byte disp_id), 4-byte aligned, zero padded to pic_dwords*4 */
    uint64_t blob;             /* out: descriptor blob, `cap` bytes, 16-byte aligned */
    uint64_t scratch;          /* per-picture scratch, hvq_gparse_scratch_bytes() */
    uint64_t nest_out;         /* I pictures: packed nest, ALIGN16(HVQ_NESTP_BYTES) bytes */
    uint32_t len;              /* picture bytes */
    uint32_t pic_dwords;
    uint32_t cap;
    uint16_t width, height;
    uint8_t frame_type;        /* 0x10 / 0x20 / 0x30 */
    uint8_t h_samp, v_samp, is15;
    uint32_t pad[4];
} HvqParseJob;

typedef struct HvqParseResult {   /* what the host needs back to size and order the reconstruction launches */
    uint32_t status;
    uint32_t flags;
    uint32_t max_items, max_pairs;
    uint32_t pool_dwords;
    uint32_t total_bytes;
    uint32_t pad[2];
} HvqParseResult;

/* ------------------------------------------------------------------ bit reader over aligned dwords */
typedef struct {
    const uint32_t *d;
    uint32_t nd;               /* readable dwords; everything past them reads as zero */
    uint32_t idx;              /* next dword */
    uint64_t acc;              /* left aligned */
    int cnt;
    int live;
} GBits;

GP_FN uint32_t gb_dword(const GBits *b, uint32_t i) { return i < b->nd ? __builtin_bswap32(b->d[i]) : 0u; }

GP_FN void gb_init(GBits *b, const uint32_t *d, uint32_t nd, uint64_t byte_off, int live)
{
    b->d = d; b->nd = nd; b->live = live;
    if (byte_off >= (uint64_t)nd * 4u) { b->idx = nd; b->acc = 0; b->cnt = 32; return; }
    const uint32_t i = (uint32_t)(byte_off >> 2), sh = (uint32_t)(byte_off & 3u) * 8u;
    b->acc = ((uint64_t)gb_dword(b, i) << 32) << sh;
    b->cnt = 32 - (int)sh;
    b->idx = i + 1;
}

GP_FN void gb_refill(GBits *b)
{
    if (b->cnt <= 32) {
        b->acc |= (uint64_t)gb_dword(b, b->idx) << (32 - b->cnt);
        b->idx++;
        b->cnt += 32;
    }
}

GP_FN uint32_t gb_take(GBits *b, int n)        /* n <= 32 */
{
    if (n == 0) return 0;
    gb_refill(b);
    const uint32_t v = (uint32_t)(b->acc >> (64 - n));
    b->acc <<= n;
    b->cnt -= n;
    return v;
}

/* ------------------------------------------------------------------ prefix trees (h4m:385-394, 604-651) */
typedef struct {
    int root;
    uint16_t lut[1 << GP_LUT_BITS];     /* [15:10] bits consumed, [9:0] leaf byte (< 256) or node id; the tree reader's stack */
    uint16_t kid[2][256];               /* children of node ids 256..511 */
    int16_t leaf[256];
} GCode;

/* serial: read the tree that heads `carrier` (h4m:604-642); iterative form of hvq_parse.c code_node */
GP_FN void gc_read(GCode *c, GBits *carrier, int is_signed, int scale, uint32_t *status, uint32_t *flags)
{
    c->root = 0;
    if (!carrier->live) { c->leaf[0] = 0; (void)flags; return; }
    uint16_t *stk = c->lut;
    int sp = 0, next = 0x100;
    for (;;) {
        int val;
        if (gb_take(carrier, 1) == 0) {
            const int byte = (int)gb_take(carrier, 8);
            const int v = (is_signed && byte > 0x7F) ? byte - 256 : byte;
            c->leaf[byte] = (int16_t)((uint32_t)v << scale);          /* int16 truncation: h4m:613-617 */
            val = byte;
        } else {
            /* a tree over 256 leaf bytes has at most 255 inner nodes (ids 256..510) and so nests at most 255 deep;
             * anything more is malformed, and would let node 511 become its own child (an endless walk in gsym) */
            if (next >= 511 || sp >= 256) { *status |= GP_ST_BADTREE; c->root = 0; return; }
            const int id = next++;
            stk[sp++] = (uint16_t)id;
            continue;
        }
        for (;;) {                                  /* hand the finished subtree to its parent */
            if (sp == 0) { c->root = val; return; }
            const uint16_t top = stk[sp - 1];
            const int id = top & 0x3FF;
            if (!(top & 0x8000u)) { c->kid[0][id - 256] = (uint16_t)val; stk[sp - 1] = (uint16_t)(top | 0x8000u); break; }
            c->kid[1][id - 256] = (uint16_t)val;
            --sp;
            val = id;
        }
    }
}

/* parallel: thread `tid` of `nthr` fills its share of the first-level table */
GP_FN void gc_fill_lut(GCode *c, int tid, int nthr)
{
    const int root = c->root;
    for (int e = tid; e < (1 << GP_LUT_BITS); e += nthr) {
        int node = root, d = 0;
        while (node >= 256 && d < GP_LUT_BITS) { node = c->kid[(e >> (GP_LUT_BITS - 1 - d)) & 1][node - 256]; ++d; }
        c->lut[e] = (uint16_t)((d << 10) | node);
    }
}

GP_FN int32_t gsym(const GCode *c, GBits *b)                                   /* h4m:644-651 */
{
    gb_refill(b);
    const uint32_t e = c->lut[b->acc >> (64 - GP_LUT_BITS)];
    const int len = (int)(e >> 10);
    int id = (int)(e & 1023u);
    b->acc <<= len;
    b->cnt -= len;
    while (id >= 256) {
        if (b->cnt == 0) gb_refill(b);
        id = c->kid[b->acc >> 63][id - 256];
        b->acc <<= 1;
        b->cnt--;
    }
    return c->leaf[id];
}

GP_FN int32_t gsym_sovf(const GCode *c, GBits *b, int32_t lo, int32_t hi)      /* h4m:654-664 */
{
    uint32_t total = 0;
    int32_t v;
    int guard = 0;
    do { v = gsym(c, b); total += (uint32_t)v; } while ((v <= lo || v >= hi) && ++guard < GP_MAX_OVF_ITER);
    return (int32_t)total;
}

GP_FN int32_t gsym_uovf(const GCode *c, GBits *b)                               /* h4m:667-677 */
{
    int32_t total = 0, v;
    int guard = 0;
    do { v = gsym(c, b); total += v; } while (v >= 0xFF && ++guard < GP_MAX_OVF_ITER);
    return total;
}

/* ------------------------------------------------------------------ per-picture state (LDS on the device) */
typedef struct {
    int hb, vb, stride;
    int bx_per, by_per, nblk;
    uint32_t nblocks, ntiles;
    uint32_t map_off, plane_off;
    uint32_t run_first;          /* index of the plane's first 64-block run in wave_base[] */
    uint32_t blk_first;          /* index of the plane's first block in blk_off[] */
} GPlane;

enum { GC_BN = 0, GC_RUN, GC_DC, GC_BT, GC_MV, GC_MCB, GC_COUNT };

typedef struct {
    /* job */
    const uint32_t *d;
    uint32_t nd, len, cap;
    uint8_t *blob;
    uint8_t *nest_out;
    int frame_type, is_pb, is_P;
    /* geometry (hvq_parser_create) */
    int w, h, is15, landscape, nest_w, nest_h, wshift, hshift, mw, mh;
    GPlane pl[3];
    uint32_t mv_off, wave_base_off, fixed_bytes, pic_bytes, total_tiles, total_runs, total_blocks;
    /* scratch */
    uint32_t *blk_off;           /* [total_blocks] pool offset of every block */
    uint16_t *run_items;         /* [total_runs] */
    uint16_t *run_pairs;         /* [total_runs] */
    uint8_t *mbtype;             /* [mw*mh] macroblock type 0..3 */
    uint8_t *procseq;            /* [mw*mh] proc value of the n-th inter macroblock */
    uint8_t *mbtag;              /* [mw*mh] (type << 5) | (proc << 4) */
    uint32_t *part;              /* [GP_PART] partial counts of the parallel phases */
    /* picture */
    int dc_shift, unk_shift, nx, ny;
    int32_t dc_lo, dc_hi;
    uint8_t res[8];              /* h0 h1 v0 v1 0 0 (h4m:2023-2026; indexed by reference 0..2 like hvq_parse.c) */
    uint32_t flags, status;
    uint32_t max_items, max_pairs, pool_dwords, total, nest_off;
    GBits bn[2], bnr[2], dc[3], bt[3], rle[3], fx[3], mvh, mvv, mtype, mproc;
} GPic;

/* part[]: [0, 512) one word per thread, [512, 992) a second word per thread, the rest single-purpose slots */
#define GP_PART 1024
#define GP_PART2 512
#define GP_MISC 992
#define GP_MAX_THREADS 480

GP_FN uint32_t gp_be32(const GPic *g, uint64_t off)
{
    if (off + 4 > g->len) return 0;
    const uint32_t i = (uint32_t)(off >> 2), sh = (uint32_t)(off & 3u) * 8u;
    const uint32_t w0 = __builtin_bswap32(g->d[i]);
    if (!sh) return w0;
    const uint32_t w1 = i + 1 < g->nd ? __builtin_bswap32(g->d[i + 1]) : 0u;
    return (w0 << sh) | (w1 >> (32 - sh));
}

GP_FN uint32_t gp_byte(const GPic *g, uint32_t off) { return (gp_be32(g, off & ~3u) >> (24 - 8 * (off & 3u))) & 0xFFu; }

/* bytes of scratch one picture of this geometry needs */
GP_FN uint32_t gp_scratch_bytes(uint32_t total_blocks, uint32_t total_runs, uint32_t nmb)
{
    return GP_ALIGN16(4u * total_blocks) + GP_ALIGN16(2u * total_runs) * 2u + GP_ALIGN16(nmb) * 3u + 4u * GP_PART;
}

/* serial (thread 0): geometry exactly as hvq_parser_create lays the blob out */
GP_FN void gp_setup(GPic *g, const HvqParseJob *job)
{
    g->d = (const uint32_t *)(uintptr_t)job->pic;
    g->nd = job->pic_dwords; g->len = job->len; g->cap = job->cap;
    g->blob = (uint8_t *)(uintptr_t)job->blob;
    g->nest_out = (uint8_t *)(uintptr_t)job->nest_out;
    g->frame_type = job->frame_type;
    g->is_pb = job->frame_type != 0x10;
    g->is_P = job->frame_type == 0x20;
    g->w = job->width; g->h = job->height; g->is15 = job->is15 != 0;
    g->landscape = g->w >= g->h;                                   /* h4m:965-975 */
    g->nest_w = g->landscape ? 70 : 38;
    g->nest_h = g->landscape ? 38 : 70;
    g->wshift = job->h_samp == 2; g->hshift = job->v_samp == 2;
    g->mw = g->w / 8; g->mh = g->h / 8;
    uint32_t off = (uint32_t)sizeof(HvqPicHeader), poff = 0, blocks = 0, runs = 0;
    g->total_tiles = 0;
    for (int i = 0; i < 3; ++i) {
        GPlane *q = &g->pl[i];
        const int ws = i ? g->wshift : 0, hs = i ? g->hshift : 0;
        q->hb = (g->w >> ws) / 4; q->vb = (g->h >> hs) / 4;        /* h4m:856-857 */
        q->stride = q->hb + 2;
        q->bx_per = 2 >> ws; q->by_per = 2 >> hs; q->nblk = q->bx_per * q->by_per;
        q->nblocks = (uint32_t)q->hb * (uint32_t)q->vb;
        q->ntiles = (q->nblocks + HVQ_TILE_BLOCKS - 1) / HVQ_TILE_BLOCKS;
        q->run_first = runs; q->blk_first = blocks;
        runs += q->ntiles * (HVQ_TILE_BLOCKS / 64);
        g->total_tiles += q->ntiles;
        q->map_off = off;
        off = GP_ALIGN16(off + 2u * (uint32_t)q->stride * (uint32_t)(q->vb + 2));
        q->plane_off = poff;
        poff += (uint32_t)(g->w >> ws) * (uint32_t)(g->h >> hs);
        blocks += q->nblocks;
    }
    g->pic_bytes = poff; g->total_runs = runs; g->total_blocks = blocks;
    g->mv_off = off;
    off = GP_ALIGN16(off + 4u * (uint32_t)g->mw * (uint32_t)g->mh);
    g->wave_base_off = off;
    off = GP_ALIGN16(off + 4u * g->total_tiles * (HVQ_TILE_BLOCKS / 64));
    g->fixed_bytes = off;
    const uint32_t nmb = (uint32_t)g->mw * (uint32_t)g->mh;
    uint8_t *s = (uint8_t *)(uintptr_t)job->scratch;
    g->blk_off = (uint32_t *)s;          s += GP_ALIGN16(4u * blocks);
    g->run_items = (uint16_t *)s;        s += GP_ALIGN16(2u * runs);
    g->run_pairs = (uint16_t *)s;        s += GP_ALIGN16(2u * runs);
    g->mbtype = s;                       s += GP_ALIGN16(nmb);
    g->procseq = s;                      s += GP_ALIGN16(nmb);
    g->mbtag = s;                        s += GP_ALIGN16(nmb);
    g->part = (uint32_t *)s;
    g->flags = 0; g->status = 0; g->max_items = 0; g->max_pairs = 0; g->pool_dwords = 0; g->total = 0; g->nest_off = 0;
    if (g->cap < g->fixed_bytes || g->len < 8 + 0x44 + 4) g->status |= GP_ST_BADARG;
}

/* sections (h4m:1061-1071, 1979-1993, 2030-2044): byte offset of the payload of section i, *live as in hvq_parse.c */
GP_FN uint64_t gp_section(const GPic *g, uint32_t data_off, uint32_t tab_off, int i, int *live)
{
    const uint64_t s = (uint64_t)data_off + gp_be32(g, tab_off + 4u * (uint32_t)i);
    if (s + 4 > g->len) { *live = 0; return g->len; }
    *live = gp_be32(g, s) != 0;
    return s + 4;
}

GP_FN void gp_section_bits(const GPic *g, GBits *b, uint32_t data_off, uint32_t tab_off, int i)
{
    int live;
    const uint64_t s = gp_section(g, data_off, tab_off, i, &live);
    gb_init(b, g->d, g->nd, s, live);
}

/* serial (thread 0): picture header fields and all section cursors */
GP_FN void gp_sections(GPic *g)
{
    if (g->status) return;
    g->dc_shift = (int)gp_byte(g, 0);
    g->unk_shift = (int)gp_byte(g, 1);
    const uint32_t tab = 8, data = g->is_pb ? 8 + 0x44 : 8 + 0x40;
    if (g->is_pb) {
        g->res[0] = (uint8_t)gp_byte(g, 2); g->res[1] = (uint8_t)gp_byte(g, 4);
        g->res[2] = (uint8_t)gp_byte(g, 3); g->res[3] = (uint8_t)gp_byte(g, 5);
        g->res[4] = g->res[5] = g->res[6] = g->res[7] = 0;
        g->nx = g->ny = 0;
    } else {
        g->nx = (int)(gp_be32(g, 4) >> 16); g->ny = (int)(gp_be32(g, 4) & 0xFFFFu);
    }
    for (int i = 0; i < 2; ++i) {
        gp_section_bits(g, &g->bn[i], data, tab, 2 * i);
        gp_section_bits(g, &g->bnr[i], data, tab, 2 * i + 1);
    }
    for (int k = 0; k < 3; ++k) {
        gp_section_bits(g, &g->dc[k], data, tab, 4 + 3 * k);
        gp_section_bits(g, &g->bt[k], data, tab, 5 + 3 * k);
        gp_section_bits(g, &g->fx[k], data, tab, 6 + 3 * k);
    }
    if (g->is_pb) {
        gp_section_bits(g, &g->mvh, data, tab, 13);
        gp_section_bits(g, &g->mvv, data, tab, 14);
        gp_section_bits(g, &g->mtype, data, tab, 15);
        gp_section_bits(g, &g->mproc, data, tab, 16);
    } else {
        for (int k = 0; k < 3; ++k) gp_section_bits(g, &g->rle[k], data, tab, 13 + k);
    }
    g->dc_hi = (int32_t)((uint32_t)0x7F << (g->dc_shift & 31));
    g->dc_lo = (int32_t)((uint32_t)-0x80 << (g->dc_shift & 31));
}

/* serial, one call per tree `t` (any lane): h4m:1996-1999 / 2045-2050 */
GP_FN void gp_read_tree(GPic *g, GCode *codes, int t)
{
    if (g->status) return;
    uint32_t st = 0, fl = 0;
    switch (t) {
    case GC_BN:  gc_read(&codes[GC_BN], &g->bn[0], 0, 0, &st, &fl); break;
    case GC_RUN: gc_read(&codes[GC_RUN], &g->bnr[0], 0, 0, &st, &fl); break;
    case GC_DC:  gc_read(&codes[GC_DC], &g->dc[0], 1, g->dc_shift & 31, &st, &fl); break;
    case GC_BT:  gc_read(&codes[GC_BT], &g->bt[0], 0, 2, &st, &fl); break;
    case GC_MV:  gc_read(&codes[GC_MV], &g->mvh, 1, 0, &st, &fl); break;
    default:     gc_read(&codes[GC_MCB], &g->mtype, 0, 0, &st, &fl); break;
    }
    /* one word per tree: no two lanes update the same location */
    g->part[GP_MISC + t] = st; g->part[GP_MISC + GC_COUNT + t] = fl;
}

GP_FN void gp_collect_tree_status(GPic *g, int ntrees)      /* serial (thread 0), after the tree reads */
{
    if (g->status) return;
    for (int t = 0; t < ntrees; ++t) { g->status |= g->part[GP_MISC + t]; g->flags |= g->part[GP_MISC + GC_COUNT + t]; }
}

/* ------------------------------------------------------------------ blob helpers */
GP_FN uint8_t *gp_map_ent(const GPic *g, int plane, int by, int bx)
{
    return g->blob + g->pl[plane].map_off + 2u * ((uint32_t)(by + 1) * (uint32_t)g->pl[plane].stride + (uint32_t)(bx + 1));
}

/* parallel: maps = zero inside, {0x7F,0xFF} on the border (h4m:951-955, 1001-1040); P/B: motion vectors = 0 */
GP_FN void gp_init_maps(const GPic *g, int tid, int nthr)
{
    if (g->status) return;
    for (int i = 0; i < 3; ++i) {
        const GPlane *q = &g->pl[i];
        uint16_t *m = (uint16_t *)(g->blob + q->map_off);
        const uint32_t n = (uint32_t)q->stride * (uint32_t)(q->vb + 2);
        for (uint32_t e = (uint32_t)tid; e < n; e += (uint32_t)nthr) {
            const uint32_t r = e / (uint32_t)q->stride, c = e - r * (uint32_t)q->stride;
            const int border = r == 0 || r == (uint32_t)q->vb + 1 || c == 0 || c == (uint32_t)q->stride - 1;
            m[e] = border ? (uint16_t)0xFF7Fu : (uint16_t)0;
        }
    }
    if (g->is_pb) {
        uint32_t *mv = (uint32_t *)(g->blob + g->mv_off);
        const uint32_t n = (uint32_t)g->mw * (uint32_t)g->mh;
        for (uint32_t e = (uint32_t)tid; e < n; e += (uint32_t)nthr) mv[e] = 0;
    }
}

/* ------------------------------------------------------------------ I picture chains */
/* chain: block kinds of the luma plane (which = 0) or of both chroma planes (which = 1); h4m:1073-1130 */
GP_FN void gp_ikinds(GPic *g, const GCode *codes, int which)
{
    if (g->status) return;
    GBits bn = g->bn[which], bnr = g->bnr[which];
    const GCode *c_bn = &codes[GC_BN], *c_run = &codes[GC_RUN];
    uint32_t run = 0;
    if (which == 0) {
        const GPlane *Y = &g->pl[0];
        for (int by = 0; by < Y->vb; ++by) {
            uint8_t *row = gp_map_ent(g, 0, by, 0);
            for (int bx = 0; bx < Y->hb; ++bx) {
                if (run) { --run; continue; }
                const int32_t k = gsym(c_bn, &bn) & 0xFFFF;
                if ((int16_t)k == 0) run = (uint32_t)gsym(c_run, &bnr);
                else row[2 * bx + 1] = (uint8_t)k;
            }
        }
    } else {
        const GPlane *C = &g->pl[1];
        for (int by = 0; by < C->vb; ++by) {
            uint8_t *ru = gp_map_ent(g, 1, by, 0), *rv = gp_map_ent(g, 2, by, 0);
            for (int bx = 0; bx < C->hb; ++bx) {
                if (run) { --run; continue; }
                const int32_t k = gsym(c_bn, &bn) & 0xFFFF;
                if ((int16_t)k == 0) run = (uint32_t)gsym(c_run, &bnr);
                else { ru[2 * bx + 1] = (uint8_t)(k & 0xF); rv[2 * bx + 1] = (uint8_t)((k >> 4) & 0xF); }
            }
        }
    }
}

/* chain: DC values of plane i (h4m:1043-1058, 1132-1164).  `rowbuf` (hb + 1 bytes, private to the chain) holds the
 * row above so that the prediction never waits for the map in HBM. */
GP_FN void gp_idc(GPic *g, const GCode *codes, int i, uint8_t *rowbuf)
{
    if (g->status) return;
    const GPlane *q = &g->pl[i];
    GBits dc = g->dc[i], rle = g->rle[i];
    const GCode *c_dc = &codes[GC_DC], *c_run = &codes[GC_RUN];
    const int32_t lo = g->dc_lo, hi = g->dc_hi;
    for (int bx = 0; bx <= q->hb; ++bx) rowbuf[bx] = 0x7F;
    uint32_t run = 0;
    for (int by = 0; by < q->vb; ++by) {
        uint8_t *row = gp_map_ent(g, i, by, 0);
        uint8_t pred = by ? rowbuf[0] : 0x7F;
        for (int bx = 0; bx < q->hb; ++bx) {
            uint32_t delta = 0;
            if (run) --run;
            else {
                delta = (uint32_t)gsym_sovf(c_dc, &dc, lo, hi);
                if (delta == 0) run = (uint32_t)gsym(c_run, &rle);
            }
            const uint8_t v = (uint8_t)(pred + delta);                  /* uint8 wrap: h4m:1145-1149 */
            row[2 * bx] = v;
            pred = (uint8_t)((v + rowbuf[bx + 1] + 1) / 2);
            rowbuf[bx] = v;
        }
    }
}

/* parallel: nest from the luma DC values (h4m:1166-1239), nibble-packed as hvq_parse.c pack_nest */
GP_FN void gp_nest(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    const GPlane *Y = &g->pl[0];
    const int cols = Y->hb < g->nest_w ? Y->hb : g->nest_w;
    const int rows = Y->vb < g->nest_h ? Y->vb : g->nest_h;
    int mcols = g->nest_w - cols; if (mcols > cols) mcols = cols;
    int mrows = g->nest_h - rows; if (mrows > rows) mrows = rows;
    int nx = g->nx, ny = g->ny, clamped = 0;
    if (nx + cols > Y->hb) { nx = Y->hb - cols; clamped = 1; }
    if (ny + rows > Y->vb) { ny = Y->vb - rows; clamped = 1; }
    if (tid == 0) g->part[GP_MISC + 2 * GC_COUNT] = clamped ? HVQ_F_CLAMPED : 0u;
    const int nbytes = (int)GP_ALIGN16(HVQ_NESTP_BYTES);
    for (int o = tid; o < nbytes; o += nthr) {
        uint32_t byte = 0;
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int idx = 2 * o + hlf;
            uint32_t v = 0;
            if (idx < HVQ_NEST_BYTES) {
                const int r = idx / g->nest_w, cidx = idx - r * g->nest_w;
                const int rr = r < rows ? r : (r < rows + mrows ? rows - 1 - (r - rows) : -1);
                const int cc = cidx < cols ? cidx : (cidx < cols + mcols ? cols - 1 - (cidx - cols) : -1);
                if (rr >= 0 && cc >= 0) v = (uint32_t)(gp_map_ent(g, 0, ny + rr, nx + cc)[0] >> 4) & 0xFu;
            }
            byte |= v << (4 * hlf);
        }
        g->nest_out[o] = (uint8_t)byte;
    }
}

/* ------------------------------------------------------------------ pool layout (hvq_parse.c layout_pool) */
GP_FN void gp_type_info(int ctx, uint32_t t, uint32_t *n, uint32_t *item, uint32_t *pairs, uint32_t *flags)
{
    const int is_pb = ctx == 2, il = ctx == 0;
    *n = hvq_payload_dwords(t, is_pb, il);
    const int inter = is_pb && (t & 0x60u);
    const uint32_t kind = il ? t : (t & 0xFu);
    *item = 0; *pairs = 0; *flags = 0;
    if (*n && kind != 6) {
        *item = 1;
        *pairs = inter ? kind - 1 : kind;
        if (!inter) *flags = HVQ_F_HAS_NEST | (kind > 15 ? HVQ_F_BIG_AOT : 0u);
    }
}

/* which plane a global run index belongs to */
GP_FN int gp_run_plane(const GPic *g, uint32_t r) { return r >= g->pl[2].run_first ? 2 : (r >= g->pl[1].run_first ? 1 : 0); }

/* parallel L1: per 64-block run, payload dwords (into wave_base[]), queued blocks, pairs; flags into part[] */
GP_FN void gp_layout_sum(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    uint32_t *wave_base = (uint32_t *)(g->blob + g->wave_base_off);
    uint32_t fl = 0;
    for (uint32_t r = (uint32_t)tid; r < g->total_runs; r += (uint32_t)nthr) {
        const int i = gp_run_plane(g, r);
        const GPlane *q = &g->pl[i];
        const int ctx = g->is_pb ? 2 : (i == 0 ? 0 : 1);
        const uint32_t b0 = (r - q->run_first) * 64u;
        uint32_t sum = 0, items = 0, pairs = 0;
        uint32_t by = b0 / (uint32_t)q->hb, bx = b0 - by * (uint32_t)q->hb;
        for (uint32_t b = b0; b < b0 + 64u && b < q->nblocks; ++b) {
            uint32_t n, it, pr, f;
            gp_type_info(ctx, gp_map_ent(g, i, (int)by, (int)bx)[1], &n, &it, &pr, &f);
            sum += n; items += it; pairs += pr; fl |= f;
            if (++bx == (uint32_t)q->hb) { bx = 0; ++by; }
        }
        wave_base[r] = sum;
        g->run_items[r] = (uint16_t)items;
        g->run_pairs[r] = (uint16_t)pairs;
    }
    g->part[tid] = fl;
}

/* serial L2 (thread 0): exclusive scan of the runs, per-tile maxima, sizes, overflow check, header */
GP_FN void gp_layout_scan(GPic *g, int nthr)
{
    if (g->status) return;
    uint32_t *wave_base = (uint32_t *)(g->blob + g->wave_base_off);
    uint32_t off = 0, fl = 0, mi = 0, mp = 0, ti = 0, tp = 0;
    for (int t = 0; t < nthr; ++t) fl |= g->part[t];
    for (uint32_t r = 0; r < g->total_runs; ++r) {
        if (r % (HVQ_TILE_BLOCKS / 64) == 0) { ti = 0; tp = 0; }
        const uint32_t s = wave_base[r];
        wave_base[r] = off;
        off += s;
        ti += g->run_items[r]; tp += g->run_pairs[r];
        if (ti > mi) mi = ti;
        if (tp > mp) mp = tp;
    }
    if (g->is_pb) { for (int t = 0; t < nthr; ++t) fl |= g->part[GP_PART2 + t]; }      /* HVQ_F_SELF_REF, gp_tags_assign */
    else fl |= g->part[GP_MISC + 2 * GC_COUNT];                                      /* nest origin clamp, gp_nest */
    g->flags |= fl;
    g->max_items = mi; g->max_pairs = mp; g->pool_dwords = off;
    uint64_t total = (uint64_t)g->fixed_bytes + 4u * (uint64_t)off;
    total = GP_ALIGN16(total);
    if (total > g->cap || off >= (1u << 22)) { g->status |= GP_ST_OVERFLOW; return; }
    g->total = (uint32_t)total;
    /* header (hvq_parse.c fill_header); nest_off stays 0: the nest travels separately */
    HvqPicHeader *h = (HvqPicHeader *)g->blob;
    uint32_t *hw = (uint32_t *)g->blob;
    for (int k = 0; k < (int)(sizeof(HvqPicHeader) / 4); ++k) hw[k] = 0;
    h->magic = HVQ_MAGIC;
    h->total_bytes = g->total;
    h->width = (uint16_t)g->w; h->height = (uint16_t)g->h;
    h->pic_kind = (uint8_t)(g->is_pb ? (g->is_P ? HVQ_PIC_P : HVQ_PIC_B) : HVQ_PIC_I);
    h->unk_shift = (uint8_t)g->unk_shift;
    h->dc_shift = (uint8_t)g->dc_shift;
    h->wshift = (uint8_t)g->wshift; h->hshift = (uint8_t)g->hshift;
    h->flags = g->flags | (g->is15 ? HVQ_F_IS15 : 0u) | (g->landscape ? HVQ_F_LANDSCAPE : 0u);
    uint32_t t = 0;
    for (int i = 0; i < 3; ++i) {
        h->hb[i] = (uint16_t)g->pl[i].hb; h->vb[i] = (uint16_t)g->pl[i].vb;
        h->plane_off[i] = g->pl[i].plane_off;
        h->map_off[i] = g->pl[i].map_off;
        h->tile_first[i] = t;
        t += g->pl[i].ntiles;
    }
    h->tile_first[3] = t;
    h->pic_bytes = g->pic_bytes;
    h->mv_off = g->is_pb ? g->mv_off : 0;
    h->wave_base_off = g->wave_base_off;
    h->pool_off = g->fixed_bytes;
    h->pool_dwords = off;
    h->nest_off = 0;
    h->mcb_w = (uint32_t)g->mw; h->mcb_h = (uint32_t)g->mh;
    h->max_items = (uint16_t)mi; h->max_pairs = mp;
}

/* parallel L3: pool offset of every block */
GP_FN void gp_layout_blocks(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    const uint32_t *wave_base = (const uint32_t *)(g->blob + g->wave_base_off);
    for (uint32_t r = (uint32_t)tid; r < g->total_runs; r += (uint32_t)nthr) {
        const int i = gp_run_plane(g, r);
        const GPlane *q = &g->pl[i];
        const int ctx = g->is_pb ? 2 : (i == 0 ? 0 : 1);
        const uint32_t b0 = (r - q->run_first) * 64u;
        uint32_t off = wave_base[r];
        uint32_t by = b0 / (uint32_t)q->hb, bx = b0 - by * (uint32_t)q->hb;
        for (uint32_t b = b0; b < b0 + 64u && b < q->nblocks; ++b) {
            uint32_t n, it, pr, f;
            gp_type_info(ctx, gp_map_ent(g, i, (int)by, (int)bx)[1], &n, &it, &pr, &f);
            g->blk_off[q->blk_first + b] = off;
            off += n;
            if (++bx == (uint32_t)q->hb) { bx = 0; ++by; }
        }
    }
}

/* ------------------------------------------------------------------ payloads */
/* `n` bases of one block: word from fixvl, coefficient from bufTree0 (h4m:691-692, 726-731 / 738-739, 767-772) */
GP_FN void gp_emit_bases(const GCode *c_bt, GBits *fx, GBits *bt, uint32_t n, uint32_t *dst)
{
    uint32_t run = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t word = gb_take(fx, 16);
        run += (uint32_t)gsym(c_bt, bt);
        dst[k] = HVQ_BASIS(word, (run + ((word >> 13) & 3u)) & 0x3FFFFu);
    }
}

GP_FN void gp_literal(GBits *fx, uint32_t *dst)                                 /* h4m:543-549 */
{
    for (int k = 0; k < 4; ++k) dst[k] = __builtin_bswap32(gb_take(fx, 32));
}

/* chain: payloads of plane i of an I picture, raster order == consumption order (h4m:2011-2015) */
GP_FN void gp_ipayload(GPic *g, const GCode *codes, int i)
{
    if (g->status) return;
    const GPlane *q = &g->pl[i];
    GBits fx = g->fx[i], bt = g->bt[i];
    const GCode *c_bt = &codes[GC_BT];
    uint32_t *pool = (uint32_t *)(g->blob + g->fixed_bytes);
    uint32_t off = ((const uint32_t *)(g->blob + g->wave_base_off))[q->run_first];
    for (int by = 0; by < q->vb; ++by) {
        const uint8_t *row = gp_map_ent(g, i, by, 0);
        for (int bx = 0; bx < q->hb; ++bx) {
            const uint32_t k = row[2 * bx + 1];
            if (k == 0 || k == 8) continue;
            if (k == 6) { gp_literal(&fx, pool + off); off += 4; }
            else { gp_emit_bases(c_bt, &fx, &bt, k, pool + off); off += k; }
        }
    }
}

/* ------------------------------------------------------------------ P/B picture chains */
/* chain: macroblock types from the mtype runs (h4m:1545-1622), then the proc runs */
GP_FN void gp_mbtypes(GPic *g, const GCode *codes)
{
    if (g->status) return;
    GBits b = g->mtype;
    const GCode *c = &codes[GC_MCB];
    uint32_t value = 0, count = 0, inter = 0;
    if (b.live) { value = gb_take(&b, 2); count = (uint32_t)gsym_uovf(c, &b); }
    const uint32_t n = (uint32_t)g->mw * (uint32_t)g->mh;
    for (uint32_t m = 0; m < n; ++m) {
        if (count == 0) {
            const uint32_t bit = gb_take(&b, 1);
            const uint32_t v = value & 3u;
            /* step table { {1,2,0,2}, {2,0,1,0} } of hvq_parse.c pb_pass1 */
            value = bit ? (v == 0 ? 2u : (v == 2 ? 1u : 0u)) : (v == 0 ? 1u : (v == 2 ? 0u : 2u));
            count = (uint32_t)gsym_uovf(c, &b);
        }
        --count;
        g->mbtype[m] = (uint8_t)value;
        inter += value != 0;
    }
    /* proc value of the n-th inter macroblock from the mproc runs (h4m:1649-1668); same lane, so that exactly as
     * many runs are read as the picture has inter macroblocks */
    b = g->mproc;
    value = 0; count = 0;
    if (b.live) { value = gb_take(&b, 1); count = (uint32_t)gsym_uovf(c, &b); }
    for (uint32_t m = 0; m < inter; ++m) {
        if (count == 0) { value ^= 1u; count = (uint32_t)gsym_uovf(c, &b); }
        --count;
        g->procseq[m] = (uint8_t)value;
    }
}

/* parallel T1: inter macroblocks per thread chunk */
GP_FN void gp_tags_count(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    const uint32_t n = (uint32_t)g->mw * (uint32_t)g->mh;
    const uint32_t per = (n + (uint32_t)nthr - 1) / (uint32_t)nthr;
    const uint32_t lo = per * (uint32_t)tid, hi = lo + per < n ? lo + per : n;
    uint32_t cnt = 0;
    for (uint32_t m = lo; m < hi; ++m) cnt += g->mbtype[m] != 0;
    g->part[tid] = cnt;
}

/* serial T2 (thread 0): exclusive scan of the chunk counts */
GP_FN void gp_tags_scan(GPic *g, int nthr)
{
    if (g->status) return;
    uint32_t run = 0;
    for (int t = 0; t < nthr; ++t) { const uint32_t c = g->part[t]; g->part[t] = run; run += c; }
}

/* parallel T3: tag of every macroblock; proc-1 macroblocks get their tag into all block types (h4m:1670-1690) */
GP_FN void gp_tags_assign(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    const uint32_t n = (uint32_t)g->mw * (uint32_t)g->mh;
    const uint32_t per = (n + (uint32_t)nthr - 1) / (uint32_t)nthr;
    const uint32_t lo = per * (uint32_t)tid, hi = lo + per < n ? lo + per : n;
    uint32_t rank = g->part[tid], fl = 0;
    for (uint32_t m = lo; m < hi; ++m) {
        const uint32_t type = g->mbtype[m];
        uint32_t tag = 0;
        if (type) {
            const uint32_t proc = g->procseq[rank++];
            tag = (type << 5) | (proc << 4);
            if (g->is_P && type >= 2) fl |= HVQ_F_SELF_REF;
            if (proc) {
                const int my = (int)(m / (uint32_t)g->mw), mx = (int)(m - (uint32_t)my * (uint32_t)g->mw);
                for (int i = 0; i < 3; ++i) {
                    const GPlane *q = &g->pl[i];
                    for (int dy = 0; dy < q->by_per; ++dy)
                        for (int dx = 0; dx < q->bx_per; ++dx)
                            gp_map_ent(g, i, my * q->by_per + dy, mx * q->bx_per + dx)[1] = (uint8_t)tag;
                }
            }
        }
        g->mbtag[m] = (uint8_t)tag;
    }
    g->part[GP_PART2 + tid] = fl;
}

/* block j of a macroblock, order TL, BL, BR, TR (h4m:447-455, 862-865) */
GP_FN int gp_dx(int j) { return j >> 1; }
GP_FN int gp_dy(int j) { return (j == 1 || j == 2) ? 1 : 0; }

/* chain: block kinds of the luma plane (which = 0) or both chroma planes (which = 1) of a P/B picture (h4m:1692-1740) */
GP_FN void gp_pbkinds(GPic *g, const GCode *codes, int which)
{
    if (g->status) return;
    GBits bn = g->bn[which], bnr = g->bnr[which];
    const GCode *c_bn = &codes[GC_BN], *c_run = &codes[GC_RUN];
    const GPlane *q = &g->pl[which];
    uint32_t rl = 0, m = 0;
    for (int my = 0; my < g->mh; ++my)
        for (int mx = 0; mx < g->mw; ++mx, ++m) {
            const uint32_t tag = g->mbtag[m];
            if (tag & 0x10u) continue;                                  /* proc 1: done by gp_tags_assign */
            for (int j = 0; j < q->nblk; ++j) {
                const int by = my * q->by_per + gp_dy(j), bx = mx * q->bx_per + gp_dx(j);
                uint32_t tu = tag, tv = tag;
                if (rl) --rl;
                else {
                    const int16_t k = (int16_t)gsym(c_bn, &bn);
                    if (k == 0) rl = (uint32_t)gsym(c_run, &bnr);
                    else if (which == 0) tu = tag | (uint32_t)k;
                    else { tu = tag | ((uint32_t)k & 0xFu); tv = tag | (((uint32_t)k >> 4) & 0xFu); }
                }
                if (which == 0) gp_map_ent(g, 0, by, bx)[1] = (uint8_t)tu;
                else { gp_map_ent(g, 1, by, bx)[1] = (uint8_t)tu; gp_map_ent(g, 2, by, bx)[1] = (uint8_t)tv; }
            }
        }
}

/* chain: DC values of the intra macroblocks of plane i (h4m:1742-1776); leaves the cursor for the payload chain */
GP_FN void gp_pbdc(GPic *g, const GCode *codes, int i)
{
    if (g->status) return;
    const GPlane *q = &g->pl[i];
    GBits dc = g->dc[i];
    const GCode *c_dc = &codes[GC_DC];
    const int32_t lo = g->dc_lo, hi = g->dc_hi;
    uint32_t pbdc = 0x7F, m = 0;
    for (int my = 0; my < g->mh; ++my)
        for (int mx = 0; mx < g->mw; ++mx, ++m) {
            if (g->mbtype[m]) { pbdc = 0x7F; continue; }
            for (int j = 0; j < q->nblk; ++j) {
                pbdc += (uint32_t)gsym_sovf(c_dc, &dc, lo, hi);
                gp_map_ent(g, i, my * q->by_per + gp_dy(j), mx * q->bx_per + gp_dx(j))[0] = (uint8_t)pbdc;
            }
        }
    g->dc[i] = dc;
}

/* chain: payloads of plane i of a P/B picture, macroblock order (h4m:1789-1827, 1862-1910, 1919-1967) */
GP_FN void gp_pbpayload(GPic *g, const GCode *codes, int i)
{
    if (g->status) return;
    const GPlane *q = &g->pl[i];
    GBits fx = g->fx[i], bt = g->bt[i], dc = g->dc[i];
    const GCode *c_bt = &codes[GC_BT], *c_dc = &codes[GC_DC];
    const int32_t lo = g->dc_lo, hi = g->dc_hi;
    const int sh_dc = g->dc_shift & 31, sh_unk = g->unk_shift & 31;
    uint32_t *pool = (uint32_t *)(g->blob + g->fixed_bytes);
    const uint32_t *blk_off = g->blk_off + q->blk_first;
    uint32_t m = 0;
    for (int my = 0; my < g->mh; ++my)
        for (int mx = 0; mx < g->mw; ++mx, ++m) {
            const uint32_t tag = g->mbtag[m];
            const int inter = (tag & 0x60u) != 0;
            if (tag & 0x10u) continue;                                  /* proc 1: plain MC, no payload (h4m:1327-1355) */
            for (int j = 0; j < q->nblk; ++j) {
                const int by = my * q->by_per + gp_dy(j), bx = mx * q->bx_per + gp_dx(j);
                const uint32_t k = gp_map_ent(g, i, by, bx)[1] & 0xFu;
                if (k == 0 || (!inter && k == 8)) continue;
                uint32_t *dst = pool + blk_off[(uint32_t)by * (uint32_t)q->hb + (uint32_t)bx];
                if (k == 6) { gp_literal(&fx, dst); continue; }
                if (!inter) { gp_emit_bases(c_bt, &fx, &bt, k, dst); continue; }
                gp_emit_bases(c_bt, &fx, &bt, k - 1, dst + 2);
                const int32_t s1 = gsym_sovf(c_dc, &dc, lo, hi);        /* h4m:1405-1406 */
                const int32_t s2 = gsym_sovf(c_dc, &dc, lo, hi);
                dst[0] = (uint32_t)(s1 >> sh_dc) << sh_unk;
                dst[1] = (uint32_t)(s2 >> sh_dc);
            }
        }
}

/* chain: one motion-vector component (comp 0: x from mvh, 1: y from mvv) of every inter macroblock
 * (h4m:1846-1860, 1943-1955); returns HVQ_F_CLAMPED when a target had to be clamped to int16 */
GP_FN uint32_t gp_mvs(GPic *g, const GCode *codes, int comp)
{
    if (g->status) return 0;
    GBits b = comp ? g->mvv : g->mvh;
    const GCode *c = &codes[GC_MV];
    int16_t *mvs = (int16_t *)(g->blob + g->mv_off);
    int cur_ref = -1;
    int32_t acc = 0;
    uint32_t fl = 0, m = 0;
    for (int my = 0; my < g->mh; ++my)
        for (int mx = 0; mx < g->mw; ++mx, ++m) {
            const int t = g->mbtype[m];
            if (t == 0) continue;
            const int r = t - 1;
            if (r != cur_ref) { cur_ref = r; acc = 0; }
            const int rbits = g->res[2 * comp + r] & 15;                  /* r = 2 only from a first type value of 3 */
            const int32_t lim = (int32_t)(1u << (rbits + 5));
            int32_t v = (int32_t)((uint32_t)gsym(c, &b) << rbits);
            v += (int32_t)gb_take(&b, rbits);
            acc += v;
            if (acc >= lim) acc -= lim << 1;
            else if (acc < -lim) acc += lim << 1;
            int32_t pos = (comp ? my : mx) * 16 + acc;
            if (pos > 32767) { pos = 32767; fl |= HVQ_F_CLAMPED; }
            if (pos < -32768) { pos = -32768; fl |= HVQ_F_CLAMPED; }
            mvs[2 * m + (uint32_t)comp] = (int16_t)pos;
        }
    return fl;
}

/* serial (thread 0): result record */
GP_FN void gp_result(const GPic *g, HvqParseResult *out, uint32_t extra_flags)
{
    out->status = g->status;
    out->flags = g->flags | extra_flags | (g->is15 ? HVQ_F_IS15 : 0u) | (g->landscape ? HVQ_F_LANDSCAPE : 0u);
    out->max_items = g->max_items; out->max_pairs = g->max_pairs;
    out->pool_dwords = g->pool_dwords; out->total_bytes = g->total;
    out->pad[0] = out->pad[1] = 0;
}

#endif
