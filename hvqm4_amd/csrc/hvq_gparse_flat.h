/*
 * hvq_gparse_flat.h -- the FLAT path of the GPU entropy parse (round 2): every prefix-coded section is decoded front to
 * back into a flat symbol array, all sections at once and independently of each other, and what the chains of
 * hvq_gparse_core.h did symbol by symbol -- zero runs, overflow grouping, DC prediction, running coefficient sums --
 * becomes scans over those arrays by all threads.
 *
 * The symbol arrays hold LEAF BYTES (round 4): a lane's table entry is 16 bits -- bits consumed, leaf flag, leaf byte or inner
 * node -- which lets the block-kind tree have a 10-bit table and the others 9-bit ones in the LDS the 8-bit tables with 32-bit
 * entries took, and the value of a leaf (h4m:613-617) is computed by whoever reads the array: 64 lanes at a time there, a
 * dozen in the decode wave.
 *
 * Why this is exact: a section's symbol boundaries depend on nothing but its own bits and its tree (h4m:644-651); what
 * the reference's loops decide from OTHER sections is only how many symbols of a section are consumed and what they
 * mean.  (The exceptions keep their chains: a motion vector is a symbol plus `res` raw bits where `res` depends on the
 * macroblock's type, h4m:1846-1860; type and proc runs mix raw bits with symbols, h4m:1545-1622.)
 *
 * Where a lane stops is a heuristic (the next section's header); how much of what it produced is consumed is decided by
 * the consumers exactly like the chains decide it.  If a consumer needs more than a lane produced (sections laid out in
 * an unusual order, capacities exceeded, overflow groups long enough to hit the chains' caps) the picture is marked
 * `retry` and decoded by the chains instead -- same blob either way, the flat path is only ever a faster way there.
 *
 * Plain C like the core, so that tests/native/gparse_emul.c runs it on the CPU; the device versions of the lane decode,
 * the scans and the DC wavefront (wave-level code) are in hvq_gparse.hip.
 */
#ifndef HVQ_GPARSE_FLAT_H
#define HVQ_GPARSE_FLAT_H

#include "hvq_gparse_core.h"

enum { GF_BN0 = 0, GF_BN1, GF_BNR0, GF_BNR1, GF_DC0, GF_DC1, GF_DC2, GF_BT0, GF_BT1, GF_BT2, GF_RLE0, GF_RLE1, GF_RLE2 };

/* part[] instances of 256 words (GF_P): 0-15 belong to the core (GP_EP(i, 0, t) = instance 4 + 4 i) */
#define GF_P(inst, tid) ((uint32_t)(inst) * 256u + (uint32_t)(tid))
#define GF_I_FX(i)    (4 + 4 * (i))    /* fixed-length bytes before the chunk (GP_EP(i, 0, .)) */
#define GF_I_ZERO(x)  (16 + (x))       /* expansion x: zero tokens before the chunk */
#define GF_I_LEN(x)   (21 + (x))       /* expansion x: blocks covered before the chunk */
#define GF_I_TERM(i)  (26 + (i))       /* DC buffer i: values completed before the chunk */
#define GF_I_DCF(i)   (29 + (i))       /*              chunk contains a value end */
#define GF_I_DCV(i)   (32 + (i))       /*              sum of the symbols after the last value end */
#define GF_I_PBF(i)   (35 + (i))       /* intra DC of plane i: chunk contains the head of a macroblock run */
#define GF_I_PBV(i)   (38 + (i))       /*                      accumulated value after it */
#define GF_I_NB(i)    (41 + (i))       /* coefficient symbols before the chunk */
#define GF_I_PREDI(i) (44 + (i))       /* MC-residual blocks before the chunk */
#define GF_I_LS       47               /* pool dwords of the tiles before the thread's tiles */
#define GF_I_MI       48               /* largest number of queued blocks / of pairs in one of the thread's tiles */
#define GF_I_MP       49

GP_FN void gp_chunk_min(uint32_t n, int tid, int nthr, uint32_t minper, uint32_t *lo, uint32_t *hi)
{
    uint32_t per = (n + (uint32_t)nthr - 1) / (uint32_t)nthr;
    if (per < minper) per = minper;
    const uint64_t a = (uint64_t)per * (uint32_t)tid;
    *lo = a < n ? (uint32_t)a : n;
    *hi = *lo + per < n ? *lo + per : n;
}

/* eight consecutive symbols / four consecutive values with one 16-byte load (p 16-byte aligned): the passes below walk
 * their chunk in such blocks -- a load instruction of a wave touches 64 cache lines whatever its width */
typedef struct { uint32_t x, y, z, w; } GfQuad;
typedef struct { uint32_t x, y; } GfPair;

/* value of leaf byte `b` of tree `tree` (gc_leaf with the tree's parameters of gp_read_tree: block kinds and run lengths are the
 * byte, DC symbols the signed byte << dc_shift, coefficient symbols the byte << 2; an empty section's tree has the one leaf 0) */
GP_FN int32_t gf_leaf_value(const GPic *g, int tree, uint32_t b)
{
    if (tree == GC_DC) return (int16_t)((uint32_t)(int32_t)(int8_t)(uint8_t)b << (g->dc_shift & 31));
    if (tree == GC_BT) return (int32_t)((b & 0xFFu) << 2);
    return (int32_t)(b & 0xFFu);
}

/* the symbols of lane `l`: leaf bytes, the lane's region of GPic.sym taken as bytes */
GP_FN const GP_G uint8_t *gf_lane_syms(const GPic *g, int l) { return (const GP_G uint8_t *)(g->sym + g->lane[l].off); }

/* eight consecutive symbols (p 8-byte aligned) as values of tree `tree` */
GP_FN void gf_ld8(const GPic *g, int tree, const GP_G uint8_t *p, int32_t *o)
{
    const GfPair q = *(const GP_G GfPair *)p;
    for (int k = 0; k < 4; ++k) { o[k] = gf_leaf_value(g, tree, q.x >> (8 * k)); o[4 + k] = gf_leaf_value(g, tree, q.y >> (8 * k)); }
}
GP_FN void gf_ld4(const GP_G uint32_t *p, uint32_t *o)
{
    const GfQuad q = *(const GP_G GfQuad *)p;
    o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
}

/* thread `tid`'s share of n items, a multiple of 8 items long and at least `minper` */
GP_FN void gp_chunk8(uint32_t n, int tid, int nthr, uint32_t minper, uint32_t *lo, uint32_t *hi)
{
    uint32_t per = (n + (uint32_t)nthr - 1) / (uint32_t)nthr;
    if (per < minper) per = minper;
    per = (per + 7u) & ~7u;
    const uint64_t a = (uint64_t)per * (uint32_t)tid;
    *lo = a < n ? (uint32_t)a : n;
    *hi = *lo + per < n ? *lo + per : n;
}

/* ------------------------------------------------------------------ lanes */
GP_FN uint32_t gf_cursor_pos(const GBits *b) { return b->idx * 32u - (uint32_t)b->cnt; }

/* serial (one thread), after the trees are read: where each lane starts and stops, where its symbols go */
GP_FN void gf_setup_lanes(GPic *g)
{
    if (g->status) return;
    static const uint8_t sec_i[GF_LANES] = { 0, 2, 1, 3, 4, 7, 10, 5, 8, 11, 13, 14, 15 };
    static const uint8_t tree[GF_LANES] = { GC_BN, GC_BN, GC_RUN, GC_RUN, GC_DC, GC_DC, GC_DC, GC_BT, GC_BT, GC_BT, GC_RUN, GC_RUN, GC_RUN };
    const GBits *cur[GF_LANES] = { &g->bn[0], &g->bn[1], &g->bnr[0], &g->bnr[1], &g->dc[0], &g->dc[1], &g->dc[2],
                                   &g->bt[0], &g->bt[1], &g->bt[2], &g->rle[0], &g->rle[1], &g->rle[2] };
    const int nl = g->is_pb ? GF_RLE0 : GF_LANES, nsec = g->is_pb ? 17 : 16;
    const uint32_t nb[3] = { g->pl[0].nblocks, g->pl[1].nblocks, g->pl[2].nblocks };
    /* never start a round of eight symbols where its window could leave the picture's dwords */
    const uint32_t hard = g->nd * 32u > 192u ? g->nd * 32u - 192u : 0u;
    uint32_t off = 0, voff = 0;
    for (int l = 0; l < GF_LANES; ++l) {
        GLane *q = &g->lane[l];
        const int i = l < GF_BNR0 ? l : (l < GF_DC0 ? l - GF_BNR0 : (l < GF_BT0 ? l - GF_DC0 : (l < GF_RLE0 ? l - GF_BT0 : l - GF_RLE0)));
        q->cap = l < GF_DC0 ? GP_CAP_BN(nb[i]) : (l < GF_BT0 ? GP_CAP_DC(nb[i]) : (l < GF_RLE0 ? GP_CAP_BT(nb[i]) : GP_CAP_BN(nb[i])));
        q->off = off; off += q->cap;
        q->tree = tree[l]; q->n = 0; q->pos = 0; q->end = 0;
        if (l >= GF_DC0 && l < GF_BT0) { g->val_off[i] = voff; voff += q->cap; g->nv[i] = 0; }
        if (l >= nl) continue;
        q->pos = gf_cursor_pos(cur[l]);
        uint32_t endb = g->len;                               /* the nearest section header after this payload */
        for (int s = 0; s < nsec; ++s) {
            const uint32_t hdr = g->sec_pay[s] >= 4 ? g->sec_pay[s] - 4u : 0u;
            if (hdr >= g->sec_pay[sec_i[l]] && s != (int)sec_i[l] && hdr < endb) endb = hdr;
        }
        uint32_t end = endb < (1u << 28) ? endb * 8u : 0u;
        if (end > hard) end = hard;
        q->end = end;
    }
}

/* The table a lane decodes with: 16-bit entries -- [5:0] bits consumed, [7] leaf reached, [15:8] the leaf byte or the inner node
 * (id - 256) -- in the LDS of the 8-bit tables with 32-bit entries the lanes' trees do not need on this path and of the 2 KB
 * behind the trees: 10 bits for the block kinds (a chroma pair's code is 9 or 10 bits in a dense stream: 45 % of them missed an
 * 8-bit table, and every miss holds all lanes up for a walk through the tree) and for the DC symbols (63 % of whose long codes
 * fit 10 bits, 38 % nine), 9 bits for run lengths and coefficient symbols (7 to 9 bits).  Where they lie: DC in the 2 KB behind
 * the trees; run lengths and coefficient symbols in their trees' `lut` arrays (1 KB each); the block kinds in their tree's `lut`
 * AND `kid` arrays, which are 2 KB in one piece -- the children of the block-kind tree (the walk of its rare longer codes needs
 * them) move into the DC tree's `lut` array first (gf_move_kids, one barrier before gf_fill_lane_tables). */
GP_FN int gf_lane_bits(int tree) { return tree == GC_BN || tree == GC_DC ? 10 : 9; }
GP_FN uint16_t *gf_lane_table(const GCode *codes, int tree)
{
    return (uint16_t *)(tree == GC_DC ? gp_stage + GP_XLUT_DWORD : (uint32_t *)codes[tree].lut);
}
/* children of inner node `id` (256..510) of a lane's tree: kid[bit * 256 + id - 256] */
GP_FN const uint16_t *gf_lane_kids(const GCode *codes, int tree)
{
    return tree == GC_BN ? (const uint16_t *)codes[GC_DC].lut : &codes[tree].kid[0][0];
}
GP_FN void gf_move_kids(const GPic *g, GCode *codes, int tid, int nthr)
{
    if (g->status) return;
    const uint32_t *src = (const uint32_t *)&codes[GC_BN].kid[0][0];
    uint32_t *dst = (uint32_t *)codes[GC_DC].lut;
    for (int k = tid; k < 256; k += nthr) dst[k] = src[k];
}
GP_FN void gf_fill_lane_tables(const GPic *g, GCode *codes, int tid, int nthr)       /* after gf_move_kids */
{
    if (g->status) return;
    for (int tree = GC_BN; tree <= GC_BT; ++tree) {
        const uint16_t *kid = gf_lane_kids(codes, tree);
        uint16_t *tab = gf_lane_table(codes, tree);
        const int bits = gf_lane_bits(tree), root = codes[tree].root;
        for (int e = tid; e < (1 << bits); e += nthr) {
            int node = root, d = 0;
            while (node >= 256 && d < bits) { node = kid[((e >> (bits - 1 - d)) & 1) * 256 + node - 256]; ++d; }
            tab[e] = (uint16_t)((uint32_t)d | (node < 256 ? 0x80u | ((uint32_t)node << 8) : (uint32_t)(node - 256) << 8));
        }
    }
}

#if !(defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__))
/* 64 bits of the picture from bit `pos` on (zeros past the end) */
GP_FN uint64_t gf_peek64(const GPic *g, uint32_t pos)
{
    const uint32_t i = pos >> 5, sh = pos & 31u;
    const uint64_t w0 = i < g->nd ? __builtin_bswap32(g->d[i]) : 0u;
    const uint64_t w1 = i + 1 < g->nd ? __builtin_bswap32(g->d[i + 1]) : 0u;
    const uint64_t w2 = i + 2 < g->nd ? __builtin_bswap32(g->d[i + 2]) : 0u;
    const uint64_t w = (w0 << 32) | w1;
    return sh ? (w << sh) | (w2 >> (32 - sh)) : w;
}

/* one lane, front to back (on the device the lanes of one wave run this in lockstep, hvq_gparse.hip) */
GP_FN void gf_decode_lane(GPic *g, const GCode *codes, int l)
{
    if (g->status) return;
    GLane *q = &g->lane[l];
    const GCode *c = &codes[q->tree];
    const uint16_t *tab = gf_lane_table(codes, (int)q->tree), *kid = gf_lane_kids(codes, (int)q->tree);
    const int bits = gf_lane_bits((int)q->tree);
    GP_G uint8_t *out = (GP_G uint8_t *)(g->sym + q->off);
    uint32_t pos = q->pos, n = 0;
    if (c->root < 256) { q->n = 0; return; }                  /* one-leaf tree: gf_fill_const */
    while (pos < q->end && n + 8 <= q->cap) {
        for (int k = 0; k < 8; ++k) {
            const uint32_t e = tab[gf_peek64(g, pos) >> (64 - bits)];
            pos += e & 63u;
            int id = (int)(e >> 8);
            if (!(e & 0x80u)) {
                id += 256;
                while (id >= 256) { id = kid[(gf_peek64(g, pos) >> 63) * 256 + id - 256]; ++pos; }
            }
            out[n++] = (uint8_t)id;
        }
    }
    q->n = n;
}
#endif

/* parallel: a lane whose tree is a single leaf decodes to that leaf for ever without consuming a bit */
GP_FN void gf_fill_const(GPic *g, const GCode *codes, int tid, int nthr)
{
    if (g->status) return;
    const int nl = g->is_pb ? GF_RLE0 : GF_LANES;
    for (int l = 0; l < nl; ++l) {
        const GLane *q = &g->lane[l];
        const GCode *c = &codes[q->tree];
        if (c->root >= 256) continue;
        const uint8_t v = (uint8_t)c->root;
        GP_G uint8_t *out = (GP_G uint8_t *)(g->sym + q->off);
        for (uint32_t j = (uint32_t)tid; j < q->cap; j += (uint32_t)nthr) out[j] = v;
    }
}

GP_FN void gf_fill_const_counts(GPic *g, const GCode *codes)          /* serial, with it */
{
    if (g->status) return;
    const int nl = g->is_pb ? GF_RLE0 : GF_LANES;
    for (int l = 0; l < nl; ++l) if (codes[g->lane[l].tree].root < 256) g->lane[l].n = g->lane[l].cap;
}

/* ------------------------------------------------------------------ scans over the 256 chunk partials */
#if !(defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__))
GP_FN void gf_scan_add(GPic *g, int inst, int nthr)                   /* exclusive, total -> tot[inst] */
{
    uint32_t run = 0;
    for (int t = 0; t < nthr; ++t) { const uint32_t v = g->part[GF_P(inst, t)]; g->part[GF_P(inst, t)] = run; run += v; }
    if (inst >= 16) g->tot[inst - 16] = run;
}

/* chunk t carries (flag: it contains a reset, value: what accumulated after its last reset, else over all of it);
 * afterwards value = what had accumulated when the chunk begins */
GP_FN void gf_scan_seg(GPic *g, int inst_flag, int inst_val, int nthr)
{
    uint32_t run = 0;
    for (int t = 0; t < nthr; ++t) {
        const uint32_t f = g->part[GF_P(inst_flag, t)], v = g->part[GF_P(inst_val, t)];
        g->part[GF_P(inst_val, t)] = run;
        run = f ? v : run + v;
    }
}
#endif

/* ------------------------------------------------------------------ DC buffer: symbols -> values (h4m:654-664) */
/* A value is the sum of its symbols up to and including the first one inside the window (lo, hi).  The chains cap a
 * value at GP_SOVF_CAP symbols; a run of that many window-edge symbols is left to them (`retry`), detected per chunk:
 * a run of 255 either has 128 symbols inside one chunk or covers a whole chunk that is not the array's last. */
GP_FN void gf_dc_count(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    const int32_t wlo = g->dc_lo, whi = g->dc_hi;
    for (int i = 0; i < 3; ++i) {
        const GLane *q = &g->lane[GF_DC0 + i];
        const GP_G uint8_t *S = gf_lane_syms(g, GF_DC0 + i);
        uint32_t lo, hi, T = 0, sum = 0, has = 0, run = 0, mx = 0;
        gp_chunk8(q->n, tid, nthr, 16, &lo, &hi);
        for (uint32_t j0 = lo; j0 < hi; j0 += 8) {
            int32_t b[8];
            gf_ld8(g, GC_DC, S + j0, b);
            const uint32_t m = hi - j0 < 8u ? hi - j0 : 8u;
            for (uint32_t k = 0; k < m; ++k) {
                const int32_t s = b[k];
                sum += (uint32_t)s;
                if (s <= wlo || s >= whi) { if (++run > mx) mx = run; }
                else { ++T; sum = 0; has = 1; run = 0; }
            }
        }
        if (mx >= 128 || (hi > lo && mx == hi - lo && hi < q->n)) g->retry = 1;
        g->part[GF_P(GF_I_TERM(i), tid)] = T;
        g->part[GF_P(GF_I_DCF(i), tid)] = has;
        g->part[GF_P(GF_I_DCV(i), tid)] = sum;
    }
}

GP_FN void gf_dc_values(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    const int32_t wlo = g->dc_lo, whi = g->dc_hi;
    for (int i = 0; i < 3; ++i) {
        const GLane *q = &g->lane[GF_DC0 + i];
        const GP_G uint8_t *S = gf_lane_syms(g, GF_DC0 + i);
        GP_G uint32_t *V = g->val + g->val_off[i];
        uint32_t lo, hi, vi = g->part[GF_P(GF_I_TERM(i), tid)], sum = g->part[GF_P(GF_I_DCV(i), tid)];
        gp_chunk8(q->n, tid, nthr, 16, &lo, &hi);
        for (uint32_t j0 = lo; j0 < hi; j0 += 8) {
            int32_t b[8];
            gf_ld8(g, GC_DC, S + j0, b);
            const uint32_t m = hi - j0 < 8u ? hi - j0 : 8u;
            for (uint32_t k = 0; k < m; ++k) {
                const int32_t s = b[k];
                sum += (uint32_t)s;
                if (!(s <= wlo || s >= whi)) { V[vi++] = sum; sum = 0; }
            }
        }
    }
}

/* ------------------------------------------------------------------ tokens with zero runs -> blocks */
/* The pattern of h4m:1090-1092, 1047-1050, 1699-1705: a non-zero token belongs to the next block; a zero token takes a
 * run length from a second section and leaves 1 + run blocks at zero.  Expansions: 0 = kinds Y, 1 = kinds U+V (slots of
 * the coded macroblocks in a P/B picture), 2-4 = DC deltas of an I picture's planes. */
typedef struct {
    const GP_G uint8_t *t8;              /* tokens: leaf bytes of the block-kind tree (their own values) ... */
    const GP_G uint32_t *t32;            /* ... or DC values */
    const GP_G uint8_t *run;             /* run lengths: leaf bytes of the run tree (their own values) */
    uint32_t ntok, nrun, N;
} GExp;

GP_FN int gf_exp(const GPic *g, int x, GExp *e)
{
    if (x < 2) {
        const GLane *t = &g->lane[GF_BN0 + x], *r = &g->lane[GF_BNR0 + x];
        e->t8 = gf_lane_syms(g, GF_BN0 + x); e->t32 = 0; e->ntok = t->n;
        e->run = gf_lane_syms(g, GF_BNR0 + x); e->nrun = r->n;
        e->N = g->is_pb ? g->ncoded * (uint32_t)g->pl[x].nblk : g->pl[x].nblocks;
    } else {
        if (g->is_pb) return 0;
        const int i = x - 2;
        const GLane *r = &g->lane[GF_RLE0 + i];
        e->t8 = 0; e->t32 = g->val + g->val_off[i]; e->ntok = g->nv[i];
        e->run = gf_lane_syms(g, GF_RLE0 + i); e->nrun = r->n;
        e->N = g->pl[i].nblocks;
    }
    if (e->ntok > e->N) e->ntok = e->N;                       /* a token covers at least one block */
    return 1;
}

/* tokens j0 .. j0 + 7 (j0 a multiple of 8; the arrays are padded to whole blocks) */
GP_FN void gf_tok8(const GExp *e, uint32_t j0, uint32_t *t)
{
    if (e->t8) { const GfPair q = *(const GP_G GfPair *)(e->t8 + j0); for (int k = 0; k < 4; ++k) { t[k] = (q.x >> (8 * k)) & 0xFFu; t[4 + k] = (q.y >> (8 * k)) & 0xFFu; } }
    else { gf_ld4(e->t32 + j0, t); gf_ld4(e->t32 + j0 + 4, t + 4); }
}

GP_FN void gf_exp_zeros(GPic *g, int x0, int x1, int tid, int nthr)           /* G1 */
{
    if (g->status || g->retry) return;
    for (int x = x0; x < x1; ++x) {
        GExp e;
        uint32_t lo, hi, z = 0;
        if (!gf_exp(g, x, &e)) continue;
        gp_chunk8(e.ntok, tid, nthr, 8, &lo, &hi);
        for (uint32_t j0 = lo; j0 < hi; j0 += 8) {
            uint32_t t[8];
            gf_tok8(&e, j0, t);
            const uint32_t m = hi - j0 < 8u ? hi - j0 : 8u;
            for (uint32_t k = 0; k < m; ++k) z += t[k] == 0;
        }
        g->part[GF_P(GF_I_ZERO(x), tid)] = z;
    }
}

GP_FN void gf_exp_lens(GPic *g, int x0, int x1, int tid, int nthr)            /* G2 */
{
    if (g->status || g->retry) return;
    for (int x = x0; x < x1; ++x) {
        GExp e;
        uint32_t lo, hi, len = 0;
        if (!gf_exp(g, x, &e)) continue;
        uint32_t z = g->part[GF_P(GF_I_ZERO(x), tid)];
        gp_chunk8(e.ntok, tid, nthr, 8, &lo, &hi);
        for (uint32_t j0 = lo; j0 < hi; j0 += 8) {
            uint32_t t[8];
            gf_tok8(&e, j0, t);
            const uint32_t m = hi - j0 < 8u ? hi - j0 : 8u;
            for (uint32_t k = 0; k < m; ++k) {
                ++len;
                if (t[k] == 0) { if (z < e.nrun) len += (uint32_t)e.run[z]; ++z; }
            }
        }
        g->part[GF_P(GF_I_LEN(x), tid)] = len;
    }
}

GP_FN void gf_exp_put(const GPic *g, int x, uint32_t at, uint32_t tok)
{
    if (x >= 2) {                                              /* DC delta; the prediction comes later */
        const GPlane *q = &g->pl[x - 2];
        const uint32_t by = at / (uint32_t)q->hb, bx = at - by * (uint32_t)q->hb;
        gp_map_ent(g, x - 2, (int)by, (int)bx)[0] = (uint8_t)tok;
        return;
    }
    const GPlane *q = &g->pl[x];
    int by, bx;
    uint32_t tag = 0;
    if (g->is_pb) {
        const uint32_t sh = q->nblk == 4 ? 2u : 0u;             /* nblk is 1 or 4 */
        const uint32_t r = q->nblk == 4 || q->nblk == 1 ? at >> sh : at / (uint32_t)q->nblk, j = at - r * (uint32_t)q->nblk;
        const uint32_t m = g->cmb[r];
        tag = g->mbtag[m];
        const int my = (int)(m / (uint32_t)g->mw), mx = (int)(m - (uint32_t)my * (uint32_t)g->mw);
        by = my * q->by_per + gp_dy((int)j); bx = mx * q->bx_per + gp_dx((int)j);
    } else {
        by = (int)(at / (uint32_t)q->hb); bx = (int)(at - (uint32_t)by * (uint32_t)q->hb);
    }
    if (x == 0) gp_map_ent(g, 0, by, bx)[1] = (uint8_t)(tag | (g->is_pb ? (tok & 0xFFu) : tok));
    else {
        gp_map_ent(g, 1, by, bx)[1] = (uint8_t)(tag | (tok & 0xFu));
        gp_map_ent(g, 2, by, bx)[1] = (uint8_t)(tag | ((tok >> 4) & 0xFu));
    }
}

GP_FN void gf_exp_write(GPic *g, int x0, int x1, int tid, int nthr)           /* G3 */
{
    if (g->status || g->retry) return;
    for (int x = x0; x < x1; ++x) {
        GExp e;
        uint32_t lo, hi;
        if (!gf_exp(g, x, &e)) continue;
        if (g->tot[GF_I_LEN(x) - 16] < e.N) { g->retry = 1; continue; }       /* not enough tokens */
        uint32_t z = g->part[GF_P(GF_I_ZERO(x), tid)], at = g->part[GF_P(GF_I_LEN(x), tid)];
        gp_chunk8(e.ntok, tid, nthr, 8, &lo, &hi);
        for (uint32_t j0 = lo; j0 < hi && at < e.N; j0 += 8) {
            uint32_t t[8];
            gf_tok8(&e, j0, t);
            const uint32_t m = hi - j0 < 8u ? hi - j0 : 8u;
            for (uint32_t k = 0; k < m && at < e.N; ++k) {
                if (t[k] == 0) {
                    if (z >= e.nrun) { g->retry = 1; at = e.N; break; }       /* not enough run lengths */
                    at += 1u + (uint32_t)e.run[z++];
                } else { gf_exp_put(g, x, at, t[k]); ++at; }
            }
        }
    }
}

/* ------------------------------------------------------------------ I picture: DC prediction (h4m:1132-1164) */
#if !(defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__))
/* the map holds the deltas; value = predictor + delta with the predictor of gp_idc.  (On the device: a wavefront over
 * the anti-diagonals, hvq_gparse.hip.) */
GP_FN void gf_idc_predict(GPic *g, int i, uint8_t *rowbuf)
{
    if (g->status || g->retry) return;
    const GPlane *q = &g->pl[i];
    for (int bx = 0; bx <= q->hb; ++bx) rowbuf[bx] = 0x7F;
    for (int by = 0; by < q->vb; ++by) {
        GP_G uint8_t *row = gp_map_ent(g, i, by, 0);
        uint8_t pred = by ? rowbuf[0] : 0x7F;
        for (int bx = 0; bx < q->hb; ++bx) {
            const uint8_t v = (uint8_t)(pred + row[2 * bx]);
            row[2 * bx] = v;
            pred = (uint8_t)((v + rowbuf[bx + 1] + 1) / 2);
            rowbuf[bx] = v;
        }
    }
}
#endif

/* ------------------------------------------------------------------ P/B picture: DC of the intra macroblocks (h4m:1742-1776) */
/* value e of plane i belongs to block e % nblk of the intra macroblock t0[e / nblk]; the DC accumulates from 0x7F
 * within a run of consecutive intra macroblocks */
GP_FN void gf_pbdc_sums(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    for (int i = 0; i < 3; ++i) {
        const uint32_t nblk = (uint32_t)g->pl[i].nblk;
        if (g->ntype0 * nblk > g->nv[i]) { g->retry = 1; continue; }
        const GP_G uint32_t *V = g->val + g->val_off[i];
        uint32_t lo, hi, has = 0, acc = 0;
        gp_chunk(g->ntype0, tid, nthr, &lo, &hi);
        for (uint32_t r = lo; r < hi; ++r) {
            if (r == 0 || g->t0[r] != g->t0[r - 1] + 1u) { has = 1; acc = 0x7F; }
            for (uint32_t j = 0; j < nblk; ++j) acc += V[r * nblk + j];
        }
        g->part[GF_P(GF_I_PBF(i), tid)] = has;
        g->part[GF_P(GF_I_PBV(i), tid)] = acc;
    }
}

GP_FN void gf_pbdc_write(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    for (int i = 0; i < 3; ++i) {
        const GPlane *q = &g->pl[i];
        const uint32_t nblk = (uint32_t)q->nblk;
        const GP_G uint32_t *V = g->val + g->val_off[i];
        uint32_t lo, hi, acc = g->part[GF_P(GF_I_PBV(i), tid)];
        gp_chunk(g->ntype0, tid, nthr, &lo, &hi);
        for (uint32_t r = lo; r < hi; ++r) {
            const uint32_t m = g->t0[r];
            if (r == 0 || m != g->t0[r - 1] + 1u) acc = 0x7F;
            const int my = (int)(m / (uint32_t)g->mw), mx = (int)(m - (uint32_t)my * (uint32_t)g->mw);
            for (uint32_t j = 0; j < nblk; ++j) {
                acc += V[r * nblk + j];
                gp_map_ent(g, i, my * q->by_per + gp_dy((int)j), mx * q->bx_per + gp_dx((int)j))[0] = (uint8_t)acc;
            }
        }
    }
}

/* ------------------------------------------------------------------ pool layout by all threads */
/* gp_layout_sum / gp_layout_scan with the scan of the runs spread out: a thread sums whole tiles (4 runs of 64 blocks),
 * leaves each run's offset relative to its own first tile in wave_base[], and the scan of the 256 thread totals makes
 * them absolute in gf_layout_blocks */
GP_FN void gf_layout_sum(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    GP_G uint32_t *wave_base = (GP_G uint32_t *)(g->blob + g->wave_base_off);
    uint32_t lo, hi, run = 0, mi = 0, mp = 0, fl = 0;
    gp_chunk(g->total_tiles, tid, nthr, &lo, &hi);
    for (uint32_t t = lo; t < hi; ++t) {
        uint32_t ti = 0, tp = 0;
        for (uint32_t k = 0; k < HVQ_TILE_BLOCKS / 64; ++k) {
            const uint32_t r = t * (HVQ_TILE_BLOCKS / 64) + k;
            const int i = gp_run_plane(g, r);
            const GPlane *q = &g->pl[i];
            const int ctx = g->is_pb ? 2 : (i == 0 ? 0 : 1);
            const uint32_t b0 = (r - q->run_first) * 64u;
            uint32_t sum = 0;
            uint32_t by = b0 / (uint32_t)q->hb, bx = b0 - by * (uint32_t)q->hb;
            const GP_G uint8_t *tp8 = gp_map_ent(g, i, (int)by, (int)bx) + 1;      /* type bytes: 2 apart, 4 more over the border */
            const uint32_t bend = b0 + 64u < q->nblocks ? b0 + 64u : q->nblocks;
            /* eight type bytes requested before the first is looked at: one at a time, a run is 64 round trips to the L2 in a row */
            for (uint32_t b = b0; b < bend; b += 8) {
                uint32_t tv[8];
                const uint32_t m = bend - b < 8u ? bend - b : 8u;
                GP_UNROLL
                for (uint32_t j = 0; j < 8; ++j) {
                    tv[j] = j < m ? *tp8 : 0u;
                    if (j < m) { tp8 += 2; if (++bx == (uint32_t)q->hb) { bx = 0; tp8 += 4; } }
                }
                GP_UNROLL
                for (uint32_t j = 0; j < 8; ++j) {
                    uint32_t n, it, pr, f;
                    if (j >= m) break;
                    gp_type_info(ctx, tv[j], &n, &it, &pr, &f);
                    sum += n; ti += it; tp += pr; fl |= f;
                }
            }
            wave_base[r] = run;
            run += sum;
        }
        if (ti > mi) mi = ti;
        if (tp > mp) mp = tp;
    }
    g->part[tid] = fl;
    g->part[GF_P(GF_I_LS, tid)] = run;
    g->part[GF_P(GF_I_MI, tid)] = mi;
    g->part[GF_P(GF_I_MP, tid)] = mp;
}

#if !(defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__))
GP_FN void gf_layout_finish(GPic *g, int nthr)               /* serial; one wave on the device (hvq_gparse.hip) */
{
    if (g->status || g->retry) return;
    uint32_t fl = 0, mi = 0, mp = 0;
    gf_scan_add(g, GF_I_LS, nthr);
    for (int t = 0; t < nthr; ++t) {
        fl |= g->part[t];
        if (g->is_pb) fl |= g->part[GP_PART2 + t];
        if (g->part[GF_P(GF_I_MI, t)] > mi) mi = g->part[GF_P(GF_I_MI, t)];
        if (g->part[GF_P(GF_I_MP, t)] > mp) mp = g->part[GF_P(GF_I_MP, t)];
    }
    if (!g->is_pb) fl |= g->part[GP_MISC + 2 * GC_COUNT];
    gp_layout_finish(g, g->tot[GF_I_LS - 16], fl, mi, mp);
}
#endif

/* gp_layout_blocks for this layout: the run offsets become absolute here */
GP_FN void gf_layout_blocks(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    GP_G uint32_t *wave_base = (GP_G uint32_t *)(g->blob + g->wave_base_off);
    const uint32_t per = (g->total_tiles + (uint32_t)nthr - 1) / (uint32_t)nthr;
    for (uint32_t r = (uint32_t)tid; r < g->total_runs; r += (uint32_t)nthr) {
        const int i = gp_run_plane(g, r);
        const GPlane *q = &g->pl[i];
        const int ctx = g->is_pb ? 2 : (i == 0 ? 0 : 1);
        const uint32_t b0 = (r - q->run_first) * 64u;
        uint32_t off = wave_base[r] + g->part[GF_P(GF_I_LS, (r / (HVQ_TILE_BLOCKS / 64)) / per)];
        wave_base[r] = off;
        uint32_t by = b0 / (uint32_t)q->hb, bx = b0 - by * (uint32_t)q->hb;
        const GP_G uint8_t *tp8 = gp_map_ent(g, i, (int)by, (int)bx) + 1;
        const uint32_t bend = b0 + 64u < q->nblocks ? b0 + 64u : q->nblocks;
        for (uint32_t b8 = b0; b8 < bend; b8 += 8) {
            uint32_t tv[8];
            const uint32_t m = bend - b8 < 8u ? bend - b8 : 8u;
            {                                                 /* the next eight type bytes, requested together (gf_layout_sum) */
                const GP_G uint8_t *p = tp8;
                uint32_t x = bx;
                GP_UNROLL
                for (uint32_t j = 0; j < 8; ++j) {
                    tv[j] = j < m ? *p : 0u;
                    if (j < m) { p += 2; if (++x == (uint32_t)q->hb) { x = 0; p += 4; } }
                }
            }
            GP_UNROLL
            for (uint32_t j = 0; j < 8; ++j) {
                if (j >= m) break;
                const uint32_t b = b8 + j, t = tv[j];
                uint32_t n, it, pr, f;
                gp_type_info(ctx, t, &n, &it, &pr, &f);
                const uint32_t kind = ctx == 0 ? t : (t & 0xFu);
                const int inter = ctx == 2 && (t & 0x60u);
                uint32_t ent = GP_ENT(off, 0, GP_MODE_NONE);
                if (n) ent = kind == 6 ? GP_ENT(off, 0, GP_MODE_LITERAL)
                           : (inter ? GP_ENT(off, kind - 1, GP_MODE_PREDI) : GP_ENT(off, kind, GP_MODE_BASES));
                uint32_t at = b;
                if (g->is_pb) {                               /* by_per, bx_per are 1 or 2: no divisions in this loop */
                    const uint32_t dy = by & (uint32_t)(q->by_per - 1), dx = bx & (uint32_t)(q->bx_per - 1);
                    const uint32_t mb = (by >> (q->by_per >> 1)) * (uint32_t)g->mw + (bx >> (q->bx_per >> 1));
                    at = mb * (uint32_t)q->nblk + (dx ? (dy ? 2u : 3u) : (dy ? 1u : 0u));
                    if (q->nblk == 1) at = mb;
                }
                g->pinfo[q->blk_first + at] = ent;
                off += n;
                tp8 += 2;
                if (++bx == (uint32_t)q->hb) { bx = 0; ++by; tp8 += 4; }
            }
        }
    }
}

/* ------------------------------------------------------------------ payloads from the arrays */
GP_FN void gf_emit_count(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    for (int i = 0; i < 3; ++i) {
        const GP_G uint32_t *ents = g->pinfo + g->pl[i].blk_first;
        uint32_t lo, hi, bytes = 0, nb = 0, np = 0;
        gp_chunk(g->pl[i].nblocks, tid, nthr, &lo, &hi);
        for (uint32_t e = lo; e < hi; ++e) {
            const uint32_t ent = ents[e], mode = ent >> 30;
            bytes += gp_ent_fx_bytes(ent);
            if (mode >= GP_MODE_BASES) nb += (ent >> 22) & 0xFFu;
            np += mode == GP_MODE_PREDI;
        }
        g->part[GF_P(GF_I_FX(i), tid)] = bytes;
        g->part[GF_P(GF_I_NB(i), tid)] = nb;
        g->part[GF_P(GF_I_PREDI(i), tid)] = np;
    }
}

GP_FN int gf_emit_short(const GPic *g)           /* after the scans of FX, NB, PREDI: do the arrays hold enough? */
{
    for (int i = 0; i < 3; ++i) {
        if (g->tot[GF_I_NB(i) - 16] > g->lane[GF_BT0 + i].n) return 1;
        if (g->is_pb && g->ntype0 * (uint32_t)g->pl[i].nblk + 2u * g->tot[GF_I_PREDI(i) - 16] > g->nv[i]) return 1;
    }
    return 0;
}

/* parallel: every payload entry completed in one go -- basis word from the fixed-length section, running sum of the
 * coefficient symbols (h4m:726-731), the two scalars of an MC-residual block from the DC buffer's values after the
 * intra DCs (h4m:1405-1406), literal blocks copied */
GP_FN void gf_emit_merge(GPic *g, int tid, int nthr)
{
    if (g->status || g->retry) return;
    if (gf_emit_short(g)) { g->retry = 1; return; }
    GP_G uint32_t *pool = (GP_G uint32_t *)(g->blob + g->fixed_bytes);
    const int sh_dc = g->dc_shift & 31, sh_unk = g->unk_shift & 31;
    for (int i = 0; i < 3; ++i) {
        const GP_G uint32_t *ents = g->pinfo + g->pl[i].blk_first;
        const GP_G uint8_t *S = gf_lane_syms(g, GF_BT0 + i);
        const GP_G uint32_t *V = g->val + g->val_off[i] + g->ntype0 * (uint32_t)g->pl[i].nblk;
        uint32_t lo, hi;
        gp_chunk(g->pl[i].nblocks, tid, nthr, &lo, &hi);
        uint64_t fx = g->fx_off[i] + g->part[GF_P(GF_I_FX(i), tid)];
        uint32_t si = g->part[GF_P(GF_I_NB(i), tid)], pi = g->part[GF_P(GF_I_PREDI(i), tid)];
        for (uint32_t e = lo; e < hi; ++e) {
            const uint32_t ent = ents[e], mode = ent >> 30;
            if (mode == GP_MODE_NONE) continue;
            GP_G uint32_t *dst = pool + (ent & 0x3FFFFFu);
            if (mode == GP_MODE_LITERAL) {
                uint32_t v[4];
                for (int k = 0; k < 4; ++k) v[k] = __builtin_bswap32(gp_be32(g, fx + 4u * (uint32_t)k));
                for (int k = 0; k < 4; ++k) dst[k] = v[k];
                fx += 16;
                continue;
            }
            const uint32_t nb = (ent >> 22) & 0xFFu;
            if (mode == GP_MODE_PREDI) {
                const int32_t s1 = (int32_t)V[2u * pi], s2 = (int32_t)V[2u * pi + 1u];
                dst[0] = (uint32_t)(s1 >> sh_dc) << sh_unk;
                dst[1] = (uint32_t)(s2 >> sh_dc);
                dst += 2; ++pi;
            }
            uint32_t run = 0;
            /* four bases at a time, all loads first: a thread's speed here is the number of HBM round trips it waits for */
            for (uint32_t k0 = 0; k0 < nb; k0 += 4) {
                uint32_t w[4];
                int32_t sv[4];
                for (uint32_t j = 0; j < 4; ++j) {
                    const int live = k0 + j < nb;
                    w[j] = live ? gp_be16(g, fx + 2u * (k0 + j)) : 0u;
                    sv[j] = live ? (int32_t)S[si + k0 + j] << 2 : 0;          /* gf_leaf_value of the coefficient tree */
                }
                for (uint32_t j = 0; j < 4; ++j) {
                    run += (uint32_t)sv[j];
                    if (k0 + j < nb) dst[k0 + j] = HVQ_BASIS(w[j], (run + ((w[j] >> 13) & 3u)) & 0x3FFFFu);
                }
            }
            si += nb; fx += 2u * nb;
        }
    }
}

#endif
