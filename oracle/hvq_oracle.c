/*
 * oracle/hvq_oracle.c -- TEST INFRASTRUCTURE ONLY (see hvq_oracle.h).
 *
 * From-scratch scalar C restatement of the reference HVQM4 decoder's picture
 * path.  Every function cites the reference lines (h4m: = h4m_audio_decode.c)
 * whose behaviour it restates.  Parse and reconstruction are interleaved exactly
 * like the reference (one pass over the bit buffers), which makes this the
 * simplest thing to compare against it; the product's two-stage design
 * (host parse -> descriptors -> HIP kernels) is checked against this file.
 *
 * All arithmetic that the reference performs in wrapping uint32/int32 is done in
 * uint32_t here so that the result is defined C and equals what gcc/clang emit
 * for the reference.
 */
#include "hvq_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

#define API __attribute__((visibility("default")))

/* ---------- bit reader: MSB-first byte stream (== h4m:552-602 over BE32 words) ---------- */
typedef struct {
    const uint8_t *base;   /* NULL when the section is empty (h4m:1061-1071) */
    uint32_t pos;          /* in bits */
} Bits;

static inline uint32_t take1(Bits *b)
{
    uint32_t v = (b->base[b->pos >> 3] >> (7 - (b->pos & 7))) & 1u;
    b->pos++;
    return v;
}

static inline uint32_t take(Bits *b, int n)
{
    uint32_t v = 0;
    while (n-- > 0) v = (v << 1) | take1(b);
    return v;
}

static uint32_t rd32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t rd16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

/* ---------- prefix code (h4m:385-394, 604-651) ---------- */
typedef struct {
    int root;              /* < 256: single leaf, >= 256: inner node */
    int next;
    int16_t kid[2][512];
    int32_t leaf[256];     /* value of leaf byte, after sign/scale/int16 truncation */
} Code;

static int code_node(Code *c, Bits *b, int is_signed, int scale)
{
    if (take1(b) == 0) {
        int byte = (int)take(b, 8);
        int v = (is_signed && byte > 0x7F) ? byte - 256 : byte;
        c->leaf[byte] = (int16_t)((uint32_t)v << scale);      /* h4m:613-617: shifted in int, kept as int16 */
        return byte;
    }
    int id = c->next < 511 ? c->next++ : 511;
    int a = code_node(c, b, is_signed, scale);
    c->kid[0][id] = (int16_t)a;
    int d = code_node(c, b, is_signed, scale);
    c->kid[1][id] = (int16_t)d;
    return id;
}

static void code_read(Code *c, Bits *carrier, int is_signed, int scale)   /* h4m:632-642 */
{
    c->next = 0x100;
    c->root = carrier->base ? code_node(c, carrier, is_signed, scale) : 0;
}

static inline int32_t sym(const Code *c, Bits *b)                          /* h4m:644-651 */
{
    int n = c->root;
    while (n >= 0x100) n = c->kid[take1(b)][n];
    return c->leaf[n];
}

static int32_t sym_sovf(const Code *c, Bits *b, int32_t lo, int32_t hi)    /* h4m:654-664 */
{
    int32_t total = 0, v;
    do { v = sym(c, b); total = (int32_t)((uint32_t)total + (uint32_t)v); } while (v <= lo || v >= hi);
    return total;
}

static int32_t sym_uovf(const Code *c, Bits *b)                            /* h4m:667-677, max always 255 */
{
    int32_t total = 0, v;
    do { v = sym(c, b); total += v; } while (v >= 0xFF);
    return total;
}

/* ---------- geometry (h4m:438-465, 843-870) ---------- */
typedef struct {
    int wshift, hshift;
    int pw, ph;            /* samples */
    int hb, vb;            /* 4x4 blocks */
    int stride;            /* map entries per row incl. border */
    int nblk;              /* blocks per macroblock */
    int bx_per;            /* blocks per macroblock horizontally */
    int by_per;
    int moff[4];           /* map-entry offsets of the blocks of one macroblock: TL, BL, BR, TR */
    int poff[4];           /* sample offsets, same order */
    uint32_t samples;
    uint8_t *map;          /* (vb+2)*(hb+2) entries of {value,type} */
} Plane;

struct HvqOracle {
    int w, h, is15, landscape, nest_w, nest_h;
    Plane pl[3];
    uint8_t nest[70 * 38];
    Code c_dc, c_run, c_bt, c_bn, c_mv, c_mcb;
    Bits bn[2], bnr[2], dc[3], bt[3], rle[3], mvh, mvv, mtype, mproc;
    const uint8_t *fx[3];
    int unk_shift, dc_shift;
    int32_t dc_lo, dc_hi;
    uint8_t res[6];        /* h0 h1 v0 v1 pad pad  (h4m:502-505 memory order) */
    int32_t divt[16];
    int32_t mcdiv[512];
    uint32_t picsize;
};

static inline uint8_t *ent(const Plane *p, int by, int bx) { return p->map + 2 * ((by + 1) * p->stride + bx + 1); }

static void tables(int32_t *divt, int32_t *mcdiv)                         /* h4m:265-273 */
{
    divt[0] = 0; mcdiv[0] = 0;
    for (int i = 1; i < 16; ++i) divt[i] = 0x1000 / (i * 16) * 16;
    for (int i = 1; i < 512; ++i) mcdiv[i] = 0x1000 / i;
}

API void hvqo_tables(int32_t div16[16], int32_t mcdiv512[512]) { tables(div16, mcdiv512); }

API HvqOracle *hvqo_create(int width, int height, int h_samp, int v_samp, int is15)
{
    HvqOracle *o = calloc(1, sizeof *o);
    o->w = width; o->h = height; o->is15 = is15;
    o->landscape = width >= height;                                       /* h4m:965-975 */
    o->nest_w = o->landscape ? 70 : 38;
    o->nest_h = o->landscape ? 38 : 70;
    tables(o->divt, o->mcdiv);
    for (int i = 0; i < 3; ++i) {
        Plane *p = &o->pl[i];
        int hs = i ? h_samp : 1, vs = i ? v_samp : 1;
        p->wshift = hs == 2; p->hshift = vs == 2;
        p->pw = width >> p->wshift; p->ph = height >> p->hshift;
        p->samples = (uint32_t)p->pw * p->ph;
        p->bx_per = 2 >> p->wshift; p->by_per = 2 >> p->hshift;
        p->nblk = p->bx_per * p->by_per;
        p->hb = width / (hs * 4); p->vb = height / (vs * 4);
        p->stride = p->hb + 2;
        p->moff[0] = 0; p->moff[1] = p->stride; p->moff[2] = p->stride + 1; p->moff[3] = 1;
        p->poff[0] = 0; p->poff[1] = p->pw * 4; p->poff[2] = p->pw * 4 + 4; p->poff[3] = 4;
        size_t n = (size_t)p->stride * (p->vb + 2);
        p->map = calloc(n + 8, 2);
        for (int r = 0; r < p->vb + 2; ++r)                               /* border {0x7F,0xFF}: h4m:951-955, 1011-1039 */
            for (int c = 0; c < p->stride; ++c)
                if (r == 0 || r == p->vb + 1 || c == 0 || c == p->stride - 1) {
                    p->map[2 * (r * p->stride + c)] = 0x7F;
                    p->map[2 * (r * p->stride + c) + 1] = 0xFF;
                }
        o->picsize += p->samples;
    }
    return o;
}

API void hvqo_destroy(HvqOracle *o)
{
    if (!o) return;
    for (int i = 0; i < 3; ++i) free(o->pl[i].map);
    free(o);
}

API uint32_t hvqo_picsize(const HvqOracle *o) { return o->picsize; }
API const uint8_t *hvqo_nest(const HvqOracle *o) { return o->nest; }
API const uint8_t *hvqo_map(const HvqOracle *o, int plane, uint32_t *stride, uint32_t *rows)
{
    if (stride) *stride = (uint32_t)o->pl[plane].stride;
    if (rows) *rows = (uint32_t)o->pl[plane].vb + 2;
    return o->pl[plane].map;
}

/* ---------- pixel primitives ---------- */
static inline uint8_t clamp255(int32_t x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }   /* h4m:288-291 */

/* h4m:293-296: the argument is taken as uint32, so sums below -4 wrap to huge values and clamp to 255 */
static inline uint8_t mean8(int32_t s) { return clamp255((int32_t)(((uint32_t)s + 4u) / 8u)); }

static void fill_block(uint8_t *dst, int stride, uint8_t v)               /* h4m:281-286 */
{
    for (int y = 0; y < 4; ++y) memset(dst + y * stride, v, 4);
}

/*
 * h4m:299-383.  The 16 outputs of the reference are 8*V + r[y] + c[x] with
 *   r[y] = a[y]*(T-V) + a[3-y]*(B-V),  c[x] = a[x]*(L-V) + a[3-x]*(R-V),  a = {2, 0, -1, -1}
 * (expand the reference's vph/vmh/tpl/... terms, or see the weight tables in its comments
 * h4m:330-347: V 6/8/10, near edge +2, far edges -1).
 */
static void weight_block(uint8_t *dst, int stride, int V, int T, int B, int L, int R)
{
    static const int a[4] = { 2, 0, -1, -1 };
    for (int y = 0; y < 4; ++y) {
        int r = a[y] * (T - V) + a[3 - y] * (B - V);
        for (int x = 0; x < 4; ++x) {
            int c = a[x] * (L - V) + a[3 - x] * (R - V);
            dst[y * stride + x] = mean8(8 * V + r + c);
        }
    }
}

API void hvqo_weight_block(uint8_t dst16[16], uint8_t v, uint8_t t, uint8_t b, uint8_t l, uint8_t r)
{
    weight_block(dst16, 4, v, t, b, l, r);
}

/* h4m:1242-1294: copy / vertical / horizontal / bilinear half-sample */
static void mc_block(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int hx, int hy)
{
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) {
            const uint8_t *s = src + y * sstride + x;
            int v;
            if (!hx && !hy) v = s[0];
            else if (hx && !hy) v = (s[0] + s[1] + 1) / 2;
            else if (!hx) v = (s[0] + s[sstride] + 1) / 2;
            else v = (s[0] + s[1] + s[sstride] + s[sstride + 1] + 2) >> 2;
            dst[y * dstride + x] = (uint8_t)v;
        }
}

API void hvqo_motion_comp(uint8_t dst16[16], const uint8_t *src, uint32_t stride, int hx, int hy)
{
    mc_block(dst16, 4, src, (int)stride, hx, hy);
}

static void literal_block(HvqOracle *o, int plane, uint8_t *dst, int stride)   /* h4m:543-549 */
{
    const uint8_t *s = o->fx[plane];
    for (int y = 0; y < 4; ++y) memcpy(dst + y * stride, s + 4 * y, 4);
    o->fx[plane] = s + 16;
}

/*
 * One AOT basis: h4m:679-732 (nest of 4-bit values, hi_nibble = 0) and h4m:734-773
 * (window of the reference luma picture, hi_nibble = 1).  Adds factor*basis into acc[16]
 * (uint32 wrap, h4m:787/809).  *run is the running coefficient sum of the block.
 */
static void aot_basis(HvqOracle *o, int plane, const uint8_t *nest, int nstride, int hi_nibble,
                      uint32_t *run, uint32_t acc[16])
{
    uint32_t word = rd16(o->fx[plane]);
    o->fx[plane] += 2;
    uint32_t off_long = word & 0x3F, off_short = (word >> 6) & 0x1F;
    uint32_t s_long = (word >> 11) & 1, s_short = (word >> 12) & 1;
    int xs, ys;
    const uint8_t *p;
    if (o->landscape) { p = nest + (size_t)nstride * off_short + off_long; xs = 1 << s_long; ys = nstride << s_short; }
    else              { p = nest + (size_t)nstride * off_long + off_short; xs = 1 << s_short; ys = nstride << s_long; }
    uint8_t e[16];
    uint8_t lo = 255, hi = 0;
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) {
            uint8_t v = p[y * ys + x * xs];
            if (hi_nibble) v = (v >> 4) & 0xF;
            e[4 * y + x] = v;
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
    *run += (uint32_t)sym(&o->c_bt, &o->bt[plane]);
    int32_t inv = o->divt[(hi - lo) & 15];
    if (word & 0x8000) inv = -inv;
    uint32_t factor = (*run + ((word >> 13) & 3)) * (uint32_t)inv;
    for (int i = 0; i < 16; ++i) acc[i] += factor * e[i];
}

/* h4m:775-817: sum of n bases, returns the (arithmetic) mean of the 16 accumulators */
static int32_t aot_sum(HvqOracle *o, int plane, int n, const uint8_t *nest, int nstride, int hi_nibble, uint32_t acc[16])
{
    uint32_t run = 0, total = 0;
    memset(acc, 0, 16 * sizeof *acc);
    for (int k = 0; k < n; ++k) aot_basis(o, plane, nest, nstride, hi_nibble, &run, acc);
    for (int i = 0; i < 16; ++i) total += acc[i];
    return (int32_t)total >> 4;
}

/* h4m:1358-1377 */
static void intra_aot_block(HvqOracle *o, int plane, uint8_t *dst, int stride, uint8_t dcv, int kind)
{
    if (kind == 6) { literal_block(o, plane, dst, stride); return; }
    uint32_t acc[16];
    int32_t mean = aot_sum(o, plane, kind, o->nest, o->nest_w, 0, acc);
    uint32_t delta = ((uint32_t)dcv << o->unk_shift) - (uint32_t)mean;
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x)
            dst[y * stride + x] = clamp255((int32_t)(acc[4 * y + x] + delta) >> o->unk_shift);
}

/* h4m:1379-1420 */
static void predi_aot_block(HvqOracle *o, int plane, uint8_t *dst, const uint8_t *src, int stride, int kind,
                            const uint8_t *window, int wstride, int hx, int hy)
{
    uint32_t acc[16];
    uint32_t mean_aot = (uint32_t)aot_sum(o, plane, kind - 1, window, wstride, 1, acc);
    uint8_t m[16];
    mc_block(m, 4, src, stride, hx, hy);
    int32_t s = 8;
    for (int i = 0; i < 16; ++i) s += m[i];
    int32_t mean = s / 16;
    int32_t lo = m[0] - mean, hi = lo;
    for (int i = 0; i < 16; ++i) {
        int32_t d = m[i] - mean;
        if (d < lo) lo = d;
        if (d > hi) hi = d;
    }
    int32_t s1 = sym_sovf(&o->c_dc, &o->dc[plane], o->dc_lo, o->dc_hi);
    int32_t s2 = sym_sovf(&o->c_dc, &o->dc[plane], o->dc_lo, o->dc_hi);
    uint32_t addend = ((uint32_t)(s1 >> o->dc_shift) << o->unk_shift) - mean_aot;
    uint32_t factor = (uint32_t)(s2 >> o->dc_shift) * (uint32_t)o->mcdiv[(hi - lo) & 511];
    for (int i = 0; i < 16; ++i) {
        uint32_t r = acc[i] + addend + (uint32_t)(m[i] - mean) * factor;
        dst[(i >> 2) * stride + (i & 3)] = clamp255(((int32_t)r >> o->unk_shift) + m[i]);
    }
}

/* ---------- picture header / sections ---------- */
static Bits section_bits(const uint8_t *data, const uint8_t *tab, int i)   /* h4m:1061-1071 */
{
    const uint8_t *s = data + rd32(tab + 4 * i);
    Bits b = { rd32(s) ? s + 4 : NULL, 0 };
    return b;
}
static const uint8_t *section_bytes(const uint8_t *data, const uint8_t *tab, int i)
{
    const uint8_t *s = data + rd32(tab + 4 * i);
    return rd32(s) ? s + 4 : NULL;
}

static void common_sections(HvqOracle *o, const uint8_t *data, const uint8_t *tab)
{
    for (int i = 0; i < 2; ++i) { o->bn[i] = section_bits(data, tab, 2 * i); o->bnr[i] = section_bits(data, tab, 2 * i + 1); }
    for (int p = 0; p < 3; ++p) {
        o->dc[p] = section_bits(data, tab, 4 + 3 * p);
        o->bt[p] = section_bits(data, tab, 5 + 3 * p);
        o->fx[p] = section_bytes(data, tab, 6 + 3 * p);
    }
}

/* ---------- I pictures ---------- */
static void ipic_kinds(HvqOracle *o)                                       /* h4m:1073-1130 */
{
    Plane *Y = &o->pl[0], *U = &o->pl[1], *V = &o->pl[2];
    uint32_t run = 0;
    for (int by = 0; by < Y->vb; ++by)
        for (int bx = 0; bx < Y->hb; ++bx) {
            uint8_t *e = ent(Y, by, bx);
            if (run) { e[1] = 0; --run; continue; }
            int32_t k = sym(&o->c_bn, &o->bn[0]) & 0xFFFF;
            if ((int16_t)k == 0) run = (uint32_t)sym(&o->c_run, &o->bnr[0]);
            e[1] = (uint8_t)k;
        }
    run = 0;
    for (int by = 0; by < U->vb; ++by)
        for (int bx = 0; bx < U->hb; ++bx) {
            uint8_t *eu = ent(U, by, bx), *ev = ent(V, by, bx);
            if (run) { eu[1] = ev[1] = 0; --run; continue; }
            int32_t k = sym(&o->c_bn, &o->bn[1]) & 0xFFFF;
            if ((int16_t)k == 0) run = (uint32_t)sym(&o->c_run, &o->bnr[1]);
            eu[1] = k & 0xF;
            ev[1] = (k >> 4) & 0xF;
        }
}

static void ipic_dc(HvqOracle *o)                                          /* h4m:1043-1058, 1132-1164 */
{
    for (int p = 0; p < 3; ++p) {
        Plane *P = &o->pl[p];
        uint32_t run = 0;
        for (int by = 0; by < P->vb; ++by) {
            uint8_t pred = ent(P, by - 1, 0)[0];
            for (int bx = 0; bx < P->hb; ++bx) {
                uint32_t delta = 0;
                if (run) --run;
                else {
                    delta = (uint32_t)sym_sovf(&o->c_dc, &o->dc[p], o->dc_lo, o->dc_hi);
                    if (delta == 0) run = (uint32_t)sym(&o->c_run, &o->rle[p]);
                }
                uint8_t v = (uint8_t)(pred + delta);
                ent(P, by, bx)[0] = v;
                pred = (uint8_t)((v + ent(P, by - 1, bx + 1)[0] + 1) / 2);
            }
        }
    }
}

static void make_nest(HvqOracle *o, int nx, int ny)                        /* h4m:1166-1239 */
{
    Plane *Y = &o->pl[0];
    int cols = Y->hb < o->nest_w ? Y->hb : o->nest_w;
    int rows = Y->vb < o->nest_h ? Y->vb : o->nest_h;
    int mcols = o->nest_w - cols; if (mcols > cols) mcols = cols;
    int mrows = o->nest_h - rows; if (mrows > rows) mrows = rows;
    uint8_t *n = o->nest;
    memset(n, 0, sizeof o->nest);
    for (int r = 0; r < rows; ++r) {
        uint8_t *row = n + r * o->nest_w;
        for (int c = 0; c < cols; ++c) row[c] = (ent(Y, ny + r, nx + c)[0] >> 4) & 0xF;
        for (int c = 0; c < mcols; ++c) row[cols + c] = row[cols - 1 - c];          /* mirror */
    }
    for (int r = 0; r < mrows; ++r)                                                    /* vertical mirror */
        memcpy(n + (rows + r) * o->nest_w, n + (rows - 1 - r) * o->nest_w, (size_t)o->nest_w);
}

static void ipic_plane(HvqOracle *o, int p, uint8_t *dst)                  /* h4m:1433-1518 */
{
    Plane *P = &o->pl[p];
    for (int by = 0; by < P->vb; ++by) {
        const uint8_t *cur = ent(P, by, 0);
        const uint8_t *top = by == 0 ? cur : ent(P, by - 1, 0);                       /* first line: prev aliases curr */
        const uint8_t *bot = (by == P->vb - 1 && P->vb > 1) ? cur : ent(P, by + 1, 0); /* last line: next aliases curr */
        uint8_t left = cur[0];
        for (int bx = 0; bx < P->hb; ++bx) {
            uint8_t v = cur[2 * bx], k = cur[2 * bx + 1];
            const uint8_t *nx = bx + 1 < P->hb ? cur + 2 * (bx + 1) : cur + 2 * bx;
            uint8_t *d = dst + (size_t)by * 4 * P->pw + bx * 4;
            if (k == 0) {
                uint8_t T = (top[2 * bx + 1] & 0x77) ? v : top[2 * bx];
                uint8_t B = (bot[2 * bx + 1] & 0x77) ? v : bot[2 * bx];
                uint8_t R = (nx[1] & 0x77) ? v : nx[0];
                weight_block(d, P->pw, v, T, B, left, R);
                left = v;
            } else if (k == 8) {
                fill_block(d, P->pw, v);
                left = v;
            } else {
                intra_aot_block(o, p, d, P->pw, v, k);
                left = nx[0];                                                          /* h4m:1453-1454 */
            }
        }
    }
}

API void hvqo_decode_ipic(HvqOracle *o, const uint8_t *pic, uint8_t *present)   /* h4m:1970-2016 */
{
    o->dc_shift = pic[0];
    o->unk_shift = pic[1];
    int nx = (int)rd16(pic + 4), ny = (int)rd16(pic + 6);
    const uint8_t *tab = pic + 8, *data = pic + 8 + 0x40;
    common_sections(o, data, tab);
    for (int p = 0; p < 3; ++p) o->rle[p] = section_bits(data, tab, 13 + p);
    code_read(&o->c_bn, &o->bn[0], 0, 0);
    code_read(&o->c_run, &o->bnr[0], 0, 0);
    code_read(&o->c_dc, &o->dc[0], 1, o->dc_shift);
    code_read(&o->c_bt, &o->bt[0], 0, 2);
    o->dc_hi = (int32_t)((uint32_t)0x7F << o->dc_shift);
    o->dc_lo = (int32_t)((uint32_t)-0x80 << o->dc_shift);
    ipic_kinds(o);
    ipic_dc(o);
    make_nest(o, nx, ny);
    for (int p = 0; p < 3; ++p) { ipic_plane(o, p, present); present += o->pl[p].samples; }
}

/* ---------- P/B pictures ---------- */
typedef struct { uint32_t value, count; } RunLen;

static void pb_kinds(HvqOracle *o, int mx, int my, uint32_t proc, uint32_t type, uint32_t rl[2])   /* h4m:1670-1740 */
{
    uint8_t tag = (uint8_t)((type << 5) | (proc << 4));
    if (proc == 1) {
        for (int p = 0; p < 3; ++p) {
            Plane *P = &o->pl[p];
            uint8_t *e = ent(P, my * P->by_per, mx * P->bx_per);
            for (int j = 0; j < P->nblk; ++j) e[2 * P->moff[j] + 1] = tag;
        }
        return;
    }
    Plane *Y = &o->pl[0];
    uint8_t *e = ent(Y, my * Y->by_per, mx * Y->bx_per);
    for (int j = 0; j < Y->nblk; ++j) {
        uint8_t *t = &e[2 * Y->moff[j] + 1];
        if (rl[0]) { *t = tag; --rl[0]; continue; }
        int16_t k = (int16_t)sym(&o->c_bn, &o->bn[0]);
        if (k) *t = (uint8_t)(tag | k);
        else { *t = tag; rl[0] = (uint32_t)sym(&o->c_run, &o->bnr[0]); }
    }
    Plane *U = &o->pl[1], *V = &o->pl[2];
    uint8_t *eu = ent(U, my * U->by_per, mx * U->bx_per), *ev = ent(V, my * V->by_per, mx * V->bx_per);
    for (int j = 0; j < U->nblk; ++j) {
        uint8_t *tu = &eu[2 * U->moff[j] + 1], *tv = &ev[2 * U->moff[j] + 1];
        if (rl[1]) { *tu = *tv = tag; --rl[1]; continue; }
        int16_t k = (int16_t)sym(&o->c_bn, &o->bn[1]);
        if (k) { *tu = (uint8_t)(tag | (k & 0xF)); *tv = (uint8_t)(tag | ((k >> 4) & 0xF)); }
        else { *tu = *tv = tag; rl[1] = (uint32_t)sym(&o->c_run, &o->bnr[1]); }
    }
}

static void pb_pass1(HvqOracle *o)                                         /* h4m:1545-1622, 1649-1668, 1742-1776 */
{
    static const uint32_t step[2][3] = { { 1, 2, 0 }, { 2, 0, 1 } };
    RunLen type = { 0, 0 }, proc = { 0, 0 };
    if (o->mproc.base) { proc.value = take1(&o->mproc); proc.count = (uint32_t)sym_uovf(&o->c_mcb, &o->mproc); }
    if (o->mtype.base) { type.value = take(&o->mtype, 2); type.count = (uint32_t)sym_uovf(&o->c_mcb, &o->mtype); }
    uint32_t rl[2] = { 0, 0 };
    uint32_t pbdc[3] = { 0x7F, 0x7F, 0x7F };
    for (int my = 0; my < o->h / 8; ++my)
        for (int mx = 0; mx < o->w / 8; ++mx) {
            if (type.count == 0) {
                type.value = step[take1(&o->mtype)][type.value % 3];
                type.count = (uint32_t)sym_uovf(&o->c_mcb, &o->mtype);
            }
            --type.count;
            if (type.value == 0) {
                for (int p = 0; p < 3; ++p) {
                    Plane *P = &o->pl[p];
                    uint8_t *e = ent(P, my * P->by_per, mx * P->bx_per);
                    for (int j = 0; j < P->nblk; ++j) {
                        pbdc[p] += (uint32_t)sym_sovf(&o->c_dc, &o->dc[p], o->dc_lo, o->dc_hi);
                        e[2 * P->moff[j]] = (uint8_t)pbdc[p];
                    }
                }
                pb_kinds(o, mx, my, 0, 0, rl);
            } else {
                pbdc[0] = pbdc[1] = pbdc[2] = 0x7F;
                if (proc.count == 0) { proc.value ^= 1; proc.count = (uint32_t)sym_uovf(&o->c_mcb, &o->mproc); }
                --proc.count;
                pb_kinds(o, mx, my, proc.value, type.value, rl);
            }
        }
}

static void mvec(HvqOracle *o, int32_t *acc, Bits *b, int rbits)          /* h4m:1846-1860 */
{
    int32_t lim = (int32_t)(1u << (rbits + 5));
    int32_t v = (int32_t)((uint32_t)sym(&o->c_mv, b) << rbits);
    for (int i = rbits - 1; i >= 0; --i) v += (int32_t)(take1(b) << i);
    *acc += v;
    if (*acc >= lim) *acc -= lim << 1;
    else if (*acc < -lim) *acc += lim << 1;
}

static void pb_intra_mcb(HvqOracle *o, int mx, int my, uint8_t *const base[3])   /* h4m:1789-1827 */
{
    for (int p = 0; p < 3; ++p) {
        Plane *P = &o->pl[p];
        const uint8_t *e = ent(P, my * P->by_per, mx * P->bx_per);
        uint8_t *d0 = base[p] + (size_t)my * (8 >> P->hshift) * P->pw + mx * (8 >> P->wshift);
        for (int j = 0; j < P->nblk; ++j) {
            const uint8_t *c = e + 2 * P->moff[j];
            uint8_t v = c[0];
            int k = c[1] & 0xF;
            uint8_t *d = d0 + P->poff[j];
            if (k == 0) {
                const uint8_t *t = c - 2 * P->stride, *b = c + 2 * P->stride, *l = c - 2, *r = c + 2;
                weight_block(d, P->pw, v, (t[1] & 0x77) ? v : t[0], (b[1] & 0x77) ? v : b[0],
                             (l[1] & 0x77) ? v : l[0], (r[1] & 0x77) ? v : r[0]);
            } else if (k == 8) fill_block(d, P->pw, v);
            else intra_aot_block(o, p, d, P->pw, v, k);
        }
    }
}

/* h4m:1327-1355 (proc = 1) and h4m:1862-1910 (proc = 0) */
static void pb_inter_mcb(HvqOracle *o, int mx, int my, int proc, int32_t rx, int32_t ry,
                         uint8_t *const base[3], const uint8_t *const ref[3])
{
    const uint8_t *window;
    if (o->landscape) window = ref[0] + rx / 2 + (ry / 2 - 16) * o->pl[0].pw - 32;
    else              window = ref[0] + rx / 2 + (ry / 2 - 32) * o->pl[0].pw - 16;
    for (int p = 0; p < 3; ++p) {
        Plane *P = &o->pl[p];
        const uint8_t *e = ent(P, my * P->by_per, mx * P->bx_per);
        uint8_t *d0 = base[p] + (size_t)my * (8 >> P->hshift) * P->pw + mx * (8 >> P->wshift);
        int32_t pdx = rx >> P->wshift, pdy = ry >> P->hshift;
        int hx = o->is15 ? (pdx & 1) : (rx & 1), hy = o->is15 ? (pdy & 1) : (ry & 1);   /* h4m:1337-1343, 1890-1896 */
        const uint8_t *s0 = ref[p] + (pdy >> 1) * P->pw + (pdx >> 1);
        for (int j = 0; j < P->nblk; ++j) {
            int k = proc ? 0 : (e[2 * P->moff[j] + 1] & 0xF);
            uint8_t *d = d0 + P->poff[j];
            const uint8_t *s = s0 + P->poff[j];
            if (k == 6) literal_block(o, p, d, P->pw);
            else if (k == 0) mc_block(d, P->pw, s, P->pw, hx, hy);
            else predi_aot_block(o, p, d, s, P->pw, k, window, o->pl[0].pw, hx, hy);
        }
    }
}

API void hvqo_decode_bpic(HvqOracle *o, const uint8_t *pic, uint8_t *present,
                          const uint8_t *past, const uint8_t *future)     /* h4m:2018-2056, 1912-1968 */
{
    o->dc_shift = pic[0];
    o->unk_shift = pic[1];
    o->res[0] = pic[2]; o->res[2] = pic[3]; o->res[1] = pic[4]; o->res[3] = pic[5];
    const uint8_t *tab = pic + 8, *data = pic + 8 + 0x44;
    common_sections(o, data, tab);
    o->mvh = section_bits(data, tab, 13);
    o->mvv = section_bits(data, tab, 14);
    o->mtype = section_bits(data, tab, 15);
    o->mproc = section_bits(data, tab, 16);
    code_read(&o->c_bn, &o->bn[0], 0, 0);
    code_read(&o->c_run, &o->bnr[0], 0, 0);
    code_read(&o->c_dc, &o->dc[0], 1, o->dc_shift);
    code_read(&o->c_bt, &o->bt[0], 0, 2);
    code_read(&o->c_mv, &o->mvh, 1, 0);
    code_read(&o->c_mcb, &o->mtype, 0, 0);
    o->dc_hi = (int32_t)((uint32_t)0x7F << o->dc_shift);
    o->dc_lo = (int32_t)((uint32_t)-0x80 << o->dc_shift);

    pb_pass1(o);

    uint8_t *base[3];
    const uint8_t *pastp[3], *futp[3];
    uint32_t off = 0;
    for (int p = 0; p < 3; ++p) { base[p] = present + off; pastp[p] = past + off; futp[p] = future + off; off += o->pl[p].samples; }
    int cur_ref = -1;
    int32_t mh = 0, mv = 0;
    Plane *Y = &o->pl[0];
    for (int my = 0; my < o->h / 8; ++my)
        for (int mx = 0; mx < o->w / 8; ++mx) {
            uint8_t tag = ent(Y, my * Y->by_per, mx * Y->bx_per)[1];
            int t = (tag >> 5) & 3;
            if (t == 0) { pb_intra_mcb(o, mx, my, base); continue; }
            int r = t - 1;
            if (r != cur_ref) { cur_ref = r; mh = mv = 0; }
            mvec(o, &mh, &o->mvh, o->res[r]);
            mvec(o, &mv, &o->mvv, o->res[2 + r]);
            int32_t rx = (int32_t)((uint32_t)mx * 16u + (uint32_t)mh), ry = (int32_t)((uint32_t)my * 16u + (uint32_t)mv);
            pb_inter_mcb(o, mx, my, (tag >> 4) & 1, rx, ry, base, r == 0 ? pastp : futp);
        }
}

API void hvqo_decode_ppic(HvqOracle *o, const uint8_t *pic, uint8_t *present, const uint8_t *past)   /* h4m:2058-2061 */
{
    hvqo_decode_bpic(o, pic, present, past, present);
}

/* ---------- container + picture rotation (h4m:2078-2138, 2427-2537; format SURVEY App. B) ---------- */
typedef struct { HvqOracle *o; uint8_t *buf[3]; /* past, present, future */ } Play;

static int play_open(Play *pl, const uint8_t *f, size_t n)
{
    if (n < 0x44) return -1;
    int v15 = !memcmp(f, "HVQM4 1.5\0\0\0\0\0\0\0", 16), v13 = !memcmp(f, "HVQM4 1.3\0\0\0\0\0\0\0", 16);
    if (!v15 && !v13) return -2;
    pl->o = hvqo_create((int)rd16(f + 0x34), (int)rd16(f + 0x36), f[0x38], f[0x39], v15);
    for (int i = 0; i < 3; ++i) pl->buf[i] = calloc(1, pl->o->picsize + 64);
    return 0;
}

static void play_close(Play *pl)
{
    for (int i = 0; i < 3; ++i) free(pl->buf[i]);
    hvqo_destroy(pl->o);
}

static void play_picture(Play *pl, int ftype, const uint8_t *pic)
{
    uint8_t *t;
    if (ftype != 0x30) { t = pl->buf[0]; pl->buf[0] = pl->buf[2]; pl->buf[2] = t; }
    if (ftype == 0x10) hvqo_decode_ipic(pl->o, pic, pl->buf[1]);
    else if (ftype == 0x20) hvqo_decode_ppic(pl->o, pic, pl->buf[1], pl->buf[0]);
    else hvqo_decode_bpic(pl->o, pic, pl->buf[1], pl->buf[0], pl->buf[2]);
}

static void play_after(Play *pl, int ftype)
{
    uint8_t *t;
    if (ftype != 0x30) { t = pl->buf[1]; pl->buf[1] = pl->buf[2]; pl->buf[2] = t; }
}

typedef void (*pic_cb)(Play *pl, int ftype, const uint8_t *pic, void *user);

static int walk(const uint8_t *f, size_t n, Play *pl, int max_pics, pic_cb cb, void *user)
{
    uint32_t gops = rd32(f + 0x18);
    size_t pos = 0x44;
    int pics = 0;
    /* the bit reader may look a few bytes past a picture: decode from a padded copy */
    uint8_t *copy = malloc(n + 16);
    memcpy(copy, f, n); memset(copy + n, 0, 16);
    for (uint32_t g = 0; g < gops && pos + 20 <= n; ++g) {
        uint32_t vc = rd32(copy + pos + 8), ac = rd32(copy + pos + 12);
        pos += 20;
        for (uint32_t v = 0, a = 0; (v < vc || a < ac) && pos + 8 <= n;) {
            uint32_t id1 = rd16(copy + pos), id2 = rd16(copy + pos + 2), size = rd32(copy + pos + 4);
            pos += 8;
            if (id1 == 1) {
                if (max_pics >= 0 && pics >= max_pics) { free(copy); return pics; }
                cb(pl, (int)id2, copy + pos + 4, user);
                play_after(pl, (int)id2);
                ++pics; ++v;
            } else ++a;
            pos += size;
        }
    }
    free(copy);
    return pics;
}

typedef struct { uint8_t *out; size_t cap; int n; } Sink;
static void cb_store(Play *pl, int ftype, const uint8_t *pic, void *user)
{
    Sink *s = user;
    play_picture(pl, ftype, pic);
    size_t ps = pl->o->picsize;
    if ((size_t)(s->n + 1) * ps <= s->cap) memcpy(s->out + (size_t)s->n * ps, pl->buf[1], ps);
    s->n++;
}

API int hvqo_decode_clip(const uint8_t *file, size_t n, uint8_t *out, size_t out_cap, int max_pics)
{
    Play pl;
    int rc = play_open(&pl, file, n);
    if (rc) return rc;
    Sink s = { out, out_cap, 0 };
    int pics = walk(file, n, &pl, max_pics, cb_store, &s);
    play_close(&pl);
    return pics;
}

typedef struct { double sec; uint64_t px; } Clock;
static void cb_time(Play *pl, int ftype, const uint8_t *pic, void *user)
{
    Clock *c = user;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    play_picture(pl, ftype, pic);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    c->sec += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    c->px += (uint64_t)pl->o->w * (uint64_t)pl->o->h;
}

API double hvqo_time_clip(const uint8_t *file, size_t n, int reps, uint64_t *pixels)
{
    Clock c = { 0.0, 0 };
    for (int r = 0; r < reps; ++r) {
        Play pl;
        if (play_open(&pl, file, n)) return -1.0;
        walk(file, n, &pl, -1, cb_time, &c);
        play_close(&pl);
    }
    if (pixels) *pixels = c.px;
    return c.sec;
}

/* ---------- display epilogue: YUV 4:2:0 -> RGB24 (h4m:897-926) ----------
 * The reference converts in single-precision float, one operation at a time (no fused multiply-add in its
 * default x86-64 build), clamps and truncates.  `volatile` pins every intermediate to a float so that no
 * compiler contracts or widens the expression. */
static uint8_t clampf255(float f) { return f < 0 ? 0 : f > 255 ? 255 : (uint8_t)f; }

API void hvqo_yuv420_to_rgb(const uint8_t *yuv, int w, int h, uint8_t *rgb)
{
    const uint8_t *yp = yuv, *up = yp + (size_t)w * h, *vp = up + (size_t)w * h / 4;
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            volatile float y = yp[(size_t)i * w + j];
            volatile float u = (float)up[(size_t)(i / 2) * w / 2 + j / 2] - 128.f;
            volatile float v = (float)vp[(size_t)(i / 2) * w / 2 + j / 2] - 128.f;
            volatile float rv = 1.402f * v, gu = 0.34414f * u, gv = 0.71414f * v, bu = 1.772f * u;
            volatile float r = y + rv, g0 = y - gu, g = g0 - gv, b = y + bu;
            *rgb++ = clampf255(r); *rgb++ = clampf255(g); *rgb++ = clampf255(b);
        }
}
