/*
 * oracle/hvq_desc_recon.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU interpreter of the product's descriptor blobs (hvqm4_amd/csrc/hvq_desc.h):
 * blob + reference pictures -> reconstructed picture.  It is the executable specification
 * of what the HIP kernels must compute from a blob, and lets the CPU test-suite check the
 * host parse (hvq_parse.c) against the oracle/reference without a GPU.  Never linked into
 * the product; the product has no CPU reconstruction path.
 *
 * Reference lines restated: block kinds h4m:1433-1455 / 1789-1827 / 1862-1910, AOT
 * h4m:679-817, 1358-1420, motion compensation h4m:1242-1294, 1327-1355.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hvq_desc.h"

#define API __attribute__((visibility("default")))

static inline uint8_t clamp255(int32_t x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }
static inline uint8_t mean8(int32_t s) { return clamp255((int32_t)(((uint32_t)s + 4u) / 8u)); }

typedef struct {
    const uint8_t *blob;
    const HvqPicHeader *h;
    const uint8_t *nest;
    int nest_w;
    uint32_t slot_bytes;
    int32_t divt[16];
} Ctx;

static inline uint8_t ref_px(const uint8_t *ref, int64_t a, uint32_t slot_bytes)
{
    if (a < 0) a = 0;
    if (a > (int64_t)slot_bytes - 1) a = (int64_t)slot_bytes - 1;
    return ref[a];
}

static void mc16(const Ctx *c, const uint8_t *ref, int64_t a, int stride, int hx, int hy, uint8_t m[16])
{
    for (int y = 0; y < 4; ++y)
        for (int x = 0; x < 4; ++x) {
            int64_t s = a + (int64_t)y * stride + x;
            int p00 = ref_px(ref, s, c->slot_bytes), p01 = ref_px(ref, s + 1, c->slot_bytes);
            int p10 = ref_px(ref, s + stride, c->slot_bytes), p11 = ref_px(ref, s + stride + 1, c->slot_bytes);
            int v;
            if (!hx && !hy) v = p00;
            else if (hx && !hy) v = (p00 + p01 + 1) / 2;
            else if (!hx) v = (p00 + p10 + 1) / 2;
            else v = (p00 + p01 + p10 + p11 + 2) >> 2;
            m[4 * y + x] = (uint8_t)v;
        }
}

/* accumulate n bases; src = nest bytes (hi=0) or reference picture luma window (hi=1, clamped) */
static int32_t aot(const Ctx *c, const uint32_t *bases, uint32_t n, const uint8_t *src, int64_t origin,
                   int stride, int hi, uint32_t acc[16])
{
    int landscape = (c->h->flags & HVQ_F_LANDSCAPE) != 0;
    memset(acc, 0, 64);
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t d = bases[k];
        uint32_t ol = d & 0x3F, os = (d >> 6) & 0x1F, sl = (d >> 11) & 1, ss = (d >> 12) & 1, neg = (d >> 13) & 1;
        uint32_t sum = d >> 14;
        int64_t o; int xs, ys;
        if (landscape) { o = origin + (int64_t)stride * os + ol; xs = 1 << sl; ys = stride << ss; }
        else           { o = origin + (int64_t)stride * ol + os; xs = 1 << ss; ys = stride << sl; }
        uint8_t e[16], lo = 255, hi_v = 0;
        for (int y = 0; y < 4; ++y)
            for (int x = 0; x < 4; ++x) {
                uint8_t v = hi ? (uint8_t)((ref_px(src, o + (int64_t)y * ys + x * xs, c->slot_bytes) >> 4) & 0xF)
                               : src[o + y * ys + x * xs];
                e[4 * y + x] = v;
                if (v < lo) lo = v;
                if (v > hi_v) hi_v = v;
            }
        int32_t inv = c->divt[(hi_v - lo) & 15];
        if (neg) inv = -inv;
        uint32_t factor = sum * (uint32_t)inv;
        for (int i = 0; i < 16; ++i) acc[i] += factor * e[i];
    }
    uint32_t total = 0;
    for (int i = 0; i < 16; ++i) total += acc[i];
    return (int32_t)total >> 4;
}

API int hvqd_recon(const uint8_t *blob, uint8_t *dst, const uint8_t *ref0, const uint8_t *ref1, uint32_t slot_bytes)
{
    static const int a4[4] = { 2, 0, -1, -1 };
    const HvqPicHeader *h = (const HvqPicHeader *)blob;
    if (h->magic != HVQ_MAGIC) return -1;
    uint8_t nest_bytes[HVQ_NEST_BYTES];                      /* the blob carries the nest nibble-packed */
    if (h->nest_off)
        for (int i = 0; i < HVQ_NEST_BYTES; ++i) nest_bytes[i] = (blob[h->nest_off + (i >> 1)] >> (4 * (i & 1))) & 0xF;
    Ctx c = { blob, h, h->nest_off ? nest_bytes : NULL, (h->flags & HVQ_F_LANDSCAPE) ? 70 : 38, slot_bytes, { 0 } };
    for (int i = 1; i < 16; ++i) c.divt[i] = 0x1000 / (i * 16) * 16;
    const uint32_t *pool = (const uint32_t *)(blob + h->pool_off);
    const uint32_t *wave_base = (const uint32_t *)(blob + h->wave_base_off);
    const int16_t *mvs = h->mv_off ? (const int16_t *)(blob + h->mv_off) : NULL;
    int is_pb = h->pic_kind != HVQ_PIC_I, is15 = (h->flags & HVQ_F_IS15) != 0;
    int lw = h->width;
    /* A P picture with future-referencing (type 2) macroblocks (HVQ_F_SELF_REF): `ref1` is the picture being written
     * (h4m:2058-2061).  Specification of what the back end does: every other macroblock is reconstructed as usual, but into a
     * side buffer; then the macroblocks are walked in raster order like BpicPlaneDec (h4m:1919-1967) -- finished ones move from
     * the side buffer into `dst` (which holds the buffer's previous content), type-2 ones are computed from `dst` as it is at
     * that moment, block by block (TL, BL, BR, TR, then U, then V), sample by sample where the reference copies in place. */
    const int selfref = h->pic_kind == HVQ_PIC_P && (h->flags & HVQ_F_SELF_REF);
    uint8_t *const real_dst = dst;
    uint8_t *side = NULL;
    uint32_t *boff[3] = { NULL, NULL, NULL };
    if (selfref) {
        side = calloc(1, slot_bytes);
        for (int p = 0; p < 3; ++p) boff[p] = malloc(sizeof(uint32_t) * ((size_t)h->hb[p] * h->vb[p] + 1));
        dst = side;
    }
    for (int p = 0; p < 3; ++p) {
        int hb = h->hb[p], vb = h->vb[p], stride = hb + 2;
        int ws = p ? h->wshift : 0, hs = p ? h->hshift : 0;
        int pw = h->width >> ws;
        const uint8_t *map = blob + h->map_off[p];
        uint8_t *plane = dst + h->plane_off[p];
        uint32_t off = 0;
        for (uint32_t b = 0; b < (uint32_t)hb * vb; ++b) {
            if ((b % 64) == 0) off = wave_base[h->tile_first[p] * (HVQ_TILE_BLOCKS / 64) + b / 64];
            int by = (int)(b / hb), bx = (int)(b % hb);
            const uint8_t *e = map + 2 * ((by + 1) * stride + bx + 1);
            uint32_t V = e[0], T = e[1];
            int I_luma = !is_pb && p == 0;
            uint32_t n = hvq_payload_dwords(T, is_pb, I_luma);
            const uint32_t *pay = pool + off;
            if (selfref) boff[p][b] = off;
            off += n;
            uint8_t out[16];
            int inter = is_pb && (T & 0x60);
            if (selfref && ((T >> 5) & 3) == 2) continue;       /* done by the raster-order walk below */
            uint32_t kind = I_luma ? T : (T & 0xF);
            if (!inter) {
                if (kind == 0) {
                    const uint8_t *t = e - 2 * stride, *bo = e + 2 * stride, *l = e - 2, *r = e + 2;
                    int Tt = (t[1] & 0x77) ? (int)V : t[0], Bb = (bo[1] & 0x77) ? (int)V : bo[0];
                    int Rr = (r[1] & 0x77) ? (int)V : r[0];
                    int Ll = is_pb ? ((l[1] & 0x77) ? (int)V : l[0]) : ((l[1] == 0 || l[1] == 8) ? l[0] : (int)V);
                    for (int y = 0; y < 4; ++y) {
                        int rr = a4[y] * (Tt - (int)V) + a4[3 - y] * (Bb - (int)V);
                        for (int x = 0; x < 4; ++x)
                            out[4 * y + x] = mean8(8 * (int)V + rr + a4[x] * (Ll - (int)V) + a4[3 - x] * (Rr - (int)V));
                    }
                } else if (kind == 8) memset(out, (int)V, 16);
                else if (kind == 6) memcpy(out, pay, 16);
                else {
                    uint32_t acc[16];
                    int32_t mean = aot(&c, pay, kind, c.nest, 0, c.nest_w, 0, acc);
                    uint32_t delta = (V << h->unk_shift) - (uint32_t)mean;
                    for (int i = 0; i < 16; ++i) out[i] = clamp255((int32_t)(acc[i] + delta) >> h->unk_shift);
                }
            } else {
                int mx = bx >> (1 - ws), my = by >> (1 - hs);
                int32_t rx = mvs[2 * (my * (int)h->mcb_w + mx)], ry = mvs[2 * (my * (int)h->mcb_w + mx) + 1];
                const uint8_t *ref = ((T >> 5) & 3) == 1 ? ref0 : ref1;
                int32_t pdx = rx >> ws, pdy = ry >> hs;
                int hx = is15 ? (pdx & 1) : (rx & 1), hy = is15 ? (pdy & 1) : (ry & 1);
                int64_t a = (int64_t)h->plane_off[p] + (int64_t)(pdy >> 1) * pw + (pdx >> 1)
                          + (int64_t)(by & (1 - hs)) * 4 * pw + (bx & (1 - ws)) * 4;
                if ((T & 0x10) || kind == 0) mc16(&c, ref, a, pw, hx, hy, out);
                else if (kind == 6) memcpy(out, pay, 16);
                else {
                    int64_t origin = (h->flags & HVQ_F_LANDSCAPE) ? (int64_t)(rx / 2) + (int64_t)(ry / 2 - 16) * lw - 32
                                                                  : (int64_t)(rx / 2) + (int64_t)(ry / 2 - 32) * lw - 16;
                    uint32_t acc[16];
                    uint32_t mean_aot = (uint32_t)aot(&c, pay + 2, kind - 1, ref, origin, lw, 1, acc);
                    uint8_t m[16];
                    mc16(&c, ref, a, pw, hx, hy, m);
                    int32_t s = 8, lo = 255, hi = 0;
                    for (int i = 0; i < 16; ++i) { s += m[i]; if (m[i] < lo) lo = m[i]; if (m[i] > hi) hi = m[i]; }
                    int32_t mean = s / 16;
                    int32_t range = hi - lo;
                    uint32_t addend = pay[0] - mean_aot;
                    uint32_t factor = pay[1] * (uint32_t)(range ? 0x1000 / range : 0);
                    for (int i = 0; i < 16; ++i) {
                        uint32_t r = acc[i] + addend + (uint32_t)((int32_t)m[i] - mean) * factor;
                        out[i] = clamp255(((int32_t)r >> h->unk_shift) + m[i]);
                    }
                }
            }
            for (int y = 0; y < 4; ++y) memcpy(plane + (size_t)(by * 4 + y) * pw + bx * 4, out + 4 * y, 4);
        }
    }
    if (selfref) {
        uint8_t *pic = real_dst;
        const int mw = (int)h->mcb_w, mh = (int)h->mcb_h;
        for (int my = 0; my < mh; ++my)
            for (int mx = 0; mx < mw; ++mx) {
                const uint32_t T = (blob + h->map_off[0])[2 * ((2 * my + 1) * (h->hb[0] + 2) + 2 * mx + 1) + 1];
                if (((T >> 5) & 3) != 2) {
                    for (int p = 0; p < 3; ++p) {
                        const int ws = p ? h->wshift : 0, hs = p ? h->hshift : 0, pw = h->width >> ws, bw = 8 >> ws, bh = 8 >> hs;
                        for (int r = 0; r < bh; ++r) {
                            const size_t o = h->plane_off[p] + (size_t)(my * bh + r) * pw + (size_t)mx * bw;
                            memcpy(pic + o, side + o, (size_t)bw);
                        }
                    }
                    continue;
                }
                const int32_t rx = mvs[2 * (my * mw + mx)], ry = mvs[2 * (my * mw + mx) + 1];
                const int proc = (T >> 4) & 1;
                const int64_t origin = (h->flags & HVQ_F_LANDSCAPE) ? (int64_t)(rx / 2) + (int64_t)(ry / 2 - 16) * lw - 32
                                                                    : (int64_t)(rx / 2) + (int64_t)(ry / 2 - 32) * lw - 16;
                for (int p = 0; p < 3; ++p) {
                    const int ws = p ? h->wshift : 0, hs = p ? h->hshift : 0, pw = h->width >> ws;
                    const int bxp = 2 >> ws, byp = 2 >> hs, nblk = bxp * byp, hb = h->hb[p];
                    const int32_t pdx = rx >> ws, pdy = ry >> hs;
                    const int hx = is15 ? (pdx & 1) : (rx & 1), hy = is15 ? (pdy & 1) : (ry & 1);
                    for (int j = 0; j < nblk; ++j) {
                        const int dx = nblk == 1 ? 0 : (j >> 1), dy = nblk == 1 ? 0 : ((j == 1 || j == 2) ? 1 : 0);
                        const int bx = mx * bxp + dx, by = my * byp + dy;
                        uint8_t *d = pic + h->plane_off[p] + (size_t)by * 4 * pw + (size_t)bx * 4;
                        const int64_t a = (int64_t)h->plane_off[p] + (int64_t)(pdy >> 1) * pw + (pdx >> 1) + (int64_t)dy * 4 * pw + dx * 4;
                        const uint32_t kind = (blob + h->map_off[p])[2 * ((by + 1) * (hb + 2) + bx + 1) + 1] & 0xF;
                        const uint32_t *pay = pool + boff[p][by * hb + bx];
                        if (!proc && kind == 6) {
                            for (int y = 0; y < 4; ++y) memcpy(d + (size_t)y * pw, (const uint8_t *)pay + 4 * y, 4);
                        } else if (proc || kind == 0) {
                            /* _MotionComp in place: every sample is read when it is needed (h4m:1242-1279) */
                            for (int y = 0; y < 4; ++y)
                                for (int x = 0; x < 4; ++x) {
                                    const int64_t sa = a + (int64_t)y * pw + x;
                                    const int p00 = ref_px(pic, sa, slot_bytes), p01 = ref_px(pic, sa + 1, slot_bytes);
                                    const int p10 = ref_px(pic, sa + pw, slot_bytes), p11 = ref_px(pic, sa + pw + 1, slot_bytes);
                                    d[(size_t)y * pw + x] = (uint8_t)(!hx && !hy ? p00 : hx && !hy ? (p00 + p01 + 1) / 2
                                                                      : !hx ? (p00 + p10 + 1) / 2 : (p00 + p01 + p10 + p11 + 2) >> 2);
                                }
                        } else {
                            uint32_t acc[16];
                            const uint32_t mean_aot = (uint32_t)aot(&c, pay + 2, kind - 1, pic, origin, lw, 1, acc);
                            uint8_t m[16];
                            mc16(&c, pic, a, pw, hx, hy, m);
                            int32_t sum = 8, lo = 255, hi = 0;
                            for (int i = 0; i < 16; ++i) { sum += m[i]; if (m[i] < lo) lo = m[i]; if (m[i] > hi) hi = m[i]; }
                            const int32_t mean = sum / 16, range = hi - lo;
                            const uint32_t addend = pay[0] - mean_aot;
                            const uint32_t factor = pay[1] * (uint32_t)(range ? 0x1000 / range : 0);
                            for (int i = 0; i < 16; ++i) {
                                const uint32_t r = acc[i] + addend + (uint32_t)((int32_t)m[i] - mean) * factor;
                                d[(size_t)(i >> 2) * pw + (i & 3)] = clamp255(((int32_t)r >> h->unk_shift) + m[i]);
                            }
                        }
                    }
                }
            }
        free(side);
        for (int p = 0; p < 3; ++p) free(boff[p]);
    }
    return 0;
}
