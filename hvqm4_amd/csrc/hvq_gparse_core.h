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

/* Everything in HBM is addressed through global-address-space pointers: a generic (flat) access counts on the LDS
 * counter as well, so every table lookup in LDS would also wait for the chain's outstanding bitstream loads and
 * blob stores -- which defeats the readers' lookahead. */
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define GP_G __attribute__((address_space(1)))
#else
#define GP_G
#endif

/* Chains and the serial steps run WAVE-UNIFORM on the device: every lane of the wave executes the same chain with the
 * same values, so the compiler keeps cursors, reservoirs and counters in scalar registers and runs the bit-level
 * logic on the scalar unit (one scalar branch instead of an exec-mask dance per `if`).  Stores to HBM are issued by all
 * lanes with the same address and value -- measured 8 % faster than masking them down to one lane, which costs five
 * more instructions per store in a chain whose speed is its instruction count. */
#define GP_ST(lvalue, value) do { (lvalue) = (value); } while (0)
/* a value every lane holds alike, said so: what was read from LDS counts as divergent for the compiler, and a chain whose
 * loop bounds come from there is compiled for the vector units with exec masking (measured: slower, the vector units are
 * what the kernel is short of) */
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define GP_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define GP_UNI(x) ((uint32_t)(x))
#endif

#if defined(__HIPCC__)
#define GP_UNROLL _Pragma("unroll")
#else
#define GP_UNROLL
#endif
#define GP_LUT_BITS 8
/* caps of the overflow-symbol loops (the reference has none: h4m:654-677 loop for as long as the stream says).  A
 * signed-overflow value (DC delta, MC-residual scalar) is a handful of symbols in any real stream; a run count is at
 * most the number of macroblocks, 255 per symbol.  The caps are the same in hvq_parse.c, so malformed streams still
 * give identical blobs, and they bound a picture's decode time whatever the stream contains. */
#define GP_SOVF_CAP 4096
#define GP_UOVF_CAP(nmb) ((int)((nmb) / 255u) + 16)
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
/* A chain is one dependent sequence, so a load that is needed at once costs its full latency.  The wave therefore
 * fetches its section GP_BLK dwords at a time -- one coalesced load by GP_BLK lanes -- into an LDS staging slot that
 * belongs to the cursor, and a refill reads the next dword from LDS: one HBM round trip per 128 bytes of bitstream
 * instead of one per 4.  (Keeping look-ahead dwords in registers does not work: the compiler's register copies of a
 * freshly requested dword make it wait for the load on the spot.) */
#define GP_BLK 32
#define GP_SLOTS 24           /* chains: 14 bitstream cursors + 5 list readers (GList), slots 0-18; 20-23: the chains that run
                                 beside the flat path's rings, which lie over slots 0-19 */
#define GP_STAGE_DWORDS (GP_SLOTS * GP_BLK)
/* On the device the staging slots and the six trees (GCode, 640 dwords each) are ONE LDS block, trees behind the slots,
 * so that the flat path (hvq_gparse_flat.h, hvq_gparse.hip) can lay its lanes' rings of 64 dwords over the slots
 * (lanes 0-9) and, in an I picture, over the two trees an I picture does not have (lanes 10-12 in tree GC_MV), and give
 * the vector chains that run beside the lanes of a P/B picture slots inside the tree GC_MCB, which is done with by then:
 * slot s >= GP_SLOTS is 32 dwords at gp_stage + 32 s all the same. */
#define GP_CODE_DWORDS 514u                                              /* sizeof(GCode) / 4 */
#define GP_XLUT_BITS 9                                                   /* the coefficient tree's table for the flat path: 512 entries behind the trees */
#define GP_XLUT_DWORD ((uint32_t)GP_STAGE_DWORDS + 6u * GP_CODE_DWORDS)
#define GP_TREE_DWORD(t) ((uint32_t)GP_STAGE_DWORDS + GP_CODE_DWORDS * (uint32_t)(t))
#define GP_TREE_SLOT(t) ((GP_TREE_DWORD(t) + 1u + 31u) / 32u)            /* first whole slot inside tree t, past its root */

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
__shared__ uint32_t gp_stage[GP_STAGE_DWORDS + 6 * GP_CODE_DWORDS + (1 << GP_XLUT_BITS)];
#define GP_LANE() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#else
static uint32_t gp_stage[GP_STAGE_DWORDS + 6 * GP_CODE_DWORDS + (1 << GP_XLUT_BITS)];
#endif

typedef struct {
    const GP_G uint32_t *d;
    uint32_t nd;               /* readable dwords; everything past them reads as zero */
    uint32_t idx;              /* next dword to consume */
    uint32_t base;             /* first dword held by the staging slot, ~0 = none */
    uint32_t slot;             /* staging slot of this cursor */
    uint64_t acc;              /* left aligned */
    int cnt;
    int live;
} GBits;

GP_FN uint32_t gb_raw(const GBits *b, uint32_t i) { return i < b->nd ? b->d[i] : 0u; }

GP_FN void gb_init(GBits *b, const GP_G uint32_t *d, uint32_t nd, uint64_t byte_off, int live, uint32_t slot)
{
    b->d = d; b->nd = nd; b->live = live; b->slot = slot; b->base = ~0u;
    if (byte_off >= (uint64_t)nd * 4u) { b->idx = nd; b->acc = 0; b->cnt = 32; return; }
    const uint32_t i = (uint32_t)(byte_off >> 2), sh = (uint32_t)(byte_off & 3u) * 8u;
    b->acc = ((uint64_t)__builtin_bswap32(gb_raw(b, i)) << 32) << sh;
    b->cnt = 32 - (int)sh;
    b->idx = i + 1;
}

/* all lanes of the (uniformly executing) wave: stage dwords [blk, blk + GP_BLK) of the section */
GP_FN void gb_fetch(GBits *b, uint32_t blk)
{
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    const int lane = GP_LANE();
    if (lane < GP_BLK) gp_stage[b->slot * GP_BLK + (uint32_t)lane] = gb_raw(b, blk + (uint32_t)lane);
#else
    for (uint32_t l = 0; l < GP_BLK; ++l) gp_stage[b->slot * GP_BLK + l] = gb_raw(b, blk + l);
#endif
    b->base = blk;
}

GP_FN void gb_refill(GBits *b)
{
    if (b->cnt <= 32) {
        const uint32_t blk = b->idx & ~(uint32_t)(GP_BLK - 1);
        if (blk != b->base) gb_fetch(b, blk);
        const uint32_t w = gp_stage[b->slot * GP_BLK + (b->idx & (GP_BLK - 1))];
        b->acc |= (uint64_t)__builtin_bswap32(w) << (32 - b->cnt);
        b->cnt += 32;
        b->idx++;
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

/* sequential reader of a u32 array in HBM (work lists written by the parallel phases), staged through LDS exactly like
 * the bitstream: one coalesced load per GP_BLK entries instead of a dependent load per entry */
typedef struct {
    const GP_G uint32_t *p;
    uint32_t n, idx, base, slot;
} GList;

GP_FN void gl_init(GList *l, const GP_G uint32_t *p, uint32_t n, uint32_t slot) { l->p = p; l->n = n; l->idx = 0; l->base = ~0u; l->slot = slot; }

GP_FN uint32_t gl_next(GList *l)
{
    const uint32_t blk = l->idx & ~(uint32_t)(GP_BLK - 1);
    if (blk != l->base) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        const int lane = GP_LANE();
        if (lane < GP_BLK) gp_stage[l->slot * GP_BLK + (uint32_t)lane] = blk + (uint32_t)lane < l->n ? l->p[blk + (uint32_t)lane] : 0u;
#else
        for (uint32_t k = 0; k < GP_BLK; ++k) gp_stage[l->slot * GP_BLK + k] = blk + k < l->n ? l->p[blk + k] : 0u;
#endif
        l->base = blk;
    }
    return gp_stage[l->slot * GP_BLK + (l->idx++ & (GP_BLK - 1))];
}

/* ------------------------------------------------------------------ prefix trees (h4m:385-394, 604-651) */
typedef struct {
    int root;
    uint8_t sgn, scale, pad[2];         /* leaf byte -> value (h4m:613-617), gc_leaf */
    uint32_t lut[1 << GP_LUT_BITS];     /* [5:0] bits consumed, [7] leaf reached, [31:16] leaf VALUE (int16) or node id;
                                           one LDS read decodes a short code.  Doubles as the tree reader's stack. */
    uint16_t kid[2][256];               /* children of node ids 256..511; a child below 256 is a leaf byte */
} GCode;
typedef char gp_code_size_check[sizeof(GCode) == 4u * GP_CODE_DWORDS ? 1 : -1];

/* value of leaf byte `b`: int16 truncation of the shifted (signed) byte, h4m:613-617 */
GP_FN int32_t gc_leaf(const GCode *c, int b)
{
    const int v = (c->sgn && b > 0x7F) ? b - 256 : b;
    return (int16_t)((uint32_t)v << c->scale);
}

/* serial: read the tree that heads `carrier` (h4m:604-642); iterative form of hvq_parse.c code_node.  The cursor is worked on in
 * registers and put back at the end: through the pointer every step of it was an LDS write */
GP_FN void gc_read(GCode *c, GBits *carrier, int is_signed, int scale, uint32_t *status, uint32_t *flags)
{
    c->root = 0; c->sgn = 0; c->scale = 0;        /* a tree from an empty section: leaf byte 0, value 0 */
    if (!carrier->live) { (void)flags; return; }
    c->sgn = (uint8_t)is_signed; c->scale = (uint8_t)scale;
    GBits b = *carrier;
    uint16_t *stk = (uint16_t *)c->lut;
    int sp = 0, next = 0x100, root = 0;
    for (;;) {
        int val;
        gb_refill(&b);                                      /* at least 33 bits: a leaf is '0' + its byte, looked at in one go */
        const uint32_t nine = (uint32_t)(b.acc >> 55);
        if (!(nine & 0x100u)) {
            val = (int)(nine & 0xFFu);
            b.acc <<= 9; b.cnt -= 9;
        } else {
            b.acc <<= 1; b.cnt -= 1;
            /* a tree over 256 leaf bytes has at most 255 inner nodes (ids 256..510) and so nests at most 255 deep;
             * anything more is malformed, and would let node 511 become its own child (an endless walk in gsym) */
            if (next >= 511 || sp >= 256) { *status |= GP_ST_BADTREE; c->sgn = 0; c->scale = 0; root = 0; break; }
            const int id = next++;
            stk[sp++] = (uint16_t)id;
            continue;
        }
        int done = 0;
        for (;;) {                                  /* hand the finished subtree to its parent */
            if (sp == 0) { root = val; done = 1; break; }
            const uint16_t top = stk[sp - 1];
            const int id = top & 0x3FF;
            if (!(top & 0x8000u)) { c->kid[0][id - 256] = (uint16_t)val; stk[sp - 1] = (uint16_t)(top | 0x8000u); break; }
            c->kid[1][id - 256] = (uint16_t)val;
            --sp;
            val = id;
        }
        if (done) break;
    }
    c->root = root;
    *carrier = b;
}

/* parallel: thread `tid` of `nthr` fills its share of a first-level table of `bits` index bits */
GP_FN void gc_fill_table(const GCode *c, uint32_t *tab, int bits, int tid, int nthr)
{
    const int root = c->root;
    for (int e = tid; e < (1 << bits); e += nthr) {
        int node = root, d = 0;
        while (node >= 256 && d < bits) { node = c->kid[(e >> (bits - 1 - d)) & 1][node - 256]; ++d; }
        tab[e] = node < 256 ? ((uint32_t)d | 0x80u | ((uint32_t)(uint16_t)gc_leaf(c, node) << 16)) : ((uint32_t)d | ((uint32_t)node << 16));
    }
}

GP_FN void gc_fill_lut(GCode *c, int tid, int nthr) { gc_fill_table(c, c->lut, GP_LUT_BITS, tid, nthr); }

GP_FN int32_t gsym(const GCode *c, GBits *b)                                   /* h4m:644-651 */
{
    gb_refill(b);
    const uint32_t e = c->lut[b->acc >> (64 - GP_LUT_BITS)];
    const int len = (int)(e & 63u);
    b->acc <<= len;
    b->cnt -= len;
    if (e & 0x80u) return (int16_t)(e >> 16);
    int id = (int)(e >> 16);
    while (id >= 256) {
        if (b->cnt == 0) gb_refill(b);
        id = c->kid[b->acc >> 63][id - 256];
        b->acc <<= 1;
        b->cnt--;
    }
    return gc_leaf(c, id);
}

/* `*fl` gets HVQ_F_CAPPED when the loop ends on its cap, not on the stream (same rule as hvq_parse.c) */
GP_FN int32_t gsym_sovf(const GCode *c, GBits *b, int32_t lo, int32_t hi, uint32_t *fl)      /* h4m:654-664 */
{
    uint32_t total = 0;
    int32_t v;
    int guard = 0;
    do { v = gsym(c, b); total += (uint32_t)v; } while ((v <= lo || v >= hi) && ++guard < GP_SOVF_CAP);
    if (v <= lo || v >= hi) *fl |= HVQ_F_CAPPED;
    return (int32_t)total;
}

/* run lengths of the type / proc runs: ending on the cap is exact and raises no flag (hvq_parse.c sym_uovf: the capped total
 * exceeds the picture's macroblocks, the run covers all that is left either way) */
GP_FN int32_t gsym_uovf(const GCode *c, GBits *b, int cap, uint32_t *fl)        /* h4m:667-677 */
{
    int32_t total = 0, v;
    int guard = 0;
    (void)fl;
    do { v = gsym(c, b); total += v; } while (v >= 0xFF && ++guard < cap);
    return total;
}

/* ------------------------------------------------------------------ per-picture state (LDS on the device) */
/* one lane of the flat symbol decode (hvq_gparse_flat.h): a prefix-coded section decoded front to back into an
 * array of leaf bytes (round 4; int16 values before), independently of every other section */
#define GF_LANES 13
#define GP_TOTS 32                 /* totals of the scan instances 16..47 (hvq_gparse_flat.h) */
typedef struct {
    uint32_t pos;                /* bit position of the next symbol, from the start of the picture */
    uint32_t end;                /* decode while pos < end (heuristic: the next section's header; never past the data) */
    uint32_t cap;                /* symbols the array can take (multiple of 8) */
    uint32_t n;                  /* out: symbols decoded */
    uint32_t tree;               /* GC_* */
    uint32_t off;                /* first symbol in GPic.sym */
} GLane;

typedef struct {
    int hb, vb, stride;
    int bx_per, by_per, nblk;
    uint32_t nblocks, ntiles;
    uint32_t map_off, plane_off;
    uint32_t run_first;          /* index of the plane's first 64-block run in wave_base[] */
    uint32_t blk_first;          /* index of the plane's first entry in pinfo[] */
} GPlane;

enum { GC_BN = 0, GC_RUN, GC_DC, GC_BT, GC_MV, GC_MCB, GC_COUNT };

typedef struct {
    /* job */
    const GP_G uint32_t *d;
    uint32_t nd, len, cap;
    GP_G uint8_t *blob;
    GP_G uint8_t *nest_out;
    int frame_type, is_pb, is_P;
    /* geometry (hvq_parser_create) */
    int w, h, is15, landscape, nest_w, nest_h, wshift, hshift, mw, mh;
    GPlane pl[3];
    uint32_t mv_off, wave_base_off, fixed_bytes, pic_bytes, total_tiles, total_runs, total_blocks;
    /* scratch */
    GP_G uint32_t *clist;        /* [total_blocks] the entries that carry bases, compacted, per plane in consumption order */
    GP_G uint32_t *pinfo;        /* [total_blocks] payload entry of every block, in CONSUMPTION order (GP_ENT) */
    GP_G uint16_t *run_items;    /* [total_runs] */
    GP_G uint16_t *run_pairs;    /* [total_runs] */
    GP_G uint8_t *mbtype;        /* [mw*mh] macroblock type 0..3 */
    GP_G uint8_t *procseq;       /* [mw*mh] proc value of the n-th inter macroblock */
    GP_G uint8_t *mbtag;         /* [mw*mh] (type << 5) | (proc << 4) */
    GP_G uint32_t *cmb;          /* [mw*mh] macroblocks whose block kinds are coded (all but proc 1), in order */
    GP_G uint32_t *t0;           /* [mw*mh] intra (type 0) macroblocks, in order */
    GP_G uint32_t *trun;         /* [mw*mh] type runs: value << 24 | first macroblock */
    GP_G uint32_t *prun;         /* [mw*mh] proc runs: value << 24 | first entry of procseq */
    GP_G uint32_t *part;         /* [GP_PART] partial counts of the parallel phases */
    /* picture */
    int dc_shift, unk_shift, nx, ny;
    int32_t dc_lo, dc_hi;
    uint8_t res[8];              /* h0 h1 v0 v1 0 0 (h4m:2023-2026; indexed by reference 0..2 like hvq_parse.c) */
    uint32_t flags, status;
    uint32_t max_items, max_pairs, pool_dwords, total;
    uint64_t fx_off[3];          /* byte offset of the fixed-length sections (basis words, literal blocks) */
    uint32_t nchain[3];          /* entries in clist per plane */
    uint32_t ncoded, ntype0;     /* entries of cmb / t0 */
    uint32_t ntrun, nprun, pend; /* entries of trun / prun; procseq entries the proc runs cover */
    uint32_t nks[2];             /* coded kinds found by the kinds chains (luma, chroma) */
    GBits bn[2], bnr[2], dc[3], bt[3], rle[3], mvh, mvv, mtype, mproc;
    /* flat path (hvq_gparse_flat.h) */
    uint32_t sec_pay[17];        /* payload start (byte) of every section, for the lanes' end heuristics */
    GLane lane[GF_LANES];
    GP_G int16_t *sym;           /* the lanes' symbol arrays: leaf bytes, in regions laid out (offsets, capacities) in 2-byte units */
    GP_G uint32_t *val;          /* grouped DC-buffer values, per plane at val_off[] */
    uint32_t val_off[3], nv[3];
    uint32_t tot[GP_TOTS];       /* totals of the scans */
    uint32_t retry;              /* the flat path cannot serve this picture: decode it with the chains */
    uint32_t spins;              /* development aid: rounds the decode wave waited for the staging wave */
    uint32_t prof[3];
} GPic;

/* part[]: [0, 512) one word per thread, [512, 992) a second word per thread, [992, 1024) single-purpose slots,
 * [1024, 4096) GP_EP: per plane and thread, the partial sums of the payload emission */
#define GP_PART 16384
#define GP_PART2 512
#define GP_MISC 992
#define GP_EP(plane, which, tid) (1024 + ((plane) * 2 + (which)) * 512 + (tid))
/* part[GP_MISC + 16 + k]: HVQ_F_CAPPED of the chains that decode overflow symbols, one word each (no two waves share one):
 * k = 0-2 I-picture DC planes, 3-5 P/B intra DC planes, 6-8 MC-residual scalars, 9 type runs, 10 proc runs */
#define GP_CAPW(k) (GP_MISC + 16 + (k))
#define GP_NCAPW 11

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

/* capacities of the flat path's arrays (symbols per lane, multiples of 8 so that every array starts 16-byte aligned):
 * block kinds and their runs one per block, DC buffer three per block (P/B: one value per intra block, two per
 * MC-residual block, and room for overflow symbols), coefficient symbols eight per block, DC runs one per block.
 * A picture that needs more is decoded by the chains. */
#define GP_CAP8(x) (((x) + 15u) & ~7u)
#define GP_CAP_BN(nb)  GP_CAP8(nb)
#define GP_CAP_DC(nb)  GP_CAP8(3u * (nb))
#define GP_CAP_BT(nb)  GP_CAP8(8u * (nb))
#define GP_SYM_TOTAL(tb) (14u * (tb) + 13u * 16u)
#define GP_VAL_TOTAL(tb) (3u * (tb) + 3u * 16u)

/* bytes of scratch one picture of this geometry needs */
GP_FN uint32_t gp_scratch_bytes(uint32_t total_blocks, uint32_t total_runs, uint32_t nmb)
{
    return GP_ALIGN16(4u * total_blocks) * 2u + GP_ALIGN16(2u * total_runs) * 2u + GP_ALIGN16(nmb + 16u) * 3u + GP_ALIGN16(4u * nmb + 16u) * 4u + 4u * GP_PART
         + GP_ALIGN16(2u * GP_SYM_TOTAL(total_blocks)) + GP_ALIGN16(4u * GP_VAL_TOTAL(total_blocks));
}

/* serial (thread 0): geometry exactly as hvq_parser_create lays the blob out */
GP_FN void gp_setup(GPic *g, const HvqParseJob *job)
{
    g->d = (const GP_G uint32_t *)(uintptr_t)job->pic;
    g->nd = job->pic_dwords; g->len = job->len; g->cap = job->cap;
    g->blob = (GP_G uint8_t *)(uintptr_t)job->blob;
    g->nest_out = (GP_G uint8_t *)(uintptr_t)job->nest_out;
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
    GP_G uint8_t *s = (GP_G uint8_t *)(uintptr_t)job->scratch;
    g->pinfo = (GP_G uint32_t *)s;       s += GP_ALIGN16(4u * blocks);
    g->clist = (GP_G uint32_t *)s;       s += GP_ALIGN16(4u * blocks);
    g->run_items = (GP_G uint16_t *)s;   s += GP_ALIGN16(2u * runs);
    g->run_pairs = (GP_G uint16_t *)s;   s += GP_ALIGN16(2u * runs);
    g->mbtype = s;                       s += GP_ALIGN16(nmb + 16u);
    g->procseq = s;                      s += GP_ALIGN16(nmb + 16u);
    g->mbtag = s;                        s += GP_ALIGN16(nmb + 16u);
    g->cmb = (GP_G uint32_t *)s;         s += GP_ALIGN16(4u * nmb + 16u);
    g->t0 = (GP_G uint32_t *)s;          s += GP_ALIGN16(4u * nmb + 16u);
    g->trun = (GP_G uint32_t *)s;        s += GP_ALIGN16(4u * nmb + 16u);
    g->prun = (GP_G uint32_t *)s;        s += GP_ALIGN16(4u * nmb + 16u);
    g->part = (GP_G uint32_t *)s;        s += 4u * GP_PART;
    g->sym = (GP_G int16_t *)s;          s += GP_ALIGN16(2u * GP_SYM_TOTAL(blocks));
    g->val = (GP_G uint32_t *)s;
    g->retry = 0; g->ncoded = 0; g->ntype0 = 0; g->ntrun = 0; g->nprun = 0; g->pend = 0; g->spins = 0;
    g->flags = 0; g->status = 0; g->max_items = 0; g->max_pairs = 0; g->pool_dwords = 0; g->total = 0;
    if (g->cap < g->fixed_bytes || g->len < 8 + 0x44 + 4) g->status |= GP_ST_BADARG;
    for (int k = 0; k < GP_NCAPW; ++k) GP_ST(g->part[GP_CAPW(k)], 0u);
}

/* sections (h4m:1061-1071, 1979-1993, 2030-2044): byte offset of the payload of section i, *live as in hvq_parse.c */
GP_FN uint64_t gp_section(const GPic *g, uint32_t data_off, uint32_t tab_off, int i, int *live)
{
    const uint64_t s = (uint64_t)data_off + gp_be32(g, tab_off + 4u * (uint32_t)i);
    if (s + 4 > g->len) { *live = 0; return g->len; }
    *live = gp_be32(g, s) != 0;
    return s + 4;
}

GP_FN void gp_section_bits(GPic *g, GBits *b, uint32_t data_off, uint32_t tab_off, int i, uint32_t slot)
{
    int live;
    const uint64_t s = gp_section(g, data_off, tab_off, i, &live);
    g->sec_pay[i] = (uint32_t)s;
    gb_init(b, g->d, g->nd, s, live, slot);
}

/* serial (thread 0): picture header fields and all section cursors */
GP_FN void gp_sections(GPic *g)
{
    if (g->status) return;
    g->dc_shift = (int)gp_byte(g, 0);
    g->unk_shift = (int)gp_byte(g, 1);
    const uint32_t tab = 8, data = g->is_pb ? 8 + 0x44 : 8 + 0x40;
    for (int i = 0; i < 17; ++i) g->sec_pay[i] = g->len;
    if (g->is_pb) {
        g->res[0] = (uint8_t)gp_byte(g, 2); g->res[1] = (uint8_t)gp_byte(g, 4);
        g->res[2] = (uint8_t)gp_byte(g, 3); g->res[3] = (uint8_t)gp_byte(g, 5);
        g->res[4] = g->res[5] = g->res[6] = g->res[7] = 0;
        g->nx = g->ny = 0;
    } else {
        g->nx = (int)(gp_be32(g, 4) >> 16); g->ny = (int)(gp_be32(g, 4) & 0xFFFFu);
    }
    /* staging slots: bn 0-1, bnr 2-3, dc 4-6, bt 7-9, then rle 10-12 (I) or mvh, mvv, mtype, mproc 10-13 (P/B) */
    for (int i = 0; i < 2; ++i) {
        gp_section_bits(g, &g->bn[i], data, tab, 2 * i, (uint32_t)i);
        gp_section_bits(g, &g->bnr[i], data, tab, 2 * i + 1, 2u + (uint32_t)i);
    }
    for (int k = 0; k < 3; ++k) {
        gp_section_bits(g, &g->dc[k], data, tab, 4 + 3 * k, 4u + (uint32_t)k);
        gp_section_bits(g, &g->bt[k], data, tab, 5 + 3 * k, 7u + (uint32_t)k);
        { int live; g->fx_off[k] = gp_section(g, data, tab, 6 + 3 * k, &live); g->sec_pay[6 + 3 * k] = (uint32_t)g->fx_off[k]; }
    }
    if (g->is_pb) {
        gp_section_bits(g, &g->mvh, data, tab, 13, 10u);
        gp_section_bits(g, &g->mvv, data, tab, 14, 11u);
        gp_section_bits(g, &g->mtype, data, tab, 15, 12u);
        gp_section_bits(g, &g->mproc, data, tab, 16, 13u);
    } else {
        for (int k = 0; k < 3; ++k) gp_section_bits(g, &g->rle[k], data, tab, 13 + k, 10u + (uint32_t)k);
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
    GP_ST(g->part[GP_MISC + t], st); GP_ST(g->part[GP_MISC + GC_COUNT + t], fl);
}

GP_FN void gp_collect_tree_status(GPic *g, int ntrees)      /* serial (thread 0), after the tree reads */
{
    if (g->status) return;
    for (int t = 0; t < ntrees; ++t) { g->status |= g->part[GP_MISC + t]; g->flags |= g->part[GP_MISC + GC_COUNT + t]; }
}

/* thread `tid`'s contiguous share [lo, hi) of n items */
GP_FN void gp_chunk(uint32_t n, int tid, int nthr, uint32_t *lo, uint32_t *hi)
{
    const uint32_t per = (n + (uint32_t)nthr - 1) / (uint32_t)nthr;
    *lo = per * (uint32_t)tid < n ? per * (uint32_t)tid : n;
    *hi = *lo + per < n ? *lo + per : n;
}

/* ------------------------------------------------------------------ blob helpers */
GP_FN GP_G uint8_t *gp_map_ent(const GPic *g, int plane, int by, int bx)
{
    return g->blob + g->pl[plane].map_off + 2u * ((uint32_t)(by + 1) * (uint32_t)g->pl[plane].stride + (uint32_t)(bx + 1));
}

/* parallel: maps = zero inside, {0x7F,0xFF} on the border (h4m:951-955, 1001-1040); P/B: motion vectors = 0 */
GP_FN void gp_init_maps(const GPic *g, int tid, int nthr)
{
    if (g->status) return;
    for (int i = 0; i < 3; ++i) {
        const GPlane *q = &g->pl[i];
        GP_G uint16_t *m = (GP_G uint16_t *)(g->blob + q->map_off);
        /* rows dealt to the waves, columns to the lanes: no division per entry (nthr is a multiple of 64) */
        const uint32_t rows = (uint32_t)q->vb + 2u, stride = (uint32_t)q->stride;
        for (uint32_t r = (uint32_t)tid >> 6; r < rows; r += (uint32_t)nthr >> 6)
            for (uint32_t c = (uint32_t)tid & 63u; c < stride; c += 64u) {
                const int border = r == 0 || r == rows - 1 || c == 0 || c == stride - 1;
                m[r * stride + c] = border ? (uint16_t)0xFF7Fu : (uint16_t)0;
            }
    }
    if (g->is_pb) {
        GP_G uint32_t *mv = (GP_G uint32_t *)(g->blob + g->mv_off);
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
            GP_G uint8_t *row = gp_map_ent(g, 0, by, 0);
            for (int bx = 0; bx < Y->hb; ++bx) {
                if (run) { --run; continue; }
                const int32_t k = gsym(c_bn, &bn) & 0xFFFF;
                if ((int16_t)k == 0) run = (uint32_t)gsym(c_run, &bnr);
                else GP_ST(row[2 * bx + 1], (uint8_t)k);
            }
        }
    } else {
        const GPlane *C = &g->pl[1];
        for (int by = 0; by < C->vb; ++by) {
            GP_G uint8_t *ru = gp_map_ent(g, 1, by, 0), *rv = gp_map_ent(g, 2, by, 0);
            for (int bx = 0; bx < C->hb; ++bx) {
                if (run) { --run; continue; }
                const int32_t k = gsym(c_bn, &bn) & 0xFFFF;
                if ((int16_t)k == 0) run = (uint32_t)gsym(c_run, &bnr);
                else { GP_ST(ru[2 * bx + 1], (uint8_t)(k & 0xF)); GP_ST(rv[2 * bx + 1], (uint8_t)((k >> 4) & 0xF)); }
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
    uint32_t run = 0, fl = 0;
    for (int by = 0; by < q->vb; ++by) {
        GP_G uint8_t *row = gp_map_ent(g, i, by, 0);
        uint8_t pred = by ? rowbuf[0] : 0x7F;
        for (int bx = 0; bx < q->hb; ++bx) {
            uint32_t delta = 0;
            if (run) --run;
            else {
                delta = (uint32_t)gsym_sovf(c_dc, &dc, lo, hi, &fl);
                if (delta == 0) run = (uint32_t)gsym(c_run, &rle);
            }
            const uint8_t v = (uint8_t)(pred + delta);                  /* uint8 wrap: h4m:1145-1149 */
            GP_ST(row[2 * bx], v);
            pred = (uint8_t)((v + rowbuf[bx + 1] + 1) / 2);
            rowbuf[bx] = v;
        }
    }
    GP_ST(g->part[GP_CAPW(i)], fl);
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
    /* flat indexing of the bordered map like the reference (h4m:1169); clamped only when the window would leave the array */
    if ((ny + rows - 1) * Y->stride + nx + cols - 1 > Y->stride * (Y->vb + 1) - 2) {
        if (nx + cols > Y->hb) nx = Y->hb - cols;
        if (ny + rows > Y->vb) ny = Y->vb - rows;
        clamped = 1;
    }
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
    GP_G uint32_t *wave_base = (GP_G uint32_t *)(g->blob + g->wave_base_off);
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

GP_FN void gp_layout_finish(GPic *g, uint32_t off, uint32_t fl, uint32_t mi, uint32_t mp);

/* serial L2 (thread 0): exclusive scan of the runs, per-tile maxima, sizes, overflow check, header */
GP_FN void gp_layout_scan(GPic *g, int nthr)
{
    if (g->status) return;
    GP_G uint32_t *wave_base = (GP_G uint32_t *)(g->blob + g->wave_base_off);
    uint32_t off = 0, fl = 0, mi = 0, mp = 0, ti = 0, tp = 0;
    for (int t = 0; t < nthr; ++t) fl |= g->part[t];
    for (uint32_t r = 0; r < g->total_runs; ++r) {
        if (r % (HVQ_TILE_BLOCKS / 64) == 0) { ti = 0; tp = 0; }
        const uint32_t s = wave_base[r];
        GP_ST(wave_base[r], off);
        off += s;
        ti += g->run_items[r]; tp += g->run_pairs[r];
        if (ti > mi) mi = ti;
        if (tp > mp) mp = tp;
    }
    if (g->is_pb) { for (int t = 0; t < nthr; ++t) fl |= g->part[GP_PART2 + t]; }      /* HVQ_F_SELF_REF, gp_tags_assign */
    else fl |= g->part[GP_MISC + 2 * GC_COUNT];                                      /* nest origin clamp, gp_nest */
    gp_layout_finish(g, off, fl, mi, mp);
}

/* serial (uniform): sizes, overflow check, header, from the totals of the layout */
GP_FN void gp_layout_finish(GPic *g, uint32_t off, uint32_t fl, uint32_t mi, uint32_t mp)
{
    g->flags |= fl;
    g->max_items = mi; g->max_pairs = mp; g->pool_dwords = off;
    uint64_t total = (uint64_t)g->fixed_bytes + 4u * (uint64_t)off;
    total = GP_ALIGN16(total);
    if (total > g->cap || off >= (1u << 22)) { g->status |= GP_ST_OVERFLOW; return; }
    g->total = (uint32_t)total;
    /* header (hvq_parse.c fill_header); nest_off stays 0: the nest travels separately */
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) != 0) return;      /* one lane writes the header */
#endif
    GP_G HvqPicHeader *h = (GP_G HvqPicHeader *)g->blob;
    GP_G uint32_t *hw = (GP_G uint32_t *)g->blob;
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

/* payload entry of one block: [21:0] pool offset (dwords), [29:22] number of bases, [31:30] mode */
#define GP_ENT(off, nb, mode) ((uint32_t)(off) | ((uint32_t)(nb) << 22) | ((uint32_t)(mode) << 30))
#define GP_MODE_NONE    0u
#define GP_MODE_LITERAL 1u
#define GP_MODE_BASES   2u     /* intra AOT: `nb` bases */
#define GP_MODE_PREDI   3u     /* MC residual: 2 parameters + `nb` bases */

/* parallel L3: one payload entry per block, stored in the order the payload chain consumes them -- raster order for an
 * I picture (h4m:2011-2015), macroblock order with blocks TL, BL, BR, TR for P/B (h4m:1919-1967) -- so that the chain
 * reads one sequential array instead of chasing map entries and offsets through HBM */
GP_FN void gp_layout_blocks(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    const GP_G uint32_t *wave_base = (const GP_G uint32_t *)(g->blob + g->wave_base_off);
    for (uint32_t r = (uint32_t)tid; r < g->total_runs; r += (uint32_t)nthr) {
        const int i = gp_run_plane(g, r);
        const GPlane *q = &g->pl[i];
        const int ctx = g->is_pb ? 2 : (i == 0 ? 0 : 1);
        const uint32_t b0 = (r - q->run_first) * 64u;
        uint32_t off = wave_base[r];
        uint32_t by = b0 / (uint32_t)q->hb, bx = b0 - by * (uint32_t)q->hb;
        for (uint32_t b = b0; b < b0 + 64u && b < q->nblocks; ++b) {
            const uint32_t t = gp_map_ent(g, i, (int)by, (int)bx)[1];
            uint32_t n, it, pr, f;
            gp_type_info(ctx, t, &n, &it, &pr, &f);
            const uint32_t kind = ctx == 0 ? t : (t & 0xFu);
            const int inter = ctx == 2 && (t & 0x60u);
            uint32_t ent = GP_ENT(off, 0, GP_MODE_NONE);
            if (n) ent = kind == 6 ? GP_ENT(off, 0, GP_MODE_LITERAL)
                       : (inter ? GP_ENT(off, kind - 1, GP_MODE_PREDI) : GP_ENT(off, kind, GP_MODE_BASES));
            uint32_t at = b;
            if (g->is_pb) {
                const uint32_t dy = by % (uint32_t)q->by_per, dx = bx % (uint32_t)q->bx_per;
                const uint32_t mb = (by / (uint32_t)q->by_per) * (uint32_t)g->mw + bx / (uint32_t)q->bx_per;
                at = mb * (uint32_t)q->nblk + (dx ? (dy ? 2u : 3u) : (dy ? 1u : 0u));
                if (q->nblk == 1) at = mb;
            }
            g->pinfo[q->blk_first + at] = ent;
            off += n;
            if (++bx == (uint32_t)q->hb) { bx = 0; ++by; }
        }
    }
}

/* ------------------------------------------------------------------ payloads */
/* The payload of a block interleaves two sources in the reference: fixed-length data (16-bit basis words, 16-byte
 * literal blocks; h4m:543-549, 691-692) whose position is a prefix sum over the blocks before it, and the coefficient
 * symbols of bufTree0 plus, for MC-residual blocks, two DC-buffer values (h4m:726-731, 1405-1406), which only a serial
 * decode can find.  So the chain decodes nothing but symbols -- it leaves the RUNNING COEFFICIENT SUM of every basis in
 * the basis' pool slot -- and all threads afterwards merge the words in (gp_emit_*). */
GP_FN uint32_t gp_ent_fx_bytes(uint32_t ent)
{
    const uint32_t mode = ent >> 30;
    return mode == GP_MODE_LITERAL ? 16u : (mode >= GP_MODE_BASES ? 2u * ((ent >> 22) & 0xFFu) : 0u);
}

/* parallel E1: per thread chunk of each plane's entries: fixed-length bytes, entries with bases */
GP_FN void gp_emit_count(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    for (int i = 0; i < 3; ++i) {
        const GP_G uint32_t *ents = g->pinfo + g->pl[i].blk_first;
        uint32_t lo, hi, bytes = 0, cnt = 0;
        gp_chunk(g->pl[i].nblocks, tid, nthr, &lo, &hi);
        for (uint32_t e = lo; e < hi; ++e) { const uint32_t ent = ents[e]; bytes += gp_ent_fx_bytes(ent); cnt += (ent >> 30) >= GP_MODE_BASES; }
        g->part[GP_EP(i, 0, tid)] = bytes;
        g->part[GP_EP(i, 1, tid)] = cnt;
    }
}

/* serial E2: exclusive scans of the chunk sums */
GP_FN void gp_emit_scan(GPic *g, int nthr)
{
    if (g->status) return;
    for (int i = 0; i < 3; ++i) {
        uint32_t bytes = 0, cnt = 0;
        for (int t = 0; t < nthr; ++t) {
            const uint32_t b = g->part[GP_EP(i, 0, t)], c = g->part[GP_EP(i, 1, t)];
            GP_ST(g->part[GP_EP(i, 0, t)], bytes); GP_ST(g->part[GP_EP(i, 1, t)], cnt);
            bytes += b; cnt += c;
        }
        g->nchain[i] = cnt;
    }
}

/* parallel E3: compact the entries that carry bases (what the chain walks) */
GP_FN void gp_emit_compact(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    for (int i = 0; i < 3; ++i) {
        const GP_G uint32_t *ents = g->pinfo + g->pl[i].blk_first;
        GP_G uint32_t *out = g->clist + g->pl[i].blk_first;
        uint32_t lo, hi, at = g->part[GP_EP(i, 1, tid)];
        gp_chunk(g->pl[i].nblocks, tid, nthr, &lo, &hi);
        for (uint32_t e = lo; e < hi; ++e) { const uint32_t ent = ents[e]; if ((ent >> 30) >= GP_MODE_BASES) out[at++] = ent; }
    }
}

/* chain: coefficient symbols of plane i, I and P/B alike: the running sum of every basis goes into its pool slot */
GP_FN void gp_payload(GPic *g, const GCode *codes, int i)
{
    if (g->status) return;
    GBits bt = g->bt[i];
    const GCode *c_bt = &codes[GC_BT];
    GP_G uint32_t *pool = (GP_G uint32_t *)(g->blob + g->fixed_bytes);
    const uint32_t n = g->nchain[i];
    GList ents;
    gl_init(&ents, g->clist + g->pl[i].blk_first, n, i == 0 ? 14u : 15u);
    for (uint32_t e = 0; e < n; ++e) {
        const uint32_t ent = gl_next(&ents);
        const uint32_t nb = (ent >> 22) & 0xFFu;
        GP_G uint32_t *dst = pool + (ent & 0x3FFFFFu) + ((ent >> 30) == GP_MODE_PREDI ? 2u : 0u);
        uint32_t run = 0;
        for (uint32_t k = 0; k < nb; ++k) { run += (uint32_t)gsym(c_bt, &bt); GP_ST(dst[k], run); }
    }
}

/* chain, concurrent with gp_payload (its own cursor): the two scalars of every MC-residual block of plane i from the DC
 * buffer, continuing where gp_pbdc stopped (h4m:1862-1910, 1405-1406) */
GP_FN void gp_predi_params(GPic *g, const GCode *codes, int i)
{
    if (g->status || !g->is_pb) return;
    GBits dc = g->dc[i];
    const GCode *c_dc = &codes[GC_DC];
    const int32_t lo = g->dc_lo, hi = g->dc_hi;
    const int sh_dc = g->dc_shift & 31, sh_unk = g->unk_shift & 31;
    GP_G uint32_t *pool = (GP_G uint32_t *)(g->blob + g->fixed_bytes);
    const uint32_t n = g->nchain[i];
    GList ents;
    uint32_t fl = 0;
    gl_init(&ents, g->clist + g->pl[i].blk_first, n, i == 0 ? 16u : 17u);   /* Y on one wave, U and V on another */
    for (uint32_t e = 0; e < n; ++e) {
        const uint32_t ent = gl_next(&ents);
        if ((ent >> 30) != GP_MODE_PREDI) continue;
        GP_G uint32_t *dst = pool + (ent & 0x3FFFFFu);
        const int32_t s1 = gsym_sovf(c_dc, &dc, lo, hi, &fl);
        const int32_t s2 = gsym_sovf(c_dc, &dc, lo, hi, &fl);
        GP_ST(dst[0], (uint32_t)(s1 >> sh_dc) << sh_unk);
        GP_ST(dst[1], (uint32_t)(s2 >> sh_dc));
    }
    GP_ST(g->part[GP_CAPW(6 + i)], fl);
}

GP_FN uint32_t gp_be16(const GPic *g, uint64_t off)
{
    if (off + 2 > g->len) return 0;
    const uint32_t i = (uint32_t)(off >> 2), sh = (uint32_t)(off & 3u);
    const uint32_t w0 = __builtin_bswap32(g->d[i]);
    if (sh < 3) return (w0 >> (16 - 8 * sh)) & 0xFFFFu;
    const uint32_t w1 = i + 1 < g->nd ? __builtin_bswap32(g->d[i + 1]) : 0u;
    return ((w0 & 0xFFu) << 8) | (w1 >> 24);
}

/* parallel E4, after the chains: basis word + running sum -> basis dword (h4m:726-731), literal blocks copied */
GP_FN void gp_emit_merge(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    GP_G uint32_t *pool = (GP_G uint32_t *)(g->blob + g->fixed_bytes);
    for (int i = 0; i < 3; ++i) {
        const GP_G uint32_t *ents = g->pinfo + g->pl[i].blk_first;
        uint32_t lo, hi;
        gp_chunk(g->pl[i].nblocks, tid, nthr, &lo, &hi);
        uint64_t fx = g->fx_off[i] + g->part[GP_EP(i, 0, tid)];
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
            if (mode == GP_MODE_PREDI) dst += 2;
            /* four bases at a time: all loads first (they are independent), then the stores -- a thread's speed here is
             * the number of HBM round trips it waits for */
            for (uint32_t k0 = 0; k0 < nb; k0 += 4) {
                uint32_t w[4], r[4];
                for (uint32_t j = 0; j < 4; ++j) {
                    const int live = k0 + j < nb;
                    w[j] = live ? gp_be16(g, fx + 2u * (k0 + j)) : 0u;
                    r[j] = live ? dst[k0 + j] : 0u;
                }
                for (uint32_t j = 0; j < 4; ++j)
                    if (k0 + j < nb) dst[k0 + j] = HVQ_BASIS(w[j], (r[j] + ((w[j] >> 13) & 3u)) & 0x3FFFFu);
            }
            fx += 2u * nb;
        }
    }
}

/* ------------------------------------------------------------------ P/B picture chains */
/* chain: macroblock types from the mtype runs (h4m:1545-1622).  The chain records the RUNS (value, first macroblock);
 * all threads spread them over the type bytes afterwards (gp_runs_expand), and the vector chains walk the runs. */
GP_FN void gp_mbtypes(GPic *g, const GCode *codes)
{
    if (g->status) return;
    GBits b = g->mtype;
    const GCode *c = &codes[GC_MCB];
    const int cap = GP_UOVF_CAP((uint32_t)g->mw * (uint32_t)g->mh);
    uint32_t value = 0, count = 0, fl = 0;
    if (b.live) { value = gb_take(&b, 2); count = (uint32_t)gsym_uovf(c, &b, cap, &fl); }
    const uint32_t n = (uint32_t)g->mw * (uint32_t)g->mh;
    uint32_t m = 0, nr = 0;
    while (m < n) {
        if (count == 0) {
            const uint32_t bit = gb_take(&b, 1);
            const uint32_t v = value & 3u;
            /* step table { {1,2,0,2}, {2,0,1,0} } of hvq_parse.c pb_pass1 */
            value = bit ? (v == 0 ? 2u : (v == 2 ? 1u : 0u)) : (v == 0 ? 1u : (v == 2 ? 0u : 2u));
            count = (uint32_t)gsym_uovf(c, &b, cap, &fl);
            if (count == 0) count = n;            /* the reference's counter wraps below zero: the run never ends */
        }
        const uint32_t len = count < n - m ? count : n - m;
        GP_ST(g->trun[nr], (value << 24) | m); ++nr;
        m += len; count -= len;
    }
    g->ntrun = nr;
    GP_ST(g->part[GP_CAPW(9)], fl);
}

/* chain, concurrent with gp_mbtypes: proc value of the n-th INTER macroblock from the mproc runs (h4m:1649-1668).
 * How many inter macroblocks there are is only known when the types are done, so this decodes runs for as many
 * entries as the picture has macroblocks (an upper bound; surplus entries are never looked at) and stops early once
 * the cursor is past the picture's data, where nothing but zero padding is left to decode. */
GP_FN void gp_mbprocs(GPic *g, const GCode *codes)
{
    if (g->status) return;
    GBits b = g->mproc;
    const GCode *c = &codes[GC_MCB];
    const int cap = GP_UOVF_CAP((uint32_t)g->mw * (uint32_t)g->mh);
    /* this chain decodes past what the picture uses (above): a run length that ended on the cap only counts when the run is
     * used, so the word holds 1 + the first entry of the first such run and gp_result compares it with the inter macroblocks */
    uint32_t value = 0, count = 0, fl = 0, capped_at = 0;
    if (b.live) { value = gb_take(&b, 1); count = (uint32_t)gsym_uovf(c, &b, cap, &fl); if (fl) capped_at = 1u; }
    const uint32_t n = (uint32_t)g->mw * (uint32_t)g->mh;
    uint32_t m = 0, nr = 0;
    while (m < n) {
        if (count == 0) {
            if (b.idx > b.nd + 2u) break;
            value ^= 1u;
            count = (uint32_t)gsym_uovf(c, &b, cap, &fl);
            if (fl && !capped_at) capped_at = m + 1u;
            if (count == 0) count = n;
        }
        const uint32_t len = count < n - m ? count : n - m;
        GP_ST(g->prun[nr], (value << 24) | m); ++nr;
        m += len; count -= len;
    }
    g->nprun = nr; g->pend = m;
    GP_ST(g->part[GP_CAPW(10)], capped_at);
}

/* parallel, after the two chains: the runs spread over mbtype[] and procseq[] */
GP_FN void gp_runs_expand(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    const uint32_t n = (uint32_t)g->mw * (uint32_t)g->mh;
    for (uint32_t r = (uint32_t)tid; r < g->ntrun; r += (uint32_t)nthr) {
        const uint32_t e = g->trun[r], m0 = e & 0xFFFFFFu, m1 = r + 1 < g->ntrun ? g->trun[r + 1] & 0xFFFFFFu : n;
        for (uint32_t m = m0; m < m1; ++m) g->mbtype[m] = (uint8_t)(e >> 24);
    }
    for (uint32_t r = (uint32_t)tid; r < g->nprun; r += (uint32_t)nthr) {
        const uint32_t e = g->prun[r], m0 = e & 0xFFFFFFu, m1 = r + 1 < g->nprun ? g->prun[r + 1] & 0xFFFFFFu : g->pend;
        for (uint32_t m = m0; m < m1; ++m) g->procseq[m] = (uint8_t)(e >> 24);
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
    for (int t = 0; t < nthr; ++t) { const uint32_t c = g->part[t]; GP_ST(g->part[t], run); run += c; }
}

/* parallel T3: tag of every macroblock, written into the type byte of all its blocks (h4m:1670-1690: the kind bits
 * of the coded blocks are OR-ed in later); counts of coded and of intra macroblocks per chunk */
GP_FN void gp_tags_assign(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    uint32_t lo, hi;
    gp_chunk((uint32_t)g->mw * (uint32_t)g->mh, tid, nthr, &lo, &hi);
    uint32_t rank = g->part[tid], fl = 0, ncoded = 0, nt0 = 0;
    for (uint32_t m = lo; m < hi; ++m) {
        const uint32_t type = g->mbtype[m];
        uint32_t tag = 0;
        if (type) {
            const uint32_t proc = g->procseq[rank++];
            tag = (type << 5) | (proc << 4);
            if (g->is_P && type >= 2) fl |= HVQ_F_SELF_REF;
            const int my = (int)(m / (uint32_t)g->mw), mx = (int)(m - (uint32_t)my * (uint32_t)g->mw);
            for (int i = 0; i < 3; ++i) {
                const GPlane *q = &g->pl[i];
                for (int dy = 0; dy < q->by_per; ++dy)
                    for (int dx = 0; dx < q->bx_per; ++dx)
                        gp_map_ent(g, i, my * q->by_per + dy, mx * q->bx_per + dx)[1] = (uint8_t)tag;
            }
        } else ++nt0;
        ncoded += !(tag & 0x10u);
        g->mbtag[m] = (uint8_t)tag;
    }
    g->part[GP_PART2 + tid] = fl;
    g->part[GP_EP(0, 0, tid)] = ncoded;
    g->part[GP_EP(0, 1, tid)] = nt0;
}

/* serial T4: exclusive scans of those counts */
GP_FN void gp_lists_scan(GPic *g, int nthr)
{
    if (g->status) return;
    uint32_t a = 0, b = 0;
    for (int t = 0; t < nthr; ++t) {
        const uint32_t ca = g->part[GP_EP(0, 0, t)], cb = g->part[GP_EP(0, 1, t)];
        GP_ST(g->part[GP_EP(0, 0, t)], a); GP_ST(g->part[GP_EP(0, 1, t)], b);
        a += ca; b += cb;
    }
    g->ncoded = a; g->ntype0 = b;
}

/* parallel T5: the two macroblock lists the chains walk */
GP_FN void gp_lists_write(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    uint32_t lo, hi;
    gp_chunk((uint32_t)g->mw * (uint32_t)g->mh, tid, nthr, &lo, &hi);
    uint32_t a = g->part[GP_EP(0, 0, tid)], b = g->part[GP_EP(0, 1, tid)];
    for (uint32_t m = lo; m < hi; ++m) {
        const uint32_t tag = g->mbtag[m];
        if (!(tag & 0x10u)) g->cmb[a++] = m;
        if (!(tag & 0x60u)) g->t0[b++] = m;
    }
}

/* block j of a macroblock, order TL, BL, BR, TR (h4m:447-455, 862-865) */
GP_FN int gp_dx(int j) { return j >> 1; }
GP_FN int gp_dy(int j) { return (j == 1 || j == 2) ? 1 : 0; }

/* chain: block kinds of the luma plane (which = 0) or of both chroma planes (which = 1) of a P/B picture
 * (h4m:1692-1740).  The coded blocks are the blocks of the macroblocks in cmb, in order: "slots".  A non-zero symbol
 * is the kind of the next slot, a zero symbol leaves that slot and `run` more at kind 0.  The chain only records
 * (slot, kind) of the non-zero ones -- a run is skipped by an addition -- and all threads put them into the map
 * afterwards (gp_kinds_scatter). */
GP_FN void gp_pbkinds(GPic *g, const GCode *codes, int which)
{
    if (g->status) return;
    GBits bn = g->bn[which], bnr = g->bnr[which];
    const GCode *c_bn = &codes[GC_BN], *c_run = &codes[GC_RUN];
    GP_G uint32_t *ks = g->clist + g->pl[which].blk_first;
    const uint32_t total = g->ncoded * (uint32_t)g->pl[which].nblk;
    uint32_t s = 0, n = 0;
    while (s < total) {
        const uint32_t k = (uint32_t)gsym(c_bn, &bn) & 0xFFFFu;
        if ((int16_t)k == 0) s += 1u + (uint32_t)gsym(c_run, &bnr);
        else { GP_ST(ks[n], (s << 8) | (k & 0xFFu)); ++n; ++s; }
    }
    g->nks[which] = n;
}

/* parallel, after the kinds chains: OR the recorded kinds into the type bytes */
GP_FN void gp_kinds_scatter(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    for (int which = 0; which < 2; ++which) {
        const GPlane *q = &g->pl[which];
        const GP_G uint32_t *ks = g->clist + q->blk_first;
        const uint32_t n = g->nks[which];
        for (uint32_t e = (uint32_t)tid; e < n; e += (uint32_t)nthr) {
            const uint32_t v = ks[e], slot = v >> 8, k = v & 0xFFu;
            const uint32_t r = slot / (uint32_t)q->nblk, j = slot - r * (uint32_t)q->nblk;
            const uint32_t m = g->cmb[r], tag = g->mbtag[m];
            const int my = (int)(m / (uint32_t)g->mw), mx = (int)(m - (uint32_t)my * (uint32_t)g->mw);
            const int by = my * q->by_per + gp_dy((int)j), bx = mx * q->bx_per + gp_dx((int)j);
            if (which == 0) gp_map_ent(g, 0, by, bx)[1] = (uint8_t)(tag | k);
            else { gp_map_ent(g, 1, by, bx)[1] = (uint8_t)(tag | (k & 0xFu)); gp_map_ent(g, 2, by, bx)[1] = (uint8_t)(tag | ((k >> 4) & 0xFu)); }
        }
    }
}

/* chain: DC values of the intra macroblocks of plane i (h4m:1742-1776): cumulative within a run of consecutive intra
 * macroblocks; values are recorded in order (all threads place them, gp_dc_scatter); leaves the cursor for the
 * coefficient chain */
GP_FN void gp_pbdc(GPic *g, const GCode *codes, int i)
{
    if (g->status) return;
    const int nblk = g->pl[i].nblk;
    GBits dc = g->dc[i];
    const GCode *c_dc = &codes[GC_DC];
    const int32_t lo = g->dc_lo, hi = g->dc_hi;
    GP_G uint8_t *dv = (GP_G uint8_t *)g->pinfo + g->pl[i].blk_first;
    const uint32_t n = g->ntype0;
    uint32_t pbdc = 0x7F, prev = ~0u, at = 0, fl = 0;
    GList t0;
    gl_init(&t0, g->t0, n, i == 0 ? 14u : 15u);
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t m = gl_next(&t0);
        if (m != prev + 1u) pbdc = 0x7F;                                /* a non-intra macroblock in between resets */
        prev = m;
        for (int j = 0; j < nblk; ++j) {
            pbdc += (uint32_t)gsym_sovf(c_dc, &dc, lo, hi, &fl);
            GP_ST(dv[at], (uint8_t)pbdc); ++at;
        }
    }
    g->dc[i] = dc;
    GP_ST(g->part[GP_CAPW(3 + i)], fl);
}

/* parallel, after the DC chains */
GP_FN void gp_dc_scatter(GPic *g, int tid, int nthr)
{
    if (g->status) return;
    for (int i = 0; i < 3; ++i) {
        const GPlane *q = &g->pl[i];
        const GP_G uint8_t *dv = (const GP_G uint8_t *)g->pinfo + q->blk_first;
        const uint32_t n = g->ntype0 * (uint32_t)q->nblk;
        for (uint32_t e = (uint32_t)tid; e < n; e += (uint32_t)nthr) {
            const uint32_t r = e / (uint32_t)q->nblk, j = e - r * (uint32_t)q->nblk;
            const uint32_t m = g->t0[r];
            const int my = (int)(m / (uint32_t)g->mw), mx = (int)(m - (uint32_t)my * (uint32_t)g->mw);
            gp_map_ent(g, i, my * q->by_per + gp_dy((int)j), mx * q->bx_per + gp_dx((int)j))[0] = dv[e];
        }
    }
}

/* chain: one motion-vector component (comp 0: x from mvh, 1: y from mvv) of every inter macroblock
 * (h4m:1846-1860, 1943-1955); returns HVQ_F_CLAMPED when a target had to be clamped to int16.
 * `comp` is a constant at every call site (the body is specialised per component), a run of macroblocks is walked row segment
 * by row segment (no end-of-row test per vector) and the clamp flag comes from the extremes at the end: the chain is bound by the
 * scalar instructions it issues. */
GP_FN __attribute__((always_inline)) uint32_t gp_mvs_comp(GPic *g, const GCode *codes, const int comp, uint32_t list_slot)
{
    if (g->status) return 0;
    GBits b = comp ? g->mvv : g->mvh;
    const GCode *c = &codes[GC_MV];
    GP_G int16_t *mvs = (GP_G int16_t *)(g->blob + g->mv_off);
    int cur_ref = -1;
    int32_t acc = 0, pos_min = 0, pos_max = 0;
    /* residual bits per reference, in registers: a table read per vector is an LDS round trip on the chain's critical path */
    const int rb0 = (int)GP_UNI(g->res[2 * comp] & 15), rb1 = (int)GP_UNI(g->res[2 * comp + 1] & 15), rb2 = (int)GP_UNI(g->res[2 * comp + 2] & 15);
    const uint32_t mw = GP_UNI(g->mw), n = mw * GP_UNI(g->mh), nr = GP_UNI(g->ntrun);
    /* row of a macroblock by one multiplication: exact for m < 65536 (m * (recip * mw - 2^32) < 2^32 as mw < 65536); the division the
     * compiler builds is 14 scalar instructions per run of macroblocks, and a dense stream's runs are one or two macroblocks long */
    const uint32_t recip = GP_UNI((uint32_t)((0x100000000ull + mw - 1u) / (mw ? mw : 1u)));
    const int small = n <= 65536u && mw > 1u;                            /* mw = 1: the reciprocal does not fit 32 bits */
    GList runs;                                                          /* the type runs (gp_mbtypes) */
    gl_init(&runs, g->trun, nr, list_slot);
    uint32_t e = nr ? GP_UNI(gl_next(&runs)) : 0u;
    for (uint32_t r = 0; r < nr; ++r) {
        const uint32_t nx = r + 1 < nr ? GP_UNI(gl_next(&runs)) : n;
        const uint32_t m0 = e & 0xFFFFFFu, m1 = r + 1 < nr ? nx & 0xFFFFFFu : n;
        const int t = (int)(e >> 24);
        e = nx;
        if (t == 0) continue;                                             /* a run of intra macroblocks */
        const int ref = t - 1;
        if (ref != cur_ref) { cur_ref = ref; acc = 0; }
        const int rbits = ref == 0 ? rb0 : (ref == 1 ? rb1 : rb2);        /* ref = 2 only from a first type value of 3 */
        const int32_t lim = (int32_t)(1u << (rbits + 5));
        uint32_t my = small ? (uint32_t)(((uint64_t)m0 * recip) >> 32) : m0 / mw, mx = m0 - my * mw;
        GP_G int16_t *out = mvs + 2 * m0 + (uint32_t)comp;
        for (uint32_t m = m0; m < m1;) {
            const uint32_t seg = m + (mw - mx) < m1 ? m + (mw - mx) : m1;  /* to the end of the macroblock row or of the run */
            int32_t at = (int32_t)(comp ? my : mx) * 16;                  /* 16 x the macroblock's row resp. column */
            for (; m < seg; ++m) {
                /* symbol and residual bits from one refill: a code from the table is at most 8 bits, the residual at most 15 */
                int32_t v = (int32_t)((uint32_t)gsym(c, &b) << rbits);
                if (rbits) {
                    if (b.cnt < rbits) gb_refill(&b);
                    v += (int32_t)(uint32_t)(b.acc >> (64 - rbits));
                    b.acc <<= rbits; b.cnt -= rbits;
                }
                acc += v;
                if (acc >= lim) acc -= lim << 1;
                else if (acc < -lim) acc += lim << 1;
                const int32_t pos = at + acc;
                if (pos > pos_max) pos_max = pos;
                if (pos < pos_min) pos_min = pos;
                GP_ST(*out, (int16_t)(pos > 32767 ? 32767 : (pos < -32768 ? -32768 : pos)));
                out += 2;
                if (comp == 0) at += 16;
            }
            mx = 0; ++my;
        }
    }
    return (pos_max > 32767 || pos_min < -32768) ? HVQ_F_CLAMPED : 0u;
}

GP_FN uint32_t gp_mvs(GPic *g, const GCode *codes, int comp, uint32_t list_slot)
{
    return comp ? gp_mvs_comp(g, codes, 1, list_slot) : gp_mvs_comp(g, codes, 0, list_slot);
}

/* serial (thread 0): result record */
GP_FN void gp_result(const GPic *g, GP_G HvqParseResult *out, uint32_t extra_flags)
{
    out->status = g->status;
    uint32_t capped = 0;
    for (int k = 0; k < GP_NCAPW - 1; ++k) capped |= g->part[GP_CAPW(k)];
    {   /* proc runs: entries [0, inter macroblocks) are the ones the picture uses */
        const uint32_t at = g->part[GP_CAPW(10)], inter = (uint32_t)g->mw * (uint32_t)g->mh - g->ntype0;
        if (g->is_pb && at && at - 1u < inter) capped |= HVQ_F_CAPPED;
    }
    out->flags = g->flags | extra_flags | capped | (g->is15 ? HVQ_F_IS15 : 0u) | (g->landscape ? HVQ_F_LANDSCAPE : 0u);
    out->max_items = g->max_items; out->max_pairs = g->max_pairs;
    out->pool_dwords = g->pool_dwords; out->total_bytes = g->total;
    out->pad[0] = out->pad[1] = 0;
}

#endif
