/*
 * hvq_desc.h -- per-picture descriptor blob: the interface between the host entropy
 * parse (hvq_parse.c) and the gfx950 reconstruction kernels (hvq_kernels.hip).
 *
 * The reference interleaves bit-buffer reads with pixel work (h4m_audio_decode.c:691, 726,
 * 1405-1406, 1950-1951).  Here the serial parse runs first and leaves everything the pixel
 * stage needs in one self-contained, position-independent blob (all offsets are byte offsets
 * from the blob start, every section 16-byte aligned):
 *
 *   HvqPicHeader
 *   map[3]      per plane (hb+2)*(vb+2) entries of {u8 value, u8 type} INCLUDING the border
 *               {0x7F,0xFF} -- byte-identical to the reference's BlockData maps (h4m:432-436,
 *               1001-1040).  I pictures: type = basis-count byte (luma) / nibble (chroma);
 *               P/B pictures: type = [6:5] macroblock type, [4] proc, [3:0] kind (h4m:1296-1304).
 *   mv          P/B only: per 8x8 macroblock {i16 ref_x, i16 ref_y}, the ABSOLUTE half-sample
 *               position of the macroblock in the reference picture (h4m:1954-1955); the
 *               predictor chain of getMVector (h4m:1846-1860) is resolved on the host.
 *   wave_base   one u32 per 64 blocks (HVQ_TILE_BLOCKS/64 per tile): dword index into `pool`
 *               of the first payload of that run of 64 blocks.
 *   pool        u32[]: block payloads in (plane, raster) order.  A block's payload length is a
 *               pure function of its map type (hvq_payload_dwords), so a wavefront finds each
 *               block's payload with one 64-lane prefix scan -- no per-block offsets are stored.
 *                 literal block (kind 6)      : 4 dwords = the 16 samples, row-major
 *                 AOT basis                   : 1 dword  = HVQ_BASIS(word, coef_sum)
 *                 MC-residual ("predi") block : 2 dwords {i32 dc_part, i32 gain_part} + bases
 *   nest        the 70x38 nest of 4-bit values packed two per byte (value n of the reference's
 *               nest_data[] = nibble n), HVQ_NESTP_BYTES; I pictures, and P/B pictures that contain
 *               intra AOT blocks (the nest of the most recent I picture, h4m:1823 -> 1367).
 *
 * A tile is HVQ_TILE_BLOCKS consecutive 4x4 blocks of ONE plane in raster order; one
 * workgroup reconstructs one tile.
 */
#ifndef HVQ_DESC_H
#define HVQ_DESC_H

#include <stdint.h>

#define HVQ_MAGIC        0x34515648u   /* "HVQ4" */
#ifndef HVQ_TILE_BLOCKS
#define HVQ_TILE_BLOCKS  256           /* blocks per tile = threads per workgroup */
#endif
#define HVQ_NEST_BYTES   (70 * 38)
#define HVQ_NESTP_BYTES  (HVQ_NEST_BYTES / 2 + 14)   /* nest packed two 4-bit values per byte, padded to 16 */

#define HVQ_PIC_I 0
#define HVQ_PIC_P 1
#define HVQ_PIC_B 2

/* header flags */
#define HVQ_F_IS15        0x0001u   /* HVQM4 1.5 stream: per-plane half-sample rule (h4m:1337-1343) */
#define HVQ_F_LANDSCAPE   0x0002u   /* width >= height (h4m:965-975) */
#define HVQ_F_HAS_NEST    0x0004u   /* blob carries a nest: some block needs intra AOT */
#define HVQ_F_SELF_REF    0x0008u   /* P picture with a future-referencing (type 2) macroblock: the reference
                                       reads the picture being written (h4m:2060).  Its other macroblocks are
                                       reconstructed data-parallel into a side buffer, then hvq_selfref_kernel walks the
                                       macroblocks in raster order like the reference does */
#define HVQ_F_BIG_AOT     0x0010u   /* some block has more than 15 bases (I-luma type byte > 15) */
#define HVQ_F_CLAMPED     0x0020u   /* malformed input: a value was clamped to keep the device in bounds */
#define HVQ_F_CAPPED      0x0040u   /* an overflow-symbol loop (h4m:654-677: the reference sums for as long as the stream says) ended
                                       on this back end's cap instead of on the stream: the value differs from the reference's --
                                       the picture is refused, never decoded differently */

typedef struct HvqPicHeader {
    uint32_t magic;
    uint32_t total_bytes;
    uint16_t width, height;        /* luma samples */
    uint8_t  pic_kind;             /* HVQ_PIC_* */
    uint8_t  unk_shift;            /* h4m:1974 / 2022: accumulator scale of the AOT */
    uint8_t  dc_shift;             /* informational; already folded into pool values */
    uint8_t  wshift, hshift;       /* chroma subsampling shifts (1,1 for 4:2:0) */
    uint8_t  pad0[3];
    uint32_t flags;
    uint16_t hb[3], vb[3];         /* 4x4 blocks per plane */
    uint32_t plane_off[3];         /* byte offset of each plane inside a picture buffer */
    uint32_t pic_bytes;            /* Y|U|V size */
    uint32_t map_off[3];
    uint32_t mv_off;               /* 0 for I pictures */
    uint32_t wave_base_off;
    uint32_t pool_off;
    uint32_t pool_dwords;
    uint32_t nest_off;             /* 0 when absent */
    uint32_t tile_first[4];        /* first tile index of plane 0,1,2 and the total */
    uint32_t mcb_w, mcb_h;
    uint16_t max_items;            /* most queued (AOT) blocks in any tile: sizes the kernel's LDS accumulators */
    uint16_t pad1;
    uint32_t max_pairs;            /* most (block, basis) pairs in any tile */
    uint32_t reserved[3];
} HvqPicHeader;

#if defined(__cplusplus)
static_assert(sizeof(HvqPicHeader) == 128, "HvqPicHeader must be 128 bytes");
#else
_Static_assert(sizeof(HvqPicHeader) == 128, "HvqPicHeader must be 128 bytes");
#endif

/* AOT basis dword.  bits 12:0 = reference word bits 12:0 (offsets + strides, h4m:683-690),
 * bit 13 = negate (word bit 15), bits 31:14 = running coefficient sum + word offset bits 14:13
 * (the `*sum + offset` of h4m:726-731; <= 255*1020+3 < 2^18). */
#define HVQ_BASIS(word, sum_plus_off) \
    ((uint32_t)((word) & 0x1FFFu) | ((uint32_t)(((word) >> 15) & 1u) << 13) | ((uint32_t)(sum_plus_off) << 14))

/* Payload length in dwords of one block, from its map type byte.
 *   intra context (I picture, or P/B macroblock type 0; h4m:1433-1455, 1789-1827):
 *       kind 0 / 8 -> none; 6 -> literal; else -> `kind` bases
 *   inter context (h4m:1862-1910): proc = 1 -> none; kind 0 -> none; 6 -> literal;
 *       else -> 2 parameters + (kind-1) bases
 * `is_I_luma` selects the full type byte as kind (h4m:1093) instead of the low nibble. */
#if defined(__HIPCC__)
#define HVQ_HD __host__ __device__
#else
#define HVQ_HD
#endif
HVQ_HD static inline uint32_t hvq_payload_dwords(uint32_t type, int is_pb, int is_I_luma)
{
    /* written with selects only: the kernel evaluates it per lane */
    const uint32_t kind = is_I_luma ? type : (type & 0xFu);
    const int inter = is_pb && (type & 0x60u);
    const uint32_t n_intra = (kind == 0u || kind == 8u) ? 0u : kind;
    const uint32_t n_inter = ((type & 0x10u) || kind == 0u) ? 0u : kind + 1u;
    return kind == 6u ? ((inter && (type & 0x10u)) ? 0u : 4u) : (inter ? n_inter : n_intra);
}

/* one reconstruction job = one picture of one stream (device-visible).  Every member is a dword or a qword: the kernel reads
 * the record with scalar loads only (16-bit members would be fetched with vector loads), and because all tiles of a picture
 * read the SAME record it stays in the scalar cache / L2 -- unlike a per-tile record, whose first touch is an HBM miss
 * (~1.2-1.9k cycles, profiles/r02d). */
typedef struct HvqPlaneRec {       /* per-plane part (32 bytes) */
    uint64_t map;                  /* device address of the plane's map (entry [-1][-1], i.e. incl. border) */
    uint64_t dst;                  /* device address of the plane inside the destination picture */
    uint32_t plane_off;            /* byte offset of the plane inside a picture buffer (reference reads) */
    uint32_t tile_first;           /* first tile index of the plane */
    uint32_t hbvb;                 /* 4x4 blocks per row | rows << 16 */
    uint32_t pw_sub;               /* samples per row | ws << 16 | hs << 24 (subsampling shifts relative to luma) */
} HvqPlaneRec;

typedef struct HvqJob {                /* 232 bytes */
    uint64_t ring;                 /* the stream's picture slots: every reference read is ring + 32-bit offset (one SGPR base) */
    uint32_t ref0_off;             /* "past"   (macroblock type 1): byte offset of its slot inside the ring */
    uint32_t ref1_off;             /* "future" (macroblock type 2) */
    uint64_t pool;                 /* device addresses of the blob sections */
    uint64_t mv;
    uint64_t tq;                   /* tile queues of the picture (HvqTileQ[total_tiles]), built by hvq_tileq_kernel */
    uint64_t nest;                 /* nibble-packed nest (HVQ_NESTP_BYTES), 0 when absent */
    uint32_t slot_bytes;           /* readable bytes of a slot (>= pic_bytes + 8) */
    uint32_t flags;                /* HVQ_F_* | picture kind << 16 | unk_shift << 20 */
    uint32_t width;                /* luma samples per row */
    uint32_t mcb_w;
    uint32_t pool_dwords;          /* payload pool size */
    uint32_t total_tiles;          /* 0: picture dropped by the flush, its workgroups exit */
    uint32_t q_lits_off;           /* byte offsets from `tq` of the picture's literal, item and pair lists */
    uint32_t q_items_off;
    HvqPlaneRec plane[3];
    uint32_t q_pairs_off;
    uint32_t q_caps;               /* list entries reserved per tile: items | pairs << 16 (literals: HVQ_TILE_BLOCKS) */
    uint64_t wave_base;            /* blob section: pool offset of every run of 64 blocks (read by hvq_tileq_kernel only) */
    uint32_t q_recs_off;           /* byte offset from `tq` of the picture's block records */
    uint32_t q_offs_off;           /* HVQ_F_SELF_REF pictures: byte offset from `tq` of the blocks' pool offsets (u32 per block), else 0 */
    uint32_t pad2[2];
    uint32_t hb_magic[3];          /* per plane, hb = blocks per row: floor(2^32 / hb) + 1 (0 for hb = 1: the quotient is the index itself)
                                      -- mulhi(b, magic) is floor(b / hb) or one more for b < 2^22 (scalar split of a wave's first block) */
    uint32_t hb_magic16[3];        /* ceil(2^16 / hb): (t * magic16) >> 16 = floor(t / hb) exactly for t < 128 + hb, hb < 64 */
} HvqJob;
#define HVQ_JOB_KIND_SHIFT  16
#define HVQ_JOB_UNK_SHIFT   20

#if defined(__cplusplus)
static_assert(sizeof(HvqPlaneRec) == 32, "HvqPlaneRec must be 32 bytes");
static_assert(sizeof(HvqJob) == 232, "HvqJob must be 232 bytes");
#else
_Static_assert(sizeof(HvqPlaneRec) == 32, "HvqPlaneRec must be 32 bytes");
_Static_assert(sizeof(HvqJob) == 232, "HvqJob must be 232 bytes");
#endif

/*
 * Tile queues (round 3).  Which blocks of a tile need the AOT machinery, in which order, with which bases -- all of it
 * follows from the type bytes alone, so it is worked out ONCE per picture when its descriptors arrive (hvq_tileq_kernel,
 * part of the parse stage) instead of by every reconstruction launch: the reconstruction kernel no longer classifies,
 * ballots, scans or builds lists, and its AOT phase does not wait for its per-block phase.
 *   HvqTileQ   per tile: counts
 *   records    8 bytes per block, HVQ_TILE_BLOCKS per tile: what the owning lane does with the block, with its operands
 *              resolved -- the reconstruction kernel reads neither the map nor the vectors:
 *              w1 = DC value | HVQ_BR_* action << 8 | half-sample flags (hx << 10, hy << 11)
 *              w0 = motion compensated (also the MC part of an MC-residual block): ring offset of the top-left source
 *                   sample (h4m:1327-1355, clamped into the slot); weighted DC: the four neighbour values the predictor
 *                   sees, top | bottom << 8 | left << 16 | right << 24 (h4m:1437-1454, 1811-1814)
 *   literals   u32 per literal block: owner (lane of the tile) | pool offset << 8
 *   items      8 bytes per queued block (intra AOT first, then MC residual): owner | map entry << 8 | HVQ_IQ_WIDE, and the two
 *              scalars of an MC-residual block (h4m:1405-1406) as 16-bit values, the first before its shift by unk_shift;
 *              HVQ_IQ_WIDE (scalars beyond 16 bits, and every item of a serial tile): the pool offset of the payload instead
 *   pairs      8 bytes per (item, basis), items in order, fully decoded (h4m:683-731 / 738-772):
 *              w0 = [17:0] coefficient sum + offset, [18] negate, [19] sample stride 2, [20] row stride 2, [21] MC residual,
 *                   [31:23] item of the tile
 *              w1 = intra: nest index of sample (0,0) (nibble units); MC residual: ring offset of sample (0,0) of the
 *                   70x38 window (origin of h4m:1865-1868 + basis offset, clamped into the slot)
 */
typedef struct HvqTileQ {
    uint32_t w0;                   /* pairs | items << 16 | HVQ_TQ_* */
    uint32_t w1;                   /* literal blocks */
} HvqTileQ;
#define HVQ_TQ_INTRA   (1u << 26)  /* the tile has intra AOT items: the nest is staged */
#define HVQ_TQ_SERIAL  (1u << 27)  /* more pairs than the picture's list reserves: no pair list, items loop over their bases */
#define HVQ_PAIR_CAP_MAX 1024u     /* pairs per tile a list may reserve (the queue build keeps two tiles' lists in LDS) */
#define HVQ_BR_LIST    0u          /* literal or AOT block: the lists do it */
#define HVQ_BR_FLAT    1u          /* flat DC (h4m:281-286) */
#define HVQ_BR_WDC     2u          /* weighted DC (h4m:299-383) */
#define HVQ_BR_MC      3u          /* motion compensated */
#define HVQ_PQ_NEG     (1u << 18)
#define HVQ_PQ_X2      (1u << 19)
#define HVQ_PQ_Y2      (1u << 20)
#define HVQ_PQ_MC      (1u << 21)
#define HVQ_PQ_ITEM_SHIFT 23
#define HVQ_IQ_WIDE    (1u << 24)

/* Block classification by map type byte, one dword per (context, type): context 0 = I-picture luma (kind = whole byte,
 * h4m:1093), 1 = I-picture chroma, 2 = P/B picture.  Filled by hvq_type_class() on the host, read by the kernel. */
#define HVQ_TC_NPAY(c)   ((c) & 0xFFu)           /* payload dwords (hvq_payload_dwords) */
#define HVQ_TC_NB(c)     (((c) >> 8) & 0xFFu)    /* AOT bases */
#define HVQ_TC_CLS(c)    (((c) >> 16) & 3u)      /* 0 done by the owning lane, 1 intra AOT, 2 motion compensation + AOT residual */
#define HVQ_TC_MC        (1u << 18)              /* motion compensated (plain, or the MC part of class 2) */
#define HVQ_TC_WDC       (1u << 19)              /* weighted-DC intra predictor (intra kind 0) */
#define HVQ_TC_LIT       (1u << 20)              /* literal block */
HVQ_HD static inline uint32_t hvq_type_class(uint32_t type, int ctx)
{
    const int is_pb = ctx == 2, il = ctx == 0;
    const uint32_t kind = il ? type : (type & 0xFu);
    const int inter = is_pb && (type & 0x60u);
    const int proc = (type & 0x10u) != 0;
    const uint32_t npay = hvq_payload_dwords(type, is_pb, il);
    const int aot = kind != 0u && kind != 6u;
    const int c1 = !inter && aot && kind != 8u;
    const int c2 = inter && !proc && aot;
    const uint32_t nb = c1 ? kind : c2 ? kind - 1u : 0u;
    uint32_t c = (npay & 0xFFu) | ((nb & 0xFFu) << 8) | ((uint32_t)(c1 ? 1 : c2 ? 2 : 0) << 16);
    if (inter && (c2 || proc || kind == 0u)) c |= HVQ_TC_MC;
    if (!inter && kind == 0u) c |= HVQ_TC_WDC;
    if (kind == 6u && npay) c |= HVQ_TC_LIT;
    return c;
}

/* launch table: one entry per picture of a launch.  The grid is (picture slots, tiles): consecutive workgroup ids differ in
 * the picture, and with the slot count a multiple of 8 all tiles of a picture run on one XCD (workgroups are dealt round-robin
 * over the 8 XCDs), keeping its map, nest and reference reads in that XCD's L2. */
typedef struct HvqTileRef {
    uint32_t job;                  /* 0xFFFFFFFF: padding slot */
    uint32_t tile;                 /* tiles of the picture */
} HvqTileRef;


#endif
