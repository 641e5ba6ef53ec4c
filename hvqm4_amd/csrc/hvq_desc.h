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
    uint64_t tq;                   /* HVQ_F_SELF_REF pictures: side section the level's launch leaves the blocks' pool offsets in (q_offs_off), else 0 */
    uint64_t nest;                 /* nibble-packed nest (HVQ_NESTP_BYTES), 0 when absent */
    uint32_t slot_bytes;           /* readable bytes of a slot (>= pic_bytes + 8) */
    uint32_t flags;                /* HVQ_F_* | picture kind << 16 | unk_shift << 20 */
    uint32_t width;                /* luma samples per row */
    uint32_t mcb_w;
    uint32_t pool_dwords;          /* payload pool size */
    uint32_t total_tiles;          /* 0: picture dropped by the flush, its workgroups exit */
    uint32_t tile_first12[2];      /* first tile of planes 1 and 2 (plane 0 starts at tile 0): a copy beside the common part, so that a workgroup knows its plane
                                      before it has loaded a plane record (the kernel addresses this record by dword index: dwords 18, 19) */
    HvqPlaneRec plane[3];
    uint32_t rsv1[2];
    uint64_t wave_base;            /* blob section: pool offset of every run of 64 blocks (scalar loads of hvq_recon_inline_kernel) */
    uint32_t rsv2;
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

#define HVQ_PQ_NEG     (1u << 18)   /* a basis' gain word inside the kernel: coefficient sum + offset [17:0], negate */

/* launch table: one entry per picture of a launch.  The grid is (picture slots, tiles): consecutive workgroup ids differ in
 * the picture, and with the slot count a multiple of 8 all tiles of a picture run on one XCD (workgroups are dealt round-robin
 * over the 8 XCDs), keeping its map, nest and reference reads in that XCD's L2. */
typedef struct HvqTileRef {
    uint32_t job;                  /* 0xFFFFFFFF: padding slot */
    uint32_t tile;                 /* tiles of the picture */
} HvqTileRef;


#endif
