/*
 * hvq_kernels.hip -- HVQM4 picture reconstruction for CDNA4 / gfx950 (MI355X).
 *
 * One launch reconstructs a BATCH of pictures (one job per picture, any mix of streams,
 * sizes and picture kinds).  One 256-thread workgroup = one tile = 256 consecutive 4x4
 * blocks of one plane; one lane = one 4x4 block, held as four packed dwords (4 samples per
 * dword).  Because intra prediction reads neighbour DC values from the descriptor map, not
 * neighbour pixels, every block of a picture is independent:
 * no intra-picture wavefront dependency exists and the whole batch is data-parallel.
 *
 * Integer/byte work, HBM-bound by design: no MFMA.  What matters here:
 *   - stores: lane i writes dword i of a 256-byte row segment -> every store instruction of a
 *     wave covers whole contiguous segments of the destination plane;
 *   - payload lookup: a block's payload length is a function of its type byte, so one
 *     workgroup prefix scan replaces per-block offsets (no offset traffic);
 *   - the 70x38 intra nest (2660 B) is staged once per workgroup in LDS and gathered from
 *     there (16 byte-gathers per basis);
 *   - sample arithmetic is SIMD-within-register: v_lerp_u8 for the 2-tap half-sample
 *     filters, 16-bit packed math for the weighted-DC predictor, v_sad_u8 for block sums;
 *   - reference pictures are addressed LINEARLY inside the Y|U|V buffer exactly like the
 *     reference's pointer arithmetic; every address is clamped to the
 *     picture slot so malformed vectors cannot fault the GPU.
 *
 * Reference behaviour restated per device function (h4m: = h4m_audio_decode.c).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvq_desc.h"

typedef uint32_t u32;
typedef int32_t i32;
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define HVQ_WG 256

/* divTable of h4m:265-273: 0x1000 / (i*16) * 16 */
__device__ __constant__ uint16_t k_div16[16] = { 0, 4096, 2048, 1360, 1024, 816, 672, 576, 512, 448, 400, 368, 336, 304, 288, 272 };

struct Blk { u32 r[4]; };

__device__ __forceinline__ u32 sat_pack(s16x2 a, s16x2 b)
{
    /* h4m:288-296 on pre-biased sums: (s+4)/8 as unsigned then clamp -> negatives become 255 */
    const u16x2 k255 = { 255, 255 };
    u16x2 ua = __builtin_elementwise_min(__builtin_bit_cast(u16x2, (s16x2)(a >> 3)), k255);
    u16x2 ub = __builtin_elementwise_min(__builtin_bit_cast(u16x2, (s16x2)(b >> 3)), k255);
    return __builtin_amdgcn_perm(__builtin_bit_cast(u32, ub), __builtin_bit_cast(u32, ua), 0x06040200u);
}

/* Weighted-DC intra block (h4m:299-383): out = sat_mean8(8V + r[y] + c[x]),
 * r[y] = a[y](T-V) + a[3-y](B-V), c[x] = a[x](L-V) + a[3-x](R-V), a = {2,0,-1,-1}. */
__device__ __forceinline__ Blk weight_block(int V, int T, int B, int L, int R)
{
    int dT = T - V, dB = B - V, dL = L - V, dR = R - V;
    int base = 8 * V + 4;
    s16x2 c01 = { (short)(base + 2 * dL - dR), (short)(base - dR) };
    s16x2 c23 = { (short)(base - dL), (short)(base - dL + 2 * dR) };
    int rr[4] = { 2 * dT - dB, -dB, -dT, 2 * dB - dT };
    Blk o;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        s16x2 r2 = { (short)rr[y], (short)rr[y] };
        o.r[y] = sat_pack(c01 + r2, c23 + r2);
    }
    return o;
}

__device__ __forceinline__ i32 clampi(i32 v, i32 lo, i32 hi) { return min(max(v, lo), hi); }

/* 4x4 motion-compensated block (h4m:1242-1294).  `a` = linear byte offset of the top-left
 * source sample inside the reference picture buffer. */
__device__ __forceinline__ Blk mc_block(const uint8_t *ref, i32 a, i32 stride, int hx, int hy, i32 amax8)
{
    uint64_t q[5];
#pragma unroll
    for (int y = 0; y < 5; ++y) q[y] = *(const u64u *)(ref + clampi(a + y * stride, 0, amax8));
    Blk o;
    if (!hy) {
        if (!hx) {
#pragma unroll
            for (int y = 0; y < 4; ++y) o.r[y] = (u32)q[y];
        } else {
#pragma unroll
            for (int y = 0; y < 4; ++y) o.r[y] = __builtin_amdgcn_lerp((u32)q[y], (u32)(q[y] >> 8), 0x01010101u);
        }
    } else if (!hx) {
#pragma unroll
        for (int y = 0; y < 4; ++y) o.r[y] = __builtin_amdgcn_lerp((u32)q[y], (u32)q[y + 1], 0x01010101u);
    } else {
        /* (a+b+c+d+2)>>2 exactly: two 16-bit lanes per dword, even and odd bytes */
        const u32 M = 0x00FF00FFu;
        u32 he[5], ho[5];
#pragma unroll
        for (int y = 0; y < 5; ++y) {
            u32 p = (u32)q[y], n = (u32)(q[y] >> 8);
            he[y] = (p & M) + (n & M);
            ho[y] = ((p >> 8) & M) + ((n >> 8) & M);
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            u32 e = ((he[y] + he[y + 1] + 0x00020002u) >> 2) & M;
            u32 d = ((ho[y] + ho[y + 1] + 0x00020002u) >> 2) & M;
            o.r[y] = e | (d << 8);
        }
    }
    return o;
}

/* exact floor(num / den) for num <= 4096, den <= 511 (0 -> 0): v_rcp_f32 estimate, integer fix-up.
 * Replaces the reference's divTable / mcdivTable lookups (h4m:265-273) -- no memory access. */
__device__ __forceinline__ u32 udiv_small(u32 num, u32 den)
{
    u32 q = (u32)((float)num * __builtin_amdgcn_rcpf((float)den));
    i32 r = (i32)(num - __umul24(q, den));
    if (r < 0) q -= 1;
    else if ((u32)r >= den) q += 1;
    return den ? q : 0u;
}

/* TOOLCHAIN HAZARD (ROCm 7.2 hipcc, gfx950): `clamp(x >> s, 0, 255)` pairs are pattern-matched into
 * v_ashr_pk_u8_i32 and OR-ed with the other two samples as if the instruction cleared the upper 16 bits
 * of its destination; on MI355X it leaves them as they were, which corrupted samples 2 and 3 of a row
 * (found by the parity suite).  The empty asm hides the shift result from that combine. */
__device__ __forceinline__ i32 sar(u32 v, i32 s)
{
    i32 r = (i32)v >> s;
    __asm__ volatile("" : "+v"(r));
    return r;
}

__device__ __forceinline__ u32 pack4(i32 a, i32 b, i32 c, i32 d)
{
    a = clampi(a, 0, 255); b = clampi(b, 0, 255); c = clampi(c, 0, 255); d = clampi(d, 0, 255);
    return (u32)a | ((u32)b << 8) | ((u32)c << 16) | ((u32)d << 24);
}

/*
 * One AOT basis into the 16 accumulators (h4m:679-732 / 734-773, 775-817).
 *   reference: factor = (sum + off) * (+-divTable[max-min]);  acc[i] += factor * e[i]   (uint32 wrap)
 * divTable[r] = 16 * (256 / r), so factor = 16 * s * q.  With <= 15 bases per block s < 2^14, q <= 256:
 * g = +-s*q fits 24 bits and sum_k g_k * e_ki < 2^31 never wraps, so the MACs run as full-rate
 * v_mad_i32_i24 and the wrap-exact value is (sum << 4).  BIG (I-luma type byte > 15, never produced by
 * real encoders) keeps the generic 32-bit wrap arithmetic.
 */
template <bool BIG>
__device__ __forceinline__ void aot_mac(u32 d, const u32 e[16], u32 lo, u32 hi, i32 acc[16])
{
    const u32 q = udiv_small(256u, (hi - lo) & 15u);
    const u32 s = d >> 14;
    if (!BIG) {
        i32 g = (i32)__umul24(s, q);
        if (d & 0x2000u) g = -g;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __mul24(g, (i32)e[i]) + acc[i];
    } else {
        u32 f = s * (q << 4);
        if (d & 0x2000u) f = 0u - f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (i32)((u32)acc[i] + f * e[i]);
    }
}

/* intra AOT block (h4m:1358-1377): nest gathers from LDS */
template <bool BIG>
__device__ __forceinline__ Blk intra_aot(const u32 *__restrict__ pay, u32 n, bool landscape, const uint8_t *s_nest,
                                         i32 V, i32 unk)
{
    i32 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    const i32 stride = landscape ? 70 : 38;
    u32 d = pay[0];
    for (u32 k = 0; k < n; ++k) {
        const u32 dn = pay[min(k + 1, n - 1)];                     /* next basis in flight while this one computes */
        i32 ol = d & 0x3F, os = (d >> 6) & 0x1F;
        u32 sl = (d >> 11) & 1, ss = (d >> 12) & 1;
        i32 o, xs, ys;
        if (landscape) { o = stride * os + ol; xs = 1 << sl; ys = stride << ss; }
        else           { o = stride * ol + os; xs = 1 << ss; ys = stride << sl; }
        u32 e[16], lo = 255, hi = 0;
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                u32 v = s_nest[o + y * ys + x * xs];
                e[4 * y + x] = v;
                lo = min(lo, v);
                hi = max(hi, v);
            }
        aot_mac<BIG>(d, e, lo, hi, acc);
        d = dn;
    }
    u32 r[16], total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = BIG ? (u32)acc[i] : ((u32)acc[i] << 4); total += r[i]; }
    const u32 delta = ((u32)V << unk) - (u32)((i32)total >> 4);
    Blk o4;
#pragma unroll
    for (int y = 0; y < 4; ++y)
        o4.r[y] = pack4(sar(r[4 * y] + delta, unk), sar(r[4 * y + 1] + delta, unk),
                        sar(r[4 * y + 2] + delta, unk), sar(r[4 * y + 3] + delta, unk));
    return o4;
}

/* MC + AOT residual block (h4m:1379-1420).  The nest is a 70x38 window of the reference LUMA plane
 * (h4m:1865-1868); each basis row (4 samples at stride 1 or 2) is one unaligned 8-byte load. */
__device__ __forceinline__ Blk predi_aot(const u32 *__restrict__ pay, u32 nb, bool landscape, const uint8_t *ref,
                                         i32 origin, i32 lw, i32 slot, Blk m, i32 unk)
{
    i32 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    u32 d = nb ? pay[2] : 0u;
    for (u32 k = 0; k < nb; ++k) {
        const u32 dn = pay[2 + min(k + 1, nb - 1)];
        i32 ol = d & 0x3F, os = (d >> 6) & 0x1F;
        u32 sl = (d >> 11) & 1, ss = (d >> 12) & 1;
        i32 o, ys; u32 x2;
        if (landscape) { o = lw * os + ol; x2 = sl; ys = lw << ss; }
        else           { o = lw * ol + os; x2 = ss; ys = lw << sl; }
        const u32 sel = x2 ? 0x06040200u : 0x03020100u;             /* stride 2: bytes 0,2,4,6 */
        u32 e[16], lo = 255, hi = 0;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            uint64_t q = *(const u64u *)(ref + clampi(origin + o + y * ys, 0, slot - 8));
            u32 w = __builtin_amdgcn_perm((u32)(q >> 32), (u32)q, sel);
            w = (w >> 4) & 0x0F0F0F0Fu;                              /* upper nibble of each sample, h4m:756-761 */
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                u32 v = (w >> (8 * x)) & 0xFFu;
                e[4 * y + x] = v;
                lo = min(lo, v);
                hi = max(hi, v);
            }
        }
        aot_mac<false>(d, e, lo, hi, acc);
        d = dn;
    }
    u32 total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) total += (u32)acc[i] << 4;
    const u32 mean_aot = (u32)((i32)total >> 4);
    u32 sum = 8;
#pragma unroll
    for (int y = 0; y < 4; ++y) sum = __builtin_amdgcn_sad_u8(m.r[y], 0u, sum);
    const i32 mean = (i32)(sum >> 4);
    i32 px[16];
    u32 lo = 255, hi = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            u32 v = (m.r[y] >> (8 * x)) & 0xFFu;
            px[4 * y + x] = (i32)v;
            lo = min(lo, v);
            hi = max(hi, v);
        }
    const u32 addend = pay[0] - mean_aot;
    const i32 gain = (i32)pay[1];
    const u32 mcd = udiv_small(0x1000u, hi - lo);                    /* mcdivTable[max-min], h4m:272, 1407 */
    const u32 factor = (u32)gain * mcd;
    const bool small = gain > -2048 && gain < 2048;                  /* |factor| < 2^23: 24-bit multiply is exact */
    Blk o4;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        i32 v[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int i = 4 * y + x;
            u32 t = small ? (u32)__mul24(px[i] - mean, (i32)factor) : (u32)(px[i] - mean) * factor;
            u32 r = ((u32)acc[i] << 4) + addend + t;
            v[x] = sar(r, unk) + px[i];
        }
        o4.r[y] = pack4(v[0], v[1], v[2], v[3]);
    }
    return o4;
}

/* wave64 inclusive prefix sum on the DPP network: Kogge-Stone inside each row of 16 lanes
 * (row_shr 1,2,4,8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3. */
__device__ __forceinline__ u32 wave_incl_scan(u32 v)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

__device__ __forceinline__ u32 lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

__device__ __forceinline__ void block_coords(u32 b, i32 hb, float rhb, i32 &bx, i32 &by)
{
    i32 q = (i32)((float)b * rhb);                 /* b < 2^22: estimate within +-1, fixed below */
    i32 r = (i32)b - q * hb;
    if (r < 0) { q -= 1; r += hb; }
    else if (r >= hb) { q += 1; r -= hb; }
    by = q; bx = r;
}

#define HVQ_NW (HVQ_WG / 64)

/*
 * Workgroup = tile of 256 consecutive blocks of one plane.
 *   phase A  every lane owns one block: descriptors are fetched with independent loads (own map
 *            entry, four neighbours, macroblock vector), cheap kinds (flat, weighted-DC, literal,
 *            plain MC) are reconstructed at once into the LDS tile; AOT blocks are queued.
 *   phase B  the queue (intra-AOT items first, then MC-residual items; entries carry everything the
 *            owner already fetched) is re-dealt one item per lane, so the expensive divergent paths
 *            run on densely packed wavefronts.
 *   phase C  the finished tile leaves LDS as 16-byte row segments: one store instruction of a wave
 *            writes four complete 256-byte runs of the destination plane (full cache lines, written
 *            once -- phase B results never reach HBM as partial lines).
 */
__global__ __launch_bounds__(HVQ_WG)
void hvq_recon_kernel(const HvqJob *__restrict__ jobs, const HvqTileRef *__restrict__ tiles)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_nest[2672];
    __shared__ __attribute__((aligned(16))) u32 s_out[4][HVQ_WG];   /* [sample row][block] packed dwords */
    __shared__ u32 s_item0[HVQ_WG];    /* owner lane | payload offset << 10 */
    __shared__ u32 s_item1[HVQ_WG];    /* map entry {value, type} */
    __shared__ u32 s_item2[HVQ_WG];    /* macroblock vector */
    __shared__ u32 s_cnt[HVQ_NW][2];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const u32 job_id = __builtin_amdgcn_readfirstlane(tiles[blockIdx.x].job);
    const u32 tile = __builtin_amdgcn_readfirstlane(tiles[blockIdx.x].tile);
    if (job_id == 0xFFFFFFFFu) return;            /* padding entry of the XCD-dealt tile table (uniform exit) */
    const HvqJob *__restrict__ J = jobs + job_id;
    const uint8_t *__restrict__ blob = (const uint8_t *)J->blob;

    /* TOOLCHAIN HAZARD (ROCm 7.2 hipcc, gfx950): a wave-uniform but run-time index into these small
     * arrays was lowered to s_load_dword with base (J + 2p) and soffset 2p; for p = 1 neither part is
     * dword aligned and the scalar memory unit truncates them separately -> element 0 is read.  All
     * per-plane fields are therefore fetched with constant indices and selected. */
#define PSEL(a) (p == 0 ? (a)[0] : p == 1 ? (a)[1] : (a)[2])
    const u32 tf1 = J->tile_first[1], tf2 = J->tile_first[2];
    const int p = (tile >= tf1) + (tile >= tf2);
    const u32 tf = p == 0 ? 0u : p == 1 ? tf1 : tf2;
    const i32 hb = PSEL(J->hb), vb = PSEL(J->vb);
    const float rhb = 1.0f / (float)hb;
    const u32 nblocks = (u32)hb * (u32)vb;
    const u32 b0 = (tile - tf) * HVQ_TILE_BLOCKS;
    const i32 ws = p ? J->wshift : 0, hs = p ? J->hshift : 0;
    const i32 pw = J->width >> ws;
    const u32 flags = J->flags;
    const bool is_pb = J->pic_kind != HVQ_PIC_I;
    const bool I_luma = !is_pb && p == 0;
    const i32 unk = J->unk_shift;
    const i32 mstride = hb + 2;
    const bool landscape = flags & HVQ_F_LANDSCAPE;
    const bool is15 = flags & HVQ_F_IS15;
    const uint8_t *map = blob + PSEL(J->map_off);
    const u32 *__restrict__ pool = (const u32 *)(blob + J->pool_off);
    const u32 *__restrict__ mvs = (const u32 *)(blob + J->mv_off);
    const i32 plane_off = (i32)PSEL(J->plane_off);
    uint8_t *plane = (uint8_t *)J->dst + plane_off;
    const i32 slot = (i32)J->slot_bytes;
    const i32 mcb_w = (i32)J->mcb_w;
#undef PSEL

    /* ---- phase A: own block ---- */
    const u32 b = b0 + (u32)tid;
    const bool valid = b < nblocks;
    i32 bx, by;
    block_coords(valid ? b : 0u, hb, rhb, bx, by);
    const uint8_t *ent = map + 2 * ((by + 1) * mstride + bx + 1);
    /* independent loads first: own entry, four neighbours, vector, wave payload base */
    const u32 e16 = *(const uint16_t *)ent;
    const u32 nt = *(const uint16_t *)(ent - 2 * mstride), nbt = *(const uint16_t *)(ent + 2 * mstride);
    const u32 nl = *(const uint16_t *)(ent - 2), nr = *(const uint16_t *)(ent + 2);
    u32 mvw = 0;
    if (is_pb) mvw = mvs[(by >> (1 - hs)) * mcb_w + (bx >> (1 - ws))];
    const u32 wbase = ((const u32 *)(blob + J->wave_base_off))[tile * HVQ_NW + (u32)wave];

    const i32 V = e16 & 0xFF;
    const u32 T = valid ? (e16 >> 8) : 0u;
    const bool inter = is_pb && (T & 0x60u);
    const u32 kind = I_luma ? T : (T & 0xFu);
    const u32 npay = valid ? hvq_payload_dwords(T, is_pb, I_luma) : 0u;
    /* class: 0 cheap (done in place), 1 intra AOT, 2 MC + AOT residual */
    int cls = 0;
    if (valid) {
        if (!inter) { if (kind != 0 && kind != 8 && kind != 6) cls = 1; }
        else if (!(T & 0x10u) && kind != 0 && kind != 6) cls = 2;
    }
    const u32 off = wbase + wave_incl_scan(npay) - npay;
    const unsigned long long m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
    if (lane == 0) { s_cnt[wave][0] = (u32)__popcll(m1); s_cnt[wave][1] = (u32)__popcll(m2); }

    if (valid && cls == 0) {
        Blk o;
        const u32 *__restrict__ pay = pool + off;
        if (!inter) {
            if (kind == 0) {
                /* neighbour DCs via the map; the border {0x7F,0xFF} never exposes (h4m:1437-1442, 1811-1814).
                 * I pictures track the left value separately: only kinds 0 and 8 expose it (h4m:1443-1454). */
                i32 Tt = (nt & 0x7700u) ? V : (i32)(nt & 0xFF);
                i32 Bb = (nbt & 0x7700u) ? V : (i32)(nbt & 0xFF);
                i32 Rr = (nr & 0x7700u) ? V : (i32)(nr & 0xFF);
                bool lexp = is_pb ? !(nl & 0x7700u) : ((nl >> 8) == 0 || (nl >> 8) == 8);
                i32 Ll = lexp ? (i32)(nl & 0xFF) : V;
                o = weight_block(V, Tt, Bb, Ll, Rr);
            } else if (kind == 8) {
                u32 v = (u32)V * 0x01010101u;                                 /* h4m:281-286 */
                o.r[0] = o.r[1] = o.r[2] = o.r[3] = v;
            } else {
                o.r[0] = pay[0]; o.r[1] = pay[1]; o.r[2] = pay[2]; o.r[3] = pay[3];   /* literal, h4m:543-549 */
            }
        } else if (!(T & 0x10u) && kind == 6) {
            o.r[0] = pay[0]; o.r[1] = pay[1]; o.r[2] = pay[2]; o.r[3] = pay[3];
        } else {
            const i32 rx = (i32)(int16_t)(mvw & 0xFFFF), ry = (i32)(int16_t)(mvw >> 16);
            const uint8_t *ref = (const uint8_t *)((((T >> 5) & 3u) == 1u) ? J->ref0 : J->ref1);
            const i32 pdx = rx >> ws, pdy = ry >> hs;
            const int hx = is15 ? (pdx & 1) : (rx & 1), hy = is15 ? (pdy & 1) : (ry & 1);   /* h4m:1337-1343 */
            const i32 a = plane_off + (pdy >> 1) * pw + (pdx >> 1) + (by & (1 - hs)) * 4 * pw + (bx & (1 - ws)) * 4;
            o = mc_block(ref, a, pw, hx, hy, slot - 8);
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) s_out[y][tid] = o.r[y];
    }

    __syncthreads();                                                           /* barrier 1: queue counts */
    u32 nI = 0, nP = 0, myI = 0, myP = 0;
#pragma unroll
    for (int w = 0; w < HVQ_NW; ++w) {
        const u32 ci = s_cnt[w][0], cp = s_cnt[w][1];
        if (w < wave) { myI += ci; myP += cp; }
        nI += ci; nP += cp;
    }
    const u32 total = nI + nP;
    if (cls) {
        const u32 slotq = cls == 1 ? myI + lanes_below(m1) : nI + myP + lanes_below(m2);
        s_item0[slotq] = (u32)tid | (off << 10);
        s_item1[slotq] = e16;
        s_item2[slotq] = mvw;
    }
    if (nI) {
        const u32 *src = (const u32 *)(blob + J->nest_off);
        for (int i = tid; i < HVQ_NEST_BYTES / 4; i += HVQ_WG) ((u32 *)s_nest)[i] = src[i];
    }
    if (total) __syncthreads();                                                /* barrier 2: queue + nest staged */

    /* ---- phase B: queued blocks, one per lane ---- */
    if ((u32)tid < total) {
        const u32 item = s_item0[tid];
        const u32 owner = item & 1023u;
        const u32 *__restrict__ pay = pool + (item >> 10);
        const u32 q16 = s_item1[tid];
        const i32 QV = q16 & 0xFF;
        const u32 QT = q16 >> 8;
        Blk o;
        if ((u32)tid < nI) {
            const u32 qkind = I_luma ? QT : (QT & 0xFu);
            if (flags & HVQ_F_BIG_AOT) o = intra_aot<true>(pay, qkind, landscape, s_nest, QV, unk);
            else                       o = intra_aot<false>(pay, qkind, landscape, s_nest, QV, unk);
        } else {
            i32 qx, qy;
            block_coords(b0 + owner, hb, rhb, qx, qy);
            const u32 qmv = s_item2[tid];
            const i32 rx = (i32)(int16_t)(qmv & 0xFFFF), ry = (i32)(int16_t)(qmv >> 16);
            const uint8_t *ref = (const uint8_t *)((((QT >> 5) & 3u) == 1u) ? J->ref0 : J->ref1);
            const i32 pdx = rx >> ws, pdy = ry >> hs;
            const int hx = is15 ? (pdx & 1) : (rx & 1), hy = is15 ? (pdy & 1) : (ry & 1);
            const i32 a = plane_off + (pdy >> 1) * pw + (pdx >> 1) + (qy & (1 - hs)) * 4 * pw + (qx & (1 - ws)) * 4;
            const i32 lw = J->width;
            const i32 origin = landscape ? rx / 2 + (ry / 2 - 16) * lw - 32 : rx / 2 + (ry / 2 - 32) * lw - 16;
            Blk m = mc_block(ref, a, pw, hx, hy, slot - 8);
            o = predi_aot(pay, (QT & 0xFu) - 1u, landscape, ref, origin, lw, slot, m, unk);
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) s_out[y][owner] = o.r[y];
    }
    if (total) __syncthreads();                                                /* barrier 3: tile complete in LDS */

    /* ---- phase C: tile -> HBM ---- */
    if ((hb & 3) == 0) {
        /* lane (g, r): sample row r of blocks 4g..4g+3 = 16 contiguous bytes of the plane */
        const int g = wave * 16 + (lane & 15), r = lane >> 4;
        const u32 gb = b0 + 4u * (u32)g;
        if (gb < nblocks) {
            i32 gx, gy;
            block_coords(gb, hb, rhb, gx, gy);
            const uint4 v = *(const uint4 *)&s_out[r][4 * g];
            *(uint4 *)(plane + (size_t)(gy * 4 + r) * pw + gx * 4) = v;
        }
    } else if (valid) {
        uint8_t *dst = plane + (size_t)(by * 4) * pw + bx * 4;
#pragma unroll
        for (int y = 0; y < 4; ++y) *(u32 *)(dst + (size_t)y * pw) = s_out[y][tid];
    }
}

extern "C" hipError_t hvq_launch_recon(const HvqJob *jobs_dev, const HvqTileRef *tiles_dev, uint32_t ntiles, hipStream_t stream)
{
    if (ntiles == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_recon_kernel, dim3(ntiles), dim3(HVQ_WG), 0, stream, jobs_dev, tiles_dev);
    return hipGetLastError();
}
