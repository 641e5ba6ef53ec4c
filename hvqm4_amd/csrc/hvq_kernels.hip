/*
 * hvq_kernels.hip -- HVQM4 picture reconstruction for CDNA4 / gfx950 (MI355X).
 *
 * One launch reconstructs a BATCH of pictures (one job per picture, any mix of streams,
 * sizes and picture kinds).  One 256-thread workgroup = one tile = 256 consecutive 4x4
 * blocks of one plane (raster order); a block is four packed dwords (4 samples per dword).  Because intra prediction reads neighbour DC values from the descriptor map, not
 * neighbour pixels, every block of a picture is independent:
 * no intra-picture wavefront dependency exists and the whole batch is data-parallel.
 *
 * Integer/byte work, no dense contraction: no MFMA.  What matters here (measurements in DESIGN.md section 5):
 *   - the tile is assembled in LDS and leaves as 16-byte row segments: every store instruction of a wave
 *     writes four complete 256-byte runs, every output line reaches HBM once and whole;
 *   - payload lookup: a block's payload length is a function of its type byte, so one 64-lane prefix scan
 *     (DPP) replaces per-block offsets (no offset traffic);
 *   - cheap block kinds are reconstructed by the lane that owns the block; AOT work is re-dealt so that one
 *     lane handles one (block, basis) pair -- the cumulative coefficient sum is resolved by the host, which
 *     makes bases independent -- and meets in LDS accumulators (ds_add);
 *   - the 70x38 intra nest is staged per workgroup in LDS, nibble-packed: one unaligned ds_read_b64 per
 *     basis row; MC-residual window rows are one unaligned 8-byte global load each;
 *   - sample arithmetic is SIMD-within-register: v_lerp_u8 for the 2-tap half-sample filters, 16-bit packed
 *     math for the weighted-DC predictor, v_sad_u8 for block sums, 24-bit multiplies for the AOT products;
 *     the reference's divTable / mcdivTable lookups are a v_rcp_f32 estimate with an exact integer fix-up;
 *   - reference pictures are addressed LINEARLY inside the Y|U|V buffer exactly like the reference's pointer
 *     arithmetic; every address is clamped to the picture slot so malformed vectors cannot
 *     fault the GPU.
 * The limiter on vector-heavy streams is the number of distinct cache lines a CU's texture addresser can
 * process (~0.43 per clock, tools/ubench/gather_rate.hip), not HBM bandwidth and not the ALUs.
 *
 * Reference behaviour restated per device function (h4m: = h4m_audio_decode.c).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvq_desc.h"

typedef uint32_t u32;
typedef int32_t i32;
typedef uint64_t __attribute__((aligned(1))) u64u;
/* Everything in HBM is addressed through global-address-space pointers.  A generic pointer becomes flat_load /
 * flat_store, which count on the LDS counter as well: every s_waitcnt for an LDS access would then also wait for all
 * outstanding HBM loads of the wave. */
#define GLB __attribute__((address_space(1)))
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define HVQ_WG HVQ_TILE_BLOCKS
#ifndef HVQ_MIN_WAVES
#define HVQ_MIN_WAVES 8               /* waves per SIMD the register allocation must allow (64 VGPRs) */
#endif

/* ablation builds only (tools/ablate.sh): 1 = no phase B, 2 = no MC loads, 3 = no stores, 4 = descriptors only,
 * 5 = no cheap-kind compute */
#ifndef HVQ_ABL
#define HVQ_ABL 0
#endif

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
struct McRows { uint64_t q[5]; };

__device__ __forceinline__ McRows mc_load(const GLB uint8_t *ref, i32 a, i32 stride, int hy, i32 amax8)
{
    McRows r;
#pragma unroll
    for (int y = 0; y < 4; ++y) r.q[y] = *(const GLB u64u *)(ref + clampi(a + y * stride, 0, amax8));
    r.q[4] = 0;
    if (hy) r.q[4] = *(const GLB u64u *)(ref + clampi(a + 4 * stride, 0, amax8));   /* 5th row only for vertical half samples */
    return r;
}

__device__ __forceinline__ Blk mc_filter(const McRows &rows, int hx, int hy)
{
    const uint64_t *q = rows.q;
    Blk o;
    if (hx & hy) {
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
    } else {
        /* copy, horizontal or vertical 2-tap: v_lerp_u8 of the row with itself is the identity, so the three cases
         * are one straight-line sequence with selected operands (no divergence) */
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const u32 p = (u32)q[y];
            const u32 other = hx ? (u32)(q[y] >> 8) : hy ? (u32)q[y + 1] : p;
            o.r[y] = __builtin_amdgcn_lerp(p, other, 0x01010101u);
        }
    }
    return o;
}

__device__ __forceinline__ Blk mc_block(const GLB uint8_t *ref, i32 a, i32 stride, int hx, int hy, i32 amax8)
{
    return mc_filter(mc_load(ref, a, stride, hy, amax8), hx, hy);
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
 * AOT arithmetic (h4m:679-817).  reference: factor = (sum + off) * (+-divTable[max-min]);
 * acc[i] += factor * e[i] (uint32 wrap).  divTable[r] = 16 * (256 / r), so factor = 16 * s * q.
 * With <= 15 bases per block s < 2^14 and q <= 256: g = +-s*q fits 24 bits and sum_k g_k * e_ki < 2^31
 * never wraps, so products are full-rate 24-bit multiplies and the wrap-exact value is (sum << 4).
 * BIG (I-luma type byte > 15, never produced by real encoders) keeps generic 32-bit wrap arithmetic.
 * The host stores the running coefficient sum in every basis dword, so bases are independent and are
 * processed one per lane; their products meet in LDS with ds_add_u32.
 */
__device__ __forceinline__ i32 basis_gain(u32 d, u32 lo, u32 hi, bool big)
{
    const u32 q = udiv_small(256u, (hi - lo) & 15u);
    const u32 s = d >> 14;
    i32 g = big ? (i32)(s * (q << 4)) : (i32)__umul24(s, q);
    return (d & 0x2000u) ? -g : g;
}

__device__ __forceinline__ void basis_scatter(i32 g, const u32 e[16], bool big, i32 *acc_lds, u32 stride)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const i32 t = big ? (i32)((u32)g * e[i]) : __mul24(g, (i32)e[i]);
        __hip_atomic_fetch_add(acc_lds + i * stride, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

/* nest gather for one basis, intra (h4m:713-725).  The LDS nest holds two 4-bit values per byte at the
 * reference's linear index (nibble n of the array = nest_data[n]); a basis row spans at most 7 values,
 * so one unaligned 8-byte LDS read per row replaces four byte reads. */
__device__ __forceinline__ void gather_nest(u32 d, bool landscape, const uint8_t *s_nest, u32 e[16], u32 &lo, u32 &hi)
{
    const i32 stride = landscape ? 70 : 38;
    i32 ol = d & 0x3F, os = (d >> 6) & 0x1F;
    u32 sl = (d >> 11) & 1, ss = (d >> 12) & 1;
    i32 o, ys; u32 x2;
    if (landscape) { o = stride * os + ol; x2 = sl; ys = stride << ss; }
    else           { o = stride * ol + os; x2 = ss; ys = stride << sl; }
    const u32 sh = x2 ? 8u : 4u;                                     /* bits between consecutive samples */
    lo = 255; hi = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const i32 n = o + y * ys;
        uint64_t q = *(const u64u *)(s_nest + (n >> 1));
        q >>= 4 * (n & 1);
        const u32 w0 = (u32)q;
        /* samples at bits 0, sh, 2sh, 3sh (sh = 4 or 8): all inside the low dword */
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            u32 v = (w0 >> (sh * x)) & 15u;
            e[4 * y + x] = v;
            lo = min(lo, v);
            hi = max(hi, v);
        }
    }
}

/* nest gather for one basis, MC residual: the nest is a 70x38 window of the reference LUMA plane
 * (h4m:1865-1868, 734-765); each basis row (4 samples at stride 1 or 2) is one unaligned 8-byte load */
__device__ __forceinline__ void gather_window(u32 d, bool landscape, const GLB uint8_t *ref, i32 origin, i32 lw, i32 slot,
                                              u32 e[16], u32 &lo, u32 &hi)
{
    i32 ol = d & 0x3F, os = (d >> 6) & 0x1F;
    u32 sl = (d >> 11) & 1, ss = (d >> 12) & 1;
    i32 o, ys; u32 x2;
    if (landscape) { o = lw * os + ol; x2 = sl; ys = lw << ss; }
    else           { o = lw * ol + os; x2 = ss; ys = lw << sl; }
    const u32 sel = x2 ? 0x06040200u : 0x03020100u;                 /* stride 2: bytes 0,2,4,6 */
    uint64_t q[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) q[y] = *(const GLB u64u *)(ref + clampi(origin + o + y * ys, 0, slot - 8));
    lo = 255; hi = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        u32 w = __builtin_amdgcn_perm((u32)(q[y] >> 32), (u32)q[y], sel);
        w = (w >> 4) & 0x0F0F0F0Fu;                                  /* upper nibble of each sample */
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            u32 v = (w >> (8 * x)) & 0xFFu;
            e[4 * y + x] = v;
            lo = min(lo, v);
            hi = max(hi, v);
        }
    }
}

/* intra AOT epilogue (h4m:1367-1376): r = wrap-exact accumulators */
__device__ __forceinline__ Blk intra_finish(const u32 r[16], i32 V, i32 unk)
{
    u32 total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) total += r[i];
    const u32 delta = ((u32)V << unk) - (u32)((i32)total >> 4);
    Blk o4;
#pragma unroll
    for (int y = 0; y < 4; ++y)
        o4.r[y] = pack4(sar(r[4 * y] + delta, unk), sar(r[4 * y + 1] + delta, unk),
                        sar(r[4 * y + 2] + delta, unk), sar(r[4 * y + 3] + delta, unk));
    return o4;
}

/* MC residual epilogue (h4m:1385-1419): m = motion-compensated block, p0/p1 = host-resolved scalars */
__device__ __forceinline__ Blk predi_finish(const u32 r[16], Blk m, u32 p0, u32 p1, i32 unk)
{
    u32 total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) total += r[i];
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
    const u32 addend = p0 - mean_aot;
    const i32 gain = (i32)p1;
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
            v[x] = sar(r[i] + addend + t, unk) + px[i];
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
    i32 q = (i32)((float)b * rhb);                 /* b < 2^22: estimate within +-1, fixed below without branches */
    i32 r = (i32)b - q * hb;
    const i32 lo = r < 0 ? 1 : 0, hi = r >= hb ? 1 : 0;
    by = q - lo + hi;
    bx = r + (lo - hi) * hb;
}

#define HVQ_NW (HVQ_WG / 64)

/* Diagnostic build only (-DHVQ_STAMPS, tools/variant.sh): s_memtime stamps of wave phases into a buffer of their own
 * (64 x u64 per workgroup: [wave][16]); the shipped kernel executes no stamp.  VM = also wait for the wave's
 * outstanding vector-memory operations first (exposes load latency at that point). */
#ifdef HVQ_STAMPS
static unsigned long long *g_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) void hvq_set_stamps(unsigned long long *p) { g_stamps = p; }
#define HVQ_STAMP_ARG , unsigned long long *stamps
#define STAMP(i, VM) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
        if (VM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
        if (stamps && (threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 64 + (threadIdx.x >> 6) * 16 + (i)] = t_; } while (0)
#else
#define HVQ_STAMP_ARG
#define STAMP(i, VM) do { } while (0)
#endif

/* ------------------------------------------------------------------------------------------------------
 * Tile records.  The host deals {job, tile} pairs into launch order; this kernel expands each into the
 * self-contained 128-byte record its workgroup reads with one scalar load (hvq_desc.h).  It runs once per
 * flush (after the entropy parse, whose wave_base[] only the device has for GPU-parsed pictures).
 */
__global__ __launch_bounds__(256)
void hvq_tilegen_kernel(const HvqJob *__restrict__ jobs, const HvqTileRef *__restrict__ tiles, HvqTileRec *__restrict__ recs, u32 n)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HvqTileRef ref = tiles[i];
    HvqTileRec r;
    __builtin_memset(&r, 0, sizeof r);
    if (ref.job != 0xFFFFFFFFu) {
        const HvqJob *J = jobs + ref.job;
        const u32 tile = ref.tile;
        const int p = (tile >= J->plane[1].tile_first) + (tile >= J->plane[2].tile_first);
        const HvqPlaneRec P = J->plane[p];
        const u32 *wb = (const u32 *)J->wave_base + (size_t)tile * HVQ_NW;
        const u32 w0 = wb[0];
        const u32 wend = tile + 1 < J->total_tiles ? wb[HVQ_NW] : J->pool_dwords;
        r.map = P.map; r.dst = P.dst;
        r.pool = J->pool + 4ull * w0;
        r.mv = J->mv; r.nest = J->nest; r.ref0 = J->ref0; r.ref1 = J->ref1;
        r.b0 = (tile - P.tile_first) * HVQ_TILE_BLOCKS;
        r.nblocks = (u32)P.hb * P.vb;
        r.pool_dwords = wend - w0;
        r.wrel[0] = wb[1] - w0; r.wrel[1] = wb[2] - w0; r.wrel[2] = wb[3] - w0;
        r.plane_off = P.plane_off; r.slot_bytes = J->slot_bytes;
        r.flags = (J->flags & 0xFFFFu) | ((u32)J->pic_kind << HVQ_TR_KIND_SHIFT) | ((u32)J->unk_shift << HVQ_TR_UNK_SHIFT) |
                  ((u32)p << HVQ_TR_PLANE_SHIFT);
        r.hb = P.hb; r.vb = P.vb; r.pw = P.pw; r.lw = J->width; r.mcb_w = (uint16_t)J->mcb_w;
        r.ws = P.ws; r.hs = P.hs;
        r.rhb = 1.0f / (float)P.hb;
    }
    recs[i] = r;
}

extern "C" hipError_t hvq_launch_tilegen(const HvqJob *jobs_dev, const HvqTileRef *tiles_dev, HvqTileRec *recs_dev, uint32_t n,
                                         hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_tilegen_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, jobs_dev, tiles_dev, recs_dev, n);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------------
 * Reconstruction.
 *
 * Workgroup = tile of 256 consecutive blocks of one plane; wave = 64 of them; lane = one block.
 * What bounds this kernel is the depth of its chain of dependent memory round trips (each ~1.7k cycles under load,
 * profiles/r02a_stamps.txt), so it is built as three levels with everything inside a level issued at once:
 *   level 0   one scalar load of the tile record;
 *   level 1   own map entry + four neighbours + macroblock vector per lane, and -- cooperatively -- the tile's whole
 *             payload (contiguous in the pool) and the nest into LDS;
 *   level 2   the gathers that depend on them: motion-compensation source rows, MC-residual window rows.
 * After the one barrier that publishes payload and nest, every wave works on its own 64 blocks without meeting the
 * others again:
 *   cheap kinds (flat, weighted DC, literal, plain MC) are reconstructed by the owning lane into the LDS tile;
 *   AOT blocks are queued PER WAVE, at most HVQ_CHUNK at a time, counting-sorted by (class, number of bases) with one
 *   returning LDS add per item; every (item, basis) pair becomes one lane's work (the serial basis loop of h4m:782-788
 *   turns into one parallel step), laid out basis-major -- pair (k, r) of the r-th item in sorted order sits at
 *   seg[k] + r -- so that the 64 lanes of one ds_add hit 64 different items = consecutive banks: no same-address
 *   serialisation and no bank conflict (item-major order made the 1..15 lanes of one item add to the same 16 words);
 *   the item's lane then turns the accumulators into samples (h4m:1367-1376 / 1385-1419);
 *   the wave's 64 blocks leave LDS as 16-byte row segments: one store instruction writes four complete 256-byte runs.
 * LDS traffic inside a wave needs no barrier: a wave's LDS instructions execute in order.
 */
#define HVQ_POOL_LDS 640            /* payload dwords of a tile staged in LDS; larger tiles read the pool from HBM */
#define HVQ_CHUNK    32             /* queued blocks accumulated at a time per wave */
#define HVQ_PAIRWIN  128            /* (item, basis) pairs listed at a time per wave */

struct WaveLds {
    i32 acc[16 * HVQ_CHUNK];        /* [sample][item slot] */
    u32 pair[HVQ_PAIRWIN];          /* item slot | payload index of the basis << 5 */
    u32 hist[32];                   /* items per (class, bases) bin; then pair segment starts per (class, basis k) */
    u32 item[HVQ_CHUNK];            /* owner lane | map entry {value, type} << 8 */
    u32 imv[HVQ_CHUNK];             /* the owner's macroblock vector */
    u32 ioff[HVQ_CHUNK];            /* the owner's payload index */
};

/* compiler-level ordering of LDS accesses between lanes of ONE wave (the hardware executes them in order) */
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__global__ __launch_bounds__(HVQ_WG, HVQ_MIN_WAVES)
void hvq_recon_kernel(const HvqTileRec *__restrict__ recs HVQ_STAMP_ARG)
{
    __shared__ __attribute__((aligned(16))) u32 s_out[4][HVQ_WG];                  /* [sample row][block] packed dwords */
    __shared__ __attribute__((aligned(16))) uint8_t s_nest[HVQ_NESTP_BYTES + 16];  /* nest, two 4-bit values per byte */
    __shared__ __attribute__((aligned(16))) u32 s_pool[HVQ_POOL_LDS];
    __shared__ __attribute__((aligned(16))) WaveLds s_w[HVQ_NW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    STAMP(0, 0);
    /* ---- level 0: the tile record ---- */
    const HvqTileRec *__restrict__ R = recs + blockIdx.x;
    const i32 hb = R->hb;
    if (hb == 0) return;                                  /* padding entry of the XCD-dealt table (uniform exit) */
    const u32 flags = R->flags;
    const u32 pic_kind = (flags >> HVQ_TR_KIND_SHIFT) & 3u;
    const i32 unk = (i32)((flags >> HVQ_TR_UNK_SHIFT) & 31u);
    const int plane_id = (int)(flags >> HVQ_TR_PLANE_SHIFT);
    const bool is_pb = pic_kind != HVQ_PIC_I;
    const bool I_luma = !is_pb && plane_id == 0;
    const bool landscape = flags & HVQ_F_LANDSCAPE;
    const bool is15 = flags & HVQ_F_IS15;
    const bool big = flags & HVQ_F_BIG_AOT;
    const float rhb = R->rhb;
    const u32 nblocks = R->nblocks;
    const u32 b0 = R->b0;
    const i32 ws = R->ws, hs = R->hs;
    const i32 pw = R->pw;
    const i32 mstride = hb + 2;
    const GLB uint8_t *map = (const GLB uint8_t *)R->map;
    const GLB u32 *__restrict__ gpool = (const GLB u32 *)R->pool;
    const GLB u32 *__restrict__ mvs = (const GLB u32 *)R->mv;
    const GLB u32 *__restrict__ gnest = (const GLB u32 *)R->nest;
    const GLB uint8_t *ref0 = (const GLB uint8_t *)R->ref0, *ref1 = (const GLB uint8_t *)R->ref1;
    const i32 plane_off = (i32)R->plane_off;
    GLB uint8_t *plane = (GLB uint8_t *)R->dst;
    const i32 slot = (i32)R->slot_bytes;
    const i32 mcb_w = (i32)R->mcb_w;
    const i32 lw = R->lw;
    const u32 pool_dwords = R->pool_dwords;
    const bool pool_lds = pool_dwords <= HVQ_POOL_LDS;
    const bool need_nest = (flags & HVQ_F_HAS_NEST) && gnest;
    STAMP(1, 0);

    /* ---- level 1: own descriptors per lane; payload and nest for the workgroup ---- */
    const u32 b = b0 + (u32)tid;
    const bool valid = b < nblocks;
    i32 bx, by;
    block_coords(valid ? b : 0u, hb, rhb, bx, by);
    const GLB uint8_t *ent = map + 2 * ((by + 1) * mstride + bx + 1);
    const u32 e16 = *(const GLB uint16_t *)ent;
    const u32 nt = *(const GLB uint16_t *)(ent - 2 * mstride), nbt = *(const GLB uint16_t *)(ent + 2 * mstride);
    const u32 nl = *(const GLB uint16_t *)(ent - 2), nr = *(const GLB uint16_t *)(ent + 2);
    u32 mvw = 0;
    if (is_pb) mvw = mvs[(by >> (1 - hs)) * mcb_w + (bx >> (1 - ws))];
    u32 pq[(HVQ_POOL_LDS + HVQ_WG - 1) / HVQ_WG];
    if (pool_lds) {
#pragma unroll
        for (int j = 0; j < (HVQ_POOL_LDS + HVQ_WG - 1) / HVQ_WG; ++j) {
            pq[j] = 0;
            if ((u32)tid + HVQ_WG * j < pool_dwords) pq[j] = gpool[tid + HVQ_WG * j];
        }
    }
    u32 nq0 = 0, nq1 = 0;
    if (need_nest) {
        nq0 = gnest[tid];
        if (tid + HVQ_WG < (HVQ_NESTP_BYTES + 3) / 4) nq1 = gnest[tid + HVQ_WG];
    }

    const i32 V = e16 & 0xFF;
    const u32 T = valid ? (e16 >> 8) : 0u;
    const bool inter = is_pb && (T & 0x60u);
    const u32 kind = I_luma ? T : (T & 0xFu);
    const u32 npay = valid ? hvq_payload_dwords(T, is_pb, I_luma) : 0u;
    /* class: 0 cheap (done in place), 1 intra AOT, 2 MC + AOT residual; nb = bases (selects only) */
    const bool aot_kind = kind != 0 && kind != 6;
    const bool c1 = valid && !inter && aot_kind && kind != 8;
    const bool c2 = valid && inter && !(T & 0x10u) && aot_kind;
    const int cls = c1 ? 1 : c2 ? 2 : 0;
    const u32 nb = c1 ? kind : c2 ? kind - 1u : 0u;
    /* payload index inside the tile: the wave's base from the record + prefix sum over the wave */
    u32 off = wave == 0 ? 0u : wave == 1 ? R->wrel[0] : wave == 2 ? R->wrel[1] : R->wrel[2];
    if (__ballot(npay != 0)) off += wave_incl_scan(npay) - npay;
    STAMP(2, 1);

    if (pool_lds) {
#pragma unroll
        for (int j = 0; j < (HVQ_POOL_LDS + HVQ_WG - 1) / HVQ_WG; ++j)
            if (tid + HVQ_WG * j < HVQ_POOL_LDS) s_pool[tid + HVQ_WG * j] = pq[j];
    }
    if (need_nest) {
        ((u32 *)s_nest)[tid] = nq0;
        if (tid + HVQ_WG < (HVQ_NESTP_BYTES + 3) / 4) ((u32 *)s_nest)[tid + HVQ_WG] = nq1;
    }
    __syncthreads();                                    /* the only workgroup barrier: payload + nest published */
    STAMP(3, 0);

    /* ---- level 2a: motion-compensation rows of the own block go out first ... ---- */
    const bool needs_mc = valid && inter && (cls == 2 || (T & 0x10u) || kind == 0);
    McRows rows;
    int hx = 0, hy = 0;
    if (needs_mc) {
        const i32 rx = (i32)(int16_t)(mvw & 0xFFFF), ry = (i32)(int16_t)(mvw >> 16);
        const GLB uint8_t *ref = (((T >> 5) & 3u) == 1u) ? ref0 : ref1;
        const i32 pdx = rx >> ws, pdy = ry >> hs;
        hx = is15 ? (pdx & 1) : (rx & 1); hy = is15 ? (pdy & 1) : (ry & 1);              /* h4m:1337-1343 */
        const i32 a = plane_off + (pdy >> 1) * pw + (pdx >> 1) + (by & (1 - hs)) * 4 * pw + (bx & (1 - ws)) * 4;
        rows = mc_load(ref, a, pw, hy, slot - 8);
    }
    /* ---- ... while the kinds that need nothing else are finished ---- */
    if (valid && !needs_mc && cls == 0) {
        Blk o;
        if (!inter && kind == 0) {
            /* neighbour DCs via the map; the border {0x7F,0xFF} never exposes (h4m:1437-1442, 1811-1814).
             * I pictures track the left value separately: only kinds 0 and 8 expose it (h4m:1443-1454). */
            i32 Tt = (nt & 0x7700u) ? V : (i32)(nt & 0xFF);
            i32 Bb = (nbt & 0x7700u) ? V : (i32)(nbt & 0xFF);
            i32 Rr = (nr & 0x7700u) ? V : (i32)(nr & 0xFF);
            bool lexp = is_pb ? !(nl & 0x7700u) : ((nl >> 8) == 0 || (nl >> 8) == 8);
            i32 Ll = lexp ? (i32)(nl & 0xFF) : V;
            o = weight_block(V, Tt, Bb, Ll, Rr);
        } else {
            /* flat DC (h4m:281-286) or literal (h4m:543-549) */
            const u32 v = (u32)V * 0x01010101u;
            o.r[0] = o.r[1] = o.r[2] = o.r[3] = v;
            if (kind == 6) {
                if (pool_lds) { o.r[0] = s_pool[off]; o.r[1] = s_pool[off + 1]; o.r[2] = s_pool[off + 2]; o.r[3] = s_pool[off + 3]; }
                else { o.r[0] = gpool[off]; o.r[1] = gpool[off + 1]; o.r[2] = gpool[off + 2]; o.r[3] = gpool[off + 3]; }
            }
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) s_out[y][tid] = o.r[y];
    }
    STAMP(4, 0);
    if (needs_mc) {
        const Blk o = mc_filter(rows, hx, hy);          /* plain MC, and the MC part of MC-residual blocks */
#pragma unroll
        for (int y = 0; y < 4; ++y) s_out[y][tid] = o.r[y];
    }
    STAMP(5, 0);

    /* ---- AOT blocks of this wave, HVQ_CHUNK at a time ---- */
    const unsigned long long imask = __ballot(cls != 0);
    const u32 nitems = (u32)__popcll(imask);
    if (nitems) {
        WaveLds &W = s_w[wave];
        const u32 idx = lanes_below(imask);
        for (u32 c0 = 0; c0 < nitems; c0 += HVQ_CHUNK) {
            const bool act = cls != 0 && idx - c0 < (u32)HVQ_CHUNK;
            const u32 nchunk = min(nitems - c0, (u32)HVQ_CHUNK);
            /* counting sort by (class, bases descending): bin = class-2 flag * 16 + 15 - min(bases, 15) */
            if (lane < 32) W.hist[lane] = 0;
#pragma unroll
            for (int j = 0; j < 16 * HVQ_CHUNK / 64; ++j) W.acc[lane + 64 * j] = 0;
            WAVE_SYNC();
            const u32 bin = (cls == 2 ? 16u : 0u) + 15u - min(nb, 15u);
            u32 rank = 0;
            if (act) rank = __hip_atomic_fetch_add(&W.hist[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            WAVE_SYNC();
            const u32 hcount = W.hist[lane & 31];
            WAVE_SYNC();
            const u32 hincl = wave_incl_scan(lane < 32 ? hcount : 0u);
            const u32 hexcl = hincl - (lane < 32 ? hcount : 0u);                  /* lane j < 32: items in bins below j */
            const u32 n1 = (u32)__builtin_amdgcn_readlane((int)hexcl, 16);        /* class-1 items of the chunk */
            /* items of class c with more than k bases: E[c*16 + 15 - k] - E[c*16]; lane c*16 + k holds it */
            const int cbase = lane & 16, kk = lane & 15;
            const u32 e_hi = (u32)__builtin_amdgcn_ds_bpermute(4 * (cbase + 15 - kk), (int)hexcl);
            const u32 e_lo = (u32)__builtin_amdgcn_ds_bpermute(4 * cbase, (int)hexcl);
            const u32 cnt = lane < 32 ? e_hi - e_lo : 0u;
            /* pair segment starts: exclusive prefix of cnt over k inside each class (a DPP row = 16 lanes), class 2 after class 1 */
            u32 segi = cnt;
            segi += (u32)__builtin_amdgcn_update_dpp(0, (int)segi, 0x111, 0xF, 0xF, false);
            segi += (u32)__builtin_amdgcn_update_dpp(0, (int)segi, 0x112, 0xF, 0xF, false);
            segi += (u32)__builtin_amdgcn_update_dpp(0, (int)segi, 0x114, 0xF, 0xF, false);
            segi += (u32)__builtin_amdgcn_update_dpp(0, (int)segi, 0x118, 0xF, 0xF, false);
            const u32 P1 = (u32)__builtin_amdgcn_readlane((int)segi, 15), P2 = (u32)__builtin_amdgcn_readlane((int)segi, 31);
            u32 npairs = P1 + P2;
            /* slot in sorted order (every lane takes part in the cross-lane read: a disabled source lane would read as 0) */
            const u32 slotq = (u32)__builtin_amdgcn_ds_bpermute(4 * (int)bin, (int)hexcl) + rank;
            u32 pstart = 0;
            if (!big) {
                if (lane < 32) W.hist[lane] = segi - cnt + (lane >= 16 ? P1 : 0u);
            } else {
                /* more than 15 bases per block possible (I-luma type bytes > 15, never produced by real encoders): the
                 * prefix property of the sorted order does not hold beyond k = 15, pairs are listed item-major */
                const u32 nbm = act ? nb : 0u;
                const u32 pin = wave_incl_scan(nbm);
                pstart = pin - nbm;
                npairs = (u32)__builtin_amdgcn_readlane((int)pin, 63);
            }
            if (act) {
                W.item[slotq] = (u32)lane | (e16 << 8);
                W.imv[slotq] = mvw;
                W.ioff[slotq] = off;
            }
            WAVE_SYNC();
            const u32 rcls = slotq - (cls == 2 ? n1 : 0u);                        /* rank inside the class */
            const u32 bidx = off + (cls == 2 ? 2u : 0u);
            for (u32 w0 = 0; w0 < npairs; w0 += HVQ_PAIRWIN) {
                if (act) {
                    for (u32 k = 0; k < nb; ++k) {
                        const u32 pos = (big ? pstart + k : W.hist[(cls == 2 ? 16u : 0u) + k] + rcls) - w0;
                        if (pos < (u32)HVQ_PAIRWIN) W.pair[pos] = slotq | ((bidx + k) << 5);
                    }
                }
                WAVE_SYNC();
                const u32 wn = min(npairs - w0, (u32)HVQ_PAIRWIN);
                /* ---- one lane per (item, basis) pair ---- */
                for (u32 pi = (u32)lane; pi < wn; pi += 64) {
                    const u32 pr = W.pair[pi];
                    const u32 it = pr & 31u;
                    const u32 d = pool_lds ? s_pool[pr >> 5] : gpool[pr >> 5];
                    u32 e[16], lo, hi;
                    if (it < n1) {
                        gather_nest(d, landscape, s_nest, e, lo, hi);
                    } else {
                        const u32 t16 = W.item[it] >> 8, mv = W.imv[it];
                        const i32 rx = (i32)(int16_t)(mv & 0xFFFF), ry = (i32)(int16_t)(mv >> 16);
                        const GLB uint8_t *ref = ((t16 >> 13) & 3u) == 1u ? ref0 : ref1;
                        const i32 origin = landscape ? rx / 2 + (ry / 2 - 16) * lw - 32 : rx / 2 + (ry / 2 - 32) * lw - 16;
                        gather_window(d, landscape, ref, origin, lw, slot, e, lo, hi);
                    }
                    basis_scatter(basis_gain(d, lo, hi, big), e, big, W.acc + it, HVQ_CHUNK);
                }
                WAVE_SYNC();
            }
            /* ---- one lane per queued block: accumulators -> samples ---- */
            if ((u32)lane < nchunk) {
                const u32 info = W.item[lane];
                const u32 owner = (u32)(wave * 64) + (info & 63u);
                const u32 q16 = info >> 8;
                u32 r[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) r[i] = (u32)W.acc[i * HVQ_CHUNK + lane];
                if (!big) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) r[i] <<= 4;
                }
                Blk o;
                if ((u32)lane >= n1) {
                    const u32 po = W.ioff[lane];
                    const u32 p0 = pool_lds ? s_pool[po] : gpool[po], p1 = pool_lds ? s_pool[po + 1] : gpool[po + 1];
                    Blk m;                                       /* the owner left the MC block in the tile */
#pragma unroll
                    for (int y = 0; y < 4; ++y) m.r[y] = s_out[y][owner];
                    o = predi_finish(r, m, p0, p1, unk);
                } else {
                    o = intra_finish(r, (i32)(q16 & 0xFF), unk);
                }
#pragma unroll
                for (int y = 0; y < 4; ++y) s_out[y][owner] = o.r[y];
            }
            WAVE_SYNC();
        }
    }
    STAMP(6, 1);
    WAVE_SYNC();

    /* ---- the wave's 64 blocks -> HBM ---- */
    if ((hb & 3) == 0) {
        /* lane (g, r): sample row r of blocks 4g..4g+3 = 16 contiguous bytes of the plane */
        const int g = wave * 16 + (lane & 15), rr = lane >> 4;
        const u32 gb = b0 + 4u * (u32)g;
        if (gb < nblocks) {
            i32 gx, gy;
            block_coords(gb, hb, rhb, gx, gy);
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = *(const u32x4 *)&s_out[rr][4 * g];
            /* B pictures are never read again by a later picture: streaming stores keep them from displacing the anchors
             * in L2 (+1 % on MC-dominated streams, neutral on the dense one; profiles/r01j_ab_nontemporal.txt) */
            if (pic_kind == HVQ_PIC_B) __builtin_nontemporal_store(v, (GLB u32x4 *)(plane + (size_t)(gy * 4 + rr) * pw + gx * 4));
            else *(GLB u32x4 *)(plane + (size_t)(gy * 4 + rr) * pw + gx * 4) = v;
        }
    } else if (valid) {
        GLB uint8_t *dst = plane + (size_t)(by * 4) * pw + bx * 4;
#pragma unroll
        for (int y = 0; y < 4; ++y) *(GLB u32 *)(dst + (size_t)y * pw) = s_out[y][tid];
    }
    STAMP(7, 0);
    STAMP(8, 1);
}

extern "C" hipError_t hvq_launch_recon(const HvqTileRec *recs_dev, uint32_t ntiles, hipStream_t stream)
{
    if (ntiles == 0) return hipSuccess;
#ifdef HVQ_STAMPS
    hipLaunchKernelGGL(hvq_recon_kernel, dim3(ntiles), dim3(HVQ_WG), 0, stream, recs_dev, g_stamps);
#else
    hipLaunchKernelGGL(hvq_recon_kernel, dim3(ntiles), dim3(HVQ_WG), 0, stream, recs_dev);
#endif
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------------
 * Display epilogue: YUV 4:2:0 -> RGB24 exactly as the reference player's dumpRGB
 * (h4m:897-926): single-precision, one rounding per operation (the intrinsics below are never contracted
 * into FMAs), clamp, truncate.  Pure streaming kernel: 1.5 B/px read, 3 B/px written; one lane = 4 samples
 * of a row = one dword of Y in, three dwords of RGB out (a wave stores 768 contiguous bytes).
 */
__device__ __forceinline__ u32 rgb_clamp(float f)
{
    return f < 0.f ? 0u : f > 255.f ? 255u : (u32)f;          /* h4m:897-900 */
}

struct HvqRgbJob { const uint8_t *yuv; uint8_t *rgb; int w, h; };

/* four samples: one dword of Y, two bytes each of U and V -> three dwords of RGB */
__device__ __forceinline__ void rgb4(u32 y4, u32 u2, u32 v2, u32 out[3])
{
    out[0] = out[1] = out[2] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float Y = (float)((y4 >> (8 * k)) & 0xFFu);
        const float U = __fsub_rn((float)((u2 >> (8 * (k >> 1))) & 0xFFu), 128.f);
        const float V = __fsub_rn((float)((v2 >> (8 * (k >> 1))) & 0xFFu), 128.f);
        const u32 px[3] = { rgb_clamp(__fadd_rn(Y, __fmul_rn(1.402f, V))),
                            rgb_clamp(__fsub_rn(__fsub_rn(Y, __fmul_rn(0.34414f, U)), __fmul_rn(0.71414f, V))),
                            rgb_clamp(__fadd_rn(Y, __fmul_rn(1.772f, U))) };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int byte = 3 * k + c;
            out[byte >> 2] |= px[c] << (8 * (byte & 3));
        }
    }
}

/* WIDE: one lane = 16 samples of TWO rows that share their chroma (two 16-byte Y loads, one 8-byte U and V load, six
 * 16-byte stores; the chroma products are computed once for the 2x2 samples they serve).  Needs width % 16 == 0 and an
 * even height (4:2:0 has both); otherwise 4 samples of one row per lane. */
template <bool WIDE>
__global__ __launch_bounds__(256)
void hvq_yuv420_rgb_kernel(const HvqRgbJob *__restrict__ jobs)
{
    const HvqRgbJob J = jobs[blockIdx.y];                    /* one picture per grid row */
    const uint8_t *__restrict__ yuv = J.yuv;
    uint8_t *__restrict__ rgb = J.rgb;
    const int w = J.w, h = J.h;
    constexpr int S = WIDE ? 16 : 4;
    const int qw = w / S;                                    /* lanes per row */
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= qw * (WIDE ? h / 2 : h)) return;
    const int yr = idx / qw, xq = idx - yr * qw;
    const int y = WIDE ? 2 * yr : yr;
    const uint8_t *yp = yuv + (size_t)y * w + S * xq;
    const uint8_t *up = yuv + (size_t)w * h + (size_t)(y >> 1) * (w >> 1) + (S / 2) * xq;
    const uint8_t *vp = up + (size_t)(w >> 1) * (h >> 1);
    u32 *dst = (u32 *)(rgb + ((size_t)y * w + S * xq) * 3);
    if (WIDE) {
        const uint4 ya = *(const uint4 *)yp, yb = *(const uint4 *)(yp + w);
        const uint2 u8 = *(const uint2 *)up, v8 = *(const uint2 *)vp;
        const u32 us[4] = { u8.x & 0xFFFFu, u8.x >> 16, u8.y & 0xFFFFu, u8.y >> 16 };
        const u32 vs[4] = { v8.x & 0xFFFFu, v8.x >> 16, v8.y & 0xFFFFu, v8.y >> 16 };
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const uint4 y16 = row ? yb : ya;
            const u32 ys[4] = { y16.x, y16.y, y16.z, y16.w };
            u32 o[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) rgb4(ys[q], us[q], vs[q], o + 3 * q);     /* the chroma terms are common subexpressions of the two rows */
            u32 *d = dst + row * (3 * w / 4);
#pragma unroll
            for (int q = 0; q < 3; ++q) ((uint4 *)d)[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
    } else {
        u32 o[3];
        rgb4(*(const u32 *)yp, *(const uint16_t *)up, *(const uint16_t *)vp, o);
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    }
}

/* jobs_dev: array of {yuv, rgb, w, h} in device memory; max_lanes = max over jobs of (w/4)*h; wide = every
 * width is a multiple of 16 */
extern "C" hipError_t hvq_launch_rgb(const void *jobs_dev, int njobs, int max_lanes, int wide, hipStream_t stream)
{
    if (njobs <= 0) return hipSuccess;
    if (wide)
        hipLaunchKernelGGL(hvq_yuv420_rgb_kernel<true>, dim3((max_lanes / 8 + 255) / 256, njobs), dim3(256), 0, stream,
                           (const HvqRgbJob *)jobs_dev);
    else
        hipLaunchKernelGGL(hvq_yuv420_rgb_kernel<false>, dim3((max_lanes + 255) / 256, njobs), dim3(256), 0, stream,
                           (const HvqRgbJob *)jobs_dev);
    return hipGetLastError();
}
