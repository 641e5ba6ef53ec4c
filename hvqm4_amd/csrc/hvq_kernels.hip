/*
 * hvq_kernels.hip -- HVQM4 picture reconstruction for CDNA4 / gfx950 (MI355X).
 *
 * One launch reconstructs a BATCH of pictures (one job per picture, any mix of streams, sizes and picture kinds).  One
 * 256-thread workgroup = one or two tiles of 256 consecutive 4x4 blocks of one plane (raster order); a block is four packed
 * dwords (4 samples per dword).  Because intra prediction reads neighbour DC values from the descriptor map, not neighbour
 * pixels, every block of a picture is independent: no intra-picture wavefront dependency
 * exists and the whole batch is data-parallel.
 *
 * Integer/byte work, no dense contraction: no MFMA.  Kernels (measurements in DESIGN.md section 5):
 *   hvq_recon_inline_kernel   per dependency level, the default: descriptors (map, vectors, payload pool) -> block records, item
 *                      queue and pair list in registers and LDS -> flat / weighted-DC / motion-compensated blocks by the owning
 *                      lane, pairs -> nest or window gather, gain, 16 products -> LDS accumulators (ds_add), items -> samples;
 *                      the tile is assembled in LDS and leaves as 16-byte row segments (every store instruction of a wave
 *                      writes four complete 256-byte runs, every output line reaches HBM once and whole)
 *   hvq_selfref_kernel P pictures with future-referencing macroblocks: the reference's raster-order walk
 *   hvq_yuv420_rgb_kernel, hvq_gather_kernel   display epilogue, bulk readback
 *   - sample arithmetic is SIMD-within-register: v_lerp_u8 for the 2-tap and 4-tap half-sample filters, 16-bit packed math for
 *     the weighted-DC predictor, v_sad_u8 for block sums; the AOT products in the reference's own uint32 wrap arithmetic
 *     (v_mul_lo_u32); the reference's divTable / mcdivTable lookups are one v_rcp_f32 and a biased multiply (udiv_table, no fix-up);
 *   - reference pictures are addressed LINEARLY inside the Y|U|V buffer exactly like the reference's pointer arithmetic
 *      as ring base + 32-bit offset; every address is clamped to the picture slot so malformed vectors cannot
 *     fault the GPU.
 * What bounds the reconstruction: vector-instruction issue (VALU 0.67-0.8 busy on every dependency level).  Round 3's
 * two-pass variant (queues built once per picture into HBM by a kernel of their own) lost as a stage for three rounds and was
 * deleted in round 6; its measurements are in profiles/HISTORY.md.
 *
 * Reference behaviour restated per device function (h4m: = h4m_audio_decode.c).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

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

__device__ __forceinline__ McRows mc_load(const GLB uint8_t *ref, i32 a, i32 stride, int hy)
{
    McRows r;
#pragma unroll
    for (int y = 0; y < 4; ++y) r.q[y] = *(const GLB u64u *)(ref + a + y * stride);
    r.q[4] = 0;
    if (hy) r.q[4] = *(const GLB u64u *)(ref + a + 4 * stride);   /* 5th row only for vertical half samples */
    return r;
}

__device__ __forceinline__ Blk mc_filter(const McRows &rows, int hx, int hy)
{
    /* ONE formula for the four half-sample cases (round 5; rounds 1-4 ran two code paths side by side in every wave):
     * (a + b' + c' + d' + 2) >> 2 with b' = the right neighbour when hx, else a itself, and c', d' = the same pair of the row below
     * when hy, else of this row -- (4a + 2) >> 2 = a, (2a + 2b + 2) >> 2 = (a + b + 1) >> 1, and the 4-tap case as it is.
     * Four samples per instruction: with h = floor((a + b') / 2) per row (v_lerp_u8, no rounding bit) the result is
     * (h + h' + 1) >> 1, plus one when both pair sums were odd and h + h' is even. */
    const uint64_t *q = rows.q;
    Blk o;
    /* rows are consumed one at a time (the pair of the previous row stays in two registers): all five at once cost six registers more
     * than the kernel has */
    u32 hp, xp;
    {
        const u32 p = (u32)q[0], n = hx ? (u32)(q[0] >> 8) : p;
        hp = __builtin_amdgcn_lerp(p, n, 0u); xp = p ^ n;
    }
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const u32 p = (u32)q[y + 1], n = hx ? (u32)(q[y + 1] >> 8) : p;
        const u32 hn = __builtin_amdgcn_lerp(p, n, 0u), xn = p ^ n;
        const u32 h1 = hy ? hn : hp, x1 = hy ? xn : xp;
        const u32 t = __builtin_amdgcn_lerp(hp, h1, 0x01010101u);
        o.r[y] = t + (xp & x1 & ~(hp ^ h1) & 0x01010101u);
        hp = hn; xp = xn;
    }
    return o;
}


/* floor(N / den) for the two numerators the tables use (N = 256 with den <= 15, N = 4096 with den <= 255), 0 for den = 0, without the
 * integer fix-up: where N / den is an integer den is a power of two and its reciprocal exact; everywhere else the quotient lies at
 * least 1 / 255 below the next integer, a hundred times the error of v_rcp_f32 (1 ulp) and the multiply -- and the bias of 1 / 1024
 * added before truncation covers the integer cases against a reciprocal that came out an ulp low.  Checked against the tables for
 * every den by the parity suite's kernels on the GPU (tests/test_gpu_parity.py::test_table_divisions_on_the_gpu). */
template <int N>
__device__ __forceinline__ u32 udiv_table(u32 den)
{
    const u32 q = (u32)__builtin_fmaf(__builtin_amdgcn_rcpf((float)den), (float)N, 0.0009765625f);
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

/*
 * AOT arithmetic (h4m:679-817).  reference: factor = (sum + off) * (+-divTable[max-min]);
 * acc[i] += factor * e[i] (uint32 wrap).  divTable[r] = 16 * (256 / r).  Everything is done in the reference's own
 * uint32 wrap arithmetic: v_mul_lo_u32 issues at the same rate as the 24-bit multiplies on gfx950
 * (profiles/r02c_ubench_valu_rate.txt), so there is no separate path for blocks with more than 15 bases.
 * The host stores the running coefficient sum in every basis dword, so bases are independent and are
 * processed one per lane; their products meet in LDS with ds_add_u32.
 */
__device__ __forceinline__ u32 basis_gain(u32 d, u32 lo, u32 hi)
{
    const u32 q = udiv_table<256>((hi - lo) & 15u);
    const u32 s = d >> 14;
    const u32 g = s * (q << 4);
    return (d & 0x2000u) ? 0u - g : g;
}

template <int STRIDE>
__device__ __forceinline__ void basis_scatter(u32 g, const u32 e[16], u32 *acc_lds)
{
#if defined(HVQ_ABL) && HVQ_ABL == 5            /* ablation: all products, ONE LDS add (prices the 16 same-address atomics) */
    u32 t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t ^= g * e[i];
    __hip_atomic_fetch_add(acc_lds, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
#pragma unroll
    for (int i = 0; i < 16; ++i)
        __hip_atomic_fetch_add(acc_lds + i * STRIDE, g * e[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

/* four samples: arithmetic shift right, clamp to [0, 255], pack -- gfx950's v_ashr_pk_u8_i32 does two samples per instruction (it
 * writes the low 16 bits of its destination and leaves the upper 16 as they were: the compiler's own pattern match of the clamp
 * assumed them cleared and corrupted samples 2 and 3, section 6 item 1 of DESIGN.md; v_perm takes exactly the two valid bytes of each) */
__device__ __forceinline__ u32 shr_sat_pack4(u32 a, u32 b, u32 c, u32 d, i32 s)
{
    const u32 lo = (u32)__builtin_amdgcn_ashr_pk_u8_i32((i32)a, (i32)b, (u32)s);
    const u32 hi = (u32)__builtin_amdgcn_ashr_pk_u8_i32((i32)c, (i32)d, (u32)s);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
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
        o4.r[y] = shr_sat_pack4(r[4 * y] + delta, r[4 * y + 1] + delta, r[4 * y + 2] + delta, r[4 * y + 3] + delta, unk);
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
    const u32 mean = sum >> 4;
    u32 lo = 255, hi = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const u32 v = (m.r[y] >> (8 * x)) & 0xFFu;
            lo = min(lo, v);
            hi = max(hi, v);
        }
    const u32 mcd = udiv_table<4096>(hi - lo);                       /* mcdivTable[max-min], h4m:272, 1407 */
    const u32 factor = p1 * mcd;
    /* value = r + addend + (px - mean) * factor  with addend = p0 - mean_aot, all in uint32 wrap arithmetic (h4m:1405-1416):
     * = r + px * factor + (p0 - mean_aot - mean * factor) */
    const u32 c = p0 - mean_aot - mean * factor;
    Blk o4;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        u32 v[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const u32 px = (m.r[y] >> (8 * x)) & 0xFFu;                 /* extracted again: sixteen live samples are registers the kernel does not have */
            v[x] = (u32)(((i32)(r[4 * y + x] + c + px * factor) >> unk) + (i32)px);    /* the sample is added BEHIND the shift: exactly so */
        }
        o4.r[y] = shr_sat_pack4(v[0], v[1], v[2], v[3], 0);
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
#ifndef HVQ_ABL
#define HVQ_ABL 0
#endif
#ifndef HVQ_NT_LOADS
#define HVQ_NT_LOADS 0         /* experiment: the map entries loaded non-temporally */
#endif
#ifndef HVQ_PRIO
#define HVQ_PRIO 0            /* 1: head at issue priority 3 until trip 2 is issued; 2: until the motion-compensation rows are requested */
#endif
#ifndef HVQ_PLANE_LOAD
#define HVQ_PLANE_LOAD 1       /* 1 (round 6): the plane record by a dependent scalar load; 0: all three records + selects (rounds 4-5).  A/B profiles/r06d_plane_load_ab.txt */
#endif
#ifndef HVQ_NT_STORES
#define HVQ_NT_STORES 1        /* B pictures leave with non-temporal stores (0: plain stores; A/B in profiles/r05_recon_steps.txt) */
#endif

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
        if (stamps && (threadIdx.x & 63) == 0) stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + (threadIdx.x >> 6) * 16 + (i)] = t_; } while (0)
#else
#define HVQ_STAMP_ARG
#define STAMP(i, VM) do { } while (0)
#endif

typedef u32 u32x2 __attribute__((ext_vector_type(2)));
/* not `volatile`: a volatile asm counts as a store to anything, and every scalar load after it would become a vector load */
#define HVQ_PIN(x) do { x = (u32)__builtin_amdgcn_readfirstlane((int)(x)); asm("" : "+s"(x)); } while (0)

/* ------------------------------------------------------------------------------------------------------
 * P pictures with future-referencing (type 2) macroblocks.  HVQM4DecodePpic passes the picture being written as `future`
 * (h4m:2058-2061), so such a macroblock reads `present` in whatever state the raster-order walk of BpicPlaneDec
 * (h4m:1919-1967) has left it: new samples where an earlier macroblock has been, the buffer's previous content elsewhere --
 * its own destination included (the copy loops of _MotionComp_* read and write sample by sample, h4m:1242-1279).
 * Encoders do not emit this; it is reproduced for parity, not for speed: every other macroblock has been reconstructed into
 * `side` by the data-parallel kernel, `dst` holds the previous content, and ONE workgroup walks the macroblocks in the
 * reference's order -- all threads move finished macroblocks from `side`, thread 0 restates the reference's scalar code for
 * the type-2 ones, reading `dst` through L2 (agent-scope accesses) in exactly the reference's order.
 */
struct SrGeo {
    GLB uint8_t *pic;                  /* the picture being written (`present`) */
    i32 lim;                           /* last readable byte offset */
};
__device__ __forceinline__ u32 sr_ld(const SrGeo &g, i32 o)
{
    o = clampi(o, 0, g.lim);
    return __hip_atomic_load(g.pic + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sr_st(const SrGeo &g, i32 o, u32 v)
{
    o = clampi(o, 0, g.lim);
    __hip_atomic_store(g.pic + o, (uint8_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* _MotionComp (h4m:1242-1294): sample (i, j) of the half-sample filtered source at `so`, read NOW */
__device__ __forceinline__ u32 sr_mc_sample(const SrGeo &g, i32 so, i32 stride, int hx, int hy, int i, int j)
{
    const i32 a = so + i * stride + j;
    if (!hx && !hy) return sr_ld(g, a);
    if (hx && !hy) return (sr_ld(g, a) + sr_ld(g, a + 1) + 1u) / 2u;
    if (!hx) return (sr_ld(g, a) + sr_ld(g, a + stride) + 1u) / 2u;
    return (sr_ld(g, a) + sr_ld(g, a + 1) + sr_ld(g, a + stride) + sr_ld(g, a + stride + 1) + 2u) >> 2;
}

__global__ __launch_bounds__(HVQ_WG)
void hvq_selfref_kernel(const HvqJob *__restrict__ J, const uint8_t *__restrict__ side, uint8_t *pic)
{
    const int tid = threadIdx.x;
    const u32 flags = J->flags;
    const bool is15 = flags & HVQ_F_IS15, landscape = flags & HVQ_F_LANDSCAPE;
    const i32 unk = (i32)((flags >> HVQ_JOB_UNK_SHIFT) & 31u);
    const i32 lw = (i32)J->width, mcb_w = (i32)J->mcb_w;
    const i32 nmb = mcb_w * (i32)((J->plane[0].hbvb >> 16) / 2u);
    const GLB u32 *__restrict__ pool = (const GLB u32 *)J->pool;
    const GLB u32 *__restrict__ mvs = (const GLB u32 *)J->mv;
    const GLB u32 *__restrict__ offs = (const GLB u32 *)((const GLB uint8_t *)J->tq + J->q_offs_off);
    SrGeo g;
    g.pic = (GLB uint8_t *)pic;
    g.lim = (i32)J->slot_bytes - 1;
    struct { const GLB uint8_t *map; i32 hb, pw, ws, hs, poff; u32 tile_first; } P[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        P[k].map = (const GLB uint8_t *)J->plane[k].map;
        P[k].hb = (i32)(J->plane[k].hbvb & 0xFFFFu);
        P[k].pw = (i32)(J->plane[k].pw_sub & 0xFFFFu);
        P[k].ws = (i32)((J->plane[k].pw_sub >> 16) & 0xFFu); P[k].hs = (i32)(J->plane[k].pw_sub >> 24);
        P[k].poff = (i32)J->plane[k].plane_off;
        P[k].tile_first = J->plane[k].tile_first;
    }
    bool moved = false;
    for (i32 m = 0; m < nmb; ++m) {
        const i32 my = m / mcb_w, mx = m - my * mcb_w;
        const u32 T = (u32)__builtin_amdgcn_readfirstlane((int)P[0].map[2 * ((2 * my + 1) * (P[0].hb + 2) + 2 * mx + 1) + 1]);
        if (((T >> 5) & 3u) != 2u) {
            /* finished by the data-parallel pass: its samples move from the side buffer into the picture */
            int k = -1, r = 0, col = 0;
            const int cw = 8 >> P[1].ws, ch = 8 >> P[1].hs, cn = cw * ch;
            if (tid < 64) { k = 0; r = tid >> 3; col = tid & 7; }
            else if (tid - 64 < cn) { k = 1; r = (tid - 64) / cw; col = (tid - 64) - r * cw; }
            else if (tid - 64 - cn < cn) { k = 2; r = (tid - 64 - cn) / cw; col = (tid - 64 - cn) - r * cw; }
            if (k >= 0) {
                const i32 o = P[k].poff + (my * (8 >> P[k].hs) + r) * P[k].pw + mx * (8 >> P[k].ws) + col;
                pic[o] = side[o];
            }
            moved = true;
            continue;
        }
        if (moved) { __syncthreads(); moved = false; }            /* what was moved is in L2 before thread 0 reads the picture */
        if (tid == 0) {
            const u32 mvw = mvs[m];
            const i32 rx = (i32)(int16_t)(mvw & 0xFFFF), ry = (i32)(int16_t)(mvw >> 16);      /* absolute half-sample target (h4m:1954-1955) */
            const bool proc = T & 0x10u;
            const i32 origin = landscape ? rx / 2 + (ry / 2 - 16) * lw - 32 : rx / 2 + (ry / 2 - 32) * lw - 16;   /* h4m:1865-1868 */
            for (int k = 0; k < 3; ++k) {
                const i32 pw = P[k].pw, ws = P[k].ws, hs = P[k].hs;
                const i32 pdx = rx >> ws, pdy = ry >> hs;
                const int hx = is15 ? (pdx & 1) : (rx & 1), hy = is15 ? (pdy & 1) : (ry & 1);       /* h4m:1337-1343, 1889-1896 */
                const int bxp = 2 >> ws, byp = 2 >> hs, nblk = bxp * byp;
                for (int j = 0; j < nblk; ++j) {                                                   /* TL, BL, BR, TR (h4m:447-455) */
                    const int dx = nblk == 1 ? 0 : (j >> 1), dy = nblk == 1 ? 0 : ((j == 1 || j == 2) ? 1 : 0);
                    const i32 bx = mx * bxp + dx, by = my * byp + dy;
                    const i32 dsto = P[k].poff + by * 4 * pw + bx * 4;
                    const i32 srco = P[k].poff + (pdy >> 1) * pw + (pdx >> 1) + dy * 4 * pw + dx * 4;
                    const u32 kind = (u32)P[k].map[2 * ((by + 1) * (P[k].hb + 2) + bx + 1) + 1] & 0xFu;
                    const u32 b = (u32)(by * P[k].hb + bx);
                    const u32 off = offs[(size_t)(P[k].tile_first + b / HVQ_TILE_BLOCKS) * HVQ_TILE_BLOCKS + b % HVQ_TILE_BLOCKS];
                    if (!proc && kind == 6u) {                                                     /* OrgBlock, h4m:543-549 */
                        for (int i = 0; i < 4; ++i) {
                            const u32 w4 = pool[off + (u32)i];
                            for (int x = 0; x < 4; ++x) sr_st(g, dsto + i * pw + x, (w4 >> (8 * x)) & 0xFFu);
                        }
                    } else if (proc || kind == 0u) {
                        /* _MotionComp straight into the picture: sample by sample, each read sees what has been written so far */
                        for (int i = 0; i < 4; ++i)
                            for (int x = 0; x < 4; ++x) sr_st(g, dsto + i * pw + x, sr_mc_sample(g, srco, pw, hx, hy, i, x));
                    } else {
                        /* PrediAotBlock (h4m:1379-1420): the AOT sum over the window first, then the MC block into a temporary,
                         * only then the destination */
                        u32 acc[16];
                        for (int i = 0; i < 16; ++i) acc[i] = 0;
                        const u32 nb = kind - 1u;
                        for (u32 q = 0; q < nb; ++q) {
                            const u32 d = pool[off + 2u + q];
                            const i32 ol = d & 0x3F, os = (d >> 6) & 0x1F;
                            const u32 sl = (d >> 11) & 1, ss = (d >> 12) & 1;
                            i32 o, ys, xs;
                            if (landscape) { o = lw * os + ol; xs = 1 << sl; ys = lw << ss; }
                            else           { o = lw * ol + os; xs = 1 << ss; ys = lw << sl; }
                            u32 e[16], lo = 255, hi = 0;
                            for (int i = 0; i < 4; ++i)
                                for (int x = 0; x < 4; ++x) {
                                    const u32 v = (sr_ld(g, origin + o + i * ys + x * xs) >> 4) & 15u;
                                    e[4 * i + x] = v; lo = min(lo, v); hi = max(hi, v);
                                }
                            const u32 gn = basis_gain(d, lo, hi);
                            for (int i = 0; i < 16; ++i) acc[i] += gn * e[i];
                        }
                        u32 total = 0;
                        for (int i = 0; i < 16; ++i) total += acc[i];
                        const u32 mean_aot = (u32)((i32)total >> 4);
                        u32 md[16], sum = 8, lo = 255, hi = 0;
                        for (int i = 0; i < 4; ++i)
                            for (int x = 0; x < 4; ++x) { const u32 v = sr_mc_sample(g, srco, pw, hx, hy, i, x); md[4 * i + x] = v; sum += v; lo = min(lo, v); hi = max(hi, v); }
                        const i32 mean = (i32)(sum >> 4);
                        const u32 addend = pool[off] - mean_aot;
                        const u32 factor = pool[off + 1u] * udiv_table<4096>(hi - lo);            /* mcdivTable[max - min] */
                        for (int i = 0; i < 16; ++i) {
                            const u32 t = (u32)((i32)md[i] - mean) * factor;
                            const i32 v = sar(acc[i] + addend + t, unk) + (i32)md[i];
                            sr_st(g, dsto + (i >> 2) * pw + (i & 3), (u32)clampi(v, 0, 255));
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

extern "C" hipError_t hvq_launch_selfref(const HvqJob *job_dev, const uint8_t *side, uint8_t *dst, hipStream_t stream)
{
    hipLaunchKernelGGL(hvq_selfref_kernel, dim3(1), dim3(HVQ_WG), 0, stream, job_dev, side, dst);
    return hipGetLastError();
}

__device__ __forceinline__ u32 gain_q(u32 w0, u32 lo, u32 hi)
{
    const u32 q = udiv_table<256>((hi - lo) & 15u);                 /* divTable[max - min] = 16 * (256 / r), h4m:265-271 */
    const u32 g = (w0 & 0x3FFFFu) * (q << 4);
    return (w0 & HVQ_PQ_NEG) ? 0u - g : g;
}

/* nest gather of one decoded pair, intra (h4m:713-725): o = index of sample (0,0), ys = row stride, both in 4-bit units */
__device__ __forceinline__ void gather_nest_q(i32 o, bool x2, i32 ys, const uint8_t *s_nest, u32 e[16], u32 &lo, u32 &hi)
{
    const u32 sh = x2 ? 8u : 4u;
    lo = 255; hi = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const i32 n = o + y * ys;
        uint64_t q = *(const u64u *)(s_nest + (n >> 1));
        q >>= 4 * (n & 1);
        const u32 w0 = (u32)q;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            u32 v = (w0 >> (sh * x)) & 15u;
            e[4 * y + x] = v;
            lo = min(lo, v);
            hi = max(hi, v);
        }
    }
}

/* window gather of one decoded pair, MC residual (h4m:734-765): voff = ring offset of sample (0,0), ys = row stride */
__device__ __forceinline__ void window_load(const GLB uint8_t *ring, u32 voff, u32 ys, uint64_t q[4])
{
    /* timing experiments (wrong pictures): 19 = two row loads instead of four; 20 = one; 21 = four, from a 64 KB window of the ring (cache hits) */
    if (HVQ_ABL == 19) { q[0] = *(const GLB u64u *)(ring + (size_t)voff); q[1] = *(const GLB u64u *)(ring + (size_t)(u32)(voff + ys)); q[2] = q[0] ^ 1; q[3] = q[1] ^ 1; return; }
    if (HVQ_ABL == 20) { q[0] = *(const GLB u64u *)(ring + (size_t)voff); q[1] = q[0] ^ 1; q[2] = q[0] ^ 2; q[3] = q[1] ^ 3; return; }
    if (HVQ_ABL == 21) voff &= 0xFFFFu;
#pragma unroll
    for (int y = 0; y < 4; ++y) q[y] = *(const GLB u64u *)(ring + (size_t)(u32)(voff + (u32)y * ys));
}

__device__ __forceinline__ void window_finish(const uint64_t q[4], bool x2, u32 e[16], u32 &lo, u32 &hi)
{
    const u32 sel = x2 ? 0x06040200u : 0x03020100u;
    lo = 255; hi = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        u32 w = __builtin_amdgcn_perm((u32)(q[y] >> 32), (u32)q[y], sel);
        w = (w >> 4) & 0x0F0F0F0Fu;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            u32 v = (w >> (8 * x)) & 0xFFu;
            e[4 * y + x] = v;
            lo = min(lo, v);
            hi = max(hi, v);
        }
    }
}


/* ------------------------------------------------------------------------------------------------------
 * Reconstruction WITHOUT a queue-build pass (round 4; front end rewritten in round 5): the workgroup derives its block records,
 * item queue and pair list from the picture's descriptors itself -- in registers and LDS, nothing of it touches HBM.
 *
 * Round 4's kernel was bound by vector-instruction issue (569 VALU per wave of 64 blocks, 0.8 busy), and two thirds of those
 * instructions were not pixel work but addressing and classification.  Round 5 keeps the phases and trims their cost:
 *   - a wave's 64 blocks are CONSECUTIVE in raster order, so their (row, column) is the wave's first block split ONCE on the
 *     scalar unit (multiply-high by a per-plane constant of the job record) plus the lane index with at most one wrap; every
 *     descriptor address is a scalar base + a 32-bit lane offset (no 64-bit vector address arithmetic, no float division);
 *   - the class of a block is computed by code specialised for the plane's context (I-picture luma, I-picture chroma, P/B
 *     picture; chosen by a scalar branch), so the selects on picture-uniform conditions are gone;
 *   - payload offset, intra pair count and MC-residual pair count come out of ONE packed wave scan instead of two;
 *   - the pair list is written by a short unrolled sequence of predicated stores at ascending addresses instead of a loop
 *     with a bounds test per entry;
 *   - phase C re-derives its rows' coordinates the same way.
 *
 *   trip 1  the picture's job record (scalar): its common part, then -- round 6 -- the workgroup's ONE plane record by a dependent load
 *   trip 2  per block: map entry with both horizontal neighbours (one unaligned 8-byte load), the vertical neighbours, the
 *           macroblock vector; per wave: its pool offset (wave_base); by LDS-DMA the nest and the tile's range of the payload pool
 *   then    class, block operands (MC source offset with the version's half-sample rule h4m:1327-1355, weighted-DC neighbours
 *           h4m:1437-1454), one packed scan, item and pair slots from ONE packed 64-bit LDS atomic per wave (intra items fill the
 *           accumulator rows from the bottom, MC-residual items from the top: no total is needed before a slot can be handed out)
 *   trip 3  motion-compensation rows
 *   barrier 1, phase B1 (lane = pair: basis dword from LDS, decoded here, nest rows from LDS or window rows from the
 *   reference), barrier 2, phase B2 (lane = item), barrier 3, phase C (16-byte row segments, complete 256-byte runs).
 */
template <int CTX>   /* 0 I-picture luma (kind = the whole type byte, h4m:1093), 1 I-picture chroma, 2 P/B picture */
__device__ __forceinline__ void inl_classify(u32 T, bool valid, u32 &cls, u32 &nb, u32 &npay, bool &lit, bool &mc, bool &wdc, bool &flat)
{
    const u32 kind = CTX == 0 ? T : (T & 0xFu);
    const bool k0 = kind == 0u, k6 = kind == 6u, k8 = kind == 8u;
    if (CTX == 2) {
        const bool inter = T & 0x60u, proc = T & 0x10u;
        const bool c1 = valid && !inter && !(k0 | k6 | k8), c2 = valid && inter && !proc && !(k0 | k6);
        lit = valid && k6 && !(inter && proc);
        mc = valid && inter && (proc || !k6); wdc = valid && !inter && k0; flat = valid && !inter && k8;
        cls = c1 ? 1u : c2 ? 2u : 0u;
        nb = c1 ? kind : c2 ? kind - 1u : 0u;
        npay = lit ? 4u : nb + (c2 ? 2u : 0u);
    } else {
        const bool c1 = valid && !(k0 | k6 | k8);
        lit = valid && k6; mc = false; wdc = valid && k0; flat = valid && k8;
        cls = c1 ? 1u : 0u; nb = c1 ? kind : 0u; npay = lit ? 4u : nb;
    }
}

template <int N> struct InlCtx { static constexpr int value = N; };
#define HVQ_W64(i) ((uint64_t)cw[i] | ((uint64_t)cw[(i) + 1] << 32))

template <int ITEMS_CAP, int TPW>
__global__ __launch_bounds__(HVQ_WG, HVQ_MIN_WAVES)
void hvq_recon_inline_kernel(const HvqJob *__restrict__ jobs, u32 pair_cap, u32 pool_cap HVQ_STAMP_ARG)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];       /* [pair_cap] item | pool index << 9, then [pool_cap] staged pool dwords */
    __shared__ __attribute__((aligned(16))) uint8_t s_nest[HVQ_NESTP_BYTES + 8];
    __shared__ __attribute__((aligned(16))) u32 s_out[TPW][4][HVQ_WG];
    __shared__ __attribute__((aligned(16))) u32 s_acc[16 * ITEMS_CAP];
    __shared__ u32 s_item0[ITEMS_CAP];   /* owner (tile of the workgroup * 256 + lane) | map entry << 10 */
    __shared__ u32 s_item1[ITEMS_CAP];   /* pool index of the block's payload */
    __shared__ u32 s_item2[ITEMS_CAP];   /* MC-residual items: the macroblock vector (the pair lanes derive the origin of the 70x38 window from it, h4m:1865-1868) */
    __shared__ unsigned long long s_ctr64;   /* intra items | MC-residual items << 10 | intra pairs << 20 | MC-residual pairs << 42 handed out */
    u32 *const s_pair = s_dyn;
    u32 *const s_pool = s_dyn + pair_cap;

    const int tid = threadIdx.x;
    const u32 lane = (u32)tid & 63u;
    STAMP(0, 0);
#if HVQ_PRIO == 1 || HVQ_PRIO == 2
    /* issue priority experiment: a new wave's head (a few instructions between long waits) goes in front of the older waves' arithmetic,
     * so that its loads are under way while they compute (the arbiter is oldest-first otherwise) */
    __builtin_amdgcn_s_setprio(3);
#endif
    const u32 slot_id = blockIdx.z * gridDim.x + blockIdx.x;
    const u32 wg = blockIdx.y;
    const HvqJob *__restrict__ J = jobs + slot_id;
#if HVQ_PLANE_LOAD
    /* The job record's common part first (with the planes' first tiles, which say which plane this workgroup belongs to), then the ONE
     * plane record it needs by a dependent scalar load: a scalar round trip more at the head (measured at zero for the kernel arguments,
     * profiles/r05_recon_steps.txt B) against 16 scalar registers and ~50 scalar instructions of loading all three records and
     * selecting among them. */
    const u32 *__restrict__ CW = (const u32 *)J;
    u32 cw[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) cw[i] = CW[i];
    u32 wb_lo = CW[46], wb_hi = CW[47], q_offs_off = CW[49];
#pragma unroll
    for (int i = 0; i < 20; ++i) HVQ_PIN(cw[i]);
    HVQ_PIN(wb_lo); HVQ_PIN(wb_hi); HVQ_PIN(q_offs_off);
    const u32 total_tiles = cw[17];
    const u32 n0 = cw[18], n1 = cw[19] - cw[18], n2 = total_tiles - cw[19];
    const u32 pf1 = (n0 + TPW - 1) / TPW, pf2 = pf1 + (n1 + TPW - 1) / TPW, pend = pf2 + (n2 + TPW - 1) / TPW;
    if (total_tiles == 0 || wg >= pend) return;                              /* picture dropped by the flush */
    const int p = (wg >= pf1) + (wg >= pf2);
    const u32 *__restrict__ PW = CW + 20 + 8 * p;
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = PW[i];
    u32 magic = CW[52 + p], magic16 = CW[55 + p];
#pragma unroll
    for (int i = 0; i < 8; ++i) HVQ_PIN(w[i]);
    HVQ_PIN(magic); HVQ_PIN(magic16);
#else
    const u32 *__restrict__ PW = (const u32 *)&J->plane[0];
    u32 w0[8], w1[8], w2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { w0[i] = PW[i]; w1[i] = PW[8 + i]; w2[i] = PW[16 + i]; }
    const u32 *__restrict__ CW = (const u32 *)J;
    u32 cw[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) cw[i] = CW[i];
    u32 wb_lo = CW[46], wb_hi = CW[47], q_offs_off = CW[49];
    u32 mg0 = CW[52], mg1 = CW[53], mg2 = CW[54], ms0 = CW[55], ms1 = CW[56], ms2 = CW[57];
#pragma unroll
    for (int i = 0; i < 8; ++i) { HVQ_PIN(w0[i]); HVQ_PIN(w1[i]); HVQ_PIN(w2[i]); }
#pragma unroll
    for (int i = 0; i < 18; ++i) HVQ_PIN(cw[i]);
    HVQ_PIN(wb_lo); HVQ_PIN(wb_hi); HVQ_PIN(q_offs_off);
    HVQ_PIN(mg0); HVQ_PIN(mg1); HVQ_PIN(mg2); HVQ_PIN(ms0); HVQ_PIN(ms1); HVQ_PIN(ms2);
    const u32 total_tiles = cw[17];
    const u32 n0 = w1[5] - w0[5], n1 = w2[5] - w1[5], n2 = total_tiles - w2[5];
    const u32 pf1 = (n0 + TPW - 1) / TPW, pf2 = pf1 + (n1 + TPW - 1) / TPW, pend = pf2 + (n2 + TPW - 1) / TPW;
    if (total_tiles == 0 || wg >= pend) return;                              /* picture dropped by the flush */
    const int p = (wg >= pf1) + (wg >= pf2);
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = p == 0 ? w0[i] : p == 1 ? w1[i] : w2[i];
    const u32 magic = p == 0 ? mg0 : p == 1 ? mg1 : mg2, magic16 = p == 0 ? ms0 : p == 1 ? ms1 : ms2;
#endif
    const u32 hbvb = w[6], pw_sub = w[7], tile_first = w[5];
    const u32 nplane_tiles = p == 0 ? n0 : p == 1 ? n1 : n2;
    const u32 pairw = wg - (p == 0 ? 0u : p == 1 ? pf1 : pf2);
    const u32 tile0 = tile_first + (u32)TPW * pairw;
    /* wg < pend: the workgroup has at least one tile; with one tile per workgroup that is all there is to know (a constant frees the
     * scalar registers the 64-bit `h < ntl` masks held from here to phase C: the kernel sits at its cap of 80) */
    const int ntl = TPW == 1 ? 1 : (int)min((u32)TPW, nplane_tiles - (u32)TPW * pairw);
    const uint64_t map_a = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    const uint64_t dst_a = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    const u32 plane_off = w[4];
    const u32 hb = hbvb & 0xFFFFu;
    const u32 flags = cw[13];
    const u32 pic_kind = (flags >> HVQ_JOB_KIND_SHIFT) & 3u;
    const i32 unk = (i32)((flags >> HVQ_JOB_UNK_SHIFT) & 31u);
    const bool is_pb = pic_kind != HVQ_PIC_I;
    const bool landscape = flags & HVQ_F_LANDSCAPE;
    const bool is15 = flags & HVQ_F_IS15;
    const u32 nblocks = hb * (hbvb >> 16);
    const u32 b0 = (u32)TPW * pairw * HVQ_TILE_BLOCKS;
    const u32 ws = (pw_sub >> 16) & 0xFFu, hs = pw_sub >> 24;
    const u32 pw = pw_sub & 0xFFFFu;
    const u32 mstride = hb + 2u;
    const GLB uint8_t *map = (const GLB uint8_t *)map_a;
    const GLB uint8_t *ring = (const GLB uint8_t *)HVQ_W64(0);
    const u32 ref0_off = cw[2], ref1_off = cw[3];
    const GLB u32 *__restrict__ pool = (const GLB u32 *)HVQ_W64(4);
    const GLB u32 *__restrict__ mvs = (const GLB u32 *)HVQ_W64(6);
    const GLB uint8_t *__restrict__ qb = (const GLB uint8_t *)HVQ_W64(8);
    const GLB u32 *__restrict__ nestp = (const GLB u32 *)HVQ_W64(10);
    const GLB u32 *__restrict__ wave_base = (const GLB u32 *)((uint64_t)wb_lo | ((uint64_t)wb_hi << 32));
    GLB uint8_t *plane = (GLB uint8_t *)dst_a;
    const i32 slot = (i32)cw[12];
    const u32 lw = cw[14];
    const u32 mcb_w = cw[15];
    const u32 pool_dwords = cw[16];
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane(tid >> 6);
    STAMP(1, 0);                                                               /* job record here */
    /* the slot counter is zero before any wave asks it: a barrier HERE, where no vector-memory operation is outstanding yet, instead
     * of one behind trip 2 that would hold every wave until the slowest wave's loads have landed */
    if (tid == 0) s_ctr64 = 0;
    __syncthreads();

    /* ---- trip 2: the blocks' descriptors into registers; the tile range of the pool and the nest straight into LDS (LDS-DMA) ---- */
    /* (wave_base through the VECTOR path -- one load, v_readlane -- was measured: -4 % dense, profiles/r05_recon_steps.txt; the scalar
     * loads stay: their latency overlaps the map loads, a vector load's wait does not) */
    u32 plo = wave_base[tile0 * HVQ_NW];
    u32 phi = tile0 + (u32)ntl < total_tiles ? wave_base[(tile0 + (u32)ntl) * HVQ_NW] : pool_dwords;
    u32 wbase[TPW];
#pragma unroll
    for (int h = 0; h < TPW; ++h) wbase[h] = h < ntl ? wave_base[(tile0 + (u32)h) * HVQ_NW + wave] : 0u;
    bool valid[TPW];
    u32 bx[TPW], by[TPW];
    uint64_t row8[TPW];
    u32 nt[TPW], nbt[TPW], mvw[TPW];
#pragma unroll
    for (int h = 0; h < TPW; ++h) {
        /* A wave's 64 blocks are consecutive in raster order: the first one is split into (row, column) on the SCALAR unit --
         * multiply-high by floor(2^32 / hb) + 1, which is the quotient or one more (block index < 2^22), put right by one compare --
         * and a lane adds its index, wrapping into the next row(s): at most once in rows of 64 blocks and more; shorter rows
         * (t < 128) divide exactly by a 16-bit reciprocal.  No per-lane division, and the map addresses below are a scalar base
         * plus a 32-bit lane offset. */
        const u32 bw = b0 + (u32)(h * HVQ_TILE_BLOCKS) + wave * 64u;          /* uniform */
        const bool livew = h < ntl && bw < nblocks;
        valid[h] = h < ntl && bw + lane < nblocks;
        const u32 bws = livew ? bw : 0u;
        u32 by0 = __umulhi(bws, magic);
        if ((i32)(bws - by0 * hb) < 0) by0 -= 1u;
        const u32 bx0 = bws - by0 * hb;
        const u32 l = valid[h] ? lane : 0u;
        const u32 t = bx0 + l;
        u32 q;
        if (hb >= 64u) q = t >= hb ? 1u : 0u;
        else q = __umul24(t, magic16) >> 16;
        bx[h] = t - __umul24(q, hb); by[h] = by0 + q;
        /* entry (by, bx) of the bordered map = b + 2 by + hb + 3 */
        const GLB uint8_t *mw = map + 2u * (size_t)(bws + 2u * by0 + hb + 3u);
        const u32 vo2 = 2u * (l + 2u * q);
#if HVQ_NT_LOADS
        row8[h] = __builtin_nontemporal_load((const GLB u64u *)((mw - 2) + vo2));
#else
        row8[h] = *(const GLB u64u *)((mw - 2) + vo2);                         /* left, own, right entries (the map has a border) */
#endif
        /* timing experiments (wrong pictures): 41 no vertical-neighbour loads, 42 neither those nor the vector load, 43 = 42 and no nest / pool staging */
        if (HVQ_ABL >= 41 && HVQ_ABL <= 43) { nt[h] = (u32)row8[h] & 0xFFFFu; nbt[h] = (u32)(row8[h] >> 32) & 0xFFFFu; }
        else { nt[h] = *(const GLB uint16_t *)((mw - 2u * (size_t)mstride) + vo2); nbt[h] = *(const GLB uint16_t *)((mw + 2u * (size_t)mstride) + vo2); }
        mvw[h] = 0;
        if (HVQ_ABL == 42 || HVQ_ABL == 43) { if (is_pb) { const u32 hh = ((bx[h] >> (1u - ws)) * 73u + (by[h] >> (1u - hs)) * 151u) * 2654435761u; mvw[h] = (((hh >> 8) & 1023u) + 16u) | ((((hh >> 18) & 511u) + 64u) << 16); } }
        else if (is_pb) mvw[h] = *(const GLB u32 *)((const GLB uint8_t *)mvs + 4u * (__umul24(by[h] >> (1u - hs), mcb_w) + (bx[h] >> (1u - ws))));
    }
    /* LDS-DMA in 16-byte pieces (gfx950: global_load_lds_dwordx4): the nest is 84 of them, a dense tile's pool range ~80 -- two
     * instructions of the first two waves each instead of two per wave (round 4 moved dwords: the staging was 8 % of the step,
     * profiles/r05_recon_steps.txt).  The pool range starts at the 16-byte boundary below `plo` (the pool section is 16-byte aligned). */
    typedef __attribute__((address_space(3))) u32 lds_u32;
    const bool has_nest = (HVQ_W64(10) != 0) && HVQ_ABL != 43;
    if (has_nest) {
        constexpr u32 NCH = HVQ_NESTP_BYTES / 16u;                       /* 84 */
        static_assert(HVQ_NESTP_BYTES % 16 == 0, "nest in 16-byte pieces");
#pragma unroll
        for (u32 r0 = 0; r0 < NCH; r0 += HVQ_WG)
            if (r0 + (u32)tid < NCH)
                __builtin_amdgcn_global_load_lds((const GLB u32 *)((const GLB uint8_t *)nestp + 16u * (r0 + (u32)tid)), (lds_u32 *)((u32 *)s_nest + 4u * (r0 + wave * 64u)), 16, 0, 0);
    }
    phi = max(phi, plo);
    const u32 plo4 = plo & ~3u;
    const u32 nst = HVQ_ABL == 43 ? 0u : min(phi - plo4, pool_cap);    /* staged dwords, from plo4 on (pool_cap is a multiple of 4) */
    {
        const GLB uint8_t *pool_lo = (const GLB uint8_t *)(pool + plo4);
        const u32 nch = (nst + 3u) >> 2;
        for (u32 r0 = 0; r0 < nch; r0 += HVQ_WG)
            if (r0 + (u32)tid < nch) __builtin_amdgcn_global_load_lds((const GLB u32 *)(pool_lo + 16u * (r0 + (u32)tid)), (lds_u32 *)(s_pool + 4u * (r0 + wave * 64u)), 16, 0, 0);
    }
    /* staged dwords as the lanes compare against: everything (every index the descriptors produce lies inside the tile's range) unless the
     * tile's payload exceeds the launch's LDS share.  One scalar, not a uniform condition: as a 64-bit mask that condition lived in two
     * scalar registers from here to the item phase, was spilled to vector-register lanes and read back at every call (two tiles). */
    const u32 nst_lim = phi - plo4 <= pool_cap ? 0xFFFFFFFFu : nst;
    auto pool_at = [&](u32 idx) -> u32 {                                       /* a dword of the payload pool: staged, or (beyond the staging cap) from HBM */
        const u32 j = idx - plo4;
        return j < nst_lim ? s_pool[j] : pool[min(idx, pool_dwords ? pool_dwords - 1u : 0u)];
    };

#if HVQ_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    STAMP(2, 0);                                                               /* trip 2 issued (the stamp itself waits for the scalar loads) */
    STAMP(3, 1);                                                               /* ... and landed (stamped builds wait here) */
    /* Everything from here to barrier 1 exists three times, once per context of the plane (I-picture luma, I-picture chroma, P/B
     * picture), chosen by ONE scalar branch: no select on a picture-uniform condition is executed per lane, the I-picture copies
     * contain no motion compensation at all, and the per-lane predicates never cross a merge point (where the compiler would park
     * them in vector registers). */
    auto front = [&](auto ctxc) {
        constexpr int CTX = decltype(ctxc)::value;
        u32 off[TPW], cls[TPW], nb[TPW], e16v[TPW], w0v[TPW], pinI[TPW], pinM[TPW], npayv[TPW];
        bool lit[TPW], mcb[TPW], wdcb[TPW], flatb[TPW], hxb[TPW], hyb[TPW];
        unsigned long long m1[TPW], m2[TPW];
        McRows rows[TPW];
#pragma unroll
        for (int h = 0; h < TPW; ++h) {
            const u32 e16 = (u32)(row8[h] >> 16) & 0xFFFFu;
            const u32 T = e16 >> 8, V = e16 & 0xFFu;
            e16v[h] = e16;
            u32 npay;
            inl_classify<CTX>(T, valid[h], cls[h], nb[h], npay, lit[h], mcb[h], wdcb[h], flatb[h]);
            npayv[h] = npay;
            if (HVQ_ABL == 36 || HVQ_ABL == 37) { cls[h] = 0; nb[h] = 0; }          /* timing experiments: no queue derivation, no AOT work at all */
            w0v[h] = 0; hxb[h] = false; hyb[h] = false;
            if (CTX == 2) {
                /* plain MC and the MC part of MC-residual blocks (h4m:1327-1355): half-sample rule per version (h4m:1337-1343);
                 * computed for every lane (a few operations), used by the motion-compensated ones */
                const i32 rx = (i32)(int16_t)(mvw[h] & 0xFFFFu), ry = (i32)mvw[h] >> 16;
                const u32 roff = (T & 0x60u) == 0x20u ? ref0_off : ref1_off;
                const u32 hsx = is15 ? ws : 0u, hsy = is15 ? hs : 0u;                /* 1.5: the plane's own vector decides, 1.3: the luma vector */
                hxb[h] = (rx >> hsx) & 1; hyb[h] = (ry >> hsy) & 1;
                /* HVQ_ABL 38 (timing experiment): every block fetched from its own position (vector 0), half-sample flags kept: prices the scatter */
                const i32 rowi = HVQ_ABL == 38 ? (i32)(by[h] << 2) : (ry >> (hs + 1u)) + (i32)((by[h] & (1u - hs)) << 2);   /* |rowi| < 2^15, pw < 2^14: a 24-bit multiply is exact */
                const i32 coli = (HVQ_ABL == 38 ? (i32)(bx[h] << 2) : (rx >> (ws + 1u)) + (i32)((bx[h] & (1u - ws)) << 2)) + (i32)plane_off;
                i32 a = __mul24(rowi, (i32)pw) + coli;
                /* one clamp for the block: legal vectors keep all rows inside the slot, malformed ones cannot fault */
                const i32 hi3 = slot - 8 - 3 * (i32)pw, hi4 = hi3 - (i32)pw;
                a = clampi(a, 0, hyb[h] ? hi4 : hi3);
                if (mcb[h]) w0v[h] = roff + (u32)a;
            }
        }
        /* ---- trip 3: motion-compensation rows, requested before the rest of the front (neighbour values, scan, slots, lists) so that
         * those 6 000 cycles run beside the gathers instead of in front of them ---- */
        if (CTX == 2) {
#pragma unroll
            for (int h = 0; h < TPW; ++h) {
                if (mcb[h]) {
                    const u32 vo = w0v[h];
                    /* timing experiments (tools/variant.sh <name> -DHVQ_ABL=n; wrong pictures): 31 no phase-A arithmetic, 32 no item epilogues,
                     * 33 no pair work, 34 no motion-compensation row loads, 35 no stores */
                    if (HVQ_ABL == 34 || HVQ_ABL == 37) {
#pragma unroll
                        for (int y = 0; y < 5; ++y) rows[h].q[y] = (uint64_t)vo * 0x0101010101ull + (uint64_t)y;
                        continue;
                    }
#pragma unroll
                    for (int y = 0; y < 4; ++y) rows[h].q[y] = *(const GLB u64u *)(ring + (size_t)(u32)(vo + (u32)y * pw));
                    rows[h].q[4] = 0;
                    if (hyb[h]) rows[h].q[4] = *(const GLB u64u *)(ring + (size_t)(u32)(vo + 4u * pw));
                }
            }
        }
#if HVQ_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int h = 0; h < TPW; ++h) {
            const u32 V = e16v[h] & 0xFFu;
            const u32 npay = npayv[h];
            if (wdcb[h]) {
                /* neighbour DCs via the map; the border {0x7F,0xFF} never exposes (h4m:1437-1442, 1811-1814).
                 * I pictures track the left value separately: only kinds 0 and 8 expose it (h4m:1443-1454). */
                const u32 nlf = (u32)row8[h] & 0xFFFFu, nr = (u32)(row8[h] >> 32) & 0xFFFFu;
                const u32 Tt = (nt[h] & 0x7700u) ? V : (nt[h] & 0xFFu);
                const u32 Bb = (nbt[h] & 0x7700u) ? V : (nbt[h] & 0xFFu);
                const u32 Rr = (nr & 0x7700u) ? V : (nr & 0xFFu);
                const bool lexp = CTX == 2 ? !(nlf & 0x7700u) : ((nlf >> 8) & ~8u) == 0u;
                const u32 Ll = lexp ? (nlf & 0xFFu) : V;
                w0v[h] = Tt | (Bb << 8) | (Ll << 16) | (Rr << 24);
            }
            /* ONE scan: payload dwords [10:0], intra bases [20:11], MC-residual bases [30:21] (P/B: kinds <= 15, a wave's sums stay
             * below 2^10 / 2^11); I pictures have no MC-residual blocks and kinds up to 255: payload [15:0], intra bases [31:16] */
            m1[h] = __ballot(cls[h] == 1); m2[h] = CTX == 2 ? __ballot(cls[h] == 2) : 0ull;
            if (CTX == 2) {
                const u32 inc = wave_incl_scan(npay + (nb[h] << (cls[h] == 1 ? 11 : 21)));
                off[h] = wbase[h] + (inc & 0x7FFu) - npay;
                pinI[h] = (inc >> 11) & 0x3FFu; pinM[h] = inc >> 21;
            } else {
                const u32 inc = wave_incl_scan(npay + (nb[h] << 16));
                off[h] = wbase[h] + (inc & 0xFFFFu) - npay;
                pinI[h] = inc >> 16; pinM[h] = 0;
            }
        }
        STAMP(4, 0);                                                           /* classes, rows requested, operands, scan */
        if (q_offs_off) {                                               /* self-referencing P picture: hvq_selfref_kernel wants the pool offsets */
#pragma unroll
            for (int h = 0; h < TPW; ++h)
                if (h < ntl) ((GLB u32 *)(qb + q_offs_off))[(size_t)(tile0 + (u32)h) * HVQ_TILE_BLOCKS + (u32)tid] = off[h];
        }

        /* ---- slots: one lane per wave asks the counters (intra items upwards, MC-residual items downwards, pairs) ---- */
        u32 slotq[TPW], pstart[TPW];
#pragma unroll
        for (int h = 0; h < TPW; ++h) {
            const u32 c1 = (u32)__popcll(m1[h]), c2 = (u32)__popcll(m2[h]);
            const u32 np1 = (u32)__builtin_amdgcn_readlane((int)pinI[h], 63), np2 = CTX == 2 ? (u32)__builtin_amdgcn_readlane((int)pinM[h], 63) : 0u;
            /* ONE returning LDS atomic per wave and tile hands out all four ranges: intra items [9:0], MC-residual items [19:10],
             * intra pairs [41:20], MC-residual pairs [63:42] */
            u32 b1 = 0, b2 = 0, bp1 = 0, bp2 = 0;
            if (c1 | c2) {
                unsigned long long got = 0;
                if (lane == 0)
                    got = __hip_atomic_fetch_add(&s_ctr64, (unsigned long long)c1 | ((unsigned long long)c2 << 10) | ((unsigned long long)np1 << 20) | ((unsigned long long)np2 << 42),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const u32 glo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)got), ghi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(got >> 32));
                b1 = glo & 1023u; b2 = (glo >> 10) & 1023u;
                bp1 = (glo >> 20) | ((ghi & 1023u) << 12); bp2 = ghi >> 10;
            }
            if (CTX == 2) {
                slotq[h] = cls[h] == 1 ? b1 + lanes_below(m1[h]) : (u32)ITEMS_CAP - 1u - (b2 + lanes_below(m2[h]));
                pstart[h] = cls[h] == 1 ? bp1 + pinI[h] - nb[h] : bp2 + pinM[h] - nb[h];      /* MC-residual pairs: counted from the top */
            } else {
                slotq[h] = b1 + lanes_below(m1[h]);
                pstart[h] = bp1 + pinI[h] - nb[h];
            }
        }
        /* accumulators zeroed: 16 * ITEMS_CAP dwords, ITEMS_CAP a multiple of 32 */
        {
            typedef u32 u32x4z __attribute__((ext_vector_type(4)));
#pragma unroll
            for (u32 i = (u32)tid; i < 4u * ITEMS_CAP; i += HVQ_WG) ((u32x4z *)s_acc)[i] = (u32x4z)(0u);
        }
#pragma unroll
        for (int h = 0; h < TPW; ++h) {
            if (cls[h] && slotq[h] < (u32)ITEMS_CAP) {
                s_item0[slotq[h]] = (u32)(h * HVQ_WG + tid) | (e16v[h] << 10);
                s_item1[slotq[h]] = off[h];
                const bool up = CTX != 2 || cls[h] == 1;
                if (!up) s_item2[slotq[h]] = mvw[h];     /* the macroblock's vector: the pair lanes derive the window origin from it (a block lane doing it
                                                            ran a dozen instructions with one or two lanes of its wave active) */
                /* the item's pairs, at ASCENDING list positions: intra pairs lie at pstart .. pstart + nb - 1 in basis order, MC-residual
                 * pairs (counted from the top of the list) at pair_cap - pstart - nb .. in reverse basis order.  Entries beyond the list
                 * (more pairs than the launch reserved: the tile turns serial and the list is not read) are cut off by the count. */
                const u32 ent0 = slotq[h] | ((off[h] + (up ? 0u : 2u)) << 9);
                const u32 room = max(pair_cap, pstart[h]) - pstart[h];         /* saturating */
                const u32 kmax = min(nb[h], room);
                const u32 pos = up ? pstart[h] : room - kmax;
                const u32 val = up ? ent0 : ent0 + (kmax << 9) - 512u;
                const u32 dv = up ? 512u : 0u - 512u;
                u32 *const sp = s_pair + pos;
#pragma unroll
                for (u32 j = 0; j < 5; ++j)
                    if (j < kmax) sp[j] = val + j * dv;
                if (kmax > 5u) {
#pragma clang loop unroll(disable) vectorize(disable)
                    for (u32 j = 5; j < kmax; ++j) sp[j] = val + j * dv;
                }
            }
        }

#if HVQ_PRIO == 4
        __builtin_amdgcn_s_setprio(3);                                         /* 4: everything from phase A on goes first */
#endif
        STAMP(5, 0);                                                           /* rows requested, slots, items and pairs in LDS */
        STAMP(6, 1);                                                           /* rows landed */
        /* ---- phase A: the blocks the owning lane reconstructs by itself ---- */
#pragma unroll
        for (int h = 0; h < TPW; ++h) {
            const i32 V = (i32)(e16v[h] & 0xFFu);
            Blk o;
            if (HVQ_ABL == 31 && (mcb[h] || wdcb[h])) {
                o.r[0] = (u32)rows[h].q[0] ^ w0v[h]; o.r[1] = (u32)rows[h].q[1]; o.r[2] = (u32)rows[h].q[2]; o.r[3] = (u32)(rows[h].q[3] ^ rows[h].q[4]);
            } else if (CTX == 2 && mcb[h]) {
                o = mc_filter(rows[h], hxb[h], hyb[h]);
            } else if (wdcb[h]) {
                const u32 nb4 = w0v[h];
                o = weight_block(V, (i32)(nb4 & 0xFF), (i32)((nb4 >> 8) & 0xFF), (i32)((nb4 >> 16) & 0xFF), (i32)(nb4 >> 24));
            } else if (flatb[h]) {
                const u32 v = (u32)V * 0x01010101u;
                o.r[0] = o.r[1] = o.r[2] = o.r[3] = v;
            } else continue;
#pragma unroll
            for (int y = 0; y < 4; ++y) s_out[h][y][tid] = o.r[y];
        }
        STAMP(7, 0);                                                           /* phase A */
        /* the nest and the pool range were staged by LDS-DMA (global_load_lds), which completes on the VECTOR-memory counter: s_barrier
         * implies no vmcnt wait and __syncthreads() emits lgkmcnt(0) only, so the issuing waves wait explicitly before they signal the
         * barrier after which every wave reads s_nest / s_pool.  Free: the wave's own loads (issued before the DMA, completed in order)
         * were waited for long ago -- the ISA had an incidental vmcnt(0) behind trip 2; this one does not depend on the scheduler. */
        __builtin_amdgcn_s_waitcnt(0x0F70);                                    /* vmcnt(0), gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait */
        __syncthreads();                                                       /* barrier 1: queues, zeroed accumulators, staged pool and nest */
        STAMP(8, 0);
        /* literal blocks (h4m:543-549): the owner copies its 16 samples from the staged pool (complete only now: other waves staged parts of it) */
#pragma unroll
        for (int h = 0; h < TPW; ++h)
            if (lit[h]) {
#pragma unroll
                for (int y = 0; y < 4; ++y) s_out[h][y][tid] = pool_at(off[h] + (u32)y);
            }
    };
    if (is_pb) front(InlCtx<2>{});
    else if (p == 0) front(InlCtx<0>{});
    else front(InlCtx<1>{});

    const unsigned long long ctr = s_ctr64;
    const u32 nI = min((u32)ctr & 1023u, (u32)ITEMS_CAP), nP = min((u32)(ctr >> 10) & 1023u, (u32)ITEMS_CAP - nI);
    const u32 npI = (u32)(ctr >> 20) & 0x3FFFFFu, npM = (u32)(ctr >> 42);
    const u32 nitems = nI + nP;
    const bool serial = npI + npM > pair_cap;                                   /* more pairs than the launch reserved (pathological): items walk their bases */
    const u32 npairs = (serial || HVQ_ABL == 33) ? 0u : npI + npM;

    if (nitems) {
        /* ---- phase B1: one lane per (item, basis) pair ---- */
        const i32 nstride = landscape ? 70 : 38;
        /* one basis of item `it` (dword d of the pool, h4m:683-711): nest or window gather, gain, 16 products into the item's accumulators */
        auto do_basis = [&](u32 it, u32 d) {
            const i32 ol = d & 0x3F, os = (d >> 6) & 0x1F;
            const u32 sl = (d >> 11) & 1, ss = (d >> 12) & 1;
            const bool x2 = landscape ? sl : ss;
            const u32 y2 = landscape ? ss : sl;
            const u32 wq0 = (d >> 14) | ((d & 0x2000u) ? HVQ_PQ_NEG : 0u);
            u32 e[16], lo, hi;
            if (it >= nI) {
                const u32 t16 = s_item0[it] >> 10;
                const u32 roff = ((t16 >> 13) & 3u) == 1u ? ref0_off : ref1_off;
                const i32 o = landscape ? (i32)lw * os + ol : (i32)lw * ol + os;
                const i32 ys = (i32)lw << y2;
                /* origin of the item's 70x38 window (h4m:1865-1868): vector / 2 truncated towards zero, minus (32, 16) or (16, 32) samples */
                const u32 mv = s_item2[it];
                const i32 rx = (i32)(int16_t)(mv & 0xFFFFu), ry = (i32)mv >> 16;
                const i32 rx2 = (rx - (rx >> 31)) >> 1, ry2 = (ry - (ry >> 31)) >> 1;
                const i32 origin = __mul24(ry2, (i32)lw) + rx2 - (landscape ? 16 * (i32)lw + 32 : 32 * (i32)lw + 16);
                const u32 voff = roff + (u32)clampi(origin + o, 0, slot - 8 - 3 * ys);     /* one clamp: legal windows lie inside the slot */
                uint64_t wq[4];
                window_load(ring, voff, (u32)ys, wq);
                window_finish(wq, x2, e, lo, hi);
            } else {
                const i32 o = landscape ? nstride * os + ol : nstride * ol + os;
                gather_nest_q(o, x2, nstride << y2, s_nest, e, lo, hi);
            }
            basis_scatter<ITEMS_CAP>(gain_q(wq0, lo, hi), e, s_acc + it);
        };
        if (!serial) {
            /* intra pairs on the FIRST lanes of the workgroup, MC-residual pairs on the LAST ones (round 5; they followed each other before):
             * with at most 256 pairs no wave runs both gathers unless the two ranges meet inside it -- a dense tile has ~60 of each */
            const bool psplit = npairs <= (u32)HVQ_WG;
            for (u32 v = (u32)tid; v < (psplit ? (u32)HVQ_WG : npairs); v += HVQ_WG) {
                u32 pi;                                                        /* list position: intra pairs from the bottom, MC-residual pairs from the top */
                if (psplit) {
                    const u32 back = (u32)(HVQ_WG - 1) - v;
                    if (v < npI) pi = v;
                    else if (back < npM) pi = pair_cap - 1u - back;
                    else continue;
                } else pi = v < npI ? v : pair_cap - 1u - (v - npI);
                const u32 pr = s_pair[pi];
                do_basis(pr & 511u, pool_at(pr >> 9));
            }
        } else {
            /* more pairs than the launch's list holds (pathological streams; HVQM4_AMD_PAIR_CAP in the tests): one lane per ITEM walks its
             * bases -- into the same accumulators, so that the item phase below is the same code either way */
            for (u32 v = (u32)tid; v < nitems; v += HVQ_WG) {
                const u32 it = v < nI ? v : (u32)ITEMS_CAP - 1u - (v - nI);
                const u32 q16 = s_item0[it] >> 10;
                const u32 kind = (q16 >> 8) & ((is_pb || p != 0) ? 0xFu : 0xFFu);   /* I-picture luma: the kind is the whole byte (h4m:1093) */
                const bool item_mc = v >= nI;
                const u32 n = item_mc ? (kind & 0xFu) - 1u : kind;
                const u32 bases = s_item1[it] + (item_mc ? 2u : 0u);
                for (u32 k = 0; k < n; ++k) do_basis(it, pool_at(bases + k));
            }
        }
        STAMP(9, 1);                                                           /* pair phase incl. its window rows */
        __syncthreads();                                                       /* barrier 2: accumulators complete */
        STAMP(10, 0);
        /* ---- phase B2: one lane per item ---- */
        /* MC-residual items (the expensive epilogue) on the first lanes, intra items on the LAST lanes of the workgroup: with at
         * most 256 items no wave runs both epilogues unless the two ranges meet inside it, and the two kinds finish side by side */
        const bool split = nitems <= (u32)HVQ_WG;
        for (u32 v = (u32)tid; v < (split ? (u32)HVQ_WG : nitems); v += HVQ_WG) {
            bool item_mc;
            u32 it;
            if (split) {
                const u32 back = (u32)(HVQ_WG - 1) - v;
                item_mc = v < nP;
                if (!item_mc && back >= nI) continue;
                it = item_mc ? (u32)ITEMS_CAP - 1u - v : back;
            } else {
                item_mc = v >= nI;
                it = v < nI ? v : (u32)ITEMS_CAP - 1u - (v - nI);
            }
            const u32 item = s_item0[it];
            const u32 owner = item & 1023u, q16 = item >> 10;
            const u32 poff = s_item1[it];
            u32 r[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = s_acc[i * ITEMS_CAP + it];
            u32 *so = &s_out[0][0][0] + (owner >> 8) * (4 * HVQ_WG) + (owner & 255u);
            Blk o;
            if (HVQ_ABL == 32) {
                o.r[0] = r[0] ^ poff; o.r[1] = r[5]; o.r[2] = r[10]; o.r[3] = r[15] ^ q16;
            } else if (item_mc) {
                Blk m;                                       /* the owner left the MC block in the tile */
#pragma unroll
                for (int y = 0; y < 4; ++y) m.r[y] = so[y * HVQ_WG];
                o = predi_finish(r, m, pool_at(poff), pool_at(poff + 1u), unk);
            } else {
                o = intra_finish(r, (i32)(q16 & 0xFF), unk);
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) so[y * HVQ_WG] = o.r[y];
        }
    }
    STAMP(11, 0);                                                              /* item phase */
#if HVQ_PRIO == 3
    __builtin_amdgcn_s_setprio(3);                                             /* 3: the stores of a finished tile go first */
#endif
    __syncthreads();                                                           /* barrier 3: tiles complete in LDS */
    STAMP(12, 0);

    /* ---- phase C: tiles -> HBM ---- */
    if ((HVQ_ABL == 35 || HVQ_ABL == 37) && s_out[0][0][tid] != 0x12345678u) return;
    /* The wave's first block is split into (row, column) AGAIN here, on the scalar unit (six instructions), from copies of b0, hb and
     * nblocks the compiler cannot connect with the head's: what trip 2 derived -- per tile the two coordinates and three 64-bit lane
     * masks -- would otherwise stay in scalar registers from the head to this point, the kernel sits at its cap of 80 (above it a CU
     * takes seven workgroups, not eight), and the surplus went to vector-register lanes: 7 (one tile) / 59 (two tiles) v_writelane and
     * as many and more v_readlane per wave, each a vector instruction in a kernel bound by those. */
    u32 b0c = b0, hbc = hb, nbc = nblocks;
    asm volatile("" : "+s"(b0c), "+s"(hbc), "+s"(nbc));
#pragma unroll
    for (int h = 0; h < TPW; ++h) {
        if (h >= ntl) continue;
        const u32 bw = b0c + (u32)(h * HVQ_TILE_BLOCKS) + wave * 64u;
        if ((hbc & 3u) == 0) {
            /* lane (g, r): sample row r of blocks 4g .. 4g + 3 = 16 contiguous bytes of the plane (rows are multiples of 4 blocks) */
            const u32 g4 = 4u * (lane & 15u), rr = lane >> 4;
            const u32 gb = bw + g4;
            if (gb < nbc) {
                /* the row's four blocks lie in one map row (rows are multiples of 4 blocks, the wave's first block is one of 64) */
                const u32 bws = bw < nbc ? bw : 0u;
                u32 cy0 = __umulhi(bws, magic);
                if ((i32)(bws - cy0 * hbc) < 0) cy0 -= 1u;
                const u32 cx0 = bws - cy0 * hbc;
                const u32 t = cx0 + g4;
                u32 q;
                if (hbc >= 64u) q = t >= hbc ? 1u : 0u;
                else q = __umul24(t, magic16) >> 16;
                const u32 gx = t - __umul24(q, hbc), gy = cy0 + q;
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = *(const u32x4 *)&s_out[h][rr][wave * 64u + g4];
                const u32 doff = (gy * 4u + rr) * pw + gx * 4u;
                /* B pictures are never read again by a later picture: streaming stores keep them from displacing the anchors in L2.
                 * Written as inline assembly: with __builtin_nontemporal_store on one side of the branch the compiler merges the two
                 * stores into one and drops the hint (rounds 2-4 shipped without it, unnoticed: the ISA had no `nt`). */
#if HVQ_NT_STORES
                if (HVQ_NT_STORES == 2 || pic_kind == HVQ_PIC_B) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(doff), "v"(v), "s"(plane) : "memory");
                else
#endif
                    *(GLB u32x4 *)(plane + (size_t)doff) = v;
            }
        } else if (bw + lane < nbc) {
            i32 sx, sy;
            block_coords(bw + lane, (i32)hbc, __builtin_amdgcn_rcpf((float)hbc), sx, sy);
            GLB uint8_t *dst = plane + (size_t)(sy * 4) * pw + sx * 4;
#pragma unroll
            for (int y = 0; y < 4; ++y) *(GLB u32 *)(dst + (size_t)y * pw) = s_out[h][y][tid];
        }
    }
    STAMP(13, 0);                                                              /* stores issued */
    STAMP(14, 1);                                                              /* stores acknowledged */
}

template <int ITEMS_CAP, int TPW>
static void launch_recon_inline(const HvqJob *jobs_dev, uint32_t nslots, uint32_t max_wgs, uint32_t pair_cap, uint32_t pool_cap, hipStream_t stream)
{
    const dim3 grid = nslots >= 8 ? dim3(8, max_wgs, nslots / 8) : dim3(nslots, max_wgs, 1);
    /* HVQM4_AMD_LDS_PAD (measurements only): unused dynamic LDS behind the pair list and the staged pool, to run a launch at a lower
     * residency than its own footprint allows (what does the step cost with 7, 6, 5 workgroups per CU?) */
    static const size_t lds_pad = getenv("HVQM4_AMD_LDS_PAD") ? (size_t)atoi(getenv("HVQM4_AMD_LDS_PAD")) : 0;
#ifdef HVQ_STAMPS
    hipLaunchKernelGGL((hvq_recon_inline_kernel<ITEMS_CAP, TPW>), grid, dim3(HVQ_WG), (size_t)(pair_cap + pool_cap) * 4u + lds_pad, stream, jobs_dev, pair_cap, pool_cap, g_stamps);
#else
    hipLaunchKernelGGL((hvq_recon_inline_kernel<ITEMS_CAP, TPW>), grid, dim3(HVQ_WG), (size_t)(pair_cap + pool_cap) * 4u + lds_pad, stream, jobs_dev, pair_cap, pool_cap);
#endif
}

/* dynamic LDS in bytes of a launch with these caps (as hvq_launch_recon_inline rounds them) */
extern "C" uint32_t hvq_recon_inline_dyn_lds(uint32_t pair_cap, uint32_t pool_cap)
{
    return 4u * (((std::max(pair_cap, 1u) + 3u) & ~3u) + ((pool_cap + 3u) & ~3u));
}

/* static LDS of hvq_recon_inline_kernel<items_cap, tpw> (the host sizes the dynamic part against the CU's 160 KB) */
extern "C" uint32_t hvq_recon_inline_static_lds(uint32_t tiles_per_wg, uint32_t items_cap)
{
    return (uint32_t)(HVQ_NESTP_BYTES + 8 + 15) / 16u * 16u + tiles_per_wg * 4u * HVQ_WG * 4u + 64u * items_cap + 12u * items_cap + 16u;
}

/* One launch = one dependency level (of one launch queue).  jobs_dev: the launch's picture slots, one job each (count a multiple of 8 when
 * there are at least 8 pictures; padding: total_tiles 0); tiles_per_wg: 1 or 2; max_wgs: the most workgroups of any picture of the launch
 * at that setting; items_cap: the most items of any workgroup of the launch (it selects the instantiation with the next larger accumulator
 * array); pair_cap / pool_cap: dwords of dynamic LDS for the pair list and the staged pool */
extern "C" hipError_t hvq_launch_recon_inline(const HvqJob *jobs_dev, uint32_t nslots, uint32_t max_wgs, uint32_t tiles_per_wg,
                                              uint32_t items_cap, uint32_t pair_cap, uint32_t pool_cap, hipStream_t stream)
{
    if (nslots == 0 || max_wgs == 0) return hipSuccess;
    /* the staged pool follows the pair list in dynamic LDS and is filled in 16-byte pieces: both sizes in multiples of 4 dwords
     * (hvq_recon_inline_dyn_lds tells the host what that makes) */
    pair_cap = (std::max(pair_cap, 1u) + 3u) & ~3u;
    pool_cap = (pool_cap + 3u) & ~3u;
    if (tiles_per_wg >= 2) {
        if (items_cap <= 32) launch_recon_inline<32, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 64) launch_recon_inline<64, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 96) launch_recon_inline<96, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 128) launch_recon_inline<128, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 192) launch_recon_inline<192, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 256) launch_recon_inline<256, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 384) launch_recon_inline<384, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else launch_recon_inline<512, 2>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
    } else {
        if (items_cap <= 32) launch_recon_inline<32, 1>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 64) launch_recon_inline<64, 1>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 96) launch_recon_inline<96, 1>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 128) launch_recon_inline<128, 1>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else if (items_cap <= 192) launch_recon_inline<192, 1>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
        else launch_recon_inline<256, 1>(jobs_dev, nslots, max_wgs, pair_cap, pool_cap, stream);
    }
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------------
 * Display epilogue: YUV 4:2:0 -> RGB24 exactly as the reference player's dumpRGB
 * (h4m:897-926): single-precision, one rounding per operation (the intrinsics below are never contracted
 * into FMAs), clamp, truncate.  Pure streaming kernel: 1.5 B/px read, 3 B/px written; one lane = 4 samples
 * of a row = one dword of Y in, three dwords of RGB out (a wave stores 768 contiguous bytes).
 */
/* clamp to [0, 255] and truncate (h4m:897-900), packed into byte `sel` of `acc`: v_floor_f32 + v_cvt_pk_u8_f32.  The pack
 * instruction saturates at both ends but rounds to nearest, the floor in front makes it exact (tools/ubench/cvt_probe.hip on
 * gfx950; the same arithmetic is proved equal to the reference's dumpRGB over all 2^24 (Y, U, V) triples in tests/test_rgb_exhaustive.py).
 * Replaces two compares, two selects, a convert, a shift and an or per sample (857 -> 434 VALU per lane of 32 samples). */
__device__ __forceinline__ u32 rgb_put(float f, u32 sel, u32 acc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(f), sel, acc);
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
        const float px[3] = { __fadd_rn(Y, __fmul_rn(1.402f, V)),
                              __fsub_rn(__fsub_rn(Y, __fmul_rn(0.34414f, U)), __fmul_rn(0.71414f, V)),
                              __fadd_rn(Y, __fmul_rn(1.772f, U)) };
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int byte = 3 * k + c;
            out[byte >> 2] = rgb_put(px[c], (u32)(byte & 3), out[byte >> 2]);
        }
    }
}

/* WIDE: one lane = 16 samples of TWO rows that share their chroma (two 16-byte Y loads, one 8-byte U and V load; the chroma
 * products are computed once for the 2x2 samples they serve).  A lane's 48 output bytes per row are contiguous, but stored
 * straight from the lane every store instruction would write 16 of every 48 bytes -- three partial passes over each line.
 * So a wave turns its 3 KB of a row around in LDS: chunk c (16 bytes) of the wave's output belongs to lane c / 3, and store
 * instruction j of lane l writes chunk l + 64 j -- every instruction a contiguous kilobyte (r03: 0.53 -> see DESIGN.md 8 f3).
 * Needs width % 16 == 0 and an even height (4:2:0 has both); otherwise 4 samples of one row per lane. */
template <bool WIDE>
__global__ __launch_bounds__(256)
void hvq_yuv420_rgb_kernel(const HvqRgbJob *__restrict__ jobs)
{
    const HvqRgbJob J = jobs[blockIdx.y];                    /* one picture per grid row */
    /* global-address-space pointers like the rest of the file: generic ones become flat_* accesses, which count on the LDS
     * counter too -- and this kernel turns its output around in LDS */
    const GLB uint8_t *__restrict__ yuv = (const GLB uint8_t *)J.yuv;
    GLB uint8_t *__restrict__ rgb = (GLB uint8_t *)J.rgb;
    const int w = J.w, h = J.h;
    constexpr int S = WIDE ? 16 : 4;
    const int qw = w / S;                                    /* lanes per row */
    const int total = qw * (WIDE ? h / 2 : h);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (WIDE) {
        __shared__ __attribute__((aligned(16))) u32 s_t[4][64 * 12];     /* per wave: 64 lanes x 48 bytes of one row */
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wave_idx0 = idx - lane;
        if (wave_idx0 >= total) return;                      /* whole wave beyond the picture (uniform) */
        const bool live = idx < total;
        const int li = live ? idx : total - 1;
        const int yr = li / qw, xq = li - yr * qw;
        const int y = 2 * yr;
        const GLB uint8_t *yp = yuv + (size_t)y * w + S * xq;
        const GLB uint8_t *up = yuv + (size_t)w * h + (size_t)(y >> 1) * (w >> 1) + (S / 2) * xq;
        const GLB uint8_t *vp = up + (size_t)(w >> 1) * (h >> 1);
        typedef u32 u32x4y __attribute__((ext_vector_type(4)));
        const u32x4y ya = *(const GLB u32x4y *)yp, yb = *(const GLB u32x4y *)(yp + w);
        const u32x2 u8 = *(const GLB u32x2 *)up, v8 = *(const GLB u32x2 *)vp;
        const u32 us[4] = { u8.x & 0xFFFFu, u8.x >> 16, u8.y & 0xFFFFu, u8.y >> 16 };
        const u32 vs[4] = { v8.x & 0xFFFFu, v8.x >> 16, v8.y & 0xFFFFu, v8.y >> 16 };
        /* where the three chunks this lane STORES live: chunk c = lane + 64 j belongs to lane c / 3 of the wave */
        size_t chunk_off[3];
        bool chunk_live[3];
        const float rqw = 1.0f / (float)qw;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = lane + 64 * j, owner = (c * 21846) >> 16, part = c - 3 * owner;       /* c / 3 for c < 192 */
            const int oi = wave_idx0 + owner;
            int oyr = (int)((float)oi * rqw);                                                   /* estimate within +-1, fixed up */
            int oxq = oi - oyr * qw;
            if (oxq < 0) { oxq += qw; --oyr; } else if (oxq >= qw) { oxq -= qw; ++oyr; }
            chunk_live[j] = oi < total;
            chunk_off[j] = ((size_t)(2 * oyr) * w + (size_t)S * oxq) * 3 + 16 * (size_t)part;
        }
        typedef u32 u32x4t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const u32x4y y16 = row ? yb : ya;
            const u32 ys[4] = { y16.x, y16.y, y16.z, y16.w };
            u32 o[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) rgb4(ys[q], us[q], vs[q], o + 3 * q);     /* the chroma terms are common subexpressions of the two rows */
            u32x4t *mine = (u32x4t *)&s_t[wave][lane * 12];
#pragma unroll
            for (int q = 0; q < 3; ++q) { const u32x4t v = { o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3] }; mine[q] = v; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4t v = *(const u32x4t *)&s_t[wave][(lane + 64 * j) * 4];
                if (chunk_live[j]) *(GLB u32x4t *)(rgb + chunk_off[j] + (size_t)row * 3 * w) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                 /* the second row overwrites the buffer */
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else {
        if (idx >= total) return;
        const int yr = idx / qw, xq = idx - yr * qw;
        const GLB uint8_t *yp = yuv + (size_t)yr * w + S * xq;
        const GLB uint8_t *up = yuv + (size_t)w * h + (size_t)(yr >> 1) * (w >> 1) + (S / 2) * xq;
        const GLB uint8_t *vp = up + (size_t)(w >> 1) * (h >> 1);
        GLB u32 *dst = (GLB u32 *)(rgb + ((size_t)yr * w + S * xq) * 3);
        u32 o[3];
        rgb4(*(const GLB u32 *)yp, *(const GLB uint16_t *)up, *(const GLB uint16_t *)vp, o);
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    }
}

/* bulk readback (hvq_read_pictures): `n` resident pictures gathered into one contiguous staging buffer, so that the copy to the
 * host is one large transfer instead of `n` small ones */
__global__ __launch_bounds__(256)
void hvq_gather_kernel(const uint64_t *__restrict__ src, uint8_t *__restrict__ dst, u32 pic_bytes)
{
    typedef u32 u32x4g __attribute__((ext_vector_type(4)));
    const GLB u32x4g *s = (const GLB u32x4g *)(uintptr_t)src[blockIdx.y];
    GLB u32x4g *d = (GLB u32x4g *)(dst + (size_t)blockIdx.y * pic_bytes);
    const u32 n16 = pic_bytes / 16u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) __builtin_nontemporal_store(s[i], d + i);
}

/* a table from pinned host memory into HBM, read over PCIe by the compute queue itself: stream-ordered without a DMA engine */
extern "C" __global__ __launch_bounds__(256)
void hvq_upload_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, u32 n16)
{
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}

/* the two table divisions of the kernels for every divisor (self-test of udiv_table on the device: tests/test_gpu_parity.py) */
__global__ void hvq_table_div_kernel(u32 *out)
{
    const u32 d = threadIdx.x;
    if (d < 16u) out[d] = udiv_table<256>(d);
    out[16u + d] = udiv_table<4096>(d);
}

extern "C" hipError_t hvq_launch_table_div(uint32_t *out_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(hvq_table_div_kernel, dim3(1), dim3(256), 0, stream, out_dev);
    return hipGetLastError();
}

extern "C" hipError_t hvq_launch_upload(const void *src_pinned, void *dst_dev, size_t bytes, hipStream_t stream)
{
    const u32 n16 = (u32)((bytes + 15u) / 16u);
    if (!n16) return hipSuccess;
    const u32 wgs = (n16 + 255u) / 256u;
    hipLaunchKernelGGL(hvq_upload_kernel, dim3(wgs < 256u ? wgs : 256u), dim3(256), 0, stream, (const uint4 *)src_pinned, (uint4 *)dst_dev, n16);
    return hipGetLastError();
}

extern "C" hipError_t hvq_launch_gather(const uint64_t *src_dev, uint8_t *dst_dev, uint32_t n, uint32_t pic_bytes, hipStream_t stream)
{
    if (!n) return hipSuccess;
    const uint32_t per = (pic_bytes / 16u + 255u) / 256u;
    hipLaunchKernelGGL(hvq_gather_kernel, dim3(per < 16u ? (per ? per : 1u) : 16u, n), dim3(256), 0, stream, src_dev, dst_dev, pic_bytes);
    return hipGetLastError();
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
