/*
 * hvq_gparse.hip -- gfx950 kernel around hvq_gparse_core.h / hvq_gparse_flat.h: one workgroup (4 wavefronts) parses one
 * picture's bitstream into its descriptor blob, entirely on the GPU.
 *
 * Two ways to the same blob (tests/native/gparse_emul.c runs both on the CPU, phase by phase):
 *
 * the FLAT path (round 2, default): every prefix-coded section is decoded front to back into a flat symbol
 * array, all sections at once -- the lanes of ONE wave are the sections -- and zero runs, overflow grouping, DC
 * prediction and coefficient sums become scans by all threads.
 *   all pictures : setup + section cursors (wave 0) | maps/MVs/tree tables cleared (all) | prefix trees read, one per wave |
 *                  the chains' 8-bit tables and the lanes' 16-bit-entry tables (10 bits block kinds, 9 bits the others) filled (all)
 *   I picture    : { decode wave: kinds, coefficients, DC runs | decode wave: the three DC sections | staging wave } |
 *                  5 scan rounds (values, zero runs -> blocks) | DC prediction as a wavefront, one wave per plane | nest, pool
 *                  layout (tile sums, one wave scan, entries) | payload positions (scans) | merge (entries on consecutive lanes)
 *   P/B picture  : { decode wave: all ten sections | staging wave | type runs then x vectors | proc runs then y vectors } |
 *                  tags + lists | 4 scan rounds | layout | payload positions | merge
 * A picture the flat path cannot serve (sections in an unusual order, array capacities, overflow groups at the chains' caps)
 * is handed back to the host (HvqParseResult.pad[0] = 2), which sends it to
 *
 * the CHAINS (round 1; HVQM4_AMD_PARSE_FLAT=0 runs them alone): the reference's loops cut into serial chains that own the
 * cursors they read, three chain phases with parallel phases between them.
 *   I picture    : 5 chains (kinds Y, kinds UV, DC Y, DC U, DC V) | nest + run sums (all) | run scan, header (wave 0) |
 *                  payload entries, fixed-length offsets, compaction (all) | 3 chains (coefficient symbols Y, U, V) | merge (all)
 *   P/B picture  : 2 chains (macroblock types; proc runs) | tags + lists (all, 5 steps) | 5 chains (kinds Y, kinds UV, DC Y,
 *                  DC U, DC V: symbols only) | kinds and DC values placed in the maps (all) | run sums | run scan, header |
 *                  payload entries, offsets, compaction | 4 waves: coefficient symbols Y; U, V + scalars Y; MV x + scalars U, V;
 *                  MV y | merge (all)
 * A chain runs wave-uniform (all lanes compute the same values, so its cursors and counters live in scalar registers and its
 * logic runs on the scalar unit); 8 workgroups share a CU, so the serial bit-level work of thousands of pictures overlaps.
 *
 * LDS per workgroup: the picture state (cursors, lanes, geometry), 24 staging slots of 128 B (the chains' bitstream blocks and
 * work lists; the flat path's rings of 256 B per lane lie over them), six prefix trees (8-bit table whose entries hold the leaf
 * value + child table: 2 KB each; the flat path's lane tables lie over four of the 8-bit tables), 2 KB behind them (10-bit block-kind table), the DC row buffers: 19.3 KB + 3 x (hb + 2), 8 workgroups
 * per CU.  At most 80 SGPRs (more costs a wave slot per SIMD on gfx950) and 64 VGPRs.
 */
#include <hip/hip_runtime.h>

#include "hvq_gparse_flat.h"

#define GPW 256

/* ------------------------------------------------------------------ flat path: wave-level pieces */
/* The decode wave and the staging wave talk through LDS: the decoder publishes each lane's bit position, the stager
 * keeps four 64-byte quarters of every lane's section in that lane's ring (64 dwords of gp_stage) and publishes how far
 * the data reaches.  The decoder never issues a load, so it never waits on the memory counter -- which on gfx950 it
 * could not do without also waiting for its own symbol stores. */
__shared__ uint32_t gf_pos_pub[16];
__shared__ uint32_t gf_avail_pub[16];
__shared__ uint32_t gf_done;
__shared__ uint32_t gf_types_done;
#ifdef GF_PROFILE
__shared__ uint32_t gf_slow_count;
#endif
/* relaxed LDS accesses the compiler may neither cache nor drop (a `volatile` LDS object becomes a flat access here) */
#define GF_LD(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GF_ST(x, v) __hip_atomic_store(&(x), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

#define GF_RING 64u
#define GF_AHEAD 640u         /* bits a round of eight codes normally touches, with room for a long one (the stager keeps
                                 at least 1537 ahead of the published position) */
#define GF_AHEAD_SLOW 384u    /* one code of any length (at most 9 + 255 bits) and its window */
#define GF_SPIN_CAP (1u << 22)

/* ring of lane l: lanes 0-9 over the staging slots, lanes 10-12 (I pictures) inside the tree an I picture lacks */
__device__ static inline uint32_t *gf_ring(int l)
{
    return l < 10 ? gp_stage + GF_RING * (uint32_t)l : gp_stage + ((GP_TREE_DWORD(GC_MV) + 8u) & ~7u) + GF_RING * (uint32_t)(l - 10);
}
#define GF_SIDE_SLOT 20u                      /* P/B: slots 20-23 for the chains that run beside the rings */

__device__ static inline uint32_t gf_ring_dword(const uint32_t *ring, uint32_t i) { return __builtin_bswap32(ring[i & (GF_RING - 1u)]); }

__device__ static inline uint64_t gf_ring_window64(const uint32_t *ring, uint32_t pos)       /* 64 valid bits */
{
    const uint32_t i = pos >> 5, sh = pos & 31u;
    const uint32_t w0 = gf_ring_dword(ring, i), w1 = gf_ring_dword(ring, i + 1), w2 = gf_ring_dword(ring, i + 2);
    return ((((uint64_t)w0 << 32) | w1) << sh & 0xFFFFFFFF00000000ull) | (((((uint64_t)w1 << 32) | w2) << sh) >> 32);
}

/* four symbols (leaf bytes) from a 64-bit window at `pos` (4 x 10 table bits fit it).  A code longer than the lane's table
 * is rare per lane but not per wave, so it is finished on the spot: the tree is walked bit by bit from the node the table
 * gave, the window is taken anew behind it, the other lanes wait for that one symbol only.  `a` = how far the lane's ring is
 * known to reach. */
__device__ static inline uint32_t gf_four(const uint16_t *kid, const uint16_t *tab, int bits, const uint32_t *ring, int l,
                                          uint32_t *ppos, uint32_t *pa, uint32_t nbits, uint32_t *guard)
{
    uint32_t pos = *ppos;
    uint64_t w = gf_ring_window64(ring, pos);
    const int down = 64 - bits;
    uint32_t sy[4];
    {   /* all four straight from the table, no lane looking left or right: with the tables of round 4 four groups in five end
         * here.  (An entry that is no leaf consumes its table bits; what is looked up behind it is not used.) */
        const uint32_t e0 = tab[w >> down];
        const uint64_t w1 = w << (e0 & 63u);
        const uint32_t e1 = tab[w1 >> down];
        const uint64_t w2 = w1 << (e1 & 63u);
        const uint32_t e2 = tab[w2 >> down];
        const uint64_t w3 = w2 << (e2 & 63u);
        const uint32_t e3 = tab[w3 >> down];
        if (!__builtin_amdgcn_ballot_w64(((e0 & e1 & e2 & e3) & 0x80u) == 0)) {
            *ppos = pos + (e0 & 63u) + (e1 & 63u) + (e2 & 63u) + (e3 & 63u);
            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(e3, e2, 0x0C0C0501u), __builtin_amdgcn_perm(e1, e0, 0x0C0C0501u), 0x05040100u);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t e = tab[w >> down];
        const uint32_t len = e & 63u;
        pos += len;
        sy[k] = e;
        if (__builtin_expect((e & 0x80u) != 0, 1)) { w <<= len; continue; }
        if (*pa < pos + GF_AHEAD_SLOW) {                    /* the walk may run up to 255 bits further */
            GF_ST(gf_pos_pub[l], pos);
            while ((*pa = GF_LD(gf_avail_pub[l])) < pos + GF_AHEAD_SLOW && ++*guard <= GF_SPIN_CAP) __builtin_amdgcn_s_sleep(1);
        }
        int id = (int)(e >> 8) + 256;
        w = gf_ring_window64(ring, pos);                   /* the bits behind the table's part of the code */
        for (uint32_t left = 64; id >= 256; --left) {
            if (left == 0) { w = gf_ring_window64(ring, pos); left = 64; }
            const uint32_t bit = pos < nbits ? (uint32_t)(w >> 63) : 0u;
            id = kid[bit * 256u + (uint32_t)id - 256u];
            w <<= 1; ++pos;
        }
        sy[k] = (uint32_t)id << 8;
#ifdef GF_PROFILE
        gf_slow_count++;
#endif
    }
    *ppos = pos;
    /* byte 1 of the four entries */
    return __builtin_amdgcn_perm(__builtin_amdgcn_perm(sy[3], sy[2], 0x0C0C0501u), __builtin_amdgcn_perm(sy[1], sy[0], 0x0C0C0501u), 0x05040100u);
}

/* the lanes of this wave decode one section each, in lockstep (gf_decode_lane is the same loop for one lane): eight
 * symbols per round */
__device__ static void gf_decode_wave(GPic *g, const GCode *codes, int lane, uint32_t lanes)
{
    const int nl = g->is_pb ? GF_RLE0 : GF_LANES;
    const int l = lane < nl ? lane : 0;
    GLane *q = &g->lane[l];
    const GCode *c = &codes[q->tree];
    const bool mine = lane < nl && ((lanes >> lane) & 1u) && c->root >= 256 && !g->status;
    const uint32_t end = mine ? q->end : 0u, cap = q->cap, nbits = g->nd * 32u;
    uint32_t pos = q->pos, n = 0, guard = 0;
    GP_G uint8_t *out = (GP_G uint8_t *)(g->sym + q->off);
    const uint32_t *ring = gf_ring(l);
    const uint16_t *tab = gf_lane_table(codes, (int)q->tree), *kid = gf_lane_kids(codes, (int)q->tree);
    const int bits = gf_lane_bits((int)q->tree);
#ifdef GF_PROFILE
    uint64_t ta = 0, tb = 0, rounds = 0;
#define GF_T(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define GF_T(x)
#endif
    for (;;) {
        GF_T(t0);
        const bool act = pos < end && n + 8 <= cap;
        if (!__builtin_amdgcn_ballot_w64(act)) break;
        if (act) GF_ST(gf_pos_pub[l], pos);
        uint32_t a;
        for (;;) {                                                   /* until the stager is far enough ahead of every lane */
            a = GF_LD(gf_avail_pub[l]);
            if (!__builtin_amdgcn_ballot_w64(act && a < pos + GF_AHEAD)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++guard > GF_SPIN_CAP) break;
        }
        /* the staging wave did not deliver in time: a scheduling matter, not a malformed stream -- the chains kernel takes the
         * picture (same blob), exactly as for anything else the flat path cannot serve */
        if (guard > GF_SPIN_CAP) { g->retry = 1; break; }
        GF_T(t1);
        if (act) {
            const uint32_t lo = gf_four(kid, tab, bits, ring, l, &pos, &a, nbits, &guard);
            const uint32_t hi = gf_four(kid, tab, bits, ring, l, &pos, &a, nbits, &guard);
            *(GP_G uint2 *)(out + n) = make_uint2(lo, hi);
            n += 8;
        }
#ifdef GF_PROFILE
        { GF_T(t2); ta += t1 - t0; tb += t2 - t1; ++rounds; }
#endif
    }
#ifdef GF_PROFILE
    if (lane == 0) { g->prof[0] = (uint32_t)ta; g->prof[1] = (uint32_t)tb; g->prof[2] = (uint32_t)rounds | (gf_slow_count << 16); }
#endif
    if (mine) q->n = n;
    if (lane == 0) { g->spins = guard; __hip_atomic_fetch_add(&gf_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
}

/* the staging wave: lane l owns ring l; every round fetches ALL quarters that are missing behind any lane (up to four
 * per ring), waits for them once and publishes */
__device__ static void gf_stage_wave(GPic *g, int lane, uint32_t decoders)
{
    const int nl = g->is_pb ? GF_RLE0 : GF_LANES;
    const bool own = lane < nl && !g->status && g->nd > 0;
    uint32_t have = own ? g->lane[lane < nl ? lane : 0].pos >> 9 : 0u;     /* next quarter (16 dwords) to fetch */
    const GP_G uint32_t *d = g->d;
    const uint32_t last = g->nd ? g->nd - 1u : 0u;
    uint32_t idle = 0;
    for (;;) {
        const uint32_t done = GF_LD(gf_done) >= decoders;
        const uint32_t p = GF_LD(gf_pos_pub[lane & 15]);
        const uint32_t want = own ? (p >> 9) + 4u : 0u;                  /* quarters up to here may be resident */
        const bool need = own && have < want;
        uint64_t mask = __builtin_amdgcn_ballot_w64(need);
        if (!mask) {
            if (done || ++idle > GF_SPIN_CAP) break;
            /* a lane uses up a quarter in 8 rounds of its decoder, some 8 us; looking every 1-2 us is plenty, and every
             * look costs the CU two dozen instructions */
            __builtin_amdgcn_s_sleep(127);
            continue;
        }
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)have, l), k1 = (uint32_t)__builtin_amdgcn_readlane((int)want, l);
            for (uint32_t k = k0; k < k1 && k < k0 + 4u; ++k) {
                if (lane < 16) {
                    uint32_t idx = 16u * k + (uint32_t)lane;
                    if (idx > last) idx = last;
                    __builtin_amdgcn_global_load_lds(d + idx, (__attribute__((address_space(3))) uint32_t *)(gf_ring(l) + 16u * (k & 3u)), 4, 0, 0);
                }
            }
        }
        if (need) have = want < have + 4u ? want : have + 4u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (need) GF_ST(gf_avail_pub[lane], have << 9);
    }
}

/* exclusive scan of the 256 chunk partials of `inst` by one wave (four per lane); total -> g->tot[inst] */
__device__ static inline uint32_t gf_wave_incl_add(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64); if (lane >= d) v += o; }
    return v;
}

__device__ static void gfd_scan_add(GPic *g, int inst, int lane, uint32_t *also)
{
    GP_G uint4 *p = (GP_G uint4 *)(g->part + GF_P(inst, 0)) + lane;
    uint4 v = *p;
    const uint32_t s = v.x + v.y + v.z + v.w;
    const uint32_t incl = gf_wave_incl_add(s, lane);
    uint32_t run = incl - s;
    uint4 o;
    o.x = run; run += v.x; o.y = run; run += v.y; o.z = run; run += v.z; o.w = run;
    *p = o;
    if (lane == 63) { if (inst >= 16) g->tot[inst - 16] = incl; if (also) *also = incl; }
}

__device__ static void gfd_scan_seg(GPic *g, int inst_flag, int inst_val, int lane)
{
    GP_G uint4 *pf = (GP_G uint4 *)(g->part + GF_P(inst_flag, 0)) + lane, *pv = (GP_G uint4 *)(g->part + GF_P(inst_val, 0)) + lane;
    const uint4 f = *pf, v = *pv;
    const uint32_t fs[4] = { f.x, f.y, f.z, f.w }, vs[4] = { v.x, v.y, v.z, v.w };
    uint32_t F = 0, V = 0;                                            /* this lane's four chunks as one */
#pragma unroll
    for (int k = 0; k < 4; ++k) { V = fs[k] ? vs[k] : V + vs[k]; F |= fs[k]; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {                                /* inclusive over the lanes */
        const uint32_t oF = (uint32_t)__shfl_up((int)F, d, 64), oV = (uint32_t)__shfl_up((int)V, d, 64);
        if (lane >= d) { V = F ? V : oV + V; F |= oF; }
    }
    uint32_t run = (uint32_t)__shfl_up((int)V, 1, 64);
    if (lane == 0) run = 0;
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = run; run = fs[k] ? vs[k] : run + vs[k]; }
    *pv = make_uint4(o[0], o[1], o[2], o[3]);
}

/* wave64 inclusive prefix sum on the DPP network (all lanes active) */
__device__ static inline uint32_t gfd_dpp_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

__device__ static inline uint32_t gfd_wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)gfd_dpp_incl(v), 63); }

__device__ static inline uint32_t gfd_below(uint64_t mask)      /* set bits of `mask` below this lane */
{
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__shared__ uint32_t gfd_sh[64];                 /* per-wave partials of the wave-cooperative passes; 0-39 one pass at a time, 40-63 the intra-DC
                                                   pass, which starts in the same barrier interval as the last expansion pass reads its own */

/* gf_emit_merge on the device.  A wave owns the entries of its 64 threads' chunks (whose starting offsets gf_emit_count + the
 * scans have already produced), walks them 64 at a time and gets every entry's position in the coefficient symbols, in the
 * fixed-length section and among the MC-residual blocks from three wave scans.  Most entries carry nothing (a dense 640x480
 * B picture: 10 000 coefficient symbols over 19 200 luma blocks), so the entries that do are QUEUED -- (entry, symbol index,
 * fixed-length offset, MC-residual rank) in the wave's 128-record ring in LDS, over the staging slots and trees nobody needs
 * any more -- and completed 64 at a time with every lane busy: round 4, a third of the vector instructions of the
 * entry-per-lane form (which was the largest single consumer of vector instructions in the kernel). */
#define GFD_MQ 128u                                                /* records per wave: 4 arrays of 128 dwords */

__device__ static inline void gfd_merge_one(const GPic *g, GP_G uint32_t *pool, const GP_G uint8_t *S, const GP_G uint32_t *V,
                                            int sh_dc, int sh_unk, uint32_t ent, uint32_t my_si, uint64_t my_fx, uint32_t my_pi)
{
    const uint32_t mode = ent >> 30;
    GP_G uint32_t *dst = pool + (ent & 0x3FFFFFu);
    if (mode == GP_MODE_LITERAL) {
        uint32_t v[4];
        for (int k = 0; k < 4; ++k) v[k] = __builtin_bswap32(gp_be32(g, my_fx + 4u * (uint32_t)k));
        for (int k = 0; k < 4; ++k) dst[k] = v[k];
        return;
    }
    const uint32_t nb = (ent >> 22) & 0xFFu;
    /* everything the entry needs is requested before its first store: the two scalars of an MC-residual block, the first eight
     * basis words and symbols (a load behind a store waits for the store's acknowledgement as well) */
    int32_t s1 = 0, s2 = 0;
    if (mode == GP_MODE_PREDI) { s1 = (int32_t)V[2u * my_pi]; s2 = (int32_t)V[2u * my_pi + 1u]; }
    uint32_t wq[4] = { 0, 0, 0, 0 }, sq[2] = { 0, 0 };
    if (nb) {
        if (__builtin_expect(my_fx + 16u <= (uint64_t)g->nd * 4u, 1)) __builtin_memcpy(wq, (const GP_G uint8_t *)g->d + my_fx, 16);
        else for (uint32_t j = 0; j < 4; ++j) wq[j] = __builtin_bswap32((gp_be16(g, my_fx + 4u * j) << 16) | gp_be16(g, my_fx + 4u * j + 2u));
        __builtin_memcpy(sq, S + my_si, 8);
    }
    if (mode == GP_MODE_PREDI) {
        dst[0] = (uint32_t)(s1 >> sh_dc) << sh_unk;
        dst[1] = (uint32_t)(s2 >> sh_dc);
        dst += 2;
    }
    uint32_t run = 0;
    /* eight bases a time from a 16-byte and an 8-byte load (words and symbols -- leaf bytes of the coefficient tree, value =
     * byte << 2 -- lie consecutive; both arrays and the picture are padded far enough to read a full block) */
    for (uint32_t k0 = 0; k0 < nb; k0 += 8) {
        if (k0) {
            const uint64_t wo = my_fx + 2u * k0;
            if (__builtin_expect(wo + 16u <= (uint64_t)g->nd * 4u, 1)) __builtin_memcpy(wq, (const GP_G uint8_t *)g->d + wo, 16);
            else for (uint32_t j = 0; j < 4; ++j) wq[j] = __builtin_bswap32((gp_be16(g, wo + 4u * j) << 16) | gp_be16(g, wo + 4u * j + 2u));
            __builtin_memcpy(sq, S + my_si + k0, 8);
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            if (k0 + j >= nb) break;
            const uint32_t half = (wq[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu;           /* bytes b0 b1 of the word, b0 low */
            const uint32_t w = ((half & 0xFFu) << 8) | (half >> 8);
            run += ((sq[j >> 2] >> (8u * (j & 3u))) & 0xFFu) << 2;
            dst[k0 + j] = HVQ_BASIS(w, (run + ((w >> 13) & 3u)) & 0x3FFFFu);
        }
    }
}

/* a wave-uniform 64-bit value (a pointer read from the picture state in LDS) into scalar registers: the kernel has 64 vector registers */
__device__ static inline uint64_t gfd_uni64(uint64_t v) { return ((uint64_t)GP_UNI(v >> 32) << 32) | GP_UNI((uint32_t)v); }
#define GFD_UNIP(type, p) ((type)gfd_uni64((uint64_t)(p)))

__device__ static void gfd_emit_merge(GPic *g, int tid)
{
    if (g->status || g->retry) return;
    if (gf_emit_short(g)) { g->retry = 1; return; }
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    GP_G uint32_t *pool = GFD_UNIP(GP_G uint32_t *, g->blob + g->fixed_bytes);
    const int sh_dc = (int)GP_UNI(g->dc_shift & 31), sh_unk = (int)GP_UNI(g->unk_shift & 31);
    uint32_t *q_ent = gp_stage + 4u * GFD_MQ * (uint32_t)wave, *q_si = q_ent + GFD_MQ, *q_fx = q_si + GFD_MQ, *q_pi = q_fx + GFD_MQ;
    for (int i = 0; i < 3; ++i) {
        const GP_G uint32_t *ents = GFD_UNIP(const GP_G uint32_t *, g->pinfo + g->pl[i].blk_first);
        const GP_G uint8_t *S = GFD_UNIP(const GP_G uint8_t *, g->sym + g->lane[GF_BT0 + i].off);
        const GP_G uint32_t *V = GFD_UNIP(const GP_G uint32_t *, g->val + g->val_off[i] + g->ntype0 * (uint32_t)g->pl[i].nblk);
        const uint32_t n = GP_UNI(g->pl[i].nblocks), per = (n + GPW - 1) / GPW;
        const uint32_t e0 = per * 64u * (uint32_t)wave < n ? per * 64u * (uint32_t)wave : n;
        const uint32_t e1 = per * 64u * (uint32_t)(wave + 1) < n ? per * 64u * (uint32_t)(wave + 1) : n;
        uint64_t fx = gfd_uni64(g->fx_off[i] + g->part[GF_P(GF_I_FX(i), 64 * wave)]);
        uint32_t si = GP_UNI(g->part[GF_P(GF_I_NB(i), 64 * wave)]), pi = GP_UNI(g->part[GF_P(GF_I_PREDI(i), 64 * wave)]);
        uint32_t head = 0, tail = 0;                                  /* records taken / put so far (wave-uniform) */
        uint32_t ent_next = e0 + (uint32_t)lane < e1 ? ents[e0 + (uint32_t)lane] : 0u;
        for (uint32_t eb = e0; eb < e1 || tail != head; eb += 64) {
            if (eb < e1) {
                /* the entries of the next step are requested before this step's records are worked on */
                const uint32_t ent = ent_next, mode = ent >> 30;
                ent_next = eb + 64u + (uint32_t)lane < e1 ? ents[eb + 64u + (uint32_t)lane] : 0u;
                const uint32_t nb = mode >= GP_MODE_BASES ? (ent >> 22) & 0xFFu : 0u, fb = gp_ent_fx_bytes(ent), ip = mode == GP_MODE_PREDI;
                const uint32_t s_nb = gfd_dpp_incl(nb), s_fb = gfd_dpp_incl(fb), s_ip = gfd_dpp_incl(ip);
                const uint64_t my_fx = fx + (s_fb - fb);
                const bool work = mode != GP_MODE_NONE;
                const uint64_t mask = __builtin_amdgcn_ballot_w64(work);
                if (work) {
                    const uint32_t slot = (tail + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))) & (GFD_MQ - 1u);
                    q_ent[slot] = ent; q_si[slot] = si + s_nb - nb; q_pi[slot] = pi + s_ip - ip;
                    q_fx[slot] = my_fx > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)my_fx;   /* past 4 GiB there are only zeros to read either way */
                }
                tail += (uint32_t)__builtin_popcountll(mask);
                si += (uint32_t)__builtin_amdgcn_readlane((int)s_nb, 63);
                fx += (uint32_t)__builtin_amdgcn_readlane((int)s_fb, 63);
                pi += (uint32_t)__builtin_amdgcn_readlane((int)s_ip, 63);
            }
            /* a full wave of records -- or, behind the plane's last entries, what is left */
            const uint32_t have = tail - head, take = have >= 64u ? 64u : (eb + 64u >= e1 ? have : 0u);
            if ((uint32_t)lane < take) {
                const uint32_t slot = (head + (uint32_t)lane) & (GFD_MQ - 1u);
                gfd_merge_one(g, pool, S, V, sh_dc, sh_unk, q_ent[slot], q_si[slot], q_fx[slot], q_pi[slot]);
            }
            head += take;
        }
    }
}

/* gf_exp_zeros / gf_exp_lens / gf_exp_write with consecutive tokens on consecutive lanes (a wave owns a quarter of an expansion's
 * tokens): the rank of a zero token among the zeros comes from a ballot, its run length from the neighbour of the previous
 * zero's, the block a token lands on from one DPP scan of the lengths, and what the 64 lanes then read (macroblock, tag) and
 * write (map bytes) lies side by side.  The thread-per-chunk passes they replace made every lane of a load or store touch a
 * cache line of its own, and the 256-entry scans between them are 4-entry sums here.  gfd_ex(x, which, wave): zeros / blocks
 * covered of expansion x in the wave's tokens. */
#define gfd_ex(x, which, w) gfd_sh[8 * (x) + 4 * (which) + (w)]

__device__ static inline uint32_t gfd_exp_token(const GExp *e, uint32_t j) { return e->t8 ? (uint32_t)e->t8[j] : e->t32[j]; }

__device__ static inline void gfd_exp_range(const GExp *e, int wave, uint32_t *lo, uint32_t *hi)
{
    const uint32_t per = ((e->ntok + GPW - 1) / GPW) * 64u;
    *lo = per * (uint32_t)wave < e->ntok ? per * (uint32_t)wave : e->ntok;
    *hi = *lo + per < e->ntok ? *lo + per : e->ntok;
}

__device__ static void gfd_exp_zeros(GPic *g, int x0, int x1, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    for (int x = x0; x < x1; ++x) {
        GExp e;
        uint32_t lo, hi, z = 0;
        if (!gf_exp(g, x, &e)) continue;
        gfd_exp_range(&e, wave, &lo, &hi);
        for (uint32_t jb = lo; jb < hi; jb += 64) {
            const uint32_t j = jb + (uint32_t)lane;
            z += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(j < hi && gfd_exp_token(&e, j) == 0));
        }
        if (lane == 0) gfd_ex(x, 0, wave) = z;
    }
}

/* the tokens jb .. jb + 63 of the wave: token, whether it takes part, its length (1 + the run behind a zero token) */
__device__ static inline uint32_t gfd_exp_step(const GExp *e, uint32_t jb, uint32_t hi, int lane, uint32_t *z, uint32_t *tok, bool *live, bool *short_runs)
{
    const uint32_t j = jb + (uint32_t)lane;
    *live = j < hi;
    *tok = *live ? gfd_exp_token(e, j) : 1u;
    const bool zero = *live && *tok == 0;
    const uint64_t bz = __builtin_amdgcn_ballot_w64(zero);
    const uint32_t zr = *z + gfd_below(bz);
    *z += (uint32_t)__builtin_popcountll(bz);
    *short_runs = zero && zr >= e->nrun;                 /* a zero token without a run length */
    return *live ? 1u + (zero && zr < e->nrun ? (uint32_t)e->run[zr] : 0u) : 0u;
}

__device__ static void gfd_exp_lens(GPic *g, int x0, int x1, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    for (int x = x0; x < x1; ++x) {
        GExp e;
        uint32_t lo, hi, z = 0, len = 0;
        if (!gf_exp(g, x, &e)) continue;
        gfd_exp_range(&e, wave, &lo, &hi);
        for (int w = 0; w < wave; ++w) z += gfd_ex(x, 0, w);
        for (uint32_t jb = lo; jb < hi; jb += 64) {
            uint32_t tok; bool live, bad;
            len += gfd_wave_sum(gfd_exp_step(&e, jb, hi, lane, &z, &tok, &live, &bad));
        }
        if (lane == 0) gfd_ex(x, 1, wave) = len;
    }
}

__device__ static void gfd_exp_write(GPic *g, int x0, int x1, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    for (int x = x0; x < x1; ++x) {
        GExp e;
        uint32_t lo, hi, z = 0, at = 0;
        if (!gf_exp(g, x, &e)) continue;
        if (gfd_ex(x, 1, 0) + gfd_ex(x, 1, 1) + gfd_ex(x, 1, 2) + gfd_ex(x, 1, 3) < e.N) { g->retry = 1; continue; }       /* not enough tokens */
        gfd_exp_range(&e, wave, &lo, &hi);
        for (int w = 0; w < wave; ++w) { z += gfd_ex(x, 0, w); at += gfd_ex(x, 1, w); }
        for (uint32_t jb = lo; jb < hi && at < e.N; jb += 64) {
            uint32_t tok; bool live, bad;
            const uint32_t len = gfd_exp_step(&e, jb, hi, lane, &z, &tok, &live, &bad);
            const uint32_t incl = gfd_dpp_incl(len), mine = at + incl - len;
            at += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (live && mine < e.N) {                      /* tokens are consumed until the blocks are covered */
                if (bad) g->retry = 1;                       /* not enough run lengths */
                else if (tok) gf_exp_put(g, x, mine, tok);
            }
        }
    }
}

/* gf_pbdc_sums / gf_pbdc_write (intra DC of a P/B picture, h4m:1742-1776) with consecutive values on consecutive lanes: value e of
 * plane i belongs to block e % nblk of the intra macroblock t0[e / nblk]; the DC accumulates from 0x7F within a run of consecutive
 * intra macroblocks, i.e. it is a segmented running sum over the values with a segment head at the first block of every run.
 * gfd_pb(i, which, wave): does the wave's range hold a head / what has accumulated behind its last head (or over all of it). */
#define gfd_pb(i, which, w) gfd_sh[40 + 8 * (i) + 4 * (which) + (w)]

__device__ static inline void gfd_pbdc_step(const GPic *g, const GP_G uint32_t *V, const GP_G uint32_t *t0, uint32_t nblk, uint32_t e, uint32_t hi,
                                            uint32_t *val, bool *head, uint32_t *mb, uint32_t *j)
{
    const bool live = e < hi;
    const uint32_t r = nblk == 4 ? e >> 2 : (nblk == 1 ? e : e / nblk);
    *j = e - r * nblk;
    *val = live ? V[e] : 0u;
    *mb = live ? t0[r] : 0u;
    *head = live && *j == 0 && (r == 0 || *mb != t0[r - 1] + 1u);
    (void)g;
}

__device__ static void gfd_pbdc_sums(GPic *g, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    for (int i = 0; i < 3; ++i) {
        const uint32_t nblk = (uint32_t)g->pl[i].nblk, n = GP_UNI(g->ntype0) * nblk;
        if (n > g->nv[i]) { g->retry = 1; continue; }
        const GP_G uint32_t *V = GFD_UNIP(const GP_G uint32_t *, g->val + g->val_off[i]), *t0 = GFD_UNIP(const GP_G uint32_t *, g->t0);
        const uint32_t per = ((n + GPW - 1) / GPW) * 64u, lo = per * (uint32_t)wave < n ? per * (uint32_t)wave : n, hi = lo + per < n ? lo + per : n;
        uint32_t has = 0, tail = 0;
        for (uint32_t eb = lo; eb < hi; eb += 64) {
            uint32_t v, mb, j; bool head;
            gfd_pbdc_step(g, V, t0, nblk, eb + (uint32_t)lane, hi, &v, &head, &mb, &j);
            const uint64_t bh = __builtin_amdgcn_ballot_w64(head);
            if (bh) {                                            /* what follows the step's last head, that head's value included */
                const int last = 63 - __builtin_clzll(bh);
                tail = gfd_wave_sum(lane >= last ? v : 0u);
                has = 1;
            } else tail += gfd_wave_sum(v);
        }
        if (lane == 0) { gfd_pb(i, 0, wave) = has; gfd_pb(i, 1, wave) = tail; }
    }
}

__device__ static void gfd_pbdc_write(GPic *g, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    const uint32_t mw = GP_UNI(g->mw);
    for (int i = 0; i < 3; ++i) {
        const GPlane *q = &g->pl[i];
        const uint32_t nblk = (uint32_t)q->nblk, n = GP_UNI(g->ntype0) * nblk;
        const GP_G uint32_t *V = GFD_UNIP(const GP_G uint32_t *, g->val + g->val_off[i]), *t0 = GFD_UNIP(const GP_G uint32_t *, g->t0);
        const uint32_t per = ((n + GPW - 1) / GPW) * 64u, lo = per * (uint32_t)wave < n ? per * (uint32_t)wave : n, hi = lo + per < n ? lo + per : n;
        uint32_t carry = 0x7F;                                   /* the DC entering this wave's first value */
        for (int w = 0; w < wave; ++w) carry = gfd_pb(i, 0, w) ? 0x7Fu + gfd_pb(i, 1, w) : carry + gfd_pb(i, 1, w);
        for (uint32_t eb = lo; eb < hi; eb += 64) {
            uint32_t v, mb, j; bool head;
            gfd_pbdc_step(g, V, t0, nblk, eb + (uint32_t)lane, hi, &v, &head, &mb, &j);
            const uint64_t bh = __builtin_amdgcn_ballot_w64(head);
            const uint32_t incl = gfd_dpp_incl(v), excl = incl - v;
            /* the last head at or before this lane: the sum restarts at 0x7F there */
            const uint64_t mine = bh & ((2ull << lane) - 1ull);
            const int hl = mine ? 63 - __builtin_clzll(mine) : 0;
            const uint32_t before = (uint32_t)__shfl((int)excl, hl, 64);
            const uint32_t acc = mine ? 0x7Fu + incl - before : carry + incl;
            if (eb + (uint32_t)lane < hi) {
                const int my = (int)(mb / mw), mx = (int)(mb - (uint32_t)my * mw);
                gp_map_ent(g, i, my * q->by_per + gp_dy((int)j), mx * q->bx_per + gp_dx((int)j))[0] = (uint8_t)acc;
            }
            carry = (uint32_t)__builtin_amdgcn_readlane((int)acc, 63);
        }
    }
}

/* gf_emit_count + its nine scans for gfd_emit_merge: what lies in front of every WAVE's entries (fixed-length bytes, coefficient
 * symbols, MC-residual blocks per plane) and the totals gf_emit_short asks for; consecutive entries on consecutive lanes */
#define gfd_ec(k, w) gfd_sh[4 * (k) + (w)]


__device__ static void gfd_emit_count(GPic *g, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    for (int i = 0; i < 3; ++i) {
        const GP_G uint32_t *ents = GFD_UNIP(const GP_G uint32_t *, g->pinfo + g->pl[i].blk_first);
        const uint32_t n = GP_UNI(g->pl[i].nblocks), per = (n + GPW - 1) / GPW;
        const uint32_t e0 = per * 64u * (uint32_t)wave < n ? per * 64u * (uint32_t)wave : n;
        const uint32_t e1 = per * 64u * (uint32_t)(wave + 1) < n ? per * 64u * (uint32_t)(wave + 1) : n;
        uint32_t bytes = 0, nb = 0, np = 0;
        for (uint32_t e = e0 + (uint32_t)lane; e < e1; e += 64) {
            const uint32_t ent = ents[e], mode = ent >> 30;
            bytes += gp_ent_fx_bytes(ent);
            if (mode >= GP_MODE_BASES) nb += (ent >> 22) & 0xFFu;
            np += mode == GP_MODE_PREDI;
        }
        const uint32_t sb = gfd_wave_sum(bytes), sn = gfd_wave_sum(nb), sp = gfd_wave_sum(np);
        if (lane == 0) { gfd_ec(3 * i, wave) = sb; gfd_ec(3 * i + 1, wave) = sn; gfd_ec(3 * i + 2, wave) = sp; }
    }
    __syncthreads();
    if (tid < 9) {                                      /* instance tid: exclusive prefix over the waves, total */
        const int i = tid / 3, what = tid - 3 * i;
        const int inst = what == 0 ? GF_I_FX(i) : (what == 1 ? GF_I_NB(i) : GF_I_PREDI(i));
        uint32_t run = 0;
        for (int w = 0; w < 4; ++w) { g->part[GF_P(inst, 64 * w)] = run; run += gfd_ec(tid, w); }
        if (inst >= 16) g->tot[inst - 16] = run;
    }
}

/* gp_tags_count .. gp_lists_write of a P/B picture with consecutive macroblocks on consecutive lanes (a wave owns a quarter of the
 * picture's macroblocks): tag of every macroblock into the type byte of all its blocks, the lists of coded and of intra
 * macroblocks.  Ranks come from ballots, so the three passes are loads of neighbouring bytes and stores to neighbouring
 * entries; the thread-per-chunk form made every lane of a load touch a cache line of its own. */
#define gfd_tl gfd_sh


__device__ static void gfd_tags_lists(GPic *g, int tid)
{
    if (g->status) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    const uint32_t mw = GP_UNI(g->mw), n = mw * GP_UNI(g->mh), per = ((n + GPW - 1) / GPW) * 64u;
    const uint32_t m0 = per * (uint32_t)wave < n ? per * (uint32_t)wave : n, m1 = m0 + per < n ? m0 + per : n;
    const GP_G uint8_t *mbtype = GFD_UNIP(const GP_G uint8_t *, g->mbtype), *procseq = GFD_UNIP(const GP_G uint8_t *, g->procseq);
    {   /* inter macroblocks of the waves before this one */
        uint32_t cnt = 0;
        for (uint32_t mb = m0; mb < m1; mb += 64) {
            const uint32_t m = mb + (uint32_t)lane;
            cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(m < m1 && mbtype[m] != 0));
        }
        if (lane == 0) gfd_tl[wave] = cnt;
    }
    __syncthreads();
    uint32_t rank0 = 0;
    for (int w = 0; w < wave; ++w) rank0 += gfd_tl[w];
    {   /* coded and intra macroblocks of the waves before this one */
        uint32_t running = rank0, ncod = 0, nt0 = 0;
        for (uint32_t mb = m0; mb < m1; mb += 64) {
            const uint32_t m = mb + (uint32_t)lane;
            const bool live = m < m1;
            const uint32_t type = live ? mbtype[m] : 0u;
            const uint64_t bi = __builtin_amdgcn_ballot_w64(type != 0);
            const uint32_t proc = type ? procseq[running + gfd_below(bi)] : 0u;
            running += (uint32_t)__builtin_popcountll(bi);
            ncod += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live && !proc));
            nt0 += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(live && !type));
        }
        if (lane == 0) { gfd_tl[4 + wave] = ncod; gfd_tl[8 + wave] = nt0; }
    }
    __syncthreads();
    uint32_t a = 0, b = 0;
    for (int w = 0; w < wave; ++w) { a += gfd_tl[4 + w]; b += gfd_tl[8 + w]; }
    if (tid == 0) { g->ncoded = gfd_tl[4] + gfd_tl[5] + gfd_tl[6] + gfd_tl[7]; g->ntype0 = gfd_tl[8] + gfd_tl[9] + gfd_tl[10] + gfd_tl[11]; }
    GP_G uint8_t *mbtag = GFD_UNIP(GP_G uint8_t *, g->mbtag);
    GP_G uint32_t *cmb = GFD_UNIP(GP_G uint32_t *, g->cmb), *t0 = GFD_UNIP(GP_G uint32_t *, g->t0);
    const bool is_P = g->is_P != 0;
    uint32_t running = rank0, fl = 0;
    for (uint32_t mb = m0; mb < m1; mb += 64) {
        const uint32_t m = mb + (uint32_t)lane;
        const bool live = m < m1;
        const uint32_t type = live ? mbtype[m] : 0u;
        const uint64_t bi = __builtin_amdgcn_ballot_w64(type != 0);
        const uint32_t proc = type ? procseq[running + gfd_below(bi)] : 0u;
        running += (uint32_t)__builtin_popcountll(bi);
        const uint32_t tag = type ? (type << 5) | (proc << 4) : 0u;
        const uint64_t bc = __builtin_amdgcn_ballot_w64(live && !proc), bt = __builtin_amdgcn_ballot_w64(live && !type);
        if (live) {
            mbtag[m] = (uint8_t)tag;
            if (!proc) cmb[a + gfd_below(bc)] = m;
            if (!type) t0[b + gfd_below(bt)] = m;
            if (type) {
                if (is_P && type >= 2) fl |= HVQ_F_SELF_REF;
                const int my = (int)(m / mw), mx = (int)(m - (uint32_t)my * mw);
                for (int i = 0; i < 3; ++i) {
                    const GPlane *q = &g->pl[i];
                    for (int dy = 0; dy < q->by_per; ++dy)
                        for (int dx = 0; dx < q->bx_per; ++dx)
                            gp_map_ent(g, i, my * q->by_per + dy, mx * q->bx_per + dx)[1] = (uint8_t)tag;
                }
            }
        }
        a += (uint32_t)__builtin_popcountll(bc); b += (uint32_t)__builtin_popcountll(bt);
    }
    g->part[GP_PART2 + tid] = fl;
}

/* gf_layout_sum / gf_layout_blocks with the 64 blocks of a run on the 64 lanes (a wave owns a quarter of the picture's tiles): one
 * load of neighbouring type bytes per run instead of 64 per thread, sums and offsets by DPP scans.  The sums go to the slots of the
 * waves' first threads (zeros elsewhere), so that gfd_layout_finish scans and reduces them as before.  Measured twice: with the roles
 * piled on single SIMDs this form was 0.05 ms slower than the thread-per-run passes with their batched loads (two wave reductions and
 * a division per run); with the roles spread evenly, where the pictures' all-thread passes run side by side and the vector-memory
 * pipeline is what they share, it is 0.07 ms faster (3.10-3.12 -> 3.02-3.04 ms, r04rv). */
__device__ static inline void gfd_run_types(const GPic *g, uint32_t r, int lane, uint32_t *t, uint32_t *by, uint32_t *bx, int *plane, bool *live)
{
    const int i = gp_run_plane(g, r);
    const GPlane *q = &g->pl[i];
    const uint32_t b = (r - q->run_first) * 64u + (uint32_t)lane;
    *plane = i;
    *live = b < q->nblocks;
    const uint32_t y = b / (uint32_t)q->hb;
    *by = y; *bx = b - y * (uint32_t)q->hb;
    *t = *live ? gp_map_ent(g, i, (int)y, (int)*bx)[1] : 0u;
}

__device__ static void gfd_layout_sum(GPic *g, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    GP_G uint32_t *wave_base = GFD_UNIP(GP_G uint32_t *, g->blob + g->wave_base_off);
    const uint32_t tiles = GP_UNI(g->total_tiles), t0 = tiles * (uint32_t)wave / 4u, t1 = tiles * (uint32_t)(wave + 1) / 4u;
    uint32_t run = 0, mi = 0, mp = 0, fl = 0;
    for (uint32_t t = t0; t < t1; ++t) {
        uint32_t ti = 0, tp = 0;
        for (uint32_t k = 0; k < HVQ_TILE_BLOCKS / 64; ++k) {
            const uint32_t r = t * (HVQ_TILE_BLOCKS / 64) + k;
            uint32_t ty, by, bx; int i; bool live;
            gfd_run_types(g, r, lane, &ty, &by, &bx, &i, &live);
            uint32_t n = 0, it = 0, pr = 0, f = 0;
            if (live) gp_type_info(g->is_pb ? 2 : (i == 0 ? 0 : 1), ty, &n, &it, &pr, &f);
            fl |= f;
            /* 64 x (<= 255 dwords, 1 item) fit 16 + 7 bits (an I picture's luma kind is the whole byte, h4m:1093); <= 255 pairs */
            const uint32_t s = gfd_wave_sum(n | (it << 16)), sp = gfd_wave_sum(pr);
            if (lane == 0) wave_base[r] = run;
            run += s & 0xFFFFu; ti += s >> 16; tp += sp;
        }
        mi = ti > mi ? ti : mi; mp = tp > mp ? tp : mp;
    }
    g->part[tid] = fl;
    g->part[GF_P(GF_I_LS, tid)] = lane ? 0u : run;
    g->part[GF_P(GF_I_MI, tid)] = lane ? 0u : mi;
    g->part[GF_P(GF_I_MP, tid)] = lane ? 0u : mp;
}

__device__ static void gfd_layout_blocks(GPic *g, int tid)
{
    if (g->status || g->retry) return;
    const int lane = tid & 63, wave = (int)GP_UNI(tid >> 6);
    GP_G uint32_t *wave_base = GFD_UNIP(GP_G uint32_t *, g->blob + g->wave_base_off);
    GP_G uint32_t *pinfo = GFD_UNIP(GP_G uint32_t *, g->pinfo);
    const uint32_t tiles = GP_UNI(g->total_tiles), t0 = tiles * (uint32_t)wave / 4u, t1 = tiles * (uint32_t)(wave + 1) / 4u;
    const uint32_t base = GP_UNI(g->part[GF_P(GF_I_LS, 64 * wave)]);          /* pool dwords of the waves before this one */
    const uint32_t mw = GP_UNI(g->mw);
    const bool is_pb = g->is_pb != 0;
    for (uint32_t r = t0 * (HVQ_TILE_BLOCKS / 64); r < t1 * (HVQ_TILE_BLOCKS / 64); ++r) {
        uint32_t t, by, bx; int i; bool live;
        gfd_run_types(g, r, lane, &t, &by, &bx, &i, &live);
        const GPlane *q = &g->pl[i];
        const int ctx = is_pb ? 2 : (i == 0 ? 0 : 1);
        uint32_t n = 0, it, pr, f;
        if (live) gp_type_info(ctx, t, &n, &it, &pr, &f);
        const uint32_t run_off = GP_UNI(wave_base[r]) + base;
        const uint32_t off = run_off + gfd_dpp_incl(n) - n;
        if (lane == 0) wave_base[r] = run_off;
        if (!live) continue;
        const uint32_t kind = ctx == 0 ? t : (t & 0xFu);
        const int inter = ctx == 2 && (t & 0x60u);
        uint32_t ent = GP_ENT(off, 0, GP_MODE_NONE);
        if (n) ent = kind == 6 ? GP_ENT(off, 0, GP_MODE_LITERAL)
                   : (inter ? GP_ENT(off, kind - 1, GP_MODE_PREDI) : GP_ENT(off, kind, GP_MODE_BASES));
        uint32_t at = (r - q->run_first) * 64u + (uint32_t)lane;
        if (is_pb) {                                          /* by_per, bx_per are 1 or 2 */
            const uint32_t dy = by & (uint32_t)(q->by_per - 1), dx = bx & (uint32_t)(q->bx_per - 1);
            const uint32_t mb = (by >> (q->by_per >> 1)) * mw + (bx >> (q->bx_per >> 1));
            at = mb * (uint32_t)q->nblk + (dx ? (dy ? 2u : 3u) : (dy ? 1u : 0u));
            if (q->nblk == 1) at = mb;
        }
        pinfo[q->blk_first + at] = ent;
    }
}

/* one wave: scan of the threads' tile totals, maxima and flags of the layout, then sizes and header (gf_layout_finish) */
__device__ static void gfd_layout_finish(GPic *g, int lane)
{
    if (g->status || g->retry) return;
    gfd_scan_add(g, GF_I_LS, lane, 0);
    const uint4 a = *((const GP_G uint4 *)(g->part + GF_P(GF_I_MI, 0)) + lane), b = *((const GP_G uint4 *)(g->part + GF_P(GF_I_MP, 0)) + lane);
    const uint4 f0 = *((const GP_G uint4 *)g->part + lane);
    uint32_t mi = max(max(a.x, a.y), max(a.z, a.w)), mp = max(max(b.x, b.y), max(b.z, b.w)), fl = f0.x | f0.y | f0.z | f0.w;
    if (g->is_pb) { const uint4 f1 = *((const GP_G uint4 *)(g->part + GP_PART2) + lane); fl |= f1.x | f1.y | f1.z | f1.w; }
    else fl |= g->part[GP_MISC + 2 * GC_COUNT];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mi = max(mi, (uint32_t)__shfl_xor((int)mi, d, 64));
        mp = max(mp, (uint32_t)__shfl_xor((int)mp, d, 64));
        fl |= (uint32_t)__shfl_xor((int)fl, d, 64);
    }
    const uint32_t off = (uint32_t)__shfl((int)(g->tot[GF_I_LS - 16]), 0, 64);   /* written by lane 63 just now: LDS is in order within the wave */
    gp_layout_finish(g, off, fl, mi, mp);
}

/* I picture: DC prediction of plane i by one wave (gf_idc_predict is the raster form).  value(by, bx) needs its left and
 * upper neighbours, so the cells of an anti-diagonal are independent: bands of 16 rows, lane = row, lane L works on
 * column t - L in step t and gets the value above from lane L - 1's previous step.  The deltas of 16 columns x 16 rows
 * are fetched one block ahead by all 64 lanes into a three-block ring in LDS (`tile`, 192 dwords). */
#define GF_LDS __attribute__((address_space(3)))
#define GF_WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

__device__ static void gf_idc_predict_wave(GPic *g, int i, uint8_t *rowbuf, uint32_t *tile, int lane)
{
    if (g->status || g->retry) return;
    const GPlane *q = &g->pl[i];
    const int hb = q->hb, vb = q->vb;
    GF_LDS uint8_t *tb = (GF_LDS uint8_t *)tile;                      /* [3][16 rows][16 columns] */
    GF_LDS uint8_t *rb = (GF_LDS uint8_t *)rowbuf;
    const int lrow = lane >> 2, lpart = lane & 3;
    for (int r0 = 0; r0 < vb; r0 += 16) {
        const int nr = vb - r0 < 16 ? vb - r0 : 16;
        const int by = r0 + lane;
        const bool rowlive = lane < nr;
        GP_G uint8_t *myrow = gp_map_ent(g, i, rowlive ? by : r0, 0);
        const GP_G uint8_t *ldrow = gp_map_ent(g, i, lrow < nr ? r0 + lrow : r0, 0);
        uint32_t pend = 0;                                            /* the four deltas this lane fetched for the next block */
        for (int k = 0; k < 4; ++k) { const int bx = 4 * lpart + k; if (lrow < nr && bx < hb) pend |= (uint32_t)ldrow[2 * bx] << (8 * k); }
        uint32_t left = 0, cur = 0;
        const int steps = hb + nr - 1;
        for (int t = 0; t < steps; ++t) {
            if ((t & 15) == 0) {
                const int c = t >> 4;
                ((GF_LDS uint32_t *)tb)[(c % 3) * 64 + lrow * 4 + lpart] = pend;
                uint32_t v = 0;
                for (int k = 0; k < 4; ++k) { const int bx = 16 * (c + 1) + 4 * lpart + k; if (lrow < nr && bx < hb) v |= (uint32_t)ldrow[2 * bx] << (8 * k); }
                pend = v;
                GF_WAVE_SYNC();
            }
            const uint32_t up = (uint32_t)__shfl_up((int)cur, 1, 64);
            const int bx = t - lane;
            if (rowlive && bx >= 0 && bx < hb) {
                const uint32_t delta = tb[((bx >> 4) % 3) * 256 + lane * 16 + (bx & 15)];
                const uint32_t above = by == 0 ? 0x7Fu : (lane == 0 ? rb[bx] : up);
                const uint32_t pred = bx == 0 ? above : (left + above + 1u) >> 1;
                const uint32_t v = (pred + delta) & 0xFFu;
                myrow[2 * bx] = (uint8_t)v;
                if (lane == nr - 1) rb[bx] = (uint8_t)v;
                left = v; cur = v;
            }
        }
        GF_WAVE_SYNC();
    }
}

/* ------------------------------------------------------------------ the kernel */
/* Two kernels from one body: FLAT = the flat path, which hands a picture it cannot serve back to the host marked "redo"
 * (HvqParseResult.pad[0] = 2); !FLAT = the chains, for those pictures (`redo` lists them) or, with HVQM4_AMD_PARSE_FLAT=0,
 * for all.  One kernel with both paths costs the flat path 5 % (93 instead of 12 spilled SGPRs in the chains it shares,
 * profiles/r02ak_*). */
#ifdef GP_PROBE
#define GP_PROBE_ARG , uint32_t exit_word
#else
#define GP_PROBE_ARG
#endif
template <bool FLAT>
__global__ __launch_bounds__(GPW) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8)))
void hvq_parse_kernel_t(const HvqParseJob *__restrict__ jobs, HvqParseResult *__restrict__ results, uint32_t rowbuf_stride,
                        const uint32_t *__restrict__ redo,           /* optional: the pictures to parse, one per workgroup */
                        uint64_t *__restrict__ timing                /* optional: 16 phase timestamps per picture (100 MHz clock) */
                        GP_PROBE_ARG)
{
    const uint32_t pic = redo ? redo[blockIdx.x] : blockIdx.x;
#ifdef GP_PROBE
    /* probe builds (tools/r04_parse_phases.sh): the whole workgroup leaves at stamp `exit_at` -- what the kernel has cost up to there */
    /* skip: 1 decode waves, 2 type/proc chains, 4 vector chains; bits 16-19: K > 0 = only the first K workgroups dealt to a CU go past the trees */
    const uint32_t probe_skip = (exit_word >> 8) & 0xFFu, probe_k = (exit_word >> 16) & 15u;
    const uint32_t exit_at = (probe_k && ((blockIdx.x >> 8) & 7u) >= probe_k) ? 1u : (exit_word & 0xFFu);
#define GP_STAMP(k) do { if (timing && tid == 0) timing[16 * pic + (k)] = wall_clock64(); \
                         if (exit_at == (uint32_t)(k) || (exit_at == 12u && !g.is_pb && (k) == 14)) return; } while (0)
#define GP_SKIP(bit) (probe_skip & (bit))
#else
#define GP_STAMP(k) do { if (timing && tid == 0) timing[16 * pic + (k)] = wall_clock64(); } while (0)
#define GP_SKIP(bit) 0
#endif
    extern __shared__ uint8_t s_rowbuf[];            /* 3 * rowbuf_stride */
    __shared__ GPic g;
    GCode *codes = (GCode *)(gp_stage + GP_STAGE_DWORDS);              /* the six trees follow the staging slots */

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         /* uniform: chains are selected per wave */
    /* Which wave decodes, stages, runs which chain: by the SIMD the wave sits on (HW_ID bits 5:4) and the order in which the CU's
     * workgroups were dealt (blockIdx / 256: one workgroup per CU in turn).  Rounds 1-3 rotated the roles by the wave's index and took
     * that for a spread over the SIMDs; the hardware starts every workgroup's waves on another SIMD, which cancelled the rotation: all
     * eight decode waves of a CU sat on one SIMD, the eight (sleeping) staging waves on a second, the type/x chains on the third, the
     * proc/y chains on the fourth (-DGP_HWID dump, round 4).  Spread evenly (GP_ROLEMAP 1: two decode, two staging, four chain waves
     * per SIMD) the chain phase ends at 2.17 instead of 2.76 ms -- and the kernel is no faster, 3.6-3.75 ms: the pictures then reach
     * their all-thread passes together, and those passes, thread-per-chunk walks whose 64 lanes touch 64 cache lines per instruction,
     * are bound by the CU's one vector-memory pipeline whatever the issue priorities.  Once the tag / list pass had
     * consecutive macroblocks on consecutive lanes the even spread won: 3.20-3.23 ms against 3.36-3.39 (map 3: SIMD 0 decode +
     * staging, 1 staging + x, 2 x + y, 3 y + decode) and 3.46 (the old arrangement), r04rk.  The wave claims its role in LDS, the
     * next free one if another wave of the workgroup has it (waves sharing a SIMD): always a bijection. */
    __shared__ uint32_t s_roles;
#ifndef GP_ROLEMAP
#define GP_ROLEMAP 1
#endif
    const uint32_t simd_id = (uint32_t)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)), dealt = (blockIdx.x >> 8) & 7u;   /* HW_REG_HW_ID, offset 4, 2 bits */
    const int want = GP_ROLEMAP == 0 ? (int)((wave + (int)(((blockIdx.x >> 3) + (blockIdx.x >> 8)) & 3u)) & 3)
                   : (int)((simd_id + (GP_ROLEMAP == 1 ? dealt : (GP_ROLEMAP == 2 ? 2u * (dealt & 1u) : (GP_ROLEMAP == 3 ? (dealt & 1u) : (GP_ROLEMAP == 4 ? (dealt >> 1) : 0u))))) & 3u);
    int role = want;
    const HvqParseJob *job = jobs + pic;
    constexpr bool flat = FLAT;

    GP_STAMP(0);
    if (tid == 0) s_roles = 0;
    /* Issue priorities follow the pictures' critical path, not the waves' age: trees, then the type and proc runs -- short chains that
     * everything else of a picture waits for -- go first.  Left to the oldest-first arbiter, the youngest workgroup of a CU had its
     * 0.27 ms of type runs done after 2.3 ms (the seven older workgroups' vector chains and decode waves took the slots); with them in
     * front it is 0.83 ms and the kernel ends 0.1 ms earlier (3.62 -> 3.53 ms, r04qp).  Raising the vector chains as well starves the
     * decode waves and the all-thread passes (4.08 ms). */
    __builtin_amdgcn_s_setprio(2);
    if (wave == 0) {
        gp_setup(&g, job);
        if ((uint32_t)(g.pl[0].hb + 2) > rowbuf_stride) g.status |= GP_ST_BADARG;
        gp_sections(&g);
        if (flat) { g.mtype.slot = GF_SIDE_SLOT; g.mproc.slot = GF_SIDE_SLOT + 1u; }     /* they run beside the rings */

        GP_ST(g.part[GP_MISC + 13], 0u); GP_ST(g.part[GP_MISC + 14], 0u);
    } else {   /* tree tables start from zero: a malformed tree must not steer a walk through stale LDS */
        uint32_t *w = (uint32_t *)codes;
        for (int k = tid - 64; k < (int)(GC_COUNT * sizeof(GCode) / 4); k += GPW - 64) w[k] = 0;
    }
    __syncthreads();
    {   /* claim the role (see above); lane 0 of every wave, the roles are first used behind two more barriers */
        uint32_t r = (uint32_t)want;
        if (lane == 0) {
            for (int k = 0; k < 4; ++k, r = (r + 1u) & 3u)
                if (!(__hip_atomic_fetch_or(&s_roles, 1u << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & (1u << r))) break;
        }
        role = __builtin_amdgcn_readfirstlane((int)r);
    }
#ifdef GP_HWID
    /* development: which SIMD each role runs on (HW_ID bits 5:4), 8 bits per role, into the slot of stamp 8 */
    if (timing && lane == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       /* HW_REG_HW_ID, all 32 bits */
        atomicAdd((unsigned long long *)&timing[16 * pic + 8], (unsigned long long)(0x80u | ((hw >> 4) & 3u) | (((hw >> 8) & 15u) << 2)) << (8 * role));
    }
#endif
    gp_init_maps(&g, tid, GPW);
    const int is_pb = g.is_pb;
    const int ntrees = is_pb ? 6 : 4;
    gp_read_tree(&g, codes, wave);                               /* BN, RUN, DC, BT: one per wave */
    if (is_pb && wave < 2) gp_read_tree(&g, codes, GC_MV + wave);  /* MV, MCB */
    __syncthreads();
    if (wave == 0) gp_collect_tree_status(&g, ntrees);
    /* the chains read 8-bit tables with the leaf's value in the entry; the flat path's lanes have their own (gf_fill_lane_tables) */
    for (int c = flat ? GC_MV : 0; c < ntrees; ++c) gc_fill_lut(&codes[c], tid, GPW);
    if (flat) {
        gf_move_kids(&g, codes, tid, GPW);
        __syncthreads();
        gf_fill_lane_tables(&g, codes, tid, GPW);
    }
    __syncthreads();
    GP_STAMP(1);
    __builtin_amdgcn_s_setprio(0);

    if (is_pb && !flat) {
        if (wave == 0) gp_mbtypes(&g, codes);
        else if (wave == 1) gp_mbprocs(&g, codes);
        __syncthreads();
        GP_STAMP(2);
        gp_runs_expand(&g, tid, GPW);
        __syncthreads();
        gp_tags_count(&g, tid, GPW);
        __syncthreads();
        if (wave == 0) gp_tags_scan(&g, GPW);
        __syncthreads();
        gp_tags_assign(&g, tid, GPW);
        __syncthreads();
        if (wave == 0) gp_lists_scan(&g, GPW);
        __syncthreads();
        gp_lists_write(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(12);
    }

    if (flat) {
        if (wave == 0) {
            gf_setup_lanes(&g);
            if (lane < 16) {
                const uint32_t p = lane < GF_LANES ? g.lane[lane].pos : 0u;
                GF_ST(gf_pos_pub[lane], p); GF_ST(gf_avail_pub[lane], (p >> 9) << 9);
            }
            GF_ST(gf_done, 0u); GF_ST(gf_types_done, 0u);
#ifdef GF_PROFILE
            gf_slow_count = 0;
#endif
        }
        __syncthreads();
        /* all prefix-coded sections at once: one wave stages the bitstream, one decodes (I: two -- the DC sections, whose
         * rare long codes would hold the other lanes up, have their own); P/B: the type and proc runs beside them */
        if (!is_pb) {
            const uint32_t dc_lanes = 7u << GF_DC0;
            if (role == 0) { gf_decode_wave(&g, codes, lane, GP_SKIP(1u) ? 0u : ~dc_lanes); if (timing && lane == 0) timing[16 * pic + 5] = wall_clock64(); }
            else if (role == 2) gf_decode_wave(&g, codes, lane, GP_SKIP(1u) ? 0u : dc_lanes);
            else if (role == 1) gf_stage_wave(&g, lane, 2u);
        } else {
            if (role == 0) {
                gf_decode_wave(&g, codes, lane, GP_SKIP(1u) ? 0u : ~0u);
#ifndef GP_SUBSTAMPS
                if (timing && lane == 0) timing[16 * pic + 5] = wall_clock64();
#endif
            }
            else if (role == 1) gf_stage_wave(&g, lane, 1u);
            else {
                /* the type runs, then the x components of the vectors (which need nothing but the type bytes); the proc runs,
                 * then -- once the types are there -- the y components */
                const int comp = role - 2;
                __builtin_amdgcn_s_setprio(3);
                if (comp == 0) {
                    if (!GP_SKIP(2u)) gp_mbtypes(&g, codes); else g.ntrun = 0;
#ifndef GP_SUBSTAMPS
                    if (timing && lane == 0) timing[16 * pic + 2] = wall_clock64();
#endif
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    GF_ST(gf_types_done, 1u);
                } else {
                    if (!GP_SKIP(2u)) gp_mbprocs(&g, codes); else { g.nprun = 0; g.pend = 0; }
                    for (uint32_t spin = 0; !GF_LD(gf_types_done) && spin < GF_SPIN_CAP; ++spin) __builtin_amdgcn_s_sleep(4);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                __builtin_amdgcn_s_setprio(0);
                GBits *b = comp ? &g.mvv : &g.mvh;
                b->slot = GF_SIDE_SLOT + (uint32_t)comp; b->base = ~0u;
                const uint32_t f = GP_SKIP(4u) ? 0u : gp_mvs(&g, codes, comp, GF_SIDE_SLOT + 2u + (uint32_t)comp);
                GP_ST(g.part[GP_MISC + 13 + comp], f);
#ifndef GP_SUBSTAMPS
                if (timing && lane == 0) timing[16 * pic + (comp ? 15 : 11)] = wall_clock64();
#endif
            }
        }
        __syncthreads();
        GP_STAMP(13);
        if (is_pb) {
            gp_runs_expand(&g, tid, GPW);
            __syncthreads();
            gfd_tags_lists(&g, tid);                            /* gp_tags_count .. gp_lists_write, lane-consecutive */
            __syncthreads();
            GP_STAMP(12);
        }
        __syncthreads();
        if (wave == 0) gf_fill_const_counts(&g, codes);
        gf_fill_const(&g, codes, tid, GPW);
        __syncthreads();
        /* round 1: DC symbols -> value ends | kinds: zero tokens */
        gf_dc_count(&g, tid, GPW);
        gfd_exp_zeros(&g, 0, 2, tid);
        __syncthreads();
        if (!g.retry && !g.status) {
            for (int k = wave; k < 6; k += 4) {
                if (k < 3) gfd_scan_add(&g, GF_I_TERM(k), lane, &g.nv[k]);
                else gfd_scan_seg(&g, GF_I_DCF(k - 3), GF_I_DCV(k - 3), lane);
            }
        }
        __syncthreads();
#ifdef GP_SUBSTAMPS
        if (is_pb) GP_STAMP(11);
#endif
        /* round 2: values | kinds: blocks covered */
        gf_dc_values(&g, tid, GPW);
        gfd_exp_lens(&g, 0, 2, tid);
        __syncthreads();
#ifdef GP_SUBSTAMPS
        if (is_pb) GP_STAMP(15);
#endif
        if (!is_pb) {
            gfd_exp_write(&g, 0, 2, tid);
            gfd_exp_zeros(&g, 2, 5, tid);
            __syncthreads();
            gfd_exp_lens(&g, 2, 5, tid);
            __syncthreads();
            gfd_exp_write(&g, 2, 5, tid);
            __syncthreads();
            GP_STAMP(14);
            if (wave < 3) gf_idc_predict_wave(&g, wave, s_rowbuf + wave * rowbuf_stride, gp_stage + 192 * wave, lane);
            __syncthreads();
            GP_STAMP(3);
            if (!g.retry) gp_nest(&g, tid, GPW);
        } else {
            gfd_exp_write(&g, 0, 2, tid);
#ifdef GP_SUBSTAMPS
            __syncthreads();
            GP_STAMP(2);
#endif
            gfd_pbdc_sums(&g, tid);
            __syncthreads();
#ifdef GP_SUBSTAMPS
            GP_STAMP(5);
#endif
            gfd_pbdc_write(&g, tid);
            __syncthreads();
            GP_STAMP(3);
        }
        if (!g.retry) {
            gfd_layout_sum(&g, tid);
            __syncthreads();
            GP_STAMP(4);
            if (wave == 0) gfd_layout_finish(&g, lane);
            __syncthreads();
#ifndef GP_HWID
            GP_STAMP(8);
#endif
            gfd_layout_blocks(&g, tid);
            __syncthreads();
            GP_STAMP(9);
            gfd_emit_count(&g, tid);                             /* gf_emit_count and its scans, per wave */
            GP_STAMP(10);
            __syncthreads();
            GP_STAMP(6);
            gfd_emit_merge(&g, tid);
            __syncthreads();
        }
        GP_STAMP(7);
        if (tid == 0) {
            gp_result(&g, (GP_G HvqParseResult *)(results + pic), g.part[GP_MISC + 13] | g.part[GP_MISC + 14]);
            results[pic].pad[0] = (g.retry && !g.status) ? 2u : 0u;       /* 2: not served, the host sends it to the chains */
            results[pic].pad[1] = g.spins;
#ifdef GF_PROFILE
            if (timing) { timing[16 * pic + 9] = g.prof[0]; timing[16 * pic + 10] = g.prof[1]; timing[16 * pic + 6] = g.prof[2]; }
#endif
        }
        return;
    }

    if (!is_pb) {
        if (wave < 2) gp_ikinds(&g, codes, wave);
        else if (wave == 2) gp_idc(&g, codes, 0, s_rowbuf);
        else { gp_idc(&g, codes, 1, s_rowbuf + rowbuf_stride); gp_idc(&g, codes, 2, s_rowbuf + 2 * rowbuf_stride); }
        __syncthreads();
        GP_STAMP(3);
        gp_nest(&g, tid, GPW);
        gp_layout_sum(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(4);
        if (wave == 0) gp_layout_scan(&g, GPW);
        __syncthreads();
        gp_layout_blocks(&g, tid, GPW);
        __syncthreads();
        gp_emit_count(&g, tid, GPW);
        __syncthreads();
        if (wave == 0) gp_emit_scan(&g, GPW);
        __syncthreads();
        gp_emit_compact(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(5);
        if (wave == 0) gp_payload(&g, codes, 0);
        else if (wave == 1) { gp_payload(&g, codes, 1); gp_payload(&g, codes, 2); }
    } else {
        if (wave < 2) gp_pbkinds(&g, codes, wave);
        else if (wave == 2) gp_pbdc(&g, codes, 0);
        else { gp_pbdc(&g, codes, 1); gp_pbdc(&g, codes, 2); }
        __syncthreads();
        GP_STAMP(13);
        gp_kinds_scatter(&g, tid, GPW);
        gp_dc_scatter(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(3);
        gp_layout_sum(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(4);
        if (wave == 0) gp_layout_scan(&g, GPW);
        __syncthreads();
        GP_STAMP(8);
        gp_layout_blocks(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(9);
        gp_emit_count(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(10);
        if (wave == 0) gp_emit_scan(&g, GPW);
        __syncthreads();
        GP_STAMP(11);
        gp_emit_compact(&g, tid, GPW);
        __syncthreads();
        GP_STAMP(5);
        /* balanced by measured chain lengths: Y coefficients | U, V coefficients + Y scalars | MV x + U, V scalars | MV y */
        if (wave == 0) gp_payload(&g, codes, 0);
        else if (wave == 1) { gp_payload(&g, codes, 1); gp_payload(&g, codes, 2); gp_predi_params(&g, codes, 0); }
        else if (wave == 2) {
            const uint32_t fx = gp_mvs(&g, codes, 0, 17u);
            GP_ST(g.part[GP_MISC + 13], fx);
            gp_predi_params(&g, codes, 1); gp_predi_params(&g, codes, 2);
        } else {
            const uint32_t fy = gp_mvs(&g, codes, 1, 18u);
            GP_ST(g.part[GP_MISC + 14], fy);
        }
    }
    __syncthreads();
    GP_STAMP(6);
    gp_emit_merge(&g, tid, GPW);
    __syncthreads();
    GP_STAMP(7);
    if (tid == 0) {
        gp_result(&g, (GP_G HvqParseResult *)(results + pic), g.part[GP_MISC + 13] | g.part[GP_MISC + 14]);
        results[pic].pad[0] = 0;
    }
#undef GP_STAMP
}

/* copy the packed nests that must outlive the batch: pairs[2k] = src, pairs[2k+1] = dst */
extern "C" __global__ __launch_bounds__(128)
void hvq_nest_commit_kernel(const uint64_t *__restrict__ pairs)
{
    const uint4 *src = (const uint4 *)(uintptr_t)pairs[2 * blockIdx.x];
    uint4 *dst = (uint4 *)(uintptr_t)pairs[2 * blockIdx.x + 1];
    const int n = (int)(GP_ALIGN16(HVQ_NESTP_BYTES) / 16);
    for (int k = (int)threadIdx.x; k < n; k += 128) dst[k] = src[k];
}

extern "C" hipError_t hvq_launch_parse(const HvqParseJob *jobs_dev, HvqParseResult *results_dev, uint32_t n,
                                       uint32_t rowbuf_stride, uint32_t use_flat, const uint32_t *redo_dev, uint64_t *timing_dev,
                                       hipStream_t stream)
{
    if (n == 0) return hipSuccess;
#ifdef GP_PROBE
    if (use_flat)
        hipLaunchKernelGGL(hvq_parse_kernel_t<true>, dim3(n), dim3(GPW), 3 * (size_t)rowbuf_stride, stream, jobs_dev, results_dev,
                           rowbuf_stride, redo_dev, timing_dev, 0xFFu);
    else
        hipLaunchKernelGGL(hvq_parse_kernel_t<false>, dim3(n), dim3(GPW), 3 * (size_t)rowbuf_stride, stream, jobs_dev, results_dev,
                           rowbuf_stride, redo_dev, timing_dev, 0xFFu);
#else
    if (use_flat)
        hipLaunchKernelGGL(hvq_parse_kernel_t<true>, dim3(n), dim3(GPW), 3 * (size_t)rowbuf_stride, stream, jobs_dev, results_dev,
                           rowbuf_stride, redo_dev, timing_dev);
    else
        hipLaunchKernelGGL(hvq_parse_kernel_t<false>, dim3(n), dim3(GPW), 3 * (size_t)rowbuf_stride, stream, jobs_dev, results_dev,
                           rowbuf_stride, redo_dev, timing_dev);
#endif
    return hipGetLastError();
}

#ifdef GP_PROBE
/* the flat kernel, leaving at stamp `exit_at`; writes no result records */
extern "C" hipError_t hvq_launch_parse_probe(const HvqParseJob *jobs_dev, HvqParseResult *results_dev, uint32_t n, uint32_t rowbuf_stride,
                                             uint32_t exit_at, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_parse_kernel_t<true>, dim3(n), dim3(GPW), 3 * (size_t)rowbuf_stride, stream, jobs_dev, results_dev,
                       rowbuf_stride, (const uint32_t *)nullptr, (uint64_t *)nullptr, exit_at);
    return hipGetLastError();
}
#endif

extern "C" hipError_t hvq_launch_nest_commit(const uint64_t *pairs_dev, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_nest_commit_kernel, dim3(n), dim3(128), 0, stream, pairs_dev);
    return hipGetLastError();
}

extern "C" int hvq_parse_occupancy(uint32_t rowbuf_stride)
{
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hvq_parse_kernel_t<true>, GPW, 3 * (size_t)rowbuf_stride) != hipSuccess) return -1;
    return n;
}

extern "C" uint32_t hvq_gparse_scratch_bytes(uint32_t total_blocks, uint32_t total_runs, uint32_t nmb)
{
    return gp_scratch_bytes(total_blocks, total_runs, nmb);
}
