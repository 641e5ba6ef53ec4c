/*
 * tests/native/gparse_emul.c -- TEST INFRASTRUCTURE: runs the GPU parse core (hvqm4_amd/csrc/hvq_gparse_core.h)
 * on the CPU, phase by phase in the order hvq_gparse.hip runs them, with the workgroup's threads emulated one
 * after the other.  tests/test_gparse_emul.py compares its blobs with the host parser's byte for byte, so that the
 * chain decomposition is proven before the kernel ever runs.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../hvqm4_amd/csrc/hvq_gparse_flat.h"

#define NTHR 256

uint32_t gparse_emul_scratch_bytes(int w, int h, int hs, int vs)
{
    uint32_t blocks = 0, runs = 0;
    for (int i = 0; i < 3; ++i) {
        const int ws = i ? hs == 2 : 0, hh = i ? vs == 2 : 0;
        const uint32_t nb = (uint32_t)((w >> ws) / 4) * (uint32_t)((h >> hh) / 4);
        blocks += nb;
        runs += (nb + HVQ_TILE_BLOCKS - 1) / HVQ_TILE_BLOCKS * (HVQ_TILE_BLOCKS / 64);
    }
    return gp_scratch_bytes(blocks, runs, (uint32_t)(w / 8) * (uint32_t)(h / 8));
}

/* one scan round: what hvq_gparse.hip does between two barriers */
static void scans_add(GPic *g, const int *inst, int n) { for (int k = 0; k < n; ++k) gf_scan_add(g, inst[k], NTHR); }

static void flat_rounds_common(GPic *g)
{
    /* round 1: DC symbols -> value ends; kinds: zero tokens */
    for (int t = 0; t < NTHR; ++t) { gf_dc_count(g, t, NTHR); gf_exp_zeros(g, 0, 2, t, NTHR); }
    { const int a[] = { GF_I_TERM(0), GF_I_TERM(1), GF_I_TERM(2), GF_I_ZERO(0), GF_I_ZERO(1) }; scans_add(g, a, 5); }
    for (int i = 0; i < 3; ++i) { gf_scan_seg(g, GF_I_DCF(i), GF_I_DCV(i), NTHR); g->nv[i] = g->tot[GF_I_TERM(i) - 16]; }
    /* round 2: values; kinds: blocks covered */
    for (int t = 0; t < NTHR; ++t) { gf_dc_values(g, t, NTHR); gf_exp_lens(g, 0, 2, t, NTHR); }
    { const int a[] = { GF_I_LEN(0), GF_I_LEN(1) }; scans_add(g, a, 2); }
}

static void flat_tail(GPic *g)
{
    for (int t = 0; t < NTHR; ++t) gf_layout_sum(g, t, NTHR);
    gf_layout_finish(g, NTHR);
    for (int t = 0; t < NTHR; ++t) gf_layout_blocks(g, t, NTHR);
    for (int t = 0; t < NTHR; ++t) gf_emit_count(g, t, NTHR);
    for (int i = 0; i < 3; ++i) { const int a[] = { GF_I_FX(i), GF_I_NB(i), GF_I_PREDI(i) }; scans_add(g, a, 3); }
    for (int t = 0; t < NTHR; ++t) gf_emit_merge(g, t, NTHR);
}

/* mode 0: chains only (round 1's schedule); 1: flat path with the chains as fallback; 2: flat path, fail instead of
 * falling back (so that tests can tell which pictures the flat path serves) */
int gparse_emul2(const uint8_t *pic, uint32_t len, int frame_type, int w, int h, int hs, int vs, int is15,
                 uint8_t *blob, uint32_t cap, uint8_t *nest_out, HvqParseResult *res, int mode)
{
    const uint32_t nd = (len + 3) / 4 + 8;
    uint32_t *d = calloc(nd, 4);
    uint8_t *scratch = calloc(gparse_emul_scratch_bytes(w, h, hs, vs) + 64, 1);
    uint8_t *rowbuf = malloc(3 * (size_t)(w / 4 + 2));
    GPic *g = calloc(1, sizeof *g);
    GCode *codes = calloc(GC_COUNT, sizeof *codes);
    if (!d || !scratch || !rowbuf || !g || !codes) return -1;
    memcpy(d, pic, len);
    HvqParseJob job;
    memset(&job, 0, sizeof job);
    job.pic = (uint64_t)(uintptr_t)d; job.blob = (uint64_t)(uintptr_t)blob; job.scratch = (uint64_t)(uintptr_t)scratch;
    job.nest_out = (uint64_t)(uintptr_t)nest_out;
    job.len = len; job.pic_dwords = nd; job.cap = cap;
    job.width = (uint16_t)w; job.height = (uint16_t)h; job.frame_type = (uint8_t)frame_type;
    job.h_samp = (uint8_t)hs; job.v_samp = (uint8_t)vs; job.is15 = (uint8_t)is15;

    uint32_t extra = 0, retried = 0;
    int flat = mode != 0;
again:
    extra = 0;
    memset(codes, 0, GC_COUNT * sizeof *codes);
    gp_setup(g, &job);
    gp_sections(g);
    for (int t = 0; t < NTHR; ++t) gp_init_maps(g, t, NTHR);
    const int ntrees = g->is_pb ? 6 : 4;
    for (int t = 0; t < ntrees; ++t) gp_read_tree(g, codes, t);
    gp_collect_tree_status(g, ntrees);
    for (int c = 0; c < ntrees; ++c)
        for (int t = 0; t < NTHR; ++t) gc_fill_lut(&codes[c], t, NTHR);
    if (g->is_pb) {
        gp_mbtypes(g, codes);
        gp_mbprocs(g, codes);
        for (int t = 0; t < NTHR; ++t) gp_runs_expand(g, t, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_tags_count(g, t, NTHR);
        gp_tags_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_tags_assign(g, t, NTHR);
        gp_lists_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_lists_write(g, t, NTHR);
    }
    if (flat) {
        for (int t = 0; t < NTHR; ++t) gf_move_kids(g, codes, t, NTHR);
        for (int t = 0; t < NTHR; ++t) gf_fill_lane_tables(g, codes, t, NTHR);
        gf_setup_lanes(g);
        for (int l = 0; l < (g->is_pb ? GF_RLE0 : GF_LANES); ++l) gf_decode_lane(g, codes, l);
        if (getenv("GF_DEBUG")) {
            fprintf(stderr, "type %#x lanes:", frame_type);
            for (int l = 0; l < (g->is_pb ? GF_RLE0 : GF_LANES); ++l) fprintf(stderr, " %u", g->lane[l].n);
            fprintf(stderr, " | ncoded %u ntype0 %u\n", g->ncoded, g->ntype0);
        }
        gf_fill_const_counts(g, codes);
        for (int t = 0; t < NTHR; ++t) gf_fill_const(g, codes, t, NTHR);
        if (g->is_pb) { extra |= gp_mvs(g, codes, 0, 22u); extra |= gp_mvs(g, codes, 1, 23u); }
        flat_rounds_common(g);
        if (!g->is_pb) {
            for (int t = 0; t < NTHR; ++t) { gf_exp_write(g, 0, 2, t, NTHR); gf_exp_zeros(g, 2, 5, t, NTHR); }
            { const int a[] = { GF_I_ZERO(2), GF_I_ZERO(3), GF_I_ZERO(4) }; scans_add(g, a, 3); }
            for (int t = 0; t < NTHR; ++t) gf_exp_lens(g, 2, 5, t, NTHR);
            { const int a[] = { GF_I_LEN(2), GF_I_LEN(3), GF_I_LEN(4) }; scans_add(g, a, 3); }
            for (int t = 0; t < NTHR; ++t) gf_exp_write(g, 2, 5, t, NTHR);
            for (int i = 0; i < 3; ++i) gf_idc_predict(g, i, rowbuf + (size_t)i * (size_t)(w / 4 + 2));
            if (!g->retry) for (int t = 0; t < NTHR; ++t) gp_nest(g, t, NTHR);
        } else {
            for (int t = 0; t < NTHR; ++t) { gf_exp_write(g, 0, 2, t, NTHR); gf_pbdc_sums(g, t, NTHR); }
            for (int i = 0; i < 3; ++i) gf_scan_seg(g, GF_I_PBF(i), GF_I_PBV(i), NTHR);
            for (int t = 0; t < NTHR; ++t) gf_pbdc_write(g, t, NTHR);
        }
        if (!g->retry) flat_tail(g);
        if (g->retry && !g->status) {
            if (mode == 2) { free(d); free(scratch); free(rowbuf); free(g); free(codes); return 1; }
            flat = 0; retried = 1;
            goto again;
        }
    } else if (!g->is_pb) {
        gp_ikinds(g, codes, 0);
        gp_ikinds(g, codes, 1);
        for (int i = 0; i < 3; ++i) gp_idc(g, codes, i, rowbuf + (size_t)i * (size_t)(w / 4 + 2));
        for (int t = 0; t < NTHR; ++t) gp_nest(g, t, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_layout_sum(g, t, NTHR);
        gp_layout_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_layout_blocks(g, t, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_emit_count(g, t, NTHR);
        gp_emit_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_emit_compact(g, t, NTHR);
        for (int i = 0; i < 3; ++i) gp_payload(g, codes, i);
        for (int t = 0; t < NTHR; ++t) gp_emit_merge(g, t, NTHR);
    } else {
        gp_pbkinds(g, codes, 0);
        gp_pbkinds(g, codes, 1);
        for (int i = 0; i < 3; ++i) gp_pbdc(g, codes, i);
        for (int t = 0; t < NTHR; ++t) gp_kinds_scatter(g, t, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_dc_scatter(g, t, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_layout_sum(g, t, NTHR);
        gp_layout_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_layout_blocks(g, t, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_emit_count(g, t, NTHR);
        gp_emit_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_emit_compact(g, t, NTHR);
        for (int i = 0; i < 3; ++i) gp_payload(g, codes, i);
        for (int i = 0; i < 3; ++i) gp_predi_params(g, codes, i);
        extra |= gp_mvs(g, codes, 0, 17u);
        extra |= gp_mvs(g, codes, 1, 18u);
        for (int t = 0; t < NTHR; ++t) gp_emit_merge(g, t, NTHR);
    }
    gp_result(g, res, extra);
    res->pad[0] = retried;
    free(d); free(scratch); free(rowbuf); free(g); free(codes);
    return 0;
}

int gparse_emul(const uint8_t *pic, uint32_t len, int frame_type, int w, int h, int hs, int vs, int is15,
                uint8_t *blob, uint32_t cap, uint8_t *nest_out, HvqParseResult *res)
{
    return gparse_emul2(pic, len, frame_type, w, h, hs, vs, is15, blob, cap, nest_out, res, 0);
}
