/*
 * tests/native/gparse_emul.c -- TEST INFRASTRUCTURE: runs the GPU parse core (hvqm4_amd/csrc/hvq_gparse_core.h)
 * on the CPU, phase by phase in the order hvq_gparse.hip runs them, with the workgroup's threads emulated one
 * after the other.  tests/test_gparse_emul.py compares its blobs with the host parser's byte for byte, so that the
 * chain decomposition is proven before the kernel ever runs.
 */
#include <stdlib.h>
#include <string.h>

#include "../../hvqm4_amd/csrc/hvq_gparse_core.h"

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

int gparse_emul(const uint8_t *pic, uint32_t len, int frame_type, int w, int h, int hs, int vs, int is15,
                uint8_t *blob, uint32_t cap, uint8_t *nest_out, HvqParseResult *res)
{
    const uint32_t nd = (len + 3) / 4 + 4;
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

    uint32_t extra = 0;
    gp_setup(g, &job);
    gp_sections(g);
    for (int t = 0; t < NTHR; ++t) gp_init_maps(g, t, NTHR);
    const int ntrees = g->is_pb ? 6 : 4;
    for (int t = 0; t < ntrees; ++t) gp_read_tree(g, codes, t);
    gp_collect_tree_status(g, ntrees);
    for (int c = 0; c < ntrees; ++c)
        for (int t = 0; t < NTHR; ++t) gc_fill_lut(&codes[c], t, NTHR);
    if (!g->is_pb) {
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
    } else {
        gp_mbtypes(g, codes);
        gp_mbprocs(g, codes);
        for (int t = 0; t < NTHR; ++t) gp_tags_count(g, t, NTHR);
        gp_tags_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_tags_assign(g, t, NTHR);
        gp_lists_scan(g, NTHR);
        for (int t = 0; t < NTHR; ++t) gp_lists_write(g, t, NTHR);
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
        extra |= gp_mvs(g, codes, 0);
        extra |= gp_mvs(g, codes, 1);
    }
    for (int t = 0; t < NTHR; ++t) gp_emit_merge(g, t, NTHR);
    gp_result(g, res, extra);
    free(d); free(scratch); free(rowbuf); free(g); free(codes);
    return 0;
}
