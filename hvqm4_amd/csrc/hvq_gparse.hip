/*
 * hvq_gparse.hip -- gfx950 kernel around hvq_gparse_core.h: one workgroup (4 wavefronts) parses one picture's
 * bitstream into its descriptor blob, entirely on the GPU.
 *
 * Phase schedule (tests/native/gparse_emul.c runs the same order on the CPU):
 *   all pictures : setup + section cursors (thread 0) | maps/MVs/tree tables cleared (all threads) |
 *                  prefix trees read, one per wave (lane 0) | first-level tables filled (all threads)
 *   I picture    : 5 chains (kinds Y, kinds UV, DC Y, DC U, DC V) | nest + run sums (all) | run scan, header (thread 0) |
 *                  payload entries, fixed-length offsets, compaction (all) | 3 chains (coefficient symbols Y, U, V) |
 *                  basis words merged in, literal blocks copied (all)
 *   P/B picture  : 2 chains (macroblock types; proc runs) | inter ranks, tags, lists of coded / intra macroblocks (all, 5
 *                  steps) | 5 chains (kinds Y, kinds UV, DC Y, DC U, DC V: symbols only) | kinds and DC values placed in
 *                  the maps (all) | run sums (all) | run scan, header (thread 0) |
 *                  payload entries, fixed-length offsets, compaction (all) | 4 waves: coefficient symbols Y; U, V + DC-buffer
 *                  scalars of the MC-residual blocks Y; MV x + scalars U, V; MV y | basis words merged in, literal blocks copied (all)
 * A chain runs wave-uniform (all lanes compute the same values, so its cursors and counters live in scalar registers
 * and its logic runs on the scalar unit); the four waves of a workgroup run different
 * chains at the same time and 8 workgroups share a CU, so the serial bit-level work of thousands of pictures overlaps.
 *
 * LDS per workgroup: the picture state (cursors, geometry), six prefix trees (8-bit table whose entries hold the leaf
 * value, child and leaf tables: 2.5 KB each), 19 staging slots of 128 B (bitstream blocks and work lists) and the DC
 * chains' row buffers: 19.4 KB, 8 workgroups per CU.  At most 80 SGPRs: more costs a wave slot per SIMD on gfx950.
 */
#include <hip/hip_runtime.h>

#include "hvq_gparse_core.h"

#define GPW 256

extern "C" __global__ __launch_bounds__(GPW) __attribute__((amdgpu_num_sgpr(80)))
void hvq_parse_kernel(const HvqParseJob *__restrict__ jobs, HvqParseResult *__restrict__ results, uint32_t rowbuf_stride,
                      uint64_t *__restrict__ timing)     /* optional: 16 phase timestamps per picture (100 MHz clock) */
{
#define GP_STAMP(k) do { if (timing && tid == 0) timing[16 * blockIdx.x + (k)] = wall_clock64(); } while (0)
    extern __shared__ uint8_t s_rowbuf[];            /* 3 * rowbuf_stride */
    __shared__ GPic g;
    __shared__ GCode codes[GC_COUNT];

    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         /* uniform: chains are selected per wave */
    const HvqParseJob *job = jobs + blockIdx.x;

    GP_STAMP(0);
    if (wave == 0) {
        gp_setup(&g, job);
        if ((uint32_t)(g.pl[0].hb + 2) > rowbuf_stride) g.status |= GP_ST_BADARG;
        gp_sections(&g);
        GP_ST(g.part[GP_MISC + 13], 0u); GP_ST(g.part[GP_MISC + 14], 0u);
    } else {   /* tree tables start from zero: a malformed tree must not steer a walk through stale LDS */
        uint32_t *w = (uint32_t *)codes;
        for (int k = tid - 64; k < (int)(sizeof(codes) / 4); k += GPW - 64) w[k] = 0;
    }
    __syncthreads();
    gp_init_maps(&g, tid, GPW);
    const int is_pb = g.is_pb;
    const int ntrees = is_pb ? 6 : 4;
    gp_read_tree(&g, codes, wave);                               /* BN, RUN, DC, BT: one per wave */
    if (is_pb && wave < 2) gp_read_tree(&g, codes, GC_MV + wave);  /* MV, MCB */
    __syncthreads();
    if (wave == 0) gp_collect_tree_status(&g, ntrees);
    for (int c = 0; c < ntrees; ++c) gc_fill_lut(&codes[c], tid, GPW);
    __syncthreads();
    GP_STAMP(1);

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
        if (wave == 0) gp_mbtypes(&g, codes);
        else if (wave == 1) gp_mbprocs(&g, codes);
        __syncthreads();
        GP_STAMP(2);
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
            const uint32_t fx = gp_mvs(&g, codes, 0);
            GP_ST(g.part[GP_MISC + 13], fx);
            gp_predi_params(&g, codes, 1); gp_predi_params(&g, codes, 2);
        } else {
            const uint32_t fy = gp_mvs(&g, codes, 1);
            GP_ST(g.part[GP_MISC + 14], fy);
        }
    }
    __syncthreads();
    GP_STAMP(6);
    gp_emit_merge(&g, tid, GPW);
    __syncthreads();
    GP_STAMP(7);
    if (tid == 0) gp_result(&g, (GP_G HvqParseResult *)(results + blockIdx.x), g.part[GP_MISC + 13] | g.part[GP_MISC + 14]);
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
                                       uint32_t rowbuf_stride, uint64_t *timing_dev, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_parse_kernel, dim3(n), dim3(GPW), 3 * (size_t)rowbuf_stride, stream, jobs_dev, results_dev,
                       rowbuf_stride, timing_dev);
    return hipGetLastError();
}

extern "C" hipError_t hvq_launch_nest_commit(const uint64_t *pairs_dev, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_nest_commit_kernel, dim3(n), dim3(128), 0, stream, pairs_dev);
    return hipGetLastError();
}

extern "C" int hvq_parse_occupancy(uint32_t rowbuf_stride)
{
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hvq_parse_kernel, GPW, 3 * (size_t)rowbuf_stride) != hipSuccess) return -1;
    return n;
}

extern "C" uint32_t hvq_gparse_scratch_bytes(uint32_t total_blocks, uint32_t total_runs, uint32_t nmb)
{
    return gp_scratch_bytes(total_blocks, total_runs, nmb);
}
