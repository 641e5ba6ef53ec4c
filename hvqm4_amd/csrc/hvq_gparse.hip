/*
 * hvq_gparse.hip -- gfx950 kernel around hvq_gparse_core.h: one workgroup (4 wavefronts) parses one picture's
 * bitstream into its descriptor blob, entirely on the GPU.
 *
 * Phase schedule (tests/native/gparse_emul.c runs the same order on the CPU):
 *   all pictures : setup + section cursors (thread 0) | maps/MVs/tree tables cleared (all threads) |
 *                  prefix trees read, one per wave (lane 0) | first-level tables filled (all threads)
 *   I picture    : 5 chains (kinds Y, kinds UV, DC Y, DC U, DC V) | nest + run sums (all) | run scan, header (thread 0) |
 *                  3 chains (payload Y, U, V)
 *   P/B picture  : 1 chain (macroblock types, procs) | inter ranks + tags (all, 3 steps) |
 *                  5 chains (kinds Y, kinds UV, DC Y, DC U, DC V) | run sums (all) | run scan, header (thread 0) |
 *                  block offsets (all) | 5 chains (payload Y, U, V, MV x, MV y)
 * A chain runs on lane 0 (and 1) of a wave; the waves of a workgroup run different chains at the same time and
 * ~8 workgroups share a CU, so the serial bit-level work of thousands of pictures overlaps.
 *
 * LDS per workgroup: the picture state (cursors, geometry), six prefix trees with 9-bit tables (2.5 KB each) and
 * the DC chains' row buffers: ~17 KB.
 */
#include <hip/hip_runtime.h>

#include "hvq_gparse_core.h"

#define GPW 256

extern "C" __global__ __launch_bounds__(GPW)
void hvq_parse_kernel(const HvqParseJob *__restrict__ jobs, HvqParseResult *__restrict__ results, uint32_t rowbuf_stride)
{
    extern __shared__ uint8_t s_rowbuf[];            /* 3 * rowbuf_stride */
    __shared__ GPic g;
    __shared__ GCode codes[GC_COUNT];

    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const HvqParseJob *job = jobs + blockIdx.x;

    if (tid == 0) {
        gp_setup(&g, job);
        if ((uint32_t)(g.pl[0].hb + 2) > rowbuf_stride) g.status |= GP_ST_BADARG;
        gp_sections(&g);
        g.part[GP_MISC + 13] = 0; g.part[GP_MISC + 14] = 0;
    }
    {   /* tree tables start from zero: a malformed tree must not steer a walk through stale LDS */
        uint32_t *w = (uint32_t *)codes;
        for (int k = tid; k < (int)(sizeof(codes) / 4); k += GPW) w[k] = 0;
    }
    __syncthreads();
    gp_init_maps(&g, tid, GPW);
    const int is_pb = g.is_pb;
    const int ntrees = is_pb ? 6 : 4;
    if (lane == 0) {
        gp_read_tree(&g, codes, wave);                               /* BN, RUN, DC, BT */
        if (is_pb && wave < 2) gp_read_tree(&g, codes, GC_MV + wave);  /* MV, MCB */
    }
    __syncthreads();
    if (tid == 0) gp_collect_tree_status(&g, ntrees);
    for (int c = 0; c < ntrees; ++c) gc_fill_lut(&codes[c], tid, GPW);
    __syncthreads();

    if (!is_pb) {
        if (lane == 0 && wave < 2) gp_ikinds(&g, codes, wave);
        if (lane == 0 && wave == 2) gp_idc(&g, codes, 0, s_rowbuf);
        if (lane < 2 && wave == 3) gp_idc(&g, codes, 1 + lane, s_rowbuf + (1 + lane) * rowbuf_stride);
        __syncthreads();
        gp_nest(&g, tid, GPW);
        gp_layout_sum(&g, tid, GPW);
        __syncthreads();
        if (tid == 0) gp_layout_scan(&g, GPW);
        __syncthreads();
        if (lane == 0 && wave < 3) gp_ipayload(&g, codes, wave);
    } else {
        if (tid == 0) gp_mbtypes(&g, codes);
        __syncthreads();
        gp_tags_count(&g, tid, GPW);
        __syncthreads();
        if (tid == 0) gp_tags_scan(&g, GPW);
        __syncthreads();
        gp_tags_assign(&g, tid, GPW);
        __syncthreads();
        if (lane == 0 && wave < 2) gp_pbkinds(&g, codes, wave);
        if (lane == 0 && wave == 2) gp_pbdc(&g, codes, 0);
        if (lane < 2 && wave == 3) gp_pbdc(&g, codes, 1 + lane);
        __syncthreads();
        gp_layout_sum(&g, tid, GPW);
        __syncthreads();
        if (tid == 0) gp_layout_scan(&g, GPW);
        __syncthreads();
        gp_layout_blocks(&g, tid, GPW);
        __syncthreads();
        if (lane == 0 && wave == 0) gp_pbpayload(&g, codes, 0);
        if (lane < 2 && wave == 1) gp_pbpayload(&g, codes, 1 + lane);
        if (lane == 0 && wave >= 2) g.part[GP_MISC + 13 + (wave - 2)] = gp_mvs(&g, codes, wave - 2);
    }
    __syncthreads();
    if (tid == 0) gp_result(&g, results + blockIdx.x, g.part[GP_MISC + 13] | g.part[GP_MISC + 14]);
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
                                       uint32_t rowbuf_stride, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_parse_kernel, dim3(n), dim3(GPW), 3 * (size_t)rowbuf_stride, stream, jobs_dev, results_dev, rowbuf_stride);
    return hipGetLastError();
}

extern "C" hipError_t hvq_launch_nest_commit(const uint64_t *pairs_dev, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hvq_nest_commit_kernel, dim3(n), dim3(128), 0, stream, pairs_dev);
    return hipGetLastError();
}

extern "C" uint32_t hvq_gparse_scratch_bytes(uint32_t total_blocks, uint32_t total_runs, uint32_t nmb)
{
    return gp_scratch_bytes(total_blocks, total_runs, nmb);
}
