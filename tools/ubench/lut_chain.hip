// Microbenchmark: the dependent table-lookup chain of a prefix-code decoder (LDS read -> 64-bit shift -> LDS read ...),
// one wave per workgroup doing the chain, the other three idle at the barrier, G workgroups per CU.
// Reports clocks per lookup.  hipcc --offload-arch=gfx950 -O3 tools/ubench/lut_chain.hip -o lut_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(256) void chain(const uint32_t *tab, uint64_t seed, uint32_t rounds, int active_lanes, uint64_t *out, uint32_t *sink, int busy)
{
    __shared__ uint32_t lut[4][256];
    __shared__ uint32_t pad[3800];        // ~19 KB per workgroup like the parse kernel: 8 workgroups per CU
    for (int k = threadIdx.x; k < 1024; k += 256) ((uint32_t *)lut)[k] = tab[k];
    if (threadIdx.x == 0) pad[blockIdx.x % 3800] = 1;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    uint32_t acc = 0;
    if (wave == 0) {
        uint64_t w = seed * (uint64_t)(lane + 1) * 0x9E3779B97F4A7C15ull;
        const uint32_t *l = lut[lane & 3];
        r0 = wall_clock64();
        t0 = __builtin_readcyclecounter();
        if (lane < active_lanes) {
            for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const uint32_t e = l[w >> 56]; w = (w << (e & 63u)) | (e >> 8); acc += e; }
            }
        }
        t1 = __builtin_readcyclecounter();
        r1 = wall_clock64();
    } else if (busy) {          // the other waves grind VALU work
        uint32_t x = threadIdx.x;
        for (uint32_t r = 0; r < rounds * (uint32_t)busy; ++r) { x = x * 1664525u + 1013904223u; x ^= x >> 7; }
        acc = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (acc == 0x12345) sink[0] = acc + pad[3];
}
int main()
{
    std::vector<uint32_t> tab(1024);
    for (int i = 0; i < 1024; ++i) tab[i] = 8u | 0x80u | ((uint32_t)(i * 2654435761u) & 0xFFFFFF00u);
    uint32_t *dt, *sink; uint64_t *out;
    hipMalloc(&dt, 4096); hipMalloc(&sink, 64); hipMalloc(&out, 16 * 4096);
    hipMemcpy(dt, tab.data(), 4096, hipMemcpyHostToDevice);
    const uint32_t rounds = 2000;
    const int grids[] = { 1, 2048 };
    for (int busy = 0; busy <= 8; busy += 4)
        for (int g : grids)
            for (int al : { 1, 13, 64 }) {
                chain<<<g, 256>>>(dt, 12345, rounds, al, out, sink, busy);
                hipDeviceSynchronize();
                std::vector<uint64_t> h(2 * g);
                hipMemcpy(h.data(), out, 16 * g, hipMemcpyDeviceToHost);
                double s = 0, w = 0; for (int i = 0; i < g; ++i) { s += (double)h[2 * i]; w += (double)h[2 * i + 1]; }
                printf("busy %d grid %5d lanes %2d: %.1f s_memtime ticks, %.1f ns per lookup\n", busy, g, al, s / g / (rounds * 8.0), w * 10.0 / g / (rounds * 8.0));
                fflush(stdout);
            }
    return 0;
}
