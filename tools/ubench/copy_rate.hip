// Microbenchmark: what does a plain streaming kernel reach on one MI355X?  (the practical HBM ceiling next to the 8 TB/s spec)
//   mode 0: copy, 16 B per lane (1 read : 1 write)      mode 1: two reads, one write (a P picture's algorithmic pattern)
//   mode 2: read only                                    mode 3: write only
//   mode 4: copy with 64-byte pieces at a pseudo-random place per 16-lane group (line-sized scattered reads, sequential writes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *__restrict__ o, size_t n, uint32_t mask)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 v = { 0, 0, 0, 0 };
        if (MODE == 0) v = a[i];
        if (MODE == 1) { u32x4 x = a[i], y = b[i]; v = x + y; }
        if (MODE == 2) { v = a[i]; if (v.x == 0x12345678u && v.y == 0x9abcdef0u) o[i] = v; continue; }
        if (MODE == 3) { v.x = (uint32_t)i; }
        if (MODE == 4) { uint32_t g = (uint32_t)(i >> 2) * 2654435761u; v = a[(((size_t)(g & mask)) << 2) + (i & 3)]; }
        o[i] = v;
    }
}

// mode 5: copy, four 16-byte loads in flight per lane, non-temporal stores (a tuned streaming copy)
__global__ __launch_bounds__(256) void k5(const u32x4 *__restrict__ a, u32x4 *__restrict__ o, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        const u32x4 v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(a + i + stride),
                    v2 = __builtin_nontemporal_load(a + i + 2 * stride), v3 = __builtin_nontemporal_load(a + i + 3 * stride);
        __builtin_nontemporal_store(v0, o + i); __builtin_nontemporal_store(v1, o + i + stride);
        __builtin_nontemporal_store(v2, o + i + 2 * stride); __builtin_nontemporal_store(v3, o + i + 3 * stride);
    }
}

template <int MODE> void run(const u32x4 *a, const u32x4 *b, u32x4 *o, size_t n, double bytes_per_elem, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t mask = (uint32_t)(n / 4 - 1);
    k<MODE><<<256 * 16, 256>>>(a, b, o, n, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<256 * 16, 256>>>(a, b, o, n, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.3f ms per pass  %7.1f GB/s\n", name, ms / 5, bytes_per_elem * (double)n * 5 / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t n = (size_t)1 << 27;                      // 2 GiB per buffer of 16-byte elements: far beyond L2 and the Infinity Cache
    u32x4 *a, *b, *o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, n * 16);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16); hipMemset(o, 0, n * 16);
    run<0>(a, b, o, n, 32, "copy, 16 B per lane (read + write)");
    run<1>(a, b, o, n, 48, "two reads + one write");
    run<2>(a, b, o, n, 16, "read only");
    run<3>(a, b, o, n, 16, "write only");
    run<4>(a, b, o, n, 32, "copy, reads in 64-byte pieces at scattered places");
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int blocks : { 256 * 4, 256 * 8, 256 * 16, 256 * 32 }) {
            k5<<<blocks, 256>>>(a, o, n);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) k5<<<blocks, 256>>>(a, o, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("copy, 4 x 16 B in flight per lane, non-temporal, %5d workgroups %8.3f ms per pass  %7.1f GB/s\n", blocks, ms / 5, 32.0 * (double)n * 5 / (ms * 1e-3) / 1e9);
        }
        hipMemcpyAsync(o, a, n * 16, hipMemcpyDeviceToDevice, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipMemcpyAsync(o, a, n * 16, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.3f ms per pass  %7.1f GB/s\n", "hipMemcpyAsync device to device (the runtime's own copy)", ms / 5, 32.0 * (double)n * 5 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
