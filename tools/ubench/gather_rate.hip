// Microbenchmark: how many scattered small loads per clock does one MI355X CU sustain?
// Each lane reads N x 8 bytes (unaligned) from pseudo-random places of a buffer that fits the L2/MALL.
// mode 0: every lane its own random address; mode 1: lane pairs (2k,2k+1) read addresses 4 bytes apart;
// mode 2: 16 consecutive lanes read consecutive 8-byte words (coalesced 128 B); mode 3: as 0 but 4-byte loads;
// mode 4: as 0 but 16-byte loads; mode 5: every lane its own random 128-byte line, 8 loads inside that line
// (1 miss + 7 L1 hits); mode 6: as 5 but 4 loads per line (two lines per 8 loads);
// mode 7: groups of 5 adjacent lanes share a random 128-byte line (8B at 16*j); mode 8: groups of 4; mode 9: groups of 8;
// mode 10: lanes i and i+32 share a line (non-adjacent); mode 11: as 0 but 16-byte loads at 2-byte aligned addresses (the
// interleaved-UV row shape); mode 12: two 8-byte loads 8 bytes apart at a 2-byte aligned address (the same bytes in two instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;
struct __attribute__((aligned(4))) q16 { uint32_t a, b, c, d; };
struct __attribute__((aligned(2))) q16b { uint32_t a, b, c, d; };

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* buf, uint32_t mask, uint32_t* out, int iters)
{
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t key = MODE == 1 ? (t >> 1) : MODE == 2 ? (t >> 4) : MODE == 7 ? (t / 5) : MODE == 8 ? (t >> 2) : MODE == 9 ? (t >> 3) : MODE == 10 ? (t & ~32u) : t;
    uint32_t x = key * 2654435761u + 12345u;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t line = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            uint32_t off = (x >> 4) & mask;
            if (MODE == 5 || MODE == 6) { if (j % (MODE == 5 ? 8 : 4) == 0) line = off & ~127u; off = line + ((j * 24 + (t & 7)) & 119); }
            if (MODE == 1) off += (t & 1) * 4;
            if (MODE == 2) off = (off & ~127u) + (t & 15) * 8;
            if (MODE == 7) off = (off & ~127u) + (t % 5) * 16 + 3;
            if (MODE == 8) off = (off & ~127u) + (t & 3) * 16 + 3;
            if (MODE == 9) off = (off & ~127u) + (t & 7) * 16 + 3;
            if (MODE == 10) off = (off & ~127u) + ((t >> 5) & 1) * 16 + 3;
            if (MODE == 11) { q16b v = *(const q16b*)(buf + (off & ~1u)); acc += v.a + v.b + v.c + v.d; }
            else if (MODE == 12) { acc += *(const u64u*)(buf + (off & ~1u)); acc += *(const u64u*)(buf + (off & ~1u) + 8); }
            else if (MODE == 3) acc += *(const u32u*)(buf + off);
            else if (MODE == 4) { q16 v = *(const q16*)(buf + (off & ~3u)); acc += v.a + v.b + v.c + v.d; }
            else acc += *(const u64u*)(buf + off);
        }
    }
    out[t] = (uint32_t)acc + (uint32_t)(acc >> 32);
}

template <int MODE> void run(const uint8_t* buf, uint32_t mask, uint32_t* out, const char* name)
{
    const int blocks = 256 * 8, iters = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(buf, mask, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(buf, mask, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double loads = (double)blocks * 256 * iters * 8;
    printf("%-34s %8.3f ms  %7.2f Glane-loads/s  %6.2f lane-loads/clk/CU (2.4GHz, 256 CU)\n", name, ms, loads / ms * 1e-6,
           loads / (ms * 1e-3) / 2.4e9 / 256);
}

int main()
{
    for (uint32_t mb : { 2u, 64u }) {
        uint32_t bytes = mb << 20, mask = bytes - 1 - 255;
        uint8_t* buf; uint32_t* out;
        hipMalloc(&buf, bytes + 4096); hipMemset(buf, 1, bytes + 4096);
        hipMalloc(&out, 256 * 8 * 256 * 4);
        printf("buffer %u MB\n", mb);
        run<0>(buf, mask, out, "random 8B per lane");
        run<1>(buf, mask, out, "lane pairs 4B apart (8B loads)");
        run<2>(buf, mask, out, "16 lanes x 8B consecutive");
        run<3>(buf, mask, out, "random 4B per lane");
        run<4>(buf, mask, out, "random 16B per lane");
        run<5>(buf, mask, out, "8 x 8B inside one private line");
        run<6>(buf, mask, out, "4 x 8B inside one private line");
        run<7>(buf, mask, out, "5 adjacent lanes share a line");
        run<8>(buf, mask, out, "4 adjacent lanes share a line");
        run<9>(buf, mask, out, "8 adjacent lanes share a line");
        run<10>(buf, mask, out, "lanes i, i+32 share a line");
        run<11>(buf, mask, out, "random 16B, 2-byte aligned");
        run<12>(buf, mask, out, "2 x 8B adjacent, 2-byte aligned");
        hipFree(buf); hipFree(out);
    }
    return 0;
}
