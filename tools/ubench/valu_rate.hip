// Microbenchmark: issue rate of the integer VALU / SALU / LDS instructions the reconstruction kernel is made of,
// at the kernel's occupancy (8 waves per SIMD, 256 CUs).  Each test runs 8 independent dependency chains per lane
// so that neither latency nor register ports serialise; result = wave-instructions per clock per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, BODY)                                                                         \
__global__ __launch_bounds__(256, 8) void k_##NAME(uint32_t *out, int iters, uint32_t seed)       \
{                                                                                                  \
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3,          \
             r4 = r0 * 11 + 4, r5 = r0 * 13 + 5, r6 = r0 * 17 + 6, r7 = r0 * 19 + 7;              \
    uint32_t a = seed | 1, b = seed + 3;                                                          \
    (void)a; (void)b;                                                                              \
    for (int i = 0; i < iters; ++i) {                                                              \
        BODY BODY BODY BODY                                                                        \
    }                                                                                              \
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                  \
}

#define V1(op) asm volatile(op " %0, %0, %1" : "+v"(r0) : "v"(a)); asm volatile(op " %0, %0, %1" : "+v"(r1) : "v"(a)); \
               asm volatile(op " %0, %0, %1" : "+v"(r2) : "v"(a)); asm volatile(op " %0, %0, %1" : "+v"(r3) : "v"(a)); \
               asm volatile(op " %0, %0, %1" : "+v"(r4) : "v"(a)); asm volatile(op " %0, %0, %1" : "+v"(r5) : "v"(a)); \
               asm volatile(op " %0, %0, %1" : "+v"(r6) : "v"(a)); asm volatile(op " %0, %0, %1" : "+v"(r7) : "v"(a));
#define V3(op) asm volatile(op " %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b)); asm volatile(op " %0, %0, %1, %2" : "+v"(r1) : "v"(a), "v"(b)); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b)); asm volatile(op " %0, %0, %1, %2" : "+v"(r3) : "v"(a), "v"(b)); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(r4) : "v"(a), "v"(b)); asm volatile(op " %0, %0, %1, %2" : "+v"(r5) : "v"(a), "v"(b)); \
               asm volatile(op " %0, %0, %1, %2" : "+v"(r6) : "v"(a), "v"(b)); asm volatile(op " %0, %0, %1, %2" : "+v"(r7) : "v"(a), "v"(b));

KERNEL(add_u32,      V1("v_add_u32"))
KERNEL(and_b32,      V1("v_and_b32"))
KERNEL(lshrrev_b32,  V1("v_lshrrev_b32"))
KERNEL(mul_i32_i24,  V1("v_mul_i32_i24"))
KERNEL(mul_lo_u32,   V1("v_mul_lo_u32"))
KERNEL(min_u32,      V1("v_min_u32"))
KERNEL(pk_add_u16,   V1("v_pk_add_u16"))
KERNEL(mad_i32_i24,  V3("v_mad_i32_i24"))
KERNEL(bfe_u32,      V3("v_bfe_u32"))
KERNEL(min3_u32,     V3("v_min3_u32"))
KERNEL(perm_b32,     V3("v_perm_b32"))
KERNEL(lerp_u8,      V3("v_lerp_u8"))
KERNEL(add3_u32,     V3("v_add3_u32"))
KERNEL(lshl_add_u32, V3("v_lshl_add_u32"))
KERNEL(sad_u8,       V3("v_sad_u8"))

#define CND asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r0) : "v"(a) : ); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r1) : "v"(a)); \
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r2) : "v"(a)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r3) : "v"(a)); \
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r4) : "v"(a)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r5) : "v"(a)); \
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r6) : "v"(a)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r7) : "v"(a));
KERNEL(cndmask, CND)
/* round 3: the 21.7-clock figure of `cndmask` above is an artefact of that test, not of the instruction: its asm reads VCC
 * without declaring it, so nothing in the kernel ever writes VCC and the selects return r0 unchanged -- but each one still
 * names the SAME register as source and destination in a chain of 4 per register per iteration with no other work between
 * them, and the implicit VCC read makes every select wait for the loop's s_cmp/s_cbranch pair.  The two tests below hold the
 * mask in an SGPR pair written once before the loop (what compiled code does: v_cmp -> s[..] -> v_cndmask). */
__global__ __launch_bounds__(256, 8) void k_cndmask_sgpr(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3,
             r4 = r0 * 11 + 4, r5 = r0 * 13 + 5, r6 = r0 * 17 + 6, r7 = r0 * 19 + 7;
    uint32_t a = seed | 1;
    unsigned long long m = __ballot((threadIdx.x * 2654435761u + seed) & 0x10000u);
    for (int i = 0; i < iters; ++i) {
#define CS(r) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "s"(m));
#define CSB CS(r0) CS(r1) CS(r2) CS(r3) CS(r4) CS(r5) CS(r6) CS(r7)
        CSB CSB CSB CSB
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
/* compare + select pairs as rgb_clamp / clampi compile them: v_cmp_gt_u32 -> SGPR pair, v_cndmask on it (2 instructions) */
__global__ __launch_bounds__(256, 8) void k_cmp_cndmask(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3,
             r4 = r0 * 11 + 4, r5 = r0 * 13 + 5, r6 = r0 * 17 + 6, r7 = r0 * 19 + 7;
    uint32_t a = seed | 1;
    for (int i = 0; i < iters; ++i) {
#define CC(r) { unsigned long long m_; asm volatile("v_cmp_gt_u32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(r), "=&s"(m_) : "v"(a)); }
#define CCB CC(r0) CC(r1) CC(r2) CC(r3) CC(r4) CC(r5) CC(r6) CC(r7)
        CCB CCB
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
#define DPP asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0)); asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r1)); \
            asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r2)); asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r3)); \
            asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r4)); asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r5)); \
            asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r6)); asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r7));
KERNEL(add_dpp, DPP)
// 64-bit shift: register pairs
__global__ __launch_bounds__(256, 8) void k_lshrrev_b64(uint32_t *out, int iters, uint32_t seed)
{
    uint64_t q0 = threadIdx.x + seed, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13, q6 = q0 * 17, q7 = q0 * 19;
    uint32_t a = (seed & 1) + 1;
    for (int i = 0; i < iters; ++i) {
#define S64(q) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q) : "v"(a));
#define B64 S64(q0) S64(q1) S64(q2) S64(q3) S64(q4) S64(q5) S64(q6) S64(q7)
        B64 B64 B64 B64
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);
}
// scalar ALU: 8 chains in SGPRs
__global__ __launch_bounds__(256, 8) void k_s_add_u32(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
    for (int i = 0; i < iters; ++i) {
#define SA(s) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
#define SB SA(s0) SA(s1) SA(s2) SA(s3) SA(s4) SA(s5) SA(s6) SA(s7)
        SB SB SB SB
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
}
// LDS: conflict-free ds_add_u32 (no return), ds_read_b32, ds_write_b32, unaligned ds_read_b64
__global__ __launch_bounds__(256, 8) void k_ds_add(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t *p = lds + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) __hip_atomic_fetch_add(p + 256 * (j & 15), seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
__global__ __launch_bounds__(256, 8) void k_ds_add_same4(uint32_t *out, int iters, uint32_t seed)
{
    // groups of 4 adjacent lanes add to the same word (what item-major pair order did with ~3 bases per block)
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t *p = lds + (threadIdx.x >> 2);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) __hip_atomic_fetch_add(p + 256 * (j & 15), seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
__global__ __launch_bounds__(256, 8) void k_ds_read_b64u(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 1024; i += 256) ((uint32_t *)lds)[i] = i * seed;
    __syncthreads();
    typedef uint64_t __attribute__((aligned(1))) u64u;
    uint32_t x = threadIdx.x * 2654435761u + seed;
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) { x = x * 1664525u + 1013904223u; acc += *(const u64u *)(lds + ((x >> 8) & 2047) ); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

template <typename K> void run(K kern, const char *name, uint32_t *out, double per_iter)
{
    const int blocks = 256 * 8, iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * per_iter;           // wave-instructions
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-16s %8.3f ms  %6.3f wave-instr/clk/SIMD  (%5.2f clk per instruction per SIMD at 2.4 GHz)\n", name, ms,
           winstr / clk / 1024.0, clk * 1024.0 / winstr); fflush(stdout);
}

int main()
{
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
#define RUN(n) run(k_##n, #n, out, 32.0)
    RUN(add_u32); RUN(and_b32); RUN(lshrrev_b32); RUN(mul_i32_i24); RUN(mul_lo_u32); RUN(min_u32); RUN(pk_add_u16);
    RUN(mad_i32_i24); RUN(bfe_u32); RUN(min3_u32); RUN(perm_b32); RUN(lerp_u8); RUN(add3_u32); RUN(lshl_add_u32); RUN(sad_u8);
    RUN(cndmask); RUN(cndmask_sgpr); RUN(cmp_cndmask); RUN(add_dpp); RUN(lshrrev_b64);
    run(k_s_add_u32, "s_add_u32", out, 32.0);
    run(k_ds_add, "ds_add_u32", out, 32.0);
    run(k_ds_add_same4, "ds_add_u32 x4same", out, 32.0);
    run(k_ds_read_b64u, "ds_read_b64 unal", out, 32.0);
    return 0;
}
