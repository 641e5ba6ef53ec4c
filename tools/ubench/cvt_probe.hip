// Probe (round 3): what do v_cvt_u32_f32 and v_cvt_pk_u8_f32 do on gfx950 with the values the RGB epilogue produces?
// dumpRGB (h4m:897-900) clamps to [0, 255] and truncates.  Prints, for f = -6 .. 262 in steps of 1/8 plus a few extremes,
// the first input where each candidate differs from the reference clamp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

__global__ void probe(const float *in, uint32_t *o_cvt, uint32_t *o_pk, uint32_t *o_floor_pk, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float f = in[i];
    uint32_t a, b = 0, c = 0;
    asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a) : "v"(f));
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(b) : "v"(f));
    float g;
    asm volatile("v_floor_f32 %0, %1" : "=v"(g) : "v"(f));
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(c) : "v"(g));
    o_cvt[i] = a < 255u ? a : 255u;
    o_pk[i] = b;
    o_floor_pk[i] = c;
}

int main()
{
    std::vector<float> v;
    for (int k = -48; k <= 262 * 8; ++k) v.push_back((float)k / 8.0f);
    for (int k = 0; k < 300; ++k) { v.push_back(nextafterf((float)k, 1e9f)); v.push_back(nextafterf((float)k, -1e9f)); }
    const float ext[] = { -1e9f, 1e9f, 4.3e9f, -4.3e9f, 65536.f, -0.0f, 1e-30f, -1e-30f };
    for (float e : ext) v.push_back(e);
    const int n = (int)v.size();
    float *d_in; uint32_t *d[3];
    hipMalloc(&d_in, n * 4); hipMemcpy(d_in, v.data(), n * 4, hipMemcpyHostToDevice);
    for (auto &p : d) hipMalloc(&p, n * 4);
    probe<<<(n + 255) / 256, 256>>>(d_in, d[0], d[1], d[2], n);
    std::vector<uint32_t> r[3];
    for (int k = 0; k < 3; ++k) { r[k].resize(n); hipMemcpy(r[k].data(), d[k], n * 4, hipMemcpyDeviceToHost); }
    const char *name[3] = { "min(v_cvt_u32_f32, 255)", "v_cvt_pk_u8_f32", "v_floor_f32 + v_cvt_pk_u8_f32" };
    for (int k = 0; k < 3; ++k) {
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            const float f = v[i];
            const uint32_t want = f < 0.f ? 0u : f > 255.f ? 255u : (uint32_t)f;
            if (r[k][i] != want) { if (bad < 6) printf("  %s: f = %.9g -> %u, reference clamp %u\n", name[k], f, r[k][i], want); ++bad; }
        }
        printf("%-32s %d of %d inputs differ from the reference clamp\n", name[k], bad, n);
    }
    return 0;
}
