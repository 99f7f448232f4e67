#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8_ref(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    for (int j = 0; j < 8; ++j) { hi[j] = (__bf16)v[j]; lo[j] = (__bf16)(v[j] - (float)hi[j]); }
}
__device__ __forceinline__ void split8_dot(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    unsigned u0 = 0x0000bf80u, u1 = 0xbf800000u;   // (-1, 0) and (0, -1) as packed bf16, kept out of the inline-constant encoder
    asm volatile("" : "+v"(u0), "+v"(u1));
    const bf16x2 m0 = __builtin_bit_cast(bf16x2, u0), m1 = __builtin_bit_cast(bf16x2, u1);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const bf16x2 h = {(__bf16)v[j], (__bf16)v[j + 1]};
        const float l0 = __builtin_amdgcn_fdot2_f32_bf16(h, m0, v[j], false);       // v0 - hi0 (exact)
        const float l1 = __builtin_amdgcn_fdot2_f32_bf16(h, m1, v[j + 1], false);
        hi[j] = h[0]; hi[j + 1] = h[1];
        lo[j] = (__bf16)l0; lo[j + 1] = (__bf16)l1;
    }
}
__global__ void k(const float* in, bf16x8* o, int mode) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + j];
    bf16x8 hi, lo;
    if (mode) split8_dot(v, hi, lo); else split8_ref(v, hi, lo);
    o[(blockIdx.x * 256 + threadIdx.x) * 2] = hi; o[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = lo;
}
int main() {
    const int N = 1 << 20;
    float* h = new float[N * 8];
    unsigned s = 12345;
    for (int i = 0; i < N * 8; ++i) { s = s * 1664525u + 1013904223u; float f; unsigned b = s; if (i % 5 == 0) b = (s & 0x807fffffu) | ((100u + (s >> 27)) << 23); memcpy(&f, &b, 4); if (!std::isfinite(f)) f = 1.5f; h[i] = (i % 7 == 0) ? f * 1e-3f : f; }
    float* din; bf16x8 *o0, *o1;
    hipMalloc(&din, N * 32); hipMalloc(&o0, N * 32); hipMalloc(&o1, N * 32);
    hipMemcpy(din, h, N * 32, hipMemcpyHostToDevice);
    k<<<N / 256, 256>>>(din, o0, 0); k<<<N / 256, 256>>>(din, o1, 1);
    unsigned short* a = new unsigned short[N * 16]; unsigned short* b = new unsigned short[N * 16];
    hipMemcpy(a, o0, N * 32, hipMemcpyDeviceToHost); hipMemcpy(b, o1, N * 32, hipMemcpyDeviceToHost);
    long diff = 0; int shown = 0;
    for (long i = 0; i < (long)N * 16; ++i) if (a[i] != b[i]) { ++diff; if (shown++ < 5) printf("diff at %ld: %04x vs %04x (v=%g)\n", i, a[i], b[i], h[(i / 16) * 8 + (i % 8)]); }
    printf("differences: %ld of %ld\n", diff, (long)N * 16);
    return 0;
}
