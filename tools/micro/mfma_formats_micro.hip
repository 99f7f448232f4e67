// mfma_formats_micro.hip -- sustained rate of the gfx950 matrix pipe per operand format, on operands that carry data (companion of
// mfma_power_micro.hip; input to tools/studies/split_formats.py: would "f16 main term + fp8 cross terms" (2 pass equivalents per f32-grade
// product instead of the 3 bf16 passes shipped) actually run faster under the package power limit?).  Register-resident streams, 256 CUs x 8
// waves, 8 accumulator sets; reported in "bf16-pass equivalents" of K = 16: one 32x32x64 fp8 MFMA covers four of them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int FMT>
__global__ __launch_bounds__(512) void body(const int* __restrict__ ops, float* out, int iters) {
    f32x16 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x16{};
    i32x8 x[4], y[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { x[i][j] = ops[((i * 2) * 512 + threadIdx.x) * 8 + j]; y[i][j] = ops[((i * 2 + 1) * 512 + threadIdx.x) * 8 + j]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const i32x8 xa = x[i & 3], yb = y[(i + (i >> 2)) & 3];
            if (FMT == 0) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 p = {xa[0], xa[1], xa[2], xa[3]}, q = {yb[0], yb[1], yb[2], yb[3]};
                a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, p), __builtin_bit_cast(bf16x8, q), a[i], 0, 0, 0);
            } else if (FMT == 1) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 p = {xa[0], xa[1], xa[2], xa[3]}, q = {yb[0], yb[1], yb[2], yb[3]};
                a[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, p), __builtin_bit_cast(f16x8, q), a[i], 0, 0, 0);
            } else {
                a[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, yb, a[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // fp8 e4m3 x e4m3, unit scales
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += a[i][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int n = 8 * 512 * 8;   // dwords
    unsigned* h = (unsigned*)malloc(n * 4);
    int* d; float* o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 512 * 4);
    const char* names[] = {"bf16 32x32x16, N(0,1)-like", "f16 32x32x16, N(0,1)-like", "fp8 e4m3 32x32x64 (f8f6f4), random"};
    for (int fmt = 0; fmt < 3; ++fmt) {
        srand(1);
        for (int i = 0; i < n; ++i) {
            unsigned v = 0;
            if (fmt == 0) for (int k = 0; k < 2; ++k) v |= (unsigned)(((rand() & 1) << 15) | ((120 + rand() % 10) << 7) | (rand() & 0x7f)) << (16 * k);
            if (fmt == 1) for (int k = 0; k < 2; ++k) v |= (unsigned)(((rand() & 1) << 15) | ((8 + rand() % 10) << 10) | (rand() & 0x3ff)) << (16 * k);
            if (fmt == 2) for (int k = 0; k < 4; ++k) v |= (unsigned)(((rand() & 1) << 7) | ((3 + rand() % 8) << 3) | (rand() & 7)) << (8 * k);
            h[i] = v;
        }
        hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
        const int iters = fmt == 2 ? 100000 : 200000;
        auto k = fmt == 0 ? body<0> : fmt == 1 ? body<1> : body<2>;
        k<<<256, 512>>>(d, o, 2000);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<256, 512>>>(d, o, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double kk = fmt == 2 ? 64.0 : 16.0;
        const double tf = 8.0 * iters * 8 * 256 * (2.0 * 32 * 32 * kk) / (ms * 1e-3) / 1e12;
        printf("%-44s %8.2f ms  %7.0f TFLOP/s  = %6.1f x 10^12 (32 x 32 x 16)-pass equivalents per second\n", names[fmt], ms, tf, tf * 1e12 / 32768.0 / 1e9);
    }
    return 0;
}
