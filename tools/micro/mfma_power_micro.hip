// mfma_power_micro.hip -- what MFMA rate does an MI355X SUSTAIN on v_mfma_f32_32x32x16_bf16 when the operands carry realistic data?
// The matrix pipe's nominal rate (one 32x32x16 MFMA per 32 cycles per SIMD at 2.4 GHz = 2.5 PFLOP/s) is reached with constant operands;
// with random operands the package runs into its power limit and the clock drops.  Every wave loops over 8 MFMAs on 8 accumulator sets
// with 4 + 4 operand registers (all 256 CUs, 2 waves per SIMD, ~40 ms per case so that the power management settles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHARE>
__global__ __launch_bounds__(512) void body(const bf16x8* __restrict__ ops, float* out, int iters) {
    f32x16 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x16{};
    bf16x8 x[4], y[4];
    for (int i = 0; i < 4; ++i) { x[i] = ops[(i * 2) * 512 + threadIdx.x]; y[i] = ops[(i * 2 + 1) * 512 + threadIdx.x]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // SHARE 0: both operands change from one MFMA to the next; 1: A changes every 4th MFMA; 2: A and B change every 2nd / alternate
            const int xi = SHARE == 0 ? (i & 3) : SHARE == 1 ? (i >> 2) : (i >> 1) & 3;
            const int yi = SHARE == 0 ? ((i + (i >> 2)) & 3) : SHARE == 1 ? (i & 3) : (i & 1) + 2 * (i >> 2);
            a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[xi], y[yi], a[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += a[i][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int n = 8 * 512 * 8;
    unsigned short* h = (unsigned short*)malloc(n * 2);
    bf16x8* d; float* o;
    hipMalloc(&d, n * 2); hipMalloc(&o, 256 * 512 * 4);
    const char* names[] = {"all operands 1.0", "random mantissas, exponents near 1", "N(0,1)-like random values (random sign, exponent, mantissa)", "3-term pattern: hi parts random, lo parts ~2^-9 of them"};
    for (int share = 0; share < 3; ++share)
    for (int mode = (share ? 2 : 0); mode < (share ? 3 : 4); ++mode) {
        srand(1);
        for (int i = 0; i < n; ++i) {
            unsigned short v = 0x3f80;
            if (mode == 1) v = 0x3f80 | (rand() & 0x7f);
            if (mode == 2) v = (unsigned short)(((rand() & 1) << 15) | ((120 + rand() % 10) << 7) | (rand() & 0x7f));
            if (mode == 3) v = (unsigned short)(((rand() & 1) << 15) | (((i / 8 / 512) & 1 ? 111 : 120) + rand() % 10) << 7 | (rand() & 0x7f));
            h[i] = v;
        }
        hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
        const int iters = 200000;
        auto k = share == 0 ? body<0> : share == 1 ? body<1> : body<2>;
        k<<<256, 512>>>(d, o, 2000);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<256, 512>>>(d, o, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double tf = 8.0 * iters * 8 * 256 * 32768.0 / (ms * 1e-3) / 1e12;
        printf("[operand order %d] %-70s %8.2f ms  %7.0f TFLOP/s = %.3f of 2.5 PFLOP/s\n", share, names[mode], ms, tf, tf / 2500);
    }
    return 0;
}
