// mfma_valu_micro.hip -- do vector-ALU instructions run in the shadow of MFMAs on one SIMD of gfx950?
// Each wave repeats a body of 4 x (1 v_mfma_f32_32x32x16_bf16 + K plain v_fma_f32 on independent registers); accumulators rotate over 4
// registers sets, so no MFMA waits for its predecessor.  Reported: cycles per body per SIMD (s_memtime of wave 0, 1 or 2 waves per SIMD,
// one workgroup per CU on all 256 CUs).  MFMA alone costs 4 x 32 cycles per wave.  hipcc --offload-arch=gfx950 -O3 mfma_valu_micro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, bool MF, int KIND>
__global__ __launch_bounds__(1024) void body(unsigned long long* out, int iters, float seed) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(seed + i); y[i] = (__bf16)(seed * 0.5f + i); }
    float c0 = seed, c1 = seed + 1, c2 = seed + 2, c3 = seed + 3, c4 = seed + 4, c5 = seed + 5, c6 = seed + 6, c7 = seed + 7;
    const float m = 1.0001f, b = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define VALU(n) do { if (K > n) { if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c##n) : "v"(m), "v"(b)); \
                                  else if (KIND == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c##n) : "v"(m)); \
                                  else asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c##n) : "v"(b)); } } while (0)
#define GROUP(acc) do { if (MF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); \
                        VALU(0); VALU(1); VALU(2); VALU(3); VALU(4); VALU(5); VALU(6); VALU(7); } while (0)
        GROUP(a0); GROUP(a1); GROUP(a2); GROUP(a3);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.678f) out[0] = 0;
}

template <int K, bool MF, int KIND>
static void run(const char* name, int threads) {
    unsigned long long* d;
    hipMalloc(&d, 256 * 8);
    const int iters = 2000;
    body<K, MF, KIND><<<256, threads>>>(d, iters, 1.0f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    body<K, MF, KIND><<<256, threads>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    const double mf = MF ? 4.0 * iters * (threads / 64) * 256 * 32768.0 / (ms * 1e-3) / 1e12 : 0.0;
    printf("%-34s %d wave(s)/SIMD: %7.1f ticks per body (4 groups); kernel %.3f ms = %.0f ns per body, %.0f TFLOP/s of MFMA, tick = %.3f ns\n", name,
           threads / 256, s / 256 / iters, ms, ms * 1e6 / iters, mf, ms * 1e6 / iters / (s / 256 / iters));
    hipFree(d);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<0, true, 0>("4 MFMA", threads);
        run<8, false, 0>("32 v_fma", threads);
        run<2, true, 0>("4 x (MFMA + 2 v_fma)", threads);
        run<4, true, 0>("4 x (MFMA + 4 v_fma)", threads);
        run<6, true, 0>("4 x (MFMA + 6 v_fma)", threads);
        run<8, true, 0>("4 x (MFMA + 8 v_fma)", threads);
        run<8, false, 1>("32 v_cndmask", threads);
        run<4, true, 1>("4 x (MFMA + 4 v_cndmask)", threads);
        run<8, true, 1>("4 x (MFMA + 8 v_cndmask)", threads);
        run<8, false, 2>("32 v_mov_dpp", threads);
        run<4, true, 2>("4 x (MFMA + 4 v_mov_dpp)", threads);
        run<8, true, 2>("4 x (MFMA + 8 v_mov_dpp)", threads);
    }
    return 0;
}
