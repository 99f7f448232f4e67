// round 5: does an in-flight v_mfma_f32_32x32x16_bf16 see a LATER write to EXEC?  A queue of MFMAs keeps the matrix pipe busy, the probed MFMA
// (D = A x [I16 | 0], exact) is issued behind them, then EXEC is cleared for a few instructions and restored, as a skipped divergent branch does.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_exec_check.hip -o tools/micro/mfma_exec_check.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int QUEUE, int TOGGLE, int GAP>
__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x & 63, lo = lane & 31, hi = lane >> 5;
    bf16x8 a, sel;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)in[lane * 8 + e];
        sel[e] = (__bf16)((lo == 8 * hi + e) ? 1.f : 0.f);
    }
    f32x16 q0 = {}, q1 = {}, d = {};
    bf16x8 b2 = a;
    asm volatile("" : "+v"(a), "+v"(sel), "+v"(b2), "+v"(q0), "+v"(q1), "+v"(d));
    if (QUEUE) {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t"
            : "+v"(q0), "+v"(q1) : "v"(a), "v"(b2));
    }
    if (TOGGLE) {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
            ".rept %3\n\ts_nop 0\n\t.endr\n\t"
            "s_mov_b64 s[10:11], exec\n\t"
            "s_mov_b64 exec, 0\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "s_mov_b64 exec, s[10:11]\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            : "+v"(d) : "v"(a), "v"(sel), "n"(GAP) : "s10", "s11");
    } else {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            : "+v"(d) : "v"(a), "v"(sel));
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(d), "+v"(q0), "+v"(q1));
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = d[r];
    if (in[0] == 1234.5f) out[0] = q0[0] + q1[0];
}
template <int Q, int T, int G>
static void run(const float* din, float* dout, const float* ref, const char* name) {
    float h[64 * 16];
    hipLaunchKernelGGL((k<Q, T, G>), dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0, bad_hi = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r)
            if (h[l * 16 + r] != ref[l * 16 + r]) { ++bad; if ((l & 31) >= 8) ++bad_hi; }
    printf("%-44s wrong entries %4d (in columns 8..15: %d)\n", name, bad, bad_hi);
}
int main() {
    float hin[64 * 8], *din, *dout, ref[64 * 16];
    for (int i = 0; i < 512; ++i) hin[i] = (float)((i * 37) % 251) - 125.f;   // exact in bf16
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(ref));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    // reference: D[i][j] = A[i][k = j] for j < 16 (0 elsewhere); lane (j, h) register r holds row i = (r & 3) + 8 (r >> 2) + 4 h
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int j = l & 31, h = l >> 5, i = (r & 3) + 8 * (r >> 2) + 4 * h;
            ref[l * 16 + r] = j < 16 ? hin[(i + 32 * (j >> 3)) * 8 + (j & 7)] : 0.f;
        }
    run<0, 0, 0>(din, dout, ref, "no queue, no EXEC write");
    run<1, 0, 0>(din, dout, ref, "queue of 6 MFMAs, no EXEC write");
    run<0, 1, 0>(din, dout, ref, "no queue, EXEC = 0 right after issue");
    run<0, 1, 8>(din, dout, ref, "no queue, EXEC = 0 after 8 cycles");
    run<1, 1, 0>(din, dout, ref, "queue of 6, EXEC = 0 right after issue");
    run<1, 1, 16>(din, dout, ref, "queue of 6, EXEC = 0 after 16 cycles");
    run<1, 1, 64>(din, dout, ref, "queue of 6, EXEC = 0 after 64 cycles");
    run<1, 1, 200>(din, dout, ref, "queue of 6, EXEC = 0 after 200 cycles");
    return 0;
}
