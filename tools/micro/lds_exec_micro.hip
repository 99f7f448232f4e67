// lds_exec_micro.hip -- does a partially-masked (EXEC) ds_read/write_b128 cost less?  8 waves per CU, throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
#define NT 512
#define ITERS 2000
#define UNR 16
// MODE: 0 all lanes, 1 lanes 0..15, 2 lanes 0..31, 3 even lanes, 4 lanes 0..15 + 32..47, 5: four passes of 16-lane groups,
//       6: two passes even / odd lanes, 7: two passes lanes (l & 16) == 0 / != 0
template <int MODE, int STRIDE>
__global__ __launch_bounds__(NT) void k(float* out) {
    __shared__ __attribute__((aligned(16))) float s[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += NT) s[i] = 0.f;
    __syncthreads();
    float4* base = reinterpret_cast<float4*>(s + wave * 2048);   // 512 float4 per wave
    auto rmw = [&](int o) {
        float4* c = base + ((lane * STRIDE + o) & 511);
        float4 v = *c; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; *c = v;
        asm volatile("" ::: "memory");
    };
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int o = (u * 64) & 511;
            if (MODE == 0) rmw(o);
            if (MODE == 1) { if (lane < 16) rmw(o); }
            if (MODE == 2) { if (lane < 32) rmw(o); }
            if (MODE == 3) { if ((lane & 1) == 0) rmw(o); }
            if (MODE == 4) { if ((lane & 16) == 0) rmw(o); }
            if (MODE == 5) { if (lane < 16) rmw(o); if (lane >= 16 && lane < 32) rmw(o); if (lane >= 32 && lane < 48) rmw(o); if (lane >= 48) rmw(o); }
            if (MODE == 6) { if ((lane & 1) == 0) rmw(o); if (lane & 1) rmw(o); }
            if (MODE == 7) { if ((lane & 16) == 0) rmw(o); if (lane & 16) rmw(o); }
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = s[5];
}
template <int MODE, int STRIDE> void run(const char* name, float* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE, STRIDE><<<256, NT>>>(out);
    (void)hipEventRecord(a);
    k<MODE, STRIDE><<<256, NT>>>(out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-64s stride %d: %8.3f ms  ~%5.1f clk per (wave, step) per CU at 2.1 GHz\n", name, STRIDE, ms, ms * 1e6 / (8.0 * ITERS * UNR) * 2.1);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    run<0, 1>("b128 RMW all 64 lanes", out);
    run<1, 1>("b128 RMW lanes 0..15", out);
    run<2, 1>("b128 RMW lanes 0..31", out);
    run<3, 1>("b128 RMW even lanes", out);
    run<4, 1>("b128 RMW lanes 0..15 and 32..47", out);
    run<5, 1>("b128 RMW as four 16-lane passes", out);
    run<6, 1>("b128 RMW as even / odd passes", out);
    run<7, 1>("b128 RMW as (lane & 16) passes", out);
    run<0, 2>("b128 RMW all 64 lanes", out);
    run<3, 2>("b128 RMW even lanes", out);
    run<6, 2>("b128 RMW as even / odd passes", out);
    run<7, 2>("b128 RMW as (lane & 16) passes", out);
    return 0;
}
