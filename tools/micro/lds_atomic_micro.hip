// lds_atomic_micro.hip -- throughput of LDS atomic flavours (8 waves per CU, distinct addresses per lane, no return value).
#include <hip/hip_runtime.h>
#include <cstdio>
#define NT 512
#define ITERS 1000
#define UNR 16
template <int MODE>
__global__ __launch_bounds__(NT) void k(float* out) {
    __shared__ __attribute__((aligned(16))) unsigned s[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += NT) s[i] = 0;
    __syncthreads();
    unsigned* base = s + wave * 2048;
    unsigned acc = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int o = (u * 64) & 1023;
            unsigned* a32 = base + ((lane + o) & 2047);
            unsigned long long* a64 = reinterpret_cast<unsigned long long*>(base) + ((lane + o) & 1023);
            if (MODE == 0) __hip_atomic_fetch_add(a32, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add(a64, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_max(a32, (unsigned)(it + u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) { unsigned e = 0; __hip_atomic_compare_exchange_strong(a32, &e, (unsigned)lane, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); acc += e; }
            if (MODE == 4) __hip_atomic_fetch_add(reinterpret_cast<double*>(a64), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 5) __hip_atomic_fetch_add(reinterpret_cast<float*>(a32), 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 6) acc += __hip_atomic_fetch_add(a32, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 7) __hip_atomic_fetch_add(base + (((lane >> 1) + o) & 2047), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // pairs share
            if (MODE == 8) __hip_atomic_fetch_add(s + (((lane * 4 + o) * 2654435761u >> 18) & 16383), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // scattered, all waves one window
            if (MODE == 9) __hip_atomic_fetch_add(base + (((lane & 31) * 4 + (lane >> 5) * 1024 + (u & 3) + o) & 2047), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // stride 16 B
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (float)(s[5] + acc);
}
template <int MODE> void run(const char* name, float* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<256, NT>>>(out);
    (void)hipEventRecord(a);
    k<MODE><<<256, NT>>>(out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-56s %8.3f ms  ~%6.1f clk per wave-instruction per CU at 2.1 GHz\n", name, ms, ms * 1e6 / (8.0 * ITERS * UNR) * 2.1);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    run<0>("ds_add_u32", out);
    run<6>("ds_add_rtn_u32", out);
    run<1>("ds_add_u64", out);
    run<2>("ds_max_u32", out);
    run<3>("ds_cmpst_rtn_b32", out);
    run<4>("ds_add_f64", out);
    run<5>("ds_add_f32", out);
    run<7>("ds_add_u32, lane pairs share an address", out);
    run<8>("ds_add_u32, hashed addresses in a 64 KB window", out);
    run<9>("ds_add_u32, stride 16 B", out);
    return 0;
}
