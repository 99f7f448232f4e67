// lds_f64_pattern_micro.hip -- ds_add_f64 with the address pattern of the DCN backward scatter: lane = (pixel 0..31, half),
// address = plane(half) + (row jitter) * STRIDE + pixel + (column jitter); jitters in {-1, 0} per lane (sub-pixel offsets of random sign).
#include <hip/hip_runtime.h>
#include <cstdio>
#define NT 512
#define ITERS 1000
#define UNR 16
template <int STRIDE, int JIT, int PLANE, int LS = 1, int TY = 0>
__global__ __launch_bounds__(NT) void k(float* out) {
    extern __shared__ double s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 8 * PLANE; i += NT) s[i] = 0.0;
    __syncthreads();
    unsigned rnd = (tid + 1) * 2654435761u;
    for (int it = 0; it < ITERS; ++it) {
        rnd = rnd * 1664525u + 1013904223u;
        const int jx = JIT ? -(int)((rnd >> 20) & 1) : 0, jy = JIT ? -(int)((rnd >> 21) & 1) : 0;
        double* q = s + (4 * hi) * PLANE + (wave + 2 + jy) * STRIDE + 3 + LS * lo + jx;
        if (TY == 1) {
            unsigned long long* u = reinterpret_cast<unsigned long long*>(q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __hip_atomic_fetch_add(u + e * PLANE, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(u + e * PLANE + 1, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(u + e * PLANE + STRIDE, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(u + e * PLANE + STRIDE + 1, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            continue;
        }
        if (TY == 2) {
            unsigned* u = reinterpret_cast<unsigned*>(s) + (4 * hi) * PLANE + (wave + 2 + jy) * STRIDE + 3 + LS * lo + jx;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __hip_atomic_fetch_add(u + e * PLANE, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(u + e * PLANE + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(u + e * PLANE + STRIDE, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(u + e * PLANE + STRIDE + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __hip_atomic_fetch_add(q + e * PLANE, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(q + e * PLANE + 1, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(q + e * PLANE + STRIDE, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(q + e * PLANE + STRIDE + 1, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (float)s[5];
}
template <int STRIDE, int JIT, int PLANE, int LS = 1, int TY = 0> void run(const char* name, float* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto kk = k<STRIDE, JIT, PLANE, LS, TY>;
    (void)hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PLANE * 8);
    kk<<<256, NT, 8 * PLANE * 8>>>(out);
    (void)hipEventRecord(a);
    kk<<<256, NT, 8 * PLANE * 8>>>(out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-40s stride %2d plane %4d: %7.3f ms  ~%5.1f clk per ds_add_f64 per CU at 2.1 GHz\n", name, STRIDE, PLANE, ms, ms * 1e6 / (8.0 * ITERS * UNR) * 2.1);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    run<39, 0, 585>("no jitter", out);
    run<39, 1, 585>("jitter {-1,0} in x and y", out);
    run<40, 1, 600>("jitter {-1,0} in x and y", out);
    run<64, 1, 960>("jitter {-1,0} in x and y", out);
    run<64, 0, 960>("no jitter", out);
    run<48, 1, 720>("jitter {-1,0} in x and y", out);
    run<32, 1, 480>("jitter (stride 32: columns wrap, timing only)", out);
    run<64, 1, 968>("jitter, plane 968 (planes 8 doubles apart mod 32)", out);
    run<64, 1, 976>("jitter, plane 976 (16 apart)", out);
    run<71, 1, 1065, 2>("lanes 2 px apart, jitter", out);
    run<71, 0, 1065, 2>("lanes 2 px apart, no jitter", out);
    run<72, 1, 1080, 2>("lanes 2 px apart, jitter", out);
    run<103, 1, 1545, 3>("lanes 3 px apart, jitter", out);
    run<39, 1, 585, 1, 1>("u64, jitter", out);
    run<39, 0, 585, 1, 1>("u64, no jitter", out);
    run<39, 1, 585, 1, 2>("u32, jitter", out);
    run<39, 0, 585, 1, 2>("u32, no jitter", out);
    run<64, 1, 960, 1, 2>("u32, jitter", out);
    return 0;
}
