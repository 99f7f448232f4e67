// lds_micro.hip -- throughput of the LDS primitives a scatter could be built from (MI355X, one 512-thread workgroup per CU).
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/micro/lds_micro.hip -o gpurun_out/lds_micro && gpurun_out/lds_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NT 512
#define ITERS 2000
#define UNR 16
__device__ __forceinline__ void lds_fadd(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// mode 0: ds_add_f32, lane -> distinct dword (stride 4 B)            mode 1: pairs of lanes share an address
// mode 2: ds_add_f32, stride 16 B (float4-per-position layout)        mode 3: ds_add_f32, random cell of a 4 K-float window
// mode 4: b128 read-modify-write, all lanes                           mode 5: b128 RMW, odd lanes predicated off
// mode 6: b32 RMW all lanes                                           mode 7: b32 RMW, lanes 32..63 predicated off
// mode 8: ds_add_f32 x4 on a float4 cell (4 consecutive dwords per lane, stride 16 B)
// mode 9: ds_read_b128 only                                           mode 10: ds_write_b128 only
// mode 11: ds_add_rtn_f32 distinct                                    mode 12: ds_pk_add_bf16? (skipped)
template <int MODE>
__global__ __launch_bounds__(NT) void k(float* out, int salt) {
    __shared__ __attribute__((aligned(16))) float s[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += NT) s[i] = 0.f;
    __syncthreads();
    float* base = s + wave * 2048;   // 8 KB per wave
    float acc = 0.f;
    unsigned rnd = tid * 2654435761u + salt;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int o = (u * 64) & 1023;
            if (MODE == 0) lds_fadd(base + ((lane + o) & 2047), 1.f);
            if (MODE == 1) lds_fadd(base + (((lane >> 1) + o) & 2047), 1.f);
            if (MODE == 2) lds_fadd(base + ((lane * 4 + o) & 2047), 1.f);
            if (MODE == 3) { rnd = rnd * 1664525u + 1013904223u; lds_fadd(s + ((rnd >> 12) & 4095), 1.f); }
            if (MODE == 4 || MODE == 5) {
                float4* c = reinterpret_cast<float4*>(base) + ((lane + o) & 511);
                float4 v = *c; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
                if (MODE == 4 || (lane & 1) == 0) *c = v;
                asm volatile("" ::: "memory");
            }
            if (MODE == 6 || MODE == 7) {
                float* c = base + ((lane + o) & 2047);
                float v = *c + 1.f;
                if (MODE == 6 || lane < 32) *c = v;
                asm volatile("" ::: "memory");
            }
            if (MODE == 8) {
                float* c = base + (((lane + o) & 511) << 2);
                lds_fadd(c, 1.f); lds_fadd(c + 1, 1.f); lds_fadd(c + 2, 1.f); lds_fadd(c + 3, 1.f);
            }
            if (MODE == 9) { float4 v = reinterpret_cast<float4*>(base)[(lane + o + it) & 511]; acc += v.x + v.w; }
            if (MODE == 10) { reinterpret_cast<float4*>(base)[(lane + o) & 511] = make_float4(acc, 1.f, 2.f, 3.f); asm volatile("" ::: "memory"); }
            if (MODE == 11) acc += __hip_atomic_fetch_add(base + ((lane + o) & 2047), 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = s[5] + acc;
}

// global float atomics into an L2-resident region (mode 0: random dword of 4 MB, mode 1: contiguous per wave)
template <int MODE>
__global__ __launch_bounds__(NT) void g(float* buf, int salt) {
    unsigned rnd = (blockIdx.x * NT + threadIdx.x) * 2654435761u + salt;
    for (int it = 0; it < 256; ++it) {
        rnd = rnd * 1664525u + 1013904223u;
        const unsigned i = MODE == 0 ? (rnd >> 10) & ((1u << 20) - 1) : ((rnd >> 10) & ((1u << 20) - 64)) & ~63u | (threadIdx.x & 63);
        atomicAdd(buf + i, 1.f);
    }
}

template <int MODE> void run(const char* name, float* out, double lanes_per_instr) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, NT>>>(out, 1);
    hipEventRecord(a);
    k<MODE><<<256, NT>>>(out, 2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = 8.0 * ITERS * UNR;   // wave-level LDS "operations" per CU
    printf("%-52s %8.3f ms  %7.1f ns per wave-op per CU  (~%.0f clk at 2.1 GHz)\n", name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
}
int main() {
    float* out; hipMalloc(&out, 4096);
    run<0>("ds_add_f32, distinct dwords", out, 64);
    run<1>("ds_add_f32, lane pairs share an address", out, 64);
    run<2>("ds_add_f32, stride 16 B", out, 64);
    run<3>("ds_add_f32, random dword of a 16 KB window (8 waves)", out, 64);
    run<11>("ds_add_rtn_f32, distinct dwords", out, 64);
    run<8>("4 x ds_add_f32 on a float4 cell", out, 64);
    run<4>("b128 read + add + write, 64 lanes", out, 64);
    run<5>("b128 read + add + write, odd lanes do not write", out, 64);
    run<6>("b32 read + add + write, 64 lanes", out, 64);
    run<7>("b32 read + add + write, upper half does not write", out, 64);
    run<9>("ds_read_b128 only", out, 64);
    run<10>("ds_write_b128 only", out, 64);
    float* buf; hipMalloc(&buf, 4 << 20); hipMemset(buf, 0, 4 << 20);
    for (int m = 0; m < 2; ++m) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        if (m == 0) g<0><<<1024, NT>>>(buf, 1); else g<1><<<1024, NT>>>(buf, 1);
        hipEventRecord(a);
        if (m == 0) g<0><<<1024, NT>>>(buf, 2); else g<1><<<1024, NT>>>(buf, 2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("global atomicAdd(float) %s: %.3f ms = %.1f G atomics/s\n", m == 0 ? "random dword of 4 MB" : "64 consecutive dwords per wave", ms, 1024.0 * NT * 256 / ms / 1e6);
    }
    return 0;
}
