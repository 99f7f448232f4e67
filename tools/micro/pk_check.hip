// round 5: semantics check of the VOP3P helper forms used by dcn6_kernels.hip (op_sel / op_sel_hi broadcasts) on real hardware.
// hipcc --offload-arch=gfx950 tools/micro/pk_check.hip -o /tmp/pk_check && /tmp/pk_check
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_fma_x(f32x2 s, f32x2 b, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(s), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_fma_y(f32x2 s, f32x2 b, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(s), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_mul_x(f32x2 s, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(s), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_mul_y(f32x2 s, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(s), "v"(b)); return r; }
__global__ void k(const float* in, float* out) {
    const int t = threadIdx.x;
    const f32x2 a = {in[t], in[t + 64]}, b = {in[t + 128], in[t + 192]}, c = {in[t + 256], in[t + 320]};
    f32x2 r[6] = {pk_sub(a, b), pk_fma(a, b, c), pk_fma_x(a, b, c), pk_fma_y(a, b, c), pk_mul_x(a, b), pk_mul_y(a, b)};
    const float e[12] = {a.x - b.x, a.y - b.y, fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.x, b.x, c.x), fmaf(a.x, b.y, c.y),
                         fmaf(a.y, b.x, c.x), fmaf(a.y, b.y, c.y), a.x * b.x, a.x * b.y, a.y * b.x, a.y * b.y};
    for (int i = 0; i < 6; ++i) {
        out[(2 * i) * 64 + t] = r[i].x - e[2 * i];
        out[(2 * i + 1) * 64 + t] = r[i].y - e[2 * i + 1];
    }
}
int main() {
    float h[384], *d, *o, ho[768];
    for (int i = 0; i < 384; ++i) h[i] = (float)((i * 7919) % 1000) / 37.f - 11.f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const char* names[12] = {"pk_sub.x", "pk_sub.y", "pk_fma.x", "pk_fma.y", "pk_fma_x.x", "pk_fma_x.y", "pk_fma_y.x", "pk_fma_y.y", "pk_mul_x.x", "pk_mul_x.y", "pk_mul_y.x", "pk_mul_y.y"};
    for (int i = 0; i < 12; ++i) {
        float m = 0;
        for (int t = 0; t < 64; ++t) m = fmaxf(m, fabsf(ho[i * 64 + t]));
        printf("%-12s max |diff| %g\n", names[i], m);
    }
    return 0;
}
