// rvsr_common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// GEMM core: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bit-for-bit an fmaf chain), one
// 32x32 output tile per wave-instruction, 64-wide wavefronts.  Operand/lane maps (gfx950):
//   A[i][k]: lane l supplies A[i = l & 31][k = l >> 5]
//   B[k][j]: lane l supplies B[k = l >> 5][j = l & 31]
//   D[i][j]: lane l, register r holds D[i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][j = l & 31]
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RVSR_WG 256  // 4 waves of 64

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// row of the 32x32 D tile held in register r of lane-half hi (= lane >> 5)
__device__ __forceinline__ int drow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// XCD-aware workgroup remap (MI355X: 8 XCDs with private L2s, linear workgroup id L is observed to run on
// XCD L % 8): give every XCD a CONTIGUOUS range of the (tile, m-block, batch) space so that the halo rows /
// columns neighbouring tiles share are L2 hits instead of HBM re-reads.  Bijective for any N; affects
// speed only, never correctness.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned L, unsigned N) {
    const unsigned q = N >> 3, r = N & 7, xcd = L & 7, idx = L >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// decode the swizzled id of a (gx, gy, gz) grid
__device__ __forceinline__ void swizzled_block(unsigned& bx, unsigned& by, unsigned& bz, int enable = 1) {
    const unsigned gx = gridDim.x, gy = gridDim.y, N = gx * gy * gridDim.z;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned S = enable ? xcd_swizzle(L, N) : L;
    bx = S % gx;
    by = (S / gx) % gy;
    bz = S / (gx * gy);
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : v * slope;
    return v;
}

// ------------------------------------------------------------------------------------------
// Read-only view of an NCHW f32 activation with the input-side fusions the conv kernels need.
//   mode 0: plain                      virtual (C, Hv, Wv) == stored (C, Hs, Ws)
//   mode 1: zero-insert x2             virtual[c][2y][2x] = stored[c][y][x], 0 elsewhere
//                                      (data-gradient of a stride-2 conv as a stride-1 conv)
//   mode 2: pixel-unshuffle x2         virtual[c][y][x] = stored[c/4][2y + (c%4)/2][2x + c%2]
//                                      (gradient arriving through nn.PixelShuffle(2))
// `act` (optional) is a tensor stored exactly like `p`; the value is multiplied by
// (act > 0 ? 1 : slope): the derivative of ReLU (slope 0) / LeakyReLU(slope) evaluated from the
// saved activation OUTPUT, fused into the load so no separate elementwise pass is needed.
struct TView {
    const float* p;
    const float* act;
    float slope;
    int C;       // virtual channels
    int Hs, Ws;  // stored spatial size
    int Hv, Wv;  // virtual spatial size
    int mode;
};

// Branch-free on purpose: the address is clamped into the tensor and the load is ALWAYS issued, validity is applied
// with a select afterwards.  A load inside a divergent `if` is waited for at the join by hipcc, which turns a
// staging loop into one L2/HBM round trip per element (measured: 8 K cycles to "issue" 24 loads).
__device__ __forceinline__ float tview_get(const TView& v, int b, int c, int y, int x) {
    bool ok = y >= 0 && x >= 0 && y < v.Hv && x < v.Wv && c >= 0 && c < v.C;
    const int yc = ok ? y : 0, xc = ok ? x : 0, cc = ok ? c : 0;
    size_t idx;
    if (v.mode == 0) {  // (uniform branches on the view mode)
        idx = (((size_t)b * v.C + cc) * v.Hs + yc) * v.Ws + xc;
    } else if (v.mode == 1) {
        const int ys = yc >> 1, xs = xc >> 1;
        ok = ok && !((yc | xc) & 1) && ys < v.Hs && xs < v.Ws;
        idx = (((size_t)b * v.C + cc) * v.Hs + (ys < v.Hs ? ys : 0)) * v.Ws + (xs < v.Ws ? xs : 0);
    } else {
        const int cs = cc >> 2, sy = (cc >> 1) & 1, sx = cc & 1;
        idx = (((size_t)b * (v.C >> 2) + cs) * v.Hs + (2 * yc + sy)) * v.Ws + (2 * xc + sx);
    }
    float val = v.p[idx];
    if (v.act != nullptr) val *= (v.act[idx] > 0.f ? 1.f : v.slope);
    return ok ? val : 0.f;
}

// N elements (c[j], y[j], x[j]) of a PLAIN (mode 0) view: all loads of the batch are issued before any value is used.
// tview_get() in a loop costs one L2/HBM round trip per element as soon as the call sits under a condition or the
// view has an `act` tensor (the uniform `act != nullptr` branch splits the loop body into basic blocks that hipcc
// does not schedule loads across): measured 52 K cycles for the 32 gradient values of a lane.
template <int N>
__device__ __forceinline__ void tview_get_batch(const TView& v, int b, const int (&c)[N], const int (&y)[N], const int (&x)[N],
                                                float (&out)[N]) {
    const size_t hw = (size_t)v.Hs * v.Ws;
    size_t idx[N];
    bool ok[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        ok[j] = y[j] >= 0 && x[j] >= 0 && y[j] < v.Hv && x[j] < v.Wv && c[j] >= 0 && c[j] < v.C;
        idx[j] = ok[j] ? ((size_t)b * v.C + c[j]) * hw + (size_t)y[j] * v.Ws + x[j] : 0;
    }
    if (v.act != nullptr) {  // (uniform, around the whole batch)
        float raw[N], a[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            raw[j] = v.p[idx[j]];
            a[j] = v.act[idx[j]];
        }
#pragma unroll
        for (int j = 0; j < N; ++j) out[j] = ok[j] ? raw[j] * (a[j] > 0.f ? 1.f : v.slope) : 0.f;
    } else {
        float raw[N];
#pragma unroll
        for (int j = 0; j < N; ++j) raw[j] = v.p[idx[j]];
#pragma unroll
        for (int j = 0; j < N; ++j) out[j] = ok[j] ? raw[j] : 0.f;
    }
}
// N consecutive channels c0 .. c0+N-1 at one pixel
template <int N>
__device__ __forceinline__ void tview_get_plain(const TView& v, int b, int c0, int y, int x, float (&out)[N]) {
    int c[N], yy[N], xx[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        c[j] = c0 + j;
        yy[j] = y;
        xx[j] = x;
    }
    tview_get_batch<N>(v, b, c, yy, xx, out);
}

// two views concatenated along channels (torch.cat([a, b], 1) without materialising it)
struct TCat {
    TView a, b;  // b.p == nullptr -> single input
};
__device__ __forceinline__ float tcat_get(const TCat& t, int bidx, int c, int y, int x) {
    if (t.b.p == nullptr) return tview_get(t.a, bidx, c, y, x);  // uniform
    const float va = tview_get(t.a, bidx, c, y, x);              // 0 when c >= a.C
    const float vb = tview_get(t.b, bidx, c - t.a.C, y, x);      // 0 when c < a.C
    return c < t.a.C ? va : vb;
}

// ------------------------------------------------------------------------------------------
// host-side error plumbing (capi.hip)
#define RVSR_OK 0
#define RVSR_ERR_UNSUPPORTED 1
#define RVSR_ERR_BAD_ARG 2
#define RVSR_ERR_LAUNCH 3
#define RVSR_ERR_WORKSPACE 4

// dst[i] (+)= sum_q part[q][i], q < P: 64 elements per block, 4 thread groups stride over the
// partials (fixed order -> run-to-run deterministic), combined through LDS.  A second segment (the bias partials of
// the same weight-gradient call: [P][n2] -> dst2) rides in the same launch: blocks beyond ceil(n / 64) take it.
__global__ void rvsr_reduce_partials_kernel(const float* __restrict__ part, int P, size_t n, float* dst,
                                            const float* __restrict__ part2, size_t n2, float* dst2, int accumulate);
#ifdef RVSR_DEFINE_REDUCE
__global__ void rvsr_reduce_partials_kernel(const float* __restrict__ part, int P, size_t n, float* dst,
                                            const float* __restrict__ part2, size_t n2, float* dst2, int accumulate) {
    __shared__ float red[4][64];
    const int ex = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const unsigned nb1 = (unsigned)((n + 63) / 64);
    if (blockIdx.x >= nb1) {   // (uniform) second segment
        part = part2; n = n2; dst = dst2;
    }
    const size_t i = (size_t)(blockIdx.x >= nb1 ? blockIdx.x - nb1 : blockIdx.x) * 64 + ex;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int q = grp;
        for (; q + 12 < P; q += 16) {   // four loads in flight per thread
            s0 += part[(size_t)q * n + i];
            s1 += part[(size_t)(q + 4) * n + i];
            s2 += part[(size_t)(q + 8) * n + i];
            s3 += part[(size_t)(q + 12) * n + i];
        }
        for (; q < P; q += 4) s0 += part[(size_t)q * n + i];
    }
    red[grp][ex] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < n) {
        const float s = (red[0][ex] + red[1][ex]) + (red[2][ex] + red[3][ex]);
        dst[i] = accumulate ? dst[i] + s : s;
    }
}
#endif
static inline void rvsr_launch_reduce(const float* part, int P, size_t n, float* dst, int accumulate, hipStream_t st,
                                      const float* part2 = nullptr, size_t n2 = 0, float* dst2 = nullptr) {
    if (!part2 || !dst2) n2 = 0;
    hipLaunchKernelGGL(rvsr_reduce_partials_kernel, dim3((unsigned)((n + 63) / 64 + (n2 + 63) / 64)), dim3(256), 0, st, part, P, n, dst,
                       part2, n2, dst2, accumulate);
}

#include <stdio.h>
#include <stdlib.h>
// (the XCD-aware workgroup remap is always on; it affects speed only)
static inline int rvsr_swizzle_enabled() { return 1; }
// one error string per host thread, shared by all translation units (defined in misc_kernels.hip)
extern thread_local char rvsr_g_err[256];
#define FAIL(code, ...)                                        \
    do {                                                       \
        snprintf(rvsr_g_err, sizeof(rvsr_g_err), __VA_ARGS__); \
        return code;                                           \
    } while (0)

// opt a kernel into > 64 KB of dynamic LDS; the attribute call is made once per (kernel, size, device)
template <typename K>
static inline int set_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    static thread_local const void* last_k[8] = {nullptr};
    static thread_local size_t last_b[8] = {0};
    static thread_local int last_dev[8] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned slot = (unsigned)(((uintptr_t)(const void*)kernel >> 4) & 7);
    if (last_k[slot] == (const void*)kernel && last_b[slot] >= bytes && last_dev[slot] == dev) return 0;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 1;
    last_k[slot] = (const void*)kernel;
    last_b[slot] = bytes;
    last_dev[slot] = dev;
    return 0;
}

// ------------------------------------------------------------------------------------------
// Raw buffer addressing: `buffer_* v, v_byte_offset, s[rsrc], s_byte_offset offen`.  A tensor plane is addressed as a uniform
// (SGPR) byte offset of the plane plus a 32-bit per-lane byte offset inside it, so the inner loops carry no per-lane 64-bit
// address arithmetic (v_lshl_add_u64 / v_mad_u64_u32 per access otherwise).  Callers check that the view spans < 4 GB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_view(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xfffffffc, 0x00020000);
}
// 2 GB view: any lane offset with bit 31 set is out of range, and a raw buffer load returns 0 for it (the range check covers
// the lane offset only, not the SGPR offset) -- used as "this lane reads zero padding"
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_view_2g(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x80000000u, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned uniform_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)lane_bytes, (int)uniform_bytes, 0));
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned uniform_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, (int)lane_bytes, (int)uniform_bytes, 0);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned uniform_bytes, float4 v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                     __builtin_bit_cast(unsigned, v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)lane_bytes, (int)uniform_bytes, 0);
}
__device__ __forceinline__ void buf_atomic_add(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned uniform_bytes, float v) {
    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rs, (int)lane_bytes, (int)uniform_bytes, 0);
}

// GEMM mode of the library (rvsr_set_gemm_mode, conv_kernels.hip): 0 = bf16x3 split products (hi*hi + hi*lo + lo*hi: f32-grade, the
// default); 1 = exact-f32 MFMA everywhere; 2 = bf16x2 (two terms: the weights -- or, in a weight gradient, the output gradient -- are
// rounded to bf16, the other operand stays a hi + lo pair); 3 = plain bf16 operands (one term).  Accumulation is f32 in every mode.  The
// reduced-term modes exist in the kernels that dominate a training step (conv_fwd5, conv_wgrad2, the DCN kernels); every other
// split-GEMM kernel keeps three terms in all of them.
extern int rvsr_g_gemm_mode;                 // process-wide default (rvsr_set_gemm_mode)
extern thread_local int rvsr_t_gemm_mode;    // the calling host thread's choice (rvsr_set_gemm_mode_thread), -1 = the default
static inline int rvsr_gemm_mode_now() { return rvsr_t_gemm_mode >= 0 ? rvsr_t_gemm_mode : rvsr_g_gemm_mode; }
static inline int rvsr_gemm_terms() { const int m = rvsr_gemm_mode_now(); return m == 2 ? 2 : (m == 3 ? 1 : 3); }
