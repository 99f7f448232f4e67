// dcn_generic.hip -- the deformable convolution operator over its WHOLE argument space: any kernel_h x kernel_w, anisotropic stride /
// padding / dilation, group >= 1, any number of channels per deformable group, DCNv1 (no mask) and DCNv2, element types f32 / f64 / f16
// (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:781, and takes every geometry argument
// separately, deform_conv_cuda.cpp:490-569, 571-685, 152-488).  The fused kernels of dcn*_kernels.hip cover what the reference's
// architectures instantiate (3 x 3, isotropic, f32); this file is the general path behind the same operator, organised the way the
// reference organises it -- per batch element: columns in a workspace, GEMMs per group -- because generality is its only job:
//   forward    col = im2col(x_b, offset_b, mask_b);  out_b[g] = W[g] col[g] (+ bias)
//   backward   colg[g] = W[g]^T gOut_b[g];  (gOffset_b, gMask_b) = coord(colg, x_b, ...);  gX_b += col2im(colg, ...);
//              gW[g] += gOut_b[g] im2col(x_b, ...)[g]^T;  gBias += rowsum(gOut_b)
// Sampling rules as everywhere in this library (kernel.cu:467-497 value, :499-524 corner weights, :526-568 coordinate weights): a sample
// contributes iff y > -1, x > -1, y < H, x < W, every corner of its 2 x 2 footprint separately valid.  Arithmetic: f32 for f16 / f32
// tensors, f64 for f64 tensors (the reference computes f16 in f16).  Not a fast path: no LDS tiling of the input, dword gathers,
// a plain tiled GEMM.
#include "rvsr_common.h"

namespace {

typedef _Float16 half_t;
template <typename T> struct AccOf { typedef float type; };
template <> struct AccOf<double> { typedef double type; };

struct GGeo {
    int C, H, W, Ho, Wo, kh, kw, ph, pw, sh, sw, dh, dw, dg;
};

template <typename A> __device__ __forceinline__ A g_floor(A v);
template <> __device__ __forceinline__ float g_floor<float>(float v) { return floorf(v); }
template <> __device__ __forceinline__ double g_floor<double>(double v) { return floor(v); }

// the four corners of a sample at (y, x): indices into the plane (clamped where invalid) and weights (0 where invalid)
template <typename A>
struct GSamp {
    int i00, i01, i10, i11;
    A w00, w01, w10, w11;   // bilinear weights, zero for corners outside the image
    A ly, lx;
    bool v00, v01, v10, v11, inside;
};
template <typename A>
__device__ __forceinline__ GSamp<A> g_sample(A y, A x, int H, int W) {
    GSamp<A> s;
    s.inside = y > (A)-1 && x > (A)-1 && y < (A)H && x < (A)W;
    const A fy = g_floor<A>(y), fx = g_floor<A>(x);
    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    s.ly = y - fy; s.lx = x - fx;
    const A hy = (A)1 - s.ly, hx = (A)1 - s.lx;
    const bool vy0 = y0 >= 0 && y0 <= H - 1, vy1 = y1 >= 0 && y1 <= H - 1, vx0 = x0 >= 0 && x0 <= W - 1, vx1 = x1 >= 0 && x1 <= W - 1;
    s.v00 = s.inside && vy0 && vx0; s.v01 = s.inside && vy0 && vx1; s.v10 = s.inside && vy1 && vx0; s.v11 = s.inside && vy1 && vx1;
    const int cy0 = vy0 ? y0 : 0, cy1 = vy1 ? y1 : 0, cx0 = vx0 ? x0 : 0, cx1 = vx1 ? x1 : 0;
    s.i00 = cy0 * W + cx0; s.i01 = cy0 * W + cx1; s.i10 = cy1 * W + cx0; s.i11 = cy1 * W + cx1;
    s.w00 = s.v00 ? hy * hx : (A)0; s.w01 = s.v01 ? hy * s.lx : (A)0; s.w10 = s.v10 ? s.ly * hx : (A)0; s.w11 = s.v11 ? s.ly * s.lx : (A)0;
    return s;
}

// sample position of tap (i, j) at output pixel (ho, wo) of deformable group g
template <typename T, typename A>
__device__ __forceinline__ void g_pos(const GGeo& q, const T* off, int g, int k, int i, int j, int ho, int wo, size_t p, A& y, A& x) {
    const size_t HWo = (size_t)q.Ho * q.Wo;
    const T* og = off + (size_t)g * 2 * q.kh * q.kw * HWo;
    y = (A)(ho * q.sh - q.ph + i * q.dh) + (A)og[(size_t)(2 * k) * HWo + p];
    x = (A)(wo * q.sw - q.pw + j * q.dw) + (A)og[(size_t)(2 * k + 1) * HWo + p];
}

// col[(c K + k)][p] = mask * bilinear(x[c]); one thread per (c, p), taps in a loop            (kernel.cu:571-633 / :190-254 for v1)
template <typename T>
__global__ void g_im2col_kernel(const T* __restrict__ x, const T* __restrict__ off, const T* __restrict__ msk, GGeo q, T* __restrict__ col) {
    typedef typename AccOf<T>::type A;
    const size_t HWo = (size_t)q.Ho * q.Wo, total = (size_t)q.C * HWo;
    const int K = q.kh * q.kw, cpg = q.C / q.dg;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx / HWo);
        const size_t p = idx - (size_t)c * HWo;
        const int ho = (int)(p / q.Wo), wo = (int)(p - (size_t)ho * q.Wo), g = c / cpg;
        const T* plane = x + (size_t)c * q.H * q.W;
        for (int i = 0; i < q.kh; ++i)
            for (int j = 0; j < q.kw; ++j) {
                const int k = i * q.kw + j;
                A y, xx;
                g_pos<T, A>(q, off, g, k, i, j, ho, wo, p, y, xx);
                const GSamp<A> s = g_sample<A>(y, xx, q.H, q.W);
                A v = (A)0;
                if (s.inside) {
                    v = s.w00 * (A)plane[s.i00] + s.w01 * (A)plane[s.i01] + s.w10 * (A)plane[s.i10] + s.w11 * (A)plane[s.i11];
                    if (msk != nullptr) v *= (A)msk[((size_t)g * K + k) * HWo + p];
                }
                col[((size_t)c * K + k) * HWo + p] = (T)v;
            }
    }
}

// gx[c][corner] += colg[(c K + k)][p] * mask * corner weight; one thread per (c, k, p)        (kernel.cu:636-693 / :256-326)
template <typename T, typename G>
__global__ void g_col2im_kernel(const T* __restrict__ colg, const T* __restrict__ off, const T* __restrict__ msk, GGeo q, G* __restrict__ gx) {
    typedef typename AccOf<T>::type A;
    const size_t HWo = (size_t)q.Ho * q.Wo;
    const int K = q.kh * q.kw, cpg = q.C / q.dg;
    const size_t total = (size_t)q.C * K * HWo;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t ck = idx / HWo, p = idx - ck * HWo;
        const int c = (int)(ck / K), k = (int)(ck - (size_t)c * K), i = k / q.kw, j = k - i * q.kw;
        const int ho = (int)(p / q.Wo), wo = (int)(p - (size_t)ho * q.Wo), g = c / cpg;
        A y, xx;
        g_pos<T, A>(q, off, g, k, i, j, ho, wo, p, y, xx);
        const GSamp<A> s = g_sample<A>(y, xx, q.H, q.W);
        if (!s.inside) continue;
        A top = (A)colg[idx];
        if (msk != nullptr) top *= (A)msk[((size_t)g * K + k) * HWo + p];
        G* plane = gx + (size_t)c * q.H * q.W;
        if (s.v00) atomicAdd(plane + s.i00, (G)(s.w00 * top));
        if (s.v01) atomicAdd(plane + s.i01, (G)(s.w01 * top));
        if (s.v10) atomicAdd(plane + s.i10, (G)(s.w10 * top));
        if (s.v11) atomicAdd(plane + s.i11, (G)(s.w11 * top));
    }
}

// gOffset[g][2k + dir][p] = sum over the group's channels of colg * mask * d(sample)/d(coordinate); gMask[g][k][p] = sum colg * sample
// one thread per (g, k, p)                                                                    (kernel.cu:696-767 / :328-399)
template <typename T>
__global__ void g_coord_kernel(const T* __restrict__ colg, const T* __restrict__ x, const T* __restrict__ off, const T* __restrict__ msk,
                               GGeo q, T* __restrict__ goff, T* __restrict__ gmsk) {
    typedef typename AccOf<T>::type A;
    const size_t HWo = (size_t)q.Ho * q.Wo;
    const int K = q.kh * q.kw, cpg = q.C / q.dg;
    const size_t total = (size_t)q.dg * K * HWo;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t gk = idx / HWo, p = idx - gk * HWo;
        const int g = (int)(gk / K), k = (int)(gk - (size_t)g * K), i = k / q.kw, j = k - i * q.kw;
        const int ho = (int)(p / q.Wo), wo = (int)(p - (size_t)ho * q.Wo);
        A y, xx;
        g_pos<T, A>(q, off, g, k, i, j, ho, wo, p, y, xx);
        const GSamp<A> s = g_sample<A>(y, xx, q.H, q.W);
        const A m = msk != nullptr ? (A)msk[idx] : (A)1;
        A gy = (A)0, gxx = (A)0, gm = (A)0;
        if (s.inside) {
            const A hx = (A)1 - s.lx, hy = (A)1 - s.ly;
            for (int cc = 0; cc < cpg; ++cc) {
                const int c = g * cpg + cc;
                const T* plane = x + (size_t)c * q.H * q.W;
                const A a00 = s.v00 ? (A)plane[s.i00] : (A)0, a01 = s.v01 ? (A)plane[s.i01] : (A)0;
                const A a10 = s.v10 ? (A)plane[s.i10] : (A)0, a11 = s.v11 ? (A)plane[s.i11] : (A)0;
                const A cg = (A)colg[((size_t)c * K + k) * HWo + p];
                gy += cg * (hx * (a10 - a00) + s.lx * (a11 - a01));     // d/dy
                gxx += cg * (hy * (a01 - a00) + s.ly * (a11 - a10));    // d/dx
                gm += cg * (hy * hx * a00 + hy * s.lx * a01 + s.ly * hx * a10 + s.ly * s.lx * a11);
            }
        }
        goff[((size_t)g * 2 * K + 2 * k) * HWo + p] = (T)(gy * m);
        goff[((size_t)g * 2 * K + 2 * k + 1) * HWo + p] = (T)(gxx * m);
        if (gmsk != nullptr) gmsk[idx] = (T)gm;
    }
}

// C(m, n) (+)= sum_k A(m, k) B(k, n) (+ bias[m]); element strides given for every operand so that the three GEMMs of the operator
// (plain, A transposed, B transposed) are one kernel.  64 x 64 tile, 256 threads, 4 x 4 outputs per thread.
template <typename T>
__global__ __launch_bounds__(256) void g_gemm_kernel(int M, int N, int K, const T* __restrict__ Ap, long sam, long sak, const T* __restrict__ Bp,
                                                     long sbk, long sbn, T* __restrict__ Cp, long scm, long scn, const T* __restrict__ bias, int accumulate) {
    typedef typename AccOf<T>::type A;
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ A As[BK][BM + 1], Bs[BK][BN + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    A acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (A)0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        for (int e = tid; e < BM * BK; e += 256) {
            const int kk = e % BK, mm = e / BK;   // (consecutive threads along k: the common A layout is k-contiguous)
            const int m = m0 + mm, k = k0 + kk;
            As[kk][mm] = (m < M && k < K) ? (A)Ap[(long)m * sam + (long)k * sak] : (A)0;
        }
        for (int e = tid; e < BN * BK; e += 256) {
            const int nn = e % BN, kk = e / BN;   // (consecutive threads along n)
            const int n = n0 + nn, k = k0 + kk;
            Bs[kk][nn] = (n < N && k < K) ? (A)Bp[(long)k * sbk + (long)n * sbn] : (A)0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            A av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[kk][ty * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = Bs[kk][tx + 16 * b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int m = m0 + ty * 4 + a;
        if (m >= M) continue;
        const A bb = bias != nullptr ? (A)bias[m] : (A)0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = n0 + tx + 16 * b;
            if (n >= N) continue;
            T* dst = Cp + (long)m * scm + (long)n * scn;
            const A v = acc[a][b] + bb + (accumulate ? (A)*dst : (A)0);
            *dst = (T)v;
        }
    }
}

// gBias[m] += sum_n gOut[m][n]; one block per row
template <typename T>
__global__ __launch_bounds__(256) void g_rowsum_kernel(const T* __restrict__ g, int N, T* __restrict__ gb) {
    typedef typename AccOf<T>::type A;
    __shared__ A red[256];
    const T* row = g + (size_t)blockIdx.x * N;
    A s = (A)0;
    for (int n = threadIdx.x; n < N; n += 256) s += (A)row[n];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) gb[blockIdx.x] = (T)((A)gb[blockIdx.x] + red[0]);
}

// f16: the scatter accumulates in an f32 scratch plane set (no f16 atomics); folded into grad_input afterwards
__global__ void g_fold_half_kernel(const float* __restrict__ scratch, half_t* __restrict__ gx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        gx[i] = (half_t)((float)gx[i] + scratch[i]);
}

unsigned g_blocks(size_t total) {
    const size_t b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65535 * 4 ? 65535 * 4 : b));
}

template <typename T>
void g_gemm(int M, int N, int K, const T* A, long sam, long sak, const T* B, long sbk, long sbn, T* C, long scm, long scn, const T* bias, int acc,
            hipStream_t st) {
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    hipLaunchKernelGGL(g_gemm_kernel<T>, grid, dim3(256), 0, st, M, N, K, A, sam, sak, B, sbk, sbn, C, scm, scn, bias, acc);
}

struct GArgs {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, group, dg;
};

int g_check(const GArgs& a, GGeo& q, const char* who) {
    if (a.B <= 0 || a.C <= 0 || a.H <= 0 || a.W <= 0 || a.Co <= 0 || a.kh <= 0 || a.kw <= 0 || a.sh <= 0 || a.sw <= 0 || a.dh <= 0 || a.dw <= 0 ||
        a.ph < 0 || a.pw < 0 || a.group <= 0 || a.dg <= 0)
        FAIL(RVSR_ERR_BAD_ARG, "%s: non-positive size", who);
    if (a.C % a.group || a.Co % a.group) FAIL(RVSR_ERR_BAD_ARG, "%s: channels (%d -> %d) not divisible into %d groups", who, a.C, a.Co, a.group);
    if (a.C % a.dg) FAIL(RVSR_ERR_BAD_ARG, "%s: input channels %d not divisible by deformable group %d", who, a.C, a.dg);
    q.C = a.C; q.H = a.H; q.W = a.W; q.kh = a.kh; q.kw = a.kw; q.ph = a.ph; q.pw = a.pw; q.sh = a.sh; q.sw = a.sw; q.dh = a.dh; q.dw = a.dw; q.dg = a.dg;
    q.Ho = (a.H + 2 * a.ph - (a.dh * (a.kh - 1) + 1)) / a.sh + 1;
    q.Wo = (a.W + 2 * a.pw - (a.dw * (a.kw - 1) + 1)) / a.sw + 1;
    if (q.Ho <= 0 || q.Wo <= 0) FAIL(RVSR_ERR_BAD_ARG, "%s: empty output (%d x %d)", who, q.Ho, q.Wo);
    if ((size_t)a.C * a.kh * a.kw * q.Ho * q.Wo >= ((size_t)1 << 40)) FAIL(RVSR_ERR_UNSUPPORTED, "%s: column buffer too large", who);
    return RVSR_OK;
}

// the forward needs one column buffer only
template <typename T> size_t g_ws_fwd_bytes(const GArgs& a, const GGeo& q) {
    const size_t n = (size_t)a.C * a.kh * a.kw * q.Ho * q.Wo * sizeof(T);
    return (n + 255) & ~(size_t)255;
}

template <typename T> size_t g_ws_bytes(const GArgs& a, const GGeo& q) {
    // two column buffers (columns, column gradients) + the f32 scatter planes of the f16 path
    size_t n = 2 * (size_t)a.C * a.kh * a.kw * q.Ho * q.Wo * sizeof(T);
    n = (n + 255) & ~(size_t)255;
    if (sizeof(T) == 2) n += (size_t)a.C * a.H * a.W * sizeof(float);
    return n;
}

template <typename T>
int g_forward(const GArgs& a, const T* x, const T* w, const T* bias, const T* off, const T* msk, T* out, void* ws, size_t ws_bytes, hipStream_t st) {
    GGeo q;
    int rc = g_check(a, q, "deform_conv_generic_forward");
    if (rc) return rc;
    if (!ws || ws_bytes < g_ws_fwd_bytes<T>(a, q))
        FAIL(RVSR_ERR_WORKSPACE, "deform_conv_generic_forward: workspace %zu B < %zu B", ws_bytes, g_ws_fwd_bytes<T>(a, q));
    const int K = a.kh * a.kw, cg = a.C / a.group, og = a.Co / a.group;
    const size_t HWo = (size_t)q.Ho * q.Wo;
    T* col = (T*)ws;
    for (int b = 0; b < a.B; ++b) {
        const T* xb = x + (size_t)b * a.C * a.H * a.W;
        const T* ob = off + (size_t)b * a.dg * 2 * K * HWo;
        const T* mb = msk ? msk + (size_t)b * a.dg * K * HWo : nullptr;
        hipLaunchKernelGGL(g_im2col_kernel<T>, dim3(g_blocks((size_t)a.C * HWo)), dim3(256), 0, st, xb, ob, mb, q, col);
        for (int g = 0; g < a.group; ++g)
            g_gemm<T>(og, (int)HWo, cg * K, w + (size_t)g * og * cg * K, (long)cg * K, 1, col + (size_t)g * cg * K * HWo, (long)HWo, 1,
                      out + ((size_t)b * a.Co + (size_t)g * og) * HWo, (long)HWo, 1, bias ? bias + g * og : nullptr, 0, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "deform_conv_generic_forward launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

template <typename T> struct ScatterOf { typedef T type; };
template <> struct ScatterOf<half_t> { typedef float type; };

template <typename T>
int g_backward(const GArgs& a, const T* x, const T* w, const T* off, const T* msk, const T* gout, T* gx, T* goff, T* gmsk, T* gw, T* gb, void* ws,
               size_t ws_bytes, hipStream_t st) {
    typedef typename ScatterOf<T>::type G;
    GGeo q;
    int rc = g_check(a, q, "deform_conv_generic_backward");
    if (rc) return rc;
    if (!ws || ws_bytes < g_ws_bytes<T>(a, q)) FAIL(RVSR_ERR_WORKSPACE, "deform_conv_generic_backward: workspace %zu B < %zu B", ws_bytes, g_ws_bytes<T>(a, q));
    if ((gx == nullptr) != (goff == nullptr)) FAIL(RVSR_ERR_BAD_ARG, "deform_conv_generic_backward: grad_input and grad_offset must be given together");
    const int K = a.kh * a.kw, cg = a.C / a.group, og = a.Co / a.group;
    const size_t HWo = (size_t)q.Ho * q.Wo, ncol = (size_t)a.C * K * HWo;
    T* col = (T*)ws;
    T* colg = col + ncol;
    float* scratch = (float*)((unsigned char*)ws + ((2 * ncol * sizeof(T) + 255) & ~(size_t)255));
    for (int b = 0; b < a.B; ++b) {
        const T* xb = x + (size_t)b * a.C * a.H * a.W;
        const T* ob = off + (size_t)b * a.dg * 2 * K * HWo;
        const T* mb = msk ? msk + (size_t)b * a.dg * K * HWo : nullptr;
        const T* gob = gout + (size_t)b * a.Co * HWo;
        if (gx != nullptr) {
            for (int g = 0; g < a.group; ++g)   // colg[g] (cg K x HWo) = W[g]^T (cg K x og) gOut[g] (og x HWo)
                g_gemm<T>(cg * K, (int)HWo, og, w + (size_t)g * og * cg * K, 1, (long)cg * K, gob + (size_t)g * og * HWo, (long)HWo, 1,
                          colg + (size_t)g * cg * K * HWo, (long)HWo, 1, nullptr, 0, st);
            hipLaunchKernelGGL(g_coord_kernel<T>, dim3(g_blocks((size_t)a.dg * K * HWo)), dim3(256), 0, st, (const T*)colg, xb, ob, mb, q,
                               goff + (size_t)b * a.dg * 2 * K * HWo, gmsk ? gmsk + (size_t)b * a.dg * K * HWo : nullptr);
            T* gxb = gx + (size_t)b * a.C * a.H * a.W;
            if (sizeof(T) == 2) {
                const size_t n = (size_t)a.C * a.H * a.W;
                if (hipMemsetAsync(scratch, 0, n * sizeof(float), st) != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "deform_conv_generic_backward: memset");
                hipLaunchKernelGGL((g_col2im_kernel<T, G>), dim3(g_blocks(ncol)), dim3(256), 0, st, (const T*)colg, ob, mb, q, (G*)scratch);
                hipLaunchKernelGGL(g_fold_half_kernel, dim3(g_blocks(n)), dim3(256), 0, st, (const float*)scratch, (half_t*)gxb, n);
            } else {
                hipLaunchKernelGGL((g_col2im_kernel<T, G>), dim3(g_blocks(ncol)), dim3(256), 0, st, (const T*)colg, ob, mb, q, (G*)gxb);
            }
        }
        if (gw != nullptr) {
            hipLaunchKernelGGL(g_im2col_kernel<T>, dim3(g_blocks((size_t)a.C * HWo)), dim3(256), 0, st, xb, ob, mb, q, col);
            for (int g = 0; g < a.group; ++g)   // gW[g] (og x cg K) += gOut[g] (og x HWo) col[g]^T (HWo x cg K)
                g_gemm<T>(og, cg * K, (int)HWo, gob + (size_t)g * og * HWo, (long)HWo, 1, col + (size_t)g * cg * K * HWo, 1, (long)HWo,
                          gw + (size_t)g * og * cg * K, (long)cg * K, 1, nullptr, 1, st);
        }
        if (gb != nullptr) hipLaunchKernelGGL(g_rowsum_kernel<T>, dim3(a.Co), dim3(256), 0, st, gob, (int)HWo, gb);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "deform_conv_generic_backward launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI (include/realvsr_hip.h section 1c)
// dtype: 0 = f32, 1 = f64, 2 = f16.  mask == NULL: DCNv1 (no modulation; grad_mask must be NULL too).
extern "C" size_t rvsr_deform_conv_generic_workspace_bytes(int dtype, int channels, int height, int width, int kernel_h, int kernel_w, int stride_h,
                                                           int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w) {
    GArgs a = {1, channels, height, width, 1, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, 1, 1};
    GGeo q;
    if (channels <= 0 || g_check(a, q, "deform_conv_generic_workspace_bytes") != RVSR_OK) return 0;
    return dtype == 1 ? g_ws_bytes<double>(a, q) : (dtype == 2 ? g_ws_bytes<half_t>(a, q) : g_ws_bytes<float>(a, q));
}

extern "C" size_t rvsr_deform_conv_generic_forward_workspace_bytes(int dtype, int channels, int height, int width, int kernel_h, int kernel_w,
                                                                   int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w) {
    GArgs a = {1, channels, height, width, 1, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, 1, 1};
    GGeo q;
    if (channels <= 0 || g_check(a, q, "deform_conv_generic_forward_workspace_bytes") != RVSR_OK) return 0;
    return dtype == 1 ? g_ws_fwd_bytes<double>(a, q) : (dtype == 2 ? g_ws_fwd_bytes<half_t>(a, q) : g_ws_fwd_bytes<float>(a, q));
}

extern "C" int rvsr_deform_conv_generic_forward(int dtype, const void* input, const void* weight, const void* bias, const void* offset, const void* mask,
                                                void* output, int batch, int channels, int height, int width, int channels_out, int kernel_h,
                                                int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                                int deformable_group, void* workspace, size_t workspace_bytes, void* stream) {
    if (!input || !weight || !offset || !output) FAIL(RVSR_ERR_BAD_ARG, "deform_conv_generic_forward: null argument");
    const GArgs a = {batch, channels, height, width, channels_out, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                     deformable_group};
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case 0: return g_forward<float>(a, (const float*)input, (const float*)weight, (const float*)bias, (const float*)offset, (const float*)mask,
                                        (float*)output, workspace, workspace_bytes, st);
        case 1: return g_forward<double>(a, (const double*)input, (const double*)weight, (const double*)bias, (const double*)offset, (const double*)mask,
                                         (double*)output, workspace, workspace_bytes, st);
        case 2: return g_forward<half_t>(a, (const half_t*)input, (const half_t*)weight, (const half_t*)bias, (const half_t*)offset, (const half_t*)mask,
                                         (half_t*)output, workspace, workspace_bytes, st);
        default: FAIL(RVSR_ERR_BAD_ARG, "deform_conv_generic_forward: dtype %d (0 = f32, 1 = f64, 2 = f16)", dtype);
    }
}

// grad_input (zero on entry: scatter-add) and grad_offset together or both NULL; grad_mask NULL for DCNv1; grad_weight / grad_bias are
// accumulated into (NULL = skip) -- the conventions of rvsr_modulated_deform_conv_backward.
extern "C" int rvsr_deform_conv_generic_backward(int dtype, const void* input, const void* weight, const void* offset, const void* mask,
                                                 const void* grad_output, void* grad_input, void* grad_offset, void* grad_mask, void* grad_weight,
                                                 void* grad_bias, int batch, int channels, int height, int width, int channels_out, int kernel_h,
                                                 int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                                 int deformable_group, void* workspace, size_t workspace_bytes, void* stream) {
    if (!input || !weight || !offset || !grad_output) FAIL(RVSR_ERR_BAD_ARG, "deform_conv_generic_backward: null argument");
    if (!mask && grad_mask) FAIL(RVSR_ERR_BAD_ARG, "deform_conv_generic_backward: grad_mask without a mask");
    const GArgs a = {batch, channels, height, width, channels_out, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                     deformable_group};
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case 0: return g_backward<float>(a, (const float*)input, (const float*)weight, (const float*)offset, (const float*)mask, (const float*)grad_output,
                                         (float*)grad_input, (float*)grad_offset, (float*)grad_mask, (float*)grad_weight, (float*)grad_bias, workspace,
                                         workspace_bytes, st);
        case 1: return g_backward<double>(a, (const double*)input, (const double*)weight, (const double*)offset, (const double*)mask,
                                          (const double*)grad_output, (double*)grad_input, (double*)grad_offset, (double*)grad_mask, (double*)grad_weight,
                                          (double*)grad_bias, workspace, workspace_bytes, st);
        case 2: return g_backward<half_t>(a, (const half_t*)input, (const half_t*)weight, (const half_t*)offset, (const half_t*)mask,
                                          (const half_t*)grad_output, (half_t*)grad_input, (half_t*)grad_offset, (half_t*)grad_mask, (half_t*)grad_weight,
                                          (half_t*)grad_bias, workspace, workspace_bytes, st);
        default: FAIL(RVSR_ERR_BAD_ARG, "deform_conv_generic_backward: dtype %d (0 = f32, 1 = f64, 2 = f16)", dtype);
    }
}
