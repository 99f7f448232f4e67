/*
 * oracle/dcn_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement of the reference's modulated deformable convolution (DCNv2)
 * device kernels.  The reference has no CPU implementation of this operator
 * (codes/models/archs/dcn/deform_conv.py:109-110 raises NotImplementedError for
 * non-CUDA tensors), so this file restates the arithmetic of
 *   codes/models/archs/dcn/src/deform_conv_cuda_kernel.cu
 * in plain C, one function per reference kernel:
 *
 *   oracle_modulated_im2col       <- modulated_deformable_im2col_gpu_kernel      (kernel.cu:571-633)
 *   oracle_modulated_col2im       <- modulated_deformable_col2im_gpu_kernel      (kernel.cu:636-693)
 *   oracle_modulated_col2im_coord <- modulated_deformable_col2im_coord_gpu_kernel(kernel.cu:696-767)
 *   bilinear_zero_outside         <- dmcn_im2col_bilinear                        (kernel.cu:467-497)
 *   corner_weight                 <- dmcn_get_gradient_weight                    (kernel.cu:499-524)
 *   coordinate_weight             <- dmcn_get_coordinate_weight                  (kernel.cu:526-568)
 *
 * The host-side loop over samples and the three GEMMs (deform_conv_cuda.cpp:490-685)
 * are restated in oracle/dcn_oracle.py on top of torch CPU matmuls.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The file is compiled twice (REAL=float -> *_f32 symbols, REAL=double -> *_f64) so the
 * same restatement can be finite-difference checked in double precision.
 *
 * Parity pin: the reference ships no tests or golden vectors;
 * this oracle is pinned by (i) known-answer identities and (ii) f64 finite differences
 * (tests/test_oracle_dcn.py), and it is the DCN plugged into the *imported* reference
 * Python when tests/golden/make_golden.py generates the committed fixtures.
 */
#include <math.h>
#include <stddef.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUFFIX
#define SUFFIX _f32
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* Sample plane[h][w] at fractional (y, x); taps outside [0,H-1]x[0,W-1] read as 0.
 * (kernel.cu:467-497: floor for the low corner, each corner guarded separately) */
static REAL bilinear_zero_outside(const REAL *plane, int H, int W, REAL y, REAL x)
{
    const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);
    const int y1 = y0 + 1, x1 = x0 + 1;
    const REAL ly = y - (REAL)y0, lx = x - (REAL)x0;
    const REAL hy = (REAL)1 - ly, hx = (REAL)1 - lx;
    REAL v00 = 0, v01 = 0, v10 = 0, v11 = 0;
    if (y0 >= 0 && x0 >= 0) v00 = plane[(size_t)y0 * W + x0];
    if (y0 >= 0 && x1 <= W - 1) v01 = plane[(size_t)y0 * W + x1];
    if (y1 <= H - 1 && x0 >= 0) v10 = plane[(size_t)y1 * W + x0];
    if (y1 <= H - 1 && x1 <= W - 1) v11 = plane[(size_t)y1 * W + x1];
    return hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11;
}

/* d(sample)/d(plane[h][w]) for a sample at (y, x) (kernel.cu:499-524). */
static REAL corner_weight(REAL y, REAL x, int h, int w, int H, int W)
{
    if (y <= -1 || y >= H || x <= -1 || x >= W) return 0;
    const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);
    const int y1 = y0 + 1, x1 = x0 + 1;
    REAL wt = 0;
    if (h == y0 && w == x0) wt = ((REAL)h + 1 - y) * ((REAL)w + 1 - x);
    if (h == y0 && w == x1) wt = ((REAL)h + 1 - y) * (x + 1 - (REAL)w);
    if (h == y1 && w == x0) wt = (y + 1 - (REAL)h) * ((REAL)w + 1 - x);
    if (h == y1 && w == x1) wt = (y + 1 - (REAL)h) * (x + 1 - (REAL)w);
    return wt;
}

/* d(sample)/d(y) (dir 0) or d(sample)/d(x) (dir 1) (kernel.cu:526-568). */
static REAL coordinate_weight(REAL y, REAL x, int H, int W, const REAL *plane, int dir)
{
    if (y <= -1 || y >= H || x <= -1 || x >= W) return 0;
    const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);
    const int y1 = y0 + 1, x1 = x0 + 1;
    REAL wt = 0;
    if (dir == 0) {
        if (y0 >= 0 && x0 >= 0) wt += -1 * ((REAL)x0 + 1 - x) * plane[(size_t)y0 * W + x0];
        if (y0 >= 0 && x1 <= W - 1) wt += -1 * (x - (REAL)x0) * plane[(size_t)y0 * W + x1];
        if (y1 <= H - 1 && x0 >= 0) wt += ((REAL)x0 + 1 - x) * plane[(size_t)y1 * W + x0];
        if (y1 <= H - 1 && x1 <= W - 1) wt += (x - (REAL)x0) * plane[(size_t)y1 * W + x1];
    } else {
        if (y0 >= 0 && x0 >= 0) wt += -1 * ((REAL)y0 + 1 - y) * plane[(size_t)y0 * W + x0];
        if (y0 >= 0 && x1 <= W - 1) wt += ((REAL)y0 + 1 - y) * plane[(size_t)y0 * W + x1];
        if (y1 <= H - 1 && x0 >= 0) wt += -1 * (y - (REAL)y0) * plane[(size_t)y1 * W + x0];
        if (y1 <= H - 1 && x1 <= W - 1) wt += (y - (REAL)y0) * plane[(size_t)y1 * W + x1];
    }
    return wt;
}

/* One image: x[C][H][W], offset[dg*2*K][Ho][Wo], mask[dg*K][Ho][Wo] -> col[C*K][Ho*Wo].
 * Row order c*K + (i*kw + j) (kernel.cu:589,597,627-628); offset channel g*2K+2k = dy,
 * +1 = dx; mask channel g*K+k (kernel.cu:602-613). */
void FN(oracle_modulated_im2col)(const REAL *x, const REAL *offset, const REAL *mask,
                                 int C, int H, int W, int Ho, int Wo, int kh, int kw,
                                 int pad_h, int pad_w, int stride_h, int stride_w,
                                 int dil_h, int dil_w, int dg, REAL *col)
{
    const int K = kh * kw, cpg = C / dg;
    const size_t HWo = (size_t)Ho * Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c) {
        for (int ho = 0; ho < Ho; ++ho) {
            const int g = c / cpg;
            const REAL *plane = x + (size_t)c * H * W;
            const REAL *off_g = offset + (size_t)g * 2 * K * HWo;
            const REAL *msk_g = mask + (size_t)g * K * HWo;
            for (int wo = 0; wo < Wo; ++wo) {
                const size_t p = (size_t)ho * Wo + wo;
                for (int i = 0; i < kh; ++i)
                    for (int j = 0; j < kw; ++j) {
                        const int k = i * kw + j;
                        const REAL dy = off_g[(size_t)(2 * k) * HWo + p];
                        const REAL dx = off_g[(size_t)(2 * k + 1) * HWo + p];
                        const REAL m = msk_g[(size_t)k * HWo + p];
                        const REAL y = (REAL)(ho * stride_h - pad_h + i * dil_h) + dy;
                        const REAL xx = (REAL)(wo * stride_w - pad_w + j * dil_w) + dx;
                        REAL v = 0;
                        if (y > -1 && xx > -1 && y < H && xx < W)
                            v = bilinear_zero_outside(plane, H, W, y, xx);
                        col[((size_t)c * K + k) * HWo + p] = v * m;
                    }
            }
        }
    }
}

/* grad_x[C][H][W] += scatter(col_grad * mask) (kernel.cu:636-693).  The reference scans a
 * 5x5 window around the truncated position and keeps cells with |delta| < 1 whose
 * corner_weight is non-zero; that visits exactly the <=4 floor/floor+1 corners.  The scan
 * is kept literally so truncation-vs-floor corner cases for negative positions match.
 * Parallel over channels -> no write conflicts, deterministic summation order. */
void FN(oracle_modulated_col2im)(const REAL *col_grad, const REAL *offset, const REAL *mask,
                                 int C, int H, int W, int Ho, int Wo, int kh, int kw,
                                 int pad_h, int pad_w, int stride_h, int stride_w,
                                 int dil_h, int dil_w, int dg, REAL *grad_x)
{
    const int K = kh * kw, cpg = C / dg;
    const size_t HWo = (size_t)Ho * Wo;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const int g = c / cpg;
        const REAL *off_g = offset + (size_t)g * 2 * K * HWo;
        const REAL *msk_g = mask + (size_t)g * K * HWo;
        REAL *gplane = grad_x + (size_t)c * H * W;
        for (int k = 0; k < K; ++k) {
            const int i = k / kw, j = k % kw;
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    const size_t p = (size_t)ho * Wo + wo;
                    const REAL dy = off_g[(size_t)(2 * k) * HWo + p];
                    const REAL dx = off_g[(size_t)(2 * k + 1) * HWo + p];
                    const REAL m = msk_g[(size_t)k * HWo + p];
                    const REAL y = (REAL)(ho * stride_h - pad_h + i * dil_h) + dy;
                    const REAL xx = (REAL)(wo * stride_w - pad_w + j * dil_w) + dx;
                    const REAL top = col_grad[((size_t)c * K + k) * HWo + p] * m;
                    const int cy = (int)y, cx = (int)xx; /* truncation, as in the reference */
                    for (int ddy = -2; ddy <= 2; ++ddy)
                        for (int ddx = -2; ddx <= 2; ++ddx) {
                            const int hh = cy + ddy, ww = cx + ddx;
                            if (hh >= 0 && hh < H && ww >= 0 && ww < W &&
                                fabs((double)(y - (REAL)hh)) < 1 && fabs((double)(xx - (REAL)ww)) < 1) {
                                gplane[(size_t)hh * W + ww] += corner_weight(y, xx, hh, ww, H, W) * top;
                            }
                        }
                }
        }
    }
}

/* grad_offset[dg*2K][Ho][Wo], grad_mask[dg*K][Ho][Wo] (kernel.cu:696-767).  For offset
 * channel (g, 2k+dir): sum over the cpg channels of group g of
 * coordinate_weight(dir) * col_grad[c*K+k] * mask; out-of-range samples contribute 0
 * (sentinel -2 in the reference).  grad_mask[g,k] = sum_c col_grad[c*K+k] * bilinear(x[c])
 * over in-range samples only, written by the dir==0 pass. */
void FN(oracle_modulated_col2im_coord)(const REAL *col_grad, const REAL *x, const REAL *offset,
                                       const REAL *mask, int C, int H, int W, int Ho, int Wo,
                                       int kh, int kw, int pad_h, int pad_w, int stride_h,
                                       int stride_w, int dil_h, int dil_w, int dg,
                                       REAL *grad_offset, REAL *grad_mask)
{
    const int K = kh * kw, cpg = C / dg;
    const size_t HWo = (size_t)Ho * Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int g = 0; g < dg; ++g) {
        for (int k = 0; k < K; ++k) {
            const int i = k / kw, j = k % kw;
            const REAL *off_g = offset + (size_t)g * 2 * K * HWo;
            const REAL *msk_g = mask + (size_t)g * K * HWo;
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    const size_t p = (size_t)ho * Wo + wo;
                    const REAL dy = off_g[(size_t)(2 * k) * HWo + p];
                    const REAL dx = off_g[(size_t)(2 * k + 1) * HWo + p];
                    const REAL m = msk_g[(size_t)k * HWo + p];
                    REAL y = (REAL)(ho * stride_h - pad_h + i * dil_h) + dy;
                    REAL xx = (REAL)(wo * stride_w - pad_w + j * dil_w) + dx;
                    const int inside = !(y <= -1 || xx <= -1 || y >= H || xx >= W);
                    if (!inside) y = xx = -2;
                    REAL gy = 0, gx = 0, gm = 0;
                    for (int cc = 0; cc < cpg; ++cc) {
                        const int c = g * cpg + cc;
                        const REAL *plane = x + (size_t)c * H * W;
                        const REAL cg = col_grad[((size_t)c * K + k) * HWo + p];
                        if (inside) gm += cg * bilinear_zero_outside(plane, H, W, y, xx);
                        gy += coordinate_weight(y, xx, H, W, plane, 0) * cg * m;
                        gx += coordinate_weight(y, xx, H, W, plane, 1) * cg * m;
                    }
                    grad_offset[((size_t)g * 2 * K + 2 * k) * HWo + p] = gy;
                    grad_offset[((size_t)g * 2 * K + 2 * k + 1) * HWo + p] = gx;
                    grad_mask[((size_t)g * K + k) * HWo + p] = gm;
                }
        }
    }
}
