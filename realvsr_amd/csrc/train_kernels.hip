// train_kernels.hip -- the HBM-bound operators either side of netG in one training step (gfx950):
//   * SSIM low-frequency term of LapPyrLoss (loss.py:203,209 -> IQA_pytorch.SSIM, restated in oracle/ssim_oracle.py)
//   * L1 / MSE / Huber / Charbonnier reductions (loss.py:10-40, PyramidLoss loss_mode :167-175)
//   * conv_gauss(img, kernel) and upsample(x) of the pyramid helpers as stand-alone operators (utils/util.py:503-516)
//   * Adam update on a flat parameter buffer (VideoSR_AllPair_model_YCbCr_Split.py:122-124,187: torch.optim.Adam)
//   * CutBlur / channel-permute / blend augmentation of an LQ/GT clip pair in one pass
//     (data/augments_video_allpair.py:6-88, called at VideoSR_AllPair_model_YCbCr_Split.py:169-173)
// One thread per output element, lanes along W (coalesced); reductions accumulate in double and finish in a second
// one-block kernel (deterministic); backward kernels are gathers (no atomics).
#include "rvsr_common.h"

#define GRID_FOR(n) dim3((unsigned)(((n) + 255) / 256 > 4096 ? 4096 : ((n) + 255) / 256))
#define LOOP(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)
#define RED_BLOCKS 1024
#define CHECK_LAUNCH(name)                                                                        \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) FAIL(RVSR_ERR_LAUNCH, name " launch: %s", hipGetErrorString(e_));   \
        return RVSR_OK;                                                                           \
    } while (0)

__device__ __forceinline__ void block_sum_to(double acc, double* __restrict__ partial) {
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// out[0] = bias + scale * sum(partial)
__global__ void affine_finish_kernel(const double* __restrict__ partial, int nb, double bias, double scale,
                                     float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(bias + red[0] * scale);
}

// ---------------------------------------------------------------- element-wise losses
// mode 0: |d|   1: d^2   2: Huber(delta = param): 0.5 q^2 + delta (|d| - q), q = min(|d|, delta)   3: sqrt(d^2 + param)
__device__ __forceinline__ float pix_loss(float d, int mode, float param) {
    const float a = fabsf(d);
    if (mode == 0) return a;
    if (mode == 1) return d * d;
    if (mode == 2) {
        const float q = fminf(a, param);
        return 0.5f * q * q + param * (a - q);
    }
    return sqrtf(d * d + param);
}
__device__ __forceinline__ float pix_loss_grad(float d, int mode, float param) {
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (mode == 0) return sgn;
    if (mode == 1) return 2.f * d;
    if (mode == 2) return fabsf(d) < param ? d : param * sgn;
    return d / sqrtf(d * d + param);
}
__global__ void pix_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n, int mode, float param,
                                    double* __restrict__ partial) {
    double acc = 0.0;
    LOOP(i, n) acc += (double)pix_loss(x[i] - y[i], mode, param);
    block_sum_to(acc, partial);
}
__global__ void pix_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gs,
                                    int mode, float param, float scale, float* __restrict__ gx, size_t n) {
    const float k = gs[0] * scale;
    LOOP(i, n) gx[i] = k * pix_loss_grad(x[i] - y[i], mode, param);
}

// ---------------------------------------------------------------- SSIM
// 11x11 window, sigma 1.5, 'valid' depthwise correlation; the 121 float weights are built on the host exactly as the
// package does (double outer product / sum, rounded to float) and travel as a kernel argument (scalar loads).
struct SsimWin {
    float w[121];
};
#define SSIM_C1 (0.01f * 0.01f)
#define SSIM_C2 (0.03f * 0.03f)

// one thread per window position q: S_q = l * relu(cs); optionally the three partial derivatives the backward needs
//   ga = dS/dmu_x, gb = dS/dE[xx], gc = dS/dE[xy]
__global__ void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t planes, int H, int W,
                                SsimWin win, double* __restrict__ partial, float* __restrict__ ga, float* __restrict__ gb,
                                float* __restrict__ gc) {
    const int Ho = H - 10, Wo = W - 10;
    const size_t n = planes * Ho * Wo;
    double acc = 0.0;
    LOOP(idx, n) {
        const int qx = (int)(idx % Wo);
        const int qy = (int)((idx / Wo) % Ho);
        const size_t base = (idx / ((size_t)Wo * Ho)) * H * W + (size_t)qy * W + qx;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
        for (int i = 0; i < 11; ++i) {
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                const float w = win.w[i * 11 + j];
                const float a = x[base + (size_t)i * W + j], b = y[base + (size_t)i * W + j];
                m1 += w * a;
                m2 += w * b;
                e11 += w * (a * a);
                e22 += w * (b * b);
                e12 += w * (a * b);
            }
        }
        const float A1 = 2.f * m1 * m2 + SSIM_C1, B1 = m1 * m1 + m2 * m2 + SSIM_C1;
        const float A2 = 2.f * (e12 - m1 * m2) + SSIM_C2, B2 = (e11 - m1 * m1) + (e22 - m2 * m2) + SSIM_C2;
        const float l = A1 / B1;
        const float cs_raw = A2 / B2;
        const bool on = cs_raw > 0.f;  // F.relu on the contrast-structure map
        const float cs = on ? cs_raw : 0.f;
        acc += (double)(l * cs);
        if (ga != nullptr) {
            ga[idx] = on ? cs * 2.f * (m2 - l * m1) / B1 + l * 2.f * (cs * m1 - m2) / B2 : 0.f;
            gb[idx] = on ? -l * cs / B2 : 0.f;
            gc[idx] = on ? 2.f * l / B2 : 0.f;
        }
    }
    block_sum_to(acc, partial);
}
// gx[p] = k * sum_q w[p - q] (ga_q + 2 x_p gb_q + y_p gc_q),  k = -gscalar * scale  (loss = 1 - scale * sum S)
__global__ void ssim_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ ga,
                                const float* __restrict__ gb, const float* __restrict__ gc, const float* __restrict__ gs,
                                float scale, SsimWin win, float* __restrict__ gx, size_t planes, int H, int W) {
    const int Ho = H - 10, Wo = W - 10;
    const size_t n = planes * H * W;
    const float k = -gs[0] * scale;
    LOOP(idx, n) {
        const int px = (int)(idx % W);
        const int py = (int)((idx / W) % H);
        const size_t pl = (idx / ((size_t)W * H)) * Ho * Wo;
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int i = 0; i < 11; ++i) {
            const int qy = py - i;
            if (qy < 0 || qy >= Ho) continue;
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                const int qx = px - j;
                const bool ok = qx >= 0 && qx < Wo;
                const size_t o = pl + (size_t)qy * Wo + (ok ? qx : 0);
                const float w = ok ? win.w[i * 11 + j] : 0.f;
                sa += w * ga[o];
                sb += w * gb[o];
                sc += w * gc[o];
            }
        }
        gx[idx] = k * (sa + 2.f * x[idx] * sb + y[idx] * sc);
    }
}

// ---------------------------------------------------------------- conv_gauss / upsample as stand-alone operators
__device__ __forceinline__ int reflect_idx(int q, int n) { return q < 0 ? -q : (q >= n ? 2 * (n - 1) - q : q); }
__device__ __forceinline__ float binom5(int i) { return i == 0 || i == 4 ? 1.f : (i == 2 ? 6.f : 4.f); }

// out[y][x] = gain/256 * sum k[i]k[j] Z[reflect(y+i-2)][reflect(x+j-2)];  up == 0: Z = in (H x W);
// up == 1: Z = zero-insert of in (H/2 x W/2) at the even positions of an H x W grid
__global__ void gauss_full_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, size_t planes, int H, int W,
                                      int up, float gain) {
    const int Hs = up ? H / 2 : H, Ws = up ? W / 2 : W;
    const size_t n = planes * H * W;
    LOOP(idx, n) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const float* p = in + (idx / ((size_t)W * H)) * Hs * Ws;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int r = reflect_idx(y + i - 2, H);
            if (up && (r & 1)) continue;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int s = reflect_idx(x + j - 2, W);
                if (up && (s & 1)) continue;
                acc += (binom5(i) * binom5(j)) * p[(size_t)(up ? r >> 1 : r) * Ws + (up ? s >> 1 : s)];
            }
        }
        out[idx] = acc * (gain * (1.f / 256.f));
    }
}
// adjoint: gin[a][b] = gain/256 * sum over outputs (y, x) in a 5x5 neighbourhood of the taps that land on Z[r][s],
// (r, s) = (a, b) (up == 0) or (2a, 2b) (up == 1).  Reflection keeps |y - r| <= 2.
__global__ void gauss_full_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, size_t planes, int H, int W,
                                      int up, float gain) {
    const int Hs = up ? H / 2 : H, Ws = up ? W / 2 : W;
    const size_t n = planes * Hs * Ws;
    LOOP(idx, n) {
        const int b = (int)(idx % Ws);
        const int a = (int)((idx / Ws) % Hs);
        const float* g = gout + (idx / ((size_t)Ws * Hs)) * H * W;
        const int r = up ? 2 * a : a, s = up ? 2 * b : b;
        float wy[5], wx[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int y = r - 2 + t, x = s - 2 + t;
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (y >= 0 && y < H && reflect_idx(y + i - 2, H) == r) u += binom5(i);
                if (x >= 0 && x < W && reflect_idx(x + i - 2, W) == s) v += binom5(i);
            }
            wy[t] = u;
            wx[t] = v;
        }
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < 5; ++ty) {
            if (wy[ty] == 0.f) continue;
            float rowacc = 0.f;
#pragma unroll
            for (int tx = 0; tx < 5; ++tx)
                if (wx[tx] != 0.f) rowacc += wx[tx] * g[(size_t)(r - 2 + ty) * W + s - 2 + tx];
            acc += wy[ty] * rowacc;
        }
        gin[idx] = acc * (gain * (1.f / 256.f));
    }
}

// ---------------------------------------------------------------- Adam on a flat buffer
// torch.optim.Adam (amsgrad=False, maximize=False), single-tensor op order:
//   g += wd * p; m = m + (g - m)(1 - b1); v = b2 v + (1 - b2) g g; p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 size_t n, float step_size, float beta1, float beta2, float eps, float wd, float bc2_sqrt) {
    const size_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    LOOP(i, n4) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        float* pf = reinterpret_cast<float*>(&pp);
        float* gf = reinterpret_cast<float*>(&gg);
        float* mf = reinterpret_cast<float*>(&mm);
        float* vf = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = wd != 0.f ? gf[j] + wd * pf[j] : gf[j];
            mf[j] = mf[j] + (gr - mf[j]) * (1.f - beta1);
            vf[j] = vf[j] * beta2 + (1.f - beta2) * gr * gr;
            pf[j] = pf[j] - step_size * (mf[j] / (sqrtf(vf[j]) / bc2_sqrt + eps));
        }
        p4[i] = pp;
        m4[i] = mm;
        v4[i] = vv;
    }
    // tail (n not a multiple of 4): first block only
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float gr = wd != 0.f ? g[i] + wd * p[i] : g[i];
        const float mm = m[i] + (gr - m[i]) * (1.f - beta1);
        const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
}

// ---------------------------------------------------------------- clip augmentation
struct AugPlan {
    int perm[3];        // output channel c reads input channel perm[c] (identity unless 'rgb')
    int box_mode;       // 0: none   1: out2 = im2 with the box taken from im1   2: out2 = im1 with the box taken from im2
    int y0, y1, x0, x1; // box [y0, y1) x [x0, x1) in the last two dimensions
    float v;            // blend weight (1 = no blend): out = v * in + (1 - v) * colour[b, n, c]
};
__global__ void augment_clips_kernel(const float* __restrict__ im1, const float* __restrict__ im2, float* __restrict__ out1,
                                     float* __restrict__ out2, const float* __restrict__ colour, size_t frames, int H, int W,
                                     AugPlan plan) {
    const size_t hw = (size_t)H * W, n = frames * 3 * hw;
    LOOP(idx, n) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int c = (int)((idx / hw) % 3);
        const size_t f = idx / (3 * hw);
        const size_t src = (f * 3 + plan.perm[c]) * hw + (size_t)y * W + x;
        const float a = im1[src], b = im2[src];
        const bool inside = y >= plan.y0 && y < plan.y1 && x >= plan.x0 && x < plan.x1;
        float o1 = a, o2 = b;
        if (plan.box_mode == 1) o2 = inside ? a : b;
        if (plan.box_mode == 2) o2 = inside ? b : a;
        if (colour != nullptr) {
            const float col = colour[f * 3 + c];
            o1 = plan.v * o1 + (1.f - plan.v) * col;
            o2 = plan.v * o2 + (1.f - plan.v) * col;
        }
        out1[idx] = o1;
        out2[idx] = o2;
    }
}

// ---------------------------------------------------------------- broadcast add + activation
// out[n][j] = act(a[n][j] + b[j]) for n < N repeats of a `per`-element block (in place on a), and its adjoint w.r.t. b:
// gb[j] = sum_n gout[n][j] * act'(out[n][j]).  Used by the algebraic split conv(cat(x, repeat(ref))) = conv_a(x) + conv_b(ref).
__global__ void bcast_add_act_kernel(float* __restrict__ a, const float* __restrict__ b, size_t per4, int N, int act, float slope) {
    float4* a4 = reinterpret_cast<float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    const float neg = act == 0 ? 1.f : (act == 1 ? 0.f : slope);
    LOOP(j, per4) {
        const float4 bb = b4[j];
        for (int n = 0; n < N; ++n) {
            float4 v = a4[(size_t)n * per4 + j];
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            v.x = v.x > 0.f ? v.x : v.x * neg; v.y = v.y > 0.f ? v.y : v.y * neg;
            v.z = v.z > 0.f ? v.z : v.z * neg; v.w = v.w > 0.f ? v.w : v.w * neg;
            a4[(size_t)n * per4 + j] = v;
        }
    }
}
__global__ void bcast_reduce_act_kernel(const float* __restrict__ gout, const float* __restrict__ out, float* __restrict__ gb,
                                        size_t per4, int N, float gslope, int has_act) {
    const float4* g4 = reinterpret_cast<const float4*>(gout);
    const float4* o4 = reinterpret_cast<const float4*>(out);
    float4* r4 = reinterpret_cast<float4*>(gb);
    LOOP(j, per4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = 0; n < N; ++n) {
            const float4 g = g4[(size_t)n * per4 + j];
            if (has_act) {
                const float4 o = o4[(size_t)n * per4 + j];
                acc.x += g.x * (o.x > 0.f ? 1.f : gslope); acc.y += g.y * (o.y > 0.f ? 1.f : gslope);
                acc.z += g.z * (o.z > 0.f ? 1.f : gslope); acc.w += g.w * (o.w > 0.f ? 1.f : gslope);
            } else {
                acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
            }
        }
        r4[j] = acc;
    }
}

// ---------------------------------------------------------------- host side
extern "C" size_t rvsr_reduce_workspace_bytes() { return RED_BLOCKS * sizeof(double); }

extern "C" int rvsr_pixel_loss_forward(const float* x, const float* y, size_t n, int mode, float param, double scale, float* out,
                                       void* workspace, void* stream) {
    if (!x || !y || !out || !workspace) FAIL(RVSR_ERR_BAD_ARG, "pixel_loss: null argument");
    if (mode < 0 || mode > 3) FAIL(RVSR_ERR_BAD_ARG, "pixel_loss: mode %d (0 l1, 1 l2, 2 huber, 3 charbonnier)", mode);
    unsigned nb = (unsigned)((n + 255) / 256);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(pix_loss_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, n, mode, param, (double*)workspace);
    hipLaunchKernelGGL(affine_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, (int)nb, 0.0,
                       scale, out);
    CHECK_LAUNCH("pixel_loss_fwd");
}
extern "C" int rvsr_pixel_loss_backward(const float* x, const float* y, const float* gscalar, int mode, float param, float scale,
                                        float* gx, size_t n, void* stream) {
    if (!x || !y || !gscalar || !gx) FAIL(RVSR_ERR_BAD_ARG, "pixel_loss backward: null argument");
    if (mode < 0 || mode > 3) FAIL(RVSR_ERR_BAD_ARG, "pixel_loss backward: mode %d", mode);
    hipLaunchKernelGGL(pix_loss_bwd_kernel, GRID_FOR(n), dim3(256), 0, (hipStream_t)stream, x, y, gscalar, mode, param, scale, gx, n);
    CHECK_LAUNCH("pixel_loss_bwd");
}

static const SsimWin& ssim_window() {
    static SsimWin win;
    static bool init = false;
    if (!init) {  // fspecial_gauss(11, 1.5): exp(-(x^2+y^2)/(2 sigma^2)) / sum, in double, rounded to float
        double g[121], sum = 0.0;
        for (int i = 0; i < 11; ++i)
            for (int j = 0; j < 11; ++j) {
                const double dy = i - 5, dx = j - 5;
                g[i * 11 + j] = exp(-(dx * dx + dy * dy) / (2.0 * 1.5 * 1.5));
                sum += g[i * 11 + j];
            }
        for (int i = 0; i < 121; ++i) win.w[i] = (float)(g[i] / sum);
        init = true;
    }
    return win;
}
extern "C" int rvsr_ssim_forward(const float* x, const float* y, size_t planes, int H, int W, double scale, float* out, float* ga,
                                 float* gb, float* gc, void* workspace, void* stream) {
    if (!x || !y || !out || !workspace) FAIL(RVSR_ERR_BAD_ARG, "ssim: null argument");
    if (H < 11 || W < 11) FAIL(RVSR_ERR_BAD_ARG, "ssim: image %dx%d is smaller than the 11x11 window", H, W);
    if ((ga == nullptr) != (gb == nullptr) || (ga == nullptr) != (gc == nullptr))
        FAIL(RVSR_ERR_BAD_ARG, "ssim: ga/gb/gc must be given together");
    const size_t n = planes * (size_t)(H - 10) * (W - 10);
    unsigned nb = (unsigned)((n + 255) / 256);
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, ssim_window(),
                       (double*)workspace, ga, gb, gc);
    hipLaunchKernelGGL(affine_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, (int)nb, 1.0,
                       -scale, out);
    CHECK_LAUNCH("ssim_fwd");
}
extern "C" int rvsr_ssim_backward(const float* x, const float* y, const float* ga, const float* gb, const float* gc,
                                  const float* gscalar, float scale, float* gx, size_t planes, int H, int W, void* stream) {
    if (!x || !y || !ga || !gb || !gc || !gscalar || !gx) FAIL(RVSR_ERR_BAD_ARG, "ssim backward: null argument");
    if (H < 11 || W < 11) FAIL(RVSR_ERR_BAD_ARG, "ssim backward: image smaller than the window");
    const size_t n = planes * (size_t)H * W;
    hipLaunchKernelGGL(ssim_bwd_kernel, GRID_FOR(n), dim3(256), 0, (hipStream_t)stream, x, y, ga, gb, gc, gscalar, scale,
                       ssim_window(), gx, planes, H, W);
    CHECK_LAUNCH("ssim_bwd");
}

extern "C" int rvsr_conv_gauss_forward(const float* in, float* out, size_t planes, int H, int W, float gain, void* stream) {
    if (!in || !out || H < 3 || W < 3) FAIL(RVSR_ERR_BAD_ARG, "conv_gauss: bad argument (reflect padding 2 needs H, W >= 3)");
    const size_t n = planes * (size_t)H * W;
    hipLaunchKernelGGL(gauss_full_fwd_kernel, GRID_FOR(n), dim3(256), 0, (hipStream_t)stream, in, out, planes, H, W, 0, gain);
    CHECK_LAUNCH("conv_gauss_fwd");
}
extern "C" int rvsr_conv_gauss_backward(const float* gout, float* gin, size_t planes, int H, int W, float gain, void* stream) {
    if (!gout || !gin || H < 3 || W < 3) FAIL(RVSR_ERR_BAD_ARG, "conv_gauss backward: bad argument");
    const size_t n = planes * (size_t)H * W;
    hipLaunchKernelGGL(gauss_full_bwd_kernel, GRID_FOR(n), dim3(256), 0, (hipStream_t)stream, gout, gin, planes, H, W, 0, gain);
    CHECK_LAUNCH("conv_gauss_bwd");
}
// in: planes x H x W  ->  out: planes x 2H x 2W
extern "C" int rvsr_pyr_upsample_forward(const float* in, float* out, size_t planes, int H, int W, void* stream) {
    if (!in || !out || H < 2 || W < 2) FAIL(RVSR_ERR_BAD_ARG, "pyr_upsample: bad argument");
    const size_t n = planes * (size_t)H * W * 4;
    hipLaunchKernelGGL(gauss_full_fwd_kernel, GRID_FOR(n), dim3(256), 0, (hipStream_t)stream, in, out, planes, 2 * H, 2 * W, 1, 4.f);
    CHECK_LAUNCH("pyr_upsample_fwd");
}
extern "C" int rvsr_pyr_upsample_backward(const float* gout, float* gin, size_t planes, int H, int W, void* stream) {
    if (!gout || !gin || H < 2 || W < 2) FAIL(RVSR_ERR_BAD_ARG, "pyr_upsample backward: bad argument");
    const size_t n = planes * (size_t)H * W;
    hipLaunchKernelGGL(gauss_full_bwd_kernel, GRID_FOR(n), dim3(256), 0, (hipStream_t)stream, gout, gin, planes, 2 * H, 2 * W, 1, 4.f);
    CHECK_LAUNCH("pyr_upsample_bwd");
}

extern "C" int rvsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float step_size,
                              float beta1, float beta2, float eps, float weight_decay, float bias_correction2_sqrt, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) FAIL(RVSR_ERR_BAD_ARG, "adam_step: null argument");
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        FAIL(RVSR_ERR_BAD_ARG, "adam_step: buffers must be 16-byte aligned");
    if (n == 0) return RVSR_OK;
    hipLaunchKernelGGL(adam_step_kernel, GRID_FOR((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       n, step_size, beta1, beta2, eps, weight_decay, bias_correction2_sqrt);
    CHECK_LAUNCH("adam_step");
}

extern "C" int rvsr_augment_clips(const float* im1, const float* im2, float* out1, float* out2, const float* colour, size_t frames,
                                  int H, int W, int perm0, int perm1, int perm2, int box_mode, int y0, int y1, int x0, int x1,
                                  float v, void* stream) {
    if (!im1 || !im2 || !out1 || !out2) FAIL(RVSR_ERR_BAD_ARG, "augment_clips: null argument");
    if (im1 == out1 || im2 == out2 || im1 == out2 || im2 == out1) FAIL(RVSR_ERR_BAD_ARG, "augment_clips: outputs must not alias inputs");
    const int seen = (1 << perm0) | (1 << perm1) | (1 << perm2);
    if (perm0 < 0 || perm0 > 2 || perm1 < 0 || perm1 > 2 || perm2 < 0 || perm2 > 2 || seen != 7)
        FAIL(RVSR_ERR_BAD_ARG, "augment_clips: (%d, %d, %d) is not a permutation of 0..2", perm0, perm1, perm2);
    if (box_mode < 0 || box_mode > 2) FAIL(RVSR_ERR_BAD_ARG, "augment_clips: box_mode %d", box_mode);
    AugPlan plan;
    plan.perm[0] = perm0;
    plan.perm[1] = perm1;
    plan.perm[2] = perm2;
    plan.box_mode = box_mode;
    plan.y0 = y0;
    plan.y1 = y1;
    plan.x0 = x0;
    plan.x1 = x1;
    plan.v = v;
    const size_t n = frames * 3 * (size_t)H * W;
    hipLaunchKernelGGL(augment_clips_kernel, GRID_FOR(n), dim3(256), 0, (hipStream_t)stream, im1, im2, out1, out2, colour, frames, H,
                       W, plan);
    CHECK_LAUNCH("augment_clips");
}

extern "C" int rvsr_bcast_add_act(float* a, const float* b, size_t per, int N, int act, float slope, void* stream) {
    if (!a || !b || N <= 0) FAIL(RVSR_ERR_BAD_ARG, "bcast_add_act: bad argument");
    if ((per & 3) != 0 || ((((uintptr_t)a) | ((uintptr_t)b)) & 15) != 0) FAIL(RVSR_ERR_BAD_ARG, "bcast_add_act: blocks must be 16-byte aligned multiples of 4 floats");
    hipLaunchKernelGGL(bcast_add_act_kernel, GRID_FOR(per / 4), dim3(256), 0, (hipStream_t)stream, a, b, per / 4, N, act, slope);
    CHECK_LAUNCH("bcast_add_act");
}
extern "C" int rvsr_bcast_reduce_act(const float* gout, const float* out, float* gb, size_t per, int N, float gslope, void* stream) {
    if (!gout || !gb || N <= 0) FAIL(RVSR_ERR_BAD_ARG, "bcast_reduce_act: bad argument");
    if ((per & 3) != 0 || ((((uintptr_t)gout) | ((uintptr_t)out) | ((uintptr_t)gb)) & 15) != 0)
        FAIL(RVSR_ERR_BAD_ARG, "bcast_reduce_act: blocks must be 16-byte aligned multiples of 4 floats");
    hipLaunchKernelGGL(bcast_reduce_act_kernel, GRID_FOR(per / 4), dim3(256), 0, (hipStream_t)stream, gout, out, gb, per / 4, N, gslope, out != nullptr);
    CHECK_LAUNCH("bcast_reduce_act");
}
