// conv_kernels.hip -- 3x3 / 1x1 convolution blocks of the EDVR hot path as implicit GEMMs on the
// gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, see rvsr_common.h).
//
// Replaces, on the hot path, the reference's nn.Conv2d calls + the elementwise ops around them
// (codes/models/archs/EDVR_arch.py:96-132,166-208,256-319; arch_util.py:121-139):
//   * bias + ReLU / LeakyReLU(0.1) + residual add fused in the epilogue,
//   * torch.cat([a, b], 1) never materialised (two input pointers),
//   * nn.PixelShuffle(2) fused into the store (forward) / the load (backward),
//   * activation derivative fused into the gradient load (TView.act),
//   * data gradient = the same kernel with transposed+flipped weight addressing (w_mode 1),
//     stride-2 data gradient through a zero-inserted view (TView mode 1),
//   * weight gradient: persistent workgroups accumulate gW tiles in registers over many pixel
//     tiles, write deterministic partials, a second kernel reduces them (no atomics).
//
// Tiling (forward): one workgroup = 4 waves = 8 x 32 output pixels x (MT * 32) output channels;
// wave w owns rows 2w, 2w+1 (two 32-pixel N tiles) x MT M tiles.  K is walked in chunks of CC
// input channels: the input tile (+halo) and the weight slice [tap][c][o] live in LDS; the B
// operand is read straight out of the input tile at the tap's shifted address, so no im2col
// buffer exists anywhere.
#include "rvsr_common.h"

#include "conv_common.h"

template <int KS, int STRIDE, int MT, int CC>
__global__ __launch_bounds__(RVSR_WG, 2) void conv_fwd_kernel(const ConvFwdParams p) {
    constexpr int T = KS * KS, PAD = KS / 2, TH = 8, TW = 32;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int MP = MT * 32, MPP = MP + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                  // [CC][IH][IW]
    float* ws = smem + CC * IH * IW;   // [T][CC][MPP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int tx = blockIdx.x % p.ntx, ty = blockIdx.x / p.ntx;
    const int x0 = tx * TW, y0 = ty * TH, mb = blockIdx.y, b = blockIdx.z;
    const int Ctot = p.in.a.C + p.in.b.C;

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m][0] = zero16();
        acc[m][1] = zero16();
    }

    for (int c0 = 0; c0 < Ctot; c0 += CC) {
#pragma unroll 2
        for (int e = tid; e < CC * IH * IW; e += RVSR_WG) {
            const int cc = e / (IH * IW);
            const int rem = e - cc * (IH * IW);
            const int r = rem / IW, s = rem - r * IW;
            const int c = c0 + cc;
            float v = 0.f;
            if (c < Ctot) v = tcat_get(p.in, b, c, y0 * STRIDE - PAD + r, x0 * STRIDE - PAD + s);
            xs[e] = v;
        }
        if (p.w_mode == 0) {  // A[o][(tap, c)] = w[o][c][tap]
#pragma unroll 4
            for (int e = tid; e < MP * CC * T; e += RVSR_WG) {
                const int m = e / (CC * T);
                const int rem = e - m * (CC * T);
                const int cc = rem / T, tap = rem - cc * T;
                const int o = mb * MP + m, c = c0 + cc;
                float v = 0.f;
                if (o < p.Co && c < Ctot) v = p.w[((size_t)o * Ctot + c) * T + tap];
                ws[(tap * CC + cc) * MPP + m] = v;
            }
        } else {  // data gradient: A[i][(tap, k)] = w[k][i][T-1-tap], w stored [Ctot][Co][T]
#pragma unroll 4
            for (int e = tid; e < CC * MP * T; e += RVSR_WG) {
                const int cc = e / (MP * T);
                const int rem = e - cc * (MP * T);
                const int m = rem / T, tap = rem - m * T;
                const int i = mb * MP + m, k = c0 + cc;
                float v = 0.f;
                if (i < p.Co && k < Ctot) v = p.w[((size_t)k * p.Co + i) * T + (T - 1 - tap)];
                ws[(tap * CC + cc) * MPP + m] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll 4
            for (int cp = 0; cp < CC / 2; ++cp) {
                const int cc = 2 * cp + hi;
                float a[MT], bv[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = ws[(tap * CC + cc) * MPP + m * 32 + lo];
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    bv[n] = xs[cc * (IH * IW) + ((wave * 2 + n) * STRIDE + dy) * IW + lo * STRIDE + dx];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m][0] = mfma32(a[m], bv[0], acc[m][0]);
                    acc[m][1] = mfma32(a[m], bv[1], acc[m][1]);
                }
            }
        }
        __syncthreads();
    }

    // epilogue: bias + activation (+ residual | pixel-shuffle | channel split), 128-B row segments
    if (p.ps)
        conv_epilogue<MT, 3>(acc, p, b, mb * (MT * 32), y0 + wave * 2, x0 + lo, hi);
    else if (p.out2 != nullptr)
        conv_epilogue<MT, 2>(acc, p, b, mb * (MT * 32), y0 + wave * 2, x0 + lo, hi);
    else if (p.res != nullptr)
        conv_epilogue<MT, 1>(acc, p, b, mb * (MT * 32), y0 + wave * 2, x0 + lo, hi);
    else
        conv_epilogue<MT, 0>(acc, p, b, mb * (MT * 32), y0 + wave * 2, x0 + lo, hi);
}

// ------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int CCW>
__global__ __launch_bounds__(RVSR_WG) void conv_wgrad_kernel(const ConvWgradParams p) {
    constexpr int T = KS * KS, PAD = KS / 2, TH = 4, TW = 32, NPX = TH * TW;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int CS = (IH * IW) | 1;  // odd channel stride: lanes walk channels conflict-free
    constexpr int NTOT = T * CCW, NT = (NTOT + 31) / 32, TPW = (2 * NT + 3) / 4;
    constexpr int GP = 65;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gT = smem;             // [NPX][GP]   gradient tile, pixel-major
    float* xs = smem + NPX * GP;  // [CCW][CS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int mb = blockIdx.y, c0 = blockIdx.z * CCW;
    const int Ctot = p.x.a.C + p.x.b.C;
    const int m = wave & 1;
    const bool m_live = (mb * 64 + m * 32) < p.Co;

    int boff[TPW];
    bool bval[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int nt = (wave >> 1) + 2 * i;
        const int n = nt * 32 + lo;
        bval[i] = (nt < NT) && (n < NTOT);
        const int tap = n / CCW, cc = n - tap * CCW;
        boff[i] = bval[i] ? cc * CS + (tap / KS) * IW + (tap % KS) : 0;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = zero16();
    float bsum = 0.f;

    const int ntiles = p.B * p.nty * p.ntx;
    for (int tile = blockIdx.x; tile < ntiles; tile += p.P) {
        const int b = tile / (p.nty * p.ntx);
        const int trem = tile - b * (p.nty * p.ntx);
        const int ty = trem / p.ntx, tx = trem - ty * p.ntx;
        const int y0 = ty * TH, x0 = tx * TW;
        // plain views (the common case) are staged four elements per batch of loads, see tview_get_batch
        if (p.g.mode == 0) {  // (uniform)
            for (int e0 = tid; e0 < 64 * NPX; e0 += 4 * RVSR_WG) {
                int cc[4], yy[4], xx[4];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j * RVSR_WG;
                    const int ol = e / NPX, px = e - ol * NPX;
                    cc[j] = e < 64 * NPX ? mb * 64 + ol : -1;
                    yy[j] = y0 + (px >> 5);
                    xx[j] = x0 + (px & 31);
                }
                tview_get_batch<4>(p.g, b, cc, yy, xx, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j * RVSR_WG;
                    const int ol = e / NPX, px = e - ol * NPX;
                    if (e < 64 * NPX) gT[px * GP + ol] = v[j];
                }
            }
        } else {
#pragma unroll 2
            for (int e = tid; e < 64 * NPX; e += RVSR_WG) {
                const int ol = e / NPX, px = e - ol * NPX;
                const int o = mb * 64 + ol;
                float v = 0.f;
                if (o < p.Co) v = tview_get(p.g, b, o, y0 + (px >> 5), x0 + (px & 31));
                gT[px * GP + ol] = v;
            }
        }
        if (p.x.b.p == nullptr && p.x.a.mode == 0) {  // (uniform)
            for (int e0 = tid; e0 < CCW * IH * IW; e0 += 4 * RVSR_WG) {
                int cc[4], yy[4], xx[4];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j * RVSR_WG;
                    const int ci = e / (IH * IW), rem = e - ci * (IH * IW);
                    const int r = rem / IW, s = rem - r * IW;
                    cc[j] = e < CCW * IH * IW ? c0 + ci : -1;
                    yy[j] = y0 * STRIDE - PAD + r;
                    xx[j] = x0 * STRIDE - PAD + s;
                }
                tview_get_batch<4>(p.x.a, b, cc, yy, xx, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j * RVSR_WG;
                    const int ci = e / (IH * IW), rem = e - ci * (IH * IW);
                    const int r = rem / IW, s = rem - r * IW;
                    if (e < CCW * IH * IW) xs[ci * CS + r * IW + s] = v[j];
                }
            }
        } else {
#pragma unroll 2
            for (int e = tid; e < CCW * IH * IW; e += RVSR_WG) {
                const int cc = e / (IH * IW);
                const int rem = e - cc * (IH * IW);
                const int r = rem / IW, s = rem - r * IW;
                const int c = c0 + cc;
                float v = 0.f;
                if (c < Ctot) v = tcat_get(p.x, b, c, y0 * STRIDE - PAD + r, x0 * STRIDE - PAD + s);
                xs[cc * CS + r * IW + s] = v;
            }
        }
        __syncthreads();
        if (p.bpart != nullptr && blockIdx.z == 0 && tid < 64) {
            float s = 0.f;
            for (int px = 0; px < NPX; ++px) s += gT[px * GP + tid];
            bsum += s;
        }
        if (m_live) {
#pragma unroll 2
            for (int ks = 0; ks < NPX / 2; ++ks) {
                const int row = ks >> 4, colp = 2 * (ks & 15) + hi;
                const float a = gT[(row * 32 + colp) * GP + m * 32 + lo];
                const int poff = row * STRIDE * IW + colp * STRIDE;
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    if ((wave >> 1) + 2 * i < NT) {
                        const float bv = bval[i] ? xs[boff[i] + poff] : 0.f;
                        acc[i] = mfma32(a, bv, acc[i]);
                    }
                }
            }
        }
        __syncthreads();
    }

    if (m_live) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int nt = (wave >> 1) + 2 * i;
            const int n = nt * 32 + lo;
            if (nt >= NT || n >= NTOT) continue;
            const int tap = n / CCW, c = c0 + (n - tap * CCW);
            if (c >= Ctot) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = mb * 64 + m * 32 + drow(r, hi);
                if (o < p.Co) p.part[(((size_t)blockIdx.x * p.Co + o) * Ctot + c) * T + tap] = acc[i][r];
            }
        }
    }
    if (p.bpart != nullptr && blockIdx.z == 0 && tid < 64) {
        const int o = mb * 64 + tid;
        if (o < p.Co) p.bpart[(size_t)blockIdx.x * p.Co + o] = bsum;
    }
}

// ------------------------------------------------------------------------------------------
// host side


template <int KS, int STRIDE, int MT, int CC>
static int launch_fwd(const ConvFwdParams& p, hipStream_t st) {
    constexpr int T = KS * KS, IH = 7 * STRIDE + KS, IW = 31 * STRIDE + KS;
    const size_t lds = sizeof(float) * (CC * IH * IW + T * CC * (MT * 32 + 1));
    auto k = conv_fwd_kernel<KS, STRIDE, MT, CC>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_fwd: cannot reserve %zu B of LDS", lds);
    const int nty = (p.Hout + 7) / 8;
    dim3 grid(p.ntx * nty, (p.Co + MT * 32 - 1) / (MT * 32), p.B);
    hipLaunchKernelGGL(k, grid, dim3(RVSR_WG), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_fwd launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

static int make_view(TView& v, const float* p, const float* act, float slope, int C, int Hs, int Ws, int mode,
                     int Hv, int Wv) {
    v.p = p;
    v.act = act;
    v.slope = slope;
    v.C = C;
    v.Hs = Hs;
    v.Ws = Ws;
    v.mode = mode;
    if (mode == 0) {
        v.Hv = Hs;
        v.Wv = Ws;
    } else if (mode == 1) {
        v.Hv = Hv;
        v.Wv = Wv;
    } else if (mode == 2) {
        if ((Hs | Ws) & 1 || C % 4) return 1;
        v.Hv = Hs / 2;
        v.Wv = Ws / 2;
    } else {
        return 1;
    }
    return 0;
}

// Fused conv block.  See include/realvsr_hip.h for the contract.
extern "C" int rvsr_conv2d_forward(const float* x1, int C1, const float* x2, int C2, const float* xact,
                                   float xact_slope, int in_mode, int Hs, int Ws, const float* weight,
                                   const float* bias, const float* residual, float* out1, int Co1, float* out2,
                                   int Co2, int B, int ksize, int stride, int w_mode, int act, float slope,
                                   int pixel_shuffle, int Hout, int Wout, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!x1 || !weight || !out1 || B <= 0 || C1 <= 0 || Co1 <= 0) FAIL(RVSR_ERR_BAD_ARG, "conv2d: null/empty argument");
    if ((x2 == nullptr) != (C2 == 0) || (out2 == nullptr) != (Co2 == 0))
        FAIL(RVSR_ERR_BAD_ARG, "conv2d: second input/output pointer and channel count disagree");
    if (ksize != 1 && ksize != 3) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: kernel size %d (1 or 3 supported)", ksize);
    if (stride != 1 && !(stride == 2 && ksize == 3)) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: stride %d", stride);
    if (in_mode != 0 && (x2 != nullptr || stride != 1)) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: view mode with concat/stride");
    if (xact && x2) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: act' fusion with concat input");
    if ((residual || pixel_shuffle) && out2) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: residual/pixel-shuffle with split output");
    if (residual && pixel_shuffle) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: residual with pixel-shuffle");
    ConvFwdParams p;
    if (make_view(p.in.a, x1, xact, xact_slope, C1, Hs, Ws, in_mode, Hout, Wout))
        FAIL(RVSR_ERR_BAD_ARG, "conv2d: bad input view (mode %d, C %d, %dx%d)", in_mode, C1, Hs, Ws);
    make_view(p.in.b, x2, nullptr, 0.f, C2, Hs, Ws, 0, 0, 0);
    const int pad = ksize / 2;
    const int He = (p.in.a.Hv + 2 * pad - ksize) / stride + 1, We = (p.in.a.Wv + 2 * pad - ksize) / stride + 1;
    if (He != Hout || We != Wout) FAIL(RVSR_ERR_BAD_ARG, "conv2d: output size %dx%d, expected %dx%d", Hout, Wout, He, We);
    p.w = weight;
    p.bias = bias;
    p.res = residual;
    p.out1 = out1;
    p.out2 = out2;
    p.Co1 = Co1;
    p.Co = Co1 + Co2;
    if (pixel_shuffle && (p.Co % 4)) FAIL(RVSR_ERR_BAD_ARG, "conv2d: pixel-shuffle needs Co %% 4 == 0");
    p.B = B;
    p.Hout = Hout;
    p.Wout = Wout;
    p.w_mode = w_mode & 1;
    p.prepacked = (w_mode >> 1) & 1;
    p.fmt = (w_mode >> 2) & 1;
    p.act = act;
    p.slope = slope;
    p.ps = pixel_shuffle;
    p.ntx = (Wout + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    p.wpack = nullptr;
    if (act == 3) {
        // out = (conv + bias) * (residual > 0 ? 1 : slope): the data gradient of a layer whose INPUT was an activation output, with that
        // activation's derivative applied on the way out (`residual` = the saved activation output).  Only the 8 x 64-tile kernel has this
        // epilogue; everything else answers "unsupported" and the caller applies the mask on the consumer side as before.
        if (!residual || ksize != 3 || stride != 1 || pixel_shuffle || out2 || rvsr_gemm_mode_now() == 1) return RVSR_ERR_UNSUPPORTED;
        return rvsr_launch_conv_fwd2(p, ksize, stride, workspace, workspace_bytes, st);
    }
    if (rvsr_conv_fwd_thin_ok(p, ksize, stride)) return rvsr_launch_conv_fwd_thin(p, st);   // <= 4 output channels: vector ALU, exact f32
    if (rvsr_gemm_mode_now() != 1 && (in_mode != 1 || ksize == 3) && (x2 == nullptr || C1 % 8 == 0)) {
        const int rc2 = rvsr_launch_conv_fwd2(p, ksize, stride, workspace, workspace_bytes, st);
        if (rc2 != RVSR_ERR_UNSUPPORTED) return rc2;   // (sizes beyond the buffer-addressed kernels: exact-f32 kernels below)
    }
    const int mt = p.Co <= 32 ? 1 : (p.Co <= 64 ? 2 : 4);
#define DISPATCH(KS, S, CC12, CC4)                                \
    do {                                                           \
        if (mt == 1) return launch_fwd<KS, S, 1, CC12>(p, st);     \
        if (mt == 2) return launch_fwd<KS, S, 2, CC12>(p, st);     \
        return launch_fwd<KS, S, 4, CC4>(p, st);                   \
    } while (0)
    if (ksize == 3 && stride == 1) DISPATCH(3, 1, 16, 8);
    if (ksize == 3 && stride == 2) DISPATCH(3, 2, 8, 4);
    DISPATCH(1, 1, 32, 32);
#undef DISPATCH
}

int rvsr_g_gemm_mode = 0;                  // process-wide default
thread_local int rvsr_t_gemm_mode = -1;    // the calling host thread's own choice (-1: follow the default)
extern "C" void rvsr_set_gemm_mode(int mode) { rvsr_g_gemm_mode = (mode >= 0 && mode <= 3) ? mode : 0; }
extern "C" void rvsr_set_gemm_mode_thread(int mode) { rvsr_t_gemm_mode = (mode >= 0 && mode <= 3) ? mode : -1; }
extern "C" int rvsr_get_gemm_mode() { return rvsr_gemm_mode_now(); }
extern "C" size_t rvsr_conv2d_forward_workspace_bytes(int C1, int C2, int Co, int ksize) {
    return rvsr_conv_fwd2_workspace_bytes(ksize, Co, C1 + C2);
}

static int wgrad_P(int ntiles, int gy, int gz, int ksize) {
    // 1x1: the GEMM kernel runs 4 small workgroups per CU and hides its load latency with occupancy
    int P = (ksize == 1 ? 1024 : 256) / (gy * gz);
    if (P < 1) P = 1;
    if (P > ntiles) P = ntiles;
    return P;
}
// stride-2 bf16x3 kernel: 2 workgroups of 4 waves per CU, pixels cut into 16-pixel units
static int wgrad_s2_P(int B, int Hout, int Wout, int gy, int gz64) {
    long units = (long)B * Hout * ((Wout + 15) / 16);
    int P = 512 / (gy * gz64);
    if (P < 1) P = 1;
    if (P > units) P = (int)units;
    return P;
}
static void wgrad_geom(int ksize, int stride, int Co, int Ctot, int& ccw, int& gy, int& gz) {
    ccw = (ksize == 3 && stride == 2) ? 32 : 64;
    gy = (Co + 63) / 64;
    gz = (Ctot + ccw - 1) / ccw;
}

extern "C" size_t rvsr_conv2d_wgrad_workspace_bytes(int C1, int C2, int Co, int B, int ksize, int stride, int Hout,
                                                    int Wout) {
    int ccw, gy, gz;
    wgrad_geom(ksize, stride, Co, C1 + C2, ccw, gy, gz);
    const int ntiles = B * ((Hout + 3) / 4) * ((Wout + 31) / 32);
    size_t P = wgrad_P(ntiles, gy, gz, ksize);
    if (ksize == 3 && stride == 2) {  // the bf16x3 stride-2 kernel slices the pixels differently
        const size_t P2 = wgrad_s2_P(B, Hout, Wout, gy, (C1 + C2 + 63) / 64);
        if (P2 > P) P = P2;
    }
    if (ksize == 3 && stride == 1 && Co <= 4) {  // the thin-layer kernel keeps one partial per workgroup of its own grid
        const size_t P3 = rvsr_conv_wgrad_thin_P(B, Hout, Wout);
        if (P3 > P) P = P3;
    }
    return sizeof(float) * P * ((size_t)Co * (C1 + C2) * ksize * ksize + Co);
}

template <int KS, int STRIDE, int CCW>
static int launch_wgrad(const ConvWgradParams& p, int gy, int gz, hipStream_t st) {
    constexpr int IH = 3 * STRIDE + KS, IW = 31 * STRIDE + KS, CS = (IH * IW) | 1;
    const size_t lds = sizeof(float) * (128 * 65 + CCW * CS);
    auto k = conv_wgrad_kernel<KS, STRIDE, CCW>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(k, dim3(p.P, gy, gz), dim3(RVSR_WG), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

extern "C" int rvsr_conv2d_backward_weight(const float* x1, int C1, const float* x2, int C2, int Hin, int Win,
                                           const float* gout, const float* gact, float gact_slope, int g_mode,
                                           int Gs_h, int Gs_w, float* grad_weight, float* grad_bias, int Co, int B,
                                           int ksize, int stride, int Hout, int Wout, int accumulate, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    if (!x1 || !gout || !grad_weight || B <= 0) FAIL(RVSR_ERR_BAD_ARG, "conv2d_backward_weight: null/empty argument");
    if ((x2 == nullptr) != (C2 == 0)) FAIL(RVSR_ERR_BAD_ARG, "conv2d_backward_weight: x2/C2 disagree");
    if (ksize != 1 && ksize != 3) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d_backward_weight: kernel size %d", ksize);
    if (stride != 1 && !(stride == 2 && ksize == 3)) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d_backward_weight: stride %d", stride);
    if (g_mode != 0 && g_mode != 2) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d_backward_weight: gradient view mode %d", g_mode);
    const size_t need = rvsr_conv2d_wgrad_workspace_bytes(C1, C2, Co, B, ksize, stride, Hout, Wout);
    if (!workspace || workspace_bytes < need)
        FAIL(RVSR_ERR_WORKSPACE, "conv2d_backward_weight: workspace %zu B < %zu B", workspace_bytes, need);
    ConvWgradParams p;
    make_view(p.x.a, x1, nullptr, 0.f, C1, Hin, Win, 0, 0, 0);
    make_view(p.x.b, x2, nullptr, 0.f, C2, Hin, Win, 0, 0, 0);
    if (make_view(p.g, gout, gact, gact_slope, Co, Gs_h, Gs_w, g_mode, 0, 0) || p.g.Hv != Hout || p.g.Wv != Wout)
        FAIL(RVSR_ERR_BAD_ARG, "conv2d_backward_weight: gradient view %dx%d (mode %d) vs output %dx%d", Gs_h, Gs_w, g_mode,
             Hout, Wout);
    const int pad = ksize / 2;
    if ((Hin + 2 * pad - ksize) / stride + 1 != Hout || (Win + 2 * pad - ksize) / stride + 1 != Wout)
        FAIL(RVSR_ERR_BAD_ARG, "conv2d_backward_weight: input %dx%d does not give output %dx%d", Hin, Win, Hout, Wout);
    int ccw, gy, gz;
    const int Ctot = C1 + C2;
    wgrad_geom(ksize, stride, Co, Ctot, ccw, gy, gz);
    p.B = B;
    p.Co = Co;
    p.Hout = Hout;
    p.Wout = Wout;
    p.ntx = (Wout + 31) / 32;
    p.nty = (Hout + 3) / 4;
    p.P = wgrad_P(B * p.nty * p.ntx, gy, gz, ksize);
    p.ring = 0;   // (the X-row ring of conv_wgrad2 measured no gain, profiles/r04_notes.md: compiled out, WGRAD2_RING in conv2_kernels.hip)
    const size_t nw = (size_t)Co * Ctot * ksize * ksize;
    p.part = (float*)workspace;
    p.bpart = grad_bias ? p.part + (size_t)p.P * nw : nullptr;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const bool aligned16 = ((((uintptr_t)x1) | ((uintptr_t)x2) | ((uintptr_t)gout) | ((uintptr_t)gact)) & 15) == 0;
    if (ksize == 3 && stride == 1 && Co <= 4 && C2 == 0 && C1 % 16 == 0 && g_mode == 0 && (Wout % 4) == 0 && aligned16 &&
        sizeof(float) * (size_t)Hout * Wout * (size_t)C1 < ((size_t)1 << 31)) {
        // thin layer (conv_last): vector-ALU kernel, exact f32 in both GEMM modes
        p.P = rvsr_conv_wgrad_thin_P(B, Hout, Wout);
        p.bpart = grad_bias ? p.part + (size_t)p.P * nw : nullptr;
        rc = rvsr_launch_conv_wgrad_thin(p, st);
        if (rc) return rc;
        rvsr_launch_reduce(p.part, p.P, nw, grad_weight, accumulate, st, p.bpart, (size_t)Co, grad_bias);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad reduce launch: %s", hipGetErrorString(e2));
        return RVSR_OK;
    }
    // conv_wgrad2 addresses one image of each tensor with 32-bit byte offsets (raw buffers, < 2 GB) and picks the input per
    // 64-channel block: a second input has to start on a multiple of 64 channels
    const size_t img_max = sizeof(float) * (size_t)Hout * Wout * (size_t)(Co > Ctot ? Co : Ctot);
    if (rvsr_gemm_mode_now() != 1 && ksize == 3 && stride == 1 && (Wout % 4) == 0 && aligned16 && img_max < ((size_t)1 << 31) &&
        (C2 == 0 || C1 % 64 == 0))
        rc = rvsr_launch_conv_wgrad2(p, gy, gz, st);
    else if (rvsr_gemm_mode_now() != 1 && ksize == 1 && g_mode == 0 && ((Hout * Wout) % 8) == 0 && aligned16)
        rc = rvsr_launch_conv_wgrad1x1(p, gy, gz, st);
    else if (ksize == 3 && stride == 1)
        rc = launch_wgrad<3, 1, 64>(p, gy, gz, st);
    else if (ksize == 3 && rvsr_gemm_mode_now() != 1 && x2 == nullptr && g_mode == 0 && (Wout % 8) == 0 && (Win % 4) == 0 && aligned16) {
        const int gz64 = (Ctot + 63) / 64;
        p.P = wgrad_s2_P(B, Hout, Wout, gy, gz64);
        p.bpart = grad_bias ? p.part + (size_t)p.P * nw : nullptr;
        rc = rvsr_launch_conv_wgrad_s2(p, gy, gz64, st);
    } else if (ksize == 3)
        rc = launch_wgrad<3, 2, 32>(p, gy, gz, st);
    else
        rc = launch_wgrad<1, 1, 64>(p, gy, gz, st);
    if (rc) return rc;
    rvsr_launch_reduce(p.part, p.P, nw, grad_weight, accumulate, st, p.bpart, (size_t)Co, grad_bias);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad reduce launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}
