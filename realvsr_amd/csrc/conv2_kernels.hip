// conv2_kernels.hip -- convolution blocks on the bf16 matrix cores with a 3-term bf16 split
// ("bf16x3"): every f32 operand a is split into a_hi = bf16(a), a_lo = bf16(a - a_hi) and the GEMM
// accumulates a_hi*b_hi + a_hi*b_lo + a_lo*b_hi in f32 (v_mfma_f32_32x32x16_bf16).  Relative error
// per product ~2^-17 (f32 is 2^-24, plain bf16 2^-9) at 3/16 of the f32-MFMA cost, which moves the
// conv blocks from MFMA-bound to roughly HBM/MFMA-balanced.
//
// Same tiling, fusions and epilogue as conv_kernels.hip; what changes is the data path:
//   * weights are packed once per call by `pack_weights_kernel` into the exact LDS image the
//     kernel wants ([m-block][chunk][hi|lo][tap][octet][row][8] bf16), so weight staging is a
//     linear 16-byte-per-lane copy;
//   * the input tile is converted to bf16 hi/lo while it is staged and stored channel-octet-major
//     ([octet][row][col][8]), so every MFMA operand fragment (8 consecutive k per lane) is one
//     conflict-free ds_read_b128;
//   * K order is (tap, channel): one k-step = 16 channels at one tap.
// MFMA operand maps (gfx950, 32x32x16): A[i = l&31][k = 8*(l>>5) + 0..7], B[k = 8*(l>>5) + 0..7][j = l&31],
// D as in rvsr_common.h.
#include "conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
}

// ------------------------------------------------------------------------------------------
// packed[mb][chunk][part][tap][oc][m][8]: part 0 = hi, 1 = lo; oc < 2*CCG octets of the chunk;
// m < MP rows of m-block mb.   mode 0: A[o][(tap,c)] = w[o][c][tap]  (w: [Co][Ctot][T])
//                               mode 1: A[i][(tap,k)] = w[k][i][T-1-tap]  (w: [Ctot][Co][T])
__global__ void pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int Ctot, int T,
                                    int MP, int CCG, int nchunks, int nmb, int mode) {
    const int noct = 2 * CCG;
    const size_t total = (size_t)nmb * nchunks * T * noct * MP;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx % MP);
        size_t r = idx / MP;
        const int oc = (int)(r % noct);
        r /= noct;
        const int tap = (int)(r % T);
        r /= T;
        const int chunk = (int)(r % nchunks);
        const int mb = (int)(r / nchunks);
        const int o = mb * MP + m, cb = (chunk * noct + oc) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb + j;
            float x = 0.f;
            if (o < Co && c < Ctot)
                x = mode == 0 ? w[((size_t)o * Ctot + c) * T + tap] : w[((size_t)c * Co + o) * T + (T - 1 - tap)];
            v[j] = x;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const size_t blk = ((size_t)mb * nchunks + chunk) * 2;  // hi block, lo block follows
        const size_t inner = ((size_t)tap * noct + oc) * MP + m;
        const size_t per = (size_t)T * noct * MP;
        packed[blk * per + inner] = hi;
        packed[(blk + 1) * per + inner] = lo;
    }
}

// ------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int MT, int CCG>
__global__ __launch_bounds__(RVSR_WG, 2) void conv_fwd2_kernel(const ConvFwdParams p) {
    constexpr int T = KS * KS, PAD = KS / 2, TH = 8, TW = 32;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int MP = MT * 32, NOCT = 2 * CCG, NPOS = IH * IW;
    constexpr int WVEC = T * NOCT * MP;  // 16-byte vectors per weight part (hi or lo)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* xs_hi = reinterpret_cast<bf16x8*>(smem_raw);  // [NOCT][IH][IW]
    bf16x8* xs_lo = xs_hi + NOCT * NPOS;
    bf16x8* ws_hi = xs_lo + NOCT * NPOS;                  // [T][NOCT][MP]
    bf16x8* ws_lo = ws_hi + WVEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int tx = blockIdx.x % p.ntx, ty = blockIdx.x / p.ntx;
    const int x0 = tx * TW, y0 = ty * TH, mb = blockIdx.y, b = blockIdx.z;
    const TView& va = p.in.a;
    const TView& vb = p.in.b;
    const int C1 = va.C, Ctot = va.C + vb.C;
    const int nchunks = (Ctot + 16 * CCG - 1) / (16 * CCG);
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)mb * nchunks * 2 * WVEC;

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m][0] = zero16();
        acc[m][1] = zero16();
    }

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * 16 * CCG;
        // ---- weights: linear copy of the pre-packed LDS image (hi block then lo block)
        {
            const bf16x8* src = wsrc + (size_t)chunk * 2 * WVEC;
#pragma unroll 4
            for (int e = tid; e < 2 * WVEC; e += RVSR_WG) ws_hi[e] = src[e];
        }
        // ---- input tile: f32 NCHW -> bf16 hi/lo, [octet][row][col][8]
        for (int it = tid; it < NOCT * NPOS; it += RVSR_WG) {
            const int oc = it / NPOS, pos = it - oc * NPOS;
            const int r = pos / IW, s = pos - r * IW;
            const int gy = y0 * STRIDE - PAD + r, gx = x0 * STRIDE - PAD + s;
            const int cb = c0 + oc * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (gy >= 0 && gx >= 0 && gy < va.Hv && gx < va.Wv && cb < Ctot) {
                if (va.mode == 0) {
                    const bool first = cb < C1;  // octets never straddle the two inputs (C1 % 8 == 0)
                    const TView& v0 = first ? va : vb;
                    const int cl = first ? cb : cb - C1;
                    const size_t hw = (size_t)v0.Hs * v0.Ws;
                    const size_t base = ((size_t)b * v0.C + cl) * hw + (size_t)gy * v0.Ws + gx;
                    const int nvalid = v0.C - cl;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (j < nvalid) v[j] = v0.p[base + j * hw];
                    if (v0.act != nullptr) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j < nvalid) v[j] *= (v0.act[base + j * hw] > 0.f ? 1.f : v0.slope);
                    }
                } else {  // mode 2: pixel-unshuffle view, virtual channel c -> stored (c>>2, 2y+((c>>1)&1), 2x+(c&1))
                    const size_t hw = (size_t)va.Hs * va.Ws;
                    const size_t base = ((size_t)b * (va.C >> 2) + (cb >> 2)) * hw + (size_t)(2 * gy) * va.Ws + 2 * gx;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const size_t idx = base + (j >> 2) * hw + ((j >> 1) & 1) * va.Ws + (j & 1);
                        float x = va.p[idx];
                        if (va.act != nullptr) x *= (va.act[idx] > 0.f ? 1.f : va.slope);
                        v[j] = x;
                    }
                }
            }
            bf16x8 h8, l8;
            split8(v, h8, l8);
            xs_hi[it] = h8;
            xs_lo[it] = l8;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll
            for (int g = 0; g < CCG; ++g) {
                const int oc = 2 * g + hi;
                bf16x8 ah[MT], al[MT], bh[2], bl[2];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[m] = ws_hi[(tap * NOCT + oc) * MP + m * 32 + lo];
                    al[m] = ws_lo[(tap * NOCT + oc) * MP + m * 32 + lo];
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int idx = (oc * IH + (wave * 2 + n) * STRIDE + dy) * IW + lo * STRIDE + dx;
                    bh[n] = xs_hi[idx];
                    bl[n] = xs_lo[idx];
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        acc[m][n] = mfma_bf16(ah[m], bh[n], acc[m][n]);
                        acc[m][n] = mfma_bf16(ah[m], bl[n], acc[m][n]);
                        acc[m][n] = mfma_bf16(al[m], bh[n], acc[m][n]);
                    }
            }
        }
        __syncthreads();
    }

    if (p.ps)
        conv_epilogue<MT, 3>(acc, p, b, mb * MP, y0 + wave * 2, x0 + lo, hi);
    else if (p.out2 != nullptr)
        conv_epilogue<MT, 2>(acc, p, b, mb * MP, y0 + wave * 2, x0 + lo, hi);
    else if (p.res != nullptr)
        conv_epilogue<MT, 1>(acc, p, b, mb * MP, y0 + wave * 2, x0 + lo, hi);
    else
        conv_epilogue<MT, 0>(acc, p, b, mb * MP, y0 + wave * 2, x0 + lo, hi);
}

// ------------------------------------------------------------------------------------------
// host side (called from conv_kernels.hip)
static void fwd2_geom(int ksize, int Co, int Ctot, int& mt, int& ccg, int& nchunks, int& nmb) {
    mt = Co <= 32 ? 1 : (Co <= 64 ? 2 : 4);
    ccg = ksize == 3 ? 1 : 2;
    nchunks = (Ctot + 16 * ccg - 1) / (16 * ccg);
    nmb = (Co + mt * 32 - 1) / (mt * 32);
}

size_t rvsr_conv_fwd2_workspace_bytes(int ksize, int Co, int Ctot) {
    int mt, ccg, nchunks, nmb;
    fwd2_geom(ksize, Co, Ctot, mt, ccg, nchunks, nmb);
    return (size_t)nmb * nchunks * 2 * (ksize * ksize) * (2 * ccg) * (mt * 32) * 16;
}

template <int KS, int STRIDE, int MT, int CCG>
static int launch_fwd2(const ConvFwdParams& p, hipStream_t st) {
    constexpr int T = KS * KS, IH = 7 * STRIDE + KS, IW = 31 * STRIDE + KS;
    const size_t lds = (size_t)16 * (2 * (2 * CCG) * IH * IW + 2 * T * (2 * CCG) * (MT * 32));
    auto k = conv_fwd2_kernel<KS, STRIDE, MT, CCG>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_fwd2: cannot reserve %zu B of LDS", lds);
    const int nty = (p.Hout + 7) / 8;
    dim3 grid(p.ntx * nty, (p.Co + MT * 32 - 1) / (MT * 32), p.B);
    hipLaunchKernelGGL(k, grid, dim3(RVSR_WG), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_fwd2 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// p.w = f32 weights; packs them into `workspace`, then runs the bf16x3 kernel
int rvsr_launch_conv_fwd2(ConvFwdParams p, int ksize, int stride, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int Ctot = p.in.a.C + p.in.b.C;
    int mt, ccg, nchunks, nmb;
    fwd2_geom(ksize, p.Co, Ctot, mt, ccg, nchunks, nmb);
    const size_t need = rvsr_conv_fwd2_workspace_bytes(ksize, p.Co, Ctot);
    if (!workspace || workspace_bytes < need) FAIL(RVSR_ERR_WORKSPACE, "conv2d: workspace %zu B < %zu B", workspace_bytes, need);
    const int T = ksize * ksize;
    const size_t total = (size_t)nmb * nchunks * T * (2 * ccg) * (mt * 32);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.w, (bf16x8*)workspace, p.Co,
                       Ctot, T, mt * 32, ccg, nchunks, nmb, p.w_mode);
    p.wpack = workspace;
#define DISPATCH2(KS, S, CCG)                                   \
    do {                                                        \
        if (mt == 1) return launch_fwd2<KS, S, 1, CCG>(p, st);  \
        if (mt == 2) return launch_fwd2<KS, S, 2, CCG>(p, st);  \
        return launch_fwd2<KS, S, 4, CCG>(p, st);               \
    } while (0)
    if (ksize == 3 && stride == 1) DISPATCH2(3, 1, 1);
    if (ksize == 3 && stride == 2) DISPATCH2(3, 2, 1);
    DISPATCH2(1, 1, 2);
#undef DISPATCH2
}
