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
#define RVSR_DEFINE_PACK
#include "conv_common.h"

#include "bf16x3.h"

// ------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int MT, int CCG, bool ACT_IN, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv_fwd2_kernel(const ConvFwdParams p) {
    constexpr int T = KS * KS, PAD = KS / 2, TH = 2 * NW, TW = 32, NTHR = NW * 64;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int MP = MT * 32, NOCT = 2 * CCG, NPOS = IH * IW;
    constexpr int WVEC = T * NOCT * MP;  // 16-byte vectors per weight part (hi or lo)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* xs_hi = reinterpret_cast<bf16x8*>(smem_raw);  // [NOCT][IH][IW]
    bf16x8* xs_lo = xs_hi + NOCT * NPOS;
    bf16x8* ws_hi = xs_lo + NOCT * NPOS;                  // [T][NOCT][MP]
    bf16x8* ws_lo = ws_hi + WVEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    unsigned sbx, sby, sbz;
    swizzled_block(sbx, sby, sbz, p.swz);
    const int tx = sbx % p.ntx, ty = sbx / p.ntx;
    const int x0 = tx * TW, y0 = ty * TH, mb = sby, b = sbz;
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

    // Software pipeline: the global loads of chunk k+1 (input octets + packed weights) are issued right
    // after the barrier that opens chunk k's MFMA phase and are only consumed (converted / written to
    // LDS) after it, so HBM/L2 latency hides under the matrix-core work.
    constexpr int NIT = (NOCT * NPOS + NTHR - 1) / NTHR;   // input items per thread
    float vin[NIT][8];
    float ain[ACT_IN ? NIT : 1][8];  // saved activation outputs (sign -> derivative), only for data gradients

    auto issue_loads = [&](int chunk) {
        const int c0 = chunk * 16 * CCG;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = tid + i * NTHR;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                vin[i][j] = 0.f;
                if (ACT_IN) ain[i][j] = 1.f;
            }
            if (it >= NOCT * NPOS) continue;
            const int oc = it / NPOS, pos = it - oc * NPOS;
            const int r = pos / IW, s = pos - r * IW;
            const int gy = y0 * STRIDE - PAD + r, gx = x0 * STRIDE - PAD + s;
            const int cb = c0 + oc * 8;
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
                        if (j < nvalid) vin[i][j] = v0.p[base + j * hw];
                    if (ACT_IN) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j < nvalid) ain[i][j] = v0.act[base + j * hw];
                    }
                } else {  // mode 2: pixel-unshuffle view, virtual channel c -> stored (c>>2, 2y+((c>>1)&1), 2x+(c&1))
                    const size_t hw = (size_t)va.Hs * va.Ws;
                    const size_t base = ((size_t)b * (va.C >> 2) + (cb >> 2)) * hw + (size_t)(2 * gy) * va.Ws + 2 * gx;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const size_t idx = base + (j >> 2) * hw + ((j >> 1) & 1) * va.Ws + (j & 1);
                        vin[i][j] = va.p[idx];
                        if (ACT_IN) ain[i][j] = va.act[idx];
                    }
                }
            }
        }
    };
    auto commit_to_lds = [&](int chunk) {
        // packed weights: straight 16-byte copy (L2-resident); issued first so its latency overlaps the conversion below
        const bf16x8* src = wsrc + (size_t)chunk * 2 * WVEC;
#if !defined(RVSR_EXP) || RVSR_EXP != 1
#pragma unroll 3
        for (int e = tid; e < 2 * WVEC; e += NTHR) ws_hi[e] = src[e];
#else
        if (chunk == 0) for (int e = tid; e < 2 * WVEC; e += NTHR) ws_hi[e] = src[e];
#endif
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = tid + i * NTHR;
            if (it >= NOCT * NPOS) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ACT_IN ? vin[i][j] * (ain[i][j] > 0.f ? 1.f : va.slope) : vin[i][j];
            bf16x8 h8, l8;
#if defined(RVSR_EXP) && RVSR_EXP == 5
            if (v[0] == 12345.f)
#endif
            {
            split8(v, h8, l8);
            xs_hi[it] = h8;
            xs_lo[it] = l8;
            }
        }
    };

    issue_loads(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        commit_to_lds(chunk);
        __syncthreads();
#if defined(RVSR_EXP) && RVSR_EXP == 2
        if (false)
#endif
        if (chunk + 1 < nchunks) issue_loads(chunk + 1);
#if defined(RVSR_EXP) && RVSR_EXP == 3
        if (p.B < 0)
#endif
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
                // the three split terms as three sweeps over the independent accumulators: consecutive
                // MFMAs never depend on each other (a dependent 32x32x16 pair costs an extra pass group)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(ah[m], bh[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(ah[m], bl[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(al[m], bh[n], acc[m][n]);
            }
        }
        __syncthreads();
    }

#if defined(RVSR_EXP) && RVSR_EXP == 4
    if (acc[0][0][0] != 12345.f) return;
#endif
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
    // 8 waves (16 rows x 32 px) per workgroup for the common stride-1 3x3 case: the weight slice is
    // amortised over twice the pixels and 2 workgroups/CU = 16 waves hide the staging latency better
    constexpr int NW = 4;  // (8 waves / 16x32 px measured 5% slower: 1.07 vs 1.00 ms on the 40x64x180x320 conv)
    constexpr int T = KS * KS, IH = (2 * NW - 1) * STRIDE + KS, IW = 31 * STRIDE + KS;
    const size_t lds = (size_t)16 * (2 * (2 * CCG) * IH * IW + 2 * T * (2 * CCG) * (MT * 32));
    auto k = p.in.a.act != nullptr ? conv_fwd2_kernel<KS, STRIDE, MT, CCG, true, NW> : conv_fwd2_kernel<KS, STRIDE, MT, CCG, false, NW>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_fwd2: cannot reserve %zu B of LDS", lds);
    const int nty = (p.Hout + 2 * NW - 1) / (2 * NW);
    dim3 grid(p.ntx * nty, (p.Co + MT * 32 - 1) / (MT * 32), p.B);
    hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, st, p);
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
    p.swz = rvsr_swizzle_enabled();
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

// ==========================================================================================
// Weight gradient on the bf16 matrix cores (3x3, stride 1, Wout % 4 == 0; other cases use the
// exact-f32 kernel of conv_kernels.hip).
//
//   gW[o][(tap, c)] = sum_px G[o][px] * X[c][px + tap],   G = grad_out (* act'), K = pixels.
//
// One workgroup = 8 waves, persistent over 4x32-pixel tiles for a fixed (64-row m-block, 64-channel
// chunk): 2 M tiles x 18 N tiles (tap x channel-half) = 36 accumulator tiles, 5/4 per wave so that
// the two waves sharing a SIMD (w, w+4) hold 9 between them.  Both operands want 8 consecutive
// PIXELS per lane, which is the natural NCHW order: G and X tiles are staged with aligned 16-byte
// global loads, split into bf16 hi/lo and kept pixel-contiguous in LDS; the +-1 column shift of a
// tap is applied in registers (v_alignbit on the two loaded octets), the row shift is an address.
// The X tile is stored from column x0-4 so that every 8-pixel group is 16-byte aligned in LDS.
#define WG2_THREADS 512
#define WG2_GP 272   // bytes per output-channel row of the G tile: 4 rows x 64 B + 16 B pad
#define WG2_XP 496   // bytes per input-channel plane of the X tile: 6 rows x 80 B + 16 B pad

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// 8 bf16 starting `shift` (3, 4 or 5) elements into the 16 held by (a, b)
template <int SHIFT>
__device__ __forceinline__ bf16x8 take8(u32x4 a, u32x4 b) {
    u32x4 r;
    if (SHIFT == 4) {
        r[0] = a[2]; r[1] = a[3]; r[2] = b[0]; r[3] = b[1];
    } else if (SHIFT == 3) {
        r[0] = __builtin_amdgcn_alignbit(a[2], a[1], 16);
        r[1] = __builtin_amdgcn_alignbit(a[3], a[2], 16);
        r[2] = __builtin_amdgcn_alignbit(b[0], a[3], 16);
        r[3] = __builtin_amdgcn_alignbit(b[1], b[0], 16);
    } else {
        r[0] = __builtin_amdgcn_alignbit(a[3], a[2], 16);
        r[1] = __builtin_amdgcn_alignbit(b[0], a[3], 16);
        r[2] = __builtin_amdgcn_alignbit(b[1], b[0], 16);
        r[3] = __builtin_amdgcn_alignbit(b[2], b[1], 16);
    }
    return as_bf16x8(r);
}

// one (dy) row of taps for one accumulator group: DXMASK selects which dx (bit 0..2) this wave owns.
// The split terms are issued as sweeps over the row's independent accumulators (no back-to-back
// dependent MFMAs).
template <int DXMASK>
__device__ __forceinline__ void wg2_row(const unsigned char* xs_hi, const unsigned char* xs_lo, int xoff, bf16x8 ah,
                                        bf16x8 al, f32x16* acc) {
    const u32x4 h0 = *reinterpret_cast<const u32x4*>(xs_hi + xoff), h1 = *reinterpret_cast<const u32x4*>(xs_hi + xoff + 16);
    const u32x4 l0 = *reinterpret_cast<const u32x4*>(xs_lo + xoff), l1 = *reinterpret_cast<const u32x4*>(xs_lo + xoff + 16);
    constexpr int N = ((DXMASK >> 0) & 1) + ((DXMASK >> 1) & 1) + ((DXMASK >> 2) & 1);
    bf16x8 bh[3], bl[3];
    int t = 0;
    if (DXMASK & 1) { bh[t] = take8<3>(h0, h1); bl[t] = take8<3>(l0, l1); ++t; }
    if (DXMASK & 2) { bh[t] = take8<4>(h0, h1); bl[t] = take8<4>(l0, l1); ++t; }
    if (DXMASK & 4) { bh[t] = take8<5>(h0, h1); bl[t] = take8<5>(l0, l1); ++t; }
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = mfma_bf16(ah, bh[i], acc[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = mfma_bf16(ah, bl[i], acc[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = mfma_bf16(al, bh[i], acc[i]);
}

__global__ __launch_bounds__(WG2_THREADS, 2) void conv_wgrad2_kernel(const ConvWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* gs_hi = smem_raw;                  // [64 o][WG2_GP]
    unsigned char* gs_lo = gs_hi + 64 * WG2_GP;
    unsigned char* xs_hi = gs_lo + 64 * WG2_GP;       // [64 c][WG2_XP]
    unsigned char* xs_lo = xs_hi + 64 * WG2_XP;
    float* bsum = reinterpret_cast<float*>(xs_lo + 64 * WG2_XP);  // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int mb = blockIdx.y, c0 = blockIdx.z * 64;
    const int Ctot = p.x.a.C + p.x.b.C, C1 = p.x.a.C;
    const int H = p.x.a.Hs, W = p.x.a.Ws;  // stride 1, pad 1: Hout == H, Wout == W
    // wave -> (M tile, channel half, first/second half of the 9 taps)
    const int m = wave & 1, chalf = (wave >> 1) & 1, second = wave >> 2;
    const bool m_live = (mb * 64 + m * 32) < p.Co;
    const bool do_bias = p.bpart != nullptr && blockIdx.z == 0;

    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = zero16();
    if (tid < 64) bsum[tid] = 0.f;
    __syncthreads();

    const int ntiles = p.B * p.nty * p.ntx;
    for (int tile = blockIdx.x; tile < ntiles; tile += p.P) {
        const int b = tile / (p.nty * p.ntx);
        const int trem = tile - b * (p.nty * p.ntx);
        const int ty = trem / p.ntx, tx = trem - ty * p.ntx;
        const int y0 = ty * 4, x0 = tx * 32;
        // ---- G tile: 64 o x 4 rows x 4 octets
        for (int it = tid; it < 64 * 16; it += WG2_THREADS) {
            const int q = it & 3, row = (it >> 2) & 3, ol = it >> 4;
            const int o = mb * 64 + ol, gy = y0 + row, gx = x0 + 8 * q;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (o < p.Co && gy < H && gx < W) {
                if (p.g.mode == 0) {
                    const size_t idx = (((size_t)b * p.Co + o) * H + gy) * W + gx;
                    const float4 a0 = *reinterpret_cast<const float4*>(p.g.p + idx);
                    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w;
                    if (gx + 4 < W) {
                        const float4 a1 = *reinterpret_cast<const float4*>(p.g.p + idx + 4);
                        v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
                    }
                    if (p.g.act != nullptr) {
                        const float4 s0 = *reinterpret_cast<const float4*>(p.g.act + idx);
                        v[0] *= s0.x > 0.f ? 1.f : p.g.slope; v[1] *= s0.y > 0.f ? 1.f : p.g.slope;
                        v[2] *= s0.z > 0.f ? 1.f : p.g.slope; v[3] *= s0.w > 0.f ? 1.f : p.g.slope;
                        if (gx + 4 < W) {
                            const float4 s1 = *reinterpret_cast<const float4*>(p.g.act + idx + 4);
                            v[4] *= s1.x > 0.f ? 1.f : p.g.slope; v[5] *= s1.y > 0.f ? 1.f : p.g.slope;
                            v[6] *= s1.z > 0.f ? 1.f : p.g.slope; v[7] *= s1.w > 0.f ? 1.f : p.g.slope;
                        }
                    }
                } else {  // mode 2: stored (Co/4, 2H, 2W) pixel-shuffled; 8 virtual px = 16 stored floats, every other one
                    const size_t idx = (((size_t)b * (p.Co >> 2) + (o >> 2)) * (2 * H) + 2 * gy + ((o >> 1) & 1)) * (size_t)(2 * W) + 2 * gx;
                    const int sx = o & 1;
#pragma unroll
                    for (int h4 = 0; h4 < 4; ++h4) {
                        if (gx + 2 * h4 < W) {
                            const float4 a = *reinterpret_cast<const float4*>(p.g.p + idx + 4 * h4);
                            float e0 = sx ? a.y : a.x, e1 = sx ? a.w : a.z;
                            if (p.g.act != nullptr) {
                                const float4 s = *reinterpret_cast<const float4*>(p.g.act + idx + 4 * h4);
                                e0 *= (sx ? s.y : s.x) > 0.f ? 1.f : p.g.slope;
                                e1 *= (sx ? s.w : s.z) > 0.f ? 1.f : p.g.slope;
                            }
                            v[2 * h4] = e0;
                            v[2 * h4 + 1] = e1;
                        }
                    }
                }
            }
            if (do_bias) {
                const float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                atomicAdd(&bsum[ol], s);
            }
            bf16x8 h8, l8;
            split8(v, h8, l8);
            const int off = ol * WG2_GP + row * 64 + q * 16;
            *reinterpret_cast<bf16x8*>(gs_hi + off) = h8;
            *reinterpret_cast<bf16x8*>(gs_lo + off) = l8;
        }
        // ---- X tile: 64 c x 6 rows x 5 octets, columns x0-4 .. x0+35
        for (int it = tid; it < 64 * 30; it += WG2_THREADS) {
            const int cl = it / 30, rem = it - cl * 30;
            const int row = rem / 5, q = rem - row * 5;
            const int c = c0 + cl, gy = y0 - 1 + row, gx = x0 - 4 + 8 * q;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (c < Ctot && gy >= 0 && gy < H) {
                const float* src = c < C1 ? p.x.a.p + ((size_t)b * C1 + c) * H * W : p.x.b.p + ((size_t)b * (Ctot - C1) + (c - C1)) * H * W;
                src += (size_t)gy * W;
                if (gx >= 0 && gx < W) {
                    const float4 a = *reinterpret_cast<const float4*>(src + gx);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                }
                if (gx + 4 >= 0 && gx + 4 < W) {
                    const float4 a = *reinterpret_cast<const float4*>(src + gx + 4);
                    v[4] = a.x; v[5] = a.y; v[6] = a.z; v[7] = a.w;
                }
            }
            bf16x8 h8, l8;
            split8(v, h8, l8);
            const int off = cl * WG2_XP + row * 80 + q * 16;
            *reinterpret_cast<bf16x8*>(xs_hi + off) = h8;
            *reinterpret_cast<bf16x8*>(xs_lo + off) = l8;
        }
        __syncthreads();
        if (m_live) {
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                const int row = ks >> 1, cb = (ks & 1) * 16 + 8 * hi;  // this lane's 8 pixels: row, cols cb..cb+7
                const int goff = (m * 32 + lo) * WG2_GP + row * 64 + cb * 2;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(gs_hi + goff);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(gs_lo + goff);
                // X octet pair holding stored cols cb .. cb+15 (= image cols x0-4+cb ..); tap dx needs +3+dx
                const int xbase = (chalf * 32 + lo) * WG2_XP + cb * 2;
                if (!second) {  // taps (0,0) (0,1) (0,2) (1,0) (1,1)
                    wg2_row<7>(xs_hi, xs_lo, xbase + (row + 0) * 80, ah, al, acc + 0);
                    wg2_row<3>(xs_hi, xs_lo, xbase + (row + 1) * 80, ah, al, acc + 3);
                } else {        // taps (1,2) (2,0) (2,1) (2,2)
                    wg2_row<4>(xs_hi, xs_lo, xbase + (row + 1) * 80, ah, al, acc + 0);
                    wg2_row<7>(xs_hi, xs_lo, xbase + (row + 2) * 80, ah, al, acc + 1);
                }
            }
        }
        __syncthreads();
    }

    if (m_live) {
        const int c = c0 + chalf * 32 + lo;
        const int ntap = second ? 4 : 5, tap0 = second ? 5 : 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i >= ntap || c >= Ctot) continue;
            const int tap = tap0 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = mb * 64 + m * 32 + drow(r, hi);
                if (o < p.Co) p.part[(((size_t)blockIdx.x * p.Co + o) * Ctot + c) * 9 + tap] = acc[i][r];
            }
        }
    }
    if (do_bias && tid < 64) {
        const int o = mb * 64 + tid;
        if (o < p.Co) p.bpart[(size_t)blockIdx.x * p.Co + o] = bsum[tid];
    }
}

int rvsr_launch_conv_wgrad2(const ConvWgradParams& p, int gy, int gz, hipStream_t st) {
    const size_t lds = 2 * 64 * WG2_GP + 2 * 64 * WG2_XP + 64 * sizeof(float);
    if (set_lds(conv_wgrad2_kernel, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad2: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(conv_wgrad2_kernel, dim3(p.P, gy, gz), dim3(WG2_THREADS), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad2 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}
