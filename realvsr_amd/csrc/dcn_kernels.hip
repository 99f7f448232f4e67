// dcn_kernels.hip -- modulated deformable convolution (DCNv2) for gfx950, fused.
//
// Replaces the reference's CUDA operator (codes/models/archs/dcn/src/deform_conv_cuda_kernel.cu
// :571-767 + deform_conv_cuda.cpp:490-685): there, every sample of the batch runs
// im2col -> `columns[C*9, H*W]` in HBM -> cuBLAS GEMM (forward) and GEMM -> columns -> col2im /
// col2im_coord + im2col again -> GEMM (backward).  Here the column tile never leaves the CU:
//
//   forward          per 4x32-pixel tile, per chunk of 8 input channels: the 72 x 128 column
//                    tile (bilinear gather * mask) is built in LDS, then multiplied by the
//                    weight slice on the f32 matrix cores (v_mfma_f32_32x32x2_f32), accumulating
//                    out[Co, 128 px] in registers; bias (+ optional LeakyReLU) in the epilogue.
//   backward (input) col_grad tile = W_chunk^T (72 x Co) * gOut (Co x 128 px) on the matrix
//                    cores -> LDS; consumed in place by the grad_offset / grad_mask reductions
//                    and the grad_input scatter (kernel.cu:636-767 fused into one pass).
//   backward (weight) persistent workgroups rebuild column tiles and accumulate
//                    gW[Co, 72(+1 ones row = gBias)] over many pixel tiles in registers; partials
//                    are reduced by a second kernel in fixed order (deterministic).
//
// Offsets/mask may be handed over either as the reference's dense tensors (offset (B,2*dg*9,H,W),
// mask (B,dg*9,H,W)) or as the raw (B,3*dg*9,H,W) output of conv_offset_mask: batch strides are
// explicit and the mask sigmoid (deform_conv.py:281-283) is then applied in-kernel.
#include "rvsr_common.h"

#include "dcn_common.h"

// Build the column tile of channel chunk [c0, c0+8) for the 128 pixels of a tile.
// COLT == false: col[k * 128 + px]      (k-major; B operand of the forward GEMM)
// COLT == true : col[px * ldc + k]       (pixel-major; B operand of the weight-gradient GEMM)
template <bool COLT, int CHS>
__device__ __forceinline__ void dcn_build_cols(const DcnGeom& d, int b, int c0, int y0, int x0, float* col, int ldc,
                                               int tid) {
    // channels sharing one offset set inside the chunk (CHS > 0: compile-time -> unrolled gathers)
    const int chs = CHS > 0 ? CHS : (d.cpg < DCN_CC ? d.cpg : DCN_CC);
    const int nslots = DCN_CC / chs;
    const int nitems = DCN_NPX * 9 * nslots;
    const size_t HW = (size_t)d.H * d.W;
    for (int it = tid; it < nitems; it += RVSR_WG) {
        const int px = it & (DCN_NPX - 1);
        const int rest = it >> 7;
        const int k = rest % 9, slot = rest / 9;
        const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
        const int cb = c0 + slot * chs;
        const bool live = oy < d.Ho && ox < d.Wo && cb < d.C;
        Samp s;
        if (live) s = dcn_sample(d, b, cb / d.cpg, k, oy, ox);
        const bool go = live && s.inside;
#pragma unroll
        for (int j = 0; j < (CHS > 0 ? CHS : chs); ++j) {
            const int c = cb + j;
            float v = 0.f;
            if (go && c < d.C) {
                const float* pl = d.x + ((size_t)b * d.C + c) * HW;
                v = (s.w00 * pl[s.i00] + s.w01 * pl[s.i01] + s.w10 * pl[s.i10] + s.w11 * pl[s.i11]) * s.m;
            }
            const int kr = (slot * chs + j) * 9 + k;
            if (COLT)
                col[px * ldc + kr] = v;
            else
                col[kr * DCN_NPX + px] = v;
        }
    }
}

// add `v` to plane[idx] (idx = y*W + x): through the LDS tile when (y, x) falls inside it
__device__ __forceinline__ void dcn_scatter(float* gplane, float* ltile, int idx, float v, int W, int ty0, int tx0,
                                            int gth, int gtw) {
    if (v == 0.f) return;
    const int y = idx / W, x = idx - y * W;
    const int ly = y - ty0, lx = x - tx0;
    if (ly >= 0 && ly < gth && lx >= 0 && lx < gtw)
        __hip_atomic_fetch_add(ltile + ly * gtw + lx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
        atomicAdd(gplane + idx, v);
}

// ------------------------------------------------------------------------------------------
template <int MT, int CHS>
__global__ __launch_bounds__(RVSR_WG, 2) void dcn_fwd_kernel(const DcnFwdParams p) {
    constexpr int MP = MT * 32, MPP = MP + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* col = smem;                     // [72][128]
    float* ws = smem + DCN_KC * DCN_NPX;   // [72][MPP]
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int tx = blockIdx.x % d.ntx, ty = blockIdx.x / d.ntx;
    const int x0 = tx * 32, y0 = ty * 4, mb = blockIdx.y, b = blockIdx.z;
    const int K = d.C * 9;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = zero16();

    for (int c0 = 0; c0 < d.C; c0 += DCN_CC) {
        // weight slice, transposed: ws[kk][o] = w[o][c0*9 + kk]
#pragma unroll 4
        for (int e = tid; e < MP * DCN_KC; e += RVSR_WG) {
            const int m = e / DCN_KC, kk = e - m * DCN_KC;
            const int o = mb * MP + m, kg = c0 * 9 + kk;
            float v = 0.f;
            if (o < d.Co && kg < K) v = p.w[(size_t)o * K + kg];
            ws[kk * MPP + m] = v;
        }
        dcn_build_cols<false, CHS>(d, b, c0, y0, x0, col, 0, tid);
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < DCN_KC / 2; ++ks) {
            const int kk = 2 * ks + hi;
            const float bv = col[kk * DCN_NPX + wave * 32 + lo];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = mfma32(ws[kk * MPP + m * 32 + lo], bv, acc[m]);
        }
        __syncthreads();
    }

    const int row = y0 + wave, colx = x0 + lo;
    if (row >= d.Ho) return;
    const bool has_bias = p.bias != nullptr;
    const float* bp = has_bias ? p.bias : p.w;
    const float neg = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
    const size_t hw = (size_t)d.Ho * d.Wo;
    const size_t pix = (size_t)row * d.Wo + colx;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = mb * MP + m * 32 + drow(r, hi);
            const bool ok = colx < d.Wo && o < d.Co;
            const int oc = ok ? o : 0;
            float v = acc[m][r];
            const float bb = bp[oc];
            v += has_bias ? bb : 0.f;
            v = v > 0.f ? v : v * neg;
            if (ok) p.out[((size_t)b * d.Co + oc) * hw + pix] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
struct DcnBwdInParams {
    DcnGeom d;
    const float* w;     // (Co, C, 3, 3)
    TView g;            // grad_output view (Co, Ho, Wo), optional fused act'
    float* gx;          // (B, C, H, W), must be zero on entry (atomics)
    float* goff;        // (b * goff_bs)[(g*18 + 2k + dir)][Ho][Wo]   overwritten
    float* gmask;       // (b * gmask_bs)[(g*9 + k)][Ho][Wo]          overwritten (d/d logit if mask_logit)
    size_t goff_bs, gmask_bs;
};

template <int CHS>
__global__ __launch_bounds__(RVSR_WG, 1) void dcn_bwd_input_kernel(const DcnBwdInParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const DcnGeom& d = p.d;
    const int CoP = (d.Co + 1) & ~1;            // K of the col_grad GEMM, padded to even
    float* gs = smem;                            // [CoP][128]     grad_output tile
    float* ws = gs + CoP * DCN_NPX;              // [CoP][72]      weight slice (natural layout)
    float* cg = ws + CoP * DCN_KC;               // [72][128]      col_grad tile
    // grad_input accumulation tile for the chunk's 8 channels: rows y0-1-R .. y0+4+R, cols x0-1-R .. x0+32+R
    // (stride 1 / dilation 1 geometry; other geometries simply hit the global-atomic fallback more often)
    constexpr int GR = 4, GTH = 4 + 2 * GR + 2, GTW = 32 + 2 * GR + 2;
    float* gxt = cg + DCN_KC * DCN_NPX;          // [8][GTH][GTW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int tx = blockIdx.x % d.ntx, ty = blockIdx.x / d.ntx;
    const int x0 = tx * 32, y0 = ty * 4, b = blockIdx.z;
    const int K = d.C * 9;
    const size_t hw = (size_t)d.Ho * d.Wo, HW = (size_t)d.H * d.W;

#pragma unroll 2
    for (int e = tid; e < CoP * DCN_NPX; e += RVSR_WG) {
        const int o = e >> 7, px = e & 127;
        gs[e] = o < d.Co ? tview_get(p.g, b, o, y0 + (px >> 5), x0 + (px & 31)) : 0.f;
    }

    const int chs = CHS > 0 ? CHS : (d.cpg < DCN_CC ? d.cpg : DCN_CC);
    const int nslots = DCN_CC / chs;
    const int nitems = DCN_NPX * 9 * nslots;
    const int ty0 = y0 * d.stride - d.pad - GR, tx0 = x0 * d.stride - d.pad - GR;  // image coords of tile cell (0,0)
    for (int e = tid; e < DCN_CC * GTH * GTW; e += RVSR_WG) gxt[e] = 0.f;

    for (int c0 = 0; c0 < d.C; c0 += DCN_CC) {
#pragma unroll 4
        for (int e = tid; e < CoP * DCN_KC; e += RVSR_WG) {
            const int o = e / DCN_KC, kk = e - o * DCN_KC;
            const int kg = c0 * 9 + kk;
            ws[e] = (o < d.Co && kg < K) ? p.w[(size_t)o * K + kg] : 0.f;
        }
        __syncthreads();
        // col_grad[72 (3 M tiles, rows >= 72 idle)][32 px of this wave] = W^T * gOut
        f32x16 acc[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = zero16();
        for (int ks = 0; ks < CoP / 2; ++ks) {
            const int o = 2 * ks + hi;
            const float bv = gs[o * DCN_NPX + wave * 32 + lo];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int kr = m * 32 + lo;
                const float a = kr < DCN_KC ? ws[o * DCN_KC + kr] : 0.f;
                acc[m] = mfma32(a, bv, acc[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = m * 32 + drow(r, hi);
                if (kr < DCN_KC) cg[kr * DCN_NPX + wave * 32 + lo] = acc[m][r];
            }
        __syncthreads();
        // consume the col_grad tile: grad_offset / grad_mask reductions + grad_input scatter
        for (int it = tid; it < nitems; it += RVSR_WG) {
            const int px = it & (DCN_NPX - 1);
            const int rest = it >> 7;
            const int k = rest % 9, slot = rest / 9;
            const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
            const int cb = c0 + slot * chs;
            if (!(oy < d.Ho && ox < d.Wo && cb < d.C)) continue;
            const int g = cb / d.cpg;
            const Samp s = dcn_sample(d, b, g, k, oy, ox);
            float gy = 0.f, gxo = 0.f, gm = 0.f;
            if (s.inside) {
                const float hy = 1.f - s.ly, hx = 1.f - s.lx;
#pragma unroll
                for (int j = 0; j < (CHS > 0 ? CHS : chs); ++j) {
                    const int c = cb + j;
                    if (c >= d.C) continue;
                    const float cgv = cg[((slot * chs + j) * 9 + k) * DCN_NPX + px];
                    const float* pl = d.x + ((size_t)b * d.C + c) * HW;
                    const float l00 = pl[s.i00], l01 = pl[s.i01], l10 = pl[s.i10], l11 = pl[s.i11];
                    const float x00 = s.v00 ? l00 : 0.f, x01 = s.v01 ? l01 : 0.f;
                    const float x10 = s.v10 ? l10 : 0.f, x11 = s.v11 ? l11 : 0.f;
                    // grad_mask (kernel.cu:751-754): col_grad * bilinear(x)
                    gm += cgv * (s.w00 * x00 + s.w01 * x01 + s.w10 * x10 + s.w11 * x11);
                    // grad_offset (kernel.cu:544-565): d(bilinear)/dy, d(bilinear)/dx
                    const float t = cgv * s.m;
                    gy += (hx * (x10 - x00) + s.lx * (x11 - x01)) * t;
                    gxo += (hy * (x01 - x00) + s.ly * (x11 - x10)) * t;
                    // grad_input (kernel.cu:674-691): scatter to the <=4 valid corners
                    // LDS-privatised: corners inside the tile go to ds_add_f32, the rest to global atomics
                    float* gp = p.gx + ((size_t)b * d.C + c) * HW;
                    float* lt = gxt + (slot * chs + j) * (GTH * GTW);
                    dcn_scatter(gp, lt, s.i00, s.w00 * t, d.W, ty0, tx0, GTH, GTW);
                    dcn_scatter(gp, lt, s.i01, s.w01 * t, d.W, ty0, tx0, GTH, GTW);
                    dcn_scatter(gp, lt, s.i10, s.w10 * t, d.W, ty0, tx0, GTH, GTW);
                    dcn_scatter(gp, lt, s.i11, s.w11 * t, d.W, ty0, tx0, GTH, GTW);
                }
            }
            if (d.mask_logit) gm *= s.m * (1.f - s.m);
            const size_t pp = (size_t)oy * d.Wo + ox;
            float* go = p.goff + (size_t)b * p.goff_bs + (size_t)(g * 18 + 2 * k) * hw + pp;
            float* gk = p.gmask + (size_t)b * p.gmask_bs + (size_t)(g * 9 + k) * hw + pp;
            if (cb % d.cpg == 0) {  // first chunk of this deformable group: overwrite
                go[0] = gy;
                go[hw] = gxo;
                gk[0] = gm;
            } else {                // group spans several chunks (cpg > 8): accumulate
                go[0] += gy;
                go[hw] += gxo;
                gk[0] += gm;
            }
        }
        __syncthreads();
        // flush the accumulation tile (one global atomic per touched cell instead of one per sample corner)
        for (int e = tid; e < DCN_CC * GTH * GTW; e += RVSR_WG) {
            const float v = gxt[e];
            if (v != 0.f) {
                gxt[e] = 0.f;
                const int cc = e / (GTH * GTW), rem = e - cc * (GTH * GTW);
                const int yy = ty0 + rem / GTW, xx = tx0 + rem % GTW;
                const int c = c0 + cc;
                if (c < d.C && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W)
                    atomicAdd(p.gx + ((size_t)b * d.C + c) * HW + (size_t)yy * d.W + xx, v);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
struct DcnBwdWParams {
    DcnGeom d;
    TView g;
    float* part;   // [4P][Co][C][9]
    float* bpart;  // [4P][Co] or nullptr
    int P, nty;
};

template <int CHS>
__global__ __launch_bounds__(RVSR_WG, 1) void dcn_bwd_weight_kernel(const DcnBwdWParams p) {
    constexpr int GP = 65, CP = 97;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gT = smem;                   // [128][65]  grad_output tile, pixel-major
    float* colT = smem + DCN_NPX * GP;  // [128][97]  column tile, pixel-major; col 72 = 1 (bias)
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int mb = blockIdx.y, c0 = blockIdx.z * DCN_CC;
    const bool m1_live = mb * 64 + 32 < d.Co;

    for (int e = tid; e < DCN_NPX * (CP - DCN_KC); e += RVSR_WG) {
        const int px = e / (CP - DCN_KC), j = e - px * (CP - DCN_KC);
        colT[px * CP + DCN_KC + j] = j == 0 ? 1.f : 0.f;
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = zero16();

    const int ntiles = d.B * p.nty * d.ntx;
    for (int tile = blockIdx.x; tile < ntiles; tile += p.P) {
        const int b = tile / (p.nty * d.ntx);
        const int trem = tile - b * (p.nty * d.ntx);
        const int ty = trem / d.ntx, tx = trem - ty * d.ntx;
        const int y0 = ty * 4, x0 = tx * 32;
#pragma unroll 2
        for (int e = tid; e < 64 * DCN_NPX; e += RVSR_WG) {
            const int ol = e >> 7, px = e & 127;
            const int o = mb * 64 + ol;
            gT[px * GP + ol] = o < d.Co ? tview_get(p.g, b, o, y0 + (px >> 5), x0 + (px & 31)) : 0.f;
        }
        dcn_build_cols<true, CHS>(d, b, c0, y0, x0, colT, CP, tid);
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
            const int px = wave * 32 + 2 * ks + hi;
            const float a0 = gT[px * GP + lo], a1 = gT[px * GP + 32 + lo];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const float bv = colT[px * CP + n * 32 + lo];
                acc[0][n] = mfma32(a0, bv, acc[0][n]);
                if (m1_live) acc[1][n] = mfma32(a1, bv, acc[1][n]);
            }
        }
        __syncthreads();
    }

    const int q = blockIdx.x * 4 + wave;
    const int K = d.C * 9;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int kr = n * 32 + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = mb * 64 + m * 32 + drow(r, hi);
                if (o >= d.Co) continue;
                const int kg = c0 * 9 + kr;
                if (kr < DCN_KC && kg < K) p.part[((size_t)q * d.Co + o) * K + kg] = acc[m][n][r];
                if (kr == DCN_KC && p.bpart != nullptr && blockIdx.z == 0) p.bpart[(size_t)q * d.Co + o] = acc[m][n][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side


static int fill_geom(DcnGeom& d, const float* x, const float* offset, size_t off_bs, const float* mask, size_t mask_bs,
                     int mask_logit, int B, int C, int H, int W, int Co, int kh, int kw, int stride_h, int stride_w,
                     int pad_h, int pad_w, int dil_h, int dil_w, int group, int dg, const char** why) {
    if (!x || !offset || !mask || B <= 0 || C <= 0 || Co <= 0 || H <= 0 || W <= 0) { *why = "null/empty argument"; return RVSR_ERR_BAD_ARG; }
    if (kh != 3 || kw != 3) { *why = "only 3x3 kernels are implemented on the HIP path"; return RVSR_ERR_UNSUPPORTED; }
    if (group != 1) { *why = "only group == 1 is implemented on the HIP path"; return RVSR_ERR_UNSUPPORTED; }
    if (stride_h != stride_w || pad_h != pad_w || dil_h != dil_w) { *why = "anisotropic stride/pad/dilation"; return RVSR_ERR_UNSUPPORTED; }
    if (dg <= 0 || C % dg) { *why = "channels not divisible by deformable_group"; return RVSR_ERR_BAD_ARG; }
    const int cpg = C / dg;
    if (!(cpg % DCN_CC == 0 || DCN_CC % cpg == 0)) { *why = "channels per deformable group must divide or be a multiple of 8"; return RVSR_ERR_UNSUPPORTED; }
    d.x = x; d.offset = offset; d.mask = mask; d.off_bs = off_bs; d.mask_bs = mask_bs; d.mask_logit = mask_logit;
    d.B = B; d.C = C; d.H = H; d.W = W; d.Co = Co;
    d.stride = stride_h; d.pad = pad_h; d.dil = dil_h; d.dg = dg; d.cpg = cpg;
    d.Ho = (H + 2 * pad_h - (dil_h * 2 + 1)) / stride_h + 1;
    d.Wo = (W + 2 * pad_w - (dil_w * 2 + 1)) / stride_w + 1;
    if (d.Ho <= 0 || d.Wo <= 0) { *why = "empty output"; return RVSR_ERR_BAD_ARG; }
    d.ntx = (d.Wo + 31) / 32;
    d.swz = rvsr_swizzle_enabled();
    return RVSR_OK;
}

static int dcn_forward_impl(DcnGeom& d, const float* weight, const float* bias, float* out, int act, float slope,
                            void* workspace, size_t workspace_bytes, hipStream_t st, int prepacked = 0, unsigned* probe = nullptr, int halo_hint = 0) {
    DcnFwdParams p;
    p.d = d; p.w = weight; p.bias = bias; p.out = out; p.act = act & 0xff; p.slope = slope; p.prepacked = prepacked;
    p.sel = dcn_halo_always();
    if (rvsr_gemm_mode_now() != 1 && workspace != nullptr) {  // bf16x3 second-generation kernel
        // `probe`: three zeroed device counters; filled with the offset statistic that selects the tile halo on the device (and that
        // the backward of the same layer reuses)
        size_t nprobe = 0;
        if (probe != nullptr && d.cpg % 8 == 0 && d.stride == 1 && d.dil == 1) nprobe = rvsr_launch_dcn_offset_probe(d, probe, st);
        const int rc = rvsr_launch_dcn_fwd2(p, workspace, workspace_bytes, st, nprobe ? probe : nullptr, nprobe, halo_hint);
        if (rc != RVSR_ERR_UNSUPPORTED) return rc;
    }
    const int nty = (d.Ho + 3) / 4;
    const bool c8 = d.cpg % DCN_CC == 0;  // every chunk of 8 channels shares one offset set
#define LAUNCH_FWD(MT)                                                                                              \
    do {                                                                                                            \
        const size_t lds = sizeof(float) * (DCN_KC * DCN_NPX + DCN_KC * (MT * 32 + 1));                             \
        const dim3 grid(d.ntx * nty, (d.Co + MT * 32 - 1) / (MT * 32), d.B);                                        \
        if (c8) {                                                                                                   \
            if (set_lds(dcn_fwd_kernel<MT, 8>, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd: cannot reserve %zu B of LDS", lds); \
            hipLaunchKernelGGL((dcn_fwd_kernel<MT, 8>), grid, dim3(RVSR_WG), lds, st, p);                           \
        } else {                                                                                                    \
            if (set_lds(dcn_fwd_kernel<MT, 0>, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd: cannot reserve %zu B of LDS", lds); \
            hipLaunchKernelGGL((dcn_fwd_kernel<MT, 0>), grid, dim3(RVSR_WG), lds, st, p);                           \
        }                                                                                                           \
    } while (0)
    if (d.Co <= 32)
        LAUNCH_FWD(1);
    else if (d.Co <= 64)
        LAUNCH_FWD(2);
    else
        LAUNCH_FWD(4);
#undef LAUNCH_FWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

static int bww_P(int ntiles, int gy, int gz) {
    int P = 256 / (gy * gz);
    if (P < 1) P = 1;
    if (P > ntiles) P = ntiles;
    return P;
}

extern "C" size_t rvsr_modulated_deform_conv_forward_workspace_bytes(int channels, int channels_out) {
    return rvsr_dcn_fwd2_workspace_bytes(channels_out, channels);
}

// Can the dcn_bwdin6 / dcn_bwdw6 pair take a call of this geometry?  (The workspace query has no deformable_groups argument: the offset planes of
// one batch element are bounded by their worst case, 18 planes per 8 channels; rvsr_launch_dcn_bwdin6 repeats the exact test.)  The query sizes the
// hand-off buffer and the backward offers it to the pair by this one predicate.
static bool dcn_bwd6_pair_possible(int channels, int height, int width, int channels_out, int stride, int dil) {
    const size_t planes = (size_t)(channels / 8) * 18 > (size_t)channels ? (size_t)(channels / 8) * 18 : (size_t)channels;
    return stride == 1 && dil == 1 && channels % 8 == 0 && channels_out <= 128 &&
           planes * (size_t)height * width * sizeof(float) < ((size_t)1 << 31);
}
// everything but the operand buffer dcn_bwdin6 hands to dcn_bwdw6 (that buffer sits behind it, 256-byte aligned)
static size_t dcn_backward_workspace_base(int batch, int channels, int height, int width, int channels_out, int stride, int pad, int dil) {
    const int Ho = (height + 2 * pad - (dil * 2 + 1)) / stride + 1, Wo = (width + 2 * pad - (dil * 2 + 1)) / stride + 1;
    const int ntiles = batch * ((Ho + 3) / 4) * ((Wo + 31) / 32);
    const int gy = (channels_out + 63) / 64, gz = (channels + DCN_CC - 1) / DCN_CC;
    const size_t Q = 8 * (size_t)bww_P(ntiles, gy, gz);
    const size_t a = sizeof(float) * Q * ((size_t)channels_out * channels * 9 + channels_out);
    size_t b2 = rvsr_dcn_bwdin5_workspace_bytes(channels_out, channels);   // weight image + column norms + the probe counters
    const size_t b6 = rvsr_dcn_bwdin6_workspace_bytes(channels_out, channels);
    if (b6 > b2) b2 = b6;
    const size_t w6 = rvsr_dcn_bwdw6_workspace_bytes(channels_out, channels);
    if (w6 > b2) b2 = w6;
    return ((a > b2 ? a : b2) + 255) & ~(size_t)255;
}
extern "C" size_t rvsr_modulated_deform_conv_backward_workspace_bytes(int batch, int channels, int height, int width,
                                                                      int channels_out, int stride, int pad, int dil) {
    const int Ho = (height + 2 * pad - (dil * 2 + 1)) / stride + 1, Wo = (width + 2 * pad - (dil * 2 + 1)) / stride + 1;
    size_t n = dcn_backward_workspace_base(batch, channels, height, width, channels_out, stride, pad, dil);
    // the gOut^T hand-off of the dcn_bwdin6 / dcn_bwdw6 pair (hi + lo bf16 copy of gOut, 8-row x 32-column tiles: ~575 MiB at B = 40, Co = 64,
    // 180 x 320): only for calls that pair can take
    if (dcn_bwd6_pair_possible(channels, height, width, channels_out, stride, dil)) n += rvsr_dcn_bwd6_agt_bytes(batch, channels_out, Ho, Wo);
    return n;
}

static int dcn_backward_impl(DcnGeom& d, const float* weight, const float* gout, const float* gact, float gact_slope,
                             float* gx, float* goff, size_t goff_bs, float* gmask, size_t gmask_bs, float* gw, float* gb,
                             void* workspace, size_t workspace_bytes, hipStream_t st, const unsigned* probe = nullptr) {
    const size_t need = rvsr_modulated_deform_conv_backward_workspace_bytes(d.B, d.C, d.H, d.W, d.Co, d.stride, d.pad, d.dil);
    if (gw && (!workspace || workspace_bytes < need)) FAIL(RVSR_ERR_WORKSPACE, "dcn backward: workspace %zu B < %zu B", workspace_bytes, need);
    TView g;
    g.p = gout; g.act = gact; g.slope = gact_slope; g.C = d.Co; g.Hs = g.Hv = d.Ho; g.Ws = g.Wv = d.Wo; g.mode = 0;
    const int nty = (d.Ho + 3) / 4;
    // dcn_bwdin6 + dcn_bwdw6 run as a pair: the first leaves gOut (x act') behind as the matrix-core operands of the second
    // developer A/B switch RVSR_DCN_BWD: 7 (default) = dcn_bwdin6 + dcn_bwdw6, 64 = dcn_bwdin6 + dcn_bwdw4, 6 = dcn_bwdin5 + dcn_bwdw4 (round 4's pair)
    static const int gen_env = [] { const char* e = getenv("RVSR_DCN_BWD"); return e ? atoi(e) : 7; }();
    static const int genw = gen_env == 7 ? 6 : 4;
    void* agt = nullptr;
    bool agt_written = false;
    if (gx && gw && rvsr_gemm_mode_now() != 1 && genw >= 6 && dcn_bwd6_pair_possible(d.C, d.H, d.W, d.Co, d.stride, d.dil))
        agt = (unsigned char*)workspace + dcn_backward_workspace_base(d.B, d.C, d.H, d.W, d.Co, d.stride, d.pad, d.dil);
    if (gx || goff || gmask) {
        if (!gx || !goff || !gmask) FAIL(RVSR_ERR_BAD_ARG, "dcn backward: grad_input/grad_offset/grad_mask must be given together");
        int rc2 = RVSR_ERR_UNSUPPORTED;
        if (rvsr_gemm_mode_now() != 1) {
            static const int gen = gen_env == 64 ? 7 : gen_env;
            static const int halo = [] { const char* e = getenv("RVSR_DCN5_HALO"); return e ? atoi(e) : -1; }();   // -1: selected on the device
            if (gen >= 7) {
                rc2 = rvsr_launch_dcn_bwdin6(d, weight, g, gx, goff, goff_bs, gmask, gmask_bs, workspace, workspace_bytes, st, halo, probe, agt);
                agt_written = rc2 == RVSR_OK && agt != nullptr;
            }
            if (rc2 == RVSR_ERR_UNSUPPORTED && gen >= 6) rc2 = rvsr_launch_dcn_bwdin5(d, weight, g, gx, goff, goff_bs, gmask, gmask_bs, workspace, workspace_bytes, st, halo, probe);
        }
        if (rc2 != RVSR_ERR_UNSUPPORTED && rc2 != RVSR_OK) return rc2;
        DcnBwdInParams p;
        p.d = d; p.w = weight; p.g = g; p.gx = gx; p.goff = goff; p.gmask = gmask; p.goff_bs = goff_bs; p.gmask_bs = gmask_bs;
        const int CoP = (d.Co + 1) & ~1;
        const size_t lds = sizeof(float) * ((size_t)CoP * DCN_NPX + (size_t)CoP * DCN_KC + DCN_KC * DCN_NPX + DCN_CC * 14 * 42);
        // (the LDS bound belongs to the first-generation kernel only: checked where that kernel is actually launched)
        if (rc2 != RVSR_OK && lds > 160 * 1024) FAIL(RVSR_ERR_UNSUPPORTED, "dcn backward: channels_out %d needs %zu B of LDS", d.Co, lds);
        if (rc2 == RVSR_OK) {
            // done by dcn_bwdin6 / dcn_bwdin5
        } else if (d.cpg % DCN_CC == 0) {
            if (set_lds(dcn_bwd_input_kernel<8>, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_bwd_input: cannot reserve %zu B of LDS", lds);
            hipLaunchKernelGGL(dcn_bwd_input_kernel<8>, dim3(d.ntx * nty, 1, d.B), dim3(RVSR_WG), lds, st, p);
        } else {
            if (set_lds(dcn_bwd_input_kernel<0>, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_bwd_input: cannot reserve %zu B of LDS", lds);
            hipLaunchKernelGGL(dcn_bwd_input_kernel<0>, dim3(d.ntx * nty, 1, d.B), dim3(RVSR_WG), lds, st, p);
        }
    }
    if (gw && agt_written) {
        const int rc6 = rvsr_launch_dcn_bwdw6(d, agt, gw, gb, workspace, workspace_bytes, st);
        if (rc6 == RVSR_OK) gw = nullptr;   // done
        else if (rc6 != RVSR_ERR_UNSUPPORTED) return rc6;
    }
    if (gw) {
        DcnBwdWParams p;
        p.d = d; p.g = g; p.nty = nty;
        const int gy = (d.Co + 63) / 64, gz = (d.C + DCN_CC - 1) / DCN_CC;
        p.P = bww_P(d.B * nty * d.ntx, gy, gz);
        size_t Q = 4 * (size_t)p.P;
        const size_t nw = (size_t)d.Co * d.C * 9;
        p.part = (float*)workspace;
        int q2 = -1;
        if (d.cpg % DCN_CC == 0) {  // second-generation builder (LDS x tile, 8 waves): 8P partials
            float* bp2 = gb ? p.part + (size_t)8 * p.P * nw : nullptr;
            q2 = rvsr_launch_dcn_bwdw2(d, g, p.part, bp2, p.P, nty, gy, gz, st);
            if (q2 == -2) FAIL(RVSR_ERR_LAUNCH, "dcn_bwdw2: cannot reserve LDS");
            if (q2 > 0) {
                Q = (size_t)q2;
                p.bpart = bp2;
            }
        }
        if (q2 <= 0) {
            p.bpart = gb ? p.part + Q * nw : nullptr;
            const size_t lds = sizeof(float) * (DCN_NPX * 65 + DCN_NPX * 97);
            if (set_lds(dcn_bwd_weight_kernel<0>, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_bwd_weight: cannot reserve %zu B of LDS", lds);
            hipLaunchKernelGGL(dcn_bwd_weight_kernel<0>, dim3(p.P, gy, gz), dim3(RVSR_WG), lds, st, p);
        }
        // the reference accumulates into caller-zeroed gW/gBias (cpp:659-671) -> accumulate = 1
        rvsr_launch_reduce(p.part, (int)Q, nw, gw, 1, st, p.bpart, (size_t)d.Co, gb);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn backward launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// ---- C ABI: drop-in for the reference's pybind functions (deform_conv_cuda.cpp:490-685, 687-701) ----
extern "C" int rvsr_modulated_deform_conv_forward(const float* input, const float* weight, const float* bias,
                                                  const float* offset, const float* mask, float* output, int batch,
                                                  int channels, int height, int width, int channels_out, int kernel_h,
                                                  int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                                  int dilation_h, int dilation_w, int group, int deformable_group,
                                                  int with_bias, void* workspace, size_t workspace_bytes, void* stream) {
    DcnGeom d;
    const char* why = "";
    if (!weight || !output) FAIL(RVSR_ERR_BAD_ARG, "modulated_deform_conv_forward: null weight/output");
    int rc = fill_geom(d, input, offset, 0, mask, 0, 0, batch, channels, height, width, channels_out, kernel_h, kernel_w,
                       stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, &why);
    if (rc) FAIL(rc, "modulated_deform_conv_forward: %s", why);
    d.off_bs = (size_t)2 * 9 * deformable_group * d.Ho * d.Wo;
    d.mask_bs = (size_t)9 * deformable_group * d.Ho * d.Wo;
    return dcn_forward_impl(d, weight, with_bias ? bias : nullptr, output, 0, 0.f, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int rvsr_modulated_deform_conv_backward(const float* input, const float* weight, const float* bias,
                                                   const float* offset, const float* mask, float* grad_input,
                                                   float* grad_weight, float* grad_bias, float* grad_offset,
                                                   float* grad_mask, const float* grad_output, int batch, int channels,
                                                   int height, int width, int channels_out, int kernel_h, int kernel_w,
                                                   int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                                                   int dilation_w, int group, int deformable_group, int with_bias,
                                                   void* workspace, size_t workspace_bytes, void* stream) {
    (void)bias;
    DcnGeom d;
    const char* why = "";
    if (!weight || !grad_output) FAIL(RVSR_ERR_BAD_ARG, "modulated_deform_conv_backward: null weight/grad_output");
    int rc = fill_geom(d, input, offset, 0, mask, 0, 0, batch, channels, height, width, channels_out, kernel_h, kernel_w,
                       stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, &why);
    if (rc) FAIL(rc, "modulated_deform_conv_backward: %s", why);
    d.off_bs = (size_t)2 * 9 * deformable_group * d.Ho * d.Wo;
    d.mask_bs = (size_t)9 * deformable_group * d.Ho * d.Wo;
    return dcn_backward_impl(d, weight, grad_output, nullptr, 0.f, grad_input, grad_offset, d.off_bs, grad_mask, d.mask_bs,
                             grad_weight, with_bias ? grad_bias : nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- DCNv1 (deform_conv_cuda.cpp:152-488: deform_conv_forward_cuda / deform_conv_backward_input_cuda / deform_conv_backward_parameters_cuda).
// The reference's v1 kernels (kernel.cu:190-465, helpers :84-188) are its modulated kernels (:571-767, helpers :467-568) without the mask
// factor and without a bias: same sampling positions, same zero-outside bilinear rule, same 5x5 scatter window and -2 sentinel.  The three
// entry points therefore run the modulated kernels on a mask of ones kept in the workspace.  `im2col_step` only shapes the reference's column
// buffer; it is checked (it has to divide the batch, deform_conv.py:41) and otherwise unused -- there is no column buffer here.
// Workspace: [ones: B*dg*9*Ho*Wo floats][grad_mask scratch: the same][the modulated operator's workspace]. ----
static size_t dcn1_mask_bytes(int batch, int height, int width, int kh, int kw, int stride, int pad, int dil, int dg) {
    const int Ho = (height + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1, Wo = (width + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const size_t n = (size_t)batch * dg * kh * kw * (Ho > 0 ? Ho : 0) * (Wo > 0 ? Wo : 0) * sizeof(float);
    return (n + 255) & ~(size_t)255;
}
extern "C" size_t rvsr_deform_conv_workspace_bytes(int batch, int channels, int height, int width, int channels_out, int kW, int kH,
                                                   int dW, int padW, int dilationW, int deformable_group) {
    const size_t fw = rvsr_modulated_deform_conv_forward_workspace_bytes(channels, channels_out);
    const size_t bw = rvsr_modulated_deform_conv_backward_workspace_bytes(batch, channels, height, width, channels_out, dW, padW, dilationW);
    return 2 * dcn1_mask_bytes(batch, height, width, kH, kW, dW, padW, dilationW, deformable_group) + (fw > bw ? fw : bw);
}
static int dcn1_setup(DcnGeom& d, const char* who, const float* input, const float* offset, int batch, int channels, int height, int width,
                      int channels_out, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                      int deformable_group, int im2col_step, void* workspace, size_t workspace_bytes, hipStream_t st, float** ones,
                      float** scratch, void** ws2, size_t* ws2_bytes) {
    const char* why = "";
    if (im2col_step <= 0 || batch % im2col_step != 0) FAIL(RVSR_ERR_BAD_ARG, "%s: im2col step must divide batchsize", who);
    const size_t mb = dcn1_mask_bytes(batch, height, width, kH, kW, dW, padW, dilationW, deformable_group);
    if (!workspace || workspace_bytes < 2 * mb) FAIL(RVSR_ERR_WORKSPACE, "%s: workspace %zu B < %zu B", who, workspace_bytes, 2 * mb);
    *ones = (float*)workspace;
    *scratch = (float*)((char*)workspace + mb);
    *ws2 = (char*)workspace + 2 * mb;
    *ws2_bytes = workspace_bytes - 2 * mb;
    int rc = fill_geom(d, input, offset, 0, *ones, 0, 0, batch, channels, height, width, channels_out, kH, kW, dH, dW, padH, padW, dilationH,
                       dilationW, group, deformable_group, &why);
    if (rc) FAIL(rc, "%s: %s", who, why);
    d.off_bs = (size_t)2 * 9 * deformable_group * d.Ho * d.Wo;
    d.mask_bs = (size_t)9 * deformable_group * d.Ho * d.Wo;
    if (hipMemsetD32Async((hipDeviceptr_t)*ones, 0x3f800000, (size_t)batch * d.mask_bs, st) != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "%s: fill", who);
    return RVSR_OK;
}
extern "C" int rvsr_deform_conv_forward(const float* input, const float* weight, const float* offset, float* output, int batch, int channels,
                                        int height, int width, int channels_out, int kW, int kH, int dW, int dH, int padW, int padH,
                                        int dilationW, int dilationH, int group, int deformable_group, int im2col_step, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (!weight || !output) FAIL(RVSR_ERR_BAD_ARG, "deform_conv_forward: null weight/output");
    DcnGeom d; float *ones, *scratch; void* ws2; size_t n2;
    int rc = dcn1_setup(d, "deform_conv_forward", input, offset, batch, channels, height, width, channels_out, kW, kH, dW, dH, padW, padH, dilationW,
                        dilationH, group, deformable_group, im2col_step, workspace, workspace_bytes, (hipStream_t)stream, &ones, &scratch, &ws2, &n2);
    if (rc) return rc;
    return dcn_forward_impl(d, weight, nullptr, output, 0, 0.f, ws2, n2, (hipStream_t)stream);
}
// grad_input is accumulated into (the caller zeroes it, deform_conv.py:62), grad_offset is written
extern "C" int rvsr_deform_conv_backward_input(const float* input, const float* offset, const float* grad_output, float* grad_input,
                                               float* grad_offset, const float* weight, int batch, int channels, int height, int width,
                                               int channels_out, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW,
                                               int dilationH, int group, int deformable_group, int im2col_step, void* workspace,
                                               size_t workspace_bytes, void* stream) {
    if (!weight || !grad_output || !grad_input || !grad_offset) FAIL(RVSR_ERR_BAD_ARG, "deform_conv_backward_input: null argument");
    DcnGeom d; float *ones, *scratch; void* ws2; size_t n2;
    int rc = dcn1_setup(d, "deform_conv_backward_input", input, offset, batch, channels, height, width, channels_out, kW, kH, dW, dH, padW, padH,
                        dilationW, dilationH, group, deformable_group, im2col_step, workspace, workspace_bytes, (hipStream_t)stream, &ones, &scratch,
                        &ws2, &n2);
    if (rc) return rc;
    return dcn_backward_impl(d, weight, grad_output, nullptr, 0.f, grad_input, grad_offset, d.off_bs, scratch, d.mask_bs, nullptr, nullptr, ws2, n2,
                             (hipStream_t)stream);
}
// grad_weight is accumulated into (the caller zeroes it, deform_conv.py:71); `scale` multiplies the contribution (the reference passes 1)
extern "C" int rvsr_deform_conv_backward_parameters(const float* input, const float* offset, const float* grad_output, float* grad_weight,
                                                    int batch, int channels, int height, int width, int channels_out, int kW, int kH, int dW,
                                                    int dH, int padW, int padH, int dilationW, int dilationH, int group, int deformable_group,
                                                    float scale, int im2col_step, void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad_output || !grad_weight) FAIL(RVSR_ERR_BAD_ARG, "deform_conv_backward_parameters: null argument");
    if (scale != 1.f) FAIL(RVSR_ERR_UNSUPPORTED, "deform_conv_backward_parameters: scale %g (only 1 is built; deform_conv.py:76 passes 1)", (double)scale);
    DcnGeom d; float *ones, *scratch; void* ws2; size_t n2;
    int rc = dcn1_setup(d, "deform_conv_backward_parameters", input, offset, batch, channels, height, width, channels_out, kW, kH, dW, dH, padW, padH,
                        dilationW, dilationH, group, deformable_group, im2col_step, workspace, workspace_bytes, (hipStream_t)stream, &ones, &scratch,
                        &ws2, &n2);
    if (rc) return rc;
    return dcn_backward_impl(d, nullptr /* the weights: not read by the weight-gradient kernels */, grad_output, nullptr, 0.f, nullptr, nullptr, d.off_bs, nullptr,
                             d.mask_bs, grad_weight, nullptr, ws2, n2, (hipStream_t)stream);
}

// ---- fused ModulatedDeformConvPack core (deform_conv.py:274-292): `om` is the raw 3*dg*9-channel output of
// conv_offset_mask; chunk/cat is pure addressing (channels [0,2*dg*9) = offsets, the rest = mask logits) and the
// sigmoid runs in-kernel.  Optional LeakyReLU/ReLU epilogue (EDVR_arch.py:107,130). ----
extern "C" int rvsr_dcn_pack_forward(const float* input, const float* weight, const float* bias, const float* om,
                                     float* output, int batch, int channels, int height, int width, int channels_out,
                                     int stride, int pad, int dilation, int deformable_group, int act, float slope,
                                     void* probe, void* workspace, size_t workspace_bytes, void* stream) {
    DcnGeom d;
    const char* why = "";
    if (!weight || !output) FAIL(RVSR_ERR_BAD_ARG, "dcn_pack_forward: null weight/output");
    int rc = fill_geom(d, input, om, 0, om, 0, 1, batch, channels, height, width, channels_out, 3, 3, stride, stride, pad, pad,
                       dilation, dilation, 1, deformable_group, &why);
    if (rc) FAIL(rc, "dcn_pack_forward: %s", why);
    const size_t hw = (size_t)d.Ho * d.Wo;
    d.off_bs = d.mask_bs = (size_t)27 * deformable_group * hw;
    d.mask = om + (size_t)18 * deformable_group * hw;
    // act bit 8: `workspace` already holds the packed weight image (rvsr_dcn_pack_weights), skip the per-call pack
    // act bits 10..13: halo of the forward's LDS tile chosen by the caller (3 / 7 / 11; 0 = from `probe`, or 3 px without one)
    return dcn_forward_impl(d, weight, bias, output, act, slope, workspace, workspace_bytes, (hipStream_t)stream, (act >> 8) & 1, (unsigned*)probe,
                            (act >> 10) & 15);
}

// The sampled offset statistic on its own (8 zeroed uint32 on the device, 6 used): lets a caller keep the counters, e.g. to choose the
// forward's halo of the NEXT step on the host without stalling the GPU (realvsr_amd.functional), and hand them to the backward.
extern "C" int rvsr_dcn_offset_probe(const float* om, int batch, int height_out, int width_out, int deformable_group, void* probe, void* stream) {
    if (!om || !probe || batch <= 0 || deformable_group <= 0) FAIL(RVSR_ERR_BAD_ARG, "dcn_offset_probe: null/empty argument");
    DcnGeom d;
    d.offset = om; d.off_bs = (size_t)27 * deformable_group * height_out * width_out;
    d.B = batch; d.C = 8 * deformable_group; d.cpg = 8; d.Ho = height_out; d.Wo = width_out;
    rvsr_launch_dcn_offset_probe(d, (unsigned*)probe, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_offset_probe launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

extern "C" int rvsr_dcn_pack_backward(const float* input, const float* weight, const float* om, const float* grad_output,
                                      const float* act_out, float act_slope, float* grad_input, float* grad_weight,
                                      float* grad_bias, float* grad_om, int batch, int channels, int height, int width,
                                      int channels_out, int stride, int pad, int dilation, int deformable_group,
                                      const void* probe, void* workspace, size_t workspace_bytes, void* stream) {
    DcnGeom d;
    const char* why = "";
    if (!weight || !grad_output) FAIL(RVSR_ERR_BAD_ARG, "dcn_pack_backward: null weight/grad_output");
    int rc = fill_geom(d, input, om, 0, om, 0, 1, batch, channels, height, width, channels_out, 3, 3, stride, stride, pad, pad,
                       dilation, dilation, 1, deformable_group, &why);
    if (rc) FAIL(rc, "dcn_pack_backward: %s", why);
    const size_t hw = (size_t)d.Ho * d.Wo;
    d.off_bs = d.mask_bs = (size_t)27 * deformable_group * hw;
    d.mask = om + (size_t)18 * deformable_group * hw;
    return dcn_backward_impl(d, weight, grad_output, act_out, act_slope, grad_input, grad_om, d.off_bs,
                             grad_om ? grad_om + (size_t)18 * deformable_group * hw : nullptr, d.off_bs, grad_weight, grad_bias,
                             workspace, workspace_bytes, (hipStream_t)stream, (const unsigned*)probe);
}
