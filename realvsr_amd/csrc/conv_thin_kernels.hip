// conv_thin_kernels.hip -- 3x3 / stride-1 convolutions with at most four output channels (EDVR's conv_last: 64 -> 3 at the
// HR resolution, 7.4 M pixels per batch of 8) on the vector ALU, exact f32.
//
// On the matrix-core kernels such a layer runs a 32-row M tile with 3 live rows: the weight gradient of conv_last took
// 2.0 ms per step in conv_wgrad2_kernel (as long as a 64 -> 64 layer), almost all of it staging the X tile as bf16 hi/lo
// for an MFMA phase that is 90 % padding.  The work itself is 1728 FMAs per pixel -- 0.16 ms of vector ALU at peak -- and the
// layer is bound by reading X once (1.9 GB, ~0.35 ms).  So: plain f32 FMAs, X tile in LDS as f32 (no split), buffer loads with
// range-check zero padding (see conv_fwd5_kernel), deterministic partial sums reduced by rvsr_reduce_partials_kernel.
// Measured: 0.84 ms per call at 8 x 64 x 720 x 1280 (conv_wgrad2: 1.45 ms); 30 TFLOP/s, issue-bound (288 packed FMAs + ~300
// other instructions per thread and tile, two barriers per tile).
#include "conv_common.h"

#define THIN_T 256           // threads per workgroup
#define THIN_XW 72           // staged columns of a row: image columns x0-4 .. x0+67
#define THIN_XPL 452         // dwords per channel plane of the X tile (6 rows x 72 + pad; == 4 mod 64: the 16 channel lanes of a
                             // 16-byte LDS read fall on distinct bank quads)

// ------------------------------------------------------------------------------------------
// Weight / bias gradient:  gW[o][c][tap] = sum_{b,y,x} G[b,o,y,x] * X[b,c,y+dy-1,x+dx-1],  Co <= 4, C % 16 == 0.
// Workgroup = 256 threads = 16 channels x 16 pixel segments (row r of the 4 x 64 pixel tile, 16 columns); the workgroup walks
// its tiles once per 16-channel group, so a thread carries 36 accumulators (4 o x 9 taps) of ONE channel; after a group's
// tiles the 16 segment partials of a channel are summed through LDS (fixed order) and written to part[blockIdx.x].
template <bool ACT>
__global__ __launch_bounds__(THIN_T, 3) void conv_wgrad_thin_kernel(const ConvWgradParams p) {
    __shared__ __attribute__((aligned(16))) float smem[16 * 16 * 37];   // 37.9 KB: the tile images, later the reduction scratch
    float* const xs = smem;                                             // [16 ch][6 rows][72 cols] (THIN_XPL dwords per channel)
    float* const gs = smem + 16 * THIN_XPL;                             // [4 rows][64 cols][4 o], x act', zero beyond Co / image
    float (*red)[16][37] = reinterpret_cast<float (*)[16][37]>(smem);   // [segment][channel][accumulator] (+1: no bank conflicts)
    static_assert(16 * THIN_XPL + 4 * 4 * 64 <= 16 * 16 * 37, "tile images fit the scratch");
    const int tid = threadIdx.x, cl = tid & 15, seg = tid >> 4, r = seg >> 2, cb = (seg & 3) * 16;
    const int C = p.x.a.C, H = p.x.a.Hs, W = p.x.a.Ws, Co = p.Co;
    const unsigned HW4 = 4u * (unsigned)(H * W);
    const int ntx = (W + 63) / 64, nty = (H + 3) / 4, ntiles = p.B * nty * ntx;
    constexpr unsigned OOB = 0x80000000u;

    // tile-independent staging items.  X: 16 ch x 6 rows x 18 float4 = 1728 = 6.75 per thread; G: 4 o x 4 rows x 16 float4 = 1 per thread
    unsigned x_vo[7];
    int x_lds[7], x_row[7], x_col[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int it = tid + i * THIN_T;
        const bool live = it < 16 * 6 * 18;
        const int c = live ? it / 108 : 0, rem = live ? it - c * 108 : 0, row = rem / 18, q = rem - row * 18;
        x_row[i] = row - 1;
        x_col[i] = 4 * q - 4;
        x_lds[i] = live ? c * THIN_XPL + row * THIN_XW + 4 * q : -1;
        x_vo[i] = live ? (unsigned)c * HW4 + 4u * (unsigned)(row * W + 4 * q) : OOB;   // view starts one row + 4 px before the image
    }
    const int g_o = tid >> 6, g_row = (tid >> 4) & 3, g_q = tid & 15;
    const unsigned g_vo = g_o < Co ? (unsigned)g_o * HW4 + 4u * (unsigned)(g_row * W + 4 * g_q) : OOB;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef float f32x2v __attribute__((ext_vector_type(2)));

    float bacc = 0.f;   // bias gradient: thread (o = g_o, row, float4 column) sums the G values it stages (first channel group only)
    for (int cg = 0; cg < C / 16; ++cg) {
        // accumulators as (o0, o1) / (o2, o3) pairs per tap: with the G tile stored o-interleaved a pixel's four output
        // channels are one 16-byte LDS read and every FMA is a packed one with the x value broadcast -- no operand shuffling
        // (the [o][pixel] layout cost one v_mov per packed FMA)
        f32x2v acc[9][2];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = f32x2v{0.f, 0.f};
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int b = tile / (nty * ntx), trem = tile - b * (nty * ntx), ty = trem / ntx;
            const int y0 = ty * 4, x0 = (trem - ty * ntx) * 64;
            const __amdgpu_buffer_rsrc_t x_rs = buf_view_2g(p.x.a.p + ((size_t)b * C + cg * 16) * H * W - (W + 4));
            const __amdgpu_buffer_rsrc_t g_rs = buf_view_2g(p.g.p + (size_t)b * Co * H * W);
            const __amdgpu_buffer_rsrc_t s_rs = buf_view_2g(ACT ? p.g.act + (size_t)b * Co * H * W : p.g.p);
            const unsigned so = 4u * (unsigned)(y0 * W + x0);
            // (a register prefetch of the next tile was measured: no change -- three resident workgroups per CU cover the round trip)
            f32x4v xv[7], gv, sv;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int gy = y0 + x_row[i], gx = x0 + x_col[i];
                const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;   // (W % 4 == 0: a float4 is inside or outside)
                xv[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)(ok ? x_vo[i] : OOB), (int)so, 0));
            }
            {
                const bool ok = y0 + g_row < H && x0 + 4 * g_q < W;
                gv = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(g_rs, (int)(ok ? g_vo : OOB), (int)so, 0));
                if (ACT) sv = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(s_rs, (int)(ok ? g_vo : OOB), (int)so, 0));
            }
            __syncthreads();   // the previous tile's readers are done with xs / gs
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (x_lds[i] >= 0) *reinterpret_cast<f32x4v*>(xs + x_lds[i]) = xv[i];
            if (ACT) {
                gv.x *= sv.x > 0.f ? 1.f : p.g.slope; gv.y *= sv.y > 0.f ? 1.f : p.g.slope;
                gv.z *= sv.z > 0.f ? 1.f : p.g.slope; gv.w *= sv.w > 0.f ? 1.f : p.g.slope;
            }
            {   // gs[row][col][o]
                float* gd = gs + ((g_row * 64 + 4 * g_q) * 4 + g_o);
                gd[0] = gv.x; gd[4] = gv.y; gd[8] = gv.z; gd[12] = gv.w;
            }
            if (cg == 0) bacc += (gv.x + gv.y) + (gv.z + gv.w);
            __syncthreads();

            // this thread: channel cl, output row r, columns cb .. cb+15.  The G values are re-read per dy and four pixels at a
            // time (48 broadcast LDS reads per tile instead of 16): 48 registers less (154 VGPRs, three workgroups per CU).
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                float xr[24];   // staged columns cb .. cb+23 = image columns x0+cb-4 .. ; output column i, tap dx reads xr[i + dx + 3]
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const f32x4v v = *reinterpret_cast<const f32x4v*>(xs + cl * THIN_XPL + (r + dy) * THIN_XW + cb + 4 * k);
                    xr[4 * k] = v.x; xr[4 * k + 1] = v.y; xr[4 * k + 2] = v.z; xr[4 * k + 3] = v.w;
                }
#pragma unroll
                for (int ic = 0; ic < 4; ++ic) {
                    f32x4v gq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) gq[j] = *reinterpret_cast<const f32x4v*>(gs + (r * 64 + cb + 4 * ic + j) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * ic + j;
                        const f32x2v g01 = {gq[j].x, gq[j].y}, g23 = {gq[j].z, gq[j].w};
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float x = xr[i + dx + 3];
                            acc[dy * 3 + dx][0] += g01 * x;
                            acc[dy * 3 + dx][1] += g23 * x;
                        }
                    }
                }
            }
        }
        // ---- sum the 16 segment partials of every channel (fixed order) and write this workgroup's partial
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            red[seg][cl][0 * 9 + t] = acc[t][0].x; red[seg][cl][1 * 9 + t] = acc[t][0].y;
            red[seg][cl][2 * 9 + t] = acc[t][1].x; red[seg][cl][3 * 9 + t] = acc[t][1].y;
        }
        __syncthreads();
        for (int e = tid; e < 16 * 36; e += THIN_T) {
            const int c = e / 36, a = e - c * 36, o = a / 9, t = a - o * 9;
            float s = 0.f;
#pragma unroll
            for (int sg = 0; sg < 16; ++sg) s += red[sg][c][a];
            if (o < Co) p.part[(((size_t)blockIdx.x * Co + o) * C + cg * 16 + c) * 9 + t] = s;
        }
    }
    if (p.bpart != nullptr) {
        // 64 threads share an output channel: wave-level sum, one value per wave = per o
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) bacc += __shfl_xor(bacc, s);
        if ((tid & 63) == 0 && g_o < Co) p.bpart[(size_t)blockIdx.x * Co + g_o] = bacc;
    }
}

int rvsr_conv_wgrad_thin_P(int B, int Hout, int Wout) {
    const long ntiles = (long)B * ((Hout + 3) / 4) * ((Wout + 63) / 64);
    return (int)(ntiles < 768 ? ntiles : 768);   // 3 workgroups of 4 waves per CU
}

int rvsr_launch_conv_wgrad_thin(const ConvWgradParams& p, hipStream_t st) {
    auto k = p.g.act != nullptr ? conv_wgrad_thin_kernel<true> : conv_wgrad_thin_kernel<false>;
    hipLaunchKernelGGL(k, dim3(p.P), dim3(THIN_T), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad_thin launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// ------------------------------------------------------------------------------------------
// Forward:  out[b,o,y,x] = act(bias[o] + sum_{c,tap} W[o][c][tap] * X[b,c,y+dy-1,x+dx-1]) (+ residual),  Co <= 4, C % 8 == 0.
// conv_last at the HR resolution on conv_fwd5_kernel<1>: 1.05 ms per call (a 32-row M tile with 3 live rows, bf16 hi/lo staging of
// 64 channels) against ~0.4 ms of HBM time.  Here one workgroup computes an 8 x 128 pixel tile, a thread four consecutive pixels
// of one row for all output channels; the input passes through LDS as f32 eight channels at a time, the weights sit in LDS as
// [c][tap][o0..o3] so that one broadcast 16-byte read feeds two packed FMAs per pixel.
#define THIN_FW 136          // staged columns per row: image columns x0-4 .. x0+131
#define THIN_FPL (10 * THIN_FW)
__global__ __launch_bounds__(THIN_T, 2) void conv_fwd_thin_kernel(const ConvFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* const xs = fsm;                    // [8 ch][10 rows][136 cols]
    float* const ws = fsm + 8 * THIN_FPL;     // [C][9][4]
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const TView& va = p.in.a;
    const int tid = threadIdx.x, r = tid >> 5, col4 = (tid & 31) * 4;
    const int C = va.C, H = p.Hout, W = p.Wout, Co = p.Co;
    const unsigned HW4 = 4u * (unsigned)(H * W);
    const int ntx = (W + 127) / 128, nty = (H + 7) / 8;
    const int b = blockIdx.x / (nty * ntx), trem = blockIdx.x - b * (nty * ntx), ty = trem / ntx;
    const int y0 = ty * 8, x0 = (trem - ty * ntx) * 128;
    constexpr unsigned OOB = 0x80000000u;

    // weights -> LDS, [c][tap][o] with zeros for o >= Co
    for (int e = tid; e < C * 9 * 4; e += THIN_T) {
        const int o = e & 3, ct = e >> 2;   // ct = c * 9 + tap
        ws[e] = o < Co ? p.w[(size_t)o * C * 9 + ct] : 0.f;
    }
    // staging items of this thread (8 ch x 10 rows x 34 float4 = 2720 = 10.6 per thread)
    unsigned x_vo[11];
    int x_lds[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int it = tid + i * THIN_T;
        const bool live = it < 8 * 10 * 34;
        const int c = live ? it / 340 : 0, rem = live ? it - c * 340 : 0, row = rem / 34, q = rem - row * 34;
        const int gy = y0 + row - 1, gx = x0 + 4 * q - 4;
        const bool ok = live && gy >= 0 && gy < H && gx >= 0 && gx < W;   // (W % 4 == 0)
        x_lds[i] = live ? c * THIN_FPL + row * THIN_FW + 4 * q : -1;
        x_vo[i] = ok ? (unsigned)c * HW4 + 4u * (unsigned)(row * W + 4 * q) : OOB;   // view starts one row + 4 px before the image
    }
    const unsigned so = 4u * (unsigned)(y0 * W + x0);

    f32x2v acc[4][2];
    {
        const float b0 = p.bias && Co > 0 ? p.bias[0] : 0.f, b1 = p.bias && Co > 1 ? p.bias[1] : 0.f;
        const float b2 = p.bias && Co > 2 ? p.bias[2] : 0.f, b3 = p.bias && Co > 3 ? p.bias[3] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j][0] = f32x2v{b0, b1}; acc[j][1] = f32x2v{b2, b3}; }
    }
    for (int cg = 0; cg < C / 8; ++cg) {
        const __amdgpu_buffer_rsrc_t x_rs = buf_view_2g(va.p + ((size_t)b * C + cg * 8) * H * W - (W + 4));
        f32x4v xv[11];
#pragma unroll
        for (int i = 0; i < 11; ++i)
            xv[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)x_vo[i], (int)so, 0));
        __syncthreads();   // readers of the previous group are done (and, the first time, ws is complete)
#pragma unroll
        for (int i = 0; i < 11; ++i)
            if (x_lds[i] >= 0) *reinterpret_cast<f32x4v*>(xs + x_lds[i]) = xv[i];
        __syncthreads();
#pragma unroll 2
        for (int c = 0; c < 8; ++c) {
            const float* wc = ws + (cg * 8 + c) * 36;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* xrow = xs + c * THIN_FPL + (r + dy) * THIN_FW + col4;   // staged column col4 = image column x0+col4-4
                const f32x4v xa = *reinterpret_cast<const f32x4v*>(xrow), xb = *reinterpret_cast<const f32x4v*>(xrow + 4);
                const float x6[6] = {xa.w, xb.x, xb.y, xb.z, xb.w, xrow[8]};   // image columns x0+col4-1 .. x0+col4+4
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4v w4 = *reinterpret_cast<const f32x4v*>(wc + (dy * 3 + dx) * 4);
                    const f32x2v w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j][0] += w01 * x6[j + dx];
                        acc[j][1] += w23 * x6[j + dx];
                    }
                }
            }
        }
    }
    // ---- epilogue: activation, residual, one 16-byte store per output channel
    const int oy = y0 + r, ox = x0 + col4;
    if (oy >= H || ox >= W) return;
    const float neg = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o >= Co) break;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = o == 0 ? acc[j][0].x : (o == 1 ? acc[j][0].y : (o == 2 ? acc[j][1].x : acc[j][1].y));
            v[j] = a > 0.f ? a : a * neg;
        }
        const size_t idx = (((size_t)b * Co + o) * H + oy) * W + ox;
        if (p.res != nullptr) {
            const float4 rr = *reinterpret_cast<const float4*>(p.res + idx);
            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        *reinterpret_cast<float4*>(p.out1 + idx) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// 3x3 / stride 1 / plain view / single input and output / no act' / f32 weights in the reference layout
bool rvsr_conv_fwd_thin_ok(const ConvFwdParams& p, int ksize, int stride) {
    const TView& va = p.in.a;
    return ksize == 3 && stride == 1 && p.Co <= 4 && p.in.b.C == 0 && va.C % 8 == 0 && va.mode == 0 && va.act == nullptr &&
           p.out2 == nullptr && !p.ps && p.w_mode == 0 && p.Wout % 4 == 0 && va.Ws == p.Wout && va.Hs == p.Hout &&
           ((((uintptr_t)va.p) | ((uintptr_t)p.out1) | ((uintptr_t)p.res)) & 15) == 0 &&
           sizeof(float) * (size_t)p.Hout * p.Wout * (size_t)va.C < ((size_t)1 << 31) && va.C <= 256;
}
int rvsr_launch_conv_fwd_thin(const ConvFwdParams& p, hipStream_t st) {
    const size_t lds = sizeof(float) * (8 * THIN_FPL + (size_t)p.in.a.C * 36);
    if (set_lds(conv_fwd_thin_kernel, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_fwd_thin: cannot reserve %zu B of LDS", lds);
    const unsigned grid = (unsigned)p.B * ((p.Hout + 7) / 8) * ((p.Wout + 127) / 128);
    hipLaunchKernelGGL(conv_fwd_thin_kernel, dim3(grid), dim3(THIN_T), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_fwd_thin launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}
