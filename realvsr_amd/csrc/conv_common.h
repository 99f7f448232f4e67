// conv_common.h -- parameter blocks + epilogue shared by the f32 (conv_kernels.hip) and bf16x3
// (conv2_kernels.hip) convolution kernels.
#pragma once
#include "rvsr_common.h"

struct ConvFwdParams {
    TCat in;
    const float* w;
    const void* wpack;  // bf16x3 path: packed weights (conv2_kernels.hip)
    int swz;            // XCD-aware workgroup remap on/off
    int vec4;           // 16-byte-store epilogue usable (Wout % 4 == 0, aligned pointers)
    const float* bias;
    const float* res;
    float* out1;
    float* out2;
    int Co1;
    int B, Co, Hout, Wout;
    int w_mode;
    int prepacked;      // `workspace` already holds the packed weight image of this call (rvsr_conv2d_pack_weights / _batched)
    int fmt;            // 1: f16 + fp8 product format (w_mode | 4): forward 3x3 / stride-1 convs of the 64-row kernels only
    int act;
    float slope;
    int ps;
    int ntx;
};


// MODE 0: plain store, 1: + residual, 2: channel split into out1/out2, 3: pixel-shuffle(2) store.
// Branch-free per element except the final predicated store (the fully unrolled 16*MT*2 stores
// otherwise explode into thousands of basic blocks and spill the accumulators).
template <int MT, int MODE>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MT][2], const ConvFwdParams& p, int b, int o0, int row0,
                                              int col, int hi) {
    const bool has_bias = p.bias != nullptr;
    const float* bp = has_bias ? p.bias : p.w;  // p.w: any valid address, value discarded
    const float neg = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
    const bool col_ok = col < p.Wout;
    const size_t HW = (size_t)p.Hout * p.Wout;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = row0 + n;
        if (row >= p.Hout) continue;  // wave-uniform
        const size_t pix = (size_t)row * p.Wout + col;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + m * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                const bool ok = col_ok && o < p.Co;
                const int oc = ok ? o : 0;
                float v = acc[m][n][r];
                const float bb = bp[oc];
                v += has_bias ? bb : 0.f;
                v = v > 0.f ? v : v * neg;
                if (MODE == 3) {
                    const size_t idx = (((size_t)b * (p.Co >> 2) + (oc >> 2)) * (2 * p.Hout) + 2 * row + ((oc >> 1) & 1)) *
                                           (2 * p.Wout) + 2 * col + (oc & 1);
                    if (ok) p.out1[idx] = v;
                } else if (MODE == 2) {
                    const bool first = oc < p.Co1;
                    float* dst = first ? p.out1 : p.out2;
                    const size_t idx = ((size_t)b * (first ? p.Co1 : p.Co - p.Co1) + (first ? oc : oc - p.Co1)) * HW + pix;
                    if (ok) dst[idx] = v;
                } else {
                    const size_t idx = ((size_t)b * p.Co + oc) * HW + pix;
                    if (ok) {
                        if (MODE == 1) v += p.res[idx];
                        p.out1[idx] = v;
                    }
                }
            }
        }
    }
}


struct ConvWgradParams {
    TCat x;      // the conv's (virtual) input
    TView g;     // gradient w.r.t. the conv output: virtual (Co, Hout, Wout); g.act fuses act'
    float* part;   // [P][Co][Ctot][T]
    float* bpart;  // [P][Co] or nullptr
    int B, Co, Hout, Wout, ntx, nty, P;
    int ring;    // conv_wgrad2: walk the tiles column by column and keep the two shared X rows of vertical neighbours in LDS
};


// bf16x3 path (conv2_kernels.hip)
size_t rvsr_conv_fwd2_workspace_bytes(int ksize, int Co, int Ctot);
int rvsr_launch_conv_fwd2(ConvFwdParams p, int ksize, int stride, void* workspace, size_t workspace_bytes, hipStream_t st);

int rvsr_launch_conv_wgrad2(const ConvWgradParams& p, int gy, int gz, hipStream_t st);
int rvsr_launch_conv_wgrad1x1(const ConvWgradParams& p, int gy, int gz, hipStream_t st);
int rvsr_launch_conv_wgrad_s2(const ConvWgradParams& p, int gy, int gz, hipStream_t st);
// conv_thin_kernels.hip: 3x3 / stride-1 layers with <= 4 output channels on the vector ALU (exact f32)
int rvsr_conv_wgrad_thin_P(int B, int Hout, int Wout);
int rvsr_launch_conv_wgrad_thin(const ConvWgradParams& p, hipStream_t st);
bool rvsr_conv_fwd_thin_ok(const ConvFwdParams& p, int ksize, int stride);
int rvsr_launch_conv_fwd_thin(const ConvFwdParams& p, hipStream_t st);
