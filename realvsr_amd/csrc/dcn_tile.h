// dcn_tile.h -- the LDS x tile shared by the second/third-generation DCN kernels (dcn2_kernels.hip, dcn3_kernels.hip).
#pragma once
#include "bf16x3.h"
#include "dcn_common.h"

#define D2_R 3  // halo radius (pixels) of the LDS x tile beyond the 3x3 footprint

// Stage the x tile of NQ channel quads (channels c0 .. c0+4*NQ-1) into LDS as [quad][row][col] float4.
// All global loads are unconditional (addresses clamped into the tensor, validity applied afterwards) and are
// issued in batches before any LDS write: a load inside a divergent `if` is waited for at the join, which turns
// the staging loop into one HBM/L2 round trip per item.
template <int NT, int NQ, int TR, int TC>
__device__ __forceinline__ void stage_x_tile(float4* xt, const DcnGeom& d, int b, int c0, int ty0, int tx0, int tid) {
    constexpr int NPOS = TR * TC, NITEMS = NQ * NPOS, PER = (NITEMS + NT - 1) / NT, BATCH = PER < 6 ? PER : 5;
    const size_t HW = (size_t)d.H * d.W;
#pragma unroll
    for (int base = 0; base < PER; base += BATCH) {
        float v[BATCH][4];
        int nv[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int it_raw = tid + (base + k) * NT;
            const bool live = (base + k < PER) && it_raw < NITEMS;
            const int it = live ? it_raw : 0;
            const int quad = it / NPOS, pos = it - quad * NPOS;
            const int r = pos / TC, s = pos - r * TC;
            const int gy = ty0 + r, gx = tx0 + s, cb = c0 + 4 * quad;
            const bool inb = live && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W && cb < d.C;
            const int gyc = gy < 0 ? 0 : (gy >= d.H ? d.H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= d.W ? d.W - 1 : gx);
            const int cbc = inb ? cb : 0;
            const float* src = d.x + ((size_t)b * d.C) * HW + (size_t)gyc * d.W + gxc;
            nv[k] = inb ? (d.C - cb < 4 ? d.C - cb : 4) : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = cbc + e < d.C ? cbc + e : d.C - 1;
                v[k][e] = src[(size_t)c * HW];
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int it = tid + (base + k) * NT;
            if ((base + k < PER) && it < NITEMS)
                xt[it] = make_float4(nv[k] > 0 ? v[k][0] : 0.f, nv[k] > 1 ? v[k][1] : 0.f, nv[k] > 2 ? v[k][2] : 0.f,
                                     nv[k] > 3 ? v[k][3] : 0.f);
        }
    }
}

