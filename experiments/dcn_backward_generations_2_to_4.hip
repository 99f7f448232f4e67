// Superseded DCN backward kernels (rounds 1-2: dcn_bwdin2 / dcn_bwdin3 / dcn_bwdin4 + the device-side `auto` dispatch, dcn_bwdw3), moved
// out of the library in round 5.  Not built.  They were cut out of realvsr_amd/csrc/dcn2_kernels.hip, which still holds what they
// shared (DcnBwdW2Params, the x tile helpers, dcn_bwdw2 for the exact-f32 mode, dcn_bwdw4); dcn_bwdin4.inc sits next to this file.

// ==========================================================================================
// Backward w.r.t. input / offsets / mask, second generation.
//
//   col_grad[(tap, c), px] = sum_o W[o, (tap, c)] * gOut[o, px]        (matrix cores, bf16x3)
//   grad_mask / grad_offset = reductions of col_grad * {bilinear(x), d bilinear/d(y,x)} over the group
//   grad_input             += scatter of col_grad * mask * corner weights
//
// The col_grad tile never leaves the accumulator registers: an M tile is ordered
// (2 taps) x (16 channels of the chunk), so in the D layout lane (px, half) owns, for each tap and
// octet, exactly channels 4*half .. 4*half+3 -- the same channel quad the x tile stores as one
// float4 per position.  Each lane therefore consumes its 16 accumulator values with 4 float4 corner
// fetches per (tap, octet), adds its partial sums to its partner lane's (lane ^ 32) and scatters
// into an LDS grad_input tile (ds_add_f32) that is flushed with one global atomic per touched cell.
// gOut (x act') is held as B fragments in registers for the whole tile (it is reused by all M tiles).
template <int NK>
__global__ void pack_weights_bwd_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int C, int nchunks) {
    // packed[chunk][tp (5)][part][ooct (2*NK)][row (32)][8 o];  row -> tap = 2*tp + (row >> 4), c = 16*chunk + (row & 15)
    const size_t total = (size_t)nchunks * 5 * (2 * NK) * 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx & 31);
        size_t r = idx >> 5;
        const int ooct = (int)(r % (2 * NK));
        r /= (2 * NK);
        const int tp = (int)(r % 5), chunk = (int)(r / 5);
        const int tap = 2 * tp + (row >> 4), c = 16 * chunk + (row & 15);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = 8 * ooct + j;
            v[j] = (tap < 9 && c < C && o < Co) ? w[((size_t)o * C + c) * 9 + tap] : 0.f;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const size_t blk = ((size_t)chunk * 5 + tp) * 2, per = (size_t)(2 * NK) * 32;
        packed[blk * per + ooct * 32 + row] = hi;
        packed[(blk + 1) * per + ooct * 32 + row] = lo;
    }
}

struct DcnBwdIn2Params {
    DcnGeom d;
    TView g;            // grad_output view (Co, Ho, Wo), optional fused act'
    float* gx;          // (B, C, H, W), zero on entry
    float* goff;
    float* gmask;
    size_t goff_bs, gmask_bs;
    int o_base, o_cnt;  // dcn_bwdin3 only: the pass covers output channels o_base .. o_base + o_cnt - 1 (o_cnt <= 64)
    int accum;          // dcn_bwdin3 only: a later pass of the same call: add to grad_offset / grad_mask instead of overwriting
    // Kernel selection on the device (rvsr_launch_dcn_bwdin_auto): every candidate is launched and returns at once unless
    // the sampled count of large offset components lies in its range.  nullptr: always run.
    const unsigned* probe;
    unsigned probe_lo, probe_hi;   // run when probe_lo <= *probe < probe_hi
};
__device__ __forceinline__ bool bwdin_not_selected(const DcnBwdIn2Params& p) {
    if (p.probe == nullptr) return false;
    const unsigned c = *p.probe;
    return c < p.probe_lo || c >= p.probe_hi;
}

// Sampled statistic behind that selection: every 16th row of every offset plane, count of components with |v| > 2.5 px
// (beyond what dcn_bwdin3/4's private windows cover on either side).  ~1/16 of the offset planes is read.
__global__ void dcn_offset_probe_kernel(const float* __restrict__ off, size_t off_bs, int B, int planes, int Ho, int Wo,
                                        unsigned* __restrict__ cnt) {
    const int nrow = (Ho + 15) / 16;
    const size_t total = (size_t)B * planes * nrow * Wo;
    unsigned mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo);
        size_t r = i / Wo;
        const int row = (int)(r % nrow);
        r /= nrow;
        const int pl = (int)(r % planes), b = (int)(r / planes);
        const int y = row * 16 + 8 < Ho ? row * 16 + 8 : Ho - 1;
        const float v = off[(size_t)b * off_bs + ((size_t)pl * Ho + y) * Wo + x];
        mine += fabsf(v) > 2.5f ? 1u : 0u;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mine += __shfl_xor(mine, s);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(cnt, mine);
}

// BR: halo of the shared tile in pixels (offsets beyond it go to global memory); 3 by default, 5 for offset-heavy inputs
template <int TH, int NK, int BR>
__global__ __launch_bounds__(TH * 64, 2) void dcn_bwdin2_kernel(const DcnBwdIn2Params p, const bf16x8* __restrict__ wpack) {
    constexpr int NT = TH * 64;
    constexpr int TR = TH + 2 * BR + 2, TC = 32 + 2 * BR + 2, NPOS = TR * TC;
    constexpr int WBLK = 2 * (2 * NK) * 32;  // vectors per (chunk, tap-pair) block (hi + lo)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);   // [4 quads][NPOS]   x tile of the chunk
    // grad_input accumulation tile, channel-PLANAR [16 ch][NPOS] so that the ds_add_f32 of consecutive pixels
    // hit consecutive banks (a float4-per-position layout is a 4-way conflict on every atomic)
    float* gt = reinterpret_cast<float*>(xt + 4 * NPOS);
    bf16x8* wsb = reinterpret_cast<bf16x8*>(gt + 16 * NPOS);  // [5][WBLK]
    if (bwdin_not_selected(p)) return;
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int tx = blockIdx.x % d.ntx, ty = blockIdx.x / d.ntx;
    const int x0 = tx * 32, y0 = ty * TH, b = blockIdx.z;
    const int ty0 = y0 * d.stride - d.pad - BR, tx0 = x0 * d.stride - d.pad - BR;
    const int nchunks = (d.C + 15) / 16;
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.Ho * d.Wo;
    const int oy = y0 + wave, ox = x0 + lo;
    const bool px_ok = oy < d.Ho && ox < d.Wo;
    const size_t pix = (size_t)oy * d.Wo + ox;

    // gOut (x act') as B fragments: lane (px, hi) holds o = 8*(2*ks + hi) .. +7 for ks < NK
    bf16x8 gh[NK], gl[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = 8 * (2 * ks + hi) + j;
            v[j] = (px_ok && o < d.Co) ? tview_get(p.g, b, o, oy, ox) : 0.f;
        }
        split8(v, gh[ks], gl[ks]);
    }

    for (int e = tid; e < 16 * NPOS; e += NT) gt[e] = 0.f;

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * 16;
        {
            const bf16x8* src = wpack + (size_t)chunk * 5 * WBLK;
#pragma unroll 4
            for (int e = tid; e < 5 * WBLK; e += NT) wsb[e] = src[e];
        }
        stage_x_tile<NT, 4, TR, TC>(xt, d, b, c0, ty0, tx0, tid);
        __syncthreads();

#pragma unroll 1
        for (int tp = 0; tp < 5; ++tp) {
            f32x16 acc = zero16();
            const bf16x8* wb_hi = wsb + tp * WBLK;
            const bf16x8* wb_lo = wb_hi + (2 * NK) * 32;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const bf16x8 ah = wb_hi[(2 * ks + hi) * 32 + lo], al = wb_lo[(2 * ks + hi) * 32 + lo];
                acc = mfma_bf16(ah, gh[ks], acc);
                acc = mfma_bf16(ah, gl[ks], acc);
                acc = mfma_bf16(al, gh[ks], acc);
            }
            // consume: regs 8*tsel + 4*oc + e  <->  tap 2*tp+tsel, octet oc of the chunk, channel 4*hi + e
#pragma unroll
            for (int tsel = 0; tsel < 2; ++tsel) {
                const int tap = 2 * tp + tsel;
                if (tap >= 9) continue;
#pragma unroll
                for (int oc = 0; oc < 2; ++oc) {
                    const int cb8 = c0 + 8 * oc;
                    if (cb8 >= d.C) continue;  // uniform
                    const int g = cb8 / d.cpg;
                    float gy_s = 0.f, gx_s = 0.f, gm_s = 0.f, m = 0.f;
                    if (px_ok) {
                        const float* offp = d.offset + (size_t)b * d.off_bs + (size_t)(g * 18 + 2 * tap) * hw + pix;
                        const float dy = offp[0], dx = offp[hw];
                        m = d.mask[(size_t)b * d.mask_bs + (size_t)(g * 9 + tap) * hw + pix];
                        if (d.mask_logit) m = 1.f / (1.f + __expf(-m));
                        const float y = (float)(oy * d.stride - d.pad + (tap / 3) * d.dil) + dy;
                        const float x = (float)(ox * d.stride - d.pad + (tap % 3) * d.dil) + dx;
                        if (y > -1.f && x > -1.f && y < (float)d.H && x < (float)d.W) {
                            const float fy = floorf(y), fx = floorf(x);
                            const int yi = (int)fy, xi = (int)fx;
                            const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
                            const bool vy0 = yi >= 0, vy1 = yi + 1 <= d.H - 1, vx0 = xi >= 0, vx1 = xi + 1 <= d.W - 1;
                            const float w00 = (vy0 && vx0) ? hy * hx : 0.f, w01 = (vy0 && vx1) ? hy * lx : 0.f;
                            const float w10 = (vy1 && vx0) ? ly * hx : 0.f, w11 = (vy1 && vx1) ? ly * lx : 0.f;
                            const int cy0 = vy0 ? yi : 0, cy1 = vy1 ? yi + 1 : d.H - 1;
                            const int cx0 = vx0 ? xi : 0, cx1 = vx1 ? xi + 1 : d.W - 1;
                            const int r0 = cy0 - ty0, r1 = cy1 - ty0, s0 = cx0 - tx0, s1 = cx1 - tx0;
                            const bool in_tile = r0 >= 0 && r1 < TR && s0 >= 0 && s1 < TC;
                            const int quad = 2 * oc + hi;
                            const int cq = cb8 + 4 * hi;  // this lane's first channel
                            float4 a00, a01, a10, a11;    // corner values of the lane's 4 channels (0 where corner invalid)
                            if (in_tile) {
                                const float4* xq = xt + quad * NPOS;
                                a00 = xq[r0 * TC + s0]; a01 = xq[r0 * TC + s1]; a10 = xq[r1 * TC + s0]; a11 = xq[r1 * TC + s1];
                            } else {
                                const float* pl = d.x + ((size_t)b * d.C + cq) * HW;
                                const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
                                float t00[4], t01[4], t10[4], t11[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const bool cok = cq + e < d.C;
                                    const float* q = pl + (size_t)e * HW;
                                    t00[e] = cok ? q[i00] : 0.f; t01[e] = cok ? q[i01] : 0.f;
                                    t10[e] = cok ? q[i10] : 0.f; t11[e] = cok ? q[i11] : 0.f;
                                }
                                a00 = make_float4(t00[0], t00[1], t00[2], t00[3]); a01 = make_float4(t01[0], t01[1], t01[2], t01[3]);
                                a10 = make_float4(t10[0], t10[1], t10[2], t10[3]); a11 = make_float4(t11[0], t11[1], t11[2], t11[3]);
                            }
                            const float z00 = (vy0 && vx0) ? 1.f : 0.f, z01 = (vy0 && vx1) ? 1.f : 0.f;
                            const float z10 = (vy1 && vx0) ? 1.f : 0.f, z11 = (vy1 && vx1) ? 1.f : 0.f;
                            const float c00[4] = {a00.x * z00, a00.y * z00, a00.z * z00, a00.w * z00};
                            const float c01[4] = {a01.x * z01, a01.y * z01, a01.z * z01, a01.w * z01};
                            const float c10[4] = {a10.x * z10, a10.y * z10, a10.z * z10, a10.w * z10};
                            const float c11[4] = {a11.x * z11, a11.y * z11, a11.z * z11, a11.w * z11};
                            float t[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float cgv = acc[8 * tsel + 4 * oc + e];
                                gm_s += cgv * (w00 * c00[e] + w01 * c01[e] + w10 * c10[e] + w11 * c11[e]);
                                t[e] = cgv * m;
                                gy_s += (hx * (c10[e] - c00[e]) + lx * (c11[e] - c01[e])) * t[e];
                                gx_s += (hy * (c01[e] - c00[e]) + ly * (c11[e] - c10[e])) * t[e];
                            }
                            if (in_tile) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float* gq = gt + (4 * quad + e) * NPOS;
                                    if (w00 != 0.f) __hip_atomic_fetch_add(gq + r0 * TC + s0, w00 * t[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (w01 != 0.f) __hip_atomic_fetch_add(gq + r0 * TC + s1, w01 * t[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (w10 != 0.f) __hip_atomic_fetch_add(gq + r1 * TC + s0, w10 * t[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (w11 != 0.f) __hip_atomic_fetch_add(gq + r1 * TC + s1, w11 * t[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            } else {
                                float* gp = p.gx + ((size_t)b * d.C + cq) * HW;
                                const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (cq + e >= d.C) continue;
                                    float* q = gp + (size_t)e * HW;
                                    if (w00 != 0.f) atomicAdd(q + i00, w00 * t[e]);
                                    if (w01 != 0.f) atomicAdd(q + i01, w01 * t[e]);
                                    if (w10 != 0.f) atomicAdd(q + i10, w10 * t[e]);
                                    if (w11 != 0.f) atomicAdd(q + i11, w11 * t[e]);
                                }
                            }
                        }
                    }
                    // add the partner lane's half of the octet (channels 4*(1-hi) ..)
                    gy_s += __shfl_xor(gy_s, 32);
                    gx_s += __shfl_xor(gx_s, 32);
                    gm_s += __shfl_xor(gm_s, 32);
                    if (px_ok && hi == 0) {
                        if (d.mask_logit) gm_s *= m * (1.f - m);
                        float* go = p.goff + (size_t)b * p.goff_bs + (size_t)(g * 18 + 2 * tap) * hw + pix;
                        float* gk = p.gmask + (size_t)b * p.gmask_bs + (size_t)(g * 9 + tap) * hw + pix;
                        if (cb8 % d.cpg == 0) {  // first octet of this deformable group: overwrite
                            go[0] = gy_s;
                            go[hw] = gx_s;
                            gk[0] = gm_s;
                        } else {                 // group wider than 8 channels: accumulate
                            go[0] += gy_s;
                            go[hw] += gx_s;
                            gk[0] += gm_s;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // flush the accumulation tile: one global atomic per touched cell (coalesced along W)
        for (int it = tid; it < 16 * NPOS; it += NT) {
            const float v = gt[it];
            if (v != 0.f) {
                gt[it] = 0.f;
                const int cc = it / NPOS, pos = it - cc * NPOS;
                const int yy = ty0 + pos / TC, xx = tx0 + pos % TC, c = c0 + cc;
                if (c < d.C && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W)
                    atomicAdd(p.gx + ((size_t)b * d.C + c) * HW + (size_t)yy * d.W + xx, v);
            }
        }
        __syncthreads();
    }
}

size_t rvsr_dcn_bwdin2_workspace_bytes(int Co, int C) {
    const int nk = Co <= 16 ? 1 : (Co <= 32 ? 2 : (Co <= 64 ? 4 : 8));
    return (size_t)((C + 15) / 16) * 5 * 2 * (2 * nk) * 32 * 16;
}

template <int NK, int BR>
static int launch_bwdin2(const DcnBwdIn2Params& p, const float* weight, void* workspace, hipStream_t st) {
    constexpr int TH = 8;
    constexpr int TR = TH + 2 * BR + 2, TC = 32 + 2 * BR + 2;
    const DcnGeom& d = p.d;
    const int nchunks = (d.C + 15) / 16;
    const size_t total = (size_t)nchunks * 5 * (2 * NK) * 32;
    hipLaunchKernelGGL(pack_weights_bwd_kernel<NK>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight,
                       (bf16x8*)workspace, d.Co, d.C, nchunks);
    const size_t lds = (size_t)16 * (8 * TR * TC + 5 * 2 * (2 * NK) * 32);
    auto k = dcn_bwdin2_kernel<TH, NK, BR>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_bwdin2: cannot reserve %zu B of LDS", lds);
    dim3 grid(d.ntx * ((d.Ho + TH - 1) / TH), 1, d.B);
    hipLaunchKernelGGL(k, grid, dim3(TH * 64), lds, st, p, (const bf16x8*)workspace);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_bwdin2 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

int rvsr_launch_dcn_bwdin2(const DcnGeom& d, const float* weight, const TView& g, float* gx, float* goff, size_t goff_bs,
                           float* gmask, size_t gmask_bs, void* workspace, size_t workspace_bytes, hipStream_t st,
                           const unsigned* probe, unsigned probe_lo, unsigned probe_hi, int halo) {
    if (d.cpg % 8 != 0 || d.Co > 128) return RVSR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < rvsr_dcn_bwdin2_workspace_bytes(d.Co, d.C)) return RVSR_ERR_UNSUPPORTED;
    DcnBwdIn2Params p;
    p.d = d; p.g = g; p.gx = gx; p.goff = goff; p.gmask = gmask; p.goff_bs = goff_bs; p.gmask_bs = gmask_bs;
    p.o_base = 0; p.o_cnt = d.Co; p.accum = 0;
    p.probe = probe; p.probe_lo = probe_lo; p.probe_hi = probe_hi;
    if (halo == 5 && d.Co <= 64) {   // wider halo (153 KB of LDS at NK = 4): pays off from a mean |offset| of ~4 px
        if (d.Co <= 16) return launch_bwdin2<1, 5>(p, weight, workspace, st);
        if (d.Co <= 32) return launch_bwdin2<2, 5>(p, weight, workspace, st);
        return launch_bwdin2<4, 5>(p, weight, workspace, st);
    }
    if (d.Co <= 16) return launch_bwdin2<1, D2_R>(p, weight, workspace, st);
    if (d.Co <= 32) return launch_bwdin2<2, D2_R>(p, weight, workspace, st);
    if (d.Co <= 64) return launch_bwdin2<4, D2_R>(p, weight, workspace, st);
    return launch_bwdin2<8, D2_R>(p, weight, workspace, st);
}



__global__ __launch_bounds__(512, 2) void dcn_bwdw3_kernel(const DcnBwdW2Params p) {
    // bf16x3 variant of dcn_bwdw2_kernel: both GEMM operands are kept PIXEL-contiguous as bf16 hi/lo ([row][128 px],
    // 272-byte row pitch), so a wave's 16 pixels are one k-step of v_mfma_f32_32x32x16_bf16 and the tile costs
    // 18 MFMAs per wave instead of 48 exact-f32 ones at twice the cycles each (6.6 K of ~18 K cycles per tile).
    constexpr int RP = 272, NT = 512;   // row pitch in bytes: 128 px x 2 B + 16 B pad (conflict-free 16-byte reads)
    constexpr int TR = 4 + 2 * D2_R + 2, TC = 32 + 2 * D2_R + 2, NPOS = TR * TC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);                      // [2 quads][NPOS]
    unsigned char* g_hi = reinterpret_cast<unsigned char*>(xt + 2 * NPOS);  // [64 o][RP]
    unsigned char* g_lo = g_hi + 64 * RP;
    unsigned char* c_hi = g_lo + 64 * RP;                                   // [96 n][RP]; row 72 = 1 (bias), 73.. = 0
    unsigned char* c_lo = c_hi + 96 * RP;
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int mb = blockIdx.y, c0 = blockIdx.z * 8;
    const bool m1_live = mb * 64 + 32 < d.Co;
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.Ho * d.Wo;
    const int g = c0 / d.cpg;

    for (int e = tid; e < (96 - DCN_KC) * DCN_NPX; e += NT) {
        const int j = e / DCN_NPX, px = e - j * DCN_NPX;
        reinterpret_cast<__bf16*>(c_hi + (DCN_KC + j) * RP)[px] = (__bf16)(j == 0 ? 1.f : 0.f);
        reinterpret_cast<__bf16*>(c_lo + (DCN_KC + j) * RP)[px] = (__bf16)0.f;
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = zero16();

    const int ntiles = d.B * p.nty * d.ntx;
    for (int tile = blockIdx.x; tile < ntiles; tile += p.P) {
        const bool st_ = tile == blockIdx.x + 2 * p.P;  // third tile of this workgroup
        if (st_) DSTAMP_W3(160);
        const int b = tile / (p.nty * d.ntx);
        const int trem = tile - b * (p.nty * d.ntx);
        const int ty = trem / d.ntx, tx = trem - ty * d.ntx;
        const int y0 = ty * 4, x0 = tx * 32;
        const int ty0 = y0 * d.stride - d.pad - D2_R, tx0 = x0 * d.stride - d.pad - D2_R;
        if (p.g.mode == 0 && (d.Wo & 3) == 0 && p.gvec) {  // (uniform)
            // 16-byte loads: item = (output channel, group of 4 pixels); 4 items per thread, all loads (value + act')
            // in flight together.  Dword loads made this phase load-instruction-bound (~9 K cycles per tile).
            float4 g4[4], a4[4];
            bool ok4[4];
            const float* ap = p.g.act != nullptr ? p.g.act : p.g.p;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int item = tid + i * NT;
                const int pg = item & 31, ol = item >> 5;
                const int o = mb * 64 + ol, yy = y0 + (pg >> 3), xx = x0 + 4 * (pg & 7);
                ok4[i] = o < d.Co && yy < d.Ho && xx < d.Wo;
                const size_t idx = ok4[i] ? (((size_t)b * d.Co + o) * d.Ho + yy) * d.Wo + xx : 0;
                g4[i] = *reinterpret_cast<const float4*>(p.g.p + idx);
                a4[i] = *reinterpret_cast<const float4*>(ap + idx);
            }
            const bool has_act = p.g.act != nullptr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int item = tid + i * NT;
                const int pg = item & 31, ol = item >> 5;
                const float f0 = has_act ? (a4[i].x > 0.f ? 1.f : p.g.slope) : 1.f, f1 = has_act ? (a4[i].y > 0.f ? 1.f : p.g.slope) : 1.f;
                const float f2 = has_act ? (a4[i].z > 0.f ? 1.f : p.g.slope) : 1.f, f3 = has_act ? (a4[i].w > 0.f ? 1.f : p.g.slope) : 1.f;
                const float v0 = ok4[i] ? g4[i].x * f0 : 0.f, v1 = ok4[i] ? g4[i].y * f1 : 0.f;
                const float v2 = ok4[i] ? g4[i].z * f2 : 0.f, v3 = ok4[i] ? g4[i].w * f3 : 0.f;
                typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                bf16x4_t h, l;
                h[0] = (__bf16)v0; h[1] = (__bf16)v1; h[2] = (__bf16)v2; h[3] = (__bf16)v3;
                l[0] = (__bf16)(v0 - (float)h[0]); l[1] = (__bf16)(v1 - (float)h[1]);
                l[2] = (__bf16)(v2 - (float)h[2]); l[3] = (__bf16)(v3 - (float)h[3]);
                *reinterpret_cast<bf16x4_t*>(g_hi + ol * RP + (4 * pg) * 2) = h;   // pixels 4pg .. 4pg+3 of row ol
                *reinterpret_cast<bf16x4_t*>(g_lo + ol * RP + (4 * pg) * 2) = l;
            }
        } else if (p.g.mode == 0) {  // (uniform)
            // thread t stages pixel (t & 127) for output channels 4*(t >> 7) + 16*i + 0..3, four at a time
            const int px = tid & 127, og = tid >> 7;
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int ol = 16 * i + 4 * og;
                float v4[4];
                tview_get_plain<4>(p.g, b, mb * 64 + ol, y0 + (px >> 5), x0 + (px & 31), v4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const __bf16 h = (__bf16)v4[j];
                    reinterpret_cast<__bf16*>(g_hi + (ol + j) * RP)[px] = h;
                    reinterpret_cast<__bf16*>(g_lo + (ol + j) * RP)[px] = (__bf16)(v4[j] - (float)h);
                }
            }
        } else {
#pragma unroll 2
            for (int e = tid; e < 64 * DCN_NPX; e += NT) {
                const int ol = e >> 7, px = e & 127;
                const int o = mb * 64 + ol;
                const float gv = o < d.Co ? tview_get(p.g, b, o, y0 + (px >> 5), x0 + (px & 31)) : 0.f;
                const __bf16 h = (__bf16)gv;
                reinterpret_cast<__bf16*>(g_hi + ol * RP)[px] = h;
                reinterpret_cast<__bf16*>(g_lo + ol * RP)[px] = (__bf16)(gv - (float)h);
            }
        }
        if (st_) DSTAMP_W3(161);
        stage_x_tile<NT, 2, TR, TC>(xt, d, b, c0, ty0, tx0, tid);
        if (st_) DSTAMP_W3(162);
        __syncthreads();
        if (st_) DSTAMP_W3(163);
        // column tile: item = (pixel, tap); 8 channels of the chunk share the sampling geometry.
        // (dy, dx, mask) of all of a thread's items are fetched first, unconditionally (clamped pixel).
        constexpr int NBI = (DCN_NPX * 9 + NT - 1) / NT;
        float b_dy[NBI], b_dx[NBI], b_m[NBI];
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int it_raw = tid + i * NT;
            const int it = it_raw < DCN_NPX * 9 ? it_raw : 0;
            const int px = it & 127, tap = it >> 7;
            const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
            const size_t pixc = (oy < d.Ho && ox < d.Wo) ? (size_t)oy * d.Wo + ox : 0;
            const float* offp = d.offset + (size_t)b * d.off_bs + (size_t)(g * 18 + 2 * tap) * hw + pixc;
            b_dy[i] = offp[0];
            b_dx[i] = offp[hw];
            b_m[i] = d.mask[(size_t)b * d.mask_bs + (size_t)(g * 9 + tap) * hw + pixc];
        }
        if (st_) DSTAMP_W3(164);
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int it = tid + i * NT;
            if (it >= DCN_NPX * 9) continue;
            const int px = it & 127, tap = it >> 7;
            const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (oy < d.Ho && ox < d.Wo) {
                const float dy = b_dy[i], dx = b_dx[i];
                float m = b_m[i];
                if (d.mask_logit) m = 1.f / (1.f + __expf(-m));
                const float y = (float)(oy * d.stride - d.pad + (tap / 3) * d.dil) + dy;
                const float x = (float)(ox * d.stride - d.pad + (tap % 3) * d.dil) + dx;
                if (y > -1.f && x > -1.f && y < (float)d.H && x < (float)d.W) {
                    const float fy = floorf(y), fx = floorf(x);
                    const int yi = (int)fy, xi = (int)fx;
                    const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
                    const bool vy0 = yi >= 0, vy1 = yi + 1 <= d.H - 1, vx0 = xi >= 0, vx1 = xi + 1 <= d.W - 1;
                    const float w00 = (vy0 && vx0) ? hy * hx : 0.f, w01 = (vy0 && vx1) ? hy * lx : 0.f;
                    const float w10 = (vy1 && vx0) ? ly * hx : 0.f, w11 = (vy1 && vx1) ? ly * lx : 0.f;
                    const int cy0 = vy0 ? yi : 0, cy1 = vy1 ? yi + 1 : d.H - 1;
                    const int cx0 = vx0 ? xi : 0, cx1 = vx1 ? xi + 1 : d.W - 1;
                    const int r0 = cy0 - ty0, r1 = cy1 - ty0, s0 = cx0 - tx0, s1 = cx1 - tx0;
                    if (r0 >= 0 && r1 < TR && s0 >= 0 && s1 < TC) {
                        const int p00 = r0 * TC + s0, p01 = r0 * TC + s1, p10 = r1 * TC + s0, p11 = r1 * TC + s1;
                        const float4 a00 = xt[p00], b00 = xt[NPOS + p00], a01 = xt[p01], b01 = xt[NPOS + p01];
                        const float4 a10 = xt[p10], b10 = xt[NPOS + p10], a11 = xt[p11], b11 = xt[NPOS + p11];
                        v[0] = w00 * a00.x + w01 * a01.x + w10 * a10.x + w11 * a11.x;
                        v[1] = w00 * a00.y + w01 * a01.y + w10 * a10.y + w11 * a11.y;
                        v[2] = w00 * a00.z + w01 * a01.z + w10 * a10.z + w11 * a11.z;
                        v[3] = w00 * a00.w + w01 * a01.w + w10 * a10.w + w11 * a11.w;
                        v[4] = w00 * b00.x + w01 * b01.x + w10 * b10.x + w11 * b11.x;
                        v[5] = w00 * b00.y + w01 * b01.y + w10 * b10.y + w11 * b11.y;
                        v[6] = w00 * b00.z + w01 * b01.z + w10 * b10.z + w11 * b11.z;
                        v[7] = w00 * b00.w + w01 * b01.w + w10 * b10.w + w11 * b11.w;
                    } else {
                        const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
                        const float* pl = d.x + ((size_t)b * d.C + c0) * HW;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (c0 + j < d.C) {
                                const float* q = pl + (size_t)j * HW;
                                v[j] = w00 * q[i00] + w01 * q[i01] + w10 * q[i10] + w11 * q[i11];
                            }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= m;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const __bf16 h = (__bf16)v[j];
                reinterpret_cast<__bf16*>(c_hi + (j * 9 + tap) * RP)[px] = h;
                reinterpret_cast<__bf16*>(c_lo + (j * 9 + tap) * RP)[px] = (__bf16)(v[j] - (float)h);
            }
        }
        if (st_) DSTAMP_W3(165);
        __syncthreads();
        if (st_) DSTAMP_W3(166);
        {   // wave w: pixels 16w .. 16w+15 = one k-step; lane (row lo, k-octet hi) reads pixels 16w + 8hi .. +7
            const int koff = (wave * 16 + 8 * hi) * 2;
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(g_hi + (m * 32 + lo) * RP + koff);
                al[m] = *reinterpret_cast<const bf16x8*>(g_lo + (m * 32 + lo) * RP + koff);
            }
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(c_hi + (n * 32 + lo) * RP + koff);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(c_lo + (n * 32 + lo) * RP + koff);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (m == 1 && !m1_live) continue;
                    acc[m][n] = mfma_bf16(ah[m], bh, acc[m][n]);
                    acc[m][n] = mfma_bf16(ah[m], bl, acc[m][n]);
                    acc[m][n] = mfma_bf16(al[m], bh, acc[m][n]);
                }
            }
        }
        if (st_) DSTAMP_W3(167);
        __syncthreads();
        if (st_) DSTAMP_W3(168);
    }

    const int q = blockIdx.x * 8 + wave;
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

// ==========================================================================================
// Backward w.r.t. input / offsets / mask, third generation: no floating-point atomics on the
// scatter.  MI355X measurements that drove this (profiles/r01_notes.md): ds_add_f32 costs ~117
// LDS cycles per wave-instruction (LDS_IDX_ACTIVE 58x the forward kernel's), global
// atomicAdd(float) tops out near 70 G atomics/s -- 2304 of them per pixel is the whole budget.
//
//   * K chunk = 8 channels (one k-octet); M tile = 4 taps x 8 channels, so in the D layout lane
//     (px, half) owns channels 4*half..4*half+3 of 4 taps = one float4 per corner.
//   * every wave scatters into its OWN private window of the grad_input tile (rows oy-1-R ..
//     oy+2+R), so waves never touch the same LDS cell;
//   * inside a wave, lanes that hit the same cell in the same instruction are serialised by a
//     claim / read-back round (write lane id, read it back, winners do a plain float4
//     read-modify-write, losers retry) -- one round in the common collision-free case;
//   * after a chunk, owner threads sum the <= 8 overlapping private windows per cell and issue one
//     global atomic per touched cell (different workgroups' halos overlap).
// Geometry: stride 1, dilation 1 (what EDVR/TDAN use); anything else takes the v2 kernel.
#define D3_R 2
#define D3_PR (2 * D3_R + 4)        // private window rows
#define D3_TC 40                    // tile columns: x0-pad-R .. (32 + 2R + 3 = 39 needed)
#define D3_TH 8

template <int NK>
__global__ void pack_weights_bwd3_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int o_base, int o_cnt, int C,
                                         int nchunks) {
    // packed[chunk][mt (3)][part][ooct (2*NK)][row (32)][8 o];  row -> tap = 4*mt + (row >> 3), c = 8*chunk + (row & 7)
    const size_t total = (size_t)nchunks * 3 * (2 * NK) * 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx & 31);
        size_t r = idx >> 5;
        const int ooct = (int)(r % (2 * NK));
        r /= (2 * NK);
        const int mt = (int)(r % 3), chunk = (int)(r / 3);
        const int tap = 4 * mt + (row >> 3), c = 8 * chunk + (row & 7);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ol = 8 * ooct + j;
            v[j] = (tap < 9 && c < C && ol < o_cnt) ? w[((size_t)(o_base + ol) * C + c) * 9 + tap] : 0.f;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const size_t blk = ((size_t)chunk * 3 + mt) * 2, per = (size_t)(2 * NK) * 32;
        packed[blk * per + ooct * 32 + row] = hi;
        packed[(blk + 1) * per + ooct * 32 + row] = lo;
    }
}

// Add c[k] to priv4[idx[k]] for the four bilinear corners of one sample, with intra-wave collision handling:
// every pending (lane, corner) writes its id to the cell's claim word, reads it back, and the winners do a plain
// 128-bit read-modify-write; losers (another lane or corner hit the same cell in this round) retry.  One LDS round
// trip per round for all four corners; one round in the common collision-free case.  LDS operations of a wave are
// executed in program order, so the read-back sees the last claim written in this round.
__device__ __forceinline__ void claim_add4(volatile int* claim, float4* priv4, const int (&idx)[4], const float4 (&c)[4],
                                           bool p0, bool p1, bool p2, bool p3, int lane) {
    bool pend[4] = {p0, p1, p2, p3};
    while (__any(pend[0] || pend[1] || pend[2] || pend[3])) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (pend[k]) claim[idx[k]] = lane * 4 + k;
        asm volatile("" ::: "memory");
        int got[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) got[k] = pend[k] ? claim[idx[k]] : -1;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (got[k] == lane * 4 + k) {
                float4 v = priv4[idx[k]];
                v.x += c[k].x; v.y += c[k].y; v.z += c[k].z; v.w += c[k].w;
                priv4[idx[k]] = v;
                pend[k] = false;
            }
        }
        asm volatile("" ::: "memory");
    }
}

template <int NK>
__global__ __launch_bounds__(D3_TH * 64, 2) void dcn_bwdin3_kernel(const DcnBwdIn2Params p, const bf16x8* __restrict__ wpack) {
    constexpr int NT = D3_TH * 64, TC = D3_TC, PR = D3_PR;
    constexpr int TR = D3_TH + PR - 1;             // shared rows: y0-pad-R .. (union of the private windows)
    constexpr int NPOS = TR * TC, PPOS = PR * TC;  // positions of the shared x tile / of one private window
    constexpr int WBLK = 2 * (2 * NK) * 32;        // vectors per (chunk, M tile) weight block (hi + lo)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);              // [2 quads][NPOS]
    float4* priv = xt + 2 * NPOS;                                  // [8 waves][2 quads][PPOS]
    int* claim = reinterpret_cast<int*>(priv + D3_TH * 2 * PPOS);  // [8 waves][2 quads][PPOS]
    bf16x8* wsb = reinterpret_cast<bf16x8*>(claim + D3_TH * 2 * PPOS);  // [3][WBLK]
    if (bwdin_not_selected(p)) return;
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    unsigned sbx, sby, sbz;
    swizzled_block(sbx, sby, sbz, d.swz);
    const int tx = sbx % d.ntx, ty = sbx / d.ntx;
    const int x0 = tx * 32, y0 = ty * D3_TH, b = sbz;
    const int ty0 = y0 - d.pad - D3_R, tx0 = x0 - d.pad - D3_R;  // stride 1
    const int nchunks = (d.C + 7) / 8;
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.Ho * d.Wo;
    const int oy = y0 + wave, ox = x0 + lo;
    const bool px_ok = oy < d.Ho && ox < d.Wo;
    const size_t pix = (size_t)oy * d.Wo + ox;
    const float4* xq = xt + hi * NPOS;                       // this lane's channel quad in the shared x tile
    float4* myp = priv + (wave * 2 + hi) * PPOS;
    volatile int* myc = claim + (wave * 2 + hi) * PPOS;

    DSTAMP(100);
    bf16x8 gh[NK], gl[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        float v[8];
        const int ol0 = 8 * (2 * ks + hi);  // first channel of this lane's octet inside the pass
        if (p.g.mode == 0) {  // (uniform)
            tview_get_plain<8>(p.g, b, p.o_base + ol0, oy, ox, v);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tview_get(p.g, b, p.o_base + ol0 + j, oy, ox);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ol0 + j < p.o_cnt ? v[j] : 0.f;
        split8(v, gh[ks], gl[ks]);
    }
    DSTAMP(101);
    for (int e = tid; e < D3_TH * 2 * PPOS; e += NT) priv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    DSTAMP(102);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * 8;
        const int g = c0 / d.cpg;
        // the 9 taps' (dy, dx, mask) of this lane's pixel for the chunk's deformable group: 27 unconditional loads
        // issued before the staging below, so their latency is paid once per chunk instead of once per tap
        float o_dy[9], o_dx[9], o_m[9];
        {
            const size_t pixc = px_ok ? pix : 0;
            const float* offp = d.offset + (size_t)b * d.off_bs + (size_t)(g * 18) * hw + pixc;
            const float* mskp = d.mask + (size_t)b * d.mask_bs + (size_t)(g * 9) * hw + pixc;
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                o_dy[t9] = offp[(size_t)(2 * t9) * hw];
                o_dx[t9] = offp[(size_t)(2 * t9 + 1) * hw];
                o_m[t9] = mskp[(size_t)t9 * hw];
            }
        }
        {
            const bf16x8* src = wpack + (size_t)chunk * 3 * WBLK;
#pragma unroll 3
            for (int e = tid; e < 3 * WBLK; e += NT) wsb[e] = src[e];
        }
        if (chunk < 3) DSTAMP(103 + 5 * chunk);
        stage_x_tile<NT, 2, TR, TC>(xt, d, b, c0, ty0, tx0, tid);
        if (chunk < 3) DSTAMP(104 + 5 * chunk);
        __syncthreads();
        if (chunk < 3) DSTAMP(105 + 5 * chunk);

        const int cq = c0 + 4 * hi;  // this lane's first channel
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            f32x16 acc = zero16();
            const bf16x8* wb_hi = wsb + mt * WBLK;
            const bf16x8* wb_lo = wb_hi + (2 * NK) * 32;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const bf16x8 ah = wb_hi[(2 * ks + hi) * 32 + lo], al = wb_lo[(2 * ks + hi) * 32 + lo];
                acc = mfma_bf16(ah, gh[ks], acc);
                acc = mfma_bf16(ah, gl[ks], acc);
                acc = mfma_bf16(al, gh[ks], acc);
            }
#pragma unroll
            for (int tsel = 0; tsel < 4; ++tsel) {
                const int tap = 4 * mt + tsel;
                if (tap >= 9) continue;  // uniform
                if (chunk == 1 && tap < 4) DSTAMP(140 + 4 * tap);
                float gy_s = 0.f, gx_s = 0.f, gm_s = 0.f, m = 0.f;
                bool inside = false, in_win = false;
                float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
                int r0 = 0, r1 = 0, s0 = 0, s1 = 0, cy0 = 0, cy1 = 0, cx0 = 0, cx1 = 0;
                float t[4] = {0.f, 0.f, 0.f, 0.f};
                if (px_ok) {
                    const float dy = o_dy[tap], dx = o_dx[tap];
                    m = o_m[tap];
                    if (d.mask_logit) m = 1.f / (1.f + __expf(-m));
                    const float y = (float)(oy - d.pad + tap / 3) + dy;
                    const float x = (float)(ox - d.pad + tap % 3) + dx;
                    inside = y > -1.f && x > -1.f && y < (float)d.H && x < (float)d.W;
                    if (inside) {
                        const float fy = floorf(y), fx = floorf(x);
                        const int yi = (int)fy, xi = (int)fx;
                        const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
                        const bool vy0 = yi >= 0, vy1 = yi + 1 <= d.H - 1, vx0 = xi >= 0, vx1 = xi + 1 <= d.W - 1;
                        w00 = (vy0 && vx0) ? hy * hx : 0.f; w01 = (vy0 && vx1) ? hy * lx : 0.f;
                        w10 = (vy1 && vx0) ? ly * hx : 0.f; w11 = (vy1 && vx1) ? ly * lx : 0.f;
                        cy0 = vy0 ? yi : 0; cy1 = vy1 ? yi + 1 : d.H - 1;
                        cx0 = vx0 ? xi : 0; cx1 = vx1 ? xi + 1 : d.W - 1;
                        r0 = cy0 - ty0; r1 = cy1 - ty0; s0 = cx0 - tx0; s1 = cx1 - tx0;  // shared-tile coords
                        // this wave's private window = shared rows wave .. wave+PR-1
                        in_win = r0 >= wave && r1 < wave + PR && s0 >= 0 && s1 < TC;
                        float4 a00, a01, a10, a11;
                        if (in_win) {
                            a00 = xq[r0 * TC + s0]; a01 = xq[r0 * TC + s1]; a10 = xq[r1 * TC + s0]; a11 = xq[r1 * TC + s1];
                        } else {
                            const float* pl = d.x + ((size_t)b * d.C + cq) * HW;
                            const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
                            float u00[4], u01[4], u10[4], u11[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool cok = cq + e < d.C;
                                const float* q = pl + (size_t)e * HW;
                                u00[e] = cok ? q[i00] : 0.f; u01[e] = cok ? q[i01] : 0.f;
                                u10[e] = cok ? q[i10] : 0.f; u11[e] = cok ? q[i11] : 0.f;
                            }
                            a00 = make_float4(u00[0], u00[1], u00[2], u00[3]); a01 = make_float4(u01[0], u01[1], u01[2], u01[3]);
                            a10 = make_float4(u10[0], u10[1], u10[2], u10[3]); a11 = make_float4(u11[0], u11[1], u11[2], u11[3]);
                        }
                        const float z00 = (vy0 && vx0) ? 1.f : 0.f, z01 = (vy0 && vx1) ? 1.f : 0.f;
                        const float z10 = (vy1 && vx0) ? 1.f : 0.f, z11 = (vy1 && vx1) ? 1.f : 0.f;
                        const float c00[4] = {a00.x * z00, a00.y * z00, a00.z * z00, a00.w * z00};
                        const float c01[4] = {a01.x * z01, a01.y * z01, a01.z * z01, a01.w * z01};
                        const float c10[4] = {a10.x * z10, a10.y * z10, a10.z * z10, a10.w * z10};
                        const float c11[4] = {a11.x * z11, a11.y * z11, a11.z * z11, a11.w * z11};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float cgv = acc[4 * tsel + e];
                            gm_s += cgv * (w00 * c00[e] + w01 * c01[e] + w10 * c10[e] + w11 * c11[e]);
                            t[e] = cgv * m;
                            gy_s += (hx * (c10[e] - c00[e]) + lx * (c11[e] - c01[e])) * t[e];
                            gx_s += (hy * (c01[e] - c00[e]) + ly * (c11[e] - c10[e])) * t[e];
                        }
                    }
                }
#ifdef RVSR_TIMELINE_DCN
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (chunk == 1 && tap < 4) DSTAMP(141 + 4 * tap);
#endif
                // ---- scatter (all lanes take part in the claim rounds; `pend` carries the per-lane predicate)
                {
                    const int pr0 = r0 - wave, pr1 = r1 - wave;
                    const bool go = inside && in_win;
                    const int cidx[4] = {go ? pr0 * TC + s0 : 0, go ? pr0 * TC + s1 : 0, go ? pr1 * TC + s0 : 0, go ? pr1 * TC + s1 : 0};
                    const float4 cval[4] = {make_float4(w00 * t[0], w00 * t[1], w00 * t[2], w00 * t[3]),
                                            make_float4(w01 * t[0], w01 * t[1], w01 * t[2], w01 * t[3]),
                                            make_float4(w10 * t[0], w10 * t[1], w10 * t[2], w10 * t[3]),
                                            make_float4(w11 * t[0], w11 * t[1], w11 * t[2], w11 * t[3])};
                    const bool p0 = go && w00 != 0.f, p1 = go && w01 != 0.f, p2 = go && w10 != 0.f, p3 = go && w11 != 0.f;
                    // Fast path without claim traffic.  Pixels of equal parity are two columns apart, so for offsets that
                    // differ by less than a pixel between neighbours (floor flips of at most 1) their left sample
                    // column s0 is strictly increasing; the wave checks exactly that (key[l] > key[l-2]).  Then, within
                    // one parity class, no two lanes share a column: the "left" corners (00, 10) hit pairwise
                    // different cells and so do the "right" ones (01, 11; s1 = s0 + 1 whenever the corner is valid,
                    // invalid corners have weight 0 and are skipped).  The four (parity, side) groups are applied one
                    // after the other as plain read-modify-writes (LDS operations of a wave execute in order).
                    // Anything else takes the exact claim rounds.  Lanes that do not scatter stand in with the
                    // zero-offset column so they do not break the test.
                    const int key = go ? s0 : lo + D3_R + tap % 3;
                    const int left2 = __shfl_up(key, 2);
                    if (__all(lo < 2 || key > left2)) {
#pragma unroll
                        for (int par = 0; par < 2; ++par) {  // even pixels, then odd pixels
                            const bool mine = (lo & 1) == par;
                            float4 v0 = myp[cidx[0]], v2 = myp[cidx[2]];
                            v0.x += cval[0].x; v0.y += cval[0].y; v0.z += cval[0].z; v0.w += cval[0].w;
                            v2.x += cval[2].x; v2.y += cval[2].y; v2.z += cval[2].z; v2.w += cval[2].w;
                            if (mine && p0) myp[cidx[0]] = v0;
                            if (mine && p2) myp[cidx[2]] = v2;
                            asm volatile("" ::: "memory");
                            float4 v1 = myp[cidx[1]], v3 = myp[cidx[3]];
                            v1.x += cval[1].x; v1.y += cval[1].y; v1.z += cval[1].z; v1.w += cval[1].w;
                            v3.x += cval[3].x; v3.y += cval[3].y; v3.z += cval[3].z; v3.w += cval[3].w;
                            if (mine && p1) myp[cidx[1]] = v1;
                            if (mine && p3) myp[cidx[3]] = v3;
                            asm volatile("" ::: "memory");
                        }
                    } else {
                        claim_add4(myc, myp, cidx, cval, p0, p1, p2, p3, lane);
                    }
                    if (chunk == 1 && tap < 4) DSTAMP(142 + 4 * tap);
                    if (inside && !in_win) {  // large offset: straight to global memory
                        float* gp = p.gx + ((size_t)b * d.C + cq) * HW;
                        const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (cq + e >= d.C) continue;
                            float* q = gp + (size_t)e * HW;
                            if (w00 != 0.f) atomicAdd(q + i00, w00 * t[e]);
                            if (w01 != 0.f) atomicAdd(q + i01, w01 * t[e]);
                            if (w10 != 0.f) atomicAdd(q + i10, w10 * t[e]);
                            if (w11 != 0.f) atomicAdd(q + i11, w11 * t[e]);
                        }
                    }
                }
                if (chunk == 1 && tap < 4) DSTAMP(143 + 4 * tap);
                gy_s += __shfl_xor(gy_s, 32);
                gx_s += __shfl_xor(gx_s, 32);
                gm_s += __shfl_xor(gm_s, 32);
                if (px_ok && hi == 0) {
                    if (d.mask_logit) gm_s *= m * (1.f - m);
                    float* go_ = p.goff + (size_t)b * p.goff_bs + (size_t)(g * 18 + 2 * tap) * hw + pix;
                    float* gk = p.gmask + (size_t)b * p.gmask_bs + (size_t)(g * 9 + tap) * hw + pix;
                    if (c0 % d.cpg == 0 && !p.accum) {
                        go_[0] = gy_s;
                        go_[hw] = gx_s;
                        gk[0] = gm_s;
                    } else {
                        go_[0] += gy_s;
                        go_[hw] += gx_s;
                        gk[0] += gm_s;
                    }
                }
            }
        }
        if (chunk < 3) DSTAMP(106 + 5 * chunk);
        __syncthreads();
        // ---- merge the private windows: owner thread per (quad, shared row, col)
        for (int it = tid; it < 2 * NPOS; it += NT) {
            const int quad = it / NPOS, pos = it - quad * NPOS;
            const int r = pos / TC, s = pos - r * TC;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            const int w_lo = r - (PR - 1) > 0 ? r - (PR - 1) : 0, w_hi = r < D3_TH - 1 ? r : D3_TH - 1;
            for (int w = w_lo; w <= w_hi; ++w) {
                float4* cell = priv + (w * 2 + quad) * PPOS + (r - w) * TC + s;
                const float4 v = *cell;
                if (v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f) {
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    *cell = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            const int yy = ty0 + r, xx = tx0 + s, cb = c0 + 4 * quad;
            if (yy >= 0 && yy < d.H && xx >= 0 && xx < d.W) {
                float* q = p.gx + ((size_t)b * d.C + cb) * HW + (size_t)yy * d.W + xx;
                if (cb < d.C && sum.x != 0.f) atomicAdd(q, sum.x);
                if (cb + 1 < d.C && sum.y != 0.f) atomicAdd(q + HW, sum.y);
                if (cb + 2 < d.C && sum.z != 0.f) atomicAdd(q + 2 * HW, sum.z);
                if (cb + 3 < d.C && sum.w != 0.f) atomicAdd(q + 3 * HW, sum.w);
            }
        }
        if (chunk < 3) DSTAMP(107 + 5 * chunk);
        __syncthreads();
    }
    DSTAMP(130);
}

#include "dcn_bwdin4.inc"

static int nk_of(int Co) { return Co <= 16 ? 1 : (Co <= 32 ? 2 : 4); }  // per pass of <= 64 output channels
size_t rvsr_dcn_bwdin3_workspace_bytes(int Co, int C) { return (size_t)((C + 7) / 8) * 3 * 2 * (2 * nk_of(Co)) * 32 * 16; }

template <int NK>
static int launch_bwdin3(const DcnBwdIn2Params& p, const float* weight, void* workspace, hipStream_t st) {
    const DcnGeom& d = p.d;
    const int nchunks = (d.C + 7) / 8;
    const size_t total = (size_t)nchunks * 3 * (2 * NK) * 32;
    hipLaunchKernelGGL(pack_weights_bwd3_kernel<NK>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight,
                       (bf16x8*)workspace, p.o_base, p.o_cnt, d.C, nchunks);
    constexpr int TR = D3_TH + D3_PR - 1, PPOS = D3_PR * D3_TC;
    const size_t lds = (size_t)16 * (2 * TR * D3_TC + D3_TH * 2 * PPOS + 3 * 2 * (2 * NK) * 32) + (size_t)4 * D3_TH * 2 * PPOS;
    static const int gen = [] { const char* e = getenv("RVSR_DCN_BWD"); return e ? atoi(e) : 4; }();  // developer A/B switch
    // dcn_bwdin4 addresses one batch element's planes with 32-bit byte offsets (raw buffers): larger frames take dcn_bwdin3
    const size_t span = sizeof(float) * (size_t)d.Ho * d.Wo * (size_t)((d.C / d.cpg) * 18 > d.C ? (d.C / d.cpg) * 18 : d.C);
    auto k = gen >= 4 && span < ((size_t)1 << 32) ? dcn_bwdin4_kernel<NK> : dcn_bwdin3_kernel<NK>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_bwdin3: cannot reserve %zu B of LDS", lds);
    dim3 grid(d.ntx * ((d.Ho + D3_TH - 1) / D3_TH), 1, d.B);
    hipLaunchKernelGGL(k, grid, dim3(D3_TH * 64), lds, st, p, (const bf16x8*)workspace);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_bwdin3 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

int rvsr_launch_dcn_bwdin3(const DcnGeom& d, const float* weight, const TView& g, float* gx, float* goff, size_t goff_bs,
                           float* gmask, size_t gmask_bs, void* workspace, size_t workspace_bytes, hipStream_t st,
                           const unsigned* probe, unsigned probe_lo, unsigned probe_hi) {
    // More than 64 output channels would need a 49 KB weight block on top of the private windows (> 160 KB of LDS) and
    // 64 more registers of gOut fragments: they are handled as passes of <= 64 output channels.  Everything downstream
    // of col_grad = W^T gOut is linear in it, so the passes simply add up (grad_input through the atomics it uses anyway,
    // grad_offset / grad_mask with `accum`); the sampling work is repeated per pass (nf = 128: 2 passes, still ~2.5x
    // faster than the LDS-atomic dcn_bwdin2 path).
    if (d.cpg % 8 != 0 || d.stride != 1 || d.dil != 1) return RVSR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < rvsr_dcn_bwdin3_workspace_bytes(d.Co, d.C)) return RVSR_ERR_UNSUPPORTED;
    DcnBwdIn2Params p;
    p.d = d; p.g = g; p.gx = gx; p.goff = goff; p.gmask = gmask; p.goff_bs = goff_bs; p.gmask_bs = gmask_bs;
    p.probe = probe; p.probe_lo = probe_lo; p.probe_hi = probe_hi;
    for (int ob = 0; ob < d.Co; ob += 64) {
        p.o_base = ob;
        p.o_cnt = d.Co - ob < 64 ? d.Co - ob : 64;
        p.accum = ob > 0;
        int rc;
        switch (nk_of(p.o_cnt)) {
            case 1: rc = launch_bwdin3<1>(p, weight, workspace, st); break;
            case 2: rc = launch_bwdin3<2>(p, weight, workspace, st); break;
            default: rc = launch_bwdin3<4>(p, weight, workspace, st); break;
        }
        if (rc != RVSR_OK) return rc;
    }
    return RVSR_OK;
}

// ------------------------------------------------------------------------------------------
// Offset-aware choice between the two generations, without a host round trip.  dcn_bwdin3/4 scatter through per-wave private
// windows that cover offsets of about +-2 px; beyond them every sample costs 16 global atomics and 16 gathers, and the step time
// at a mean |offset| of 2 / 3 / 5 px was 181 / 240 / 327 ms against 117 at ~0 px.  dcn_bwdin2 (shared LDS tile, ds_add_f32,
// +-3 px) is 2.2x slower at small offsets but flat up to ~3 px: 181 / 180 / 236 ms at the same three points.  A sampled
// statistic of the offsets decides on the device; both kernels are enqueued and the one not selected returns immediately
// (~25 us per call for memset + probe + the empty launches).  From a mean |offset| of ~4 px dcn_bwdin2 runs with a 5 px halo
// (153 KB of LDS): 5 px 234 -> 198 ms, 8 px 345 -> 308; at 2-3 px the 3 px halo is 1.5 % faster (smaller tile to stage and flush).
size_t rvsr_dcn_bwdin_auto_workspace_bytes(int Co, int C) {
    const size_t a3 = (rvsr_dcn_bwdin3_workspace_bytes(Co, C) + 255) & ~(size_t)255;
    const size_t a2 = (rvsr_dcn_bwdin2_workspace_bytes(Co, C) + 255) & ~(size_t)255;
    return a3 + a2 + 256;
}
int rvsr_launch_dcn_bwdin_auto(const DcnGeom& d, const float* weight, const TView& g, float* gx, float* goff, size_t goff_bs,
                               float* gmask, size_t gmask_bs, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (!workspace || workspace_bytes < rvsr_dcn_bwdin_auto_workspace_bytes(d.Co, d.C)) return RVSR_ERR_UNSUPPORTED;
    if (d.cpg % 8 != 0 || d.stride != 1 || d.dil != 1 || d.Co > 128) return RVSR_ERR_UNSUPPORTED;
    const size_t a3 = (rvsr_dcn_bwdin3_workspace_bytes(d.Co, d.C) + 255) & ~(size_t)255;
    const size_t a2 = (rvsr_dcn_bwdin2_workspace_bytes(d.Co, d.C) + 255) & ~(size_t)255;
    unsigned char* ws = (unsigned char*)workspace;
    unsigned* cnt = (unsigned*)(ws + a3 + a2);
    if (hipMemsetAsync(cnt, 0, sizeof(unsigned), st) != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn backward: memset of the probe counter failed");
    const int planes = (d.C / d.cpg) * 18, nrow = (d.Ho + 15) / 16;
    const size_t total = (size_t)d.B * planes * nrow * d.Wo;
    // P(|v| > 2.5 px) = 0.25 <=> Gaussian offsets of std 2.2 px (mean |v| 1.75 px); 0.55 <=> std 4.2 px (mean 3.3 px)
    const unsigned thr = (unsigned)(total / 4), thr2 = d.Co <= 64 ? (unsigned)(total * 11 / 20) : 0xffffffffu;
    const unsigned nb = (unsigned)((total + 2047) / 2048 < 2048 ? (total + 2047) / 2048 : 2048);
    hipLaunchKernelGGL(dcn_offset_probe_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, d.offset, d.off_bs, d.B, planes, d.Ho, d.Wo, cnt);
    int rc = rvsr_launch_dcn_bwdin3(d, weight, g, gx, goff, goff_bs, gmask, gmask_bs, ws, a3, st, cnt, 0, thr);
    if (rc != RVSR_OK) return rc;
    rc = rvsr_launch_dcn_bwdin2(d, weight, g, gx, goff, goff_bs, gmask, gmask_bs, ws + a3, a2, st, cnt, thr, thr2, 3);
    if (rc != RVSR_OK || thr2 == 0xffffffffu) return rc;
    return rvsr_launch_dcn_bwdin2(d, weight, g, gx, goff, goff_bs, gmask, gmask_bs, ws + a3, a2, st, cnt, thr2, 0xffffffffu, 5);
}
