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

// Energy ablations of scratch builds (tools/build_variant.sh conv2_kernels <name> -DRVSR_ABL5=<bits>; results wrong by construction; the 8 x 64
// kernel runs at the package power cap, so its time follows the energy of what it does -- profiles/r05_notes.md): 1 = LDS fragment reads of
// taps 2-8 dropped, 2 = output stores dropped, 4 = input loads dropped
#ifdef RVSR_ABL5
constexpr int ABL5 = RVSR_ABL5;
#else
constexpr int ABL5 = 0;
#endif

#ifdef RVSR_TIMELINE
__device__ unsigned long long rvsr_dbg[512];
extern "C" int rvsr_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rvsr_dbg), sizeof(unsigned long long) * 512); }
#define STAMP(i) do { if (blockIdx.x == 77 && tid == 0) rvsr_dbg[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------
// Epilogue of the bf16x3 kernel: bias comes from LDS (staged once per tile), residual values are fetched in
// one batch per 32x32 tile before any store -- a per-element "load, wait, store" chain costs an L2 round trip
// per element (measured 26 K cycles per tile for the 64 stores of a lane).
template <int MT, int MODE>
__device__ __forceinline__ void conv2_epilogue(f32x16 (&acc)[MT][2], const ConvFwdParams& p, const float* bias_s, int b,
                                               int o0, int row0, int col, int hi) {
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
            float resv[16];
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = o0 + m * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                    const bool ok = col_ok && o < p.Co;
                    resv[r] = p.res[ok ? ((size_t)b * p.Co + o) * HW + pix : 0];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ol = m * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                const int o = o0 + ol;
                const bool ok = col_ok && o < p.Co;
                const int oc = ok ? o : 0;
                float v = acc[m][n][r] + bias_s[ol];
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
                    if (MODE == 1) v += resv[r];
                    if (ok) p.out1[((size_t)b * p.Co + oc) * HW + pix] = v;
                }
            }
        }
    }
}

// 16-byte-store epilogue for the plain / residual cases (MODE 0 / 1) when Wout % 4 == 0 and the output (and residual)
// base pointers are 16-byte aligned: each 4-register group (4 consecutive channels x this lane's pixel) is transposed
// inside the lane quad, after which lane j holds channel j of the group at 4 consecutive pixels.
template <int MT, int MODE>
__device__ __forceinline__ void conv2_epilogue_v4(f32x16 (&acc)[MT][2], const ConvFwdParams& p, const float* bias_s, int b,
                                                  int o0, int row0, int x0, int lo, int hi) {
    const float neg = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
    const int j = lo & 3, col4 = x0 + (lo & ~3);  // this lane's channel-in-group and first pixel column after the transpose
    const bool col_ok = col4 < p.Wout;            // Wout % 4 == 0: the float4 is entirely inside or outside
    // Raw buffer addressing (p.vec4 implies one batch element of the output spans < 2 GB, rvsr_launch_conv_fwd2): a 32-bit lane offset
    // (channel within the group, pixel) computed once per row + the channel group's byte offset (one add); a lane without an
    // element gets an offset beyond the view (its load returns 0, its store is dropped).  The pointer form cost ~13 vector
    // instructions per store incl. a quarter-rate 32-bit multiply and 64-bit adds: a quarter of the epilogue's issue time.
    // The group offset is added on the vector side ON PURPOSE: passed as the store's SGPR soffset, which the next store's s_add then
    // rewrites a few instructions later, the second wave of every SIMD wrote some float4s of its rows to the wrong channel group
    // (measured: wrong x components in lanes 12-15 / 28-31 of waves 4-7, varying from run to run; loads with a changing soffset are
    // fine everywhere in this library) -- a store appears to read its soffset later than the SALU may overwrite it.
    const unsigned HW = (unsigned)(p.Hout * p.Wout), HW4 = 4u * HW;
    const __amdgpu_buffer_rsrc_t out_rs = buf_view_2g(p.out1 + (size_t)b * p.Co * HW);
    const __amdgpu_buffer_rsrc_t res_rs = buf_view_2g(MODE == 1 ? p.res + (size_t)b * p.Co * HW : p.out1);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = row0 + n;
        if (row >= p.Hout) continue;  // wave-uniform
        const unsigned lane_off = 4u * ((unsigned)(4 * hi + j) * HW + (unsigned)row * p.Wout + col4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float4 resv[4];
            unsigned off[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ob = o0 + m * 32 + 8 * rg;   // (uniform) first channel of the group of 8
                off[rg] = col_ok && ob + 4 * hi + j < p.Co ? lane_off : 0x80000000u;
                if (MODE == 1) {
                    typedef float f32x4v __attribute__((ext_vector_type(4)));
                    const f32x4v q = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(res_rs, (int)(off[rg] + (unsigned)ob * HW4), 0, 0));
                    resv[rg] = make_float4(q.x, q.y, q.z, q.w);
                }
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float r0 = acc[m][n][4 * rg + 0], r1 = acc[m][n][4 * rg + 1], r2 = acc[m][n][4 * rg + 2], r3 = acc[m][n][4 * rg + 3];
                quad_transpose4(r0, r1, r2, r3, lo);
                const int ol = m * 32 + 8 * rg + 4 * hi + j;
                const float bb = bias_s[ol];
                float4 v = make_float4(r0 + bb, r1 + bb, r2 + bb, r3 + bb);
                v.x = v.x > 0.f ? v.x : v.x * neg; v.y = v.y > 0.f ? v.y : v.y * neg;
                v.z = v.z > 0.f ? v.z : v.z * neg; v.w = v.w > 0.f ? v.w : v.w * neg;
                if (MODE == 1) { v.x += resv[rg].x; v.y += resv[rg].y; v.z += resv[rg].z; v.w += resv[rg].w; }
                buf_store4(out_rs, off[rg] + (unsigned)(o0 + m * 32 + 8 * rg) * HW4, 0u, v);
            }
        }
    }
}

// Epilogue of the 8 x 64 tile (conv_fwd5_kernel<.., WIDE>): acc[m][0] / acc[m][1] are the left / right 32 pixels of ONE row.  After
// the quad transpose a lane holds 4 consecutive pixels of channel 8 rg + 4 hi + j in each of the two register sets; one
// v_permlane32_swap per register then gives the lower lane half both pixel halves of channel 8 rg + j (register set 0) and the
// upper lane half both halves of channel 8 rg + 4 + j (register set 1), so that a store instruction covers 4 channels x 64 pixels
// = 4 runs of 256 bytes.  (Inline assembly: this hipcc lowers the builtin's second result to a copy of the first.)
__device__ __forceinline__ void half_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
}
template <int MT, int MODE>
__device__ __forceinline__ void conv2_epilogue_wide(f32x16 (&acc)[MT][2], const ConvFwdParams& p, const float* bias_s, int b,
                                                    int o0, int row, int x0, int lo, int hi) {
    if (row >= p.Hout) return;   // wave-uniform
    const float neg = (p.act == 0 || MODE == 4) ? 1.f : (p.act == 1 ? 0.f : p.slope);
    // After transpose + half swap lane 4 G + j holds channel j, pixel quad G = 8 hi + (lo >> 2) of the row.  The store path wants
    // CONSECUTIVE lanes on consecutive addresses (store_micro.hip: 4 x 256 B per instruction = 26 B/clk with lanes 16 c .. 16 c + 15
    // on channel c, 15 B/clk with the channels interleaved lane by lane): one ds_bpermute per register moves the value of lane
    // 4 G + j to lane 16 j + G (LDS crossbar only, no LDS memory).
    const int lane = 32 * hi + lo, jn = lane >> 4, Gn = lane & 15;
    const int src4 = 4 * (4 * Gn + jn);                       // byte index of the source lane
    const int col4 = x0 + 4 * Gn;
    const bool col_ok = col4 < p.Wout;                        // Wout % 4 == 0
    // raw buffer addressing as in conv2_epilogue_v4: a 32-bit lane offset computed once per row + the channel group's offset, added on the vector side
    const unsigned HW = (unsigned)(p.Hout * p.Wout), HW4 = 4u * HW;
    const __amdgpu_buffer_rsrc_t out_rs = buf_view_2g(p.out1 + (size_t)b * p.Co * HW);
    const __amdgpu_buffer_rsrc_t res_rs = buf_view_2g(MODE == 1 || MODE == 4 ? p.res + (size_t)b * p.Co * HW : p.out1);
    const unsigned lane_off = 4u * ((unsigned)jn * HW + (unsigned)row * p.Wout + col4);
    const int j = lo & 3;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float4 resv[2];
            unsigned off[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ob = o0 + m * 32 + 8 * rg + 4 * s;   // (uniform) first of the instruction's 4 channels
                off[s] = col_ok && ob + jn < p.Co ? lane_off : 0x80000000u;
                if (MODE == 1 || MODE == 4) {
                    typedef float f32x4v __attribute__((ext_vector_type(4)));
                    const f32x4v q = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(res_rs, (int)(off[s] + (unsigned)ob * HW4), 0, 0));
                    resv[s] = make_float4(q.x, q.y, q.z, q.w);
                }
            }
            float r[2][4];
            const float bb = bias_s[m * 32 + 8 * rg + 4 * hi + j];   // (before the swap a lane's channel is 8 rg + 4 hi + j in both sets)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                r[n][0] = acc[m][n][4 * rg + 0]; r[n][1] = acc[m][n][4 * rg + 1]; r[n][2] = acc[m][n][4 * rg + 2]; r[n][3] = acc[m][n][4 * rg + 3];
                quad_transpose4(r[n][0], r[n][1], r[n][2], r[n][3], lo);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = r[n][e] + bb;
                    r[n][e] = v > 0.f ? v : v * neg;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) half_swap(r[0][e], r[1][e]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    r[s][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, r[s][e])));
#pragma unroll
            for (int s = 0; s < 2; ++s) {   // set s: lanes 16 c .. 16 c + 15 = channel 8 rg + 4 s + c, the 64 pixels of the row
                float4 v = make_float4(r[s][0], r[s][1], r[s][2], r[s][3]);
                if (MODE == 1) { v.x += resv[s].x; v.y += resv[s].y; v.z += resv[s].z; v.w += resv[s].w; }
                if (MODE == 4) {   // gradient mask: the derivative of the activation whose output `res` is (p.act == 3, neg == 1 above)
                    v.x *= resv[s].x > 0.f ? 1.f : p.slope; v.y *= resv[s].y > 0.f ? 1.f : p.slope;
                    v.z *= resv[s].z > 0.f ? 1.f : p.slope; v.w *= resv[s].w > 0.f ? 1.f : p.slope;
                }
                if (!(ABL5 & 2) || (m == 0 && rg == 0 && s == 0))   // (ABL5 & 2: one store in 16 kept so that the epilogue's arithmetic stays alive)
                buf_store4(out_rs, off[s] + (unsigned)(o0 + m * 32 + 8 * rg + 4 * s) * HW4, 0u, v);
            }
        }
    }
}

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
    float* bias_s = reinterpret_cast<float*>(ws_lo + WVEC);  // [MP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const TView& va = p.in.a;
    const TView& vb = p.in.b;
    const int C1 = va.C, Ctot = va.C + vb.C;
    const int nchunks = (Ctot + 16 * CCG - 1) / (16 * CCG);

    // ---- persistent schedule: the work items (tile, m-block, batch) are split into 8 contiguous ranges, one per
    // XCD (workgroup w is observed on XCD w % 8: neighbouring tiles then share an L2); inside a range the XCD's
    // workgroups take items round-robin.
    const unsigned nmb = (p.Co + MP - 1) / MP, nty = (p.Hout + TH - 1) / TH, ntx = (p.Wout + TW - 1) / TW;
    const unsigned items = ntx * nty * nmb * p.B;
    const unsigned xcd = blockIdx.x & 7, wq = blockIdx.x >> 3, nwq = (gridDim.x + 7 - xcd) >> 3;
    const unsigned q8 = items >> 3, r8 = items & 7;
    const unsigned range0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned range1 = range0 + q8 + (xcd < r8 ? 1 : 0);

    int x0 = 0, y0 = 0, mb = 0, b = 0;
    auto decode = [&](unsigned S) {
        const unsigned t = S % (p.ntx * nty);
        x0 = (int)(t % p.ntx) * TW;
        y0 = (int)(t / p.ntx) * TH;
        mb = (int)((S / (p.ntx * nty)) % nmb);
        b = (int)(S / (p.ntx * nty * nmb));
    };

    // ---- staging.  EVERY global load below is unconditional (addresses are clamped into the tensor, validity is
    // kept as a count and applied when the value is written to LDS): a load inside a divergent `if` makes hipcc
    // wait for it at the join, which serialises the loads of a thread into L2/HBM round trips (measured: 8 K
    // cycles to "issue" 24 loads).  Loads of chunk k+1 are issued after the barrier that opens chunk k's MFMA
    // phase and consumed only after it.
    constexpr int NIT = (NOCT * NPOS + NTHR - 1) / NTHR;   // input items (octet x position) per thread
    constexpr int NWV = (2 * WVEC + NTHR - 1) / NTHR;      // weight vectors per thread
    float vin[NIT][8];
    float ain[ACT_IN ? NIT : 1][8];  // saved activation outputs (sign -> derivative), only for data gradients
    int nvalid[NIT];                 // number of real channels in the item's octet (0: position outside the image)

    __amdgpu_buffer_rsrc_t xa_rs = buf_view_2g(va.p), xb_rs = buf_view_2g(va.p), act_rs = buf_view_2g(va.p);
    auto tile_views = [&]() {   // buffer views of the current tile's batch element (plain view only)
        if (va.mode != 0) return;
        const size_t hw = (size_t)va.Hs * va.Ws;
        xa_rs = buf_view_2g(va.p + (size_t)b * va.C * hw);
        if (vb.C) xb_rs = buf_view_2g(vb.p + (size_t)b * vb.C * hw);
        if (ACT_IN) act_rs = buf_view_2g(va.act + (size_t)b * va.C * hw);
    };
    // items i_lo .. i_hi-1 only (compile-time bounds after unrolling): the stride-2 instantiations stage NIT = 9 items = 72
    // registers per thread and spilled 130-180 of them when all were in flight at once; they go in batches of NB
    auto issue_loads = [&](int chunk, int i_lo = 0, int i_hi = 1 << 20) {
        const int c0 = chunk * 16 * CCG;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (i < i_lo || i >= i_hi) continue;
            const int it_raw = tid + i * NTHR;
            const bool live = it_raw < NOCT * NPOS;
            const int it = live ? it_raw : 0;
            const int oc = it / NPOS, pos = it - oc * NPOS;
            const int r = pos / IW, s = pos - r * IW;
            const int gy = y0 * STRIDE - PAD + r, gx = x0 * STRIDE - PAD + s;
            const int cb = c0 + oc * 8;
            const bool inb = live && gy >= 0 && gx >= 0 && gy < va.Hv && gx < va.Wv && cb < Ctot;
            const int gyc = gy < 0 ? 0 : (gy >= va.Hv ? va.Hv - 1 : gy), gxc = gx < 0 ? 0 : (gx >= va.Wv ? va.Wv - 1 : gx);
            if (va.mode == 0) {  // (uniform branch)
                // raw buffer loads (see conv_fwd5_kernel): the channel plane is a uniform byte offset, the lane offset is 32-bit,
                // and everything that must read as zero (padding, channels beyond the tensor, threads without an item) gets
                // an offset beyond the 2 GB view.  The scalar path paid a 64-bit multiply-add chain per loaded value.
                const bool second = c0 >= C1;   // (uniform: rvsr_launch_conv_fwd2 requires C1 % chunk == 0 for a concat)
                const int Cb = second ? vb.C : va.C, cl0 = second ? c0 - C1 : c0;
                const unsigned hw4 = 4u * (unsigned)(va.Hs * va.Ws);
                const bool pos_ok = live && gy >= 0 && gx >= 0 && gy < va.Hv && gx < va.Wv;
                const unsigned vo = pos_ok ? 4u * (unsigned)(gy * va.Ws + gx) + (unsigned)(8 * oc) * hw4 : 0x80000000u;
                const int lane_nch = Cb - cl0 - 8 * oc;
                nvalid[i] = 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned vo_j = j < lane_nch ? vo : 0x80000000u;
                    const unsigned so = (unsigned)(cl0 + j) * hw4;
                    vin[i][j] = buf_load(second ? xb_rs : xa_rs, vo_j, so);
                    if (ACT_IN) ain[i][j] = buf_load(act_rs, vo_j, so);
                }
            } else {  // mode 2: pixel-unshuffle view, virtual channel c -> stored (c>>2, 2y+((c>>1)&1), 2x+(c&1))
                const size_t hw = (size_t)va.Hs * va.Ws;
                const int cbc = inb ? cb : 0;
                const size_t base = ((size_t)b * (va.C >> 2) + (cbc >> 2)) * hw + (size_t)(2 * gyc) * va.Ws + 2 * gxc;
                nvalid[i] = inb ? 8 : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const size_t idx = base + (j >> 2) * hw + ((j >> 1) & 1) * va.Ws + (j & 1);
                    vin[i][j] = va.p[idx];
                    if (ACT_IN) ain[i][j] = va.act[idx];
                }
            }
        }
    };
    auto commit_inputs = [&](int i_lo = 0, int i_hi = 1 << 20) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (i < i_lo || i >= i_hi) continue;
            const int it = tid + i * NTHR;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = ACT_IN ? vin[i][j] * (ain[i][j] > 0.f ? 1.f : va.slope) : vin[i][j];
                v[j] = j < nvalid[i] ? x : 0.f;
            }
            bf16x8 h8, l8;
            split8(v, h8, l8);
            if (it < NOCT * NPOS) {
                xs_hi[it] = h8;
                xs_lo[it] = l8;
            }
        }
    };
    auto commit_to_lds = [&](int chunk, bool inputs_done = false) {
        // packed weights: straight 16-byte copy (L2-resident), in batches of <= 9 vectors per thread: all loads of
        // a batch are issued before its LDS writes.  When registers allow (forward variants) the first batch's
        // loads fly while the input tile is converted.
        const bf16x8* src = reinterpret_cast<const bf16x8*>(p.wpack) + ((size_t)mb * nchunks + chunk) * 2 * WVEC;
        constexpr int WB = MT >= 4 ? 3 : 9;  // MT = 4 keeps 128 accumulator registers live: smaller batches
        constexpr bool OVERLAP = !ACT_IN && MT <= 2;
        if (!OVERLAP && !inputs_done) commit_inputs();
#pragma unroll
        for (int base = 0; base < NWV; base += WB) {
            bf16x8 wv[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int e = tid + (base + i) * NTHR;
                wv[i] = src[(base + i < NWV && e < 2 * WVEC) ? e : 0];
            }
            if (OVERLAP && base == 0 && !inputs_done) commit_inputs();
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int e = tid + (base + i) * NTHR;
                if (base + i < NWV && e < 2 * WVEC) ws_hi[e] = wv[i];
            }
        }
        if (chunk == 0 && tid < MP) {
            const int o = mb * MP + tid;
            bias_s[tid] = (p.bias != nullptr && o < p.Co) ? p.bias[o] : 0.f;
        }
    };

    // register prefetch only where the register file has room for it; the other variants stage synchronously
    constexpr bool PF = !ACT_IN && MT <= 2 && STRIDE == 1;   // (stride 2 stages 9 items = 72 registers per thread: prefetching them spilled 181)
    unsigned S = range0 + wq;
    if (S < range1) {
        decode(S);
        tile_views();
        if (PF) issue_loads(0);
    }
    for (; S < range1; S += nwq) {
        f32x16 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][0] = zero16();
            acc[m][1] = zero16();
        }
        const int cx0 = x0, cy0 = y0, cmb = mb, cb_ = b;
        STAMP(0);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            constexpr int NB = 3;
            if (!PF && STRIDE == 2) {   // batched: NB items in flight at a time
#pragma unroll
                for (int i0 = 0; i0 < NIT; i0 += NB) {
                    issue_loads(chunk, i0, i0 + NB);
                    commit_inputs(i0, i0 + NB);
                }
                commit_to_lds(chunk, true);
            } else {
                if (!PF) issue_loads(chunk);
                commit_to_lds(chunk);
            }
            STAMP(1 + chunk * 5);
            __syncthreads();
            STAMP(2 + chunk * 5);
            if (PF && chunk + 1 < nchunks) issue_loads(chunk + 1);
            STAMP(3 + chunk * 5);
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
            STAMP(4 + chunk * 5);
            __syncthreads();
            STAMP(5 + chunk * 5);
        }
        if (p.ps)
            conv2_epilogue<MT, 3>(acc, p, bias_s, cb_, cmb * MP, cy0 + wave * 2, cx0 + lo, hi);
        else if (p.out2 != nullptr)
            conv2_epilogue<MT, 2>(acc, p, bias_s, cb_, cmb * MP, cy0 + wave * 2, cx0 + lo, hi);
        else if (p.res != nullptr) {
            if (p.vec4) conv2_epilogue_v4<MT, 1>(acc, p, bias_s, cb_, cmb * MP, cy0 + wave * 2, cx0, lo, hi);
            else conv2_epilogue<MT, 1>(acc, p, bias_s, cb_, cmb * MP, cy0 + wave * 2, cx0 + lo, hi);
        } else {
            if (p.vec4) conv2_epilogue_v4<MT, 0>(acc, p, bias_s, cb_, cmb * MP, cy0 + wave * 2, cx0, lo, hi);
            else conv2_epilogue<MT, 0>(acc, p, bias_s, cb_, cmb * MP, cy0 + wave * 2, cx0 + lo, hi);
        }
        STAMP(30);
        if (S + nwq < range1) {  // the stores above are asynchronous: the next tile's loads go out right behind them
            decode(S + nwq);
            tile_views();
            if (PF) issue_loads(0);
        }
        STAMP(31);
        // bias_s of this tile is read by the epilogue above and rewritten by the next tile's chunk-0 commit:
        // that commit is followed by a barrier before any MFMA, and every wave passed the last barrier of this
        // tile before its epilogue; a wave still in its epilogue while another already commits the next tile
        // would race on bias_s, hence one more barrier.
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 / stride-1 forward (and data gradient), software-pipelined per wave: ONE workgroup of 8 waves per CU, a 16x32-pixel
// tile, a flat sequence of stages (tile, 16-channel chunk) over two LDS buffers (input chunk + weight chunk) with one barrier
// per stage; every wave runs the MFMA loop of every stage and carries its share of the
// staging work INSIDE that loop, a slice per tap.  Measured on MI355X: a wave's own VALU / LDS / load instructions
// issue in the gaps of its MFMA stream (up to ~5 per 32x32x16 MFMA), whereas instructions of the OTHER wave of the
// SIMD get roughly one issue slot per MFMA (~18 cycles per instruction in round 1's ping-pong kernel with dedicated staging
// slots, removed in round 3; a variant with dedicated staging waves ran 2x slower).  Registers of stage q+1 are published to LDS during taps 0-3 (inputs) and
// 4-8 (weights) of stage q and refilled at once with the loads of stage q+2, so the staging registers are not doubled.
// VEC: 0 = scalar staging (any view), 1 = vector staging of a plain view, 2 = vector staging of a pixel-unshuffle view
// WIDE: the workgroup tile is 8 rows x 64 pixels (a wave = one row, its two N tiles side by side) instead of 16 x 32 (a wave = two
// rows): a wave then owns 256 contiguous bytes of every output channel row, and the epilogue (conv2_epilogue_wide) writes them as
// 4 channels x 256 B per instruction instead of 8 channels x 128 B -- the CU's store path takes a 256-byte run at the price of a
// 128-byte one (tools/micro/store_micro.hip: 26.5 vs 13.5 B/clk), and the output burst of a tile was 15 % of the training step.
// NT: terms of the bf16 product (rvsr_set_gemm_mode): 3 = hi*hi + hi*lo + lo*hi (f32-grade, the default); 2 = the weights' lo part is
// dropped (W rounded to bf16, activations / gradients full: two MFMAs per product, the lo half of the weight image is neither loaded nor
// published); 1 = plain bf16 operands (one MFMA, no lo image of the input tile either).  Accumulation is f32 in every mode.
// NT = 4 is not a term count but the f16 + fp8 product FORMAT (ConvFwdParams.fmt; DESIGN.md 5h, experiments/f16fp8/README.md): a1 * b1 in f16 +
// a1 * b2 + a2 * b1 in fp8 e4m3 with a1 = f16(a), a2 = a - a1 -- per tap four v_mfma_f32_32x32x16_f16, per pair of taps four
// v_mfma_scale_f32_32x32x64_f8f6f4: 56 matrix instructions per stage instead of 108, ~1.2e-5 per 576-deep GEMM instead of 4.6e-6.
template <int MT, bool ACT_IN, int VEC, bool WIDE = false, int NT = 3>
__global__ __launch_bounds__(512, 2) void conv_fwd5_kernel(const ConvFwdParams p) {
    constexpr int KS = 3, T = 9, PAD = 1, NW = 8, TH = WIDE ? NW : 2 * NW, TW = WIDE ? 64 : 32, NTHR = NW * 64;
    constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
    constexpr int MP = MT * 32, NOCT = 2, NPOS = IH * IW, NX = NOCT * NPOS;
    constexpr int WVEC = T * NOCT * MP;  // 16-byte vectors per weight part (hi or lo)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* xs_base = reinterpret_cast<bf16x8*>(smem_raw);                // [2 buffers][hi|lo][NX]
    bf16x8* ws_base = xs_base + 2 * 2 * NX;                               // [2 buffers][hi|lo][WVEC]
    float* bias_base = reinterpret_cast<float*>(ws_base + 2 * 2 * WVEC);  // [4][MP]
    // write-only slot: LDS stores of lanes without a valid destination land here, so that the publishing code has no
    // divergent branch and shares a basic block (and the MFMA issue gaps) with the tap's matrix instructions
    bf16x8* const sink = reinterpret_cast<bf16x8*>(bias_base + 4 * MP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const TView& va = p.in.a;
    const TView& vb = p.in.b;
    const int C1 = va.C, Ctot = va.C + vb.C;
    const int nchunks = (Ctot + 15) / 16;

    // persistent schedule as in conv_fwd2_kernel: 8 contiguous item ranges, one per XCD
    const unsigned nmb = (p.Co + MP - 1) / MP, nty = (p.Hout + TH - 1) / TH, ntx = (p.Wout + TW - 1) / TW;
    const unsigned items = ntx * nty * nmb * p.B;
    const unsigned xcd = blockIdx.x & 7, wq = blockIdx.x >> 3, nwq = (gridDim.x + 7 - xcd) >> 3;
    const unsigned q8 = items >> 3, r8 = items & 7;
    const unsigned range0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned range1 = range0 + q8 + (xcd < r8 ? 1 : 0);
    const unsigned S0 = range0 + wq;
    if (S0 >= range1) return;
    const int ntile = (int)((range1 - S0 + nwq - 1) / nwq);
    const int Q = ntile * nchunks;

    struct Tile { int x0, y0, mb, b; };
    auto tile_of = [&](int k) {
        const unsigned S = S0 + (unsigned)k * nwq;
        Tile t;
        const unsigned u = S % (ntx * nty);
        t.x0 = (int)(u % ntx) * TW;
        t.y0 = (int)(u / ntx) * TH;
        t.mb = (int)((S / (ntx * nty)) % nmb);
        t.b = (int)(S / (ntx * nty * nmb));
        return t;
    };

    // Staging.  VEC (stored width % 4 == 0, 16-byte aligned planes, plain view): one item = (octet, tile row, group of
    // 4 pixels starting at x0 - 4 + 4g), fetched with 8 aligned 16-byte loads (one per channel) -- a third of the
    // load instructions and address arithmetic of the scalar path.  The 10 groups of a row cover 40 pixels of which
    // the tile uses 34 (groups 0 and 9 contribute one pixel each).  All loads are unconditional (clamped addresses,
    // validity applied at the LDS write), see conv_fwd2_kernel.
    constexpr int NG = TW / 4 + 2;   // 4-pixel groups of a staged row: x0 - 4 .. x0 + TW + 3
    // buffer views (VEC staging): the packed weight image, and the current tile's batch element of each input / of act'
    __amdgpu_buffer_rsrc_t w_rs = buf_view_2g(p.wpack), xa_rs = buf_view_2g(va.p), xb_rs = buf_view_2g(va.p), act_rs = buf_view_2g(va.p);
    constexpr int NIT = VEC ? 1 : (NX + NTHR - 1) / NTHR;  // input items per thread
    constexpr int WPARTS = NT >= 3 ? 2 : 1;                // parts of the weight image this kernel reads: [hi | lo] or hi only
    constexpr int NWV = (WPARTS * WVEC + NTHR - 1) / NTHR; // weight vectors per thread
    constexpr int NV = VEC ? 4 : 1;                        // pixels per item
    int it_oc[NIT], it_sp[NIT], it_dst[NIT];
    bool it_pos_ok[NIT];
    auto item_geom = [&](const Tile& t) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (VEC) {
                // byte offset of the 4-pixel group inside a channel plane, or an offset beyond every buffer view for a group
                // outside the image / a thread without an item: raw buffer loads return 0 there, which IS the zero padding
                // (no clamped address, no select per loaded value)
                const bool live = tid < NOCT * IH * NG;
                const int it = live ? tid : 0;
                const int oc = it / (IH * NG), rem = it - oc * (IH * NG);
                const int r = rem / NG, g = rem - r * NG;
                const int gy = t.y0 - PAD + r, gx = t.x0 - 4 + 4 * g;
                it_oc[i] = oc;
                it_pos_ok[i] = live && gy >= 0 && gy < va.Hv && gx >= 0 && gx < va.Wv;
                if (VEC == 3)   // zero-insert view: only even virtual rows hold data, at stored (gy / 2, gx / 2) and (gy / 2, gx / 2 + 1)
                    it_sp[i] = it_pos_ok[i] && !(gy & 1) && (gy >> 1) < va.Hs ? 4 * ((gy >> 1) * va.Ws + (gx >> 1)) : (int)0x80000000;
                else
                    it_sp[i] = it_pos_ok[i] ? (VEC == 2 ? 8 * (gy * va.Ws + gx) : 4 * (gy * va.Ws + gx)) : (int)0x80000000;   // (mode 2: stored row 2 gy, column 2 gx)
                it_dst[i] = live ? (oc * IH + r) * IW + 4 * g - 3 : -100;  // LDS slot of the group's first pixel
            } else {
                const int it_raw = tid + i * NTHR;
                const bool live = it_raw < NX;
                const int it = live ? it_raw : 0;
                const int oc = it / NPOS, pos = it - oc * NPOS;
                const int r = pos / IW, s = pos - r * IW;
                const int gy = t.y0 - PAD + r, gx = t.x0 - PAD + s;
                it_oc[i] = oc;
                bool ok = live && gy >= 0 && gx >= 0 && gy < va.Hv && gx < va.Wv;
                const int gyc = gy < 0 ? 0 : (gy >= va.Hv ? va.Hv - 1 : gy), gxc = gx < 0 ? 0 : (gx >= va.Wv ? va.Wv - 1 : gx);
                if (va.mode == 1) {  // zero-insert view: only even (y, x) carry data
                    const int ys = gyc >> 1, xs = gxc >> 1;
                    ok = ok && !((gyc | gxc) & 1) && ys < va.Hs && xs < va.Ws;
                    it_sp[i] = (ys < va.Hs ? ys : 0) * va.Ws + (xs < va.Ws ? xs : 0);
                } else {
                    it_sp[i] = va.mode == 0 ? gyc * va.Ws + gxc : (2 * gyc) * va.Ws + 2 * gxc;
                }
                it_pos_ok[i] = ok;
                it_dst[i] = live ? it : -100;
            }
        }
    };
    float vin[NIT][8][NV];
    float ain[ACT_IN ? NIT : 1][8][NV];
    int nvalid[NIT];
    bf16x8 wv[NWV];
    // slices: wsel < 0: all weight vectors, else only vector wsel (>= NWV: none); xsel < 0: all input loads, else
    // VEC: channels 4*xsel .. 4*xsel+3 of the item, non-VEC: item xsel (out-of-range: none)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    auto issue_loads = [&](const Tile& t, int chunk, int wsel, int xsel) {
        if (VEC) {
            // Raw buffer addressing: weight image and the tile's batch element are buffer views, the (chunk, channel) part of an
            // address is a uniform byte offset in an SGPR, the per-lane part is the 32-bit it_sp (+ the lane's octet).  The
            // staging carries no 64-bit per-lane address arithmetic and no validity selects (launch_fwd5 checks the 4 GB spans).
            const unsigned wbase = (unsigned)(((size_t)t.mb * nchunks + chunk) * 2 * WVEC) * 16u;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                if (wsel >= 0 && wsel != i) continue;
                const int e = tid + i * NTHR;
                const unsigned vo = e < WPARTS * WVEC ? (unsigned)tid * 16u : 0x80000000u;   // (only the last vector is ragged)
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)vo, (int)(wbase + (unsigned)i * NTHR * 16u), 0);
                wv[i] = __builtin_bit_cast(bf16x8, w);
            }
            if (xsel >= 8) return;
            const int c0 = chunk * 16;
            const bool second = c0 >= C1;                 // (uniform: C1 % 16 == 0 whenever there is a second input)
            const int Cb = second ? vb.C : va.C, cl0 = second ? c0 - C1 : c0;
            const unsigned hw4 = 4u * (unsigned)(va.Hs * va.Ws);
            const int lane_nch = Cb - cl0 - 8 * it_oc[0];   // channels of the lane's octet that exist (<= 0: none, >= 8: all)
            nvalid[0] = 8;
            if (VEC == 2) {
                // pixel-unshuffle view (the data gradient of conv + PixelShuffle): virtual channel c of pixel (y, x) is stored
                // channel c >> 2 at (2y + ((c >> 1) & 1), 2x + (c & 1)).  The item's 8 channels x 4 pixels are 2 stored
                // channels x 2 stored rows x 8 stored columns = eight 16-byte loads (csl, sy, half); element e of a load is
                // channel 4 csl + 2 sy + (e & 1) of pixel 2 half + (e >> 1) -- a compile-time renaming of registers.
                const unsigned row4 = 4u * (unsigned)va.Ws, pl4 = row4 * (unsigned)va.Hs;   // stored row / plane in bytes
                const unsigned vo = (unsigned)it_sp[0] + (unsigned)(2 * it_oc[0]) * pl4;
                const unsigned vo_l = lane_nch > 0 ? vo : 0x80000000u;   // (virtual channel counts are multiples of 4 here)
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const int csl = l >> 2, sy = (l >> 1) & 1, half = l & 1;
                    if (xsel >= 0 && csl != xsel) continue;
                    const unsigned so = (unsigned)((cl0 >> 2) + csl) * pl4 + (unsigned)sy * row4 + 16u * half;
                    const f32x4v q = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xa_rs, (int)vo_l, (int)so, 0));
                    const int j0 = 4 * csl + 2 * sy, e0 = (2 * half) & (NV - 1), e1 = (2 * half + 1) & (NV - 1);
                    vin[0][j0][e0] = q.x; vin[0][j0 + 1][e0] = q.y; vin[0][j0][e1] = q.z; vin[0][j0 + 1][e1] = q.w;
                    if (ACT_IN) {
                        const f32x4v a4 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(act_rs, (int)vo_l, (int)so, 0));
                        ain[0][j0][e0] = a4.x; ain[0][j0 + 1][e0] = a4.y; ain[0][j0][e1] = a4.z; ain[0][j0 + 1][e1] = a4.w;
                    }
                }
                return;
            }
            const unsigned vo = (unsigned)it_sp[0] + (unsigned)(8 * it_oc[0]) * hw4;
            if (VEC == 3) {
                // zero-insert view (the data gradient of a stride-2 conv): of an item's 4 virtual pixels only 0 and 2 hold data --
                // one 8-byte load per channel; pixels 1 and 3, and every odd virtual row, are zeros
                typedef float f32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (xsel >= 0 && (j >> 2) != xsel) continue;
                    const unsigned so = (unsigned)(cl0 + j) * hw4;
                    const unsigned vo_j = j < lane_nch ? vo : 0x80000000u;
                    const f32x2v q = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(xa_rs, (int)vo_j, (int)so, 0));
                    vin[0][j][0] = q.x; vin[0][j][NV > 2 ? 2 : 0] = q.y; vin[0][j][NV > 1 ? 1 : 0] = 0.f; vin[0][j][NV > 3 ? 3 : 0] = 0.f;
                    if (ACT_IN) {
                        const f32x2v a2 = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(act_rs, (int)vo_j, (int)so, 0));
                        ain[0][j][0] = a2.x; ain[0][j][NV > 2 ? 2 : 0] = a2.y; ain[0][j][NV > 1 ? 1 : 0] = 1.f; ain[0][j][NV > 3 ? 3 : 0] = 1.f;
                    }
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (xsel >= 0 && (j >> 2) != xsel) continue;
                const unsigned so = (unsigned)(cl0 + j) * hw4;
                const unsigned vo_j = j < lane_nch ? vo : 0x80000000u;   // a channel beyond the tensor reads zeros, not memory
                // (the whole vector is bit-cast at once: per-element extraction of the builtin's result is mis-folded by this
                // hipcc into four copies of element 0)
                const f32x4v q = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(second ? xb_rs : xa_rs, (int)vo_j, (int)so, 0));
                vin[0][j][0] = q.x; vin[0][j][NV > 1 ? 1 : 0] = q.y; vin[0][j][NV > 2 ? 2 : 0] = q.z; vin[0][j][NV > 3 ? 3 : 0] = q.w;
                if (ACT_IN) {
                    const f32x4v a4 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(act_rs, (int)vo_j, (int)so, 0));
                    ain[0][j][0] = a4.x; ain[0][j][NV > 1 ? 1 : 0] = a4.y; ain[0][j][NV > 2 ? 2 : 0] = a4.z; ain[0][j][NV > 3 ? 3 : 0] = a4.w;
                }
            }
            return;
        }
        const bf16x8* src = reinterpret_cast<const bf16x8*>(p.wpack) + ((size_t)t.mb * nchunks + chunk) * 2 * WVEC;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            if (wsel >= 0 && wsel != i) continue;
            const int e = tid + i * NTHR;
            wv[i] = src[e < WPARTS * WVEC ? e : 0];
        }
        if (xsel >= 8) return;
        const int c0 = chunk * 16;
        const size_t hw = (size_t)va.Hs * va.Ws;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (xsel >= 0 && xsel != i) continue;
            const int cb = c0 + it_oc[i] * 8;
            const bool inb = it_pos_ok[i] && cb < Ctot;
            if (va.mode != 2) {  // (uniform branch)
                const bool second = inb && cb >= C1;  // octets never straddle the two inputs (C1 % 8 == 0)
                const float* bp = second ? vb.p : va.p;
                const float* ap = va.act;
                const int Cb = second ? vb.C : va.C;
                const int cl = inb ? (second ? cb - C1 : cb) : 0;
                nvalid[i] = inb ? (Cb - cl < 8 ? Cb - cl : 8) : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int cj = cl + j < Cb ? cl + j : Cb - 1;
                    const size_t idx = ((size_t)t.b * Cb + cj) * hw + it_sp[i];
                    vin[i][j][0] = bp[idx];
                    if (ACT_IN) ain[i][j][0] = ap[idx];
                }
            } else {  // mode 2: pixel-unshuffle view, virtual channel c -> stored (c>>2, 2y+((c>>1)&1), 2x+(c&1))
                const int cbc = inb ? cb : 0;
                const size_t base = ((size_t)t.b * (va.C >> 2) + (cbc >> 2)) * hw + it_sp[i];
                nvalid[i] = inb ? 8 : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const size_t idx = base + (j >> 2) * hw + ((j >> 1) & 1) * va.Ws + (j & 1);
                    vin[i][j][0] = va.p[idx];
                    if (ACT_IN) ain[i][j][0] = va.act[idx];
                }
            }
        }
    };
    // xsel < 0: the whole input share, else VEC: pixel xsel of the item, non-VEC: item xsel; wsel as in issue_loads
    auto commit = [&](int buf, int wsel, int xsel) {
        bf16x8* xs_hi = xs_base + buf * 2 * NX;
        bf16x8* xs_lo = xs_hi + NX;
        bf16x8* ws = ws_base + buf * 2 * WVEC;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (xsel >= 8 || (!VEC && xsel >= 0 && xsel != i)) continue;
            unsigned vm[8];   // (VEC: the loads already returned zeros wherever the tile needs them)
#pragma unroll
            for (int j = 0; j < 8; ++j) vm[j] = VEC || j < nvalid[i] ? 0xffffffffu : 0u;
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (VEC && xsel >= 0 && xsel != e) continue;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = ACT_IN ? vin[i][j][e] * (ain[i][j][e] > 0.f ? 1.f : va.slope) : vin[i][j][e];
                    v[j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & vm[j]);
                }
                const int dst = it_dst[i] + e;
                bool ok = dst >= 0;
                if (VEC) {  // groups 0 and 9 straddle the tile's 34-pixel row
                    const int s = 4 * (tid % NG) - 3 + e;
                    ok = ok && s >= 0 && s < IW;
                }
                if (NT == 4) {
                    // f16 + fp8 format (NT == 4): hi image = the octet as f16; lo image = [kind][position][16 channels] fp8 e4m3, kind 0: (v - f16(v)) * 2^12,
                    // kind 1: f16(v) -- this octet's 8 bytes of each (the matrix instruction's block scales take the 2^12 back)
                    typedef _Float16 f16x8c __attribute__((ext_vector_type(8)));
                    typedef int i32x2c __attribute__((ext_vector_type(2)));
                    f16x8c h;
                    float r2[8], r1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        h[j] = (_Float16)v[j];
                        r1[j] = (float)h[j];
                        r2[j] = (v[j] - r1[j]) * 4096.f;
                    }
                    i32x2c q2, q1;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        int w2 = 0, w1 = 0;
                        w2 = __builtin_amdgcn_cvt_pk_fp8_f32(r2[4 * k], r2[4 * k + 1], w2, false);
                        w2 = __builtin_amdgcn_cvt_pk_fp8_f32(r2[4 * k + 2], r2[4 * k + 3], w2, true);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(r1[4 * k], r1[4 * k + 1], w1, false);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(r1[4 * k + 2], r1[4 * k + 3], w1, true);
                        q2[k] = w2; q1[k] = w1;
                    }
                    const int pos = dst - it_oc[i] * NPOS;
                    unsigned char* lo_b = reinterpret_cast<unsigned char*>(xs_lo);
                    unsigned char* sink_b = reinterpret_cast<unsigned char*>(sink);
                    *(ok ? xs_hi + dst : sink) = __builtin_bit_cast(bf16x8, h);
                    *reinterpret_cast<i32x2c*>(ok ? lo_b + (size_t)pos * 16 + 8 * it_oc[i] : sink_b) = q2;
                    *reinterpret_cast<i32x2c*>(ok ? lo_b + ((size_t)NPOS + pos) * 16 + 8 * it_oc[i] : sink_b) = q1;
                    continue;
                }
                bf16x8 h8, l8;
                split8(v, h8, l8);
                bf16x8* const dh = ok ? xs_hi + dst : sink;
                bf16x8* const dl = ok ? xs_lo + dst : sink;
                *dh = h8;
                if (NT >= 2) *dl = l8;
            }
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            if (wsel >= 0 && wsel != i) continue;
            const int e = tid + i * NTHR;
            *(e < WPARTS * WVEC ? ws + e : sink) = wv[i];
        }
    };
    // bias of tile k lives in slot k & 3: tiles k-1 .. k+1 can be alive at once when a tile is a single stage
    Tile itile = tile_of(0);
    // geometry + bias of the stage's tile when a new tile starts (bias of tile k lives in slot k & 3: tiles k-1 .. k+1
    // can be alive at once when a tile is a single stage)
    // (tile index, chunk index) of a stage are carried as counters: `q / nchunks` by a run-time divisor expands to ~25 vector
    // instructions incl. quarter-rate integer multiplies, and the stage loop evaluated it seven times per stage -- in a loop
    // whose SQ counters show 6.2 VALU instructions per MFMA (issue-bound, profiles/r02_notes.md)
    auto stage_tile = [&](int q, int k, int ch) {
        if (q >= Q) return;
        if (ch == 0) {
            itile = tile_of(k);
            item_geom(itile);
            if (VEC) {
                const size_t hw = (size_t)va.Hs * va.Ws;
                const size_t img_a = (size_t)(VEC == 2 ? va.C >> 2 : va.C) * hw;   // stored elements per batch element
                xa_rs = buf_view_2g(va.p + (size_t)itile.b * img_a);
                if (vb.C) xb_rs = buf_view_2g(vb.p + (size_t)itile.b * vb.C * hw);
                if (ACT_IN) act_rs = buf_view_2g(va.act + (size_t)itile.b * img_a);
            }
            if (tid < MP) {
                const int o = itile.mb * MP + tid;
                bias_base[(k & 3) * MP + tid] = (p.bias != nullptr && o < p.Co) ? p.bias[o] : 0.f;
            }
        }
    };
    // Unconditional: beyond the last stage the loads repeat an earlier (tile, chunk) and the data is never used.  A uniform
    // `if (q < Q)` around loads makes the compiler's s_waitcnt vmcnt(N) the minimum over both paths -- vmcnt(0) in
    // mid-stage, i.e. the requests of tap 4 were waited for at tap 5 (a full L2/HBM round trip per stage and wave).
    auto issue_stage = [&](int q, int ch, int wsel, int xsel) {
        (void)q;
        issue_loads(itile, ch, wsel, xsel);
    };

    STAMP(60);
    stage_tile(0, 0, 0);
    issue_stage(0, 0, -1, -1);
    commit(0, -1, -1);
    stage_tile(1, 1 / nchunks, 1 % nchunks);
    issue_stage(1, 1 % nchunks, -1, -1);
    __syncthreads();
    int k_cur = 0, c_cur = 0;                          // stage q
    int k_nx2 = 2 / nchunks, c_nx2 = 2 % nchunks;      // stage q + 2

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m][0] = zero16();
        acc[m][1] = zero16();
    }
    for (int q = 0; q < Q; ++q) {
        STAMP(1 + (q & 7) * 3);
        const int buf = q & 1;
        const bf16x8* xs_hi = xs_base + buf * 2 * NX;
        const bf16x8* xs_lo = xs_hi + NX;
        const bf16x8* ws_hi = ws_base + buf * 2 * WVEC;
        const bf16x8* ws_lo = ws_hi + WVEC;
        constexpr bool pub = true;    // publish stage q+1 into the other buffer during this stage (garbage after the last stage)
        bf16x8 ah[2][MT], al[2][MT], bh[2][2], bl[2][2];
        auto fetch = [&](int tap, int slot) {
            const int dy = tap / KS, dx = tap % KS;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[slot][m] = ws_hi[(tap * NOCT + hi) * MP + m * 32 + lo];
                if (NT >= 3) al[slot][m] = ws_lo[(tap * NOCT + hi) * MP + m * 32 + lo];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int idx = WIDE ? (hi * IH + wave + dy) * IW + lo + 32 * n + dx : (hi * IH + wave * 2 + n + dy) * IW + lo + dx;
                bh[slot][n] = xs_hi[idx];
                if (NT >= 2) bl[slot][n] = xs_lo[idx];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const int sl = tap & 1;
#ifdef RVSR_TIMELINE
            if (blockIdx.x == 77 && lane == 0 && q == Q - 6) rvsr_dbg[300 + wave * 12 + tap] = __builtin_amdgcn_s_memtime();
#endif
            if (NT == 4) {
                // f16 + fp8 format (NT == 4): the cross terms a1 * b2 + a2 * b1 of TWO taps in one 64-deep fp8 instruction (lane half 0 holds the
                // a1 / b2 * 2^12 pieces, half 1 the a2 * 2^12 / b1 pieces: the E8M0 block scale of a lane's 32 k undoes the 2^12), issued
                // before the next fetch overwrites the previous tap's fragments; then the main term a1 * b1 in f16
                typedef _Float16 f16x8m __attribute__((ext_vector_type(8)));
                typedef int i32x4m __attribute__((ext_vector_type(4)));
                typedef int i32x8m __attribute__((ext_vector_type(8)));
                const int sa = hi ? 127 - 12 : 127, sb = hi ? 127 : 127 - 12;
                if (tap & 1) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const i32x8m pa = __builtin_shufflevector(__builtin_bit_cast(i32x4m, al[sl ^ 1][m]), __builtin_bit_cast(i32x4m, al[sl][m]), 0, 1, 2, 3, 4, 5, 6, 7);
                            const i32x8m pb = __builtin_shufflevector(__builtin_bit_cast(i32x4m, bl[sl ^ 1][n]), __builtin_bit_cast(i32x4m, bl[sl][n]), 0, 1, 2, 3, 4, 5, 6, 7);
                            acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[m][n], 0, 0, 0, sa, 0, sb);
                        }
                }
                if (tap + 1 < T) fetch(tap + 1, sl ^ 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8m, ah[sl][m]), __builtin_bit_cast(f16x8m, bh[sl][n]), acc[m][n], 0, 0, 0);
                if (tap == T - 1) {   // the ninth tap has no partner: the upper half of its K is zero on the weight side
                    const i32x4m z = {0, 0, 0, 0};
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const i32x8m pa = __builtin_shufflevector(__builtin_bit_cast(i32x4m, al[sl][m]), z, 0, 1, 2, 3, 4, 5, 6, 7);
                            const i32x8m pb = __builtin_shufflevector(__builtin_bit_cast(i32x4m, bl[sl][n]), __builtin_bit_cast(i32x4m, bl[sl][n]), 0, 1, 2, 3, 4, 5, 6, 7);
                            acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(pa, pb, acc[m][n], 0, 0, 0, sa, 0, sb);
                        }
                }
            } else {
            if (tap + 1 < ((ABL5 & 1) ? 2 : T)) fetch(tap + 1, sl ^ 1);   // (ABL5 & 1: fragments of taps 0 and 1 reused for taps 2-8)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(ah[sl][m], bh[sl][n], acc[m][n]);
            if (NT >= 2) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(ah[sl][m], bl[sl][n], acc[m][n]);
            }
            if (NT >= 3) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(al[sl][m], bh[sl][n], acc[m][n]);
            }
            }
            // ---- this wave's own staging work, a slice per tap, in the shadow of the MFMAs above (a wave's own
            // VALU/LDS instructions fill its MFMA issue gaps; another wave's barely do):
            //   taps 0-3: publish pixel / item `tap` of stage q+1 (registers loaded during stage q-1)
            //   taps 4-8: publish weight vector tap-4, then refill the freed registers with stage q+2
            if (tap < 4) {
                if (pub) commit(buf ^ 1, NWV, tap);
            } else {
                if (pub) commit(buf ^ 1, tap - 4, 8);
                if (tap == 4) stage_tile(q + 2, k_nx2, c_nx2);
                issue_stage(q + 2, c_nx2, tap - 4, (ABL5 & 4) ? 8 : VEC ? (tap < 6 ? tap - 4 : 8) : (tap - 4 < NIT ? tap - 4 : 8));   // (ABL5 & 4: no input loads after the first stages)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(2 + (q & 7) * 3);
#ifdef RVSR_TIMELINE
        if (blockIdx.x == 77 && lane == 0 && q == Q - 3) rvsr_dbg[70 + wave] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 77 && lane == 0 && q == Q - 4) rvsr_dbg[90 + wave] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 77 && lane == 0 && q == Q - 6) rvsr_dbg[300 + wave * 12 + 9] = __builtin_amdgcn_s_memtime();
#endif
        const int k = k_cur;
        const bool last_chunk = c_cur == nchunks - 1;
        if (++c_cur == nchunks) { c_cur = 0; ++k_cur; }
        if (++c_nx2 == nchunks) { c_nx2 = 0; ++k_nx2; }
        if (last_chunk) {  // last chunk of tile k: epilogue, fresh accumulators
            const Tile cur = tile_of(k);
            const float* bias_s = bias_base + (k & 3) * MP;
            if (WIDE) {   // (launch_fwd5: 16-byte stores, no pixel shuffle, one output tensor)
                if (p.act == 3) conv2_epilogue_wide<MT, 4>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave, cur.x0, lo, hi);
                else if (p.res != nullptr) conv2_epilogue_wide<MT, 1>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave, cur.x0, lo, hi);
                else conv2_epilogue_wide<MT, 0>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave, cur.x0, lo, hi);
            } else if (p.ps)
                conv2_epilogue<MT, 3>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave * 2, cur.x0 + lo, hi);
            else if (p.out2 != nullptr)
                conv2_epilogue<MT, 2>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave * 2, cur.x0 + lo, hi);
            else if (p.res != nullptr) {
                if (p.vec4) conv2_epilogue_v4<MT, 1>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave * 2, cur.x0, lo, hi);
                else conv2_epilogue<MT, 1>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave * 2, cur.x0 + lo, hi);
            } else {
                if (p.vec4) conv2_epilogue_v4<MT, 0>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave * 2, cur.x0, lo, hi);
                else conv2_epilogue<MT, 0>(acc, p, bias_s, cur.b, cur.mb * MP, cur.y0 + wave * 2, cur.x0 + lo, hi);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m][0] = zero16();
                acc[m][1] = zero16();
            }
        }
        __syncthreads();
        STAMP(3 + (q & 7) * 3);
#ifdef RVSR_TIMELINE
        if (blockIdx.x == 77 && lane == 0 && q == Q - 4) rvsr_dbg[80 + wave] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 77 && lane == 0 && q == Q - 6) rvsr_dbg[300 + wave * 12 + 10] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 77 && lane == 0 && q == Q - 7) rvsr_dbg[300 + wave * 12 + 11] = __builtin_amdgcn_s_memtime();
#endif
    }
    STAMP(61);
#ifdef RVSR_TIMELINE
    if (blockIdx.x == 77 && tid == 0) rvsr_dbg[62] = (unsigned long long)Q;
#endif
}


// ------------------------------------------------------------------------------------------
// host side (called from conv_kernels.hip)
// The f16 + fp8 product format (ConvFwdParams.fmt) exists in the 3x3 / stride-1 kernels with 64-row m-blocks, forward weights only
static inline bool f16fp8_ok(int ksize, int stride, int mt, int w_mode) { return ksize == 3 && stride == 1 && mt == 2 && w_mode == 0; }
static void fwd2_geom(int ksize, int Co, int Ctot, int& mt, int& ccg, int& nchunks, int& nmb) {
    // never more than 2 M tiles per workgroup: the MT = 4 instantiation keeps 128 accumulator registers live and
    // spills (130-190 VGPRs to scratch); two 64-row m-blocks re-stage the input tile but run spill-free
    mt = Co <= 32 ? 1 : 2;
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
    const size_t lds = (size_t)16 * (2 * (2 * CCG) * IH * IW + 2 * T * (2 * CCG) * (MT * 32)) + sizeof(float) * MT * 32;
    auto k = p.in.a.act != nullptr ? conv_fwd2_kernel<KS, STRIDE, MT, CCG, true, NW> : conv_fwd2_kernel<KS, STRIDE, MT, CCG, false, NW>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_fwd2: cannot reserve %zu B of LDS", lds);
    const int nty = (p.Hout + 2 * NW - 1) / (2 * NW);
    const long items = (long)p.ntx * nty * ((p.Co + MT * 32 - 1) / (MT * 32)) * p.B;
    const int slots = 256 * (lds > 80 * 1024 ? 1 : 2);  // persistent: 2 workgroups per CU when LDS allows
    dim3 grid((unsigned)(items < slots ? items : slots), 1, 1);
    hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_fwd2 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

template <int MT>
static int launch_fwd5(const ConvFwdParams& p, hipStream_t st) {
    constexpr int NX = 2 * 18 * 34, WVEC = 9 * 2 * MT * 32;
    const size_t lds = (size_t)16 * (2 * 2 * NX + 2 * 2 * WVEC) + sizeof(float) * 4 * MT * 32 + 16;
    const TView& va = p.in.a;
    const TView& vb = p.in.b;
    // VEC staging addresses one batch element of an input with 32-bit byte offsets (raw buffers) and selects the input per
    // 16-channel chunk: planes of one element < 2 GB, a second input only behind a multiple of 16 channels
    const size_t plane = sizeof(float) * (size_t)va.Hs * va.Ws;
    const bool al16 = ((((uintptr_t)va.p) | ((uintptr_t)va.act) | ((uintptr_t)vb.p)) & 15) == 0;
    int vec = 0;
    if (va.mode == 0 && va.Ws % 4 == 0 && va.Wv == va.Ws && al16 &&
        plane * (size_t)(va.C > vb.C ? va.C : vb.C) < ((size_t)1 << 31) && (vb.C == 0 || va.C % 16 == 0))
        vec = 1;
    else if (va.mode == 2 && vb.C == 0 && va.C % 16 == 0 && va.Wv % 4 == 0 && va.Ws == 2 * va.Wv && va.Hs == 2 * va.Hv && al16 &&
             plane * (size_t)(va.C >> 2) < ((size_t)1 << 31))
        vec = 2;
    else if (va.mode == 1 && vb.C == 0 && va.Wv % 4 == 0 && va.Wv == 2 * va.Ws && va.Hv <= 2 * va.Hs && va.Ws % 2 == 0 && al16 &&
             plane * (size_t)va.C < ((size_t)1 << 31))
        vec = 3;
    auto k = va.act != nullptr ? (vec == 3 ? conv_fwd5_kernel<MT, true, 3> : vec == 2 ? conv_fwd5_kernel<MT, true, 2>
                                  : vec ? conv_fwd5_kernel<MT, true, 1> : conv_fwd5_kernel<MT, true, 0>)
                               : (vec == 3 ? conv_fwd5_kernel<MT, false, 3> : vec == 2 ? conv_fwd5_kernel<MT, false, 2>
                                  : vec ? conv_fwd5_kernel<MT, false, 1> : conv_fwd5_kernel<MT, false, 0>);
    // reduced-term products (gemm modes 2 / 3): the 64-row m-block kernels on the vector-staged views; everything else keeps three terms
    const int nt = (MT == 2 && vec != 0 && !p.fmt) ? rvsr_gemm_terms() : 3;
    if constexpr (MT == 2) {
#define FWD5_NT(NTV)                                                                                                                   \
        k = va.act != nullptr ? (vec == 3 ? conv_fwd5_kernel<MT, true, 3, false, NTV> : vec == 2 ? conv_fwd5_kernel<MT, true, 2, false, NTV> \
                                                                                      : conv_fwd5_kernel<MT, true, 1, false, NTV>)      \
                              : (vec == 3 ? conv_fwd5_kernel<MT, false, 3, false, NTV> : vec == 2 ? conv_fwd5_kernel<MT, false, 2, false, NTV> \
                                                                                       : conv_fwd5_kernel<MT, false, 1, false, NTV>)
        if (nt == 2) { FWD5_NT(2); } else if (nt == 1) { FWD5_NT(1); }
#undef FWD5_NT
    }
    if (p.fmt) {   // f16 + fp8 images: only the vector-staged kernels without act' read them
        if constexpr (MT == 2) {
            if (vec != 1 || va.act != nullptr) FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: the f16 + fp8 format needs a plain 16-byte-aligned input view with W %% 4 == 0 and no act' tensor");
            k = conv_fwd5_kernel<MT, false, 1, false, 4>;
        } else {
            FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: the f16 + fp8 format needs more than 32 output channels");
        }
    }
    // tile shape: 8 x 64 (256-byte output runs, 16-byte stores, plain vector-staged view) when it wastes no more pixels than 16 x 32
    int th = 16, tw = 32;
    size_t lds_k = lds;
    if (MT != 2 && p.act == 3) return RVSR_ERR_UNSUPPORTED;   // (mask epilogue: 8 x 64 tile, 64-row m-blocks only)
    if constexpr (MT == 2) {
        constexpr bool wide_ok = true;   // (the 8 x 64 tile wherever it tiles the frame at least as well as 16 x 32)
        const long px_n = (long)((p.Hout + 15) / 16 * 16) * ((p.Wout + 31) / 32 * 32), px_w = (long)((p.Hout + 7) / 8 * 8) * ((p.Wout + 63) / 64 * 64);
        if (p.act == 3 && !(wide_ok && vec == 1 && p.vec4 && px_w <= px_n)) return RVSR_ERR_UNSUPPORTED;   // (mask epilogue: 8 x 64 tile only)
        if (wide_ok && vec == 1 && p.vec4 && px_w <= px_n) {
            k = va.act != nullptr ? conv_fwd5_kernel<MT, true, 1, true> : conv_fwd5_kernel<MT, false, 1, true>;
            if (p.fmt) k = conv_fwd5_kernel<MT, false, 1, true, 4>;
            if (nt == 2) k = va.act != nullptr ? conv_fwd5_kernel<MT, true, 1, true, 2> : conv_fwd5_kernel<MT, false, 1, true, 2>;
            if (nt == 1) k = va.act != nullptr ? conv_fwd5_kernel<MT, true, 1, true, 1> : conv_fwd5_kernel<MT, false, 1, true, 1>;
            th = 8; tw = 64;
            constexpr int NXW = 2 * 10 * 66;
            lds_k = (size_t)16 * (2 * 2 * NXW + 2 * 2 * WVEC) + sizeof(float) * 4 * MT * 32 + 16;
        }
    }
    if (set_lds(k, lds_k)) FAIL(RVSR_ERR_LAUNCH, "conv_fwd5: cannot reserve %zu B of LDS", lds_k);
    const int nty = (p.Hout + th - 1) / th;
    const long items = (long)((p.Wout + tw - 1) / tw) * nty * ((p.Co + MT * 32 - 1) / (MT * 32)) * p.B;
    dim3 grid((unsigned)(items < 256 ? items : 256), 1, 1);
    hipLaunchKernelGGL(k, grid, dim3(512), lds_k, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_fwd5 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

int rvsr_launch_conv_fwd2(ConvFwdParams p, int ksize, int stride, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int Ctot = p.in.a.C + p.in.b.C;
    int mt, ccg, nchunks, nmb;
    fwd2_geom(ksize, p.Co, Ctot, mt, ccg, nchunks, nmb);
    const size_t need = rvsr_conv_fwd2_workspace_bytes(ksize, p.Co, Ctot);
    if (!workspace || workspace_bytes < need) FAIL(RVSR_ERR_WORKSPACE, "conv2d: workspace %zu B < %zu B", workspace_bytes, need);
    if ((ksize == 1 || stride == 2) && p.in.a.mode == 0) {
        // conv_fwd2_kernel stages a plain view through raw buffer loads: 32-bit byte offsets inside one batch element (< 2 GB),
        // and a concat boundary on a chunk boundary; anything else goes back to the exact-f32 kernels of conv_kernels.hip
        const size_t span = sizeof(float) * (size_t)p.in.a.Hs * p.in.a.Ws * (size_t)(p.in.a.C > p.in.b.C ? p.in.a.C : p.in.b.C);
        if (span >= ((size_t)1 << 31) || (p.in.b.C != 0 && p.in.a.C % (16 * ccg) != 0)) return RVSR_ERR_UNSUPPORTED;
    }
    const int T = ksize * ksize;
    const size_t total = (size_t)nmb * nchunks * T * (2 * ccg) * (mt * 32);
    if (p.fmt && (!f16fp8_ok(ksize, stride, mt, p.w_mode) || rvsr_gemm_mode_now() == 1))
        FAIL(RVSR_ERR_UNSUPPORTED, "conv2d: the f16 + fp8 format (w_mode | 4) is for forward 3x3 / stride-1 convs with more than 32 output channels");
    if (!p.prepacked)
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.w, (bf16x8*)workspace, p.Co,
                           Ctot, T, mt * 32, ccg, nchunks, nmb, p.w_mode | (p.fmt ? 0x100 : 0));
    p.wpack = workspace;
    p.swz = rvsr_swizzle_enabled();
    // (the 16-byte-store epilogues address one batch element of the output / residual with 32-bit byte offsets in a 2 GB buffer view)
    p.vec4 = (p.Wout % 4 == 0) && ((((uintptr_t)p.out1) | ((uintptr_t)p.res)) & 15) == 0 && !p.ps && p.out2 == nullptr &&
             sizeof(float) * (size_t)p.Co * p.Hout * p.Wout < ((size_t)1 << 31);
#define DISPATCH2(KS, S, CCG)                                   \
    do {                                                        \
        if (mt == 1) return launch_fwd2<KS, S, 1, CCG>(p, st);  \
        if (mt == 2) return launch_fwd2<KS, S, 2, CCG>(p, st);  \
        return launch_fwd2<KS, S, 4, CCG>(p, st);               \
    } while (0)
    if (ksize == 3 && stride == 1) return mt == 1 ? launch_fwd5<1>(p, st) : launch_fwd5<2>(p, st);
    if (ksize == 3 && stride == 2) DISPATCH2(3, 2, 1);
    DISPATCH2(1, 1, 2);
#undef DISPATCH2
}

// Pre-packed weight images (include/realvsr_hip.h section 2b): the image rvsr_conv2d_forward would build in its workspace,
// written to caller-owned memory so that it can be reused until the weights change; `desc` (10 x int64, host) describes the image
// for rvsr_pack_weights_batched.
extern "C" size_t rvsr_conv2d_pack_weights(const float* weight, int C_in, int Co, int ksize, int w_mode, void* out, size_t out_bytes,
                                           long long* desc, void* stream) {
    int mt, ccg, nchunks, nmb;
    fwd2_geom(ksize, Co, C_in, mt, ccg, nchunks, nmb);
    const size_t need = rvsr_conv_fwd2_workspace_bytes(ksize, Co, C_in);
    if (!weight || !out || out_bytes < need) return 0;
    if ((w_mode & 4) && !f16fp8_ok(ksize, 1, mt, w_mode & 1)) return 0;   // (the caller vouches for stride 1)
    const int fmtflag = (w_mode & 4) ? 0x100 : 0;
    const int T = ksize * ksize;
    const size_t total = (size_t)nmb * nchunks * T * (2 * ccg) * (mt * 32);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, (bf16x8*)out, Co,
                       C_in, T, mt * 32, ccg, nchunks, nmb, (w_mode & 1) | fmtflag);
    if (desc) {
        desc[0] = (long long)(uintptr_t)weight; desc[1] = (long long)(uintptr_t)out;
        desc[2] = Co; desc[3] = C_in; desc[4] = T; desc[5] = mt * 32; desc[6] = ccg; desc[7] = nchunks; desc[8] = nmb; desc[9] = (w_mode & 1) | fmtflag;
    }
    return need;
}
// descs: device array of n PackDesc records (built from the 10 x int64 descriptors: two pointers + eight 32-bit fields)
extern "C" int rvsr_pack_weights_batched(const void* descs, int n, void* stream) {
    if (n <= 0) return RVSR_OK;
    if (!descs) FAIL(RVSR_ERR_BAD_ARG, "pack_weights_batched: null table");
    static_assert(sizeof(PackDesc) == 48, "PackDesc layout is part of the C ABI (two pointers + eight ints)");
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(16, (unsigned)n), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "pack_weights_batched launch: %s", hipGetErrorString(e));
    return RVSR_OK;
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
// NT: terms of the product (rvsr_common.h): 3 = hi*hi + hi*lo + lo*hi; 2 = without the lo part of the output gradient; 1 = hi*hi
template <int DXMASK, int NT>
__device__ __forceinline__ void wg2_row(const unsigned char* xs_hi, const unsigned char* xs_lo, int xoff, bf16x8 ah,
                                        bf16x8 al, f32x16* acc) {
    const u32x4 h0 = *reinterpret_cast<const u32x4*>(xs_hi + xoff), h1 = *reinterpret_cast<const u32x4*>(xs_hi + xoff + 16);
    constexpr int N = ((DXMASK >> 0) & 1) + ((DXMASK >> 1) & 1) + ((DXMASK >> 2) & 1);
    bf16x8 bh[3], bl[3];
    int t = 0;
    if (DXMASK & 1) { bh[t] = take8<3>(h0, h1); ++t; }
    if (DXMASK & 2) { bh[t] = take8<4>(h0, h1); ++t; }
    if (DXMASK & 4) { bh[t] = take8<5>(h0, h1); ++t; }
    if (NT >= 2) {
        const u32x4 l0 = *reinterpret_cast<const u32x4*>(xs_lo + xoff), l1 = *reinterpret_cast<const u32x4*>(xs_lo + xoff + 16);
        t = 0;
        if (DXMASK & 1) { bl[t] = take8<3>(l0, l1); ++t; }
        if (DXMASK & 2) { bl[t] = take8<4>(l0, l1); ++t; }
        if (DXMASK & 4) { bl[t] = take8<5>(l0, l1); ++t; }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = mfma_bf16(ah, bh[i], acc[i]);
    if (NT >= 2) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = mfma_bf16(ah, bl[i], acc[i]);
    }
    if (NT >= 3) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = mfma_bf16(al, bh[i], acc[i]);
    }
}

// The X-row ring of round 4 (tile rows fastest, the two rows shared with the upper neighbour kept in LDS) measured no gain and is compiled out;
// scratch builds: -DWGRAD2_RING=1 (and p.ring = 1 in conv_kernels.hip).
#ifndef WGRAD2_RING
#define WGRAD2_RING 0
#endif
template <bool ACT, int GMODE, int NT = 3>
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

    // ---- staging registers.  All global loads are unconditional 16-byte loads from clamped addresses (a load
    // inside a divergent `if` is waited for at the join); validity travels as flags and is applied on the way to
    // LDS.  The loads of tile t+1 are issued after the barrier that opens tile t's MFMA phase.
    constexpr int NGI = 2;                  // G items per thread: 64 o x 4 rows x 4 octets / 512
    constexpr int NGV = GMODE == 0 ? 2 : 4; // float4 per G item (8 virtual px; pixel-shuffled storage holds 16 floats)
    constexpr int NXI = 4;                  // X items per thread: 64 c x 6 rows x 5 octets = 1920 / 512 -> 3.75
    float4 gv[NGI][NGV], sv[ACT ? NGI : 1][NGV], xv[NXI][2];
    float bacc[NGI] = {0.f, 0.f};           // bias-gradient partial sums of this thread's two G items

    // Raw buffer addressing (see conv_fwd5_kernel): a thread's item has a tile-independent 32-bit byte offset inside one image of
    // its tensor, the tile contributes a uniform byte offset (SGPR), and a float4 outside the image -- or an item without a
    // channel -- gets an offset beyond the 2 GB view, for which the load returns zeros: no clamped addresses, no validity flags,
    // no selects on the way to LDS.  (The 64-bit address chains and selects were ~14 vector instructions per load, inside the
    // MFMA phase.)  rvsr_launch_conv_wgrad2 checks the spans and that a second input starts on a multiple of 64 channels.
    constexpr unsigned OOB = 0x80000000u;
    unsigned g_vo[NGI], x_vo[NXI];
    int g_row[NGI], g_gx8[NGI], g_o[NGI], g_lds[NGI];
#pragma unroll
    for (int i = 0; i < NGI; ++i) {
        const int it = tid + i * WG2_THREADS;
        const int q = it & 3, row = (it >> 2) & 3, ol = it >> 4;
        g_row[i] = row;
        g_gx8[i] = 8 * q;
        g_o[i] = mb * 64 + ol;
        g_lds[i] = ol * WG2_GP + row * 64 + q * 16;
        const int o = g_o[i];
        g_vo[i] = o >= p.Co ? OOB
                : 4u * (GMODE == 0 ? (unsigned)((o * H + row) * W + 8 * q)
                                   : (unsigned)((((o >> 2) * 2 * H) + 2 * row + ((o >> 1) & 1)) * (2 * W) + 16 * q));
    }
    const bool sec = c0 >= C1;              // (uniform) this workgroup's 64 channels come from the second input
    int x_row[NXI], x_gx[NXI], x_lds[NXI];   // (x_lds: LDS offset of the item WITHOUT its row term; the row slot is a ring, see below)
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int it_raw = tid + i * WG2_THREADS;
        const bool live = it_raw < 64 * 30;
        const int it = live ? it_raw : 0;
        const int cl = it / 30, rem = it - cl * 30;
        const int row = rem / 5, q = rem - row * 5;
        const int c = c0 + cl;
        x_row[i] = row - 1;
        x_gx[i] = 8 * q - 4;
        x_lds[i] = live ? cl * WG2_XP + q * 16 : -1;
        // relative to a view that starts one row and four pixels BEFORE the image (so that row -1 / column -4 are offset >= 0)
        x_vo[i] = live && c < Ctot ? 4u * (unsigned)(((sec ? c - C1 : c) * H + row) * W + 8 * q) : OOB;
    }
    const size_t img_g = (size_t)p.Co * H * W;  // elements per image of G (same count for the pixel-shuffled storage)
    const size_t img_x1 = (size_t)C1 * H * W, img_x2 = (size_t)(Ctot - C1) * H * W;

    // `part` < 0: everything at once (prologue).  Otherwise part 0..7 = one eighth of the tile's loads: the main loop
    // issues one part per k-step of the MFMA phase so the requests trickle out under the matrix work instead of as one
    // burst in front of it (the burst took ~5.5 K cycles to issue: the memory pipeline back-pressures).
    // parts 0-3: X item `part`; parts 4, 5: G item 0, 1; parts 6, 7: nothing.
    // Tile walk.  p.ring: tile ROWS fastest -- consecutive tiles of a workgroup are vertical neighbours, and of the six X rows a
    // 4-row tile needs (y0 - 1 .. y0 + 4) the first two are the last two of the tile above: they stay in LDS (the row slots form a
    // ring of six, rotated by four per tile) and only four rows are fetched -- X traffic 1.25x instead of 1.875x the tile's pixels,
    // 73 instead of 93 KB per tile (the kernel moves as fast as its loads issue, profiles/r03_notes.md).  `fresh`: first tile of a
    // column (or of the workgroup's range): all six rows are fetched.
    struct TilePos { int b, y0, x0; bool fresh; };
    auto tile_pos = [&](int tile) {
        TilePos t;
        t.b = tile / (p.nty * p.ntx);
        const int trem = tile - t.b * (p.nty * p.ntx);
        if (WGRAD2_RING && p.ring) {
            const int tx = trem / p.nty, ty = trem - tx * p.nty;
            t.y0 = ty * 4;
            t.x0 = tx * 32;
            t.fresh = ty == 0;
        } else {
            const int ty = trem / p.ntx;
            t.y0 = ty * 4;
            t.x0 = (trem - ty * p.ntx) * 32;
            t.fresh = true;
        }
        return t;
    };
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) {
        const f32x4v q = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)so, 0));
        return make_float4(q.x, q.y, q.z, q.w);
    };
    auto issue_loads = [&](const TilePos& tp, int part) {
        const int b = tp.b;
        int y0 = tp.y0, x0 = tp.x0;
        // opaque per call: otherwise the (now cheap) lane offsets of all eight slices are computed once per tile, ahead of the
        // k-step loop, and live across it (56 spilled registers in the <ACT, pixel-shuffle> instantiation)
        asm volatile("" : "+s"(y0), "+s"(x0));
        if (part < 0 || part >= 4) {
            const __amdgpu_buffer_rsrc_t g_rs = buf_view_2g(p.g.p + (size_t)b * img_g);
            const __amdgpu_buffer_rsrc_t s_rs = buf_view_2g(ACT ? p.g.act + (size_t)b * img_g : p.g.p);
            const unsigned so = GMODE == 0 ? 4u * (unsigned)(y0 * W + x0) : 4u * (unsigned)(2 * y0 * 2 * W + 2 * x0);
#pragma unroll
            for (int i = 0; i < NGI; ++i) {
                if (part >= 0 && part != 4 + i) continue;
                const int gx = x0 + g_gx8[i];
                const bool ok = y0 + g_row[i] < H;
#pragma unroll
                for (int kk = 0; kk < NGV; ++kk) {
                    const bool okk = ok && gx + (GMODE == 0 ? 4 : 2) * kk < W;
                    const unsigned vo = okk ? g_vo[i] + 16u * kk : OOB;
                    gv[i][kk] = ld4(g_rs, vo, so);
                    if (ACT) sv[i][kk] = ld4(s_rs, vo, so);
                }
            }
        }
        if (part < 0 || part < 4) {
            const float* ximg = sec ? p.x.b.p + (size_t)b * img_x2 : p.x.a.p + (size_t)b * img_x1;
            const __amdgpu_buffer_rsrc_t x_rs = buf_view_2g(ximg - (W + 4));
            const unsigned so = 4u * (unsigned)(y0 * W + x0);
#pragma unroll
            for (int i = 0; i < NXI; ++i) {
                if (part >= 0 && part != i) continue;
                const int gy = y0 + x_row[i], gx = x0 + x_gx[i];
                const bool ok = gy >= 0 && gy < H && (tp.fresh || x_row[i] >= 1);   // (rows y0 - 1, y0 of a non-fresh tile are in LDS already)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int gxx = gx + 4 * kk;
                    xv[i][kk] = ld4(x_rs, ok && gxx >= 0 && gxx < W ? x_vo[i] + 16u * kk : OOB, so);
                }
            }
        }
    };

    auto commit = [&](bool fresh, int rbase) {
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
            const int ol = g_o[i] - mb * 64;
            float v[8];
            if (GMODE == 0) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float4 a = gv[i][k];
                    if (ACT) {
                        const float4 s_ = sv[i][k];
                        a.x *= s_.x > 0.f ? 1.f : p.g.slope; a.y *= s_.y > 0.f ? 1.f : p.g.slope;
                        a.z *= s_.z > 0.f ? 1.f : p.g.slope; a.w *= s_.w > 0.f ? 1.f : p.g.slope;
                    }
                    v[4 * k + 0] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
                }
            } else {
                const int sx = (mb * 64 + ol) & 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 a = gv[i][k];
                    float e0 = sx ? a.y : a.x, e1 = sx ? a.w : a.z;
                    if (ACT) {
                        const float4 s_ = sv[i][k];
                        e0 *= (sx ? s_.y : s_.x) > 0.f ? 1.f : p.g.slope;
                        e1 *= (sx ? s_.w : s_.z) > 0.f ? 1.f : p.g.slope;
                    }
                    v[2 * k] = e0;
                    v[2 * k + 1] = e1;
                }
            }
            // bias gradient: a G item's output channel does not depend on the tile, so every thread keeps its own
            // running sums and the workgroup reduces them once after the tile loop (LDS float atomics here cost
            // ~117 LDS cycles per wave-instruction, twice per tile)
            if (do_bias) bacc[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            bf16x8 h8, l8;
            split8(v, h8, l8);
            *reinterpret_cast<bf16x8*>(gs_hi + g_lds[i]) = h8;
            if (NT >= 3) *reinterpret_cast<bf16x8*>(gs_lo + g_lds[i]) = l8;
        }
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            if (x_lds[i] < 0 || (!fresh && x_row[i] < 1)) continue;
            float v[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float4 a = xv[i][k];
                v[4 * k + 0] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
            }
            bf16x8 h8, l8;
            split8(v, h8, l8);
            int slot = x_row[i] + 1 + rbase;          // ring slot of the item's row
            slot = slot >= 6 ? slot - 6 : slot;
            const int dst = x_lds[i] + slot * 80;
            *reinterpret_cast<bf16x8*>(xs_hi + dst) = h8;
            if (NT >= 2) *reinterpret_cast<bf16x8*>(xs_lo + dst) = l8;
        }
    };

    // contiguous tile range per workgroup: consecutive tiles share halo rows/columns in L2
    const int ntiles = p.B * p.nty * p.ntx;
    const int per = (ntiles + p.P - 1) / p.P;
    const int t_begin = blockIdx.x * per, t_end = t_begin + per < ntiles ? t_begin + per : ntiles;
    TilePos cur_pos = tile_pos(t_begin < t_end ? t_begin : 0);
    cur_pos.fresh = true;                   // (the first tile of the range has no predecessor in LDS)
    if (t_begin < t_end) issue_loads(cur_pos, -1);
    int rbase = 0;                          // ring slot of the tile's row y0 - 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        TilePos next_pos = tile_pos(tile + 1 < t_end ? tile + 1 : tile);
        if (tile + 1 >= t_end) next_pos.fresh = true;   // (the re-read of the last tile: everything, nothing is committed)
        const int ti = tile - t_begin;
        rbase = cur_pos.fresh ? 0 : (rbase + 4 >= 6 ? rbase - 2 : rbase + 4);
        int roff[6];                        // LDS byte offset of tile row r (image row y0 - 1 + r)
#pragma unroll
        for (int r = 0; r < 6; ++r) roff[r] = (r + rbase >= 6 ? r + rbase - 6 : r + rbase) * 80;
        if (ti < 6) STAMP(200 + ti * 5);
        commit(cur_pos.fresh, rbase);
        if (ti < 6) STAMP(201 + ti * 5);
        __syncthreads();
        if (ti < 6) STAMP(202 + ti * 5);
        if (ti < 6) STAMP(203 + ti * 5);
        const bool more = tile + 1 < t_end;
        if (!m_live && more) issue_loads(next_pos, -1);
        if (m_live) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#ifdef RVSR_TIMELINE
                if (blockIdx.x == 77 && blockIdx.z == 0 && lane == 0 && ti == 3) rvsr_dbg[120 + wave * 10 + ks] = __builtin_amdgcn_s_memtime();
#endif
                if (GMODE != 2 || more) issue_loads(next_pos, ks);   // (unconditional for the plain view: after the last tile it re-reads that tile)
                __builtin_amdgcn_sched_barrier(0);
                const int row = ks >> 1, cb = (ks & 1) * 16 + 8 * hi;  // this lane's 8 pixels: row, cols cb..cb+7
                const int goff = (m * 32 + lo) * WG2_GP + row * 64 + cb * 2;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(gs_hi + goff);
                const bf16x8 al = NT >= 3 ? *reinterpret_cast<const bf16x8*>(gs_lo + goff) : ah;
                // X octet pair holding stored cols cb .. cb+15 (= image cols x0-4+cb ..); tap dx needs +3+dx
                const int xbase = (chalf * 32 + lo) * WG2_XP + cb * 2;
                if (!second) {  // taps (0,0) (0,1) (0,2) (1,0) (1,1)
                    wg2_row<7, NT>(xs_hi, xs_lo, xbase + roff[row + 0], ah, al, acc + 0);
                    wg2_row<3, NT>(xs_hi, xs_lo, xbase + roff[row + 1], ah, al, acc + 3);
                } else {        // taps (1,2) (2,0) (2,1) (2,2)
                    wg2_row<4, NT>(xs_hi, xs_lo, xbase + roff[row + 1], ah, al, acc + 0);
                    wg2_row<7, NT>(xs_hi, xs_lo, xbase + roff[row + 2], ah, al, acc + 1);
                }
            }
        }
        cur_pos = next_pos;
        if (ti < 6) STAMP(204 + ti * 5);
#ifdef RVSR_TIMELINE
        if (blockIdx.x == 77 && blockIdx.z == 0 && lane == 0 && ti == 3) rvsr_dbg[120 + wave * 10 + 8] = __builtin_amdgcn_s_memtime();
#endif
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
    if (do_bias) {
        // item i of thread t covers output channel (t + 512 i) >> 4: the 16 lanes of a row group share it
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
            float v = bacc[i];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if ((tid & 15) == 0) bsum[(tid + i * WG2_THREADS) >> 4] = v;   // one writer per channel
        }
        __syncthreads();
        if (tid < 64) {
            const int o = mb * 64 + tid;
            if (o < p.Co) p.bpart[(size_t)blockIdx.x * p.Co + o] = bsum[tid];
        }
    }
}

int rvsr_launch_conv_wgrad2(const ConvWgradParams& p, int gy, int gz, hipStream_t st) {
    const size_t lds = 2 * 64 * WG2_GP + 2 * 64 * WG2_XP + 64 * sizeof(float);
    const bool act = p.g.act != nullptr;
    auto k = p.g.mode == 0 ? (act ? conv_wgrad2_kernel<true, 0> : conv_wgrad2_kernel<false, 0>)
                           : (act ? conv_wgrad2_kernel<true, 2> : conv_wgrad2_kernel<false, 2>);
    const int nt = rvsr_gemm_terms();   // reduced-term products (gemm modes 2 / 3)
    if (nt == 2) k = p.g.mode == 0 ? (act ? conv_wgrad2_kernel<true, 0, 2> : conv_wgrad2_kernel<false, 0, 2>)
                                   : (act ? conv_wgrad2_kernel<true, 2, 2> : conv_wgrad2_kernel<false, 2, 2>);
    if (nt == 1) k = p.g.mode == 0 ? (act ? conv_wgrad2_kernel<true, 0, 1> : conv_wgrad2_kernel<false, 0, 1>)
                                   : (act ? conv_wgrad2_kernel<true, 2, 1> : conv_wgrad2_kernel<false, 2, 1>);
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad2: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(k, dim3(p.P, gy, gz), dim3(WG2_THREADS), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad2 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// ==========================================================================================
// Weight gradient of a 1x1 convolution on the bf16 matrix cores: a plain GEMM
//   gW[o][c] = sum_px G[o][px] * X[c][px],   K = pixels.
// Both MFMA operands want 8 consecutive K (= pixels) per lane, which is the NCHW memory order, so the fragments
// are loaded straight from global memory with 16-byte loads (no LDS): lane l of a wave reads pixels
// 8*(l>>5) .. +7 of row (l & 31).  One workgroup = 4 waves = a 64(o) x 64(c) block of gW, one 32x32 tile per wave,
// over a slice of the pixels (K split, partial sums reduced by rvsr_reduce_partials_kernel).  The kernel is
// L1-bandwidth-bound (every wave streams its G and X rows), ~10x faster than the exact-f32 LDS-staged kernel it
// replaces for the 320->64 / 64->64 fusion convs of TSA.
template <bool ACT>
__global__ __launch_bounds__(256, 4) void conv_wgrad1x1_kernel(const ConvWgradParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int m = wave & 1, n = wave >> 1;
    const int C1 = p.x.a.C, Ctot = C1 + p.x.b.C;
    const int HW = p.Hout * p.Wout;
    const int o = blockIdx.y * 64 + m * 32 + lo;   // this lane's G row
    const int c = blockIdx.z * 64 + n * 32 + lo;   // this lane's X row
    const bool o_ok = o < p.Co, c_ok = c < Ctot;
    const bool second = c_ok && c >= C1;
    const float* xrow0 = second ? p.x.b.p : p.x.a.p;
    const int Cb = second ? p.x.b.C : C1, cl = c_ok ? (second ? c - C1 : c) : 0;
    const int oc = o_ok ? o : 0;
    const int U = (HW + 31) / 32;  // 32-pixel units per image
    const long units = (long)p.B * U;
    const long per = (units + p.P - 1) / p.P;
    long u0 = (long)blockIdx.x * per, u1 = u0 + per;
    if (u1 > units) u1 = units;
    f32x16 acc = zero16();
    float bsum = 0.f;
    for (long u = u0; u < u1; ++u) {
        const int b = (int)(u / U), px0 = (int)(u - (long)b * U) * 32;
        const float* gp = p.g.p + ((size_t)b * p.Co + oc) * HW;
        const float* ap = ACT ? p.g.act + ((size_t)b * p.Co + oc) * HW : nullptr;
        const float* xp = xrow0 + ((size_t)b * Cb + cl) * HW;
        float4 g4[4], a4[ACT ? 4 : 1], x4[4];
        bool pv[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int px = px0 + 16 * s2 + 8 * hi;
            pv[s2] = px < HW;  // HW % 8 == 0: the 8-pixel group is entirely inside or outside
            const int pc = pv[s2] ? px : 0;
            g4[2 * s2] = *reinterpret_cast<const float4*>(gp + pc);
            g4[2 * s2 + 1] = *reinterpret_cast<const float4*>(gp + pc + 4);
            x4[2 * s2] = *reinterpret_cast<const float4*>(xp + pc);
            x4[2 * s2 + 1] = *reinterpret_cast<const float4*>(xp + pc + 4);
            if (ACT) {
                a4[2 * s2] = *reinterpret_cast<const float4*>(ap + pc);
                a4[2 * s2 + 1] = *reinterpret_cast<const float4*>(ap + pc + 4);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float gv[8] = {g4[2 * s2].x, g4[2 * s2].y, g4[2 * s2].z, g4[2 * s2].w,
                           g4[2 * s2 + 1].x, g4[2 * s2 + 1].y, g4[2 * s2 + 1].z, g4[2 * s2 + 1].w};
            float xv[8] = {x4[2 * s2].x, x4[2 * s2].y, x4[2 * s2].z, x4[2 * s2].w,
                           x4[2 * s2 + 1].x, x4[2 * s2 + 1].y, x4[2 * s2 + 1].z, x4[2 * s2 + 1].w};
            if (ACT) {
                const float av[8] = {a4[2 * s2].x, a4[2 * s2].y, a4[2 * s2].z, a4[2 * s2].w,
                                     a4[2 * s2 + 1].x, a4[2 * s2 + 1].y, a4[2 * s2 + 1].z, a4[2 * s2 + 1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[j] *= av[j] > 0.f ? 1.f : p.g.slope;
            }
            const bool gok = pv[s2] && o_ok, xok = pv[s2] && c_ok;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gv[j] = gok ? gv[j] : 0.f;
                xv[j] = xok ? xv[j] : 0.f;
                bsum += gv[j];
            }
            bf16x8 gh, gl, xh, xl;
            split8(gv, gh, gl);
            split8(xv, xh, xl);
            acc = mfma_bf16(gh, xh, acc);
            acc = mfma_bf16(gh, xl, acc);
            acc = mfma_bf16(gl, xh, acc);
        }
    }
    float* part = p.part + (size_t)blockIdx.x * p.Co * Ctot;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int orow = blockIdx.y * 64 + m * 32 + drow(r, hi);
        if (orow < p.Co && c_ok) part[(size_t)orow * Ctot + c] = acc[r];
    }
    if (p.bpart != nullptr && blockIdx.z == 0 && n == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (hi == 0 && o_ok) p.bpart[(size_t)blockIdx.x * p.Co + o] = bsum;
    }
}

int rvsr_launch_conv_wgrad1x1(const ConvWgradParams& p, int gy, int gz, hipStream_t st) {
    auto k = p.g.act != nullptr ? conv_wgrad1x1_kernel<true> : conv_wgrad1x1_kernel<false>;
    hipLaunchKernelGGL(k, dim3(p.P, gy, gz), dim3(256), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad1x1 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// ==========================================================================================
// Weight gradient of a 3x3 STRIDE-2 convolution (pad 1) on the bf16 matrix cores, direct from global memory like
// conv_wgrad1x1_kernel:   gW[o][c][dy][dx] = sum_{b, py, px} G[o][py][px] * X[c][2 py + dy - 1][2 px + dx - 1].
// K = output pixels; lane (row, k-octet) holds 8 consecutive output pixels of one output row.  For X that is a
// stride-2 walk over input columns 2 px0 + dx - 1 + 2 i: the lane loads the 20 input values of columns
// 2 px0 - 4 .. 2 px0 + 15 of an input row with five aligned 16-byte loads and picks the three dx variants out of
// registers (static indices, no shuffles); the three dy variants are three input rows.  One workgroup = 4 waves =
// a 64(o) x 64(c) block of all 9 taps (9 accumulator tiles per wave) over a slice of the pixels; deterministic
// partial sums.  Replaces the exact-f32 LDS-staged fallback (2.5 ms/step for the two pyramid convs of config 2).
template <bool ACT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_s2_kernel(const ConvWgradParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int m = wave & 1, n = wave >> 1;
    const int C = p.x.a.C;                       // single input (no concat on this path)
    const int Hin = p.x.a.Hs, Win = p.x.a.Ws;
    const int Ho = p.Hout, Wo = p.Wout;
    const int o = blockIdx.y * 64 + m * 32 + lo;   // this lane's G row
    const int c = blockIdx.z * 64 + n * 32 + lo;   // this lane's X row
    const bool o_ok = o < p.Co, c_ok = c < C;
    const int oc = o_ok ? o : 0, cc = c_ok ? c : 0;
    const int UW = (Wo + 15) / 16;                 // 16-pixel units per output row
    const long units = (long)p.B * Ho * UW;
    const long per = (units + p.P - 1) / p.P;
    long u0 = (long)blockIdx.x * per, u1 = u0 + per;
    if (u1 > units) u1 = units;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = zero16();
    float bsum = 0.f;
    for (long u = u0; u < u1; ++u) {
        const int b = (int)(u / ((long)Ho * UW));
        const int rem = (int)(u - (long)b * Ho * UW);
        const int py = rem / UW, px0 = (rem - py * UW) * 16 + 8 * hi;   // this lane's 8 output pixels: px0 .. px0+7
        const bool pv = px0 < Wo;                                        // Wo % 8 == 0: all 8 or none
        // ---- G (and act') : two 16-byte loads
        const size_t gidx = (((size_t)b * p.Co + oc) * Ho + py) * Wo + (pv ? px0 : 0);
        const float4 g0 = *reinterpret_cast<const float4*>(p.g.p + gidx), g1 = *reinterpret_cast<const float4*>(p.g.p + gidx + 4);
        float4 a0, a1;
        if (ACT) {
            a0 = *reinterpret_cast<const float4*>(p.g.act + gidx);
            a1 = *reinterpret_cast<const float4*>(p.g.act + gidx + 4);
        }
        // ---- X: three input rows x five 16-byte groups (columns 2 px0 - 4 + 4 j .. +3)
        float xr[3][20];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * py + dy - 1;
            const bool rok = pv && c_ok && iy >= 0 && iy < Hin;
            const float* row = p.x.a.p + (((size_t)b * C + cc) * Hin + (rok ? iy : 0)) * Win;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int col = 2 * px0 - 4 + 4 * j;
                const bool ok = rok && col >= 0 && col < Win;   // Win % 4 == 0: the group is entirely in or out
                const float4 q = *reinterpret_cast<const float4*>(row + (ok ? col : 0));
                xr[dy][4 * j + 0] = ok ? q.x : 0.f; xr[dy][4 * j + 1] = ok ? q.y : 0.f;
                xr[dy][4 * j + 2] = ok ? q.z : 0.f; xr[dy][4 * j + 3] = ok ? q.w : 0.f;
            }
        }
        float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        if (ACT) {
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] *= av[j] > 0.f ? 1.f : p.g.slope;
        }
        const bool gok = pv && o_ok;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gv[j] = gok ? gv[j] : 0.f;
            bsum += gv[j];
        }
        bf16x8 gh, gl;
        split8(gv, gh, gl);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = xr[dy][3 + dx + 2 * i];   // column 2 (px0 + i) + dx - 1
                bf16x8 xh, xl;
                split8(xv, xh, xl);
                f32x16& a = acc[dy * 3 + dx];
                a = mfma_bf16(gh, xh, a);
                a = mfma_bf16(gh, xl, a);
                a = mfma_bf16(gl, xh, a);
            }
        }
    }
    float* part = p.part + (size_t)blockIdx.x * p.Co * C * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = blockIdx.y * 64 + m * 32 + drow(r, hi);
            if (orow < p.Co && c_ok) part[((size_t)orow * C + c) * 9 + t] = acc[t][r];
        }
    }
    if (p.bpart != nullptr && blockIdx.z == 0 && n == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (hi == 0 && o_ok) p.bpart[(size_t)blockIdx.x * p.Co + o] = bsum;
    }
}

int rvsr_launch_conv_wgrad_s2(const ConvWgradParams& p, int gy, int gz, hipStream_t st) {
    auto k = p.g.act != nullptr ? conv_wgrad_s2_kernel<true> : conv_wgrad_s2_kernel<false>;
    hipLaunchKernelGGL(k, dim3(p.P, gy, gz), dim3(256), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_wgrad_s2 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}
