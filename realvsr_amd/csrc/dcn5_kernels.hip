// dcn5_kernels.hip -- input / offset / mask gradient of the modulated DCN, fifth generation (gfx950).
//
// Replaces the reference's modulated_deformable_col2im + col2im_coord pair (kernel.cu:636-767, called from
// deform_conv_cuda.cpp:623-643) together with the `columns = W^T gOut` GEMM in front of them.
//
// What changed against dcn_bwdin4 (dcn_bwdin4.inc) comes from micro-benchmarks of the LDS on the MI355X (tools/micro/, results in
// profiles/r03_notes.md), cycles per wave instruction per CU with 8 waves issuing:
//     ds_add_u32 3.7    ds_add_u64 5.6    ds_add_f64 7.3    ds_add_f32 169 (!)    128-bit read-modify-write pair 15-23,
//     the same whether 16 or 64 lanes are active; with the address pattern of this scatter at sub-pixel offsets of random sign
//     (neighbouring lanes land on the same or the neighbouring cell): ds_add_u32 5.9, ds_add_u64 11.3, ds_add_f64 19.7.
//   The fourth generation scattered through per-wave PRIVATE f32 windows (two parity passes of four read-modify-writes per tap,
//   claim rounds for irregular offsets, an 8-window merge per chunk) because f32 LDS atomics are unusable; its tap phase was
//   LDS-instruction-bound (~220 LDS cycles per wave and tap) and its windows (82 KB + 20 KB of claim words) covered +-2 px.
//   * Here the grad_input tile of a workgroup is ONE shared window of 32-bit FIXED-POINT cells that every lane updates with
//     ds_add_u32: no ownership, no parity passes, no claims, no merge, any offset pattern, 16 atomics per (lane, tap) = ~95 LDS
//     cycles per (wave, tap).  Integer addition is associative: the in-tile sum is bit-reproducible whatever the order.
//     Scale: a contribution is w_corner * mask * col_grad with w, mask in [0, 1] and |col_grad[t, c, px]| = |<W[:, c, t], gOut[:, px]>|
//     <= ||W[:, c, t]||_2 * ||gOut[:, px]||_2 (Cauchy-Schwarz); with Wn = the largest column norm of the chunk (pack kernel) and Gn = the
//     largest pixel norm of the tile (prologue), S = 2^31 / (2304 * Wn * Gn) maps every contribution to |q| <= 2^19.8 and a cell
//     can receive at most 8 * 32 * 9 = 2304 of them per chunk (each (pixel, tap) at most once): NO overflow for any input.
//     One unit is ~1e-6 of the bound, round-to-nearest (magic-number add), so the quantisation noise of a cell (~6 units at 36
//     contributions) stays below the bf16x3 error of col_grad itself; measured against the f64 oracle in tests/test_gpu_dcn*.py.
//   * 4 B per (cell, channel) instead of 8 overlapping f32 windows: the halo R is a template parameter, R = 2 / 5 / 8 / 12 px
//     around the 8 x 32 pixel tile (19 / 30 / 44 / 66 KB), selected on the device from the offsets (two-counter probe).
//     Samples beyond the tile keep the global gather / atomic path.
//   * Co > 64 runs in ONE pass: the K loop of col_grad = W^T gOut covers all output channels (NK = 8 k-steps: 64 VGPRs of
//     gOut fragments, a 48 KB weight block per chunk), so sampling, scatter, flush and the offset / mask stores are done once
//     (the fourth generation repeated the whole kernel per 64 output channels).
//   * the next chunk's offsets, weight DMA and x-tile loads are issued before the current chunk's window is flushed; the
//     partner-lane sums of grad_offset / grad_mask use v_permlane32_swap (VALU) instead of ds_bpermute, which queued behind
//     the wave's own atomics.
// Layout of the work is unchanged: workgroup = 8 waves = 8 rows x 32 pixels, K chunk = 8 input channels, M tile = 4 taps x 8
// channels, so lane (pixel, half) owns channels 4*half .. 4*half+3 of four taps per M tile, straight from the accumulators.
#include "dcn_tile.h"

#ifdef RVSR_TIMELINE_DCN5   // s_memtime stamps of one wave of one workgroup (tools/dcn5_timeline.py)
__device__ unsigned long long rvsr_dbg_dcn5[256];
extern "C" int rvsr_debug_read_dcn5(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rvsr_dbg_dcn5), sizeof(unsigned long long) * 256); }
#define TS5(i) do { if (blockIdx.x == 77 && blockIdx.z == 1 && threadIdx.x == 192) rvsr_dbg_dcn5[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TS5(i) do {} while (0)
#endif
#ifndef RVSR_ABL5
#define RVSR_ABL5 0   // scratch ablation builds (tools/build_variant5.sh): bit mask of deleted ingredients, results wrong by construction
#endif

struct DcnBwdIn5Params {
    DcnGeom d;
    TView g;            // grad_output view (Co, Ho, Wo), optional fused act'
    float* gx;          // (B, C, H, W): accumulated into (zero or a partial gradient on entry)
    float* goff;
    float* gmask;
    size_t goff_bs, gmask_bs;
    DcnHaloSel sel;     // kernel selection on the device (dcn_common.h): every candidate halo is launched, one runs
    const float* wnorm;            // [chunk]: max over the chunk's 72 (tap, channel) columns of ||W[:, c, tap]||_2
};

// Wn[chunk] for the fixed-point scale: one block per chunk, 576 threads = 72 (tap, channel) columns x 8 slices of the output
// channels; slice sums combined through LDS, then the maximum over the columns
__global__ __launch_bounds__(576) void dcn_bwd5_wnorm_kernel(const float* __restrict__ w, float* __restrict__ wn, int Co, int C) {
    __shared__ float red[576];
    const int chunk = blockIdx.x, t = threadIdx.x, col = t % 72, sl = t / 72;
    const int tap = col >> 3, c = 8 * chunk + (col & 7);
    float s = 0.f;
    if (c < C)
        for (int o = sl; o < Co; o += 8) {
            const float v = w[((size_t)o * C + c) * 9 + tap];
            s += v * v;
        }
    red[t] = s;
    __syncthreads();
    if (t < 72) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += red[t + 72 * k];
        red[t] = a;
    }
    __syncthreads();
    if (t < 64) red[t] = fmaxf(red[t], t + 64 < 72 ? red[t + 64] : 0.f);
    __syncthreads();
    for (int k = 32; k > 0; k >>= 1) {
        if (t < k) red[t] = fmaxf(red[t], red[t + k]);
        __syncthreads();
    }
    if (t == 0) wn[chunk] = sqrtf(red[0]);
}

// packed[chunk][mt (3)][part (hi, lo)][o-octet (2*NK)][row (32)][8 o];  row -> tap = 4*mt + (row >> 3), c = 8*chunk + (row & 7)
template <int NK>
__global__ void pack_weights_bwd5_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int C, int nchunks) {
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
            const int o = 8 * ooct + j;
            v[j] = (tap < 9 && c < C && o < Co) ? w[((size_t)o * C + c) * 9 + tap] : 0.f;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const size_t blk = ((size_t)chunk * 3 + mt) * 2, per = (size_t)(2 * NK) * 32;
        packed[blk * per + ooct * 32 + row] = hi;
        packed[(blk + 1) * per + ooct * 32 + row] = lo;
    }
}

// Sampled statistic behind the halo selection of both DCN directions: every 16th row of every offset plane;
// cnt[0..5] = components with |v| > 2.5 / 3.5 / 5.5 / 7.5 / 8.5 / 11.5 px (DcnHaloSel).
__global__ void dcn_offset_probe2_kernel(const float* __restrict__ off, size_t off_bs, int B, int planes, int Ho, int Wo,
                                         unsigned* __restrict__ cnt) {
    const int nrow = (Ho + 15) / 16;
    const size_t total = (size_t)B * planes * nrow * Wo;
    const float lim[DCN_PROBE_COUNTERS] = {2.5f, 3.5f, 5.5f, 7.5f, 8.5f, 11.5f};
    unsigned n[DCN_PROBE_COUNTERS] = {0, 0, 0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo);
        size_t r = i / Wo;
        const int row = (int)(r % nrow);
        r /= nrow;
        const int pl = (int)(r % planes), b = (int)(r / planes);
        const int y = row * 16 + 8 < Ho ? row * 16 + 8 : Ho - 1;
        const float v = fabsf(off[(size_t)b * off_bs + ((size_t)pl * Ho + y) * Wo + x]);
#pragma unroll
        for (int k = 0; k < DCN_PROBE_COUNTERS; ++k) n[k] += v > lim[k] ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < DCN_PROBE_COUNTERS; ++k)
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) n[k] += __shfl_xor(n[k], s);
    // one atomic per counter and BLOCK (per wave, 2 K blocks x 4 waves on the same few addresses took 226 us at 3 px offsets)
    __shared__ unsigned part[DCN_PROBE_COUNTERS][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < DCN_PROBE_COUNTERS; ++k) part[k][wv] = n[k];
    }
    __syncthreads();
    if (threadIdx.x < DCN_PROBE_COUNTERS) {
        const unsigned t = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        if (t) atomicAdd(cnt + threadIdx.x, t);
    }
}

__device__ __forceinline__ void lds_add_i32(int* p, int v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u32
}
// a + b of the two lane halves in every lane (v_permlane32_swap: VALU, no LDS queue)
__device__ __forceinline__ float half_sum(float v) {
    // v_permlane32_swap a, b: lanes 32..63 of a <-> lanes 0..31 of b; with a = b = v every lane then holds both halves' values.
    // Inline assembly: this hipcc lowers the second result of __builtin_amdgcn_permlane32_swap to a copy of the first
    // (`v_add_f32 v2, v1, v1` in the ISA).  The s_nop covers the VALU-write -> permlane-read distance by hand.
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
    return a + b;
}
// fixed-point units of a contribution: round-to-nearest-even of w * ts through the 1.5 * 2^23 magic number (|w * ts| < 2^22)
__device__ __forceinline__ int fx_units(float w, float ts) {
    return (int)(__builtin_bit_cast(unsigned, __builtin_fmaf(w, ts, 12582912.f)) - 0x4B400000u);
}

// TERMS: terms of the bf16 product W^T * gOut (rvsr_common.h: gemm modes): 3 = hi*hi + hi*lo + lo*hi; 2 = without the weights' lo part;
// 1 = hi*hi
template <int NK, int R, int TERMS = 3>
__global__ __launch_bounds__(512, 2) void dcn_bwdin5_kernel(const DcnBwdIn5Params p, const bf16x8* __restrict__ wpack) {
    constexpr int TH = 8, NT = TH * 64;
    constexpr int TR = TH + 2 * R + 3, TC = 32 + 2 * R + 3, NPOS = TR * TC;
    constexpr int WBLK = 2 * (2 * NK) * 32;                        // vectors per (chunk, M tile): hi + lo
    constexpr int NXI = (2 * NPOS + NT - 1) / NT;                  // x-tile items (float4 of one position and quad) per thread
    constexpr int NWV = (3 * WBLK + NT - 1) / NT;                  // weight vectors per thread
    static_assert((3 * WBLK) % 64 == 0, "whole waves of weight vectors");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);              // [2 quads][NPOS], zero outside the image
    int* gwin = reinterpret_cast<int*>(xt + 2 * NPOS);             // [8 channels][NPOS]: the grad_input tile of this chunk, fixed point
    bf16x8* wsb = reinterpret_cast<bf16x8*>(gwin + 8 * NPOS);      // [3][WBLK]
    float* gn_red = reinterpret_cast<float*>(wsb + 3 * WBLK);      // [8 waves]: largest ||gOut[:, px]||^2 of the wave's row
    if (dcn_halo_not_selected(p.sel)) return;   // (uniform) not the halo the offsets of this call ask for
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    unsigned sbx, sby, sbz;
    swizzled_block(sbx, sby, sbz, d.swz);
    const int tx = sbx % d.ntx, ty = sbx / d.ntx;
    const int x0 = tx * 32, y0 = ty * TH, b = sbz;
    const int ty0 = y0 - d.pad - R, tx0 = x0 - d.pad - R;         // image coordinates of tile cell (0, 0); stride 1
    const int nchunks = (d.C + 7) / 8;
    const unsigned HW = (unsigned)(d.H * d.W);
    const size_t hw = (size_t)d.Ho * d.Wo;
    const int oy = y0 + wave, ox = x0 + lo;
    const bool px_ok = oy < d.Ho && ox < d.Wo;
    const size_t pix = (size_t)oy * d.Wo + ox;
    const float4* xq = xt + hi * NPOS;
    int* gq = gwin + (4 * hi) * NPOS;
    // raw buffer views of this batch element (32-bit byte offsets: the launcher checks that each spans < 2 GB)
    const __amdgpu_buffer_rsrc_t x_rs = buf_view_2g(d.x + (size_t)b * d.C * HW), gx_rs = buf_view(p.gx + (size_t)b * d.C * HW);
    const __amdgpu_buffer_rsrc_t off_rs = buf_view(d.offset + (size_t)b * d.off_bs), msk_rs = buf_view(d.mask + (size_t)b * d.mask_bs);
    const __amdgpu_buffer_rsrc_t goff_rs = buf_view(p.goff + (size_t)b * p.goff_bs), gmsk_rs = buf_view(p.gmask + (size_t)b * p.gmask_bs);
    const unsigned pl4 = 4u * (unsigned)hw, HW4 = 4u * HW;        // bytes per offset / mask plane, per x plane
    const unsigned pix4 = px_ok ? 4u * (unsigned)pix : 0u;
    const float by = (float)(oy - d.pad), bx = (float)(ox - d.pad);   // image coordinates of tap (0, 0) at zero offset

    // x-tile items of this thread, once per tile: item = (quad, row, col); a position outside the image (or no item) gets a
    // lane offset beyond the 2 GB view, for which the buffer load returns 0 -- the zero padding costs no clamp and no select
    unsigned xoff[NXI];
#pragma unroll
    for (int k = 0; k < NXI; ++k) {
        const int it = tid + k * NT;
        const int quad = it >= NPOS ? 1 : 0, pos = it - quad * NPOS;
        const int rr = pos / TC, ss = pos - rr * TC;
        const int gy = ty0 + rr, gx = tx0 + ss;
        const bool ok = it < 2 * NPOS && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        xoff[k] = ok ? 4u * ((unsigned)(gy * d.W + gx) + (unsigned)(4 * quad) * HW) : 0x80000000u;
    }

    // gOut (x act') of this lane's pixel as bf16 hi / lo MFMA fragments: K = output channels, NK k-steps of 16
    bf16x8 gh[NK], gl[NK];
    float gsq = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int ol0 = 8 * (2 * ks + hi);
        float v[8];
        if (p.g.mode == 0) {  // (uniform)
            tview_get_plain<8>(p.g, b, ol0, oy, ox, v);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tview_get(p.g, b, ol0 + j, oy, ox);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (px_ok && ol0 + j < d.Co) ? v[j] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) gsq = __builtin_fmaf(v[j], v[j], gsq);
        split8(v, gh[ks], gl[ks]);
    }
    {   // Gn^2 = the largest squared pixel norm of the tile (read back after the first barrier)
        gsq = half_sum(gsq);
#pragma unroll
        for (int sft = 16; sft > 0; sft >>= 1) gsq = fmaxf(gsq, __shfl_xor(gsq, sft));
        if (lane == 0) gn_red[wave] = gsq;
    }
    for (int e = tid; e < 8 * NPOS; e += NT) gwin[e] = 0;

    float o_dy[9], o_dx[9], o_m[9];
    float xv[NXI][4];
    // requests of a chunk: offsets / masks of this lane's pixel (27 loads), the weight blocks (LDS-DMA: lane l of a wave lands at
    // M0 + 16 l, no registers), the x tile (registers; committed to LDS once the previous chunk's taps are done)
    auto request = [&](int chunk) {
        const int c0 = chunk * 8, g = c0 / d.cpg;
        const unsigned ob = (unsigned)(g * 18) * pl4, mb_ = (unsigned)(g * 9) * pl4;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            o_dy[t9] = buf_load(off_rs, pix4, ob + (unsigned)(2 * t9) * pl4);
            o_dx[t9] = buf_load(off_rs, pix4, ob + (unsigned)(2 * t9 + 1) * pl4);
            o_m[t9] = buf_load(msk_rs, pix4, mb_ + (unsigned)t9 * pl4);
        }
        const bf16x8* src = wpack + (size_t)chunk * 3 * WBLK;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = tid + i * NT;
            if (e - lane + 63 < 3 * WBLK)   // (wave-uniform)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + e),
                                                 (__attribute__((address_space(3))) void*)(wsb + e), 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NXI; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[k][e] = buf_load(x_rs, xoff[k], (unsigned)(c0 + e) * HW4);   // (C % 8 == 0)
    };
    TS5(0);
    request(0);
    TS5(1);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * 8;
        const int g = c0 / d.cpg;
        if (chunk < 4) TS5(10 + 8 * chunk);
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int it = tid + k * NT;
            if (it < 2 * NPOS) xt[it] = make_float4(xv[k][0], xv[k][1], xv[k][2], xv[k][3]);
        }
        if (chunk < 4) TS5(11 + 8 * chunk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the weight DMA has landed
        if (chunk < 4) TS5(12 + 8 * chunk);
        __syncthreads();
        if (chunk < 4) TS5(13 + 8 * chunk);

        // fixed-point scale of this chunk (see the header): |contribution| * S <= 0.995 * 2^31 / 2304
        float S, invS;
        {
            float g2 = gn_red[0];
#pragma unroll
            for (int k = 1; k < TH; ++k) g2 = fmaxf(g2, gn_red[k]);
            const float bound = 1.002f * p.wnorm[chunk] * sqrtf(g2);
            S = bound > 0.f ? 927407.f / bound : 0.f;
            invS = bound > 0.f ? bound * (1.f / 927407.f) : 0.f;
        }
        const int cq = c0 + 4 * hi;  // this lane's first channel
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            f32x16 acc = zero16();
            const bf16x8* wb_hi = wsb + mt * WBLK;
            const bf16x8* wb_lo = wb_hi + (2 * NK) * 32;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const bf16x8 ah = wb_hi[(2 * ks + hi) * 32 + lo], al = wb_lo[(2 * ks + hi) * 32 + lo];
                if (RVSR_ABL5 & 16) { acc[0] += (float)ah[0] * (float)gh[ks][0] + (float)al[1] * (float)gl[ks][1]; continue; }
                acc = ks == 0 ? mfma_bf16_first(ah, gh[0]) : mfma_bf16(ah, gh[ks], acc);   // (bf16x3.h: the untied first MFMA must keep its operands)
                if (TERMS >= 2) acc = mfma_bf16(ah, gl[ks], acc);
                if (TERMS >= 3) acc = mfma_bf16(al, gh[ks], acc);
            }
            if (chunk == 1) TS5(90 + mt);
#pragma unroll
            for (int tsel = 0; tsel < 4; ++tsel) {
                const int tap = 4 * mt + tsel;
                if (tap >= 9) continue;  // uniform
                if (chunk == 1) TS5(50 + 4 * tap);
                // lanes without a pixel: zero offset, zero mask (their col_grad is already 0: gOut was read as 0)
                const float dy = px_ok ? o_dy[tap] : 0.f, dx = px_ok ? o_dx[tap] : 0.f;
                float m = o_m[tap];
                if (d.mask_logit) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));
                m = px_ok ? m : 0.f;
                // sample position in IMAGE coordinates exactly as the reference forms it (kernel.cu:594-616, 722-737)
                const float y = (by + (float)(tap / 3)) + dy, x = (bx + (float)(tap % 3)) + dx;
                const float fy = floorf(y), fx = floorf(x);
                const int yi = (int)fy, xi = (int)fx;
                const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
                const int r0 = yi - ty0, s0 = xi - tx0;   // tile coordinates of the top-left corner
                const bool in_tile = (unsigned)r0 < (unsigned)(TR - 1) && (unsigned)s0 < (unsigned)(TC - 1);
                const bool inside = y > -1.f && x > -1.f && y < (float)d.H && x < (float)d.W;
                const bool far = (RVSR_ABL5 & 64) ? false : (!in_tile && inside && px_ok);   // beyond the halo: global gather / atomics with the full rule set
                const int pos0 = in_tile ? r0 * TC + s0 : 0;
                float4 a00 = xq[pos0], a01 = xq[pos0 + 1], a10 = xq[pos0 + TC], a11 = xq[pos0 + TC + 1];
                int i00 = 0, i01 = 0, i10 = 0, i11 = 0;
                float z00 = 1.f, z01 = 1.f, z10 = 1.f, z11 = 1.f;   // corner validity (far path only)
                if (far) {
                    const bool vy0 = yi >= 0, vy1 = yi + 1 <= d.H - 1, vx0 = xi >= 0, vx1 = xi + 1 <= d.W - 1;
                    const int cy0 = vy0 ? yi : 0, cy1 = vy1 ? yi + 1 : d.H - 1, cx0 = vx0 ? xi : 0, cx1 = vx1 ? xi + 1 : d.W - 1;
                    i00 = cy0 * d.W + cx0; i01 = cy0 * d.W + cx1; i10 = cy1 * d.W + cx0; i11 = cy1 * d.W + cx1;
                    z00 = (vy0 && vx0) ? 1.f : 0.f; z01 = (vy0 && vx1) ? 1.f : 0.f;
                    z10 = (vy1 && vx0) ? 1.f : 0.f; z11 = (vy1 && vx1) ? 1.f : 0.f;
                    const float* pl = d.x + (size_t)b * d.C * HW + (unsigned)cq * HW;
                    float u00[4], u01[4], u10[4], u11[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // 16 loads in flight together
                        const float* q = pl + (unsigned)e * HW;
                        u00[e] = q[i00]; u01[e] = q[i01]; u10[e] = q[i10]; u11[e] = q[i11];
                    }
                    a00 = make_float4(u00[0] * z00, u00[1] * z00, u00[2] * z00, u00[3] * z00);
                    a01 = make_float4(u01[0] * z01, u01[1] * z01, u01[2] * z01, u01[3] * z01);
                    a10 = make_float4(u10[0] * z10, u10[1] * z10, u10[2] * z10, u10[3] * z10);
                    a11 = make_float4(u11[0] * z11, u11[1] * z11, u11[2] * z11, u11[3] * z11);
                }
                const bool live = in_tile || far;   // anything else (outside the sampler's range, or no pixel) contributes nothing
                const float c00[4] = {a00.x, a00.y, a00.z, a00.w}, c01[4] = {a01.x, a01.y, a01.z, a01.w};
                const float c10[4] = {a10.x, a10.y, a10.z, a10.w}, c11[4] = {a11.x, a11.y, a11.z, a11.w};
                float gy_s = 0.f, gx_s = 0.f, gm_s = 0.f;
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float cgv = live ? acc[4 * tsel + e] : 0.f;
                    const float B = c01[e] - c00[e], Cc = c10[e] - c00[e], D = (c11[e] - c01[e]) - Cc;
                    const float dxv = B + ly * D, dyv = Cc + lx * D;
                    const float val = (c00[e] + ly * Cc) + lx * dxv;
                    gm_s += cgv * val;
                    t[e] = cgv * m;
                    gy_s += dyv * t[e];
                    gx_s += dxv * t[e];
                }
                const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
                if (chunk == 1) TS5(51 + 4 * tap);
                if (in_tile && !(RVSR_ABL5 & 2)) {   // ---- scatter: 16 LDS integer atomics into the shared window (cells outside the image are dropped by the flush)
                    int* q = gq + pos0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ts = t[e] * S;
                        lds_add_i32(q + e * NPOS, fx_units(w00, ts));
                        lds_add_i32(q + e * NPOS + 1, fx_units(w01, ts));
                        lds_add_i32(q + e * NPOS + TC, fx_units(w10, ts));
                        lds_add_i32(q + e * NPOS + TC + 1, fx_units(w11, ts));
                    }
                }
                if (chunk == 1) TS5(52 + 4 * tap);
                if (far) {
                    float* gp = p.gx + ((size_t)b * d.C + cq) * HW;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* q = gp + (size_t)e * HW;
                        if (z00 * w00 != 0.f) atomicAdd(q + i00, w00 * t[e]);
                        if (z01 * w01 != 0.f) atomicAdd(q + i01, w01 * t[e]);
                        if (z10 * w10 != 0.f) atomicAdd(q + i10, w10 * t[e]);
                        if (z11 * w11 != 0.f) atomicAdd(q + i11, w11 * t[e]);
                    }
                }
                if (!(RVSR_ABL5 & 4)) {
                gy_s = half_sum(gy_s);
                gx_s = half_sum(gx_s);
                gm_s = half_sum(gm_s);
                }
                if (px_ok && hi == 0 && !(RVSR_ABL5 & 8)) {
                    if (d.mask_logit) gm_s *= m * (1.f - m);
                    const unsigned go_ = (unsigned)(g * 18 + 2 * tap) * pl4, gk = (unsigned)(g * 9 + tap) * pl4;
                    if (c0 % d.cpg == 0) {
                        buf_store(goff_rs, pix4, go_, gy_s);
                        buf_store(goff_rs, pix4, go_ + pl4, gx_s);
                        buf_store(gmsk_rs, pix4, gk, gm_s);
                    } else {   // a later chunk of the same deformable group (cpg > 8)
                        buf_store(goff_rs, pix4, go_, buf_load(goff_rs, pix4, go_) + gy_s);
                        buf_store(goff_rs, pix4, go_ + pl4, buf_load(goff_rs, pix4, go_ + pl4) + gx_s);
                        buf_store(gmsk_rs, pix4, gk, buf_load(gmsk_rs, pix4, gk) + gm_s);
                    }
                }
            }
        }
        if (chunk < 4) TS5(14 + 8 * chunk);
        __syncthreads();
        if (chunk < 4) TS5(15 + 8 * chunk);
        if (chunk + 1 < nchunks) request(chunk + 1);   // (uniform) in flight while the window is flushed
        if (chunk < 4) TS5(16 + 8 * chunk);
        // ---- flush: wave w owns channel c0 + w; one global atomic per touched cell inside the image (the halos of
        // neighbouring workgroups overlap), cell back to zero for the next chunk
        {
            const unsigned cpl = (unsigned)(c0 + wave) * HW4;
            int* gc = gwin + wave * NPOS;
            const bool ch_ok = c0 + wave < d.C;
            for (int base = lane; base < ((RVSR_ABL5 & 32) ? 0 : NPOS); base += 256) {   // four cells per round trip
                int v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = base + 64 * j < NPOS ? gc[base + 64 * j] : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (v[j] != 0) {
                        const int pos = base + 64 * j;
                        gc[pos] = 0;
                        const int r = pos / TC, s = pos - r * TC;
                        const int yy = ty0 + r, xx = tx0 + s;
                        if (ch_ok && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W) {
                            const float f = (float)v[j] * invS;
                            if (RVSR_ABL5 & 1) buf_store(gx_rs, 4u * (unsigned)(yy * d.W + xx), cpl, f);
                            else buf_atomic_add(gx_rs, 4u * (unsigned)(yy * d.W + xx), cpl, f);
                        }
                    }
                }
            }
        }
        if (chunk < 4) TS5(17 + 8 * chunk);
        // (the barrier after the next commit orders this flush before the next chunk's atomics)
    }
}

// memset-free: the caller zeroes the three counters (a fresh torch.zeros in the Python layer, hipMemsetAsync in the backward's own path)
size_t rvsr_launch_dcn_offset_probe(const DcnGeom& d, unsigned* cnt, hipStream_t st) {
    const int oplanes = (d.C / d.cpg) * 18, nrow = (d.Ho + 15) / 16;
    const size_t nprobe = (size_t)d.B * oplanes * nrow * d.Wo;
    const unsigned nb = (unsigned)((nprobe + 2047) / 2048 < 2048 ? (nprobe + 2047) / 2048 : 2048);
    hipLaunchKernelGGL(dcn_offset_probe2_kernel, dim3(nb ? nb : 1), dim3(256), 0, st, d.offset, d.off_bs, d.B, oplanes, d.Ho, d.Wo, cnt);
    return nprobe;
}

static int nk5_of(int Co) { return Co <= 16 ? 1 : (Co <= 32 ? 2 : (Co <= 64 ? 4 : 8)); }
size_t rvsr_dcn_bwdin5_workspace_bytes(int Co, int C) {
    // weight image + per-chunk column norms + the probe's counters
    return (size_t)((C + 7) / 8) * 3 * 2 * (2 * nk5_of(Co)) * 32 * 16 + (((size_t)((C + 7) / 8) * 4 + 255) & ~(size_t)255) + 256;
}

template <int NK, int R>
static int launch_bwdin5(const DcnBwdIn5Params& p, const bf16x8* wpack, hipStream_t st) {
    constexpr int TH = 8, TR = TH + 2 * R + 3, TC = 32 + 2 * R + 3, NPOS = TR * TC;
    const size_t lds = (size_t)NPOS * (2 * 16 + 8 * 4) + (size_t)3 * 2 * (2 * NK) * 32 * 16 + 8 * sizeof(float);
    auto k = dcn_bwdin5_kernel<NK, R>;
    if constexpr (NK >= 4) {   // reduced-term products (gemm modes 2 / 3): the kernels of the nf64 / nf128 packs
        const int nt = rvsr_gemm_terms();
        if (nt == 2) k = dcn_bwdin5_kernel<NK, R, 2>;
        if (nt == 1) k = dcn_bwdin5_kernel<NK, R, 1>;
    }
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_bwdin5: cannot reserve %zu B of LDS", lds);
    const DcnGeom& d = p.d;
    dim3 grid(d.ntx * ((d.Ho + TH - 1) / TH), 1, d.B);
    hipLaunchKernelGGL(k, grid, dim3(TH * 64), lds, st, p, wpack);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_bwdin5 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

template <int NK>
static int launch_bwdin5_halo(const DcnBwdIn5Params& p, const bf16x8* wpack, int halo, hipStream_t st) {
    if (halo <= 2) return launch_bwdin5<NK, 2>(p, wpack, st);
    if constexpr (NK <= 4) if (halo <= 4) return launch_bwdin5<NK, 4>(p, wpack, st);   // (77 KB: the largest window that still fits twice per CU)
    if (halo <= 5) return launch_bwdin5<NK, 5>(p, wpack, st);
    if (halo <= 8) return launch_bwdin5<NK, 8>(p, wpack, st);
    if constexpr (NK <= 4) return launch_bwdin5<NK, 12>(p, wpack, st);   // (12 px + the 48 KB weight block of NK = 8 exceed 160 KB)
    return launch_bwdin5<NK, 8>(p, wpack, st);
}

// halo < 0: selected on the device from the offsets (probe + one launch per candidate halo, no host round trip)
int rvsr_launch_dcn_bwdin5(const DcnGeom& d, const float* weight, const TView& g, float* gx, float* goff, size_t goff_bs,
                           float* gmask, size_t gmask_bs, void* workspace, size_t workspace_bytes, hipStream_t st, int halo,
                           const unsigned* probe_in) {
    if (d.cpg % 8 != 0 || d.C % 8 != 0 || d.stride != 1 || d.dil != 1 || d.Co > 128) return RVSR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < rvsr_dcn_bwdin5_workspace_bytes(d.Co, d.C)) return RVSR_ERR_UNSUPPORTED;
    // 32-bit byte offsets into one batch element's planes (x through a 2 GB view: bit 31 marks the zero padding)
    const size_t planes = (size_t)(d.C / d.cpg) * 18 > (size_t)d.C ? (size_t)(d.C / d.cpg) * 18 : (size_t)d.C;
    if (planes * (size_t)d.H * d.W * sizeof(float) >= ((size_t)1 << 31) || planes * (size_t)d.Ho * d.Wo * sizeof(float) >= ((size_t)1 << 31))
        return RVSR_ERR_UNSUPPORTED;
    const int NK = nk5_of(d.Co), nchunks = (d.C + 7) / 8;
    const size_t wbytes = (size_t)nchunks * 3 * 2 * (2 * NK) * 32 * 16;
    bf16x8* wpack = (bf16x8*)workspace;
    float* wnorm = (float*)((unsigned char*)workspace + wbytes);
    unsigned* cnt = (unsigned*)((unsigned char*)workspace + wbytes + (((size_t)nchunks * 4 + 255) & ~(size_t)255));
    hipLaunchKernelGGL(dcn_bwd5_wnorm_kernel, dim3(nchunks), dim3(576), 0, st, weight, wnorm, d.Co, d.C);
    const size_t total = (size_t)nchunks * 3 * (2 * NK) * 32;
    const dim3 pg((unsigned)((total + 255) / 256)), pb(256);
    switch (NK) {
        case 1: hipLaunchKernelGGL(pack_weights_bwd5_kernel<1>, pg, pb, 0, st, weight, wpack, d.Co, d.C, nchunks); break;
        case 2: hipLaunchKernelGGL(pack_weights_bwd5_kernel<2>, pg, pb, 0, st, weight, wpack, d.Co, d.C, nchunks); break;
        case 4: hipLaunchKernelGGL(pack_weights_bwd5_kernel<4>, pg, pb, 0, st, weight, wpack, d.Co, d.C, nchunks); break;
        default: hipLaunchKernelGGL(pack_weights_bwd5_kernel<8>, pg, pb, 0, st, weight, wpack, d.Co, d.C, nchunks); break;
    }
    DcnBwdIn5Params p;
    p.d = d; p.g = g; p.gx = gx; p.goff = goff; p.gmask = gmask; p.goff_bs = goff_bs; p.gmask_bs = gmask_bs;
    p.sel = dcn_halo_always(); p.wnorm = wnorm;
#define BWDIN5_DISPATCH(HALO)                                                   \
    switch (NK) {                                                               \
        case 1: rc = launch_bwdin5_halo<1>(p, wpack, HALO, st); break;          \
        case 2: rc = launch_bwdin5_halo<2>(p, wpack, HALO, st); break;          \
        case 4: rc = launch_bwdin5_halo<4>(p, wpack, HALO, st); break;          \
        default: rc = launch_bwdin5_halo<8>(p, wpack, HALO, st); break;         \
    }
    int rc = RVSR_OK;
    if (halo >= 0) {
        BWDIN5_DISPATCH(halo);
        return rc;
    }
    const int oplanes = (d.C / d.cpg) * 18, nrow = (d.Ho + 15) / 16;
    const size_t nprobe = (size_t)d.B * oplanes * nrow * d.Wo;
    if (probe_in != nullptr) {
        cnt = const_cast<unsigned*>(probe_in);   // the forward of this layer already counted these offsets (rvsr_dcn_pack_forward's probe)
    } else {
        if (hipMemsetAsync(cnt, 0, DCN_PROBE_COUNTERS * sizeof(unsigned), st) != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn backward: memset of the probe counters failed");
        rvsr_launch_dcn_offset_probe(d, cnt, st);
    }
    // A sample beyond the halo costs 16 global gathers + 16 global atomics (~50x an in-tile sample), a larger halo costs
    // staging and flush work in proportion to its cells (585 / 945 / 1377 / 2065): switch up as soon as a few percent of the
    // offset components leave the smaller tile.
    const unsigned thr = (unsigned)(nprobe * (size_t)2 / 100) + 1;
    const bool has12 = NK <= 4;
    p.sel.probe = cnt;
    p.sel.thr_ge = p.sel.thr_lt = thr;
    // R = 2: few components beyond 2.5 px; R = 4 (round 4: two workgroups per CU like R = 2, where R = 5 fits once): else, few beyond
    // 3.5 px (hence fewer still beyond its 4.5); R = 5: else, few beyond 5.5; R = 8: else, few beyond 8.5 (or no larger tile); R = 12: the
    // rest.  The counters are monotone, so the chain is a partition.
    if (has12) {
        const int halos[5] = {2, 4, 5, 8, 12}, ge[5] = {-1, 0, 1, 2, 4}, lt[5] = {0, 1, 2, 4, -1};
        for (int k = 0; k < 5; ++k) {
            p.sel.ge = ge[k]; p.sel.lt = lt[k];
            BWDIN5_DISPATCH(halos[k]);
            if (rc != RVSR_OK) return rc;
        }
    } else {
        const int halos[4] = {2, 5, 8, 12}, ge[4] = {-1, 0, 2, 4}, lt[4] = {0, 2, has12 ? 4 : -1, -1};
        for (int k = 0; k < (has12 ? 4 : 3); ++k) {
            p.sel.ge = ge[k]; p.sel.lt = lt[k];
            BWDIN5_DISPATCH(halos[k]);
            if (rc != RVSR_OK) return rc;
        }
    }
#undef BWDIN5_DISPATCH
    return rc;
}
