// dcn6_kernels.hip -- the WHOLE backward of the modulated DCN in one kernel, sixth generation (gfx950).
//
// Replaces, for the shape the EDVR-M training step runs (3 x 3, stride 1, dilation 1, 8 channels per deformable group, <= 64 output
// channels), the pair dcn_bwdin5 (input / offset / mask gradient, dcn5_kernels.hip) + dcn_bwdw4 (weight / bias gradient,
// dcn_bwdw4.inc), i.e. the reference's modulated_deform_conv_cuda_backward (deform_conv_cuda.cpp:571-685) with all of its kernels
// (kernel.cu:636-767 col2im + col2im_coord, :571-633 the im2col recompute) and its three GEMMs (cpp:623-626, 659-671).
//
// Why (profiles/r04_*): the pair cost 7.4 ms per L1 launch against 1.2 ms of forward -- 1.6 G vector wave-instructions against the
// forward's 0.24 G.  Both kernels walked all 72 sampling geometries per pixel, both read x / offsets / masks / gOut, dcn_bwdin5 spent
// half of its vector work on things the lane layout forced (two lane halves computing one geometry, partner-lane sums), and the weight
// gradient rebuilt the column tile through 2-byte LDS stores.  Here:
//   * ONE sampling pass.  bilinear(x) is formed once per (pixel, tap, channel); grad_mask, grad_offset, the grad_input scatter AND
//     the column value col = mask * bilinear(x) of the weight gradient all come from it.
//   * The M rows of col_grad = W^T gOut are PERMUTED in the packed weight image so that the accumulator registers of lane (pixel, half)
//     hold all 8 channels of two taps (instead of 4 channels of four taps): a lane owns a whole (pixel, tap) -- one geometry per lane,
//     no partner-lane sums, and the two halves of a wave work on different taps (5 lane iterations per chunk instead of 9 taps).
//   * Packed f32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel broadcasts) on channel pairs.
//   * The weight gradient gW[o, (c, tap)] = sum_px gOut[o, px] col[(c, tap), px] needs both operands with PIXELS along the MFMA's K
//     (register) dimension, but the sampling pass produces pixels along the LANE dimension.  The transposition is done by the matrix
//     core itself: an MFMA with a 0/1 selector as B operand, D = A x [I16 | 0] (+ A' x [0 | I16]), moves A[i = pixel][k = channel] to
//     D[i = pixel][j = channel] whose register layout is (lane = j, register = i): exact (one non-zero product per output, f32
//     accumulate), no LDS round trip, no 2-byte stores.  The same trick turns the gOut fragments (lane = pixel, K = output channel)
//     into the A operand of the weight-gradient GEMM (lane = output channel, K = pixel).  The pixel order along K is whatever the D
//     layout makes it -- the same for both operands, which is all a dot product needs.
//   * CHUNK-MAJOR persistent schedule: a workgroup owns ONE 8-channel chunk (= one deformable group) and walks a contiguous range of
//     8 x 32 pixel tiles, so the 64 x 80 weight-gradient block of its chunk stays in registers for the whole launch (96 accumulator
//     registers per wave; deterministic partials at the end, reduced by rvsr_reduce_partials_kernel as before), the chunk's weight
//     block is fetched once per workgroup, and the grad_offset / grad_mask planes of the group are written exactly once.  The
//     workgroups of the nchunks chunks of one tile stream sit on the same XCD (linear workgroup id % 8), so the gOut tile they all
//     read comes out of that XCD's L2.
// The grad_input scatter is dcn_bwdin5's: one shared LDS window of 32-bit fixed-point cells, ds_add_u32 from every lane, scale from
// Cauchy-Schwarz norms (no overflow for any input), flushed with one f32 global atomic per touched cell.
#include "dcn_tile.h"

struct DcnBwd6Params {
    DcnGeom d;
    TView g;            // grad_output view (Co, Ho, Wo), plain, optional fused act'
    float* gx;          // (B, C, H, W): accumulated into
    float* goff;
    float* gmask;
    size_t goff_bs, gmask_bs;
    DcnHaloSel sel;
    const float* wnorm; // [chunk]: max over the chunk's 72 (tap, channel) columns of ||W[:, c, tap]||_2  (dcn_bwd5_wnorm_kernel)
    float* part;        // [ns][Co][C * 9] weight-gradient partials
    float* bpart;       // [ns][Co] bias-gradient partials (nullptr: not wanted)
    int ns;             // tile streams (= partials)
    int nty;            // tile rows of 8 output rows
    int ntiles;         // B * nty * ntx
};

// Lane iteration `it` (0..4) of lane half h works on tap:  it < 4: 4 (it >> 1) + 2 h + (it & 1);  it == 4: 8 for h = 0, none for h = 1.
__host__ __device__ __forceinline__ int bwd6_tap(int it, int h) { return it < 4 ? 4 * (it >> 1) + 2 * h + (it & 1) : (h == 0 && it == 4 ? 8 : -1); }

// packed[chunk][mt (3)][part (hi, lo)][o-octet (8)][row (32)][8 o].  Row i of M tile mt lands in accumulator register
// r = (i & 3) + 4 (i >> 3) of lane half h = (i >> 2) & 1 (D layout of the 32x32 MFMA); that register is to hold channel r & 7 of the tap of
// lane iteration it = 2 mt + (r >> 3).
__global__ void pack_weights_bwd6_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int C, int nchunks) {
    const size_t total = (size_t)nchunks * 3 * 8 * 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx & 31);
        size_t r = idx >> 5;
        const int ooct = (int)(r & 7);
        r >>= 3;
        const int mt = (int)(r % 3), chunk = (int)(r / 3);
        const int h = (row >> 2) & 1, reg = (row & 3) + 4 * (row >> 3);
        const int tap = bwd6_tap(2 * mt + (reg >> 3), h), c = 8 * chunk + (reg & 7);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = 8 * ooct + j;
            v[j] = (tap >= 0 && c < C && o < Co) ? w[((size_t)o * C + c) * 9 + tap] : 0.f;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const size_t blk = ((size_t)chunk * 3 + mt) * 2, per = (size_t)8 * 32;
        packed[blk * per + ooct * 32 + row] = hi;
        packed[(blk + 1) * per + ooct * 32 + row] = lo;
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// s.x * b + c  /  s.y * b + c  (op_sel picks the half of s that both result lanes read)
__device__ __forceinline__ f32x2 pk_fma_x(f32x2 s, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(s), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma_y(f32x2 s, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(s), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul_x(f32x2 s, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(s), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul_y(f32x2 s, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(s), "v"(b));
    return r;
}
__device__ __forceinline__ void lds_add_i32_6(int* p, int v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u32
}
// a + b of the two lane halves in every lane (see dcn5_kernels.hip: half_sum)
__device__ __forceinline__ float half_sum6(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
    return a + b;
}
// 8 f32 registers holding exact bf16 values -> one bf16x8 MFMA operand
__device__ __forceinline__ bf16x8 pack8_exact(const f32x16& d, int r0) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)d[r0 + e];
    return o;
}

// R: halo of the LDS x tile / grad_input window around the 8 x 32 pixel tile.  TERMS: terms of the bf16 products (rvsr_common.h: gemm modes):
// 3 = hi*hi + hi*lo + lo*hi; 2 = without the weights' lo part (col_grad) / the output gradient's lo part (weight gradient); 1 = hi*hi.
//
// FOUR waves per workgroup, ONE per SIMD, each with the whole 512-register budget, two pixel rows of the tile per wave (rows w and w + 4,
// one after the other): the 96 accumulator registers of the chunk's weight gradient exist once per SIMD instead of once per wave -- with
// eight waves of 256 registers the kernel spilled 90-170 registers (the first build of this file) -- and the transposed gOut operands stay in
// registers.  What a second wave per SIMD would have hidden is hidden by the wave's own instruction stream instead: the main path of a lane
// iteration is branch-free (dead lanes add 0 to spread cells, store beyond the buffer view), so hipcc schedules LDS reads, packed math and
// MFMAs of neighbouring iterations into each other; only the far path (samples beyond the window) is a branch.
template <int R, int TERMS, int NW>
__global__ __launch_bounds__(NW * 64) void dcn_bwd6_kernel(const DcnBwd6Params p, const bf16x8* __restrict__ wpack) {
    constexpr int NK = 4, TH = 8, NT = NW * 64, ROWS = TH / NW;
    constexpr int TR = TH + 2 * R + 3, TC = 32 + 2 * R + 3, NPOS = TR * TC;
    constexpr int WBLK = 2 * (2 * NK) * 32;                        // vectors per M tile: hi + lo
    constexpr int NXI = (2 * NPOS + NT - 1) / NT;                  // x-tile items (float4 of one position and quad) per thread
    constexpr int NWV = (3 * WBLK + NT - 1) / NT;                  // weight vectors per thread
    static_assert((3 * WBLK) % 64 == 0, "whole waves of weight vectors");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);              // [2 quads][NPOS], zero outside the image
    int* gwin = reinterpret_cast<int*>(xt + 2 * NPOS);             // [8 channels][NPOS]: the grad_input tile, fixed point
    bf16x8* wsb = reinterpret_cast<bf16x8*>(gwin + 8 * NPOS);      // [3][WBLK]: the chunk's weight block (whole launch)
    float* gn_red = reinterpret_cast<float*>(wsb + 3 * WBLK);      // [NW]
    if (dcn_halo_not_selected(p.sel)) return;   // (uniform)
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int nchunks = d.C >> 3;
    // workgroup -> (tile stream, chunk): the chunks of a stream on ONE XCD (linear id % 8), streams spread over the XCDs
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    // (integer divisions run on the vector ALU; readfirstlane brings the uniform results back to SGPRs -- left in VGPRs they made every
    // buffer descriptor "divergent": a waterfall loop around each of the kernel's ~400 buffer instructions)
    const int chunk = __builtin_amdgcn_readfirstlane(slot % nchunks), stream = __builtin_amdgcn_readfirstlane(xcd + 8 * (slot / nchunks));
    const int c0 = chunk * 8, g = chunk;                           // (8 channels per deformable group)
    const int t_begin = __builtin_amdgcn_readfirstlane((int)((long long)p.ntiles * stream / p.ns));
    const int t_end = __builtin_amdgcn_readfirstlane((int)((long long)p.ntiles * (stream + 1) / p.ns));
    const int per_b = p.nty * d.ntx;
    const unsigned HW = (unsigned)(d.H * d.W);
    const unsigned hw = (unsigned)(d.Ho * d.Wo);
    const unsigned pl4 = 4u * hw, HW4 = 4u * HW;
    const bool has_act = p.g.act != nullptr;

    // ---- once per workgroup: weight block (LDS-DMA), window zero, constant operands
    {
        const bf16x8* src = wpack + (size_t)chunk * 3 * WBLK;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = tid + i * NT;
            if (e - lane + 63 < 3 * WBLK)   // (wave-uniform)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + e),
                                                 (__attribute__((address_space(3))) void*)(wsb + e), 16, 0, 0);
        }
    }
    for (int e = tid; e < 8 * NPOS; e += NT) gwin[e] = 0;
    // 0/1 selectors of the transposing MFMAs as B operands: lane (j = lo, h = hi) supplies B[k = 8 h + e][j], e < 8
    //   even: B[k][j] = (j == k), j < 16        odd: B[k][j] = (j == k + 16)
    bf16x8 sel_e, sel_o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sel_e[e] = (__bf16)((lo == 8 * hi + e) ? 1.f : 0.f);
        sel_o[e] = (__bf16)((lo == 8 * hi + e + 16) ? 1.f : 0.f);
    }
    // x-tile items of this thread: (quad, row, col) are tile-independent
    int x_it[NXI];   // row << 16 | column << 1 | quad; -1: no item
#pragma unroll
    for (int k = 0; k < NXI; ++k) {
        const int it = tid + k * NT;
        const int quad = it >= NPOS ? 1 : 0, pos = it - quad * NPOS;
        const int rr_ = pos / TC;
        x_it[k] = it < 2 * NPOS ? (rr_ << 16 | (pos - rr_ * TC) << 1 | quad) : -1;
    }

    f32x16 gw_acc[2][3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) gw_acc[mb][nb] = zero16();

    // ---- requests of a tile (registers).  req_misc: the (dy, dx, mask) triples of the lane's two pixels' five taps + the x tile; req_g: gOut
    // (+ act) of ONE of its pixels -- the two rows' gOut is requested, and converted to bf16 fragments, one after the other between the two
    // halves of the window flush, so that at most 64 raw values wait in registers (all 128 at once made hipcc park them in scratch).
    float graw[32], araw[32];
    float o_dy[ROWS][5], o_dx[ROWS][5], o_m[ROWS][5];
    float xv[NXI][4];
    bf16x8 gh[ROWS][NK], gl[ROWS][NK];
    float gsq_row[ROWS];
    auto tile_coords = [&](int t, int& b, int& y0, int& x0) {
        b = __builtin_amdgcn_readfirstlane(t / per_b);
        const int rem = t - b * per_b, ty = __builtin_amdgcn_readfirstlane(rem / d.ntx);
        y0 = ty * TH;
        x0 = (rem - ty * d.ntx) * 32;
    };
    auto req_g = [&](int t, int rr) {
        int b, y0, x0;
        tile_coords(t, b, y0, x0);
        const __amdgpu_buffer_rsrc_t g_rs = buf_view_2g(p.g.p + (size_t)b * d.Co * hw);
        const __amdgpu_buffer_rsrc_t a_rs = buf_view_2g((has_act ? p.g.act : p.g.p) + (size_t)b * d.Co * hw);
        const int oy = y0 + wave + NW * rr, ox = x0 + lo;
        const bool px_ok = oy < d.Ho && ox < d.Wo;
        // o = 16 ks + 8 hi + j; whole k-steps beyond Co read as zero (Co % 16 == 0): lane offset beyond the 2 GB view
        const unsigned vo = px_ok ? 4u * (unsigned)(oy * d.Wo + ox) + (unsigned)(8 * hi) * pl4 : 0x80000000u;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const unsigned vk = 16 * ks < d.Co ? vo : 0x80000000u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                graw[8 * ks + j] = buf_load(g_rs, vk, (unsigned)(16 * ks + j) * pl4);
                if (has_act) araw[8 * ks + j] = buf_load(a_rs, vk, (unsigned)(16 * ks + j) * pl4);
            }
        }
    };
    auto conv_g = [&](int rr) {   // gOut (x act') as bf16 hi / lo fragments: K = output channels, 4 k-steps of 16; squared pixel norm
        float gsq = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = graw[8 * ks + j];
                if (has_act) f *= araw[8 * ks + j] > 0.f ? 1.f : p.g.slope;
                v[j] = f;
                gsq = __builtin_fmaf(f, f, gsq);
            }
            split8(v, gh[rr][ks], gl[rr][ks]);
        }
        gsq_row[rr] = gsq;
    };
    auto req_misc = [&](int t) {
        int b, y0, x0;
        tile_coords(t, b, y0, x0);
        const __amdgpu_buffer_rsrc_t off_rs = buf_view(d.offset + (size_t)b * d.off_bs), msk_rs = buf_view(d.mask + (size_t)b * d.mask_bs);
        const unsigned ob = (unsigned)(g * 18) * pl4, mb_ = (unsigned)(g * 9) * pl4;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int oy = y0 + wave + NW * rr, ox = x0 + lo;
            const bool px_ok = oy < d.Ho && ox < d.Wo;
            const unsigned pv = 4u * (unsigned)(px_ok ? oy * d.Wo + ox : y0 * d.Wo + x0);   // (lanes without a pixel read the tile's first pixel: masked later)
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int t0 = bwd6_tap(it, 0), t1 = it < 4 ? bwd6_tap(it, 1) : t0;   // (it == 4, half 1: no tap; reads tap 8, unused)
                const unsigned tp = (unsigned)(hi ? t1 : t0) * pl4;
                o_dy[rr][it] = buf_load(off_rs, pv + 2u * tp, ob);
                o_dx[rr][it] = buf_load(off_rs, pv + 2u * tp, ob + pl4);
                o_m[rr][it] = buf_load(msk_rs, pv + tp, mb_);
            }
        }
        const __amdgpu_buffer_rsrc_t x_rs = buf_view_2g(d.x + (size_t)b * d.C * HW);
        const int ty0 = y0 - d.pad - R, tx0 = x0 - d.pad - R;
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int gy = ty0 + (x_it[k] >> 16), gx = tx0 + ((x_it[k] >> 1) & 0x7fff);
            const bool ok = x_it[k] >= 0 && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
            const unsigned xo = ok ? 4u * ((unsigned)(gy * d.W + gx) + (unsigned)(4 * (x_it[k] & 1)) * HW) : 0x80000000u;
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[k][e] = buf_load(x_rs, xo, (unsigned)(c0 + e) * HW4);
        }
    };
    if (t_begin < t_end) {
        req_misc(t_begin);
        req_g(t_begin, 0);
        if (ROWS == 2) {
            conv_g(0);
            req_g(t_begin, 1);
        }
    }

    unsigned mg0 = 0x4B400000u;          // 1.5 * 2^23: the magic number of the fixed-point rounding, kept out of the literal encoder
    asm volatile("" : "+s"(mg0));
    const f32x2 MAGIC = {__builtin_bit_cast(float, mg0), __builtin_bit_cast(float, mg0)};

    for (int t = t_begin; t < t_end; ++t) {
        int b, y0, x0;
        tile_coords(t, b, y0, x0);
        const int ty0 = y0 - d.pad - R, tx0 = x0 - d.pad - R;
        const __amdgpu_buffer_rsrc_t gx_rs = buf_view(p.gx + (size_t)b * d.C * HW);
        const __amdgpu_buffer_rsrc_t goff_rs = buf_view(p.goff + (size_t)b * p.goff_bs), gmsk_rs = buf_view(p.gmask + (size_t)b * p.gmask_bs);

        conv_g(ROWS - 1);   // (two rows per wave: row 0 was converted between the two halves of the previous tile's flush)
        float gsq_max = half_sum6(gsq_row[0]);
        if (ROWS == 2) gsq_max = fmaxf(gsq_max, half_sum6(gsq_row[ROWS - 1]));
        {   // Gn^2 = the largest squared pixel norm of the tile (read back after the barrier)
#pragma unroll
            for (int sft = 16; sft > 0; sft >>= 1) gsq_max = fmaxf(gsq_max, __shfl_xor(gsq_max, sft));
            if (lane == 0) gn_red[wave] = gsq_max;
        }
        // ---- commit the x tile
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int it = tid + k * NT;
            if (it < 2 * NPOS) xt[it] = make_float4(xv[k][0], xv[k][1], xv[k][2], xv[k][3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (first tile: this wave's share of the weight DMA has landed)
        __syncthreads();

        // fixed-point scale (dcn5_kernels.hip header): |contribution| * S <= 0.995 * 2^31 / 2304
        float S, invS;
        {
            float g2 = gn_red[0];
#pragma unroll
            for (int k = 1; k < NW; ++k) g2 = fmaxf(g2, gn_red[k]);
            const float bound = 1.002f * p.wnorm[chunk] * sqrtf(g2);
            S = bound > 0.f ? 927407.f / bound : 0.f;
            invS = bound > 0.f ? bound * (1.f / 927407.f) : 0.f;
        }

#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int oy = y0 + wave + NW * rr, ox = x0 + lo;
            const bool px_ok = oy < d.Ho && ox < d.Wo;
            const unsigned pix4 = px_ok ? 4u * (unsigned)(oy * d.Wo + ox) : 0u;
            const float by = (float)(oy - d.pad), bx = (float)(ox - d.pad);
            // ---- gOut transposed by the matrix core: G[i = pixel][j = o] -> registers (lane = o, register = pixel) = the A operands of the
            // weight-gradient GEMM: ag[mb][ks][hi, lo]
            bf16x8 ag[2][2][2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x16 th = mfma_bf16(gh[rr][2 * mb], sel_e, zero16());
                th = mfma_bf16(gh[rr][2 * mb + 1], sel_o, th);
                ag[mb][0][0] = pack8_exact(th, 0);
                ag[mb][1][0] = pack8_exact(th, 8);
                if (TERMS >= 3) {
                    f32x16 tl = mfma_bf16(gl[rr][2 * mb], sel_e, zero16());
                    tl = mfma_bf16(gl[rr][2 * mb + 1], sel_o, tl);
                    ag[mb][0][1] = pack8_exact(tl, 0);
                    ag[mb][1][1] = pack8_exact(tl, 8);
                }
            }
            f32x16 dt_h, dt_l;   // column values of two lane iterations, transposed: D[i = pixel][j = 16 (it & 1) + 8 h + ch]
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                f32x16 acc = zero16();
                const bf16x8* wb_hi = wsb + mt * WBLK;
                const bf16x8* wb_lo = wb_hi + (2 * NK) * 32;
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    const bf16x8 ah = wb_hi[(2 * ks + hi) * 32 + lo];
                    acc = mfma_bf16(ah, gh[rr][ks], acc);
                    if (TERMS >= 2) acc = mfma_bf16(ah, gl[rr][ks], acc);
                    if (TERMS >= 3) acc = mfma_bf16(wb_lo[(2 * ks + hi) * 32 + lo], gh[rr][ks], acc);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int it = 2 * mt + s;
                    if (it >= 5) continue;   // (compile time)
                    const int t0 = bwd6_tap(it, 0), t1 = it < 4 ? bwd6_tap(it, 1) : 0;
                    const bool has_tap = it < 4 || hi == 0;
                    const int tap = hi ? t1 : t0;
                    const bool act_lane = px_ok && has_tap;
                    const float dy = act_lane ? o_dy[rr][it] : 0.f, dx = act_lane ? o_dx[rr][it] : 0.f;
                    float m = o_m[rr][it];
                    if (d.mask_logit) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));
                    const float kyf = hi ? (float)(t1 / 3) : (float)(t0 / 3), kxf = hi ? (float)(t1 % 3) : (float)(t0 % 3);
                    // sample position in IMAGE coordinates exactly as the reference forms it (kernel.cu:594-616, 722-737)
                    const float y = (by + kyf) + dy, x = (bx + kxf) + dx;
                    const float fy = floorf(y), fx = floorf(x);
                    const int yi = (int)fy, xi = (int)fx;
                    const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
                    const int r0 = yi - ty0, s0 = xi - tx0;
                    const bool in_tile = act_lane && (unsigned)r0 < (unsigned)(TR - 1) && (unsigned)s0 < (unsigned)(TC - 1);
                    const bool inside = y > -1.f && x > -1.f && y < (float)d.H && x < (float)d.W;
                    const bool far = !in_tile && inside && act_lane;   // beyond the window: global gather / atomics with the full rule set
                    const int pos0 = in_tile ? r0 * TC + s0 : lo;      // (dead lanes: distinct cells of the first row; they add 0)
                    const float ml = in_tile ? m : 0.f;               // dead and far lanes contribute nothing on the main path
                    const f32x2 L2 = {ly, lx}, MS = {ml, ml * S};
                    const f32x2 W01 = {hy * hx, hy * lx}, W23 = {ly * hx, ly * lx};
                    f32x2 gm2 = {0.f, 0.f}, gy2 = {0.f, 0.f}, gx2 = {0.f, 0.f};
                    float colv[8];
                    int* wq = gwin + pos0;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float4* xq = xt + q * NPOS + pos0;
                        const float4 a00 = xq[0], a01 = xq[1], a10 = xq[TC], a11 = xq[TC + 1];
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {       // channel pairs (2 pr, 2 pr + 1), pr = 2 q + ph
                            const int pr = 2 * q + ph;
                            const f32x2 c00 = ph ? f32x2{a00.z, a00.w} : f32x2{a00.x, a00.y};
                            const f32x2 c01 = ph ? f32x2{a01.z, a01.w} : f32x2{a01.x, a01.y};
                            const f32x2 c10 = ph ? f32x2{a10.z, a10.w} : f32x2{a10.x, a10.y};
                            const f32x2 c11 = ph ? f32x2{a11.z, a11.w} : f32x2{a11.x, a11.y};
                            const f32x2 cg = {acc[8 * s + 2 * pr], acc[8 * s + 2 * pr + 1]};
                            const f32x2 Bv = pk_sub(c01, c00), Cv = pk_sub(c10, c00), Dv = pk_sub(pk_sub(c11, c01), Cv);
                            const f32x2 dxv = pk_fma_x(L2, Dv, Bv), dyv = pk_fma_y(L2, Dv, Cv);   // d val / d x, d val / d y
                            const f32x2 val = pk_fma_y(L2, dxv, pk_fma_x(L2, Cv, c00));          // bilinear(x)
                            gm2 = pk_fma(cg, val, gm2);
                            const f32x2 tv = pk_mul_x(MS, cg);                                    // col_grad * mask
                            gy2 = pk_fma(dyv, tv, gy2);
                            gx2 = pk_fma(dxv, tv, gx2);
                            const f32x2 cv = pk_mul_x(MS, val);                                   // the column value of the weight gradient
                            colv[2 * pr] = cv.x; colv[2 * pr + 1] = cv.y;
                            // ---- scatter: 8 LDS integer atomics per channel pair into the shared window (unconditional)
                            const f32x2 ts = pk_mul_y(MS, cg);                                    // col_grad * mask * S
                            const f32x2 u00 = pk_fma_x(W01, ts, MAGIC), u01 = pk_fma_y(W01, ts, MAGIC);
                            const f32x2 u10 = pk_fma_x(W23, ts, MAGIC), u11 = pk_fma_y(W23, ts, MAGIC);
                            // (.x / .y spelled out: `u00[e]` under an unrolled e compiled to element 0 twice with this hipcc)
                            int* q0 = wq + (2 * pr) * NPOS;
                            int* q1 = q0 + NPOS;
                            lds_add_i32_6(q0, (int)(__float_as_uint(u00.x) - 0x4B400000u));
                            lds_add_i32_6(q0 + 1, (int)(__float_as_uint(u01.x) - 0x4B400000u));
                            lds_add_i32_6(q0 + TC, (int)(__float_as_uint(u10.x) - 0x4B400000u));
                            lds_add_i32_6(q0 + TC + 1, (int)(__float_as_uint(u11.x) - 0x4B400000u));
                            lds_add_i32_6(q1, (int)(__float_as_uint(u00.y) - 0x4B400000u));
                            lds_add_i32_6(q1 + 1, (int)(__float_as_uint(u01.y) - 0x4B400000u));
                            lds_add_i32_6(q1 + TC, (int)(__float_as_uint(u10.y) - 0x4B400000u));
                            lds_add_i32_6(q1 + TC + 1, (int)(__float_as_uint(u11.y) - 0x4B400000u));
                        }
                    }
                    float gm_s = gm2.x + gm2.y, gy_s = gy2.x + gy2.y, gx_s = gx2.x + gx2.y;
                    gm_s = in_tile ? gm_s : 0.f;
                    if (far) {   // ---- rare: the whole (pixel, tap) from global memory with the reference's rule set, plain arithmetic
                        const bool vy0 = yi >= 0, vy1 = yi + 1 <= d.H - 1, vx0 = xi >= 0, vx1 = xi + 1 <= d.W - 1;
                        const int cy0 = vy0 ? yi : 0, cy1 = vy1 ? yi + 1 : d.H - 1, cx0 = vx0 ? xi : 0, cx1 = vx1 ? xi + 1 : d.W - 1;
                        const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
                        const float z00 = (vy0 && vx0) ? 1.f : 0.f, z01 = (vy0 && vx1) ? 1.f : 0.f;
                        const float z10 = (vy1 && vx0) ? 1.f : 0.f, z11 = (vy1 && vx1) ? 1.f : 0.f;
                        const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
                        const float* pl = d.x + ((size_t)b * d.C + c0) * HW;
                        float* gp = p.gx + ((size_t)b * d.C + c0) * HW;
                        gm_s = gy_s = gx_s = 0.f;
#pragma unroll 1
                        for (int e = 0; e < 8; ++e) {   // (a real loop: ten unrolled copies of this block made the register allocator spill on the main path)
                            const float* qp = pl + (size_t)e * HW;
                            const float c00 = qp[i00] * z00, c01 = qp[i01] * z01, c10 = qp[i10] * z10, c11 = qp[i11] * z11;
                            const float Bq = c01 - c00, Cq = c10 - c00, Dq = (c11 - c01) - Cq;
                            const float dxq = Bq + ly * Dq, dyq = Cq + lx * Dq, vq = (c00 + ly * Cq) + lx * dxq;
                            float cgq = acc[8 * s];
#pragma unroll
                            for (int j = 1; j < 8; ++j) cgq = e == j ? acc[8 * s + j] : cgq;
                            const float tq = cgq * m;
                            gm_s += cgq * vq;
                            gy_s += dyq * tq;
                            gx_s += dxq * tq;
#pragma unroll
                            for (int j = 0; j < 8; ++j) colv[j] = e == j ? vq * m : colv[j];
                            float* gq = gp + (size_t)e * HW;
                            if (z00 * w00 != 0.f) atomicAdd(gq + i00, w00 * tq);
                            if (z01 * w01 != 0.f) atomicAdd(gq + i01, w01 * tq);
                            if (z10 * w10 != 0.f) atomicAdd(gq + i10, w10 * tq);
                            if (z11 * w11 != 0.f) atomicAdd(gq + i11, w11 * tq);
                        }
                    }
                    {   // grad_offset / grad_mask of (pixel, tap): this lane holds the sum over the group's 8 channels; dead lanes store beyond the view
                        if (d.mask_logit) gm_s *= m * (1.f - m);
                        const unsigned tp = (unsigned)tap * pl4;
                        const unsigned so = act_lane ? pix4 + 2u * tp : 0xfffffffcu, sm = act_lane ? pix4 + tp : 0xfffffffcu;
                        buf_store(goff_rs, so, (unsigned)(g * 18) * pl4, gy_s);
                        buf_store(goff_rs, so, (unsigned)(g * 18) * pl4 + pl4, gx_s);
                        buf_store(gmsk_rs, sm, (unsigned)(g * 9) * pl4, gm_s);
                    }
                    // ---- the column values of this iteration into the transposing MFMAs (spare slot of iteration 4, half 1: the ones column of
                    // the bias gradient)
                    if (it == 4) colv[0] = hi ? (px_ok ? 1.f : 0.f) : colv[0];
                    bf16x8 ch_, cl_;
                    split8(colv, ch_, cl_);
                    if (s == 0) {
                        dt_h = mfma_bf16(ch_, sel_e, zero16());
                        if (TERMS >= 2) dt_l = mfma_bf16(cl_, sel_e, zero16());
                    } else {
                        dt_h = mfma_bf16(ch_, sel_o, dt_h);
                        if (TERMS >= 2) dt_l = mfma_bf16(cl_, sel_o, dt_l);
                    }
                }
                // ---- weight gradient of n-block mt (two lane iterations): gw_acc[mb][mt] += gOut^T[mb] x col, K = this row's 32 pixels
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 bh = pack8_exact(dt_h, 8 * ks);
                    bf16x8 bl = bh;
                    if (TERMS >= 2) bl = pack8_exact(dt_l, 8 * ks);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        gw_acc[mb][mt] = mfma_bf16(ag[mb][ks][0], bh, gw_acc[mb][mt]);
                        if (TERMS >= 2) gw_acc[mb][mt] = mfma_bf16(ag[mb][ks][0], bl, gw_acc[mb][mt]);
                        if (TERMS >= 3) gw_acc[mb][mt] = mfma_bf16(ag[mb][ks][1], bh, gw_acc[mb][mt]);
                    }
                }
            }
        }
        __syncthreads();
        const bool more = t + 1 < t_end;   // (uniform)
        if (more) {                         // the next tile's requests fly while the window is flushed
            req_misc(t + 1);
            req_g(t + 1, 0);
        }
        // ---- flush: wave w owns channels c0 + w and c0 + w + 4; one global atomic per touched cell inside the image, cell back to zero
#pragma unroll
        for (int cc = 0; cc < 8 / NW; ++cc) {
            const int ch = wave + NW * cc;
            const unsigned cpl = (unsigned)(c0 + ch) * HW4;
            int* gc = gwin + ch * NPOS;
            for (int base = lane; base < NPOS; base += 256) {   // four cells per round trip
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
                        if (yy >= 0 && yy < d.H && xx >= 0 && xx < d.W)
                            buf_atomic_add(gx_rs, 4u * (unsigned)(yy * d.W + xx), cpl, (float)v[j] * invS);
                    }
                }
            }
            if (ROWS == 2 && cc == 0 && more) {
                conv_g(0);
                req_g(t + 1, 1);
            }
        }
        // (the barrier after the next commit orders this flush before the next tile's atomics)
    }

    // ---- weight / bias gradient partial of this (stream, chunk): sum of the 4 waves (8 rows), fixed order, through LDS
    __syncthreads();
    {
        float* red = reinterpret_cast<float*>(smem_raw);   // [4 waves][16 registers][64 lanes] = 16 KB
        const int K = d.C * 9;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = gw_acc[mb][nb][r];
                __syncthreads();
#pragma unroll
                for (int rq = 0; rq < 16 / NW; ++rq) {
                    const int r = wave + NW * rq;     // thread (wave, lane) sums register r of lane `lane` over the waves
                    float sum = 0.f;
#pragma unroll
                    for (int w4 = 0; w4 < NW; ++w4) sum += red[(w4 * 16 + r) * 64 + lane];
                    const int o = 32 * mb + drow(r, hi);
                    const int it = 2 * nb + (lo >> 4), h2 = (lo >> 3) & 1, ch = lo & 7;
                    const int tap = bwd6_tap(it, h2);
                    if (o < d.Co) {
                        if (tap >= 0) p.part[((size_t)stream * d.Co + o) * K + (size_t)(c0 + ch) * 9 + tap] = sum;
                        else if (it == 4 && h2 == 1 && ch == 0 && chunk == 0 && p.bpart != nullptr) p.bpart[(size_t)stream * d.Co + o] = sum;
                    }
                }
                __syncthreads();
            }
        }
    }
}

size_t rvsr_dcn_bwd6_workspace_bytes(int Co, int C, int* ns_out) {
    const int nchunks = C / 8;
    const int ns = nchunks > 0 && 32 % nchunks == 0 ? 256 / nchunks : 0;
    if (ns_out) *ns_out = ns;
    const size_t wbytes = (size_t)nchunks * 3 * 2 * 8 * 32 * 16;
    const size_t nrm = ((size_t)nchunks * 4 + 255) & ~(size_t)255;
    return wbytes + nrm + (size_t)ns * ((size_t)Co * C * 9 + Co) * sizeof(float) + 256;
}

// (dcn5_kernels.hip)
__global__ void dcn_bwd5_wnorm_kernel(const float* __restrict__ w, float* __restrict__ wn, int Co, int C);

int rvsr_dcn_bwd6_supported(const DcnGeom& d, const TView& g) {
    if (d.cpg != 8 || d.C % 8 != 0 || d.stride != 1 || d.dil != 1 || d.Co > 64 || d.Co % 16 != 0 || g.mode != 0) return 0;
    const int nchunks = d.C / 8;
    if (nchunks > 32 || 32 % nchunks != 0) return 0;
    const size_t planes = (size_t)(d.C / d.cpg) * 18 > (size_t)d.C ? (size_t)(d.C / d.cpg) * 18 : (size_t)d.C;
    if (planes * (size_t)d.H * d.W * sizeof(float) >= ((size_t)1 << 31) || planes * (size_t)d.Ho * d.Wo * sizeof(float) >= ((size_t)1 << 31))
        return 0;
    if ((size_t)d.Co * d.Ho * d.Wo * sizeof(float) >= ((size_t)1 << 31)) return 0;
    return 1;
}

template <int R>
static int launch_bwd6(const DcnBwd6Params& p, const bf16x8* wpack, hipStream_t st) {
    constexpr int TH = 8, TR = TH + 2 * R + 3, TC = 32 + 2 * R + 3, NPOS = TR * TC;
    const size_t lds = (size_t)NPOS * (2 * 16 + 8 * 4) + (size_t)3 * 2 * 8 * 32 * 16 + 8 * sizeof(float);
    static const int nw = [] { const char* e = getenv("RVSR_DCN6_NW"); return e ? atoi(e) : 8; }();   // developer A/B switch: waves per workgroup
    auto k = nw == 4 ? dcn_bwd6_kernel<R, 3, 4> : dcn_bwd6_kernel<R, 3, 8>;
#ifndef RVSR_DCN6_DEV
    const int nt = rvsr_gemm_terms();
    if (nt == 2) k = dcn_bwd6_kernel<R, 2, 8>;
    if (nt == 1) k = dcn_bwd6_kernel<R, 1, 8>;
#endif
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_bwd6: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(k, dim3(p.ns * (p.d.C / 8)), dim3(nw == 4 ? 256 : 512), lds, st, p, wpack);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_bwd6 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// All five gradients of the fused pack / the operator.  gw / gb are ACCUMULATED into (the reference's convention, cpp:659-671).
// halo: 2 / 4 / 6 (window around the tile); < 0: 4.
int rvsr_launch_dcn_bwd6(const DcnGeom& d, const float* weight, const TView& g, float* gx, float* goff, size_t goff_bs, float* gmask,
                         size_t gmask_bs, float* gw, float* gb, void* workspace, size_t workspace_bytes, hipStream_t st, int halo) {
    if (!rvsr_dcn_bwd6_supported(d, g)) return RVSR_ERR_UNSUPPORTED;
    int ns = 0;
    if (!workspace || workspace_bytes < rvsr_dcn_bwd6_workspace_bytes(d.Co, d.C, &ns)) return RVSR_ERR_UNSUPPORTED;
    const int nchunks = d.C / 8;
    const size_t wbytes = (size_t)nchunks * 3 * 2 * 8 * 32 * 16;
    bf16x8* wpack = (bf16x8*)workspace;
    float* wnorm = (float*)((unsigned char*)workspace + wbytes);
    float* part = (float*)((unsigned char*)workspace + wbytes + (((size_t)nchunks * 4 + 255) & ~(size_t)255));
    const size_t nw = (size_t)d.Co * d.C * 9;
    hipLaunchKernelGGL(dcn_bwd5_wnorm_kernel, dim3(nchunks), dim3(576), 0, st, weight, wnorm, d.Co, d.C);
    const size_t total = (size_t)nchunks * 3 * 8 * 32;
    hipLaunchKernelGGL(pack_weights_bwd6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight, wpack, d.Co, d.C, nchunks);
    DcnBwd6Params p;
    p.d = d; p.g = g; p.gx = gx; p.goff = goff; p.gmask = gmask; p.goff_bs = goff_bs; p.gmask_bs = gmask_bs;
    p.sel = dcn_halo_always(); p.wnorm = wnorm;
    p.part = part; p.bpart = gb ? part + (size_t)ns * nw : nullptr;
    p.ns = ns; p.nty = (d.Ho + 7) / 8; p.ntiles = d.B * p.nty * d.ntx;
    int rc;
#ifdef RVSR_DCN6_DEV
    rc = launch_bwd6<4>(p, wpack, st);
#else
    if (halo == 2) rc = launch_bwd6<2>(p, wpack, st);
    else if (halo == 6) rc = launch_bwd6<6>(p, wpack, st);
    else rc = launch_bwd6<4>(p, wpack, st);
#endif
    if (rc != RVSR_OK) return rc;
    rvsr_launch_reduce(part, ns, nw, gw, 1, st, p.bpart, (size_t)d.Co, gb);
    return RVSR_OK;
}
