// dcn4_kernels.hip -- fused modulated-DCN forward, fourth generation (gfx950): ONE software pipeline per wave.
//
// Same arithmetic as dcn_fwd3_kernel (out[Co, px] = W[Co, (tap, c)] * col[(tap, c), px], the column values built by the lane the
// matrix core expects them from out of a zero-padded LDS x tile, bf16x3 products, f32 everything else; reference semantics:
// deform_conv_cuda_kernel.cu:467-497, 571-633, deform_conv_cuda.cpp:490-569).  What changed is the SCHEDULE.  The third generation
// ran, per wave and k-step, "8 corner reads -> wait -> blend + split (62 VALU) -> 4 weight reads -> 6 MFMAs" as phases, two
// barriers per 16-channel chunk, four waves per SIMD: the SQ counters (profiles/r03_dcn_sq_counters.json) show LDS 50 %, VALU 25 %,
// matrix pipe 27 % busy -- adding up to the kernel time.  In-order waves whose MFMAs come in bursts of six queue behind each
// other on the SIMD's matrix pipe and fall into step; nothing overlaps.  Here:
//   * the k-steps of a tile form ONE flat sequence j = 0 .. 9*C/16 - 1 and every wave runs them as a pipeline: in iteration i it
//     blends the corners of k-step i, issues the corner reads of k-step i+1, splits k-step i into bf16 hi / lo, feeds the matrix
//     core with k-step i-1 and computes the sampling geometry of k-step i+2 from offsets requested an iteration earlier -- one
//     basic block per nine k-steps, MFMAs spread between the vector work of OTHER k-steps of the same wave;
//   * a k-step pairs two UNITS u = 2j, 2j+1 in the two lane halves, a unit = (8-channel chunk u / 9, tap u % 9): the x tile of a
//     chunk is half as large as the third generation's, which buys a 5 x 7 px halo (rows x columns), a ring of three chunk slots
//     and a ring of ten k-step weight slices in 160 KB: ONE barrier per chunk, placed where the next chunk's slot and slices are
//     free anyway, x tile and weights of later k-steps in flight underneath (global loads -> registers -> LDS across a barrier
//     interval; weights by LDS-DMA);
//   * no branch in the pipeline: a sample whose 2x2 footprint leaves the tile contributes zero there and sets a flag bit; after
//     each period of nine k-steps the flagged samples (at the 1 px offsets the benchmark runs: ~1e-5 of them) are gathered from
//     global memory with the reference's full rule set and added through extra MFMAs (the product is linear in the column).
// One workgroup of NW waves (= NW output rows x 32 columns x all output channels of an m-block) per CU.
#include "dcn_tile.h"

#ifdef RVSR_TIMELINE_DCN4
__device__ unsigned long long rvsr_dbg_dcn4[16 * 64];
extern "C" int rvsr_debug_read_dcn4(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rvsr_dbg_dcn4), sizeof(unsigned long long) * 16 * 64); }
// per-wave stamps of one workgroup: [wave][32]
#define TSTAMP4(i) do { if (blockIdx.x == 77 && lane == 0) rvsr_dbg_dcn4[wave * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP4(i) do {} while (0)
#endif

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2v lo2v(const float4& a) { return f32x2v{a.x, a.y}; }
__device__ __forceinline__ f32x2v hi2v(const float4& a) { return f32x2v{a.z, a.w}; }
// s.x * a00 + t.x * a01 + s.y * a10 + t.y * a11 on a pair of channels (VOP3P op_sel / op_sel_hi pick the broadcast half)
__device__ __forceinline__ f32x2v blend4v(f32x2v s, f32x2v t, f32x2v a00, f32x2v a01, f32x2v a10, f32x2v a11) {
    f32x2v r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(s), "v"(a00));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(r) : "v"(t), "v"(a01));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(r) : "v"(s), "v"(a10));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(r) : "v"(t), "v"(a11));
    return r;
}

// Rotating wave priority.  The SIMD arbiter serves the OLDER of two ready waves ("priority, then age"): the first-dispatched waves of
// a workgroup run at full speed, the later ones on what is left, and at every barrier the former wait 1.5-2 K cycles for the latter
// (tools/dcn4_timeline.py) while the LDS -- what this kernel is bound by -- idles.  s_setprio takes an immediate, so the group-
// dependent value is set through a three-way scalar branch inside ONE asm statement (no basic-block split for the compiler).
__device__ __forceinline__ void fwd4_setprio(int pr /* wave-uniform, 0 .. 2 */) {
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 10f\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 11f\n\ts_setprio 2\n\ts_branch 12f\n"
                 "10:\n\ts_setprio 0\n\ts_branch 12f\n11:\n\ts_setprio 1\n12:" : : "s"(pr) : "scc");
}

template <int NW, int RY, int RX, int MT>
struct Fwd4 {
    static constexpr int NT = NW * 64;
    static constexpr int TR = NW + 2 * RY + 2, TC = 32 + 2 * RX + 2, NPOS = TR * TC, TC4 = TC / 4;
    static constexpr int MP = MT * 32;
    static constexpr int XSLOT = 2 * NPOS;     // float4 per chunk slot: [2 quads][NPOS]
    static constexpr int NXS = 3;              // chunk slots
    static constexpr int WSLOT = 4 * MP;       // 16-byte vectors per k-step weight slice: [hi | lo part][lane half][MP]
    static constexpr int NWS = 10;             // weight slots
    static constexpr int WPI = WSLOT / 64;     // LDS-DMA wave-instructions per weight slice
    static constexpr int NITEM = 2 * TR * TC4; // x staging items (quad, row, group of 4 columns) per chunk
    static constexpr size_t LDS = (size_t)16 * (NXS * XSLOT + NWS * WSLOT) + sizeof(float) * 4 * MP;   // (+ bias of up to 4 m-blocks)
    static_assert(TC % 4 == 0 && (RX + 1) % 4 == 0, "tile rows are whole, 16-byte aligned groups of 4 columns");
    static_assert(NITEM <= NT, "one x staging item per thread");
};

// tap of unit u_local (0 .. 17) of a period, and which of the period's two chunks it belongs to
__host__ __device__ constexpr int f4_cl(int ul) { return ul >= 9 ? 1 : 0; }
__host__ __device__ constexpr int f4_tap(int ul) { return ul >= 9 ? ul - 9 : ul; }

struct Fwd4Geo {          // sampling geometry of one k-step, per lane
    unsigned addr;        // LDS byte address of the top-left corner's float4 (quad 0)
    f32x2v wsd, wt;       // corner weights x mask: (w00, w10), (w01, w11)
};

// Per-period context: everything a pipeline stage needs to know about the TILE its k-step belongs to.  Two of them are live --
// `C` for the period the loop body is in, `N` for the next one (the same tile, or the workgroup's next tile): which one a stage
// uses is a compile-time property of its position in the body, so a tile boundary costs no selects inside the pipeline.
struct Fwd4Ctx {
    float Y0, X0;            // (float)(oy - pad), (float)(ox - pad)
    unsigned pix4;           // byte offset of the lane's output pixel inside a plane (0 for lanes without work)
    unsigned x_voff;         // byte offset of the lane's x staging item inside batch element b (bit 31: reads zero)
    bool px_ok;
    int ty0, tx0;            // image coordinates of tile (0, 0)
    int y0, x0, b, mb;
    int per;                 // period inside the tile
};

template <int NW, int RY, int RX, int MT>
__global__ __launch_bounds__(NW * 64) void dcn_fwd4_kernel(const DcnFwdParams p, const bf16x8* __restrict__ wpack, const int cpg8s, const int ntiles, const int prio_rot, const int dbg) {
    using F = Fwd4<NW, RY, RX, MT>;
    constexpr int TR = F::TR, TC = F::TC, NPOS = F::NPOS, TC4 = F::TC4, MP = F::MP;
    constexpr int XSLOT = F::XSLOT, WSLOT = F::WSLOT, NWS = F::NWS, WPI = F::WPI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* const xs = reinterpret_cast<float4*>(smem_raw);            // [3 slots][2 quads][NPOS], zero outside the image
    bf16x8* const ws = reinterpret_cast<bf16x8*>(xs + F::NXS * XSLOT); // [10 slots][part][half][MP]
    float* const bias_s = reinterpret_cast<float*>(ws + NWS * WSLOT);  // [nmb][MP]
    const DcnGeom& d = p.d;
    if (dcn_halo_not_selected(p.sel)) return;   // (uniform) not the kernel the offsets of this call ask for
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgrp = wave >> 2;                                        // dispatch-age group of this wave on its SIMD (4 SIMDs)
    const int nper = d.C / 16, nk = 9 * nper;
    const int nty = (d.Ho + NW - 1) / NW, nmb = (d.Co + MP - 1) / MP;
    const unsigned HW4 = 4u * (unsigned)(d.H * d.W), hw4 = 4u * (unsigned)(d.Ho * d.Wo);
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.Ho * d.Wo;
    // the tiles of this workgroup: a contiguous range of (batch, m-block, tile row, tile column), XCD by XCD (neighbouring tiles
    // share their halos through the XCD's L2 -- and mostly through this CU's own fetches)
    const unsigned lw = d.swz ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
    const int t_begin = (int)(((long long)lw * ntiles) / gridDim.x), t_end = (int)(((long long)(lw + 1) * ntiles) / gridDim.x);
    if (t_begin >= t_end) return;   // (uniform)

    // x staging item of this thread: (quad q, tile row r, columns 4 g4 .. 4 g4 + 3): four 16-byte loads (one per channel of the
    // quad) land as the float4s of four positions.  Rows / groups outside the image and threads without an item read zeros
    // through the buffer range check (lane offset beyond the 2 GB view); W % 4 == 0 (launcher) keeps a group whole.
    const int it_q = tid / (TR * TC4), it_rem = tid - it_q * (TR * TC4);
    const int it_r = it_rem / TC4, it_g4 = it_rem - it_r * TC4;
    const int xs_dst = tid < F::NITEM ? it_q * NPOS + it_r * TC + 4 * it_g4 : -1;
    const unsigned hi2 = hi ? 2u * hw4 : 0u, hi1 = hi ? hw4 : 0u;   // the hi lane half is one tap further: + 2 offset planes / + 1 mask plane

    auto make_ctx = [&](Fwd4Ctx& c, int tile, int per) {
        int t = tile;
        const int tx = t % d.ntx; t /= d.ntx;
        const int ty = t % nty; t /= nty;
        c.mb = t % nmb; c.b = t / nmb;
        c.x0 = tx * 32; c.y0 = ty * NW;
        c.ty0 = c.y0 - d.pad - RY; c.tx0 = c.x0 - d.pad - RX;
        const int oy = c.y0 + wave, ox = c.x0 + lo;
        c.px_ok = oy < d.Ho && ox < d.Wo;
        c.pix4 = c.px_ok ? 4u * (unsigned)(oy * d.Wo + ox) : 0u;   // lanes without work read pixel 0 (loads stay unconditional)
        // Sample positions are formed in IMAGE coordinates exactly as the reference does (float(h_in + i) + offset,
        // kernel.cu:594-616) so that floor() and the fractional weights round identically
        c.Y0 = (float)(oy - d.pad); c.X0 = (float)(ox - d.pad);
        const int gy = c.ty0 + it_r, gx = c.tx0 + 4 * it_g4;
        const bool ok = tid < F::NITEM && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        c.x_voff = ok ? 4u * (unsigned)(gy * d.W + gx) + (unsigned)(4 * it_q) * HW4 : 0x80000000u;
        c.per = per;
    };

    // ---- staging: global -> registers -> LDS, one chunk (8 channels) at a time
    auto x_load = [&](f32x4v (&xv)[4], const Fwd4Ctx& c, int cl) {   // chunk 2 c.per + cl of c's tile
        const __amdgpu_buffer_rsrc_t x2g_rs = buf_view_2g(d.x + (size_t)c.b * d.C * HW);   // bit 31 of a lane offset = "reads zero"
        const int chunk = 2 * c.per + cl;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            xv[e] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(x2g_rs, (int)c.x_voff, (int)((unsigned)(8 * chunk + e) * HW4), 0));
    };
    auto x_write2 = [&](const f32x4v (&xv)[4], int slot, int half) {   // two of the item's four positions
        if (xs_dst >= 0) {
            float4* dst = xs + slot * XSLOT + xs_dst;
            if (half == 0) {
                dst[0] = make_float4(xv[0].x, xv[1].x, xv[2].x, xv[3].x);
                dst[1] = make_float4(xv[0].y, xv[1].y, xv[2].y, xv[3].y);
            } else {
                dst[2] = make_float4(xv[0].z, xv[1].z, xv[2].z, xv[3].z);
                dst[3] = make_float4(xv[0].w, xv[1].w, xv[2].w, xv[3].w);
            }
        }
    };
    // weight slices of `nj` k-steps starting `ahead` k-steps after the current one (slot `slot_now`, k-step `j_now` of m-block mb_now;
    // past the tile's last k-step: the first ones of the next tile's m-block) -> their ring slots, by LDS-DMA
    const unsigned ws_base = (unsigned)(F::NXS * XSLOT * 16);   // LDS byte address of the weight ring (the dynamic LDS segment starts at 0)
    auto w_issue = [&](int j_now, int slot_now, int ahead, int nj, int mb_now, int mb_next) {
#pragma unroll
        for (int k = 0; k < (5 * WPI + NW - 1) / NW; ++k) {
            const int idx = wave + k * NW;               // (uniform)
            if (idx < nj * WPI) {
                const int dj = ahead + idx / WPI, sub = idx % WPI;
                int j = j_now + dj, mbj = mb_now;
                if (j >= nk) { j -= nk; mbj = mb_next; }
                int slot = slot_now + dj;
                while (slot >= NWS) slot -= NWS;
                // Inline assembly on purpose: next to a __builtin_amdgcn_global_load_lds in flight hipcc waits vmcnt(0) at the use of EVERY
                // ordinary load (here: the offset / mask requests of each k-step), which drains the requests of the following
                // k-steps once per iteration.  The DMA has no register destination, so hiding it from the compiler's counters is safe
                // (its waits can only become conservative); completion is waited for explicitly before the next barrier event.
                const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ws_base + (unsigned)((slot * WSLOT + sub * 64) * 16)));
                const bf16x8* src = wpack + ((size_t)mbj * nk + j) * WSLOT + sub * 64 + lane;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
            }
        }
    };

    // ---- offset / mask requests of k-step jj of c's period: three dword loads per lane
    struct Req { float dy, dx, m; };
    auto request = [&](Req& q, const Fwd4Ctx& c, int jj) {
        const __amdgpu_buffer_rsrc_t off_rs = buf_view(d.offset + (size_t)c.b * d.off_bs);
        const __amdgpu_buffer_rsrc_t msk_rs = buf_view(d.mask + (size_t)c.b * d.mask_bs);
        // (uniform scalar arithmetic) deformable groups of the period's two chunks; unit u reads offset channels 18 g + 2 t (+1)
        // and mask channel 9 g + t
        const int g0 = (2 * c.per) >> cpg8s, g1 = (2 * c.per + 1) >> cpg8s;
        if (jj != 4) {
            const int ul = 2 * jj, g = f4_cl(ul) ? g1 : g0, t = f4_tap(ul);
            const unsigned so = (unsigned)(18 * g + 2 * t) * hw4, sm = (unsigned)(9 * g + t) * hw4;
            q.dy = buf_load(off_rs, c.pix4 + hi2, so);
            q.dx = buf_load(off_rs, c.pix4 + hi2, so + hw4);
            q.m = buf_load(msk_rs, c.pix4 + hi1, sm);
        } else {
            // units 8 (chunk 0, tap 8) and 9 (chunk 1, tap 0): their planes are 2 apart only when the chunks are different groups
            const int cl = 18 * g0 + 16, ch = 18 * g1, base = cl < ch ? cl : ch;
            const int ml = 9 * g0 + 8, mh = 9 * g1, mbase = ml < mh ? ml : mh;
            const unsigned vo = c.pix4 + (unsigned)(hi ? ch - base : cl - base) * hw4, vm = c.pix4 + (unsigned)(hi ? mh - mbase : ml - mbase) * hw4;
            q.dy = buf_load(off_rs, vo, (unsigned)base * hw4);
            q.dx = buf_load(off_rs, vo, (unsigned)(base + 1) * hw4);
            q.m = buf_load(msk_rs, vm, (unsigned)mbase * hw4);
        }
    };
    // ---- sampling geometry of that k-step from its offsets; `s0` = x slot of the period's first chunk
    auto geometry = [&](Fwd4Geo& G, unsigned& flags, const Req& q, const Fwd4Ctx& c, int jj, int s0) {
        const int ul0 = 2 * jj, ul1 = 2 * jj + 1;
        const int t0 = f4_tap(ul0), t1 = f4_tap(ul1);
        const int ky0 = t0 / 3, kx0 = t0 % 3, ky1 = t1 / 3, kx1 = t1 % 3;
        // (c.Y0 + k is exact: the same float as float(oy - pad + k))
        const float ybase = c.Y0 + (ky0 == ky1 ? (float)ky0 : (hi ? (float)ky1 : (float)ky0));
        const float xbase = c.X0 + (kx0 == kx1 ? (float)kx0 : (hi ? (float)kx1 : (float)kx0));
        const float yr = ybase + q.dy, xr = xbase + q.dx;
        const float fy = floorf(yr), fx = floorf(xr);
        const int r0 = (int)fy - c.ty0, c0 = (int)fx - c.tx0;      // tile coordinates of the top-left corner
        const float ly = yr - fy, lx = xr - fx;
        // 2x2 footprint inside the tile <=> 0 <= r0 <= TR-2 and 0 <= c0 <= TC-2 (NaN / huge offsets fail the test)
        const bool in_tile = (unsigned)r0 < (unsigned)(TR - 1) && (unsigned)c0 < (unsigned)(TC - 1);
        float m = q.m;
        if (d.mask_logit) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));   // (uniform)
        const bool live = in_tile && c.px_ok;
        m = live ? m : 0.f;                                                // outside the tile: zero here, flagged for the fix-up pass
        flags |= (unsigned)(c.px_ok && !in_tile) << jj;
        // (24-bit multiply-add, computed unconditionally: a full-rate instruction, and nothing for hipcc to branch around)
        const unsigned pos_raw = __umul24((unsigned)r0, (unsigned)TC) + (unsigned)c0;
        const unsigned pos = in_tile ? pos_raw : 0u;
        // chunk slot of this lane's unit (uniform except at jj == 4, where the lane halves are in different chunks)
        const int sl0 = s0 + f4_cl(ul0), sl1 = s0 + f4_cl(ul1);
        const unsigned b0 = (unsigned)((sl0 >= 3 ? sl0 - 3 : sl0) * XSLOT * 16), b1 = (unsigned)((sl1 >= 3 ? sl1 - 3 : sl1) * XSLOT * 16);
        G.addr = pos * 16u + (f4_cl(ul0) == f4_cl(ul1) ? b0 : (hi ? b1 : b0));
        const float wy1 = ly * m;
        const f32x2v wy = {m - wy1, wy1};
        G.wt = wy * lx;
        G.wsd = wy - G.wt;
    };
    auto corners = [&](float4 (&c)[8], const Fwd4Geo& G) {
        const float4* q0 = reinterpret_cast<const float4*>(smem_raw + G.addr);
        c[0] = q0[0]; c[1] = q0[NPOS]; c[2] = q0[1]; c[3] = q0[NPOS + 1];
        c[4] = q0[TC]; c[5] = q0[NPOS + TC]; c[6] = q0[TC + 1]; c[7] = q0[NPOS + TC + 1];
    };
    auto blend = [&](float (&v)[8], const float4 (&c)[8], const Fwd4Geo& G) {
        const f32x2v p0 = blend4v(G.wsd, G.wt, lo2v(c[0]), lo2v(c[2]), lo2v(c[4]), lo2v(c[6]));
        const f32x2v p1 = blend4v(G.wsd, G.wt, hi2v(c[0]), hi2v(c[2]), hi2v(c[4]), hi2v(c[6]));
        const f32x2v p2 = blend4v(G.wsd, G.wt, lo2v(c[1]), lo2v(c[3]), lo2v(c[5]), lo2v(c[7]));
        const f32x2v p3 = blend4v(G.wsd, G.wt, hi2v(c[1]), hi2v(c[3]), hi2v(c[5]), hi2v(c[7]));
        v[0] = p0.x; v[1] = p0.y; v[2] = p1.x; v[3] = p1.y; v[4] = p2.x; v[5] = p2.y; v[6] = p3.x; v[7] = p3.y;
    };

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = zero16();
    auto mma = [&](int slot, const bf16x8& bh, const bf16x8& bl) {
        const bf16x8* w = ws + slot * WSLOT + hi * MP + lo;
        bf16x8 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = w[mt * 32];
            al[mt] = w[2 * MP + mt * 32];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(ah[mt], bh, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(ah[mt], bl, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(al[mt], bh, acc[mt]);
    };

    TSTAMP4(0);
    // =============================================================================== prologue (first tile of the workgroup)
    int tile = t_begin;
    Fwd4Ctx C, N;
    make_ctx(C, tile, 0);
    if (nper > 1) make_ctx(N, tile, 1);
    else make_ctx(N, tile + 1 < t_end ? tile + 1 : tile, 0);
    // (requests run THREE k-steps ahead of the geometry stage: the offset tensor streams from HBM -- 50 MB per batch element -- and
    // with two or three waves per SIMD nobody else covers a 2 us round trip per k-step)
    Req rq0, rq1, rq2, rqa, rqb, rqc;
    request(rq0, C, 0);
    request(rq1, C, 1);
    request(rq2, C, 2);
    request(rqa, C, 3);
    request(rqb, C, 4);
    request(rqc, C, 5);
    f32x4v xv[4];
    {
        f32x4v xw[4];
        x_load(xv, C, 0);
        x_load(xw, C, 1);
        w_issue(0, 0, 0, 5, C.mb, C.mb);
        w_issue(0, 0, 5, 2, C.mb, C.mb);
        for (int o = tid; o < nmb * MP; o += NW * 64) bias_s[o] = (p.bias != nullptr && o < d.Co) ? p.bias[o] : 0.f;
        x_write2(xv, 0, 0); x_write2(xv, 0, 1);
        x_write2(xw, 1, 0); x_write2(xw, 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the weight DMAs of this wave have landed)
    __syncthreads();
    x_load(xv, N, 0);       // the next period's first chunk: stored at the first barrier event
    TSTAMP4(1);

    Fwd4Geo gC, gN, gNN;                 // geometry of k-steps i, i+1, i+2
    unsigned flC = 0, flN = 0;           // out-of-tile flags of the current / the next period
    float4 cq[8];
    bf16x8 bhP, blP;                     // B fragments of k-step i-1
    int s0 = 0;                          // x slot of the period's first chunk
    {   // iteration 0 of the first period, without a predecessor to multiply
        Fwd4Geo g0;
        geometry(g0, flC, rq0, C, 0, 0);
        corners(cq, g0);
        geometry(gC, flC, rq1, C, 1, 0);
        float v[8];
        blend(v, cq, g0);
        corners(cq, gC);
        split8(v, bhP, blP);
        geometry(gN, flC, rq2, C, 2, 0);
    }
    int wslP = 0, wslC = 1;              // weight slots of k-steps i-1 and i

    // =============================================================================== the pipeline
    // One body = iterations 1 .. 8 of period C.per and iteration 0 of the next period ("jj == 9"): a tile's last product (k-step 8
    // of its last period) is multiplied in that final iteration, so the tile's epilogue sits at a body boundary.
    for (;;) {
        const int s0n = s0 + 2 >= 3 ? s0 - 1 : s0 + 2;           // (s0 + 2) % 3: x slot of the NEXT period's first chunk
        const int jbase = 9 * C.per;
#ifdef RVSR_TIMELINE_DCN4
        const bool tl = tile == t_begin + 1 && C.per < 4;     // the workgroup's second tile
        const int tb = 8 + 12 * C.per;
        if (tl) TSTAMP4(tb);
#endif
#pragma unroll
        for (int jj = 1; jj <= 9; ++jj) {
            if (prio_rot) {   // (uniform, kernel argument)
                constexpr int NG = NW / 4;          // waves per SIMD = age groups
                int pr = jj % NG + wgrp;             // the group that leads changes every iteration
                pr = pr >= NG ? pr - NG : pr;
                fwd4_setprio(pr);
            }
            // ---- V1(i): four-corner blend of k-step i
            float v[8];
            blend(v, cq, gC);
            // ---- barrier events: before the first corner read of the period's second chunk (jj == 3) / of the next period's first (jj == 8)
            if (jj == 3 || jj == 8) {
                // The weight DMAs of the previous event must have landed.  VMEM reads return in order and at least twelve offset / mask
                // requests (four k-steps) were issued after them, so "at most 9 operations outstanding" implies it -- without draining
                // the requests of the next three k-steps (vmcnt(0) exposes one HBM round trip per event: measured 2-5 K cycles).
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                if (!(dbg & 8)) __syncthreads();
                // weight slices of the k-steps that follow the NEXT event
                if (!(dbg & 1)) { if (jj == 3) w_issue(jbase + 3, wslC, 4, 4, C.mb, N.mb); else w_issue(jbase + 8, wslC, 3, 5, C.mb, N.mb); }
            }
            // ---- staging, trickled over the iterations after an event so that the eight waves' LDS stores and 16-byte loads do not
            // pile up behind the barrier: the registers hold the next period's first chunk at jj == 3 (slot s0 + 2), its second
            // chunk at jj == 8 (slot s0 + 3 = s0); they are refilled two iterations after their last store
            if (!(dbg & 2)) {
                if (jj == 3) x_write2(xv, s0n, 0);
                if (jj == 4) x_write2(xv, s0n, 1);
                if (jj == 8) x_write2(xv, s0, 0);
                if (jj == 9) x_write2(xv, s0, 1);
            }
            if (jj == 5 && !(dbg & 4)) x_load(xv, N, 1);
            // ---- L(i+1): corner reads of k-step i+1
            corners(cq, gN);
            // ---- V2(i): bf16 hi / lo split of k-step i
            bf16x8 bh, bl;
            split8(v, bh, bl);
            // ---- M(i-1): the matrix core runs k-step i-1
            mma(wslP, bhP, blP);
            // ---- G(i+2): geometry of k-step i+2, requests of k-step i+5
            {
                const int j2 = jj + 2, j5 = jj + 5;
                if (j2 < 9) geometry(gNN, flC, rqa, C, j2, s0); else geometry(gNN, flN, rqa, N, j2 - 9, s0n);
                rqa = rqb; rqb = rqc;
                if (j5 < 9) request(rqc, C, j5); else request(rqc, N, j5 - 9);
            }
            gC = gN; gN = gNN;
            bhP = bh; blP = bl;
            wslP = wslC;
            wslC = wslC + 1 == NWS ? 0 : wslC + 1;
#ifdef RVSR_TIMELINE_DCN4
            if (tl) TSTAMP4(tb + jj);
#endif
        }
        // ---- fix-up pass: samples of this period whose footprint left the tile (cold path; uniform branch)
        if (__any((int)(flC != 0))) {
            const int oy = C.y0 + wave, ox = C.x0 + lo;
            for (int jj = 0; jj < 9; ++jj) {
                const bool mine = (flC >> jj) & 1;
                if (!__any((int)mine)) continue;
                const int j = jbase + jj, u = 2 * j + hi, chunk = u / 9, t = u - 9 * chunk, g = chunk >> cpg8s;
                float vv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = 0.f;
                if (mine) {
                    const float* ob = d.offset + (size_t)C.b * d.off_bs + (size_t)(18 * g + 2 * t) * hw + (C.pix4 >> 2);
                    const float dy = ob[0], dx = ob[hw];
                    float m = d.mask[(size_t)C.b * d.mask_bs + (size_t)(9 * g + t) * hw + (C.pix4 >> 2)];
                    if (d.mask_logit) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));
                    const float y = (float)(oy - d.pad + t / 3) + dy, x = (float)(ox - d.pad + t % 3) + dx;
                    // the reference's rules spelled out (image coordinates; kernel.cu:467-497,618)
                    if (y > -1.f && x > -1.f && y < (float)d.H && x < (float)d.W) {
                        const float gy_ = floorf(y), gx_ = floorf(x);
                        const int yi = (int)gy_, xi = (int)gx_;
                        const float qy = y - gy_, qx = x - gx_, py = 1.f - qy, px = 1.f - qx;
                        const bool vy0 = yi >= 0, vy1 = yi + 1 <= d.H - 1, vx0 = xi >= 0, vx1 = xi + 1 <= d.W - 1;
                        const float u00 = (vy0 && vx0) ? py * px * m : 0.f, u01 = (vy0 && vx1) ? py * qx * m : 0.f;
                        const float u10 = (vy1 && vx0) ? qy * px * m : 0.f, u11 = (vy1 && vx1) ? qy * qx * m : 0.f;
                        const int cy0 = vy0 ? yi : 0, cy1 = vy1 ? yi + 1 : d.H - 1;
                        const int cx0 = vx0 ? xi : 0, cx1 = vx1 ? xi + 1 : d.W - 1;
                        const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
                        const float* pl = d.x + ((size_t)C.b * d.C + 8 * chunk) * HW;
                        float q00[8], q01[8], q10[8], q11[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {   // all 32 loads in flight together
                            const float* q = pl + (size_t)e * HW;
                            q00[e] = q[i00]; q01[e] = q[i01]; q10[e] = q[i10]; q11[e] = q[i11];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) vv[e] = u00 * q00[e] + u01 * q01[e] + u10 * q10[e] + u11 * q11[e];
                    }
                }
                bf16x8 fh, fl;
                split8(vv, fh, fl);
                const bf16x8* w = wpack + ((size_t)C.mb * nk + j) * WSLOT + hi * MP + lo;   // (weights of a cold k-step: straight from global memory)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 ah = w[mt * 32], al = w[2 * MP + mt * 32];
                    acc[mt] = mfma_bf16(ah, fh, acc[mt]);
                    acc[mt] = mfma_bf16(ah, fl, acc[mt]);
                    acc[mt] = mfma_bf16(al, fh, acc[mt]);
                }
            }
        }
        flC = flN; flN = 0;
        s0 = s0n;
#ifdef RVSR_TIMELINE_DCN4
        if (tl) TSTAMP4(tb + 10);
#endif
        const bool tile_done = C.per == nper - 1;   // (uniform)
        if (tile_done) {
            // ========================================================================== epilogue of the finished tile
            const int oy = C.y0 + wave, ox = C.x0 + lo;
            const __amdgpu_buffer_rsrc_t out_rs = buf_view(p.out + (size_t)C.b * d.Co * hw);
            const float* bias_m = bias_s + C.mb * MP;
            if (oy < d.Ho) {
                const float neg = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
                if ((d.Wo & 3) == 0 && (((uintptr_t)p.out) & 15) == 0) {   // (uniform)
                    // 16-byte stores: a 4x4 transpose inside every quad of lanes turns "lane = pixel, 4 registers = 4 consecutive
                    // channels" into "lane = channel, 4 consecutive pixels"
                    const int j = lo & 3, col4 = C.x0 + (lo & ~3);
                    const bool col_ok = col4 < d.Wo;   // Wo % 4 == 0: the float4 is entirely inside or outside
                    const unsigned lane_off = 4u * ((unsigned)(4 * hi + j) * (unsigned)hw + (unsigned)oy * d.Wo + col4);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            float r0 = acc[mt][4 * rg + 0], r1 = acc[mt][4 * rg + 1], r2 = acc[mt][4 * rg + 2], r3 = acc[mt][4 * rg + 3];
                            quad_transpose4(r0, r1, r2, r3, lo);
                            const int ol = mt * 32 + 8 * rg + 4 * hi + j;
                            const int o = C.mb * MP + ol;
                            const float bb = bias_m[ol];
                            float4 v = make_float4(r0 + bb, r1 + bb, r2 + bb, r3 + bb);
                            v.x = v.x > 0.f ? v.x : v.x * neg; v.y = v.y > 0.f ? v.y : v.y * neg;
                            v.z = v.z > 0.f ? v.z : v.z * neg; v.w = v.w > 0.f ? v.w : v.w * neg;
                            // (channel-group offset added on the vector side, not passed as the store's SGPR soffset: conv2_epilogue_v4 has the reason)
                            if (col_ok && o < d.Co) buf_store4(out_rs, lane_off + 4u * (unsigned)(C.mb * MP + mt * 32 + 8 * rg) * (unsigned)hw, 0u, v);
                        }
                    }
                } else {
                    const size_t pix = (size_t)oy * d.Wo + ox;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ol = mt * 32 + drow(r, hi);
                            const int o = C.mb * MP + ol;
                            const bool ok = ox < d.Wo && o < d.Co;
                            float v = acc[mt][r] + bias_m[ol];
                            v = v > 0.f ? v : v * neg;
                            if (ok) p.out[((size_t)C.b * d.Co + o) * hw + pix] = v;
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = zero16();
#ifdef RVSR_TIMELINE_DCN4
            if (tl) TSTAMP4(57);
#endif
            ++tile;
            if (tile >= t_end) break;
        }
        // ---- contexts of the next body: C <- N; N <- the period after it (the same tile, the next tile, or -- behind the last
        // tile -- the same context again: its look-ahead results are never used)
        C = N;
        if (C.per + 1 < nper) { N = C; N.per = C.per + 1; }
        else make_ctx(N, tile + 1 < t_end ? tile + 1 : tile, 0);
        // the staging registers were last stored from in iteration "jj == 9": refill them with the second period ahead's first chunk
        if (!(dbg & 4)) x_load(xv, N, 0);
    }
    TSTAMP4(58);
}

// ------------------------------------------------------------------------------------------ host side
// The weight image of the fourth generation ("layout 2"): packed[m-block][k-step j][hi | lo part][lane half h][MP rows][8] with
// A[o][8 h + c'] = w[o][8 (u / 9) + c'][u % 9], u = 2 j + h -- pack_weights_kernel mode 2 (bf16x3.h).
int rvsr_dcn_fwd4_geom(int Co, int C, int& mt, int& nk, int& nmb) {
    if (C % 16 != 0 || C < 16) return 0;
    mt = Co <= 32 ? 1 : 2;                  // (Co > 64: m-blocks of 64 output channels, the sampling work is repeated per m-block)
    nk = 9 * (C / 16);
    nmb = (Co + mt * 32 - 1) / (mt * 32);
    return 1;
}
size_t rvsr_dcn_fwd4_image_bytes(int Co, int C) {
    int mt, nk, nmb;
    if (!rvsr_dcn_fwd4_geom(Co, C, mt, nk, nmb)) return 0;
    return (size_t)nmb * nk * 4 * (mt * 32) * 16;
}

template <int NW, int RY, int RX, int MT>
static int launch_dcn_fwd4(const DcnFwdParams& p, const bf16x8* wpack, int cpg8s, hipStream_t st) {
    using F = Fwd4<NW, RY, RX, MT>;
    auto k = dcn_fwd4_kernel<NW, RY, RX, MT>;
    if (set_lds(k, F::LDS)) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd4: cannot reserve %zu B of LDS", F::LDS);
    const DcnGeom& d = p.d;
    const int nmb = (d.Co + MT * 32 - 1) / (MT * 32);
    if (nmb > 4) return RVSR_ERR_UNSUPPORTED;
    const long long ntiles = (long long)d.ntx * ((d.Ho + NW - 1) / NW) * nmb * d.B;
    if (ntiles >= (1ll << 30)) return RVSR_ERR_UNSUPPORTED;
    // persistent: one workgroup per CU walks a contiguous range of tiles (the pipeline runs across tile boundaries)
    static thread_local int ncu[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16) dev = 0;
    if (ncu[dev] == 0) {
        hipDeviceProp_t prop;
        ncu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const unsigned nwg = (unsigned)(ntiles < ncu[dev] ? ntiles : ncu[dev]);
    static const int prio = [] { const char* e = getenv("RVSR_DCN4_PRIO"); return e ? atoi(e) : 0; }();   // developer A/B switch
    static const int dbg = [] { const char* e = getenv("RVSR_DCN4_DBG"); return e ? atoi(e) : 0; }();     // timing ablations (WRONG results): 1 no weight DMA, 2 no x stores, 4 no x loads, 8 no barriers
    hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), F::LDS, st, p, wpack, cpg8s, (int)ntiles, prio, dbg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd4 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// can this call run on the fourth-generation kernel?  (everything else keeps the third generation and its weight image)
int rvsr_dcn_fwd4_supported(const DcnGeom& d) {
    static const int gen = [] { const char* e = getenv("RVSR_DCN_FWD"); return e ? atoi(e) : 3; }();   // developer A/B switch
    if (gen < 4) return 0;
    int mt, nk, nmb;
    if (!rvsr_dcn_fwd4_geom(d.Co, d.C, mt, nk, nmb)) return 0;
    if (d.stride != 1 || d.dil != 1 || d.pad != 1 || d.cpg % 8 != 0) return 0;
    const int cpg8 = d.cpg / 8;
    if (cpg8 & (cpg8 - 1)) return 0;
    if (d.W % 4 != 0 || (((uintptr_t)d.x) & 15) != 0 || d.H != d.Ho || d.W != d.Wo) return 0;
    // 32-bit byte offsets into one batch element's planes; x through a 2 GB view
    const size_t planes = (size_t)(d.C / d.cpg) * 18 > (size_t)(d.C > d.Co ? d.C : d.Co) ? (size_t)(d.C / d.cpg) * 18 : (size_t)(d.C > d.Co ? d.C : d.Co);
    if (planes * d.H * d.W * sizeof(float) >= ((size_t)1 << 32) || (size_t)d.C * d.H * d.W * sizeof(float) >= ((size_t)1 << 31)) return 0;
    return 1;
}

int rvsr_launch_dcn_fwd4(const DcnFwdParams& p, const void* wpack2, hipStream_t st) {
    const DcnGeom& d = p.d;
    int mt, nk, nmb;
    if (!rvsr_dcn_fwd4_supported(d) || !rvsr_dcn_fwd4_geom(d.Co, d.C, mt, nk, nmb)) return RVSR_ERR_UNSUPPORTED;
    int cpg8s = 0;
    while ((8 << cpg8s) < d.cpg) ++cpg8s;
    const bf16x8* wp = (const bf16x8*)wpack2;
    static const int nw = [] { const char* e = getenv("RVSR_DCN4_NW"); return e ? atoi(e) : 8; }();   // developer A/B switch
    if (mt == 1) return launch_dcn_fwd4<8, 5, 7, 1>(p, wp, cpg8s, st);
    if (nw == 12) return launch_dcn_fwd4<12, 5, 7, 2>(p, wp, cpg8s, st);   // (3 waves per SIMD: 168 registers, spills)
    return launch_dcn_fwd4<8, 5, 7, 2>(p, wp, cpg8s, st);
}
