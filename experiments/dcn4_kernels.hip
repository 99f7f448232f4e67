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
//     and a ring of nine k-step weight slices in 160 KB: ONE barrier per chunk, placed where the next chunk's slot and slices are
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

// Timing ablations of scratch builds (tools/build_variant.sh dcn4_kernels <name> -DRVSR_ABL4=<bits>; results wrong by construction):
// 1 no MFMAs (nor weight-fragment reads), 2 weight fragments from registers, 4 no corner reads, 8 no geometry arithmetic, 16 no blend / split,
// 32 no offset / mask requests, 64 no barrier events (barriers, weight DMA, x staging)
#ifdef RVSR_ABL4
constexpr int ABL4 = RVSR_ABL4;
#else
constexpr int ABL4 = 0;
#endif

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

template <int NW, int RY, int RX, int MT>
struct Fwd4 {
    static constexpr int NT = NW * 64;
    static constexpr int TR = NW + 2 * RY + 2, TC = 32 + 2 * RX + 2, NPOS = TR * TC, TC4 = TC / 4;
    static constexpr int MP = MT * 32;
    static constexpr int XSLOT = 2 * NPOS;     // float4 per chunk slot: [2 quads][NPOS]
    static constexpr int NXS = 3;              // chunk slots
    static constexpr int WSLOT = 4 * MP;       // 16-byte vectors per k-step weight slice: [hi | lo part][lane half][MP]
    static constexpr int NWS = 9;              // weight slots: k-step j lives in slot j % 9 -- its index inside the period, a compile-time
                                               // constant at every place of the pipeline body
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

// The kernel's own parameter block: only what the pipeline reads, sizes premultiplied on the host.  (DcnFwdParams is 40 dwords; with two
// tile contexts alive the scalar registers spilled 63-68 values into vector lanes and parts of the uniform arithmetic -- the DMA
// addressing -- ended up on the vector ALU.)
struct Fwd4Params {
    const float* x;        // (B, C, H, W)
    const float* offset;   // offset planes of batch element 0
    const float* mask;
    const float* bias;     // nullable
    float* out;            // (B, Co, H, W)
    unsigned long long off_bs, mask_bs;   // elements between batch elements
    int H, W, C, Co;
    int ntx, nty, nmb, ntiles;
    unsigned hw4, HW4;     // bytes per offset / output plane (= 4 H W: stride 1, pad 1), per x plane
    unsigned mdelta;       // byte distance from the offset planes to the mask planes of the same batch element (one buffer view serves both)
    int mask_logit, act, swz;
    float slope;
    DcnHaloSel sel;
};

// Per-period context: everything a pipeline stage needs to know about the TILE its k-step belongs to.  Two of them are live --
// `C` for the period the loop body is in, `N` for the next one (the same tile, or the workgroup's next tile): which one a stage
// uses is a compile-time property of its position in the body, so a tile boundary costs no selects inside the pipeline.
struct Fwd4Ctx {
    float Y0, X0;            // (float)(oy - pad), (float)(ox - pad)
    unsigned pix4;           // byte offset of the lane's output pixel inside a plane (0 for lanes without work)
    unsigned x_voff;         // byte offset of the lane's x staging item inside batch element b (bit 31: reads zero)
    unsigned voff, vmsk;     // RUNNING byte offsets of the lane's next offset / mask request (advanced by every request: no scalar
                             // plane arithmetic in the pipeline)
    __amdgpu_buffer_rsrc_t rs;   // offset planes of batch element b (masks: + Fwd4Params::mdelta)
    unsigned long long wsrc;     // (uniform) address of the weight slices of this period: image of m-block mb, k-step 9 per
    bool px_ok;
    int ty0, tx0;            // image coordinates of tile (0, 0)
    int y0, x0, b, mb;
    int per;                 // period inside the tile
};

template <int NW, int RY, int RX, int MT, int CPG8S>   // CPG8S = log2(channels per deformable group / 8): 0 or 1
__global__ __launch_bounds__(NW * 64) void dcn_fwd4_kernel(const Fwd4Params d, const bf16x8* __restrict__ wpack) {
    using F = Fwd4<NW, RY, RX, MT>;
    constexpr int TR = F::TR, TC = F::TC, NPOS = F::NPOS, TC4 = F::TC4, MP = F::MP;
    constexpr int XSLOT = F::XSLOT, WSLOT = F::WSLOT, NWS = F::NWS, WPI = F::WPI;
    constexpr int PAD = 1;
    constexpr unsigned XB = (unsigned)(XSLOT * 16);   // bytes per x chunk slot
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* const xs = reinterpret_cast<float4*>(smem_raw);            // [3 slots][2 quads][NPOS], zero outside the image
    bf16x8* const ws = reinterpret_cast<bf16x8*>(xs + F::NXS * XSLOT); // [10 slots][part][half][MP]
    float* const bias_s = reinterpret_cast<float*>(ws + NWS * WSLOT);  // [nmb][MP]
    if (dcn_halo_not_selected(d.sel)) return;   // (uniform) not the kernel the offsets of this call ask for
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nper = d.C / 16, nk = 9 * nper;
    const int nty = d.nty, nmb = d.nmb, ntiles = d.ntiles;
    constexpr int cpg8s = CPG8S;
    const unsigned HW4 = d.HW4, hw4 = d.hw4;
    const size_t HW = (size_t)d.H * d.W, hw = HW;
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
    const unsigned hw4x2 = 2u * hw4, hw4x4 = 4u * hw4;
    const unsigned hi2 = hi ? hw4x2 : 0u, hi1 = hi ? hw4 : 0u;   // the hi lane half is one tap further: + 2 offset planes / + 1 mask plane

    // running request offsets at k-step 0 of period c.per: the lane's unit is u = 18 per + hi = (chunk 2 per, tap hi); unit (chunk, t)
    // reads offset planes 18 (chunk >> CPG8S) + 2 t (+1) and mask plane 9 (chunk >> CPG8S) + t
    auto start_period = [&](Fwd4Ctx& c) {
        const unsigned pm = (unsigned)(CPG8S == 0 ? 18 : 9) * (unsigned)c.per * hw4;   // (uniform)
        c.voff = c.pix4 + hi2 + 2u * pm;
        c.vmsk = c.pix4 + hi1 + pm;
        const unsigned long long a = (unsigned long long)(uintptr_t)(wpack + ((size_t)c.mb * nk + 9 * c.per) * WSLOT);
        c.wsrc = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32) |
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    };
    auto make_ctx = [&](Fwd4Ctx& c, int tile, int per) {
        int t = tile;
        const int tx = t % d.ntx; t /= d.ntx;
        const int ty = t % nty; t /= nty;
        c.mb = t % nmb; c.b = t / nmb;
        c.x0 = tx * 32; c.y0 = ty * NW;
        c.ty0 = c.y0 - PAD - RY; c.tx0 = c.x0 - PAD - RX;
        const int oy = c.y0 + wave, ox = c.x0 + lo;
        c.px_ok = oy < d.H && ox < d.W;
        c.pix4 = c.px_ok ? 4u * (unsigned)(oy * d.W + ox) : 0u;   // lanes without work read pixel 0 (loads stay unconditional)
        // Sample positions are formed in IMAGE coordinates exactly as the reference does (float(h_in + i) + offset,
        // kernel.cu:594-616) so that floor() and the fractional weights round identically
        c.Y0 = (float)(oy - PAD); c.X0 = (float)(ox - PAD);
        const int gy = c.ty0 + it_r, gx = c.tx0 + 4 * it_g4;
        const bool ok = tid < F::NITEM && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        c.x_voff = ok ? 4u * (unsigned)(gy * d.W + gx) + (unsigned)(4 * it_q) * HW4 : 0x80000000u;
        c.per = per;
        c.rs = buf_view(d.offset + (size_t)c.b * d.off_bs);
        start_period(c);
    };

    // ---- staging: global -> registers -> LDS, one chunk (8 channels) at a time
    auto x_load = [&](f32x4v (&xv)[4], const Fwd4Ctx& c, int cl) {   // chunk 2 c.per + cl of c's tile
        const __amdgpu_buffer_rsrc_t x2g_rs = buf_view_2g(d.x + (size_t)c.b * d.C * HW);   // bit 31 of a lane offset = "reads zero"
        const int chunk = 2 * c.per + cl;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            xv[e] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(x2g_rs, (int)c.x_voff, (int)((unsigned)(8 * chunk + e) * HW4), 0));
    };
    auto x_write2 = [&](const f32x4v (&xv)[4], unsigned xb, int half) {   // two of the item's four positions; xb = slot byte address
        if (xs_dst >= 0) {
            float4* dst = reinterpret_cast<float4*>(smem_raw + xb) + xs_dst;
            if (half == 0) {
                dst[0] = make_float4(xv[0].x, xv[1].x, xv[2].x, xv[3].x);
                dst[1] = make_float4(xv[0].y, xv[1].y, xv[2].y, xv[3].y);
            } else {
                dst[2] = make_float4(xv[0].z, xv[1].z, xv[2].z, xv[3].z);
                dst[3] = make_float4(xv[0].w, xv[1].w, xv[2].w, xv[3].w);
            }
        }
    };
    // Weight slices by LDS-DMA, two k-steps per call: waves 0 .. WPI-1 move the 1 KB pieces of k-step jl of c's period, waves WPI ..
    // 2 WPI - 1 those of k-step jl + 1 (`both` = false: only the first).  k-step j lives in slot j % 9, so with jl a compile-time
    // constant everything but the wave's own piece offset is an immediate: one vector add (source offset), one scalar add (M0).
    // Inline assembly on purpose: next to a __builtin_amdgcn_global_load_lds in flight hipcc waits vmcnt(0) at the use of EVERY
    // ordinary load (here: the offset / mask requests of each k-step), which drains the requests of the following k-steps once per
    // iteration.  The DMA has no register destination, so hiding it from the compiler's counters is safe (its waits can only
    // become conservative); completion is waited for explicitly before the next barrier event.
    const unsigned ws_base = (unsigned)(F::NXS * XSLOT * 16);   // LDS byte address of the weight ring (the dynamic LDS segment starts at 0)
    const int wq = wave / WPI;                                  // (uniform) which k-step of a pair this wave moves; >= 2: none
    const unsigned wconst = (unsigned)__builtin_amdgcn_readfirstlane((wq * WSLOT + (wave % WPI) * 64) * 16);
    const unsigned wv = 16u * (unsigned)lane + wconst;
    auto w_pair = [&](const Fwd4Ctx& c, int jl, bool both) {
        if (wq < (both ? 2 : 1)) {   // (uniform)
            const unsigned voff = wv + (unsigned)(jl * WSLOT * 16);
            const unsigned m0v = ws_base + (unsigned)(jl * WSLOT * 16) + wconst;
            unsigned keep;   // (M0 is a reserved register: saved and restored, not clobbered)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(c.wsrc), "s"(m0v) : "memory");
        }
    };

    // ---- offset / mask requests of k-step jj of c's period: three dword loads per lane
    struct Req { float dy, dx, m; };
    auto request = [&](Req& q, Fwd4Ctx& c, int jj) {
        if (ABL4 & 32) { q.dy = __builtin_bit_cast(float, c.voff) * 1e-30f; q.dx = q.dy; q.m = 0.5f; }
        else {
        q.dy = buf_load(c.rs, c.voff, 0u);
        q.dx = buf_load(c.rs, c.voff, hw4);
        q.m = buf_load(c.rs, c.vmsk, d.mdelta);
        }
        // advance to the lane's next unit (u + 2): four offset planes / two mask planes further, except where the step crosses from
        // the period's first chunk into its second and both are the same deformable group (CPG8S == 1: taps 7, 8 -> 0, 1: -14 / -7)
        const int ul0 = 2 * jj, ul1 = 2 * jj + 1;
        const bool back0 = CPG8S == 1 && ul0 < 9 && ul0 + 2 >= 9, back1 = CPG8S == 1 && ul1 < 9 && ul1 + 2 >= 9;
        if (!back0 && !back1) { c.voff += hw4x4; c.vmsk += hw4x2; }
        else {
            const unsigned bo = 0u - 14u * hw4, bm = 0u - 7u * hw4;
            const bool back = hi ? back1 : back0;
            c.voff += back ? bo : hw4x4;
            c.vmsk += back ? bm : hw4x2;
        }
    };
    // ---- sampling geometry of that k-step from its offsets; xb0 / xb1 = (uniform) LDS byte addresses of the period's two chunk slots
    auto geometry = [&](Fwd4Geo& G, unsigned& flags, const Req& q, const Fwd4Ctx& c, int jj, unsigned xb0, unsigned xb1) {
        if (ABL4 & 8) { G.addr = xb0 + 16u * (unsigned)lane; G.wsd = f32x2v{q.dy, q.dx}; G.wt = f32x2v{q.m, q.m}; return; }
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
        const unsigned b0 = f4_cl(ul0) ? xb1 : xb0, b1 = f4_cl(ul1) ? xb1 : xb0;
        G.addr = pos * 16u + (f4_cl(ul0) == f4_cl(ul1) ? b0 : (hi ? b1 : b0));
        const float wy1 = ly * m;
        const f32x2v wy = {m - wy1, wy1};
        G.wt = wy * lx;
        G.wsd = wy - G.wt;
    };
    auto corners = [&](float4 (&c)[8], const Fwd4Geo& G) {
        if (ABL4 & 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = make_float4(G.wsd.x, G.wsd.y, G.wt.x, G.wt.y);
            return;
        }
        const float4* q0 = reinterpret_cast<const float4*>(smem_raw + G.addr);
        c[0] = q0[0]; c[1] = q0[NPOS]; c[2] = q0[1]; c[3] = q0[NPOS + 1];
        c[4] = q0[TC]; c[5] = q0[NPOS + TC]; c[6] = q0[TC + 1]; c[7] = q0[NPOS + TC + 1];
    };
    auto blend = [&](float (&v)[8], const float4 (&c)[8], const Fwd4Geo& G) {
#ifdef RVSR_F4_SCALAR_BLEND   // (scratch variant: plain v_fma_f32 instead of the packed form -- packed f32 beside MFMAs is said to be expensive)
        {
            const float w00 = G.wsd.x, w10 = G.wsd.y, w01 = G.wt.x, w11 = G.wt.y;
            auto b1 = [&](float a00, float a01, float a10, float a11) {
                float r;
                asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(w00), "v"(a00));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(w01), "v"(a01));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(w10), "v"(a10));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(w11), "v"(a11));
                return r;
            };
            v[0] = b1(c[0].x, c[2].x, c[4].x, c[6].x); v[1] = b1(c[0].y, c[2].y, c[4].y, c[6].y);
            v[2] = b1(c[0].z, c[2].z, c[4].z, c[6].z); v[3] = b1(c[0].w, c[2].w, c[4].w, c[6].w);
            v[4] = b1(c[1].x, c[3].x, c[5].x, c[7].x); v[5] = b1(c[1].y, c[3].y, c[5].y, c[7].y);
            v[6] = b1(c[1].z, c[3].z, c[5].z, c[7].z); v[7] = b1(c[1].w, c[3].w, c[5].w, c[7].w);
            return;
        }
#endif
        if (ABL4 & 16) { v[0] = c[0].x; v[1] = c[1].x; v[2] = c[2].x; v[3] = c[3].x; v[4] = c[4].x; v[5] = c[5].x; v[6] = c[6].x; v[7] = c[7].x; return; }
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
        if (ABL4 & 1) { acc[0][0] += (float)bh[0] + (float)bl[1]; return; }
        const bf16x8* w = ws + slot * WSLOT + hi * MP + lo;
        bf16x8 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = (ABL4 & 2) ? bh : w[mt * 32];
            al[mt] = (ABL4 & 2) ? bl : w[2 * MP + mt * 32];
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
        w_pair(C, 0, true); w_pair(C, 2, true); w_pair(C, 4, true); w_pair(C, 6, false);   // k-steps 0 .. 6
        for (int o = tid; o < nmb * MP; o += NW * 64) bias_s[o] = (d.bias != nullptr && o < d.Co) ? d.bias[o] : 0.f;
        x_write2(xv, 0u, 0); x_write2(xv, 0u, 1);
        x_write2(xw, XB, 0); x_write2(xw, XB, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the weight DMAs of this wave have landed)
    __syncthreads();
    x_load(xv, N, 0);       // the next period's first chunk: stored at the first barrier event
    TSTAMP4(1);

    Fwd4Geo gC, gN, gNN;                 // geometry of k-steps i, i+1, i+2
    unsigned flC = 0, flN = 0;           // out-of-tile flags of the current / the next period
    float4 cq[8];
    bf16x8 bhP, blP;                     // B fragments of k-step i-1
    // (uniform) LDS byte addresses of the three x chunk slots in ring order: the period's chunks are in xa, xb; the next period's go
    // to xc and -- once the period's first chunk is used up -- xa
    unsigned xa = 0u, xb = XB, xc = 2u * XB;
    {   // iteration 0 of the first period, without a predecessor to multiply
        Fwd4Geo g0;
        geometry(g0, flC, rq0, C, 0, 0u, XB);
        corners(cq, g0);
        geometry(gC, flC, rq1, C, 1, 0u, XB);
        float v[8];
        blend(v, cq, g0);
        corners(cq, gC);
        split8(v, bhP, blP);
        geometry(gN, flC, rq2, C, 2, 0u, XB);
    }

    // =============================================================================== the pipeline
    // One body = iterations 1 .. 8 of period C.per and iteration 0 of the next period ("jj == 9"): a tile's last product (k-step 8
    // of its last period) is multiplied in that final iteration, so the tile's epilogue sits at a body boundary.
    for (;;) {
        const int jbase = 9 * C.per;
#ifdef RVSR_TIMELINE_DCN4
        const bool tl = tile == t_begin + 1 && C.per < 4;     // the workgroup's second tile
        const int tb = 8 + 12 * C.per;
        if (tl) TSTAMP4(tb);
#endif
#pragma unroll
        for (int jj = 1; jj <= 9; ++jj) {
            // ---- V1(i): four-corner blend of k-step i
            float v[8];
            blend(v, cq, gC);
            // ---- barrier events: before the first corner read of the period's second chunk (jj == 3) / of the next period's first (jj == 8)
            if (!(ABL4 & 64) && (jj == 3 || jj == 8)) {
                // The weight DMAs of the previous event must have landed.  VMEM reads return in order and at least twelve offset / mask
                // requests (four k-steps) were issued after them, so "at most 9 operations outstanding" implies it -- without draining
                // the requests of the next three k-steps (vmcnt(0) exposes one HBM round trip per event: measured 2-5 K cycles).
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                __syncthreads();
                // weight slices of the k-steps that follow the NEXT event: 7, 8 of this period and 0, 1 of the next / 2 .. 6 of the next
                if (jj == 3) { w_pair(C, 7, true); w_pair(N, 0, true); }
                else { w_pair(N, 2, true); w_pair(N, 4, true); w_pair(N, 6, false); }
            }
            // ---- staging, trickled over the iterations after an event so that the eight waves' LDS stores and 16-byte loads do not
            // pile up behind the barrier: the registers hold the next period's first chunk at jj == 3 (slot xc), its second
            // chunk at jj == 8 (slot xa); they are refilled two iterations after their last store
            if (!(ABL4 & 64)) {
                if (jj == 3) x_write2(xv, xc, 0);
                if (jj == 4) x_write2(xv, xc, 1);
                if (jj == 8) x_write2(xv, xa, 0);
                if (jj == 9) x_write2(xv, xa, 1);
                if (jj == 5) x_load(xv, N, 1);
            }
            // ---- L(i+1): corner reads of k-step i+1
            corners(cq, gN);
            // ---- V2(i): bf16 hi / lo split of k-step i
            bf16x8 bh, bl;
            if (ABL4 & 16) { bh = __builtin_bit_cast(bf16x8, make_float4(v[0], v[1], v[2], v[3])); bl = __builtin_bit_cast(bf16x8, make_float4(v[4], v[5], v[6], v[7])); }
            else {
#ifdef RVSR_F4_NODOT   // (scratch variant: lo = v - float(hi) through a shift / mask and a subtraction instead of v_dot2c_f32_bf16)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    const bf2 h = {(__bf16)v[j], (__bf16)v[j + 1]};
                    const unsigned hp = __builtin_bit_cast(unsigned, h);
                    const float l0 = v[j] - __builtin_bit_cast(float, hp << 16), l1 = v[j + 1] - __builtin_bit_cast(float, hp & 0xffff0000u);
                    bh[j] = h[0]; bh[j + 1] = h[1];
                    bl[j] = (__bf16)l0; bl[j + 1] = (__bf16)l1;
                }
#else
                split8(v, bh, bl);
#endif
            }
            // ---- M(i-1): the matrix core runs k-step i-1
            mma(jj - 1, bhP, blP);
            // ---- G(i+2): geometry of k-step i+2, requests of k-step i+5
            {
                const int j2 = jj + 2, j5 = jj + 5;
                if (j2 < 9) geometry(gNN, flC, rqa, C, j2, xa, xb); else geometry(gNN, flN, rqa, N, j2 - 9, xc, xa);
                rqa = rqb; rqb = rqc;
                if (j5 < 9) request(rqc, C, j5); else request(rqc, N, j5 - 9);
            }
#ifdef RVSR_F4_SGB   // (scratch variant: ask the scheduler for one MFMA per RVSR_F4_SGB vector instructions (+ 2 LDS reads) instead of its clusters)
#pragma unroll
            for (int k = 0; k < 3 * MT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, RVSR_F4_SGB, 0);
#ifdef RVSR_F4_SGB_DS
                __builtin_amdgcn_sched_group_barrier(0x100, RVSR_F4_SGB_DS, 0);
#endif
            }
#endif
            gC = gN; gN = gNN;
            bhP = bh; blP = bl;
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
                    const float y = (float)(oy - PAD + t / 3) + dy, x = (float)(ox - PAD + t % 3) + dx;
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
        { const unsigned t = xa; xa = xc; xc = xb; xb = t; }   // (xa, xb, xc) <- (xc, xa, xb)
#ifdef RVSR_TIMELINE_DCN4
        if (tl) TSTAMP4(tb + 10);
#endif
        const bool tile_done = C.per == nper - 1;   // (uniform)
        if (tile_done) {
            // ========================================================================== epilogue of the finished tile
            const int oy = C.y0 + wave, ox = C.x0 + lo;
            const __amdgpu_buffer_rsrc_t out_rs = buf_view(d.out + (size_t)C.b * d.Co * hw);
            const float* bias_m = bias_s + C.mb * MP;
            if (oy < d.H) {
                const float neg = d.act == 0 ? 1.f : (d.act == 1 ? 0.f : d.slope);
                if ((((uintptr_t)d.out) & 15) == 0) {   // (uniform; W % 4 == 0 is a launch condition)
                    // 16-byte stores: a 4x4 transpose inside every quad of lanes turns "lane = pixel, 4 registers = 4 consecutive
                    // channels" into "lane = channel, 4 consecutive pixels"
                    const int j = lo & 3, col4 = C.x0 + (lo & ~3);
                    const bool col_ok = col4 < d.W;   // W % 4 == 0: the float4 is entirely inside or outside
                    const unsigned lane_off = 4u * ((unsigned)(4 * hi + j) * (unsigned)hw + (unsigned)oy * d.W + col4);
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
                    const size_t pix = (size_t)oy * d.W + ox;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ol = mt * 32 + drow(r, hi);
                            const int o = C.mb * MP + ol;
                            const bool ok = ox < d.W && o < d.Co;
                            float v = acc[mt][r] + bias_m[ol];
                            v = v > 0.f ? v : v * neg;
                            if (ok) d.out[((size_t)C.b * d.Co + o) * hw + pix] = v;
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
        if (C.per + 1 < nper) { N = C; N.per = C.per + 1; start_period(N); }
        else make_ctx(N, tile + 1 < t_end ? tile + 1 : tile, 0);
        // the staging registers were last stored from in iteration "jj == 9": refill them with the second period ahead's first chunk
        if (!(ABL4 & 64)) x_load(xv, N, 0);
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

template <int NW, int RY, int RX, int MT, int CPG8S>
static int launch_dcn_fwd4(const DcnFwdParams& p, const bf16x8* wpack, hipStream_t st) {
    using F = Fwd4<NW, RY, RX, MT>;
    auto k = dcn_fwd4_kernel<NW, RY, RX, MT, CPG8S>;
    if (set_lds(k, F::LDS)) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd4: cannot reserve %zu B of LDS", F::LDS);
    const DcnGeom& d = p.d;
    const int nmb = (d.Co + MT * 32 - 1) / (MT * 32);
    if (nmb > 4) return RVSR_ERR_UNSUPPORTED;
    const int nty = (d.Ho + NW - 1) / NW;
    const long long ntiles = (long long)d.ntx * nty * nmb * d.B;
    if (ntiles >= (1ll << 30)) return RVSR_ERR_UNSUPPORTED;
    Fwd4Params q;
    q.x = d.x; q.offset = d.offset; q.mask = d.mask; q.bias = p.bias; q.out = p.out;
    q.off_bs = d.off_bs; q.mask_bs = d.mask_bs;
    q.H = d.H; q.W = d.W; q.C = d.C; q.Co = d.Co;
    q.ntx = d.ntx; q.nty = nty; q.nmb = nmb; q.ntiles = (int)ntiles;
    q.hw4 = 4u * (unsigned)(d.H * d.W); q.HW4 = q.hw4;
    q.mdelta = (unsigned)((d.mask - d.offset) * (ptrdiff_t)sizeof(float));
    q.mask_logit = d.mask_logit; q.act = p.act; q.swz = d.swz; q.slope = p.slope;
    q.sel = p.sel;
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
    hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), F::LDS, st, q, wpack);
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
    if (d.cpg != 8 && d.cpg != 16) return 0;
    // one buffer view serves the offset and the mask planes of a batch element (the fused offset/mask tensor, or two tensors that
    // happen to lie within 4 GB of each other)
    if (d.mask < d.offset || d.off_bs != d.mask_bs) return 0;
    if ((size_t)(d.mask - d.offset) * sizeof(float) + (size_t)(d.C / d.cpg) * 9 * d.H * d.W * sizeof(float) >= ((size_t)1 << 32)) return 0;
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
    const bf16x8* wp = (const bf16x8*)wpack2;
    if (d.cpg == 8) return mt == 1 ? launch_dcn_fwd4<8, 5, 7, 1, 0>(p, wp, st) : launch_dcn_fwd4<8, 5, 7, 2, 0>(p, wp, st);
    return mt == 1 ? launch_dcn_fwd4<8, 5, 7, 1, 1>(p, wp, st) : launch_dcn_fwd4<8, 5, 7, 2, 1>(p, wp, st);
}
