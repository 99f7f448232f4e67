// dcn3_kernels.hip -- fused modulated-DCN forward, third generation (gfx950).
//
// Same decomposition as dcn_fwd2_kernel (dcn2_kernels.hip): out[Co, px] = W[Co, (tap, c)] * col[(tap, c), px] with the
// column values built by the lane the matrix core expects them from, never stored anywhere.  What changed follows
// the SQ counters of the second generation (profiles/r02_dcn_pmc.md): 188 VALU instructions per wave and k-step
// against 6 MFMAs, 43 % of the wave time parked in s_waitcnt/s_barrier around two serial staging round trips per
// 16-channel chunk.
//   * The LDS x tile is ZERO outside the image, so every validity rule of the reference's bilinear sampler
//     (kernel.cu:467-497: corners outside [0,H-1]x[0,W-1] contribute 0; whole sample 0 unless -1 < y < H, -1 < x < W,
//     :618) is implied by reading the tile: a sample whose 2x2 footprint lies inside the tile needs no per-corner
//     masks, clamps or range tests at all -- one unsigned compare per axis decides "inside the tile".
//   * Sampling runs in tile coordinates; the modulation mask is folded into the row weights (6 multiplies/subtracts
//     for the four corner weights instead of 4 + 8), sigmoid uses v_exp/v_rcp directly.
//   * Weight slice and x tile of a chunk are requested together (one L2/HBM round trip per chunk, not two).
//   * Samples that leave the tile (|offset| > R) still fall back to per-lane global gathers with the full rule set.
// Geometry: stride 1, dilation 1, groups of >= 8 channels (what EDVR / TDAN instantiate); everything else keeps the
// second-generation kernel.
#include "dcn_tile.h"

#ifdef RVSR_TIMELINE_DCN
__device__ unsigned long long rvsr_dbg_dcn3[256];
extern "C" int rvsr_debug_read_dcn3(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rvsr_dbg_dcn3), sizeof(unsigned long long) * 256); }
#define TSTAMP(i) do { if (blockIdx.x == 77 && blockIdx.z == 1 && threadIdx.x == 0) rvsr_dbg_dcn3[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do {} while (0)
#endif

// The x tile of a 16-channel chunk for an 8 x 32 pixel workgroup: 16 rows x 40 columns x 4 channel quads = 2560 float4
// items = 5 per thread of a 512-thread workgroup.  Item decode is shifts and masks only and four of the five items
// share one address (same position, consecutive quads): the staging of the second generation spent ~9 K cycles per
// chunk just ISSUING its loads (div/mod by 40 and 640, 64-bit address chains) next to co-resident waves in their
// VALU-bound tap loops.
//   items 0..3: position (row tid >> 5, column tid & 31), quad j
//   item  4   : position (row (tid >> 3) & 15, column 32 + (tid & 7)), quad tid >> 7
struct XTile16x40 {
    float v[5][4];
    bool inb0, inb4;
};
__device__ __forceinline__ void xtile16x40_load(XTile16x40& r, const DcnGeom& d, __amdgpu_buffer_rsrc_t x_rs, int c0, int ty0, int tx0, int tid) {
    // x_rs: raw buffer view of this batch element's planes; plane offsets are uniform (SGPR), lane offsets 32-bit bytes
    const unsigned HW4 = 4u * (unsigned)(d.H * d.W);
    {
        const int gy = ty0 + (tid >> 5), gx = tx0 + (tid & 31);
        r.inb0 = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        const int gyc = gy < 0 ? 0 : (gy >= d.H ? d.H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= d.W ? d.W - 1 : gx);
        const unsigned off = 4u * (unsigned)(gyc * d.W + gxc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + 4 * j + e;   // (uniform)
                r.v[j][e] = buf_load(x_rs, off, (unsigned)(c < d.C ? c : d.C - 1) * HW4);
            }
    }
    {
        const int gy = ty0 + ((tid >> 3) & 15), gx = tx0 + 32 + (tid & 7), q = tid >> 7;
        r.inb4 = gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
        const int gyc = gy < 0 ? 0 : (gy >= d.H ? d.H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= d.W ? d.W - 1 : gx);
        const int qe = c0 + 4 * q < d.C ? q : 0;   // (C % 4 == 0: a quad exists entirely or not at all)
        const unsigned off = 4u * (unsigned)(gyc * d.W + gxc) + (unsigned)(4 * qe) * HW4;
#pragma unroll
        for (int e = 0; e < 4; ++e) r.v[4][e] = buf_load(x_rs, off, (unsigned)(c0 + e) * HW4);
    }
}
// zero outside the image and beyond the last channel: the tap loop relies on it
__device__ __forceinline__ void xtile16x40_commit(float4* xt, const XTile16x40& r, const DcnGeom& d, int c0, int tid) {
    constexpr int TC = 40, NPOS = 16 * 40;
    const int p0 = (tid >> 5) * TC + (tid & 31);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int left = r.inb0 ? d.C - (c0 + 4 * j) : 0;
        xt[j * NPOS + p0] = make_float4(left > 0 ? r.v[j][0] : 0.f, left > 1 ? r.v[j][1] : 0.f, left > 2 ? r.v[j][2] : 0.f,
                                        left > 3 ? r.v[j][3] : 0.f);
    }
    const int q = tid >> 7, p4 = ((tid >> 3) & 15) * TC + 32 + (tid & 7);
    const int left = r.inb4 ? d.C - (c0 + 4 * q) : 0;
    xt[q * NPOS + p4] = make_float4(left > 0 ? r.v[4][0] : 0.f, left > 1 ? r.v[4][1] : 0.f, left > 2 ? r.v[4][2] : 0.f,
                                    left > 3 ? r.v[4][3] : 0.f);
}


typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lo2(const float4& a) { return f32x2{a.x, a.y}; }
__device__ __forceinline__ f32x2 hi2(const float4& a) { return f32x2{a.z, a.w}; }
// s.x * a00 + t.x * a01 + s.y * a10 + t.y * a11 on a pair of channels (VOP3P op_sel / op_sel_hi pick the broadcast half)
__device__ __forceinline__ f32x2 blend4(f32x2 s, f32x2 t, f32x2 a00, f32x2 a01, f32x2 a10, f32x2 a11) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(s), "v"(a00));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(r) : "v"(t), "v"(a01));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(r) : "v"(s), "v"(a10));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(r) : "v"(t), "v"(a11));
    return r;
}

// R: halo of the LDS x tile beyond the 3x3 footprint.  R = 3 is the kernel of round 2 (two workgroups per CU).  R = 7 / 11 (one
// workgroup per CU, a 24 x 48 / 32 x 56 tile) are launched next to it when the caller passes probe counters of the offsets
// (rvsr_launch_dcn_fwd3): every candidate returns at once unless the counters select it, so that px-scale offsets sample from LDS
// instead of gathering from global memory lane by lane (3 px mean |offset|: 0.19 -> see profiles/r03_notes.md).  pad + R is a
// multiple of 4 for all three: tile rows start on 16-byte boundaries and the large tiles are staged with 16-byte loads.
// TERMS: terms of the bf16 product (rvsr_common.h: gemm modes): 3 = hi*hi + hi*lo + lo*hi; 2 = without the weights' lo part (that half of
// the weight slice is not fetched); 1 = hi*hi (no lo part of the column values either)
// Timing ablations of scratch builds (tools/build_variant.sh dcn3_kernels <name> -DRVSR_ABL3=<bits>; results wrong by construction):
// 1 no MFMAs (nor weight-fragment reads), 4 no corner reads, 8 no far path, 16 no blend / split, 32 no offset / mask requests,
// 64 no staging (x tile loads + stores, weight DMA; the barriers stay)
#ifdef RVSR_ABL3
constexpr int ABL3 = RVSR_ABL3;
#else
constexpr int ABL3 = 0;
#endif
template <int MT, int R, int TERMS = 3>
__global__ __launch_bounds__(512, (MT <= 2 && R == 3) ? 4 : 2) void dcn_fwd3_kernel(const DcnFwdParams p, const bf16x8* __restrict__ wpack) {
    constexpr int TH = 8, NT = TH * 64;
    constexpr int TR = TH + 2 * R + 2, TC = 32 + 2 * R + 2, NPOS = TR * TC;
    constexpr int MP = MT * 32, WVEC = 9 * 2 * MP;  // 16-byte vectors per weight part
    constexpr int WPARTS = TERMS >= 3 ? 2 : 1;      // parts of the weight slice this kernel uses: [hi | lo] or hi only
    constexpr int NWV = (WPARTS * WVEC + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);           // [2 octets][2 halves][NPOS], zero outside the image
    bf16x8* ws_hi = reinterpret_cast<bf16x8*>(xt + 4 * NPOS);   // [9 taps][2 octets][MP]
    bf16x8* ws_lo = ws_hi + WVEC;
    float* bias_s = reinterpret_cast<float*>(ws_lo + WVEC);     // [MP]
    const DcnGeom& d = p.d;
    if (dcn_halo_not_selected(p.sel)) return;   // (uniform) not the halo the offsets of this call ask for
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    unsigned sbx, sby, sbz;
    swizzled_block(sbx, sby, sbz, d.swz);
    const int tx = sbx % d.ntx, ty = sbx / d.ntx;
    const int x0 = tx * 32, y0 = ty * TH, mb = sby, b = sbz;
    const int ty0 = y0 - d.pad - R, tx0 = x0 - d.pad - R;  // image coords of tile (0,0); stride 1
    const int nchunks = (d.C + 15) / 16;
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.Ho * d.Wo;
    const int oy = y0 + wave, ox = x0 + lo;
    const bool px_ok = oy < d.Ho && ox < d.Wo;
    const size_t pix = (size_t)oy * d.Wo + ox;
    // Sample positions are formed in IMAGE coordinates exactly as the reference does (float(h_in + i) + offset,
    // kernel.cu:594-616) so that floor() and the fractional weights round identically; only the integer corner is
    // moved into tile coordinates.
    const float by = (float)(oy - d.pad), bx = (float)(ox - d.pad);

    // raw buffer views of this batch element's offset / mask planes (rvsr_launch_dcn_fwd3 checks that they span < 4 GB)
    const __amdgpu_buffer_rsrc_t off_rs = buf_view(d.offset + (size_t)b * d.off_bs);
    const __amdgpu_buffer_rsrc_t msk_rs = buf_view(d.mask + (size_t)b * d.mask_bs);
    const __amdgpu_buffer_rsrc_t x_rs = buf_view(d.x + (size_t)b * d.C * HW), out_rs = buf_view(p.out + (size_t)b * d.Co * hw);
    const __amdgpu_buffer_rsrc_t x2g_rs = buf_view_2g(d.x + (size_t)b * d.C * HW);   // (large tiles: bit 31 of a lane offset = "reads zero")

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = zero16();

    TSTAMP(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * 16;
        const int cb8 = c0 + 8 * hi;              // first channel of this lane's octet
        const bool oct_ok = px_ok && cb8 < d.C;
        // offsets / mask: buffer loads with a uniform (SGPR) plane offset and a 32-bit per-lane byte offset -- no per-lane
        // 64-bit address arithmetic in the tap loop (three v_lshl_add_u64 per tap otherwise)
        const int g0 = c0 / d.cpg, dg = oct_ok ? cb8 / d.cpg - g0 : 0;
        const unsigned pixc = oct_ok ? (unsigned)pix : 0u;   // lanes without work read pixel 0 (loads stay unconditional)
        const unsigned off_lane = 4u * (pixc + (unsigned)(dg * 18) * (unsigned)hw), msk_lane = 4u * (pixc + (unsigned)(dg * 9) * (unsigned)hw);
        const unsigned off_pl = 4u * (unsigned)(g0 * 18) * (unsigned)hw, msk_pl = 4u * (unsigned)(g0 * 9) * (unsigned)hw, pl = 4u * (unsigned)hw;
        float n_dy = buf_load(off_rs, off_lane, off_pl), n_dx = buf_load(off_rs, off_lane, off_pl + pl), n_m = buf_load(msk_rs, msk_lane, msk_pl);
        if (ABL3 & 32) { n_dy = (float)lane * 1e-3f; n_dx = n_dy; n_m = 0.5f; }
        if (!(ABL3 & 64) || chunk == 0) {   // weight slice + x tile of the chunk: ALL global loads first, then the LDS writes (one round trip)
            // weight slice: LDS-DMA (global_load_lds_dwordx4: lane l of a wave lands at M0 + 16 l, so a linear copy needs
            // no registers, no ds_write and no wait before the barrier's)
            const bf16x8* src = wpack + ((size_t)mb * nchunks + chunk) * 2 * WVEC;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int e = tid + i * NT;
                if (e - lane + 63 < WPARTS * WVEC)   // (wave-uniform; WVEC is a multiple of 64)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + e),
                                                     (__attribute__((address_space(3))) void*)(ws_hi + e), 16, 0, 0);
            }
            if constexpr (R == 3) {
                static_assert(R != 3 || (TR == 16 && TC == 40 && NT == 512), "xtile16x40 is written for this tile");
                XTile16x40 xr;
                xtile16x40_load(xr, d, x_rs, c0, ty0, tx0, tid);
                TSTAMP(1 + 6 * chunk);
                TSTAMP(2 + 6 * chunk);
                xtile16x40_commit(xt, xr, d, c0, tid);
                TSTAMP(3 + 6 * chunk);
            } else {
                // item = (quad, row, group of 4 columns): four 16-byte loads (one per channel of the quad) land as the float4s of four
                // positions -- a renaming of registers, no shuffles.  Rows / column groups outside the image and channels beyond C read
                // zeros through the buffer range check (lane offset beyond the 2 GB view); W % 4 == 0 (launcher) keeps a group whole.
                constexpr int TC4 = TC / 4, NITEM = 4 * TR * TC4, NXI = (NITEM + NT - 1) / NT;
                static_assert(TC % 4 == 0, "tile rows are whole 16-byte groups");
                typedef float f32x4v __attribute__((ext_vector_type(4)));
                const unsigned HW4b = 4u * (unsigned)(d.H * d.W);
                f32x4v xv[NXI][4];
#pragma unroll
                for (int k = 0; k < NXI; ++k) {
                    const int it = tid + k * NT;
                    const int q = it / (TR * TC4), rem = it - q * (TR * TC4);
                    const int r = rem / TC4, g4 = rem - r * TC4;
                    const int gy = ty0 + r, gx = tx0 + 4 * g4;
                    const bool ok = it < NITEM && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W;
                    const unsigned off = ok ? 4u * (unsigned)(gy * d.W + gx) : 0x80000000u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = c0 + 4 * q + e;
                        xv[k][e] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(x2g_rs, (int)(c < d.C ? off : 0x80000000u),
                                                                                                  (int)((unsigned)(c < d.C ? c : 0) * HW4b), 0));
                    }
                }
#pragma unroll
                for (int k = 0; k < NXI; ++k) {
                    const int it = tid + k * NT;
                    if (it < NITEM) {
                        const int q = it / (TR * TC4), rem = it - q * (TR * TC4);
                        const int r = rem / TC4, g4 = rem - r * TC4;
                        float4* dst = xt + q * NPOS + r * TC + 4 * g4;
                        dst[0] = make_float4(xv[k][0].x, xv[k][1].x, xv[k][2].x, xv[k][3].x);
                        dst[1] = make_float4(xv[k][0].y, xv[k][1].y, xv[k][2].y, xv[k][3].y);
                        dst[2] = make_float4(xv[k][0].z, xv[k][1].z, xv[k][2].z, xv[k][3].z);
                        dst[3] = make_float4(xv[k][0].w, xv[k][1].w, xv[k][2].w, xv[k][3].w);
                    }
                }
            }
            if (chunk == 0 && tid < MP) {
                const int o = mb * MP + tid;
                bias_s[tid] = (p.bias != nullptr && o < d.Co) ? p.bias[o] : 0.f;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight DMA of this wave has landed
        __syncthreads();
        TSTAMP(4 + 6 * chunk);

        const float4* xq0 = xt + (2 * hi) * NPOS;  // channels cb8..cb8+3 (xq0[NPOS + pos]: cb8+4..cb8+7)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (chunk == 1) TSTAMP(60 + 5 * tap);
            const float dy = n_dy, dx = n_dx;
            float m = n_m;
            if (ABL3 & 32) { n_dy = (float)lane * 1e-3f; n_dx = n_dy; n_m = 0.5f; }
            else if (tap < 8) {  // (compile-time) prefetch the next tap's offsets/mask under this tap's math
                n_dy = buf_load(off_rs, off_lane, off_pl + (unsigned)(2 * tap + 2) * pl);
                n_dx = buf_load(off_rs, off_lane, off_pl + (unsigned)(2 * tap + 3) * pl);
                n_m = buf_load(msk_rs, msk_lane, msk_pl + (unsigned)(tap + 1) * pl);
            }
            if (d.mask_logit) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));  // (uniform)
            m = oct_ok ? m : 0.f;                                            // lanes without work contribute zeros
            const float yr = (by + (float)(tap / 3)) + dy, xr_ = (bx + (float)(tap % 3)) + dx;
            const float fy = floorf(yr), fx = floorf(xr_);
            const int r0 = (int)fy - ty0, s0 = (int)fx - tx0;   // tile coordinates of the top-left corner
            const float ly = yr - fy, lx = xr_ - fx;
            // 2x2 footprint inside the tile <=> 0 <= r0 <= TR-2 and 0 <= s0 <= TC-2 (NaN/huge offsets fail the test)
            const bool in_tile = (unsigned)r0 < (unsigned)(TR - 1) && (unsigned)s0 < (unsigned)(TC - 1);
            // corner weights (x mask) as two register pairs S = (w00, w10), T = (w01, w11); the four-corner blend of a
            // channel pair is then four packed ops with one weight broadcast each.  Spelled in assembly: left to itself the
            // compiler pairs the first channels ACROSS corners and spends 14 v_mov per tap rearranging registers.
            const float wy1 = ly * m;
            const f32x2 wy = {m - wy1, wy1};
            const f32x2 wt = wy * lx, wsd = wy - wt;
            const int pos = in_tile ? r0 * TC + s0 : 0;
            float4 a00, b00, a01, b01, a10, b10, a11, b11;
            if (ABL3 & 4) { a00 = make_float4(wsd.x, wsd.y, wt.x, wt.y); b00 = a01 = b01 = a10 = b10 = a11 = b11 = a00; }
            else {
                a00 = xq0[pos]; b00 = xq0[NPOS + pos]; a01 = xq0[pos + 1]; b01 = xq0[NPOS + pos + 1];
                a10 = xq0[pos + TC]; b10 = xq0[NPOS + pos + TC]; a11 = xq0[pos + TC + 1]; b11 = xq0[NPOS + pos + TC + 1];
            }
            if (chunk == 1) TSTAMP(61 + 5 * tap);
            const f32x2 p0 = blend4(wsd, wt, lo2(a00), lo2(a01), lo2(a10), lo2(a11));
            const f32x2 p1 = blend4(wsd, wt, hi2(a00), hi2(a01), hi2(a10), hi2(a11));
            const f32x2 p2 = blend4(wsd, wt, lo2(b00), lo2(b01), lo2(b10), lo2(b11));
            const f32x2 p3 = blend4(wsd, wt, hi2(b00), hi2(b01), hi2(b10), hi2(b11));
            float v[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
            if (ABL3 & 16) { v[0] = a00.x; v[1] = b00.x; v[2] = a01.x; v[3] = b01.x; v[4] = a10.x; v[5] = b10.x; v[6] = a11.x; v[7] = b11.x; }
            if (!(ABL3 & 8) && !in_tile && oct_ok) {
                // large offset: this lane gathers its corners from global memory, with the reference's rules spelled out
                // (image coordinates; kernel.cu:467-497,618)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
                const float y = yr, x = xr_;
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
                    const float* pl = d.x + ((size_t)b * d.C + cb8) * HW;
                    float q00[8], q01[8], q10[8], q11[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {   // all 32 loads in flight together (channel clamped, masked below)
                        const float* q = pl + (size_t)(cb8 + j < d.C ? j : 0) * HW;
                        q00[j] = q[i00]; q01[j] = q[i01]; q10[j] = q[i10]; q11[j] = q[i11];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        v[j] = cb8 + j < d.C ? u00 * q00[j] + u01 * q01[j] + u10 * q10[j] + u11 * q11[j] : 0.f;
                }
            }
            if (chunk == 1) TSTAMP(62 + 5 * tap);
            bf16x8 bh, bl;
            if (ABL3 & 16) { bh = __builtin_bit_cast(bf16x8, make_float4(v[0], v[1], v[2], v[3])); bl = __builtin_bit_cast(bf16x8, make_float4(v[4], v[5], v[6], v[7])); }
            else split8(v, bh, bl);
            if (ABL3 & 1) { acc[0][0] += (float)bh[0] + (float)bl[1]; continue; }
            if (chunk == 1) TSTAMP(63 + 5 * tap);
            bf16x8 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ah[mt] = ws_hi[(tap * 2 + hi) * MP + mt * 32 + lo];
                if (TERMS >= 3) al[mt] = ws_lo[(tap * 2 + hi) * MP + mt * 32 + lo];
            }
#ifdef RVSR_F3_PRIO   // (scratch variant, tools/build_variant.sh dcn3_kernels <name> -DRVSR_F3_PRIO: priority of the wave while it feeds the matrix core)
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(ah[mt], bh, acc[mt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) if (TERMS >= 2) acc[mt] = mfma_bf16(ah[mt], bl, acc[mt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) if (TERMS >= 3) acc[mt] = mfma_bf16(al[mt], bh, acc[mt]);
#ifdef RVSR_F3_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#ifdef RVSR_F3_SGB    // (scratch variant: one MFMA per five vector instructions of the next tap's geometry instead of hipcc's clusters)
#pragma unroll
            for (int k = 0; k < 3 * MT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);
            }
#endif
            if (chunk == 1) TSTAMP(40 + tap);
        }
        TSTAMP(5 + 6 * chunk);
        __syncthreads();
        TSTAMP(6 + 6 * chunk);
    }

    if (oy >= d.Ho) return;
    const float neg = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
    if ((d.Wo & 3) == 0 && (((uintptr_t)p.out) & 15) == 0) {   // (uniform)
        // 16-byte stores: a 4x4 transpose inside every quad of lanes turns "lane = pixel, 4 registers = 4 consecutive channels"
        // into "lane = channel, 4 consecutive pixels" (dword stores are store-issue-bound: 32 per lane, ~7.5 K cycles per tile)
        const int j = lo & 3, col4 = x0 + (lo & ~3);
        const bool col_ok = col4 < d.Wo;   // Wo % 4 == 0: the float4 is entirely inside or outside
        const unsigned lane_off = 4u * ((unsigned)(4 * hi + j) * (unsigned)hw + (unsigned)oy * d.Wo + col4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float r0 = acc[mt][4 * rg + 0], r1 = acc[mt][4 * rg + 1], r2 = acc[mt][4 * rg + 2], r3 = acc[mt][4 * rg + 3];
                quad_transpose4(r0, r1, r2, r3, lo);
                const int ol = mt * 32 + 8 * rg + 4 * hi + j;
                const int o = mb * MP + ol;
                const float bb = bias_s[ol];
                float4 v = make_float4(r0 + bb, r1 + bb, r2 + bb, r3 + bb);
                v.x = v.x > 0.f ? v.x : v.x * neg; v.y = v.y > 0.f ? v.y : v.y * neg;
                v.z = v.z > 0.f ? v.z : v.z * neg; v.w = v.w > 0.f ? v.w : v.w * neg;
                // (channel-group offset added on the vector side, not passed as the store's SGPR soffset: conv2_epilogue_v4 has the reason)
                if (col_ok && o < d.Co) buf_store4(out_rs, lane_off + 4u * (unsigned)(mb * MP + mt * 32 + 8 * rg) * (unsigned)hw, 0u, v);
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ol = mt * 32 + drow(r, hi);
                const int o = mb * MP + ol;
                const bool ok = ox < d.Wo && o < d.Co;
                const int oc = ok ? o : 0;
                float v = acc[mt][r] + bias_s[ol];
                v = v > 0.f ? v : v * neg;
                if (ok) p.out[((size_t)b * d.Co + oc) * hw + pix] = v;
            }
        }
    }
    TSTAMP(30);
}

template <int MT, int R>
static int launch_dcn_fwd3(const DcnFwdParams& p, const bf16x8* wpack, hipStream_t st) {
    constexpr int TH = 8, TR = TH + 2 * R + 2, TC = 32 + 2 * R + 2;
    const size_t lds = (size_t)16 * (4 * TR * TC + 2 * 9 * 2 * MT * 32) + sizeof(float) * MT * 32;
    auto k = dcn_fwd3_kernel<MT, R>;
    if constexpr (MT >= 2) {   // reduced-term products (gemm modes 2 / 3): the kernels of the nf64 / nf128 packs
        const int nt = rvsr_gemm_terms();
        if (nt == 2) k = dcn_fwd3_kernel<MT, R, 2>;
        if (nt == 1) k = dcn_fwd3_kernel<MT, R, 1>;
    }
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd3: cannot reserve %zu B of LDS", lds);
    const DcnGeom& d = p.d;
    dim3 grid(d.ntx * ((d.Ho + TH - 1) / TH), (d.Co + MT * 32 - 1) / (MT * 32), d.B);
    hipLaunchKernelGGL(k, grid, dim3(TH * 64), lds, st, p, wpack);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd3 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}
template <int MT>
static int launch_dcn_fwd3_halo(const DcnFwdParams& p, const bf16x8* wpack, int halo, hipStream_t st) {
    if (halo <= 3) return launch_dcn_fwd3<MT, 3>(p, wpack, st);
    if (halo <= 7) return launch_dcn_fwd3<MT, 7>(p, wpack, st);
    if constexpr (MT <= 2) return launch_dcn_fwd3<MT, 11>(p, wpack, st);   // (11 px + the 74 KB weight slice of MT = 4 exceed 160 KB)
    return launch_dcn_fwd3<MT, 7>(p, wpack, st);
}

// `wpack`: the image pack_weights_kernel(mode 0, CCG 1) wrote for (mt, nchunks, nmb) -- built by rvsr_launch_dcn_fwd2's caller.
// `probe` (nullable): the three device counters of dcn_offset_probe2_kernel for this call's offsets: the halo is then chosen on the device.
int rvsr_launch_dcn_fwd3(const DcnFwdParams& p_in, const void* wpack, int mt, hipStream_t st, const unsigned* probe, size_t nprobe, int halo_hint) {
    DcnFwdParams p = p_in;
    const DcnGeom& d = p.d;
    if (d.cpg % 8 != 0 || d.stride != 1 || d.dil != 1) return RVSR_ERR_UNSUPPORTED;
    // 32-bit byte offsets into one batch element's x / offset / output planes: larger frames take dcn_fwd2
    const size_t planes = (size_t)(d.C / d.cpg) * 18 > (size_t)(d.C > d.Co ? d.C : d.Co) ? (size_t)(d.C / d.cpg) * 18 : (size_t)(d.C > d.Co ? d.C : d.Co);
    if (planes * d.H * d.W * sizeof(float) >= ((size_t)1 << 32)) return RVSR_ERR_UNSUPPORTED;
    const bf16x8* wp = (const bf16x8*)wpack;
    static const int fixed = [] { const char* e = getenv("RVSR_DCN3_HALO"); return e ? atoi(e) : -1; }();   // developer A/B switch
    const bool big_ok = (d.W % 4 == 0) && ((((uintptr_t)d.x) & 15) == 0) && (size_t)d.C * d.H * d.W * sizeof(float) < ((size_t)1 << 31);
    p.sel = dcn_halo_always();
    int rc = RVSR_OK;
#define FWD3_DISPATCH(HALO)                                                        \
    do {                                                                           \
        if (mt == 1) rc = launch_dcn_fwd3_halo<1>(p, wp, HALO, st);                \
        else if (mt == 2) rc = launch_dcn_fwd3_halo<2>(p, wp, HALO, st);           \
        else rc = launch_dcn_fwd3_halo<4>(p, wp, HALO, st);                        \
    } while (0)
    if (fixed >= 0 || probe == nullptr || !big_ok) {
        // no counters: the caller's hint (a halo chosen on the host from an earlier statistic of this layer's offsets), else 3 px
        FWD3_DISPATCH(big_ok ? (fixed >= 0 ? fixed : (halo_hint > 0 ? halo_hint : 3)) : 3);
        return rc;
    }
    // The smallest tile that leaves (almost) no sample outside: a k-step in which ANY of a wave's 64 lanes left the tile pays the global
    // gather for all of them, and at one workgroup per CU the large tiles hide that latency worse than the small one -- measured
    // (profiles/r03_notes.md): a 7 px halo with 10 % of the samples outside is slower than the 3 px halo with 65 % outside.
    //   R = 3: < 8 % of the offset components beyond 3.5 px;  R = 7: else, < 1 % beyond 7.5 px (or no larger tile);  R = 11: the rest
    // (crossovers of the fixed-halo timings at offset std 1.25 / 2.5 / 3.75 / 6.25 px; functional.dcn_forward_halo applies the same rule
    // on the host to the counters of the previous step).
    const unsigned thr3 = (unsigned)(nprobe * 8 / 100) + 1, thr7 = (unsigned)(nprobe / 100) + 1;
    const bool has11 = mt <= 2;
    p.sel.probe = probe;
    p.sel.ge = -1; p.sel.lt = 1; p.sel.thr_lt = thr3;
    FWD3_DISPATCH(3);
    if (rc != RVSR_OK) return rc;
    p.sel.ge = 1; p.sel.thr_ge = thr3; p.sel.lt = has11 ? 3 : -1; p.sel.thr_lt = thr7;
    FWD3_DISPATCH(7);
    if (rc != RVSR_OK || !has11) return rc;
    p.sel.ge = 3; p.sel.thr_ge = thr7; p.sel.lt = -1; p.sel.ge2 = 1; p.sel.thr_ge2 = thr3;   // (a partition: not when R = 3 runs)
    FWD3_DISPATCH(11);
#undef FWD3_DISPATCH
    return rc;
}
