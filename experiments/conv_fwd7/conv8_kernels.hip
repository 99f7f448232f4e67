// conv7_kernels.hip -- 3x3 / stride-1 forward (and data gradient) on the 8 x 64 tile with the tile's OUTPUT BURST INSIDE THE MFMA SHADOW.
//
// conv_fwd5_kernel<2, false, 1, true> (conv2_kernels.hip) spends ~10 K of a tile's ~55 K cycles in its epilogue: all eight waves of the
// CU's one workgroup issue their 16 `buffer_store_dwordx4` at the same time, the CU's store path (and, with every CU in step, the HBM
// write bandwidth) drains them at 13-26 B/clk and the matrix pipe idles meanwhile (profiles/r03_notes.md, r04_notes.md).  Two source-level
// attempts to move the stores under the MFMAs failed on registers: the kernel sits at 223 of the 256 registers a wave has at two waves per
// SIMD, and hipcc spilled 95-139 of them as soon as half of a finished tile's accumulators stayed live through another pass.
//
// This kernel hand-allocates the one thing the compiler would not: the finished accumulators are PARKED IN 32 AGPRs (`v_accvgpr_write`,
// inline asm with "a" constraints; the build gives the kernel exactly 32 accumulation registers through the `amdgpu-agpr-alloc`
// function attribute -- hipcc_agpr.sh -- instead of hipcc's default half / half split of the register file), and the stage loop rotates
// the two 32-row m-blocks of the 64 output channels so that one of them is always final while the other one computes:
//
//   chunk 1 .. n-2 of a tile ("mid"):  both m-blocks, 12 MFMAs per tap (as conv_fwd5)
//   last chunk:   pass A  m-block 0, 6 MFMAs per tap + the staging slices      -> m-block 0 final, parked
//                 pass B  m-block 1, 6 MFMAs per tap + 8 epilogue slices of the parked m-block 0 (2 stores each)
//                                                                               -> m-block 1 final, parked
//   first chunk of the NEXT tile:
//                 pass A  m-block 0 (fresh accumulators) + 8 epilogue slices of the parked m-block 1
//                 pass B  m-block 1 (fresh accumulators) + the staging slices
//
// The split passes read the input fragments twice (LDS is ~40 % busy in this kernel).  In a split pass the other m-block's accumulators
// are dead (parked or not yet started) and only one m-block's weight fragments are held, so the slices' temporaries fit under the
// 224-register ceiling.  Arithmetic, operand order and the epilogue's operation order are those of conv_fwd5_kernel: results are
// bit-identical to it (tests/test_gpu_conv.py::test_fwd7_bit_identical_to_fwd5).
// Reference semantics: arch_util.py:121-139 (ResidualBlock_noBN), EDVR_arch.py:96-132 (the PCD conv stack).
#include "conv_common.h"

#include "bf16x3.h"

#ifdef RVSR_TIMELINE7   // (developer build: s_memtime stamps of one tile of workgroup 77, all eight waves; tools/conv7_timeline.py)
__device__ unsigned long long rvsr_dbg8[8 * 96];
extern "C" int rvsr_debug_read8(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rvsr_dbg8), sizeof(unsigned long long) * 8 * 96); }
#define STAMP7() do { if (dbg_on) { if (lane == 0 && dbg_i < 96) rvsr_dbg8[wave * 96 + dbg_i] = __builtin_amdgcn_s_memtime(); ++dbg_i; } } while (0)
#else
#define STAMP7() do {} while (0)
#endif

typedef unsigned u32x4_7 __attribute__((ext_vector_type(4)));
typedef float f32x4_7 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void half_swap7(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
}

// EPI: 0 plain store, 1 + residual, 4 gradient mask (p.act == 3): conv2_epilogue_wide's MODE
template <int EPI, int NT>
__global__ __launch_bounds__(512, 2) void conv_fwd8_kernel(const ConvFwdParams p) {
    constexpr int KS = 3, T = 9, PAD = 1, NW = 8, TH = NW, TW = 64, NTHR = NW * 64;
    constexpr int IH = TH + KS - 1, IW = TW + KS - 1;
    constexpr int MP = 64, NOCT = 2, NPOS = IH * IW, NX = NOCT * NPOS;
    constexpr int WVEC = T * NOCT * MP;  // 16-byte vectors per weight part (hi or lo)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* xs_base = reinterpret_cast<bf16x8*>(smem_raw);                // [2 buffers][hi|lo][NX]
    bf16x8* ws_base = xs_base + 2 * 2 * NX;                               // [2 buffers][hi|lo][WVEC]
    float* bias_base = reinterpret_cast<float*>(ws_base + 2 * 2 * WVEC);  // [4][MP]
    bf16x8* const sink = reinterpret_cast<bf16x8*>(bias_base + 4 * MP);   // write-only slot (conv_fwd5_kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
#ifdef RVSR_F7_PRIO   // (scratch variant: static priority for the second-dispatched half, which loses every issue arbitration by age)
    if (wave >= 4) __builtin_amdgcn_s_setprio(RVSR_F7_PRIO);
#endif
    const TView& va = p.in.a;
    const TView& vb = p.in.b;
    const int C1 = va.C, Ctot = va.C + vb.C;
    const int nchunks = (Ctot + 15) / 16;   // (>= 2: launcher)

    // persistent schedule of conv_fwd5_kernel: 8 contiguous item ranges, one per XCD
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

    // ---- staging (conv_fwd5_kernel's vector staging of a plain view): one item per thread = (octet, tile row, group of 4 pixels
    // starting at x0 - 4 + 4 g), eight aligned 16-byte loads; zero padding comes from the buffer range check
    constexpr int NG = TW / 4 + 2;
    __amdgpu_buffer_rsrc_t w_rs = buf_view_2g(p.wpack), xa_rs = buf_view_2g(va.p), xb_rs = buf_view_2g(va.p);
    constexpr int WPARTS = NT >= 3 ? 2 : 1;
    constexpr int NWV = (WPARTS * WVEC + NTHR - 1) / NTHR;
    int it_oc, it_sp, it_dst, it_s0, itile_mb = 0;
    auto item_geom = [&](const Tile& t) {
        const bool live = tid < NOCT * IH * NG;
        const int it = live ? tid : 0;
        const int oc = it / (IH * NG), rem = it - oc * (IH * NG);
        const int r = rem / NG, g = rem - r * NG;
        const int gy = t.y0 - PAD + r, gx = t.x0 - 4 + 4 * g;
        it_oc = oc;
        const bool ok = live && gy >= 0 && gy < va.Hv && gx >= 0 && gx < va.Wv;
        it_sp = ok ? 4 * (gy * va.Ws + gx) : (int)0x80000000;
        it_dst = live ? (oc * IH + r) * IW + 4 * g - 3 : -100;
        it_s0 = 4 * g - 3;
    };
    float vin[8][4];
    bf16x8 wv[NWV];
    // wsel: weight vector to fetch (>= NWV: none); xsel: channels 4 xsel .. 4 xsel + 3 of the item (>= 2: none)
    auto issue_loads = [&](int chunk, int wsel, int xsel) {
        const unsigned wbase = (unsigned)(((size_t)itile_mb * nchunks + chunk) * 2 * WVEC) * 16u;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            if (wsel != i && wsel >= 0) continue;
            const int e = tid + i * NTHR;
            const unsigned vo = e < WPARTS * WVEC ? (unsigned)tid * 16u : 0x80000000u;
            const u32x4_7 w = __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)vo, (int)(wbase + (unsigned)i * NTHR * 16u), 0);
            wv[i] = __builtin_bit_cast(bf16x8, w);
        }
        if (xsel >= 2) return;
        const int c0 = chunk * 16;
        const bool second = c0 >= C1;   // (uniform: C1 % 16 == 0 whenever there is a second input)
        const int Cb = second ? vb.C : va.C, cl0 = second ? c0 - C1 : c0;
        const unsigned hw4 = 4u * (unsigned)(va.Hs * va.Ws);
        const int lane_nch = Cb - cl0 - 8 * it_oc;
        const unsigned vo = (unsigned)it_sp + (unsigned)(8 * it_oc) * hw4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (xsel >= 0 && (j >> 2) != xsel) continue;
            const unsigned so = (unsigned)(cl0 + j) * hw4;
            const unsigned vo_j = j < lane_nch ? vo : 0x80000000u;
            const f32x4_7 q = __builtin_bit_cast(f32x4_7, __builtin_amdgcn_raw_buffer_load_b128(second ? xb_rs : xa_rs, (int)vo_j, (int)so, 0));
            vin[j][0] = q.x; vin[j][1] = q.y; vin[j][2] = q.z; vin[j][3] = q.w;
        }
    };
    // xsel: pixel of the item to publish (< 0: all, >= 4: none); wsel: weight vector (< 0: all, >= NWV: none)
    auto commit = [&](int buf, int wsel, int xsel) {
        bf16x8* xs_hi = xs_base + buf * 2 * NX;
        bf16x8* xs_lo = xs_hi + NX;
        bf16x8* ws = ws_base + buf * 2 * WVEC;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (xsel >= 4 || (xsel >= 0 && xsel != e)) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = vin[j][e];
            bf16x8 h8, l8;
            split8(v, h8, l8);
            const int s = it_s0 + e;
            const bool ok = it_dst + e >= 0 && s >= 0 && s < IW;
            bf16x8* const dh = ok ? xs_hi + it_dst + e : sink;
            bf16x8* const dl = ok ? xs_lo + it_dst + e : sink;
            *dh = h8;
            if (NT >= 2) *dl = l8;
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            if (wsel >= 0 && wsel != i) continue;
            const int e = tid + i * NTHR;
            *(e < WPARTS * WVEC ? ws + e : sink) = wv[i];
        }
    };
    Tile itile = tile_of(0);
    auto stage_tile = [&](int q, int k, int ch) {
        if (q >= Q) return;
        if (ch == 0) {
            itile = tile_of(k);
            itile_mb = itile.mb;
            item_geom(itile);
            const size_t hw = (size_t)va.Hs * va.Ws;
            xa_rs = buf_view_2g(va.p + (size_t)itile.b * va.C * hw);
            if (vb.C) xb_rs = buf_view_2g(vb.p + (size_t)itile.b * vb.C * hw);
            if (tid < MP) {
                const int o = itile.mb * MP + tid;
                bias_base[(k & 3) * MP + tid] = (p.bias != nullptr && o < p.Co) ? p.bias[o] : 0.f;
            }
        }
    };

    stage_tile(0, 0, 0);
    issue_loads(0, -1, -1);
    commit(0, -1, -1);
    stage_tile(1, 1 / nchunks, 1 % nchunks);
    issue_loads(1 % nchunks, -1, -1);
    __syncthreads();
    int k_nx2 = 2 / nchunks, c_nx2 = 2 % nchunks;      // stage q + 2

    // ---- the parked m-block (32 AGPRs: [n][16]) and the epilogue of the tile it belongs to
    float park[32];
    auto park_block = [&](const f32x16& a0, const f32x16& a1) {
        // (the accumulators were last written by the matrix core: inline asm is outside hipcc's hazard bookkeeping, so the wait states
        //  an XDL result needs before a vector read are spelled out once)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[r]) : "v"(a0[r]));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[16 + r]) : "v"(a1[r]));
        }
    };
    struct Epi {
        __amdgpu_buffer_rsrc_t out_rs, res_rs;
        unsigned lane_off;   // 0x80000000: this lane stores nothing (column / row outside the image, no pending tile)
        int bias_slot, o0;
    } ep;
    const unsigned HWo = (unsigned)(p.Hout * p.Wout), HW4 = 4u * HWo;
    const int jn = lane >> 4, Gn = lane & 15, j = lo & 3;
    const int src4 = 4 * (4 * Gn + jn);
    const float neg = (p.act == 0 || EPI == 4) ? 1.f : (p.act == 1 ? 0.f : p.slope);
    ep.out_rs = buf_view_2g(p.out1);
    ep.res_rs = buf_view_2g(p.out1);
    ep.lane_off = 0x80000000u;
    ep.bias_slot = 0;
    ep.o0 = 0;
    auto epi_setup = [&](int k) {
        const Tile t = tile_of(k);
        const int row = t.y0 + wave, col4 = t.x0 + 4 * Gn;
        ep.out_rs = buf_view_2g(p.out1 + (size_t)t.b * p.Co * HWo);
        ep.res_rs = buf_view_2g(EPI == 1 || EPI == 4 ? p.res + (size_t)t.b * p.Co * HWo : p.out1);
        ep.lane_off = (row < p.Hout && col4 < p.Wout) ? 4u * ((unsigned)jn * HWo + (unsigned)row * p.Wout + col4) : 0x80000000u;
        ep.bias_slot = (k & 3) * MP;
        ep.o0 = t.mb * MP;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        acc[m][0] = zero16();
        acc[m][1] = zero16();
    }
    // The parked m-block `m` leaves in nine steps, one per tap (conv2_epilogue_wide, one register group rg = 8 channels x 64 pixels per
    // two steps), software-pipelined so that no step waits for its own LDS-crossbar or memory results:
    //   step 2 rg    : the two stores of group rg - 1 | residual requests of group rg | left 32 pixels of group rg: AGPR -> quad
    //                  transpose -> + bias -> activation
    //   step 2 rg + 1: right 32 pixels likewise | half swap | lane permutation (8 ds_bpermute, consumed by the next step's stores)
    //   step 8       : the two stores of group 3
    // bias and activation run as packed pairs: act(v) = max(v, v * neg) for 0 <= neg <= 1 is bit for bit `v > 0 ? v : v * neg`
    typedef float f32x2_7 __attribute__((ext_vector_type(2)));
    float e_r[2][4];      // the group in flight: [pixel half][register]
    float4 e_res[2];      // its residual / mask values (requested two steps before their use)
    auto epi_store = [&](int m, int rg) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ob = ep.o0 + m * 32 + 8 * rg + 4 * s;
            const unsigned off = ob + jn < p.Co ? ep.lane_off : 0x80000000u;
            float4 v = make_float4(e_r[s][0], e_r[s][1], e_r[s][2], e_r[s][3]);
            if (EPI == 1) { v.x += e_res[s].x; v.y += e_res[s].y; v.z += e_res[s].z; v.w += e_res[s].w; }
            if (EPI == 4) {
                v.x *= e_res[s].x > 0.f ? 1.f : p.slope; v.y *= e_res[s].y > 0.f ? 1.f : p.slope;
                v.z *= e_res[s].z > 0.f ? 1.f : p.slope; v.w *= e_res[s].w > 0.f ? 1.f : p.slope;
            }
            buf_store4(ep.out_rs, off + (unsigned)ob * HW4, 0u, v);
        }
    };
    auto epi_step = [&](int m, int t, int from_acc) {
        if (t >= 2 && !(t & 1)) epi_store(m, (t >> 1) - 1);
        if (t >= 8) return;
        const int rg = t >> 1, part = t & 1;
        if ((EPI == 1 || EPI == 4) && part == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ob = ep.o0 + m * 32 + 8 * rg + 4 * s;
                const unsigned off = ob + jn < p.Co ? ep.lane_off : 0x80000000u;
                const f32x4_7 q = __builtin_bit_cast(f32x4_7, __builtin_amdgcn_raw_buffer_load_b128(ep.res_rs, (int)(off + (unsigned)ob * HW4), 0, 0));
                e_res[s] = make_float4(q.x, q.y, q.z, q.w);
            }
        }
        const float bb = bias_base[ep.bias_slot + m * 32 + 8 * rg + 4 * hi + j];
        float r[4];
        if (from_acc) {   // (m-block 0 at the tile's end: straight from the accumulators)
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = acc[0][part][4 * rg + e];
        } else {
            float t0, t1, t2, t3;
            asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7\n\ts_nop 1"
                         : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3)
                         : "a"(park[16 * part + 4 * rg + 0]), "a"(park[16 * part + 4 * rg + 1]), "a"(park[16 * part + 4 * rg + 2]),
                           "a"(park[16 * part + 4 * rg + 3]));
            r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
        }
        quad_transpose4(r[0], r[1], r[2], r[3], lo);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const f32x2_7 v = f32x2_7{r[e], r[e + 1]} + f32x2_7{bb, bb};
            const f32x2_7 w = v * f32x2_7{neg, neg};
            e_r[part][e] = __builtin_fmaxf(v.x, w.x);
            e_r[part][e + 1] = __builtin_fmaxf(v.y, w.y);
        }
        if (part == 0) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) half_swap7(e_r[0][e], e_r[1][e]);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                e_r[s][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, e_r[s][e])));
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    int q = 0;
#ifdef RVSR_TIMELINE7
    bool dbg_on = false;
    int dbg_i = 0;
#endif
    // one stage = one 16-channel chunk of one tile; KIND 0: first chunk of a tile, 1: a middle chunk, 2: the last chunk
    auto stage = [&](auto kind_c, int k) {
        constexpr int KIND = decltype(kind_c)::value;
        const int buf = q & 1;
        const bf16x8* xs_hi = xs_base + buf * 2 * NX;
        const bf16x8* xs_lo = xs_hi + NX;
        const bf16x8* ws_hi = ws_base + buf * 2 * WVEC;
        const bf16x8* ws_lo = ws_hi + WVEC;
        bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        // the staging slice of tap `tap` (conv_fwd5_kernel): taps 0-3 publish pixel `tap` of stage q+1's item, taps 4-8 publish weight
        // vector tap-4 and refill the freed registers with stage q+2
        auto staging = [&](int tap) {
            if (tap < 4) {
                commit(buf ^ 1, NWV, tap);
            } else {
                commit(buf ^ 1, tap - 4, 4);
                if (tap == 4) stage_tile(q + 2, k_nx2, c_nx2);
                issue_loads(c_nx2, tap - 4, tap < 6 ? tap - 4 : 2);
            }
        };
        // nine taps of the m-blocks [ML, MH) with `work(tap)` in their shadow
        auto taps = [&](auto ml_c, auto mh_c, auto&& work) {
            constexpr int ML = decltype(ml_c)::value, MH = decltype(mh_c)::value;
            auto fetch = [&](int tap, int slot) {
                const int dy = tap / KS, dx = tap % KS;
#pragma unroll
                for (int m = ML; m < MH; ++m) {
                    ah[slot][m] = ws_hi[(tap * NOCT + hi) * MP + m * 32 + lo];
                    if (NT >= 3) al[slot][m] = ws_lo[(tap * NOCT + hi) * MP + m * 32 + lo];
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int idx = (hi * IH + wave + dy) * IW + lo + 32 * n + dx;
                    bh[slot][n] = xs_hi[idx];
                    if (NT >= 2) bl[slot][n] = xs_lo[idx];
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                const int sl = tap & 1;
                STAMP7();
                if (tap + 1 < T) fetch(tap + 1, sl ^ 1);
#pragma unroll
                for (int m = ML; m < MH; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(ah[sl][m], bh[sl][n], acc[m][n]);
                if (NT >= 2) {
#pragma unroll
                    for (int m = ML; m < MH; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(ah[sl][m], bl[sl][n], acc[m][n]);
                }
                if (NT >= 3) {
#pragma unroll
                    for (int m = ML; m < MH; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[m][n] = mfma_bf16(al[sl][m], bh[sl][n], acc[m][n]);
                }
                work(tap);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        STAMP7();
        if constexpr (KIND == 0) {
            // m-block 1 of the previous tile is parked (nothing on the first tile: its lanes store nowhere).  Its nine steps ride on the
            // taps whose staging slice is light (taps 4-8: a weight vector + loads; taps 0-3 carry the bf16 split of the input pixels)
            taps(I0{}, I2{}, [&](int tap) { if (tap >= 4) epi_step(1, tap - 4, 0); staging(tap); });
        } else if constexpr (KIND == 3) {
            taps(I0{}, I2{}, [&](int tap) { if (tap >= 4) epi_step(1, tap + 1, 0); staging(tap); });
        } else {
            taps(I0{}, I2{}, staging);
        }
        if constexpr (KIND == 2) {
            epi_setup(k);
#pragma unroll
            for (int t = 0; t < 9; ++t) epi_step(0, t, 1);          // m-block 0: not hidden (its registers restart with the next tile)
            park_block(acc[1][0], acc[1][1]);                       // m-block 1: leaves under the next tile's first stage
#pragma unroll
            for (int m = 0; m < 2; ++m) { acc[m][0] = zero16(); acc[m][1] = zero16(); }
        }
        if (++c_nx2 == nchunks) { c_nx2 = 0; ++k_nx2; }
        ++q;
        STAMP7();
        __syncthreads();
        STAMP7();
    };
    for (int k = 0; k < ntile; ++k) {
#ifdef RVSR_TIMELINE7
        dbg_on = blockIdx.x == 77 && k == 2;
#endif
        stage(I0{}, k);
        stage(I3{}, k);
        for (int c = 3; c < nchunks; ++c) stage(I1{}, k);
        stage(I2{}, k);
    }
    // the last tile's m-block 1 is still parked
#pragma unroll
    for (int t = 0; t < 9; ++t) epi_step(1, t, 0);
}

int rvsr_launch_conv_fwd8(const ConvFwdParams& p, size_t lds, unsigned grid, hipStream_t st) {
    auto k = p.act == 3 ? conv_fwd8_kernel<4, 3> : (p.res != nullptr ? conv_fwd8_kernel<1, 3> : conv_fwd8_kernel<0, 3>);
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "conv_fwd8: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "conv_fwd8 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}
