// conv_fwd6 (round 3): conv_fwd5's 8 x 64 tile with one wave per SIMD; measured slower (0.59-0.60 vs 0.54 ms), kept here as the prototype.
// Cut out of realvsr_amd/csrc/conv2_kernels.hip in round 5; not built (it uses that file's helpers).

// ------------------------------------------------------------------------------------------
// conv_fwd6: the 8 x 64 tile of conv_fwd5<.., WIDE> with FOUR waves per workgroup -- one wave per SIMD, which then owns the SIMD's
// whole register file (512: accumulators of 2 rows x 64 pixels x 64 output channels = 128 registers, two fragment slots 96, the
// staging registers of two x items and nine weight vectors 100-164).  Why: with two waves per SIMD the matrix pipe retired one
// 32x32x16 MFMA per ~41 cycles inside the tap loop (floor 32; profiles/r03_notes.md) -- the two waves arbitrate for the pipe and
// for VALU issue, the younger one is starved and the stage ends when it does.  A single wave per SIMD has no partner to lose to,
// issues 24 MFMAs per tap on 8 independent accumulators, and re-reads a weight fragment once per 4 pixel tiles instead of per 2.
// Same LDS images, stage sequence (tile, 16-channel chunk) over two buffers, persistent XCD-contiguous schedule and epilogue
// (conv2_epilogue_wide) as conv_fwd5; plain vector-staged views only (launch_fwd5 decides).
// MEASURED (profiles/r03_notes.md): 0.59-0.60 ms per 40 x 64 x 180 x 320 launch against 0.54 ms for conv_fwd5<.., WIDE>, +6 ms per step -- the
// compiler-scheduled single wave stalls on its own LDS waits and barriers with nobody to cover for it (letting the scheduler work across taps, or
// asking for an MFMA / 2 VALU / LDS / VMEM interleave through sched_group_barrier, is slower still).  Kept behind RVSR_CONV_FWD6=1 as the
// measured prototype; not the default.
template <bool ACT_IN>
__global__ __launch_bounds__(256, 1) void conv_fwd6_kernel(const ConvFwdParams p) {
    constexpr int MT = 2, T = 9, PAD = 1, NW = 4, TH = 8, TW = 64, NTHR = NW * 64;
    constexpr int IH = TH + 2, IW = TW + 2, MP = MT * 32, NOCT = 2, NPOS = IH * IW, NX = NOCT * NPOS;
    constexpr int WVEC = T * NOCT * MP;
    constexpr int NG = TW / 4 + 2, NITEMS = NOCT * IH * NG;          // 360 x items of (octet, row, 4-pixel group)
    constexpr int NIT = (NITEMS + NTHR - 1) / NTHR;                   // 2 per thread
    constexpr int NWV = (2 * WVEC) / NTHR;                            // 9 weight vectors per thread
    static_assert((2 * WVEC) % NTHR == 0 && NIT == 2 && NWV == 9, "slice schedule below is written for these counts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* xs_base = reinterpret_cast<bf16x8*>(smem_raw);                // [2 buffers][hi|lo][NX]
    bf16x8* ws_base = xs_base + 2 * 2 * NX;                               // [2 buffers][hi|lo][WVEC]
    float* bias_base = reinterpret_cast<float*>(ws_base + 2 * 2 * WVEC);  // [4][MP]
    bf16x8* const sink = reinterpret_cast<bf16x8*>(bias_base + 4 * MP);   // write-only slot for lanes without a destination
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const TView& va = p.in.a;
    const TView& vb = p.in.b;
    const int C1 = va.C, Ctot = va.C + vb.C;
    const int nchunks = (Ctot + 15) / 16;

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
    __amdgpu_buffer_rsrc_t w_rs = buf_view_2g(p.wpack), xa_rs = buf_view_2g(va.p), xb_rs = buf_view_2g(va.p), act_rs = buf_view_2g(va.p);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4v __attribute__((ext_vector_type(4)));

    // ---- per-thread staging items (tile-independent LDS slots, tile-dependent source offsets)
    int it_oc[NIT], it_sp[NIT], it_dst[NIT], it_s0[NIT];
    auto item_geom = [&](const Tile& t) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it_raw = tid + i * NTHR;
            const bool live = it_raw < NITEMS;
            const int it = live ? it_raw : 0;
            const int oc = it / (IH * NG), rem = it - oc * (IH * NG);
            const int r = rem / NG, g = rem - r * NG;
            const int gy = t.y0 - PAD + r, gx = t.x0 - 4 + 4 * g;
            it_oc[i] = oc;
            // outside the image / no item: an offset beyond every buffer view, the load returns the zero padding
            it_sp[i] = live && gy >= 0 && gy < va.Hv && gx >= 0 && gx < va.Wv ? 4 * (gy * va.Ws + gx) : (int)0x80000000;
            it_dst[i] = live ? (oc * IH + r) * IW + 4 * g - 3 : -100;   // LDS slot of the group's first pixel
            it_s0[i] = 4 * g - 3;                                        // its column inside the 66-pixel row
        }
    };
    float vin[NIT][8][4];
    float ain[ACT_IN ? NIT : 1][8][4];
    bf16x8 wv[NWV];
    Tile itile = tile_of(0);
    int ld_chunk = 0;   // (tile, chunk) the next loads belong to
    auto issue_w = [&](int i) {
        const unsigned wbase = (unsigned)(((size_t)itile.mb * nchunks + ld_chunk) * 2 * WVEC) * 16u;
        wv[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (int)((unsigned)tid * 16u), (int)(wbase + (unsigned)i * NTHR * 16u), 0));
    };
    auto issue_x = [&](int i) {
        const int c0 = ld_chunk * 16;
        const bool second = c0 >= C1;                 // (uniform: C1 % 16 == 0 whenever there is a second input)
        const int Cb = second ? vb.C : va.C, cl0 = second ? c0 - C1 : c0;
        const unsigned hw4 = 4u * (unsigned)(va.Hs * va.Ws);
        const int lane_nch = Cb - cl0 - 8 * it_oc[i];
        const unsigned vo = (unsigned)it_sp[i] + (unsigned)(8 * it_oc[i]) * hw4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned so = (unsigned)(cl0 + j) * hw4;
            const unsigned vo_j = j < lane_nch ? vo : 0x80000000u;
            const f32x4v q = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(second ? xb_rs : xa_rs, (int)vo_j, (int)so, 0));
            vin[i][j][0] = q.x; vin[i][j][1] = q.y; vin[i][j][2] = q.z; vin[i][j][3] = q.w;
            if (ACT_IN) {
                const f32x4v a4 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(act_rs, (int)vo_j, (int)so, 0));
                ain[i][j][0] = a4.x; ain[i][j][1] = a4.y; ain[i][j][2] = a4.z; ain[i][j][3] = a4.w;
            }
        }
    };
    auto commit_x = [&](int buf, int i, int e) {
        bf16x8* xs_hi = xs_base + buf * 2 * NX;
        bf16x8* xs_lo = xs_hi + NX;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ACT_IN ? vin[i][j][e] * (ain[i][j][e] > 0.f ? 1.f : va.slope) : vin[i][j][e];
        bf16x8 h8, l8;
        split8(v, h8, l8);
        const int s = it_s0[i] + e;
        const bool ok = it_dst[i] + e >= 0 && s >= 0 && s < IW;   // (no item: slot -100; groups 0 and 17 straddle the tile's 66-pixel row)
        *(ok ? xs_hi + it_dst[i] + e : sink) = h8;
        *(ok ? xs_lo + it_dst[i] + e : sink) = l8;
    };
    auto commit_w = [&](int buf, int i) { ws_base[buf * 2 * WVEC + tid + i * NTHR] = wv[i]; };
    // geometry + bias of the tile whose chunk 0 is about to be loaded (bias of tile k lives in slot k & 3)
    auto stage_tile = [&](int k, int ch, bool real) {
        ld_chunk = ch;
        if (ch != 0) return;
        itile = tile_of(k);
        item_geom(itile);
        const size_t hw = (size_t)va.Hs * va.Ws;
        xa_rs = buf_view_2g(va.p + (size_t)itile.b * va.C * hw);
        if (vb.C) xb_rs = buf_view_2g(vb.p + (size_t)itile.b * vb.C * hw);
        if (ACT_IN) act_rs = buf_view_2g(va.act + (size_t)itile.b * va.C * hw);
        if (real && tid < MP) {   // (beyond the last stage the loads repeat tile 0: its bias slot may belong to a live tile)
            const int o = itile.mb * MP + tid;
            bias_base[(k & 3) * MP + tid] = (p.bias != nullptr && o < p.Co) ? p.bias[o] : 0.f;
        }
    };

    // ---- prologue: stage 0 into LDS buffer 0, stage 1 into the registers
    stage_tile(0, 0, true);
#pragma unroll
    for (int i = 0; i < NWV; ++i) issue_w(i);
#pragma unroll
    for (int i = 0; i < NIT; ++i) issue_x(i);
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) commit_x(0, i, e);
#pragma unroll
    for (int i = 0; i < NWV; ++i) commit_w(0, i);
    // (beyond the last stage the loads repeat an earlier (tile, chunk): unconditional on purpose, see conv_fwd5)
    int k_ld = 1 / nchunks, c_ld = 1 % nchunks;
    if (1 >= Q) { k_ld = 0; c_ld = 0; }
    stage_tile(k_ld, c_ld, 1 < Q);
#pragma unroll
    for (int i = 0; i < NWV; ++i) issue_w(i);
#pragma unroll
    for (int i = 0; i < NIT; ++i) issue_x(i);
    __syncthreads();
    int k_cur = 0, c_cur = 0;
    int k_nx2 = 2 / nchunks, c_nx2 = 2 % nchunks;      // stage q + 2

    f32x16 acc[2][MT][2];   // [row of the wave][M tile][left / right 32 pixels]
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[r2][m][0] = zero16(); acc[r2][m][1] = zero16(); }

    for (int q = 0; q < Q; ++q) {
        const int buf = q & 1;
        const bf16x8* xs_hi = xs_base + buf * 2 * NX;
        const bf16x8* xs_lo = xs_hi + NX;
        const bf16x8* ws_hi = ws_base + buf * 2 * WVEC;
        const bf16x8* ws_lo = ws_hi + WVEC;
        bf16x8 ah[2][MT], al[2][MT], bh[2][2][2], bl[2][2][2];
        auto fetch = [&](int tap, int slot) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[slot][m] = ws_hi[(tap * NOCT + hi) * MP + m * 32 + lo];
                al[slot][m] = ws_lo[(tap * NOCT + hi) * MP + m * 32 + lo];
            }
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int idx = (hi * IH + 2 * wave + r2 + dy) * IW + lo + 32 * h + dx;
                    bh[slot][r2][h] = xs_hi[idx];
                    bl[slot][r2][h] = xs_lo[idx];
                }
        };
        const bool in_range = q + 2 < Q;
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const int sl = tap & 1;
            if (tap + 1 < T) fetch(tap + 1, sl ^ 1);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc[r2][m][h] = mfma_bf16(term == 2 ? al[sl][m] : ah[sl][m], term == 1 ? bl[sl][r2][h] : bh[sl][r2][h], acc[r2][m][h]);
            // ---- the wave's own staging slices in the shadow of the MFMAs above:
            //   taps 0-3: split + publish pixel `tap` of both x items of stage q+1
            //   taps 4-8: publish two weight vectors of stage q+1, refill the freed registers with stage q+2
            if (tap < 4) {
                commit_x(buf ^ 1, 0, tap);
                commit_x(buf ^ 1, 1, tap);
            } else {
                const int i0 = 2 * (tap - 4);
                commit_w(buf ^ 1, i0);
                if (i0 + 1 < NWV) commit_w(buf ^ 1, i0 + 1);
                if (tap == 4) stage_tile(in_range ? k_nx2 : 0, in_range ? c_nx2 : 0, in_range);
                issue_w(i0);
                if (i0 + 1 < NWV) issue_w(i0 + 1);
                if (tap == 4) issue_x(0);
                if (tap == 5) issue_x(1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int k = k_cur;
        const bool last_chunk = c_cur == nchunks - 1;
        if (++c_cur == nchunks) { c_cur = 0; ++k_cur; }
        if (++c_nx2 == nchunks) { c_nx2 = 0; ++k_nx2; }
        if (last_chunk) {
            const Tile cur = tile_of(k);
            const float* bias_s = bias_base + (k & 3) * MP;
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                if (p.res != nullptr) conv2_epilogue_wide<MT, 1>(acc[r2], p, bias_s, cur.b, cur.mb * MP, cur.y0 + 2 * wave + r2, cur.x0, lo, hi);
                else conv2_epilogue_wide<MT, 0>(acc[r2], p, bias_s, cur.b, cur.mb * MP, cur.y0 + 2 * wave + r2, cur.x0, lo, hi);
#pragma unroll
                for (int m = 0; m < MT; ++m) { acc[r2][m][0] = zero16(); acc[r2][m][1] = zero16(); }
            }
        }
        __syncthreads();
    }
}
