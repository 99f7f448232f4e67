// dcn2_kernels.hip -- modulated deformable convolution, second generation (gfx950).
//
// Forward: out[Co, px] = W[Co, (tap, c)] * col[(tap, c), px] where col = mask * bilinear(x).  The
// column matrix is never stored anywhere -- not in HBM (reference: `columns`, 132.7 MB per call),
// not in LDS: the lane that the matrix core expects to hold B[k-octet][pixel] *computes* those 8
// values itself and feeds them straight into v_mfma_f32_32x32x16_bf16 (3-term bf16 split, see
// bf16x3.h).  With K ordered (tap, channel) one k-octet = the 8 channels of one deformable group
// slice at one tap, i.e. exactly the values that share one (dy, dx, mask) triple, so a lane does
// one sampling-geometry computation + 4 corner fetches of 8 channels per MFMA k-step.
//
//   workgroup   = TH waves = TH rows x 32 pixels of output, all Co (M tiles in registers)
//   K chunk     = 16 input channels: k-step = (tap, 16 channels); lane half hi picks the octet
//   LDS         = x tile for the chunk, channel-quad-major [octet][half][row][col][4] f32 with a
//                 halo of R pixels (corner fetch = 2 conflict-free ds_read_b128), + the packed
//                 bf16 hi/lo weight slice of the chunk (linear copy of the pre-packed image).
//   out-of-tile samples (|offset| > R) fall back to global loads for that lane -- correct for any
//   offset, fast for the common small ones.
//   offsets / mask are read exactly once, coalesced along W; sigmoid of the mask logits in-kernel.
#include "bf16x3.h"
#include "dcn_common.h"

#ifdef RVSR_TIMELINE_DCN
__device__ unsigned long long rvsr_dbg_dcn[256];
extern "C" int rvsr_debug_read_dcn(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rvsr_dbg_dcn), sizeof(unsigned long long) * 256); }
#define DSTAMP(i) do { if (blockIdx.x == 77 && blockIdx.z == 1 && threadIdx.x == 0) rvsr_dbg_dcn[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DSTAMP_W(i) do { if (blockIdx.x == 7 && blockIdx.z == 1 && threadIdx.x == 0) rvsr_dbg_dcn[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DSTAMP_W3(i) do { if (blockIdx.x == 7 && blockIdx.z == 1 && threadIdx.x == 0) rvsr_dbg_dcn[(i) + 20] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DSTAMP(i) do {} while (0)
#define DSTAMP_W(i) do {} while (0)
#define DSTAMP_W3(i) do {} while (0)
#endif

#include "dcn_tile.h"

template <int TH, int MT>
__global__ __launch_bounds__(TH * 64, MT <= 2 ? 4 : 2) void dcn_fwd2_kernel(const DcnFwdParams p, const bf16x8* __restrict__ wpack) {
    constexpr int NT = TH * 64;
    constexpr int TR = TH + 2 * D2_R + 2, TC = 32 + 2 * D2_R + 2, NPOS = TR * TC;
    constexpr int MP = MT * 32, WVEC = 9 * 2 * MP;  // 16-byte vectors per weight part
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);           // [2 octets][2 halves][NPOS]
    bf16x8* ws_hi = reinterpret_cast<bf16x8*>(xt + 4 * NPOS);   // [9 taps][2 octets][MP]
    bf16x8* ws_lo = ws_hi + WVEC;
    float* bias_s = reinterpret_cast<float*>(ws_lo + WVEC);  // [MP]
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    unsigned sbx, sby, sbz;
    swizzled_block(sbx, sby, sbz, d.swz);
    const int tx = sbx % d.ntx, ty = sbx / d.ntx;
    const int x0 = tx * 32, y0 = ty * TH, mb = sby, b = sbz;
    const int ty0 = y0 * d.stride - d.pad - D2_R, tx0 = x0 * d.stride - d.pad - D2_R;  // image coords of tile (0,0)
    const int nchunks = (d.C + 15) / 16;
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.Ho * d.Wo;
    const int oy = y0 + wave, ox = x0 + lo;
    const bool px_ok = oy < d.Ho && ox < d.Wo;
    const size_t pix = (size_t)oy * d.Wo + ox;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = zero16();

    DSTAMP(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * 16;
        const int cb8 = c0 + 8 * hi;              // first channel of this lane's octet
        const bool oct_ok = px_ok && cb8 < d.C;
        const int g = oct_ok ? cb8 / d.cpg : 0;
        // offsets / mask pointers (clamped to pixel 0 for lanes outside the image: loads stay unconditional);
        // tap 0's values are requested before the staging below, tap t+1's under tap t's math.  (Fetching all 9
        // taps up front was measured slower: +27 live registers -> spills at the 128-VGPR / 4-waves-per-SIMD point.)
        const size_t pixc = oct_ok ? pix : 0;
        const float* offp = d.offset + (size_t)b * d.off_bs + (size_t)(g * 18) * hw + pixc;
        const float* mskp = d.mask + (size_t)b * d.mask_bs + (size_t)(g * 9) * hw + pixc;
        float n_dy = offp[0], n_dx = offp[hw], n_m = mskp[0];
        {   // weight slice: linear copy of the pre-packed LDS image (hi block, then lo block)
            const bf16x8* src = wpack + ((size_t)mb * nchunks + chunk) * 2 * WVEC;
            constexpr int NWV = (2 * WVEC + NT - 1) / NT;
            bf16x8 wv[NWV];
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int e = tid + i * NT;
                wv[i] = src[e < 2 * WVEC ? e : 0];
            }
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int e = tid + i * NT;
                if (e < 2 * WVEC) ws_hi[e] = wv[i];
            }
            if (chunk == 0 && tid < MP) {
                const int o = mb * MP + tid;
                bias_s[tid] = (p.bias != nullptr && o < d.Co) ? p.bias[o] : 0.f;
            }
        }
        DSTAMP(1 + chunk * 5);
        stage_x_tile<NT, 4, TR, TC>(xt, d, b, c0, ty0, tx0, tid);
        DSTAMP(2 + chunk * 5);
        __syncthreads();
        DSTAMP(3 + chunk * 5);

        const float4* xq0 = xt + (2 * hi) * NPOS;  // channels cb8..cb8+3
        const float4* xq1 = xq0 + NPOS;            // channels cb8+4..cb8+7
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float dy = n_dy, dx = n_dx;
            float m = n_m;
            if (tap < 8) {  // (compile-time) prefetch the next tap's offsets/mask under this tap's math
                n_dy = offp[(size_t)(2 * tap + 2) * hw];
                n_dx = offp[(size_t)(2 * tap + 3) * hw];
                n_m = mskp[(size_t)(tap + 1) * hw];
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (oct_ok) {
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
                    if (r0 >= 0 && r1 < TR && s0 >= 0 && s1 < TC) {  // all four corners inside the LDS tile
                        const int p00 = r0 * TC + s0, p01 = r0 * TC + s1, p10 = r1 * TC + s0, p11 = r1 * TC + s1;
                        const float4 a00 = xq0[p00], b00 = xq1[p00], a01 = xq0[p01], b01 = xq1[p01];
                        const float4 a10 = xq0[p10], b10 = xq1[p10], a11 = xq0[p11], b11 = xq1[p11];
                        v[0] = w00 * a00.x + w01 * a01.x + w10 * a10.x + w11 * a11.x;
                        v[1] = w00 * a00.y + w01 * a01.y + w10 * a10.y + w11 * a11.y;
                        v[2] = w00 * a00.z + w01 * a01.z + w10 * a10.z + w11 * a11.z;
                        v[3] = w00 * a00.w + w01 * a01.w + w10 * a10.w + w11 * a11.w;
                        v[4] = w00 * b00.x + w01 * b01.x + w10 * b10.x + w11 * b11.x;
                        v[5] = w00 * b00.y + w01 * b01.y + w10 * b10.y + w11 * b11.y;
                        v[6] = w00 * b00.z + w01 * b01.z + w10 * b10.z + w11 * b11.z;
                        v[7] = w00 * b00.w + w01 * b01.w + w10 * b10.w + w11 * b11.w;
                    } else {  // large offset: fetch this lane's corners from global memory
                        const int i00 = cy0 * d.W + cx0, i01 = cy0 * d.W + cx1, i10 = cy1 * d.W + cx0, i11 = cy1 * d.W + cx1;
                        const float* pl = d.x + ((size_t)b * d.C + cb8) * HW;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (cb8 + j < d.C) {
                                const float* q = pl + (size_t)j * HW;
                                v[j] = w00 * q[i00] + w01 * q[i01] + w10 * q[i10] + w11 * q[i11];
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= m;
                }
            }
            bf16x8 bh, bl;
            split8(v, bh, bl);
            bf16x8 ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ah[mt] = ws_hi[(tap * 2 + hi) * MP + mt * 32 + lo];
                al[mt] = ws_lo[(tap * 2 + hi) * MP + mt * 32 + lo];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(ah[mt], bh, acc[mt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(ah[mt], bl, acc[mt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(al[mt], bh, acc[mt]);
        }
        DSTAMP(4 + chunk * 5);
        __syncthreads();
        DSTAMP(5 + chunk * 5);
    }

    if (oy >= d.Ho) return;
    const float neg = p.act == 0 ? 1.f : (p.act == 1 ? 0.f : p.slope);
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
    DSTAMP(30);
}

// ------------------------------------------------------------------------------------------
static void fwd2_geom(int Co, int C, int& mt, int& nchunks, int& nmb) {
    mt = Co <= 32 ? 1 : (Co <= 64 ? 2 : 4);
    nchunks = (C + 15) / 16;
    nmb = (Co + mt * 32 - 1) / (mt * 32);
}

static size_t fwd2_image_bytes(int Co, int C) {
    int mt, nchunks, nmb;
    fwd2_geom(Co, C, mt, nchunks, nmb);
    return (size_t)nmb * nchunks * 2 * 9 * 2 * (mt * 32) * 16;
}
size_t rvsr_dcn_fwd2_workspace_bytes(int Co, int C) { return fwd2_image_bytes(Co, C); }
template <int TH, int MT>
static int launch_dcn_fwd2(const DcnFwdParams& p, const bf16x8* wpack, hipStream_t st) {
    constexpr int TR = TH + 2 * D2_R + 2, TC = 32 + 2 * D2_R + 2;
    const size_t lds = (size_t)16 * (4 * TR * TC + 2 * 9 * 2 * MT * 32) + sizeof(float) * MT * 32;
    auto k = dcn_fwd2_kernel<TH, MT>;
    if (set_lds(k, lds)) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd2: cannot reserve %zu B of LDS", lds);
    const DcnGeom& d = p.d;
    dim3 grid(d.ntx * ((d.Ho + TH - 1) / TH), (d.Co + MT * 32 - 1) / (MT * 32), d.B);
    hipLaunchKernelGGL(k, grid, dim3(TH * 64), lds, st, p, wpack);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(RVSR_ERR_LAUNCH, "dcn_fwd2 launch: %s", hipGetErrorString(e));
    return RVSR_OK;
}

// the forward's weight image in caller-owned memory (+ its descriptor for rvsr_pack_weights_batched), see include/realvsr_hip.h
extern "C" size_t rvsr_dcn_pack_weights(const float* weight, int C, int Co, void* out, size_t out_bytes, long long* desc, void* stream) {
    int mt, nchunks, nmb;
    fwd2_geom(Co, C, mt, nchunks, nmb);
    const size_t need = rvsr_dcn_fwd2_workspace_bytes(Co, C);
    if (!weight || !out || out_bytes < need) return 0;
    const size_t total = (size_t)nmb * nchunks * 9 * 2 * (mt * 32);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, (bf16x8*)out, Co,
                       C, 9, mt * 32, 1, nchunks, nmb, 0);
    if (desc) {
        desc[0] = (long long)(uintptr_t)weight; desc[1] = (long long)(uintptr_t)out;
        desc[2] = Co; desc[3] = C; desc[4] = 9; desc[5] = mt * 32; desc[6] = 1; desc[7] = nchunks; desc[8] = nmb; desc[9] = 0;
    }
    if (desc) for (int i = 10; i < 20; ++i) desc[i] = 0;   // (second descriptor: the slot of the removed dcn_fwd4 image, kept zero for the ABI)
    return need;
}

int rvsr_launch_dcn_fwd2(const DcnFwdParams& p, void* workspace, size_t workspace_bytes, hipStream_t st, const unsigned* probe, size_t nprobe, int halo_hint) {
    const DcnGeom& d = p.d;
    if (d.cpg % 8 != 0) return RVSR_ERR_UNSUPPORTED;  // a k-octet must lie inside one deformable group
    int mt, nchunks, nmb;
    fwd2_geom(d.Co, d.C, mt, nchunks, nmb);
    const size_t need = rvsr_dcn_fwd2_workspace_bytes(d.Co, d.C);
    if (!workspace || workspace_bytes < need) FAIL(RVSR_ERR_WORKSPACE, "dcn forward: workspace %zu B < %zu B", workspace_bytes, need);
    const size_t total = (size_t)nmb * nchunks * 9 * 2 * (mt * 32);
    if (!p.prepacked) {
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.w, (bf16x8*)workspace, d.Co,
                           d.C, 9, mt * 32, 1, nchunks, nmb, 0);
    }
    const bf16x8* wp = (const bf16x8*)workspace;
    {   // third generation where it covers the geometry (stride 1, dilation 1); else the second
        const int rc = rvsr_launch_dcn_fwd3(p, workspace, mt, st, probe, nprobe, halo_hint);
        if (rc != RVSR_ERR_UNSUPPORTED) return rc;
    }
    if (mt == 1) return launch_dcn_fwd2<8, 1>(p, wp, st);
    if (mt == 2) return launch_dcn_fwd2<8, 2>(p, wp, st);
    return launch_dcn_fwd2<8, 4>(p, wp, st);
}

// ==========================================================================================
// Backward w.r.t. weight / bias, second generation: same GEMM as dcn_bwd_weight_kernel
// (gW[o, k] += sum_px gOut[o, px] * col[k, px], exact-f32 MFMA, deterministic partials) but the column
// tile is built from an LDS x tile (no dependent global gathers) by 8 waves instead of 4.
struct DcnBwdW2Params {
    DcnGeom d;
    TView g;
    float* part;   // [8P][Co][C][9]
    float* bpart;  // [8P][Co] or nullptr
    int P, nty;
    int gvec;      // g.p / g.act 16-byte aligned: the gradient tile may be staged with 16-byte loads
};

__global__ __launch_bounds__(512, 2) void dcn_bwdw2_kernel(const DcnBwdW2Params p) {
    constexpr int GP = 65, CP = 97, NT = 512;
    constexpr int TR = 4 + 2 * D2_R + 2, TC = 32 + 2 * D2_R + 2, NPOS = TR * TC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* xt = reinterpret_cast<float4*>(smem_raw);       // [2 quads][NPOS]
    float* gT = reinterpret_cast<float*>(xt + 2 * NPOS);    // [128][65]
    float* colT = gT + DCN_NPX * GP;                        // [128][97]; col 72 = 1 (bias), 73.. = 0
    const DcnGeom& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int mb = blockIdx.y, c0 = blockIdx.z * 8;
    const bool m1_live = mb * 64 + 32 < d.Co;
    const size_t HW = (size_t)d.H * d.W, hw = (size_t)d.Ho * d.Wo;
    const int g = c0 / d.cpg;

    for (int e = tid; e < DCN_NPX * (CP - DCN_KC); e += NT) {
        const int px = e / (CP - DCN_KC), j = e - px * (CP - DCN_KC);
        colT[px * CP + DCN_KC + j] = j == 0 ? 1.f : 0.f;
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = zero16();

    const int ntiles = d.B * p.nty * d.ntx;
    for (int tile = blockIdx.x; tile < ntiles; tile += p.P) {
        const bool st_ = tile == blockIdx.x + 2 * p.P;  // third tile of this workgroup
        if (st_) DSTAMP_W(160);
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
                float* dst = gT + (4 * pg) * GP + ol;
                dst[0] = ok4[i] ? g4[i].x * f0 : 0.f;
                dst[GP] = ok4[i] ? g4[i].y * f1 : 0.f;
                dst[2 * GP] = ok4[i] ? g4[i].z * f2 : 0.f;
                dst[3 * GP] = ok4[i] ? g4[i].w * f3 : 0.f;
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
                for (int j = 0; j < 4; ++j) gT[px * GP + ol + j] = v4[j];
            }
        } else {
#pragma unroll 2
            for (int e = tid; e < 64 * DCN_NPX; e += NT) {
                const int ol = e >> 7, px = e & 127;
                const int o = mb * 64 + ol;
                gT[px * GP + ol] = o < d.Co ? tview_get(p.g, b, o, y0 + (px >> 5), x0 + (px & 31)) : 0.f;
            }
        }
        if (st_) DSTAMP_W(161);
        stage_x_tile<NT, 2, TR, TC>(xt, d, b, c0, ty0, tx0, tid);
        if (st_) DSTAMP_W(162);
        __syncthreads();
        if (st_) DSTAMP_W(163);
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
        if (st_) DSTAMP_W(164);
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
            for (int j = 0; j < 8; ++j) colT[px * CP + j * 9 + tap] = v[j];
        }
        if (st_) DSTAMP_W(165);
        __syncthreads();
        if (st_) DSTAMP_W(166);
#pragma unroll 2
        for (int ks = 0; ks < 8; ++ks) {  // wave w: pixels 16w .. 16w+15
            const int px = wave * 16 + 2 * ks + hi;
            const float a0 = gT[px * GP + lo], a1 = gT[px * GP + 32 + lo];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const float bv = colT[px * CP + n * 32 + lo];
                acc[0][n] = mfma32(a0, bv, acc[0][n]);
                if (m1_live) acc[1][n] = mfma32(a1, bv, acc[1][n]);
            }
        }
        if (st_) DSTAMP_W(167);
        __syncthreads();
        if (st_) DSTAMP_W(168);
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

#include "dcn_bwdw4.inc"

// returns the number of partials written (8 * P), or -1 if the geometry is not covered
int rvsr_launch_dcn_bwdw2(const DcnGeom& d, const TView& g, float* part, float* bpart_or_null, int P, int nty, int gy, int gz,
                          hipStream_t st) {
    if (d.cpg % 8 != 0) return -1;
    DcnBwdW2Params p;
    p.d = d; p.g = g; p.part = part; p.bpart = bpart_or_null; p.P = P; p.nty = nty;
    p.gvec = ((((uintptr_t)g.p) | ((uintptr_t)g.act)) & 15) == 0;
    constexpr int TR = 4 + 2 * D2_R + 2, TC = 32 + 2 * D2_R + 2;
    if (rvsr_gemm_mode_now() != 1) {  // bf16 split products
        const size_t lds3 = (size_t)16 * 2 * TR * TC + (size_t)2 * (64 + 96) * 272;
        // (dcn_bwdw4 addresses 64 gOut planes, 27 offset / mask planes and 8 x planes with 32-bit byte offsets inside 2 GB buffer views)
        const bool spans_ok = (size_t)256 * d.Ho * d.Wo < ((size_t)1 << 31) && (size_t)32 * d.H * d.W < ((size_t)1 << 31);
        if (d.stride == 1 && d.dil == 1 && g.mode == 0 && (d.Wo & 3) == 0 && p.gvec && spans_ok) {
            const int nt = rvsr_gemm_terms();   // reduced-term products (gemm modes 2 / 3)
            auto k4 = nt == 2 ? dcn_bwdw4_kernel<2> : (nt == 1 ? dcn_bwdw4_kernel<1> : dcn_bwdw4_kernel<3>);
            if (set_lds(k4, lds3)) return -2;
            hipLaunchKernelGGL(k4, dim3(P, gy, gz), dim3(512), lds3, st, p);
            return 8 * P;
        }
        // (views / geometries dcn_bwdw4 does not take: the exact-f32 kernel below)
    }
    const size_t lds = (size_t)16 * 2 * TR * TC + sizeof(float) * (DCN_NPX * 65 + DCN_NPX * 97);
    if (set_lds(dcn_bwdw2_kernel, lds)) return -2;
    hipLaunchKernelGGL(dcn_bwdw2_kernel, dim3(P, gy, gz), dim3(512), lds, st, p);
    return 8 * P;
}

