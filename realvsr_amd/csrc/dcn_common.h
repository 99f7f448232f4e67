// dcn_common.h -- geometry / sampling helpers shared by dcn_kernels.hip (f32 MFMA) and dcn2_kernels.hip (bf16x3).
#pragma once
#include "rvsr_common.h"

#define DCN_CC 8      // input channels per K chunk
#define DCN_KC 72     // = DCN_CC * 9 column rows per chunk
#define DCN_NPX 128   // pixels per tile (4 rows x 32)

struct DcnGeom {
    const float* x;       // (B, C, H, W)
    const float* offset;  // (b * off_bs)[(g*18 + 2k + {0:dy,1:dx})][Ho][Wo]
    const float* mask;    // (b * mask_bs)[(g*9 + k)][Ho][Wo]
    size_t off_bs, mask_bs;
    int mask_logit;       // 1: mask holds logits, sigmoid applied here
    int B, C, H, W, Co, Ho, Wo;
    int stride, pad, dil, dg, cpg;
    int ntx;
    int swz;  // XCD-aware workgroup remap on/off
};

struct Samp {
    float w00, w01, w10, w11;  // bilinear corner weights (0 for corners outside the image)
    float ly, lx;
    int i00, i01, i10, i11;    // clamped plane indices (always safe to load)
    float m;                   // modulation mask
    bool inside;
    bool v00, v01, v10, v11;   // corner inside the image
};

// sampling geometry of tap k at output pixel (oy, ox) for deformable group g  (kernel.cu:594-618)
__device__ __forceinline__ Samp dcn_sample(const DcnGeom& d, int b, int g, int k, int oy, int ox) {
    const size_t hw = (size_t)d.Ho * d.Wo;
    const size_t p = (size_t)oy * d.Wo + ox;
    const float* ob = d.offset + (size_t)b * d.off_bs + (size_t)(g * 18 + 2 * k) * hw + p;
    const float dy = ob[0], dx = ob[hw];
    float m = d.mask[(size_t)b * d.mask_bs + (size_t)(g * 9 + k) * hw + p];
    if (d.mask_logit) m = 1.f / (1.f + __expf(-m));
    const float y = (float)(oy * d.stride - d.pad + (k / 3) * d.dil) + dy;
    const float x = (float)(ox * d.stride - d.pad + (k % 3) * d.dil) + dx;
    Samp s;
    s.m = m;
    s.inside = (y > -1.f) && (x > -1.f) && (y < (float)d.H) && (x < (float)d.W);
    s.w00 = s.w01 = s.w10 = s.w11 = 0.f;
    s.ly = s.lx = 0.f;
    s.i00 = s.i01 = s.i10 = s.i11 = 0;
    s.v00 = s.v01 = s.v10 = s.v11 = false;
    if (s.inside) {
        const float fy = floorf(y), fx = floorf(x);
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
        const float ly = y - fy, lx = x - fx, hy = 1.f - ly, hx = 1.f - lx;
        s.ly = ly;
        s.lx = lx;
        const bool vy0 = y0 >= 0, vy1 = y1 <= d.H - 1, vx0 = x0 >= 0, vx1 = x1 <= d.W - 1;
        const int cy0 = vy0 ? y0 : 0, cy1 = vy1 ? y1 : d.H - 1, cx0 = vx0 ? x0 : 0, cx1 = vx1 ? x1 : d.W - 1;
        s.i00 = cy0 * d.W + cx0;
        s.i01 = cy0 * d.W + cx1;
        s.i10 = cy1 * d.W + cx0;
        s.i11 = cy1 * d.W + cx1;
        s.v00 = vy0 && vx0;
        s.v01 = vy0 && vx1;
        s.v10 = vy1 && vx0;
        s.v11 = vy1 && vx1;
        s.w00 = s.v00 ? hy * hx : 0.f;
        s.w01 = s.v01 ? hy * lx : 0.f;
        s.w10 = s.v10 ? ly * hx : 0.f;
        s.w11 = s.v11 ? ly * lx : 0.f;
    }
    return s;
}


// Device-side choice among tile halos: every candidate kernel is launched and returns at once unless
//   (ge < 0 || probe[ge] >= thr_ge) && (lt < 0 || probe[lt] < thr_lt)
// probe[0..5] = sampled counts of offset components with |v| > 2.5 / 3.5 / 5.5 / 7.5 / 8.5 / 11.5 px (dcn_offset_probe2_kernel): the
// backward's windows (2 / 5 / 8 / 12 px) use counters 0, 2, 4, the forward's tiles (3 / 7 / 11 px) counters 1, 3, 5.  probe == nullptr: run.
#define DCN_PROBE_COUNTERS 6
//   (ge < 0 || probe[ge] >= thr_ge) && (ge2 < 0 || probe[ge2] >= thr_ge2) && (lt < 0 || probe[lt] < thr_lt)
// (two lower conditions: the candidates of a cascade "A while x < a, else B while y < b, else C" form a partition only if C also
// carries "x >= a" -- a heavy-tailed offset field can have y >= b with x < a, and A and C would both run)
struct DcnHaloSel {
    const unsigned* probe;
    int ge, lt, ge2;
    unsigned thr_ge, thr_lt, thr_ge2;
};
__device__ __forceinline__ bool dcn_halo_not_selected(const DcnHaloSel& s) {
    if (s.probe == nullptr) return false;
    if (s.ge >= 0 && s.probe[s.ge] < s.thr_ge) return true;
    if (s.ge2 >= 0 && s.probe[s.ge2] < s.thr_ge2) return true;
    if (s.lt >= 0 && s.probe[s.lt] >= s.thr_lt) return true;
    return false;
}
static inline DcnHaloSel dcn_halo_always() { DcnHaloSel s; s.probe = nullptr; s.ge = s.lt = s.ge2 = -1; s.thr_ge = s.thr_lt = s.thr_ge2 = 0; return s; }

struct DcnFwdParams {
    DcnGeom d;
    const float* w;     // (Co, C, 3, 3)
    const float* bias;  // nullable
    float* out;         // (B, Co, Ho, Wo)
    int act;
    float slope;
    int prepacked;      // the workspace already holds the packed weight image (rvsr_dcn_pack_weights / rvsr_pack_weights_batched)
    DcnHaloSel sel;     // halo selection on the device (dcn_fwd3_kernel<MT, R>)
};


// bf16x3 forward (dcn2_kernels.hip); returns RVSR_ERR_UNSUPPORTED if the geometry is not covered
size_t rvsr_dcn_fwd2_workspace_bytes(int Co, int C);
int rvsr_launch_dcn_fwd2(const DcnFwdParams& p, void* workspace, size_t workspace_bytes, hipStream_t st, const unsigned* probe = nullptr,
                         size_t nprobe = 0, int halo_hint = 0);
// third-generation forward (dcn3_kernels.hip): consumes the weight image rvsr_launch_dcn_fwd2 packs; stride 1, dilation 1
int rvsr_launch_dcn_fwd3(const DcnFwdParams& p, const void* wpack, int mt, hipStream_t st, const unsigned* probe = nullptr, size_t nprobe = 0,
                         int halo_hint = 0);
// the three-counter offset statistic both DCN directions select their tile halo from (dcn5_kernels.hip); returns the sample count
size_t rvsr_launch_dcn_offset_probe(const DcnGeom& d, unsigned* cnt, hipStream_t st);
int rvsr_launch_dcn_bwdw2(const DcnGeom& d, const TView& g, float* part, float* bpart_or_null, int P, int nty, int gy, int gz,
                          hipStream_t st);
// sixth-generation input / offset / mask gradient (dcn6_kernels.hip): dcn_bwdin5's window with a lane = (pixel, tap) layout and packed math;
// same calling protocol.  RVSR_ERR_UNSUPPORTED: the caller falls back to dcn_bwdin5.
size_t rvsr_dcn_bwdin6_workspace_bytes(int Co, int C);
int rvsr_launch_dcn_bwdin6(const DcnGeom& d, const float* weight, const TView& g, float* gx, float* goff, size_t goff_bs,
                           float* gmask, size_t gmask_bs, void* workspace, size_t workspace_bytes, hipStream_t st, int halo = -1,
                           const unsigned* probe_in = nullptr, void* agt = nullptr);
// sixth-generation weight / bias gradient (dcn6_kernels.hip): column values transposed by the matrix core, chunk-major persistent; gw / gb
// accumulated into.  RVSR_ERR_UNSUPPORTED: the caller falls back to dcn_bwdw4.
// Its gOut operand comes pre-transposed from dcn_bwdin6 (`agt`, rvsr_dcn_bwd6_agt_bytes): the two run as a pair.
size_t rvsr_dcn_bwdw6_workspace_bytes(int Co, int C);
size_t rvsr_dcn_bwd6_agt_bytes(int B, int Co, int Ho, int Wo);
int rvsr_launch_dcn_bwdw6(const DcnGeom& d, const void* agt, float* gw, float* gb, void* workspace, size_t workspace_bytes, hipStream_t st);
// fifth-generation input / offset / mask gradient (dcn5_kernels.hip): shared f64 LDS window; halo < 0 = chosen on the device
size_t rvsr_dcn_bwdin5_workspace_bytes(int Co, int C);
int rvsr_launch_dcn_bwdin5(const DcnGeom& d, const float* weight, const TView& g, float* gx, float* goff, size_t goff_bs,
                           float* gmask, size_t gmask_bs, void* workspace, size_t workspace_bytes, hipStream_t st, int halo = -1,
                           const unsigned* probe_in = nullptr);

