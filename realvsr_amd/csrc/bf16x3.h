// bf16x3.h -- helpers for the 3-term bf16 split GEMM (see conv2_kernels.hip header comment).
#pragma once
#include "rvsr_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// First MFMA of an accumulator (SrcC = 0).  hipcc selects the untied form, `v_mfma ... vdst, a, b, 0`, and -- when an operand dies at this
// instruction -- allocates vdst ON TOP of that operand's registers (seen in the listings: `v_mfma_f32_32x32x16_bf16 v[98:113], v[110:113],
// v[150:153], 0`).  The matrix core reads the k = 8..15 half of the operands passes after the k = 0..7 half and by then may have begun writing
// vdst: the products of the upper lane half came out wrong, differently from run to run (round 5, profiles/r05_notes.md; tools/
// check_mfma_overlap.py scans the library for the pattern).  The empty asm keeps `a` and `b` alive across the instruction, so the allocator
// cannot overlap them with the result; it costs nothing.
__device__ __forceinline__ f32x16 mfma_bf16_first(bf16x8 a, bf16x8 b) {
    f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero16(), 0, 0, 0);
    asm volatile("" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
// hi = bf16(v) (round to nearest even), lo = bf16(v - hi).  The exact residual v - hi comes from v_dot2c_f32_bf16 on the PACKED
// hi pair -- D = v + <(hi0, hi1), (-1, 0)> -- so the pair is never unpacked to f32 again: 4 v_cvt_pk + 8 v_dot2c + 4 v_cvt_pk = 16
// vector instructions per 8 values, against 24 for the shift / mask / subtract form hipcc makes of `v - (float)hi`.  hi0 * -1 and
// the sum are exact (Sterbenz), so the result is bit-identical to that form (tools/micro/split_check.hip: 16.7 M random values) unless
// the OTHER value of a pair rounds to +-inf in bf16 (|v| > 3.39e38: 0 * inf).  The two selector constants are kept out of the
// inline-constant encoder on purpose: as literals this hipcc emits `1.0` for the packed pair (1, 0), which the hardware reads as (0, 1).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    unsigned u0 = 0x0000bf80u, u1 = 0xbf800000u;   // (-1, 0) and (0, -1) as packed bf16
    asm volatile("" : "+s"(u0), "+s"(u1));
    const bf16x2 m0 = __builtin_bit_cast(bf16x2, u0), m1 = __builtin_bit_cast(bf16x2, u1);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const bf16x2 h = {(__bf16)v[j], (__bf16)v[j + 1]};
        const float l0 = __builtin_amdgcn_fdot2_f32_bf16(h, m0, v[j], false);
        const float l1 = __builtin_amdgcn_fdot2_f32_bf16(h, m1, v[j + 1], false);
        hi[j] = h[0];
        hi[j + 1] = h[1];
        lo[j] = (__bf16)l0;
        lo[j + 1] = (__bf16)l1;
    }
}

// 4x4 transpose inside every quad of lanes: on entry lane j (= lane & 3) holds r[i] = M[i][j]; on exit it holds
// r[i] = M[j][i].  Two butterfly stages over DPP quad_perm (lane ^ 1, lane ^ 2): 12 VALU ops per block.  Used to turn
// the MFMA D layout (lane = pixel column, registers = 4 consecutive channels) into "lane = channel, registers = 4
// consecutive pixels", so the epilogue can issue 16-byte stores (dword stores are store-issue-bound: ~3.5 B/clk/CU).
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose4(float& r0, float& r1, float& r2, float& r3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    {   // stage 1: exchange across lane ^ 1 within register pairs (0,1) and (2,3)
        const float x01 = b0 ? r0 : r1, y01 = dpp_xor1(x01);
        const float x23 = b0 ? r2 : r3, y23 = dpp_xor1(x23);
        if (b0) { r0 = y01; r2 = y23; } else { r1 = y01; r3 = y23; }
    }
    {   // stage 2: exchange across lane ^ 2 within register pairs (0,2) and (1,3)
        const float x02 = b1 ? r0 : r2, y02 = dpp_xor2(x02);
        const float x13 = b1 ? r1 : r3, y13 = dpp_xor2(x13);
        if (b1) { r0 = y02; r1 = y13; } else { r2 = y02; r3 = y13; }
    }
}

// ------------------------------------------------------------------------------------------
// packed[mb][chunk][part][tap][oc][m][8]: part 0 = hi, 1 = lo; oc < 2*CCG octets of the chunk;
// (mode | 0x100, CCG = 1: the f16 + fp8 format -- part 0 = the octets as f16, part 1 = [tap][kind][m][16 channels of the chunk] fp8 e4m3, kind 0 = f16(w),
//  kind 1 = (w - f16(w)) * 2^12; conv_fwd5_kernel<.., NT = 4>)
// m < MP rows of m-block mb.   mode 0: A[o][(tap,c)] = w[o][c][tap]  (w: [Co][Ctot][T])
//                               mode 1: A[i][(tap,k)] = w[k][i][T-1-tap]  (w: [Ctot][Co][T])
//                               mode 2 (dcn_fwd4_kernel; T = 1, CCG = 1, "chunk" = k-step j, oc = lane half h):
//                                       A[o][8 h + c'] = w[o][8 (u / 9) + c'][u % 9], u = 2 j + h  (w: [Co][Ctot][9])
// One pre-packed weight image: what pack_weights_kernel writes for (w, Co, Ctot, T, MP, CCG, nchunks, nmb, mode).
// rvsr_pack_weights_batched re-packs a whole table of them in ONE launch (once per optimizer step, realvsr_amd.functional).
struct PackDesc {
    const float* w;
    bf16x8* out;
    int Co, Ctot, T, MP, CCG, nchunks, nmb, mode;
};
__device__ __forceinline__ void pack_weights_body(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int Ctot, int T,
                                                  int MP, int CCG, int nchunks, int nmb, int mode, size_t first, size_t step) {
    const int noct = 2 * CCG;
    const size_t total = (size_t)nmb * nchunks * T * noct * MP;
    for (size_t idx = first; idx < total; idx += step) {
        const int m = (int)(idx % MP);
        size_t r = idx / MP;
        const int oc = (int)(r % noct);
        r /= noct;
        const int tap = (int)(r % T);
        r /= T;
        const int chunk = (int)(r % nchunks);
        const int mb = (int)(r / nchunks);
        const int o = mb * MP + m;
        if (mode & 0x100) {   // "f16 main term + fp8 cross terms" image (rvsr_conv2d_forward w_mode | 4): format flag in bit 8 of `mode`
            // hi image: the octet's 8 channels as f16; lo image slot (tap, kind = oc, m): the chunk's 16 channels as fp8 e4m3 --
            // kind 0: a1 = f16(w), kind 1: (w - a1) * 2^12 (the MFMA's block scale takes the 2^12 back)
            typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
            typedef int i32x4p __attribute__((ext_vector_type(4)));
            f16x8p h;
            float q[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = chunk * 16 + j;
                const float x = (o < Co && c < Ctot) ? w[((size_t)o * Ctot + c) * T + tap] : 0.f;
                const float a1 = (float)(_Float16)x;
                q[j] = oc ? (x - a1) * 4096.f : a1;
                if ((j >> 3) == oc) h[j & 7] = (_Float16)x;
            }
            i32x4p f8;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int wd = 0;
                wd = __builtin_amdgcn_cvt_pk_fp8_f32(q[4 * k], q[4 * k + 1], wd, false);
                wd = __builtin_amdgcn_cvt_pk_fp8_f32(q[4 * k + 2], q[4 * k + 3], wd, true);
                f8[k] = wd;
            }
            const size_t blk = ((size_t)mb * nchunks + chunk) * 2, inner = ((size_t)tap * noct + oc) * MP + m, per = (size_t)T * noct * MP;
            packed[blk * per + inner] = __builtin_bit_cast(bf16x8, h);
            packed[(blk + 1) * per + inner] = __builtin_bit_cast(bf16x8, f8);
            continue;
        }
        const int unit = chunk * 2 + oc;                                   // (mode 2)
        const int cb = mode == 2 ? 8 * (unit / 9) : (chunk * noct + oc) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb + j;
            float x = 0.f;
            if (o < Co && c < Ctot)
                x = mode == 0 ? w[((size_t)o * Ctot + c) * T + tap]
                  : mode == 1 ? w[((size_t)c * Co + o) * T + (T - 1 - tap)]
                              : w[((size_t)o * Ctot + c) * 9 + unit % 9];
            v[j] = x;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        const size_t blk = ((size_t)mb * nchunks + chunk) * 2;  // hi block, lo block follows
        const size_t inner = ((size_t)tap * noct + oc) * MP + m;
        const size_t per = (size_t)T * noct * MP;
        packed[blk * per + inner] = hi;
        packed[(blk + 1) * per + inner] = lo;
    }
}
#ifdef RVSR_DEFINE_PACK
__global__ void pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int Ctot, int T,
                                    int MP, int CCG, int nchunks, int nmb, int mode) {
    pack_weights_body(w, packed, Co, Ctot, T, MP, CCG, nchunks, nmb, mode, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                      (size_t)gridDim.x * blockDim.x);
}
// grid (blocks per image, images)
__global__ void pack_weights_batched_kernel(const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    pack_weights_body(d.w, d.out, d.Co, d.Ctot, d.T, d.MP, d.CCG, d.nchunks, d.nmb, d.mode,
                      (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
#else
__global__ void pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int Ctot, int T,
                                    int MP, int CCG, int nchunks, int nmb, int mode);
__global__ void pack_weights_batched_kernel(const PackDesc* __restrict__ descs);
#endif
