// bf16x3.h -- helpers for the 3-term bf16 split GEMM (see conv2_kernels.hip header comment).
#pragma once
#include "rvsr_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
}

// ------------------------------------------------------------------------------------------
// packed[mb][chunk][part][tap][oc][m][8]: part 0 = hi, 1 = lo; oc < 2*CCG octets of the chunk;
// m < MP rows of m-block mb.   mode 0: A[o][(tap,c)] = w[o][c][tap]  (w: [Co][Ctot][T])
//                               mode 1: A[i][(tap,k)] = w[k][i][T-1-tap]  (w: [Ctot][Co][T])
#ifdef RVSR_DEFINE_PACK
__global__ void pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int Ctot, int T,
                                    int MP, int CCG, int nchunks, int nmb, int mode) {
    const int noct = 2 * CCG;
    const size_t total = (size_t)nmb * nchunks * T * noct * MP;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx % MP);
        size_t r = idx / MP;
        const int oc = (int)(r % noct);
        r /= noct;
        const int tap = (int)(r % T);
        r /= T;
        const int chunk = (int)(r % nchunks);
        const int mb = (int)(r / nchunks);
        const int o = mb * MP + m, cb = (chunk * noct + oc) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb + j;
            float x = 0.f;
            if (o < Co && c < Ctot)
                x = mode == 0 ? w[((size_t)o * Ctot + c) * T + tap] : w[((size_t)c * Co + o) * T + (T - 1 - tap)];
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
#else
__global__ void pack_weights_kernel(const float* __restrict__ w, bf16x8* __restrict__ packed, int Co, int Ctot, int T,
                                    int MP, int CCG, int nchunks, int nmb, int mode);
#endif
