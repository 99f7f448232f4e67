// store_micro.hip -- global store throughput per CU (one 512-thread workgroup per CU writes TILES x 128 KB, no compute).
//   pattern 0: dword stores, a wave covers 256 contiguous bytes          pattern 1: dwordx4 stores, a wave covers 1 KB contiguous
//   pattern 2: dwordx4 stores in the conv epilogue pattern: lane quad j -> channel plane (stride PLANE), 8 lane quads -> 128 contiguous bytes
//   pattern 3: as 2 through raw buffer stores
// Each run: all 256 workgroups, then 32 workgroups (one per 8: is the limit per CU or shared?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define NT 512
template <int PAT, int NTS = 0>
__global__ __launch_bounds__(NT) void k(float* out, size_t plane, int tiles, int stride_wg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t wg = (size_t)blockIdx.x * stride_wg;
    for (int t = 0; t < tiles; ++t) {
        // a "tile": 64 channel planes x 16 rows x 32 px; wave w owns rows 2w, 2w+1
        const size_t tbase = (wg * tiles + t) * 512;   // 512 floats = 16 rows x 32 px per plane-tile (planes are `plane` floats apart)
        if (PAT == 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {   // 64 dword stores: (channel i, row 2w + lane/32, px lane%32)
                out[(size_t)i * plane + tbase + (2 * wave + (lane >> 5)) * 32 + (lane & 31)] = (float)i;
            }
        } else if (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {   // 16 dwordx4 stores, each wave-instruction 1 KB contiguous = two channel planes' rows
                float4* p = reinterpret_cast<float4*>(out + (size_t)(4 * i + (lane >> 4)) * plane + tbase + (2 * wave) * 32) + (lane & 15);
                if (NTS) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f4v{1.f, 2.f, 3.f, (float)i}, reinterpret_cast<f4v*>(p));
                } else
                    *p = make_float4(1.f, 2.f, 3.f, (float)i);
            }
        } else if (PAT == 7 || PAT == 8) {
            // 16 instructions, each 4 planes x 256 B (PAT 7) or 2 planes x 512 B (PAT 8) of one 1280-byte-pitch row
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int per = PAT == 7 ? 16 : 32;                 // lanes per plane
                const int ch = i * (64 / per) + lane / per;
                float4* p = reinterpret_cast<float4*>(out + (size_t)ch * plane + tbase * 8 + (size_t)wave * 320) + (lane % per);
                *p = make_float4(1.f, 2.f, 3.f, (float)i);
            }
        } else if (PAT == 9) {
            // 4 planes x 256 B per instruction with the lane order of the wide conv epilogue: lane & 3 -> plane, lane >> 2 -> 16-byte group
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ch = i * 4 + (lane & 3);
                float4* p = reinterpret_cast<float4*>(out + (size_t)ch * plane + tbase * 8 + (size_t)wave * 320) + (lane >> 2);
                *p = make_float4(1.f, 2.f, 3.f, (float)i);
            }
        } else if (PAT == 5 || PAT == 6) {
            // tile of 8 rows x 64 px (wave = one row): instruction (m, rg, n) covers 8 planes x 128 B, n = left / right half of the row
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int lo = lane & 31, hi = lane >> 5, j = lo & 3;
                        const int ch = m * 32 + 8 * rg + 4 * hi + j;
                        // PAT 5: row of 64 px = 256 contiguous bytes per plane (tile rows 256 B apart in this synthetic plane); PAT 6: image-like row pitch 1280 B
                        const size_t rowoff = PAT == 5 ? (size_t)wave * 64 : (size_t)wave * 320;
                        float4* p = reinterpret_cast<float4*>(out + (size_t)ch * plane + tbase * (PAT == 5 ? 1 : 8) + rowoff + 32 * n + (lo & ~3));
                        *p = make_float4(1.f, 2.f, 3.f, (float)rg);
                    }
        } else {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int lo = lane & 31, hi = lane >> 5, j = lo & 3;
                        const int ch = m * 32 + 8 * rg + 4 * hi + j;
                        float4* p;
                        if (PAT == 2) p = reinterpret_cast<float4*>(out + (size_t)ch * plane + tbase + (2 * wave + n) * 32 + (lo & ~3));
                        else if (PAT == 3)   // the 8 segments of an instruction are 8 ROWS (1280 B apart) of one plane instead of 8 planes
                            p = reinterpret_cast<float4*>(out + (size_t)(m * 4 + rg) * plane + tbase * 8 + (size_t)((2 * wave + n) * 8 + 4 * hi + j) * 320 + (lo & ~3));
                        else                  // PAT 4: 8 planes, but each lane QUAD covers 64 contiguous bytes (4 lanes x 16 B): lanes lo&3 -> pixels, lo>>2 .. -> channel
                            p = reinterpret_cast<float4*>(out + (size_t)(m * 32 + 8 * rg + (lo >> 3) + 4 * hi) * plane + tbase + (2 * wave + n) * 32 + 4 * (lo & 7));
                        if (NTS) {
                            typedef float f4v __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store(f4v{1.f, 2.f, 3.f, (float)rg}, reinterpret_cast<f4v*>(p));
                        } else
                            *p = make_float4(1.f, 2.f, 3.f, (float)rg);
                    }
        }
    }
}
template <int PAT, int NTS = 0> void run(const char* name, float* buf, size_t plane, int nwg, int stride) {
    const int tiles = 64;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<PAT, NTS><<<nwg, NT>>>(buf, plane, tiles, stride);
    (void)hipEventRecord(a);
    k<PAT, NTS><<<nwg, NT>>>(buf, plane, tiles, stride);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)nwg * tiles * 131072.0;
    printf("%-46s %3d workgroups: %7.3f ms  %7.1f GB/s total  %6.1f B/clk per CU at 2.1 GHz\n", name, nwg, ms, bytes / ms / 1e6, bytes / nwg / (ms * 1e-3 * 2.1e9));
}
int main() {
    const size_t plane = (size_t)256 * 64 * 512 * 8 + 4096 + (getenv("SKEW") ? atoi(getenv("SKEW")) : 0);   // floats per channel plane (all tiles of all workgroups)
    float* buf; if (hipMalloc(&buf, plane * 64 * sizeof(float) + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int nwg : {64}) {
        const int stride = 256 / nwg;
        if (nwg == 256 || nwg == 32) run<0>("dword stores (256 B contiguous per wave)", buf, plane, nwg, stride);
        run<1>("dwordx4 stores (1 KB contiguous per wave)", buf, plane, nwg, stride);
        run<2>("dwordx4 stores, conv epilogue pattern", buf, plane, nwg, stride);
        run<7>("4 planes x 256 B per instruction", buf, plane, nwg, stride);
        run<8>("2 planes x 512 B per instruction", buf, plane, nwg, stride);
        run<9>("4 planes x 256 B, lane & 3 -> plane (wide epilogue)", buf, plane, nwg, stride);
        if (nwg == 256 || nwg == 32) run<3>("dwordx4, 8 rows x 128 B of ONE plane per instr", buf, plane, nwg, stride);
        if (nwg == 256 || nwg == 32) run<4>("dwordx4, 8 planes x 128 B, lanes 0-7 contiguous", buf, plane, nwg, stride);
    }
    return 0;
}
