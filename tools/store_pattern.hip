// Microbenchmark: how fast one workgroup per CU (8 waves) can write a 256x256 output tile to HBM, as a function of
// the per-instruction access pattern.  Explains the GEMM epilogue: the MFMA C/D fragment layout gives every store
// instruction 16 rows x 32 B (fp16) or 16 rows x 64 B (fp32) pieces; transposing through LDS first gives
// 8 rows x 128 B / 4 rows x 256 B.
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// every workgroup writes `tiles` tiles of 256 rows x 256 cols of a (M, ldc) matrix; wave tile 128x64
template <int MODE>
__global__ __launch_bounds__(512) void k(char* out, long ldc, int tiles, int n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave % 2, wn = wave / 2;
    const int frow = lane & 15, fq = lane >> 4;
    for (int t = 0; t < tiles; ++t) {
        const int tile = t * gridDim.x + blockIdx.x;
        const long m0 = (long)(tile / n_tiles) * 256 + wm * 128;
        const long n0 = (long)(tile % n_tiles) * 256 + wn * 64;
        const float base = (float)tile;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {          // fp16 fragment layout: lane -> row frow, 8 B at col i*16 + fq*4
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f16x4 v = {(f16)base, (f16)j, (f16)i, (f16)lane};
                    *reinterpret_cast<f16x4*>(out + ((m0 + j * 16 + frow) * ldc + n0 + i * 16 + fq * 4) * 2) = v;
                }
            } else if (MODE == 1) {   // fp16 row layout: lane -> row (lane>>3) + 8h, 16 B at col (lane&7)*8
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f16x8 v = {(f16)base, (f16)j, (f16)h, (f16)lane, (f16)1, (f16)2, (f16)3, (f16)4};
                    *reinterpret_cast<f16x8*>(out + ((m0 + j * 16 + h * 8 + (lane >> 3)) * ldc + n0 + (lane & 7) * 8) * 2) = v;
                }
            } else if (MODE == 2) {   // fp32 fragment layout: 16 B at row frow, col i*16 + fq*4
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = {base, (float)j, (float)i, (float)lane};
                    *reinterpret_cast<f32x4*>(out + ((m0 + j * 16 + frow) * ldc + n0 + i * 16 + fq * 4) * 4) = v;
                }
            } else if (MODE == 3) {   // fp32 row layout: row (lane>>4) + 4h, 16 B at col (lane&15)*4
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    f32x4 v = {base, (float)j, (float)h, (float)lane};
                    *reinterpret_cast<f32x4*>(out + ((m0 + j * 16 + h * 4 + (lane >> 4)) * ldc + n0 + (lane & 15) * 4) * 4) = v;
                }
            } else if (MODE == 4) {   // fp16 row layout, 8 B per lane: row (lane>>4) + 4h, col (lane&15)*4
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    f16x4 v = {(f16)base, (f16)j, (f16)h, (f16)lane};
                    *reinterpret_cast<f16x4*>(out + ((m0 + j * 16 + h * 4 + (lane >> 4)) * ldc + n0 + (lane & 15) * 4) * 2) = v;
                }
            }
        }
    }
}

template <int MODE> void run(const char* name, int esz, int grid) {
    const long M = 100800 / 256 * 256, N = 1536;
    char* out; hipMalloc(&out, (size_t)M * N * esz);
    const int n_tiles = N / 256, total = (int)(M / 256) * n_tiles;
    const int tiles = total / grid;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, N, tiles, n_tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, N, tiles, n_tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)tiles * grid * 256 * 256 * esz;
    printf("%-44s grid %3d: %7.1f us  %6.2f us/tile/CU  %7.1f GB/s\n", name, grid, ms * 1e3, ms * 1e3 / tiles, bytes / ms / 1e6);
    hipFree(out);
}

int main() {
    for (int grid : {256, 64}) {
        run<0>("fp16 fragment layout (16 rows x 32 B / instr)", 2, grid);
        run<1>("fp16 row layout b128 (8 rows x 128 B / instr)", 2, grid);
        run<4>("fp16 row layout b64  (4 rows x 128 B / instr)", 2, grid);
        run<2>("fp32 fragment layout (16 rows x 64 B / instr)", 4, grid);
        run<3>("fp32 row layout b128 (4 rows x 256 B / instr)", 4, grid);
    }
    return 0;
}
