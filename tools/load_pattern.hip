// Microbenchmark: per-CU cost of READING a 256x256 fp16 tile with the MFMA C/D fragment access pattern (8 B per lane,
// 16 rows x 32 B per instruction) versus whole rows (16 B per lane, 8 rows x 128 B) -- the load-side twin of
// tools/store_pattern.hip.  Decides whether a residual kept as fp16 (+ 8-bit correction) could be read row-major.
//   hipcc --offload-arch=gfx950 -O3 tools/load_pattern.hip -o tools/bin/load_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* in, long ldc, int tiles, int n_tiles, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave % 2, wn = wave / 2;
    const int frow = lane & 15, fq = lane >> 4;
    float acc = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const int tile = t * gridDim.x + blockIdx.x;
        const long m0 = (long)(tile / n_tiles) * 256 + wm * 128;
        const long n0 = (long)(tile % n_tiles) * 256 + wn * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {          // fp16 fragment layout
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f16x4 v = *reinterpret_cast<const f16x4*>(in + ((m0 + j * 16 + frow) * ldc + n0 + i * 16 + fq * 4) * 2);
                    acc += (float)v[0] + (float)v[3];
                }
            } else if (MODE == 1) {   // fp16 rows, 16 B per lane
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(in + ((m0 + j * 16 + h * 8 + (lane >> 3)) * ldc + n0 + (lane & 7) * 8) * 2);
                    acc += (float)v[0] + (float)v[7];
                }
            } else if (MODE == 2) {   // fp32 fragment layout
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((m0 + j * 16 + frow) * ldc + n0 + i * 16 + fq * 4) * 4);
                    acc += v[0] + v[3];
                }
            } else {                  // fp32 rows
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((m0 + j * 16 + h * 4 + (lane >> 4)) * ldc + n0 + (lane & 15) * 4) * 4);
                    acc += v[0] + v[3];
                }
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE> void run(const char* name, int esz, int grid) {
    const long M = 100800 / 256 * 256, N = 1536;
    char* in; hipMalloc(&in, (size_t)M * N * esz); hipMemset(in, 0, (size_t)M * N * esz);
    float* sink; hipMalloc(&sink, 4);
    const int n_tiles = N / 256, total = (int)(M / 256) * n_tiles;
    const int tiles = total / grid;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, in, N, tiles, n_tiles, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, in, N, tiles, n_tiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)tiles * grid * 256 * 256 * esz;
    printf("%-44s grid %3d: %7.1f us  %6.2f us/tile/CU  %7.1f GB/s\n", name, grid, ms * 1e3, ms * 1e3 / tiles, bytes / ms / 1e6);
    hipFree(in); hipFree(sink);
}

int main() {
    for (int grid : {256, 64}) {
        run<0>("fp16 fragment layout (16 rows x 32 B / instr)", 2, grid);
        run<1>("fp16 rows b128       (8 rows x 128 B / instr)", 2, grid);
        run<2>("fp32 fragment layout (16 rows x 64 B / instr)", 4, grid);
        run<3>("fp32 rows b128       (4 rows x 256 B / instr)", 4, grid);
    }
    return 0;
}
