// Microbenchmark: issue rate of the f16 / bf16 MFMA shapes used by libjegal_hip (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a, b; bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); ab[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0); }
        if (MODE == 1) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0); }
        if (MODE == 2) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0); d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0); }
        if (MODE == 3) { d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d1, 0, 0, 0); d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d3, 0, 0, 0); }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, double flop_per_mfma) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000, blocks = 256;   // one 4-wave block per CU -> one wave per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 4;
    printf("%-22s %8.3f ms  %8.1f TFLOP/s   %.1f ns per MFMA per SIMD\n", name, ms, mfmas * flop_per_mfma / ms / 1e9, ms * 1e6 / (iters * 4.0));
    hipFree(out);
}
template <int NACC>
__global__ __launch_bounds__(256) void kchain(float* out, int iters, const _Float16* src) {
    f16x8 a[8], b;
    for (int j = 0; j < 8; ++j) a[j] = *(const f16x8*)(src + (threadIdx.x * 8 + j) * 8);
    b = *(const f16x8*)(src + 4096 + threadIdx.x * 8);
    f32x16 c0 = {0}, c1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (NACC == 1 || (j & 1) == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b, c0, 0, 0, 0);
            else c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b, c1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void kchain16(float* out, int iters, const _Float16* src) {
    f16x8 a[8], b;
    for (int j = 0; j < 8; ++j) a[j] = *(const f16x8*)(src + (threadIdx.x * 8 + j) * 8);
    b = *(const f16x8*)(src + 4096 + threadIdx.x * 8);
    f32x4 c[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b, c[j % NACC], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c[0][i] + c[1][i] + c[2][i] + c[3][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the GEMM kernel's patterns: 16 accumulators acc[i][j], A fragment per i (4), B fragment per j (4), two k-steps (kk)
//   ORDER 0: kk outer, j, i inner (gemm.hip round 1-3): every MFMA switches the accumulator, each accumulator gets one MFMA per kk
//   ORDER 1: j, i, kk inner: two consecutive MFMAs per accumulator
//   ORDER 2: one accumulator per (i, j) fed by 4 consecutive MFMAs (a K = 128 k-tile)
template <int ORDER>
__global__ __launch_bounds__(512) void kgemm16(float* out, int iters, const _Float16* src) {
    f16x8 a[4][4], b[4][4];      // [kk][i], [kk][j]
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j) {
            a[k][j] = *(const f16x8*)(src + ((threadIdx.x * 16 + k * 4 + j) * 8 & 32767));
            b[k][j] = *(const f16x8*)(src + 32768 + ((threadIdx.x * 16 + k * 4 + j) * 8 & 16383));
        }
    f32x4 c[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk][i], b[kk][j], c[i][j], 0, 0, 0);
        } else if (ORDER == 1) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k2 * 2 + kk][i], b[k2 * 2 + kk][j], c[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk][i], b[kk][j], c[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ORDER> void rungemm16(const char* name, int wg) {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    _Float16* src; hipMalloc(&src, 65536 * 2);
    static _Float16 hsrc[65536]; for (int i = 0; i < 65536; ++i) hsrc[i] = (_Float16)(((i * 2654435761u) >> 16 & 1023) / 512.0f - 1.0f);
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    const int iters = 40000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kgemm16<ORDER>, dim3(blocks), dim3(wg), 0, 0, out, 1000, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kgemm16<ORDER>, dim3(blocks), dim3(wg), 0, 0, out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 64.0) / (wg / 256);
    printf("%-44s %8.3f ms  %.1f ns per MFMA per SIMD -> %.0f TFLOP/s\n", name, ms, per, 1024.0 * 16 * 16 * 32 * 2 / per / 1e3);
}
template <int NACC> void runchain16(const char* name) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    _Float16* src; hipMalloc(&src, 65536 * 2);
    static _Float16 hsrc[65536]; for (int i = 0; i < 65536; ++i) hsrc[i] = (_Float16)(((i * 2654435761u) >> 16 & 1023) / 512.0f - 1.0f);
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    const int iters = 10000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kchain16<NACC>, dim3(blocks), dim3(256), 0, 0, out, 1000, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kchain16<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %.1f ns per MFMA per SIMD  (random data) -> %.0f TFLOP/s\n", name, ms, ms * 1e6 / (iters * 8.0), 1024.0 * 16 * 16 * 32 * 2 / (ms * 1e6 / (iters * 8.0)) / 1e3);
}
template <int NACC> void runchain(const char* name) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    _Float16* src; hipMalloc(&src, 65536 * 2);
    _Float16 hsrc[65536]; for (int i = 0; i < 65536; ++i) hsrc[i] = (_Float16)(((i * 2654435761u) >> 16 & 1023) / 512.0f - 1.0f);
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    const int iters = 10000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kchain<NACC>, dim3(blocks), dim3(256), 0, 0, out, 1000, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kchain<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %.1f ns per MFMA per SIMD  (random data)\n", name, ms, ms * 1e6 / (iters * 8.0));
}
int main() {
    rungemm16<0>("gemm pattern, acc switch every MFMA, 1 wave/SIMD", 256);
    rungemm16<1>("gemm pattern, 2 MFMAs per acc, 1 wave/SIMD", 256);
    rungemm16<2>("gemm pattern, 4 MFMAs per acc, 1 wave/SIMD", 256);
    rungemm16<0>("gemm pattern, acc switch every MFMA, 2 waves/SIMD", 512);
    rungemm16<1>("gemm pattern, 2 MFMAs per acc, 2 waves/SIMD", 512);
    rungemm16<2>("gemm pattern, 4 MFMAs per acc, 2 waves/SIMD", 512);
    runchain16<1>("16x16x32 f16 1 acc chain");
    runchain16<2>("16x16x32 f16 2 acc chains");
    runchain16<4>("16x16x32 f16 4 acc chains");
    runchain<1>("32x32x16 f16 1 acc chain");
    runchain<2>("32x32x16 f16 2 acc chains");
    run<0>("32x32x16 f16", 32.0 * 32 * 16 * 2);
    run<1>("32x32x16 bf16", 32.0 * 32 * 16 * 2);
    run<2>("16x16x32 f16", 16.0 * 16 * 32 * 2);
    run<3>("16x16x32 bf16", 16.0 * 16 * 32 * 2);
    return 0;
}
