// Microbenchmark: what does the L2 -> LDS path (global_load_lds_dwordx4, "LDS-DMA") deliver per CU when EVERY CU streams at once,
// with nothing else going on?  Every GEMM-shaped kernel of the path sits at ~45 GB/s per CU of operand intake beside its MFMAs
// (DESIGN.md section 5); this asks whether that is the path's ceiling or the kernels' doing.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_rate.hip -o tools/bin/lds_dma_rate
// One workgroup of 8 waves per CU (as the GEMM kernels), 128 KB of LDS in two 64 KB stages.  One "k-tile" = 64 pieces of 1 KB per
// workgroup (8 per wave), a piece = 8 rows x 128 B like the GEMM's operand pieces (row pitch `pitch` bytes).  Source footprints:
//   shared  : all workgroups read the same 1.5 MB (weights-like: L2-resident in every XCD)
//   xcd     : the 32 workgroups of an XCD share a 2 MB region (activation-panel-like)
//   private : every workgroup streams its own 16 MB (HBM / Infinity Cache)
// Sync modes: "barrier" = s_waitcnt vmcnt(0) + workgroup barrier per k-tile (the GEMM k loop's structure, two stages),
//             "free"    = vmcnt(8): one k-tile of pieces stays in flight per wave, no barrier.
// MFMA variants: the same loop with 64 v_mfma_f32_16x16x32_f16 per wave and k-tile on register operands (the 256x256 tile's count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int SYNC, int MFMA>
__global__ __launch_bounds__(512) void k(const char* src, long region_bytes, int mode, int ktiles, long pitch, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lrow = lane >> 3, pc = lane & 7;
    const long base = mode == 0 ? 0 : (mode == 1 ? (long)(blockIdx.x & 7) * region_bytes : (long)blockIdx.x * region_bytes);
    const char* rb = src + base;
    f32x4 acc[16];
    f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a k-tile of the workgroup covers 64 pieces x 8 rows = 512 rows of 128 B at row pitch `pitch`; consecutive k-tiles advance by 128 B
    // along the row (like a GEMM's K walk) and wrap inside the region
    const long rows_in_region = region_bytes / pitch;
    for (int kt = 0; kt < ktiles; ++kt) {
        char* stage = smem + (kt & 1) * 65536;
        const long col = ((long)kt * 128) % pitch;
        const long rblock = (((long)kt * 128) / pitch) * 512 % (rows_in_region > 512 ? rows_in_region - 512 : 1);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long row = rblock + (wave * 8 + p) * 8 + lrow;
            const char* s = rb + row * pitch + col + pc * 16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)s, (lds_ptr_t)(stage + (wave * 8 + p) * 1024), 16, 0, 0);
            if (MFMA) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[(p * 8 + m) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(p * 8 + m) & 15], 0, 0, 0);
            }
        }
        if (SYNC == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc[i].x;
    if (t == 123.456f || smem[threadIdx.x] == 77) sink[0] = t;
}

template <int SYNC, int MFMA> void run(const char* name, const char* src, int mode, long region, long pitch, float* sink) {
    const int ktiles = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SYNC, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SYNC, MFMA>), dim3(256), dim3(512), 131072, 0, src, region, mode, 200, pitch, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SYNC, MFMA>), dim3(256), dim3(512), 131072, 0, src, region, mode, ktiles, pitch, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * ktiles * 65536.0;
    const double mfma_tf = MFMA ? 256.0 * 8 * ktiles * 64 * (16.0 * 16 * 32 * 2) / (ms * 1e-3) / 1e12 : 0.0;
    printf("%-58s %8.1f us  %6.3f us/k-tile  %6.1f GB/s per CU  %6.2f TB/s total", name, ms * 1e3, ms * 1e3 / ktiles, bytes / 256 / ms / 1e6, bytes / ms / 1e9);
    if (MFMA) printf("  (%.0f TFLOP/s of MFMA beside it)", mfma_tf);
    printf("\n");
}

int main() {
    const size_t total = (size_t)256 * 16 * 1024 * 1024;
    char* src; hipMalloc(&src, total); hipMemset(src, 1, total);
    float* sink; hipMalloc(&sink, 4);
    const long P = 1024;         // row pitch: K = 512 fp16
    run<0, 0>("shared 1.5 MB, barrier per k-tile", src, 0, 1536 * 1024, P, sink);
    run<1, 0>("shared 1.5 MB, free running", src, 0, 1536 * 1024, P, sink);
    run<0, 0>("per-XCD 2 MB, barrier per k-tile", src, 1, 2048 * 1024, P, sink);
    run<1, 0>("per-XCD 2 MB, free running", src, 1, 2048 * 1024, P, sink);
    run<0, 0>("private 16 MB per workgroup, barrier per k-tile", src, 2, 16 * 1024 * 1024, P, sink);
    run<1, 0>("private 16 MB per workgroup, free running", src, 2, 16 * 1024 * 1024, P, sink);
    run<0, 1>("shared 1.5 MB, barrier, + 64 MFMA per wave and k-tile", src, 0, 1536 * 1024, P, sink);
    run<1, 1>("shared 1.5 MB, free running, + 64 MFMA per wave and k-tile", src, 0, 1536 * 1024, P, sink);
    run<0, 1>("per-XCD 2 MB, barrier, + 64 MFMA per wave and k-tile", src, 1, 2048 * 1024, P, sink);
    run<1, 1>("per-XCD 2 MB, free running, + 64 MFMA per wave and k-tile", src, 1, 2048 * 1024, P, sink);
    run<1, 1>("private 16 MB, free running, + 64 MFMA per wave and k-tile", src, 2, 16 * 1024 * 1024, P, sink);
    // wide rows (K = 2048 fp16): fewer distinct rows per byte
    run<1, 0>("per-XCD 2 MB, free running, pitch 4096", src, 1, 2048 * 1024, 4096, sink);
    return 0;
}
