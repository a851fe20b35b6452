// Micro-benchmark: k-loop of a 256x256x(BK) fp16 MFMA tile on gfx950, two staging pipelines with identical MFMA work
//   V0  two LDS stages of BK = 64 (64 KB each), all DMA of the next k-tile issued behind the barrier, vmcnt(0) + barrier per k-tile
//       (the structure of gemm_glds_kernel)
//   V1  ring of NS half-stages of BK = 32 (32 KB each), counted vmcnt: up to NS-1 half-stages in flight ACROSS the barriers
// out[m][n] = sum_k A[m][k] W[n][k]; 8 waves (2 x 4), wave tile 128 x 64, operands by global_load_lds_dwordx4 with the XOR swizzle on
// the source address.  Prints microseconds per 64-deep k-tile per workgroup and the TFLOP/s of the whole launch, and checks V1 == V0.
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_kloop gemm_kloop.hip     Run: ./gemm_kloop [K] [row_tiles]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int BM = 256, BN = 256;

template <int VARIANT, int NS>
__global__ __launch_bounds__(512) void kloop(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ out, int M, int N, int K, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % 2, wn = wave / 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int n_tiles = N / BN;
    f32x4 acc[4][8];
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int n0 = (tile % n_tiles) * BN, m0 = (tile / n_tiles) * BM;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (VARIANT == 0) {
            // ---- two full stages, rows of 128 B, pieces of 8 rows, swizzle chunk ^= (row >> 1) & 7
            constexpr int XB = BM * 128, STAGE = XB + BN * 128;
            const int lrow = lane >> 3, pc = lane & 7;
            const f16* xs[4]; const f16* ws[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (wave * 4 + i) * 8 + lrow;
                const int c = pc ^ ((row >> 1) & 7);
                xs[i] = A + (long)(m0 + row) * K + c * 8;
                ws[i] = W + (long)(n0 + row) * K + c * 8;
            }
            auto stage = [&](int kt, int buf) {
                char* base = smem + buf * STAGE;
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[i] + kt * 64), (lds_ptr_t)(base + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws[i] + kt * 64), (lds_ptr_t)(base + XB + (wave * 4 + i) * 1024), 16, 0, 0);
            };
            const int nk = K / 64;
            const int fsw = (frow >> 1) & 7;
            stage(0, 0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
                const char* sX = smem + (kt & 1) * STAGE;
                const char* sW = sX + XB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int choff = ((kk * 4 + fq) ^ fsw) << 4;
                    f16x8 wf[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sW + (wn * 64 + i * 16 + frow) * 128 + choff);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f16x8 xf = *reinterpret_cast<const f16x8*>(sX + (wm * 128 + j * 16 + frow) * 128 + choff);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf, acc[i][j], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        } else {
            // ---- ring of NS half-stages, rows of 64 B, pieces of 16 rows, swizzle chunk ^= f[(row >> 2) & 3], f = {0,2,3,1}
            constexpr int XB = BM * 64, SLOT = XB + BN * 64;
            constexpr int DIST = NS - 1;
            const int lrow = lane >> 2, pc = lane & 3;
            auto fswz = [](int row) -> int { return (0x1e >> (((row >> 2) & 3) * 2)) & 3; };      // {0,2,3,1} packed: 0b01'11'10'00 = 0x78?  see below
            const f16* xs[2]; const f16* ws[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave * 2 + i) * 16 + lrow;
                const int g = (row >> 2) & 3;
                const int f = g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 3 : 1;
                const int c = pc ^ f;
                xs[i] = A + (long)(m0 + row) * K + c * 8;
                ws[i] = W + (long)(n0 + row) * K + c * 8;
            }
            (void)fswz;
            auto stage = [&](int hs) {
                char* base = smem + (hs % NS) * SLOT;
#pragma unroll
                for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[i] + hs * 32), (lds_ptr_t)(base + (wave * 2 + i) * 1024), 16, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws[i] + hs * 32), (lds_ptr_t)(base + XB + (wave * 2 + i) * 1024), 16, 0, 0);
            };
            const int nh = K / 32;
            const int g = (frow >> 2) & 3;
            const int ff = g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 3 : 1;
            const int choff = (fq ^ ff) << 4;
#pragma unroll
            for (int d = 0; d < DIST; ++d)
                if (d < nh) stage(d);
            for (int hs = 0; hs < nh; ++hs) {
                const int later = nh - 1 - hs;              // half-stages issued after hs that exist
                if (DIST >= 3 && later >= 2) wait_vmcnt<8>();
                else if (DIST >= 2 && later >= 1) wait_vmcnt<4>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();               // hs has landed for everyone; slot (hs-1) % NS is free
                if (hs + DIST < nh) stage(hs + DIST);
                const char* sX = smem + (hs % NS) * SLOT;
                const char* sW = sX + XB;
                f16x8 wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sW + (wn * 64 + i * 16 + frow) * 64 + choff);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f16x8 xf = *reinterpret_cast<const f16x8*>(sX + (wm * 128 + j * 16 + frow) * 64 + choff);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf, acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_barrier();                   // everyone is done reading before the next tile restages
        }
        // epilogue: plain fragment-order fp16 stores (both variants identical; excluded from the per-k-tile figure by using a long K)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + wn * 64 + i * 16 + fq * 4, m = m0 + wm * 128 + j * 16 + frow;
                f16* o = out + (long)m * N + n;
                o[0] = (f16)acc[i][j].x; o[1] = (f16)acc[i][j].y; o[2] = (f16)acc[i][j].z; o[3] = (f16)acc[i][j].w;
            }
    }
}

template <int V, int NS>
static float run(const f16* A, const f16* W, f16* out, int M, int N, int K, int iters) {
    const size_t lds = V == 0 ? 2 * (size_t)(BM * 128 + BN * 128) : NS * (size_t)(BM * 64 + BN * 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kloop<V, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (M / BM) * (N / BN);
    const int grid = tiles < 256 ? tiles : 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kloop<V, NS>), dim3(grid), dim3(512), lds, 0, A, W, out, M, N, K, tiles);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((kloop<V, NS>), dim3(grid), dim3(512), lds, 0, A, W, out, M, N, K, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4096;
    const int rt = argc > 2 ? atoi(argv[2]) : 128;               // row tiles: M = 256 * rt
    const int M = 256 * rt, N = 1536;
    std::vector<f16> hA((size_t)M * K), hW((size_t)N * K);
    unsigned x = 12345;
    for (auto& v : hA) { x = x * 1664525u + 1013904223u; v = (f16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f); }
    for (auto& v : hW) { x = x * 1664525u + 1013904223u; v = (f16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f); }
    f16 *A, *W, *o0, *o1;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&o0, (size_t)M * N * 2); hipMalloc(&o1, (size_t)M * N * 2);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    const double flop = 2.0 * M * N * K;
    const int tiles = (M / BM) * (N / BN), rounds = (tiles + 255) / 256;
    for (int rep = 0; rep < 2; ++rep) {
        const float t0 = run<0, 2>(A, W, o0, M, N, K, 5);
        const float t3 = run<1, 3>(A, W, o1, M, N, K, 5);
        const float t4 = run<1, 4>(A, W, o1, M, N, K, 5);
        printf("K=%d M=%d: 2-stage %.3f ms (%.0f TF/s, %.2f us/k-tile)  ring3 %.3f ms (%.0f TF/s, %.2f us)  ring4 %.3f ms (%.0f TF/s, %.2f us)\n", K, M,
               t0, flop / t0 / 1e9, t0 * 1e3 / rounds / (K / 64), t3, flop / t3 / 1e9, t3 * 1e3 / rounds / (K / 64), t4, flop / t4 / 1e9, t4 * 1e3 / rounds / (K / 64));
    }
    std::vector<f16> r0((size_t)M * N), r1((size_t)M * N);
    hipMemcpy(r0.data(), o0, r0.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), o1, r1.size() * 2, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    printf("ring vs 2-stage: %zu differing outputs of %zu\n", bad, r0.size());
    return 0;
}
