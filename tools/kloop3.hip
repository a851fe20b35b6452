// Micro-benchmark (round 5): does a second workgroup per CU hide the plain GEMM's epilogue?
//
// The production 256x256 tile (8 waves, one 128-KB workgroup per CU) spends 8 x 1.6 us in the k loop and ~2.7 us in its epilogue per
// tile at K = 512 (profiles/r5_gemm_timeline.txt): the matrix pipes idle while the accumulators leave.  Two INDEPENDENT workgroups per
// CU (4 waves, 256x128 tile each, a ring of three BK = 32 half-stages = 72 KB of LDS) would overlap one's epilogue with the other's
// k loop -- at 1.5x the L2 -> LDS bytes per FLOP.  Both forms here run the REAL GEMM (persistent over the real tiles, XCD-aware order,
// fp16 output through the row-transposing LDS epilogue with nontemporal stores) and are checked against each other and the host.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value tools/kloop3.hip -o tools/bin/kloop3
//   tools/bin/kloop3 [K = 512] [N = 1536] [M = 100864]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr int TP16 = 144;

// tile id of this workgroup in `round` (workgroups b, b + 8, ... share an XCD and take consecutive tile ids)
__device__ __forceinline__ int tile_of(int round, int G, int total) {
    const int v0 = round * G;
    if (v0 >= total) return -1;
    const int cnt = total - v0 < G ? total - v0 : G;
    const int b = blockIdx.x;
    if (b >= cnt) return -1;
    const int q = cnt / 8, rr = cnt % 8, xcd = b % 8, loc = b / 8;
    return v0 + (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
}

// the wave's 128 x 64 accumulator block -> fp16 rows of out (row pitch N), 16 rows at a time through `tsc` (16 x TP16 bytes of LDS)
__device__ __forceinline__ void store_block(const f32x4 (&acc)[4][8], f16* out, long row0, int col0, int N, char* tsc, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = acc[i][j];
            const f16x4 hv = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
            *reinterpret_cast<f16x4*>(tsc + frow * TP16 + i * 32 + fq * 8) = hv;
        }
        wave_sync();
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const f16x8 o = *reinterpret_cast<const f16x8*>(tsc + (h2 * 8 + (lane >> 3)) * TP16 + (lane & 7) * 16);
            __builtin_nontemporal_store(o, reinterpret_cast<f16x8*>(out + (row0 + j * 16 + h2 * 8 + (lane >> 3)) * N + col0 + (lane & 7) * 8));
        }
        wave_sync();
    }
}

// ---- BASE: the production structure.  8 waves, 256 x 256 tile, two BK = 64 stages, DMA burst behind the barrier.
__global__ __launch_bounds__(512) void base_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ out, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, XB = BM * 128, STAGE = XB + BN * 128;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % 2, wn = wave / 2;
    const int frow = lane & 15, fq = lane >> 4, fsw = (frow >> 1) & 7;
    const int lrow = lane >> 3, pc = lane & 7;
    const int n_tiles = N / BN, total = (M / BM) * n_tiles, G = gridDim.x, nk = K / 64;
    char* tsc = smem + 2 * STAGE + wave * (16 * TP16);
    const f16* xs[4]; const f16* ws[4];
    int m0 = 0, n0 = 0;
    auto setup = [&](int tile) {
        n0 = (tile % n_tiles) * BN; m0 = (tile / n_tiles) * BM;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + lrow;
            const int c = pc ^ ((row >> 1) & 7);
            xs[i] = A + (long)(m0 + row) * K + c * 8;
            ws[i] = W + (long)(n0 + row) * K + c * 8;
        }
    };
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[p] + kt * 64), (lds_ptr_t)(base + (wave * 4 + p) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws[p] + kt * 64), (lds_ptr_t)(base + XB + (wave * 4 + p) * 1024), 16, 0, 0);
    };
    int round = 0, gk = 0;
    int tile = tile_of(0, G, total);
    if (tile < 0) return;
    setup(tile);
    stage(0, 0);
    while (true) {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int cm0 = m0, cn0 = n0;
        int next = -1;
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) stage(kt + 1, (gk + 1) & 1);
            else {                                   // the next tile's first k-tile goes out in front of this tile's epilogue, as in production
                next = tile_of(++round, G, total);
                if (next >= 0) { setup(next); stage(0, (gk + 1) & 1); }
            }
            const char* sX = smem + (gk & 1) * STAGE;
            const char* sW = sX + XB;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int choff = ((kk * 4 + fq) ^ fsw) << 4;
                auto ldx = [&](int j) -> f16x8 { return *reinterpret_cast<const f16x8*>(sX + (wm * 128 + j * 16 + frow) * 128 + choff); };
                f16x8 wf[4], xq[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sW + (wn * 64 + i * 16 + frow) * 128 + choff);
                xq[0] = ldx(0);
                xq[1] = ldx(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xq[j % 3], acc[i][j], 0, 0, 0);
                    if (j + 2 < 8) xq[(j + 2) % 3] = ldx(j + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        store_block(acc, out, (long)cm0 + wm * 128, cn0 + wn * 64, N, tsc, lane);
        if (next < 0) break;
    }
}

// ---- DUO: two workgroups per CU.  4 waves, 256 x 128 tile, ring of three BK = 32 half-stages (two half-steps in flight).
__global__ __launch_bounds__(256, 2) void duo_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ out, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 128, XB = BM * 64, SLOT = XB + BN * 64, NS = 3;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % 2, wn = wave / 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 2, pc = lane & 3;
    const int n_tiles = N / BN, total = (M / BM) * n_tiles, G = gridDim.x, nh = K / 32;
    auto swz = [](int row) -> int { const int g = (row >> 2) & 3; return g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 3 : 1; };
    // issue side: runs two half-steps ahead of the compute side, across tile boundaries
    const f16* xs[4]; const f16* ws[2];
    int iround = 0, ih = 0;
    long ghs = 0;
    bool idone = false;
    auto isetup = [&](int tile) {
        const int n0 = (tile % n_tiles) * BN, m0 = (tile / n_tiles) * BM;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 16 + lrow;
            xs[i] = A + (long)(m0 + row) * K + (pc ^ swz(row)) * 8;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 16 + lrow;
            ws[i] = W + (long)(n0 + row) * K + (pc ^ swz(row)) * 8;
        }
    };
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (idone) return;
        char* base = smem + (ghs % NS) * SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[i] + ih * 32), (lds_ptr_t)(base + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws[i] + ih * 32), (lds_ptr_t)(base + XB + (wave * 2 + i) * 1024), 16, 0, 0);
        ++ghs;
        if (++ih == nh) {
            ih = 0;
            const int nt = tile_of(++iround, G, total);
            if (nt < 0) idone = true; else isetup(nt);
        }
    };
    int tile = tile_of(0, G, total);
    if (tile < 0) return;
    isetup(tile);
    issue_next();
    issue_next();
    const int ff = swz(frow);
    const int choff = (fq ^ ff) << 4;
    long chs = 0;
    int round = 0;
    while (tile >= 0) {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n0 = (tile % n_tiles) * BN, m0 = (tile / n_tiles) * BM;
        for (int hs = 0; hs < nh; ++hs, ++chs) {
            // all but the youngest half-step in flight have landed (6 pieces per half-step and wave; the epilogue's stores are older)
            if (idone && chs + 1 >= ghs) wait_vmcnt<0>(); else wait_vmcnt<6>();
            __builtin_amdgcn_s_barrier();             // chs has landed for everyone; everyone has finished reading slot chs - 1
            issue_next();                             // refills slot (chs + 2) % 3 = the slot read in half-step chs - 1
            const char* sX = smem + (chs % NS) * SLOT;
            const char* sW = sX + XB;
            f16x8 wf[4], xq[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sW + (wn * 64 + i * 16 + frow) * 64 + choff);
            auto ldx = [&](int j) -> f16x8 { return *reinterpret_cast<const f16x8*>(sX + (wm * 128 + j * 16 + frow) * 64 + choff); };
            xq[0] = ldx(0);
            xq[1] = ldx(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xq[j % 3], acc[i][j], 0, 0, 0);
                if (j + 2 < 8) xq[(j + 2) % 3] = ldx(j + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue scratch = the ring slot read in the LAST half-step (the two others are being filled); the barrier makes sure every wave
        // is done reading it, the barrier at the top of the next half-step -- in front of the issue that refills it -- that every wave is
        // done with its scratch (two workgroups x 72 KB per CU leave no room for a separate area)
        __builtin_amdgcn_s_barrier();
        char* tsc = smem + ((chs + NS - 1) % NS) * SLOT + wave * (16 * TP16);
        store_block(acc, out, (long)m0 + wm * 128, n0 + wn * 64, N, tsc, lane);
        tile = tile_of(++round, G, total);
    }
}

template <class Kern>
static float run(const char* name, Kern kern, int grid, int block, size_t lds, const f16* A, const f16* W, f16* out, int M, int N, int K) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, A, W, out, M, N, K);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, A, W, out, M, N, K);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-72s %8.1f us  %6.0f TFLOP/s\n", name, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
    return best;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512;
    const int N = argc > 2 ? atoi(argv[2]) : 1536;
    const int M = argc > 3 ? atoi(argv[3]) : 100864;
    if (M % 256 || N % 256 || K % 64) { printf("need M %% 256 == 0, N %% 256 == 0, K %% 64 == 0\n"); return 1; }
    std::vector<f16> hA((size_t)M * K), hW((size_t)N * K);
    unsigned x = 12345;
    for (auto& v : hA) { x = x * 1664525u + 1013904223u; v = (f16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f); }
    for (auto& v : hW) { x = x * 1664525u + 1013904223u; v = (f16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f); }
    f16 *A, *W, *o1, *o2;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2);
    hipMalloc(&o1, (size_t)M * N * 2); hipMalloc(&o2, (size_t)M * N * 2);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemset(o1, 0xff, (size_t)M * N * 2); hipMemset(o2, 0xff, (size_t)M * N * 2);
    printf("M = %d, N = %d, K = %d, random operands, fp16 output (nontemporal row stores), persistent, XCD-aware tile order\n", M, N, K);
    const size_t lds_base = 2 * (256 * 128 + 256 * 128) + 8 * 16 * TP16;
    const size_t lds_duo = 3 * (256 * 64 + 128 * 64);
    for (int rep = 0; rep < 2; ++rep) {
        run("BASE one workgroup per CU: 8 waves, 256x256 tile, 2 stages of BK = 64", base_kernel, 256, 512, lds_base, A, W, o1, M, N, K);
        run("DUO  two workgroups per CU: 4 waves, 256x128 tile, ring of 3 x BK = 32", duo_kernel, 512, 256, lds_duo, A, W, o2, M, N, K);
    }
    // check: DUO == BASE bit for bit (same k order per output element), and a sample of elements against the host
    std::vector<f16> h1((size_t)M * N), h2((size_t)M * N);
    hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t i = 0; i < h1.size(); ++i) diff += (float)h1[i] != (float)h2[i];
    double worst = 0;
    for (int s = 0; s < 200; ++s) {
        x = x * 1664525u + 1013904223u; const int m = (x >> 8) % M;
        x = x * 1664525u + 1013904223u; const int n = (x >> 8) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)m * K + k] * (double)(float)hW[(size_t)n * K + k];
        worst = std::fmax(worst, std::fabs(ref - (double)(float)h1[(size_t)m * N + n]) / (1.0 + std::fabs(ref)));
    }
    printf("DUO vs BASE: %zu of %zu elements differ; BASE vs host (200 samples): worst rel error %.2e\n", diff, h1.size(), worst);
    printf("LDS per workgroup: BASE %zu B, DUO %zu B (two per CU: %zu)\n", lds_base, lds_duo, 2 * lds_duo);
    return diff == 0 && worst < 2e-3 ? 0 : 2;
}
