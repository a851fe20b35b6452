// Micro-benchmark (round 4): the k loop of the LN-fused GEMM's row-wide 128 x 512 tile (8 waves side by side along n, each all 128
// rows), operands L2-resident (the 512 x K weights are shared by EVERY workgroup, the 128-row activation panel is private), no
// epilogue.  tools/kernel_power.py shows this kernel is the one MFMA kernel of the path that is NOT power-bound (1 276 W at the full
// 2.38 GHz), so unlike the 256 x 256 tile (tools/kloop2.hip) its k loop can gain from a better pipeline: a k-tile brings in 80 KB
// (16 KB activations + 64 KB weights) and takes 1.75-2.1 us in the library = its MFMA time (0.86 us) PLUS its DMA time at the path's
// barrier-regime rate (0.92 us): the two do not overlap.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/kloop_ln.hip -o tools/bin/kloop_ln
//   L0  two 80 KB stages, vmcnt(0) + barrier per k-tile, the 10 pieces of the next k-tile in one burst behind the barrier
//   L1  the same, one piece per MFMA group (the library's schedule)
//   L2  ring of 4 half-stages (BK = 32, 40 KB each), 2 half-steps in flight across the barriers
//   L3  ring of 4 half-stages, 3 half-steps in flight
//   L4  L0 with the burst issued BEFORE the barrier wait of the current k-tile's ... (not possible with 2 stages) -> omitted
//   L5  two stages, but each wave waits only for ITS OWN share of the weights: the weight rows a wave multiplies are the rows it
//       loaded itself (wave w owns columns 64w..64w+63 = weight rows 64w..), so only the 16 KB activation tile needs the barrier;
//       weights: per-wave vmcnt, no cross-wave dependency.  Barrier per k-tile stays (stage reuse) but the weight DMA of k-tile kt+2
//       can be issued right after the wave's own last read of stage kt -- two k-tiles of weights in flight.  [3 weight slots per wave]
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

constexpr int BM = 128, BN = 512;

template <int V>
__global__ __launch_bounds__(512) void kloop(const f16* __restrict__ A, const f16* __restrict__ W, float* __restrict__ sink, int M, int K, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave;
    const int frow = lane & 15, fq = lane >> 4;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int G = gridDim.x;
    auto tile_of = [&](int round) -> int { return (round * G + blockIdx.x) % tiles; };
    const int rounds = 64;
    if constexpr (V == 0 || V == 1) {
        constexpr int XB = BM * 128, STAGE = XB + BN * 128;          // 16 KB + 64 KB
        const int lrow = lane >> 3, pc = lane & 7;
        const int fsw = (frow >> 1) & 7;
        const int nk = K / 64;
        const f16* xs[2]; const f16* ws[8];
        auto setup = [&](int tile) {
            const int m0 = tile * BM;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave * 2 + i) * 8 + lrow;
                xs[i] = A + (long)(m0 + row) * K + (pc ^ ((row >> 1) & 7)) * 8;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (wave * 8 + i) * 8 + lrow;
                ws[i] = W + (long)row * K + (pc ^ ((row >> 1) & 7)) * 8;
            }
        };
        auto piece = [&](int p, int kt, int buf) __attribute__((always_inline)) {
            char* base = smem + buf * STAGE;
            if (p < 2) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[p] + kt * 64), (lds_ptr_t)(base + (wave * 2 + p) * 1024), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws[p - 2] + kt * 64), (lds_ptr_t)(base + XB + (wave * 8 + p - 2) * 1024), 16, 0, 0);
        };
        int gk = 0;
        setup(tile_of(0));
#pragma unroll
        for (int p = 0; p < 10; ++p) piece(p, 0, 0);
        for (int r = 0; r < rounds; ++r) {
            for (int kt = 0; kt < nk; ++kt, ++gk) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                const bool last = kt + 1 == nk;
                if (last) setup(tile_of(r + 1));
                const int nkt = last ? 0 : kt + 1;
                if (V == 0) {
#pragma unroll
                    for (int p = 0; p < 10; ++p) piece(p, nkt, (gk + 1) & 1);
                }
                const char* sX = smem + (gk & 1) * STAGE;
                const char* sW = sX + XB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int choff = ((kk * 4 + fq) ^ fsw) << 4;
                    auto ldx = [&](int j) -> f16x8 { return *reinterpret_cast<const f16x8*>(sX + (j * 16 + frow) * 128 + choff); };
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
                        if (V == 1 && (kk * 8 + j) < 10) piece(kk * 8 + j, nkt, (gk + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    } else if constexpr (V == 2 || V == 3) {
        // ---- ring of 4 half-stages (BK = 32): rows of 64 B, pieces of 16 rows, swizzle chunk ^= f[(row >> 2) & 3], f = {0,2,3,1}
        constexpr int NS = 4;
        constexpr int XB = BM * 64, SLOT = XB + BN * 64;             // 8 KB + 32 KB
        constexpr int DIST = V == 2 ? 2 : 3;
        const int lrow = lane >> 2, pc = lane & 3;
        const int nh = K / 32;
        const f16* xs[1]; const f16* ws[4];
        auto fsel = [](int row) -> int { const int g = (row >> 2) & 3; return g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 3 : 1; };
        auto setup = [&](int tile) {
            const int m0 = tile * BM;
            { const int row = wave * 16 + lrow; xs[0] = A + (long)(m0 + row) * K + (pc ^ fsel(row)) * 8; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int row = (wave * 4 + i) * 16 + lrow; ws[i] = W + (long)row * K + (pc ^ fsel(row)) * 8; }
        };
        long ghs = 0;
        auto stage = [&](int hs, long slot) __attribute__((always_inline)) {
            char* base = smem + (slot % NS) * SLOT;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[0] + hs * 32), (lds_ptr_t)(base + wave * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws[i] + hs * 32), (lds_ptr_t)(base + XB + (wave * 4 + i) * 1024), 16, 0, 0);
        };
        const int choff = (fq ^ fsel(frow)) << 4;
        int ir = 0, ih = 0;
        setup(tile_of(0));
        auto issue_next = [&]() __attribute__((always_inline)) {
            stage(ih, ghs);
            ++ghs;
            if (++ih == nh) { ih = 0; ++ir; setup(tile_of(ir)); }
        };
#pragma unroll
        for (int d = 0; d < DIST; ++d) issue_next();
        long chs = 0;
        for (int r = 0; r < rounds; ++r) {
            for (int hs = 0; hs < nh; ++hs, ++chs) {
                if (DIST == 3) wait_vmcnt<10>(); else wait_vmcnt<5>();
                __builtin_amdgcn_s_barrier();
                issue_next();
                const char* sX = smem + (chs % NS) * SLOT;
                const char* sW = sX + XB;
                f16x8 wf[4], xq[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sW + (wn * 64 + i * 16 + frow) * 64 + choff);
                auto ldx = [&](int j) -> f16x8 { return *reinterpret_cast<const f16x8*>(sX + (j * 16 + frow) * 64 + choff); };
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
    } else if constexpr (V == 5) {
        // ---- wave-private weights: wave w multiplies weight rows 64w .. 64w+63 only, and it can load exactly those rows itself.
        // LDS: 2 activation stages (16 KB each, shared, barrier) + per wave 3 weight slots of 8 KB (64 rows x 128 B) = 32 + 192 KB?
        // -> too much; 2 activation stages + per wave TWO weight slots of 8 KB = 32 + 128 = 160 KB: same footprint as the library.
        // The gain is in the dependency structure: a wave's weight DMA for k-tile kt+1 is issued as soon as IT has read slot (kt+1)&1
        // for the last time (its own k-tile kt-1), needs no barrier, and is waited for with the wave's own vmcnt; only the 2
        // activation pieces per wave cross waves.
        constexpr int XB = BM * 128;                                  // 16 KB per activation stage
        constexpr int WSLOT = 64 * 128;                               // 8 KB per wave and slot
        char* const sA = smem;                                        // 2 stages
        char* const sWv = smem + 2 * XB + wave * 2 * WSLOT;           // this wave's 2 weight slots
        const int lrow = lane >> 3, pc = lane & 7;
        const int fsw = (frow >> 1) & 7;
        const int nk = K / 64;
        const f16* xs[2]; const f16* ws[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + lrow;                              // row within the wave's 64
            ws[i] = W + (long)(wave * 64 + row) * K + (pc ^ ((row >> 1) & 7)) * 8;
        }
        auto setup = [&](int tile) {
            const int m0 = tile * BM;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave * 2 + i) * 8 + lrow;
                xs[i] = A + (long)(m0 + row) * K + (pc ^ ((row >> 1) & 7)) * 8;
            }
        };
        auto wpieces = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws[i] + kt * 64), (lds_ptr_t)(sWv + buf * WSLOT + i * 1024), 16, 0, 0);
        };
        auto apieces = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[i] + kt * 64), (lds_ptr_t)(sA + buf * XB + (wave * 2 + i) * 1024), 16, 0, 0);
        };
        int gk = 0;
        setup(tile_of(0));
        apieces(0, 0);
        wpieces(0, 0);
        for (int r = 0; r < rounds; ++r) {
            for (int kt = 0; kt < nk; ++kt, ++gk) {
                wait_vmcnt<0>();                                       // own weights of this k-tile + own activation pieces
                __builtin_amdgcn_s_barrier();                          // everyone's activation pieces
                const bool last = kt + 1 == nk;
                if (last) setup(tile_of(r + 1));
                const int nkt = last ? 0 : kt + 1;
                apieces(nkt, (gk + 1) & 1);
                wpieces(nkt, (gk + 1) & 1);
                const char* sX = sA + (gk & 1) * XB;
                const char* sW = sWv + (gk & 1) * WSLOT;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int choff = ((kk * 4 + fq) ^ fsw) << 4;
                    auto ldx = [&](int j) -> f16x8 { return *reinterpret_cast<const f16x8*>(sX + (j * 16 + frow) * 128 + choff); };
                    f16x8 wf[4], xq[3];
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sW + (i * 16 + frow) * 128 + choff);
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
        }
    } else if constexpr (V == 6) {
        // ---- weights straight into REGISTERS (no LDS for them): wave w needs, per k-step of 32, four 16 x 32 fragments of ITS 64
        // weight rows = per lane 4 x 16 B at W[64w + 16i + frow][k0 + 8 fq ..]: global_load_dwordx4, 64-byte segments per row.
        // Only the 16 KB activation tile goes through LDS (2 pieces per wave and k-tile).  LDS-DMA volume per k-tile: 16 KB instead
        // of 80; the weights come through the vector-memory path into VGPRs, one k-step ahead (8 fragments in flight = 32 VGPRs).
        constexpr int XB = BM * 128;
        const int lrow = lane >> 3, pc = lane & 7;
        const int fsw = (frow >> 1) & 7;
        const int nk = K / 64;
        const f16* xs[2];
        const f16* wrow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wrow[i] = W + (long)(wave * 64 + i * 16 + frow) * K + fq * 8;
        auto setup = [&](int tile) {
            const int m0 = tile * BM;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave * 2 + i) * 8 + lrow;
                xs[i] = A + (long)(m0 + row) * K + (pc ^ ((row >> 1) & 7)) * 8;
            }
        };
        auto apieces = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs[i] + kt * 64), (lds_ptr_t)(smem + buf * XB + (wave * 2 + i) * 1024), 16, 0, 0);
        };
        auto ldw = [&](int i, int kstep) -> f16x8 { return *reinterpret_cast<const f16x8*>(wrow[i] + kstep * 32); };
        int gk = 0;
        setup(tile_of(0));
        apieces(0, 0);
        f16x8 wcur[4], wnxt[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wcur[i] = ldw(i, 0);
        for (int r = 0; r < rounds; ++r) {
            for (int kt = 0; kt < nk; ++kt, ++gk) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                const bool last = kt + 1 == nk;
                if (last) setup(tile_of(r + 1));
                const int nkt = last ? 0 : kt + 1;
                apieces(nkt, (gk + 1) & 1);
                const char* sX = smem + (gk & 1) * XB;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int choff = ((kk * 4 + fq) ^ fsw) << 4;
                    auto ldx = [&](int j) -> f16x8 { return *reinterpret_cast<const f16x8*>(sX + (j * 16 + frow) * 128 + choff); };
                    // next k-step's weights (the k walk wraps inside the tile's K: same weights for every tile)
                    const int ks_next = (kt * 2 + kk + 1) % (2 * nk);
#pragma unroll
                    for (int i = 0; i < 4; ++i) wnxt[i] = ldw(i, ks_next);
                    f16x8 xq[3];
                    xq[0] = ldx(0);
                    xq[1] = ldx(1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur[i], xq[j % 3], acc[i][j], 0, 0, 0);
                        if (j + 2 < 8) xq[(j + 2) % 3] = ldx(j + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) wcur[i] = wnxt[i];
                }
            }
        }
    }
    wait_vmcnt<0>();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j].x + acc[i][j].w;
    if (s == 123.456f) sink[0] = s;
}

template <int V>
static void run(const char* name, const f16* A, const f16* W, float* sink, int M, int K) {
    const size_t lds = 163840;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kloop<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = M / BM;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kloop<V>), dim3(256), dim3(512), lds, 0, A, W, sink, M, K, tiles);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kloop<V>), dim3(256), dim3(512), lds, 0, A, W, sink, M, K, tiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double kt = 64.0 * (K / 64);
    printf("%-72s %8.1f us  %6.3f us per k-tile  %6.0f TFLOP/s\n", name, best * 1e3, best * 1e3 / kt, 256.0 * kt * 2.0 * BM * BN * 64 / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512;
    const int M = 100864 / 128 * 128, N = BN;
    std::vector<f16> hA((size_t)M * K), hW((size_t)N * K);
    unsigned x = 12345;
    for (auto& v : hA) { x = x * 1664525u + 1013904223u; v = (f16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f); }
    for (auto& v : hW) { x = x * 1664525u + 1013904223u; v = (f16)(((x >> 9) & 0xffff) / 65536.0f - 0.5f); }
    f16 *A, *W; float* sink;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&sink, 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    printf("128 x 512 tile, K = %d (random operands), 256 workgroups x 64 tiles, no epilogue\n", K);
    run<0>("L0 2 stages, barrier per k-tile, 10 pieces in a burst (warm-up)", A, W, sink, M, K);
    run<0>("L0 2 stages, barrier per k-tile, 10 pieces in a burst", A, W, sink, M, K);
    run<1>("L1 2 stages, pieces spread one per MFMA group (the library)", A, W, sink, M, K);
    run<2>("L2 ring of 4 half-stages, 2 half-steps in flight", A, W, sink, M, K);
    run<3>("L3 ring of 4 half-stages, 3 half-steps in flight", A, W, sink, M, K);
    run<5>("L5 wave-private weight slots (only the activations cross waves)", A, W, sink, M, K);
    run<6>("L6 weights straight into registers (global_load), activations via LDS", A, W, sink, M, K);
    run<0>("L0 again", A, W, sink, M, K);
    return 0;
}
