// MFMA fp16 GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   out[m][n] = epi( sum_k A[m][k] * (Wh[n][k] + Wl[n][k]) ),   A, W both K-contiguous.
//
// * v_mfma_f32_16x16x32_f16, fp32 accumulate.  The WEIGHT tile is the MFMA "A" operand and the
//   activation tile the "B" operand, so a lane ends up with 4 consecutive n for one m and the
//   epilogue stores 16 B (fp32) / 8 B (fp16) per lane.
// * Workgroup = 4 waves, each wave a 64(m) x 64(n) sub-tile (4x4 MFMA tiles, 64 accumulator
//   VGPRs); block tile 128x128 (2x2 waves) or 256x64 (4x1 waves, for N <= 64 convolutions).
// * K-step 64: both operand tiles are staged through LDS in full 128-B rows (8 lanes x 16 B per
//   row -> whole cache lines from HBM/L2), XOR-swizzled (chunk ^= (row>>1)&7) so the
//   ds_read_b128 fragment reads of all four 16-lane groups are bank-conflict free.
// * Register-staged software pipeline: the global loads of tile t+1 are issued before the MFMAs
//   of tile t and written to LDS after them (guide T14).
// * W2 mode: the fp32 weight is carried as hi+lo fp16 pair (2 MFMAs per k-step) -- weight
//   rounding, not activation rounding, dominates the end-to-end error (precision).
// * CONV: the activation "row" m is an output pixel (img,oh,ow) of an NHWC tensor and
//   k = (kh,kw,c); out-of-image taps are zero-filled at staging time.
#include "common.h"

template <int WM, int WN, bool CONV, bool W2>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    constexpr int BM = 64 * WM, BN = 64 * WN, XR = BM / 32, WR = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;
    char* sWh = smem + BM * 128;
    char* sWl = sWh + BN * 128;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int n_tiles = (a.N + BN - 1) / BN;
    const int n0 = (blockIdx.x % n_tiles) * BN;
    const int m0 = (blockIdx.x / n_tiles) * BM;
    const int c = t & 7, r0 = t >> 3;

    long xbase[XR];
    int ih0[XR], iw0[XR];
    bool xok[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + r0 + 32 * i;
        xok[i] = m < a.M;
        if (CONV) {
            const int per = a.g.OH * a.g.OW;
            const int img = m / per, rem = m - img * per;
            const int oh = rem / a.g.OW, ow = rem - oh * a.g.OW;
            ih0[i] = oh * a.g.SH - a.g.PH;
            iw0[i] = ow * a.g.SW - a.g.PW;
            xbase[i] = (long)img * a.g.H * a.g.W * a.g.C;
        } else {
            ih0[i] = iw0[i] = 0;
            xbase[i] = (long)m * a.lda;
        }
    }
    long wbase[WR];
    bool wok[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int n = n0 + r0 + 32 * i;
        wok[i] = n < a.N;
        wbase[i] = (long)n * a.ldw;
    }

    uint4 xr[XR], whr[WR], wlr[WR];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tile = [&](int kt) {
        const int k = kt * 64 + c * 8;
        const bool kok = k < a.K;
        int kh = 0, kw = 0, ci = 0;
        if (CONV) {
            ci = k & (a.g.C - 1);
            const int kp = k >> a.g.cshift;
            kh = kp / a.g.KW;
            kw = kp - kh * a.g.KW;
        }
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            uint4 v = zero4;
            if (xok[i] && kok) {
                if (CONV) {
                    const int ih = ih0[i] + kh, iw = iw0[i] + kw;
                    if ((unsigned)ih < (unsigned)a.g.H && (unsigned)iw < (unsigned)a.g.W)
                        v = *reinterpret_cast<const uint4*>(a.A + xbase[i] + ((long)ih * a.g.W + iw) * a.g.C + ci);
                } else {
                    v = *reinterpret_cast<const uint4*>(a.A + xbase[i] + k);
                }
            }
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const bool ok = wok[i] && kok;
            whr[i] = ok ? *reinterpret_cast<const uint4*>(a.Wh + wbase[i] + k) : zero4;
            if (W2) wlr[i] = ok ? *reinterpret_cast<const uint4*>(a.Wl + wbase[i] + k) : zero4;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sX + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int r = r0 + 32 * i;
            const int off = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
            *reinterpret_cast<uint4*>(sWh + off) = whr[i];
            if (W2) *reinterpret_cast<uint4*>(sWl + off) = wlr[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + 63) / 64;
    const int frow = lane & 15, fq = lane >> 4;
    const int fsw = (frow >> 1) & 7;   // tile row bases are multiples of 16 -> swizzle depends on lane only

    load_tile(0);
    store_tile();
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int choff = ((kk * 4 + fq) ^ fsw) << 4;
            f16x8 wf[4], wl[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + frow;
                wf[i] = *reinterpret_cast<const f16x8*>(sWh + row * 128 + choff);
                if (W2) wl[i] = *reinterpret_cast<const f16x8*>(sWl + row * 128 + choff);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + frow;
                xf[j] = *reinterpret_cast<const f16x8*>(sX + row * 128 + choff);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                    if (W2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xf[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds D[n = 4*fq + r][m = frow] of each 16x16 tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + fq * 4;
        if (n >= a.N) continue;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
        if (a.bias) bi = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + frow;
            if (m >= a.M) continue;
            f32x4 v = acc[i][j] * sc + bi;
            if (a.res) {
                const int rr = a.res_mod ? (m % a.res_mod) : m;
                v += *reinterpret_cast<const f32x4*>(a.res + (long)rr * a.ldr + n);
            }
            if (a.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (a.out32) *reinterpret_cast<f32x4*>(a.out32 + (long)m * a.ldc + n) = v;
            if (a.out16) {
                f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                *reinterpret_cast<f16x4*>(a.out16 + (long)m * a.ldc + n) = h;
            }
        }
    }
}

template <int WM, int WN, bool CONV, bool W2>
static hipError_t launch_variant(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const long mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)(BM + BN * (W2 ? 2 : 1)) * 128;
    hipLaunchKernelGGL((gemm_kernel<WM, WN, CONV, W2>), dim3((unsigned)(mt * nt)), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, bool conv, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    const bool w2 = a.Wl != nullptr;
    const bool narrow = a.N <= 64;
    if (conv) {
        if (narrow) return w2 ? launch_variant<4, 1, true, true>(a, s) : launch_variant<4, 1, true, false>(a, s);
        return w2 ? launch_variant<2, 2, true, true>(a, s) : launch_variant<2, 2, true, false>(a, s);
    }
    if (narrow) return w2 ? launch_variant<4, 1, false, true>(a, s) : launch_variant<4, 1, false, false>(a, s);
    return w2 ? launch_variant<2, 2, false, true>(a, s) : launch_variant<2, 2, false, false>(a, s);
}
