// fp32 audit kernels (gfx950): see audit32.h.  Exact fp32 products on v_mfma_f32_32x32x2_f32, fp32 everywhere else.
#include "audit32.h"
#include <type_traits>

namespace {

__device__ __forceinline__ float act1(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}

// ---------------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution.  Block tile 128 (m) x 128 (n), 4 waves of 64 x 64 (2 x 2 MFMA blocks of 32 x 32), k-step 16 through
// LDS stored [k][row] so the one-float MFMA fragment reads (lane -> row l & 31, k l >> 5) are conflict free; the next k-tile's
// global loads are issued before the MFMAs of the current one.  The WEIGHT tile is the MFMA "A" operand: a lane ends up with four
// consecutive n of one m per accumulator quad (16-byte stores).
// VEC: every 4-aligned k-quad of a row is contiguous in memory and 16-byte aligned (plain: lda % 4 == 0; conv: C % 4 == 0).
constexpr int G32_BK = 16;

// BT = block tile edge (128: 4 waves of 64 x 64 = 2 x 2 MFMA blocks each; 64: 4 waves of 32 x 32, for the small-M launches of the JEGAL
// branch -- 4 x the workgroups, 2 x the LDS reads per MFMA, which the fp32 MFMA's 64 cycles hide)
template <bool CONV, bool VEC, int BT>
__global__ __launch_bounds__(256) void gemm32_kernel(Gemm32Args a) {
    constexpr int LD = BT + 4, NB = BT / 64, NL = BT / 64;     // MFMA blocks per wave along n / m; loads (float4) per thread and operand
    __shared__ float sX[G32_BK][LD];
    __shared__ float sW[G32_BK][LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int n_tiles = (a.N + BT - 1) / BT;
    const int n0 = (blockIdx.x % n_tiles) * BT;
    const int m0 = (int)(blockIdx.x / n_tiles) * BT;
    // loader: float4 number idx = t + 256 u -> (row idx >> 2, k quad (idx & 3) * 4) of the BT x 16 tile
    long xbase[NL], wbase[NL];
    int ih0[NL], iw0[NL];
    bool xok[NL], wok[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int r = (t + 256 * u) >> 2;
        const int mrow = m0 + r, nrow = n0 + r;
        xok[u] = mrow < a.M;
        wok[u] = nrow < a.N;
        ih0[u] = iw0[u] = 0;
        if (CONV) {
            const int mm = xok[u] ? mrow : 0;
            const int per = a.g.OH * a.g.OW;
            const int img = mm / per, rem = mm - img * per;
            const int oh = rem / a.g.OW, ow = rem - oh * a.g.OW;
            ih0[u] = oh * a.g.SH - a.g.PH;
            iw0[u] = ow * a.g.SW - a.g.PW;
            xbase[u] = (long)img * a.g.H * a.g.W * a.g.C;
        } else {
            xbase[u] = (long)(xok[u] ? mrow : 0) * a.lda;
        }
        wbase[u] = (long)(wok[u] ? nrow : 0) * a.ldw;
    }
    const int kq = (t & 3) * 4;

    auto load_x1 = [&](int u, int k) -> float {
        if (!xok[u] || k >= a.K) return 0.f;
        if (CONV) {
            const int ci = k & (a.g.C - 1), kp = k >> a.g.cshift;
            int kh, kw;
            tap_decode(a.g, kp, kh, kw);
            const int ih = ih0[u] + kh, iw = iw0[u] + kw;
            if ((unsigned)ih >= (unsigned)a.g.H || (unsigned)iw >= (unsigned)a.g.W || kh >= a.g.KH) return 0.f;
            return a.A[xbase[u] + ((long)ih * a.g.W + iw) * a.g.C + ci];
        }
        return a.A[xbase[u] + k];
    };
    auto load_x4 = [&](int u, int k) -> f32x4 {
        if (VEC) {
            if (!xok[u] || k >= a.K) return f32x4{0.f, 0.f, 0.f, 0.f};
            if (CONV) {
                const int ci = k & (a.g.C - 1), kp = k >> a.g.cshift;
                int kh, kw;
                tap_decode(a.g, kp, kh, kw);
                const int ih = ih0[u] + kh, iw = iw0[u] + kw;
                if ((unsigned)ih >= (unsigned)a.g.H || (unsigned)iw >= (unsigned)a.g.W || kh >= a.g.KH) return f32x4{0.f, 0.f, 0.f, 0.f};
                return *reinterpret_cast<const f32x4*>(a.A + xbase[u] + ((long)ih * a.g.W + iw) * a.g.C + ci);
            }
            return *reinterpret_cast<const f32x4*>(a.A + xbase[u] + k);
        }
        return f32x4{load_x1(u, k), load_x1(u, k + 1), load_x1(u, k + 2), load_x1(u, k + 3)};
    };
    auto load_w4 = [&](int u, int k) -> f32x4 {
        if (!wok[u]) return f32x4{0.f, 0.f, 0.f, 0.f};
        if (VEC && k + 3 < a.K) return *reinterpret_cast<const f32x4*>(a.W + wbase[u] + k);
        f32x4 v;
        v.x = k < a.K ? a.W[wbase[u] + k] : 0.f;
        v.y = k + 1 < a.K ? a.W[wbase[u] + k + 1] : 0.f;
        v.z = k + 2 < a.K ? a.W[wbase[u] + k + 2] : 0.f;
        v.w = k + 3 < a.K ? a.W[wbase[u] + k + 3] : 0.f;
        return v;
    };

    f32x16 acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

    const int nk = (a.K + G32_BK - 1) / G32_BK;
    f32x4 xr[NL], wr[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        xr[u] = load_x4(u, kq);
        wr[u] = load_w4(u, kq);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int r = (t + 256 * u) >> 2;
            sX[kq + 0][r] = xr[u].x; sX[kq + 1][r] = xr[u].y; sX[kq + 2][r] = xr[u].z; sX[kq + 3][r] = xr[u].w;
            sW[kq + 0][r] = wr[u].x; sW[kq + 1][r] = wr[u].y; sW[kq + 2][r] = wr[u].z; sW[kq + 3][r] = wr[u].w;
        }
        __syncthreads();
        if (kt + 1 < nk) {
            const int k = (kt + 1) * G32_BK + kq;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                xr[u] = load_x4(u, k);
                wr[u] = load_w4(u, k);
            }
        }
#pragma unroll
        for (int kk = 0; kk < G32_BK; kk += 2) {
            const int kr = kk + (lane >> 5);
            float w[NB], x[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                w[i] = sW[kr][wn * (BT / 2) + i * 32 + (lane & 31)];
                x[i] = sX[kr][wm * (BT / 2) + i * 32 + (lane & 31)];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i], x[j], acc[i][j], 0, 0, 0);
        }
    }
    // D layout: B index (m) = lane & 31, A index (n) = (x & 3) + 8 (x >> 2) + 4 (lane >> 5)
    const bool n4ok = (a.N & 3) == 0 && (a.ldc & 3) == 0 && (!a.res || (a.ldr & 3) == 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int m = m0 + wm * (BT / 2) + j * 32 + (lane & 31);
        if (m >= a.M) continue;
        const long rr = a.res ? (long)(a.res_mod ? m % a.res_mod : m) * a.ldr : 0;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * (BT / 2) + i * 32 + 8 * q + 4 * (lane >> 5);
                if (n >= a.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = acc[i][j][q * 4 + e];
                    if (n + e < a.N) {
                        if (a.scale) y *= a.scale[n + e];
                        if (a.bias) y += a.bias[n + e];
                        if (a.res) y += a.res[rr + n + e];
                    }
                    v[e] = act1(y, a.act);
                }
                if (n4ok) {
                    *reinterpret_cast<f32x4*>(a.out + (long)m * a.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.N) a.out[(long)m * a.ldc + n + e] = v[e];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Attention, one thread per query row, keys / values through LDS in chunks of KC, online softmax (fp32).  A block holds the queries of
// ONE (sequence, head); blockDim = min(256, S rounded up to 64).
template <int DK>
__global__ __launch_bounds__(256) void attention32_kernel(const float* __restrict__ qkv, const float* __restrict__ keymask, int S, int H,
                                                          float* __restrict__ out) {
    constexpr int KC = 32;
    __shared__ float sK[KC][DK];
    __shared__ float sV[KC][DK];
    __shared__ float sM[KC];
    const int bh = blockIdx.x, b = bh / H, hd = bh - b * H;
    const int D = H * DK;
    const int qi = blockIdx.y * blockDim.x + threadIdx.x;
    const bool live = qi < S;
    const float scale = 1.0f / sqrtf((float)DK);
    float q[DK], o[DK];
    const float* qp = qkv + ((long)b * S + (live ? qi : 0)) * 3 * D + hd * DK;
#pragma unroll
    for (int d = 0; d < DK; ++d) { q[d] = qp[d]; o[d] = 0.f; }
    float mx = -INFINITY, l = 0.f;
    for (int c0 = 0; c0 < S; c0 += KC) {
        __syncthreads();
        for (int i = threadIdx.x; i < KC * DK; i += blockDim.x) {
            const int j = i / DK, d = i - j * DK;
            const bool ok = c0 + j < S;
            const float* kp = qkv + ((long)b * S + (ok ? c0 + j : 0)) * 3 * D + D + hd * DK + d;
            sK[j][d] = ok ? kp[0] : 0.f;
            sV[j][d] = ok ? kp[D] : 0.f;
        }
        for (int j = threadIdx.x; j < KC; j += blockDim.x) sM[j] = (c0 + j < S && keymask) ? keymask[(long)b * S + c0 + j] : 1.f;
        __syncthreads();
        const int nj = S - c0 < KC ? S - c0 : KC;
        for (int j = 0; j < nj; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < DK; ++d) sc = __builtin_fmaf(q[d], sK[j][d], sc);
            sc *= scale;
            if (sM[j] == 0.f) sc = -1e9f;                 // masked_fill(mask == 0, -1e9), modules.py:69-70
            const float mn = fmaxf(mx, sc);
            const float corr = expf(mx - mn), p = expf(sc - mn);
            l = l * corr + p;
#pragma unroll
            for (int d = 0; d < DK; ++d) o[d] = __builtin_fmaf(p, sV[j][d], o[d] * corr);
            mx = mn;
        }
    }
    if (!live) return;
    const float inv = 1.f / l;
    float* op = out + ((long)b * S + qi) * D + hd * DK;
#pragma unroll
    for (int d = 0; d < DK; ++d) op[d] = o[d] * inv;
}


// ---------------------------------------------------------------------------------------------
// Split-operand GEMM (GemmX3Args, audit32.h).  These launches are small (M = B * T rows of the JEGAL branch) and LATENCY-bound: a first
// version with 64 x 128 tiles and one k-tile of prefetch took 63 us for 4800 x 512 x 512 (3.8 us per k-tile: one exposed memory round
// trip each, ~1 workgroup per CU).  Hence: tile 32 (m) x 128 (n) x 64 (k) for 4 x the workgroups, and a ring of FOUR register stages --
// the loads of k-tiles kt + 1 .. kt + 3 are in flight while k-tile kt is split, staged and multiplied.  4 waves side by side along n
// (each 32 m x 32 n = 2 x 2 tiles of v_mfma_f32_16x16x32_f16), operands through LDS in 128-B rows with the XOR swizzle of gemm.hip.
// K % 256 == 0 (the k loop is unrolled by the ring depth).
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmX3Args a) {
    constexpr int BM = 32, BN = 128, DEPTH = 4;
    __shared__ __attribute__((aligned(16))) char sXh[BM * 128];
    __shared__ __attribute__((aligned(16))) char sXl[BM * 128];
    __shared__ __attribute__((aligned(16))) char sWh[BN * 128];
    __shared__ __attribute__((aligned(16))) char sWl[BN * 128];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n_tiles = a.N / BN;
    const int n0 = (blockIdx.x % n_tiles) * BN;
    const int m0 = (int)(blockIdx.x / n_tiles) * BM;
    // A loader: row t >> 3, the 8 consecutive k of 16-B chunk t & 7 (two float4)
    const int xr = t >> 3, xc = t & 7;
    const bool xok = m0 + xr < a.M;
    const float* xp = a.A + (long)(xok ? m0 + xr : 0) * a.lda + xc * 8;
    // W loader: row t >> 1, chunks 4 (t & 1) .. + 3
    const int wr = t >> 1, wc = (t & 1) * 4;
    const f16* whp = a.Wh + (long)(n0 + wr) * a.ldw + (t & 1) * 32;
    const f16* wlp = a.Wl + (long)(n0 + wr) * a.ldw + (t & 1) * 32;

    f32x4 xv[DEPTH][2];
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t whv[DEPTH][4], wlv[DEPTH][4];
    static_assert(DEPTH == 4, "the k loop below is written out for a ring of four stages");
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = a.K / 64;
    const int frow = lane & 15, fq = lane >> 4, fsw = (frow >> 1) & 7;
    // (written as macros with literal stage indices: through lambdas, or with run-time indices, hipcc kept the ring in scratch memory)
#define X3_LOAD(ST, KT)                                                                                          \
    {                                                                                                            \
        const int k_ = (KT) * 64;                                                                                \
        xv[ST][0] = xok ? *reinterpret_cast<const f32x4*>(xp + k_) : f32x4{0.f, 0.f, 0.f, 0.f};                  \
        xv[ST][1] = xok ? *reinterpret_cast<const f32x4*>(xp + k_ + 4) : f32x4{0.f, 0.f, 0.f, 0.f};              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                          \
            whv[ST][q] = *reinterpret_cast<const u32x4_t*>(whp + k_ + 8 * q);                                   \
            wlv[ST][q] = *reinterpret_cast<const u32x4_t*>(wlp + k_ + 8 * q);                                   \
        }                                                                                                        \
    }
#define X3_TILE(ST, KT)                                                                                          \
    {                                                                                                            \
        if ((KT) + DEPTH - 1 < nk) X3_LOAD(((ST) + DEPTH - 1) % DEPTH, (KT) + DEPTH - 1)                          \
        __syncthreads();                                                                                         \
        {                                                                                                        \
            const f32x4 x0_ = xv[ST][0], x1_ = xv[ST][1];                                                        \
            const f16x8 hi_ = {(f16)x0_.x, (f16)x0_.y, (f16)x0_.z, (f16)x0_.w, (f16)x1_.x, (f16)x1_.y, (f16)x1_.z, (f16)x1_.w};          \
            const f16x8 lo_ = {(f16)(x0_.x - (float)hi_[0]), (f16)(x0_.y - (float)hi_[1]), (f16)(x0_.z - (float)hi_[2]), (f16)(x0_.w - (float)hi_[3]),   \
                               (f16)(x1_.x - (float)hi_[4]), (f16)(x1_.y - (float)hi_[5]), (f16)(x1_.z - (float)hi_[6]), (f16)(x1_.w - (float)hi_[7])};  \
            const int off_ = xr * 128 + ((xc ^ ((xr >> 1) & 7)) << 4);                                           \
            *reinterpret_cast<f16x8*>(sXh + off_) = hi_;                                                         \
            *reinterpret_cast<f16x8*>(sXl + off_) = lo_;                                                         \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
                const int o2_ = wr * 128 + (((wc + q) ^ ((wr >> 1) & 7)) << 4);                                  \
                *reinterpret_cast<u32x4_t*>(sWh + o2_) = whv[ST][q];                                            \
                *reinterpret_cast<u32x4_t*>(sWl + o2_) = wlv[ST][q];                                            \
            }                                                                                                    \
        }                                                                                                        \
        __syncthreads();                                                                                         \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                       \
            const int choff = ((kk * 4 + fq) ^ fsw) << 4;                                                        \
            f16x8 wf[2], wl[2], xh[2], xl[2];                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
                const int row = wave * 32 + i * 16 + frow;                                                       \
                wf[i] = *reinterpret_cast<const f16x8*>(sWh + row * 128 + choff);                                \
                wl[i] = *reinterpret_cast<const f16x8*>(sWl + row * 128 + choff);                                \
            }                                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                      \
                const int row = j * 16 + frow;                                                                   \
                xh[j] = *reinterpret_cast<const f16x8*>(sXh + row * 128 + choff);                                \
                xl[j] = *reinterpret_cast<const f16x8*>(sXl + row * 128 + choff);                                \
            }                                                                                                    \
            /* smallest terms first: the two cross terms, then the main product */                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
                    acc[i][j] = JG_MFMA_16x16x32(wl[i], xh[j], acc[i][j]);                                       \
                    acc[i][j] = JG_MFMA_16x16x32(wf[i], xl[j], acc[i][j]);                                       \
                    acc[i][j] = JG_MFMA_16x16x32(wf[i], xh[j], acc[i][j]);                                       \
                }                                                                                                \
        }                                                                                                        \
    }
    X3_LOAD(0, 0)          // nk >= DEPTH (K % 256 == 0)
    X3_LOAD(1, 1)
    X3_LOAD(2, 2)
    for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
        X3_TILE(0, kt0)
        X3_TILE(1, kt0 + 1)
        X3_TILE(2, kt0 + 2)
        X3_TILE(3, kt0 + 3)
    }
#undef X3_TILE
#undef X3_LOAD
    // lane holds D[n = 4 fq + r][m = frow] of each 16 x 16 tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + wave * 32 + i * 16 + fq * 4;
        const f32x4 bi = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + j * 16 + frow;
            if (m >= a.M) continue;
            f32x4 v = acc[i][j] + bi;
            if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + (long)(a.res_mod ? m % a.res_mod : m) * a.ldr + n);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<f32x4*>(a.out + (long)m * a.ldc + n) = v;
        }
    }
}

template <typename SRC>
__global__ void stack_frames32_kernel(const SRC* __restrict__ src, long sb, long st, long sh, long sw, long sc, int B, int T, int pad, int H, int W,
                                      float* __restrict__ dst) {
    const int P = T + 2 * pad - 4;
    const long total = (long)B * P * H * W;
    constexpr bool U8 = sizeof(SRC) == 1;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int w = idx % W;
        long r = idx / W;
        const int h = r % H;
        r /= H;
        const int p = r % P;
        const int b = r / P;
        float v[16];
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
            int f = p + dt - pad;
            f = f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
            const SRC* s = src + b * sb + f * st + h * sh + w * sw;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[dt * 3 + c] = U8 ? (float)s[c * sc] / 255.0f : (float)s[c * sc];      // IEEE division, as numpy's / 255.
        }
        v[15] = 0.f;
        f32x4* d = reinterpret_cast<f32x4*>(dst + idx * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = f32x4{v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
    }
}

__global__ void maxpool32_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C, int OH, int OW) {
    const int cv = C / 4;
    const long total = (long)N * OH * OW * cv;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = idx % cv;
        long r = idx / cv;
        const int ow = r % OW;
        r /= OW;
        const int oh = r % OH;
        const long n = r / OH;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((n * H + oh * 2 + kh) * W + ow * 2 + kw) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        *reinterpret_cast<f32x4*>(out + idx * 4) = m;
    }
}

__global__ void group_mean32_kernel(const float* __restrict__ in, int groups, int L, int D, float* __restrict__ out) {
    const int dv = D / 4;
    const long total = (long)groups * dv;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int d4 = idx % dv;
        const long g = idx / dv;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < L; ++j) acc += *reinterpret_cast<const f32x4*>(in + (g * L + j) * D + d4 * 4);
        *reinterpret_cast<f32x4*>(out + idx * 4) = acc / (float)L;
    }
}

__global__ __launch_bounds__(256) void zero_tail32_kernel(float* __restrict__ x, const int* __restrict__ valid, int halvings, int H, int row_vec) {
    const int b = blockIdx.x / H, hrow = blockIdx.x - b * H;
    int len = max(valid[b], 0);
    for (int i = 0; i < halvings; ++i) len = len > 0 ? (len - 1) / 2 + 1 : 0;
    if (hrow < len) return;
    f32x4* p = reinterpret_cast<f32x4*>(x) + (long)blockIdx.x * row_vec;
    for (int i = threadIdx.x; i < row_vec; i += 256) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ void poison_kernel(uint4* __restrict__ p, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
}

inline int grid_for(long total) { return (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4); }

}  // namespace

template <int BT>
static hipError_t launch_gemm32_bt(const Gemm32Args& a, hipStream_t s) {
    const long mt = (a.M + BT - 1) / BT, nt = (a.N + BT - 1) / BT;
    const dim3 grid((unsigned)(mt * nt)), block(256);
    const bool walign = (a.ldw & 3) == 0 && ((uintptr_t)a.W & 15) == 0 && ((uintptr_t)a.A & 15) == 0;
    if (a.conv) {
        if ((1 << a.g.cshift) != a.g.C || a.g.rowmap || a.g.const_in) return hipErrorInvalidValue;
        if (walign && (a.g.C & 3) == 0) hipLaunchKernelGGL((gemm32_kernel<true, true, BT>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm32_kernel<true, false, BT>), grid, block, 0, s, a);
    } else {
        if (walign && (a.lda & 3) == 0 && (a.K & 3) == 0) hipLaunchKernelGGL((gemm32_kernel<false, true, BT>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm32_kernel<false, false, BT>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_gemm32(const Gemm32Args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (a.K <= 0 || !a.A || !a.W || !a.out) return hipErrorInvalidValue;
    // fewer than ~1.5 rounds of 128 x 128 tiles on 256 CUs: 64 x 64 tiles (the small-M launches of the JEGAL branch)
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (tiles128 < 384) return launch_gemm32_bt<64>(a, s);
    return launch_gemm32_bt<128>(a, s);
}

hipError_t launch_attention32(const float* qkv, const float* keymask, int B, int S, int H, int dk, float* out, hipStream_t s) {
    if (B <= 0 || S <= 0) return hipSuccess;
    const int threads = S >= 256 ? 256 : (S + 63) / 64 * 64;
    const dim3 grid((unsigned)(B * H), (unsigned)((S + threads - 1) / threads));
    if (dk == 64) hipLaunchKernelGGL(attention32_kernel<64>, grid, dim3(threads), 0, s, qkv, keymask, S, H, out);
    else if (dk == 96) hipLaunchKernelGGL(attention32_kernel<96>, grid, dim3(threads), 0, s, qkv, keymask, S, H, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_stack_frames32(const void* src, int src_is_u8, long sb, long st, long sh, long sw, long sc, int B, int T, int pad, int H, int W,
                                 float* dst, hipStream_t s) {
    const long total = (long)B * (T + 2 * pad - 4) * H * W;
    if (total <= 0) return hipSuccess;
    if (src_is_u8)
        hipLaunchKernelGGL(stack_frames32_kernel<uint8_t>, dim3(grid_for(total)), dim3(256), 0, s, (const uint8_t*)src, sb, st, sh, sw, sc, B, T, pad, H, W, dst);
    else
        hipLaunchKernelGGL(stack_frames32_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)src, sb, st, sh, sw, sc, B, T, pad, H, W, dst);
    return hipGetLastError();
}

hipError_t launch_maxpool3x3s2_32(const float* in, float* out, int N, int H, int W, int C, hipStream_t s) {
    if (C % 4) return hipErrorInvalidValue;
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const long total = (long)N * OH * OW * (C / 4);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(maxpool32_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, N, H, W, C, OH, OW);
    return hipGetLastError();
}

hipError_t launch_group_mean32(const float* in, int groups, int L, int D, float* out, hipStream_t s) {
    if (D % 4) return hipErrorInvalidValue;
    const long total = (long)groups * (D / 4);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(group_mean32_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, groups, L, D, out);
    return hipGetLastError();
}

hipError_t launch_zero_tail32(float* x, const int* valid, int halvings, int B, int H, long row_elems, hipStream_t s) {
    if (B <= 0 || H <= 0 || !valid) return hipSuccess;
    if (row_elems % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(zero_tail32_kernel, dim3((unsigned)(B * H)), dim3(256), 0, s, x, valid, halvings, H, (int)(row_elems / 4));
    return hipGetLastError();
}

hipError_t launch_gemm_x3(const GemmX3Args& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (!a.A || !a.Wh || !a.Wl || !a.out || a.K <= 0 || a.K % 256 || a.N <= 0 || a.N % 128 || (a.lda & 3) || (a.ldw & 7) || (a.ldc & 3) || (a.res && (a.ldr & 3)) ||
        ((uintptr_t)a.A & 15))
        return hipErrorInvalidValue;
    const long tiles = (long)((a.M + 31) / 32) * (a.N / 128);
    hipLaunchKernelGGL(gemm_x3_kernel, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_poison(void* p, size_t bytes, hipStream_t s) {
    if (!p || bytes < 16) return hipSuccess;
    hipLaunchKernelGGL(poison_kernel, dim3(4096), dim3(256), 0, s, reinterpret_cast<uint4*>(p), bytes / 16);
    return hipGetLastError();
}
