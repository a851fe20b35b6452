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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifndef JG_LNF_JD
#define JG_LNF_JD 2            // LN-fused epilogue: 16-row block in front of which the next tile's first DMA is issued (experiments: 1 / 3 / 4)
#endif

JG_NS_BEGIN

// epilogue activation (GemmArgs::relu): 0 none, 1 ReLU, 2 exact GELU 0.5 v (1 + erf(v / sqrt 2)) (XLM-RoBERTa's intermediate layer;
// the conv instances are compiled with ReLU only)
__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
    if (act == 1) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (act == 2) {
        v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678118654752f)); v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678118654752f));
        v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678118654752f)); v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678118654752f));
    }
    return v;
}

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
            tap_decode(a.g, kp, kh, kw);
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
                    acc[i][j] = JG_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
                    if (W2) acc[i][j] = JG_MFMA_16x16x32(wl[i], xf[j], acc[i][j]);
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
            v = act4(v, CONV ? (a.relu != 0) : a.relu);
            if (a.out32) *reinterpret_cast<f32x4*>(a.out32 + (long)m * a.ldc + n) = v;
            if (a.out16) {
                f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                *reinterpret_cast<f16x4*>(a.out16 + (long)m * a.ldc + n) = h;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// LDS-DMA variant for the plain (Linear) GEMMs, K % 64 == 0:  tile 256(m) x 128(n) x 64(k), 8 waves
// (4 x 2, each wave 64 x 64), both operand tiles brought in by global_load_lds_dwordx4 (no VGPR
// round trip, no ds_write), two LDS stages (activations 32 KB + weights 16 KB (+16 KB lo) each).
// The LDS image is lane-linear per wave-instruction (1 KB = 8 rows x 128 B), so the XOR swizzle is
// applied to the per-lane SOURCE address and again on the fragment read (guide rule 21).
// Rows beyond M / N are clamped to the last valid row (their results are never stored).
// Workgroups that share an activation row panel (the N tiles of one M tile) are remapped onto one
// XCD so the panel is fetched into that L2 once.
// Diagnostic build only (-DJG_CLOCK_STAMPS, see conv1.hip): wave 0 of every workgroup stamps both clocks around its persistent loop.
#if defined(JG_CLOCK_STAMPS) && !defined(JG_BF16)
__device__ unsigned long long jg_clock_stamps_gemm[2 * 1024];
extern "C" int jg_clock_read_gemm(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jg_clock_stamps_gemm), sizeof(unsigned long long) * (n < 2048 ? n : 2048)) == hipSuccess ? 0 : 1;
}
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MI = 16-row MFMA tiles per wave along m (4: 64x64 wave tile, 8: 128x64); WM x WN waves.
//   <4,4,2>: 256x128 block tile (hi+lo weights fit two LDS stages);  <8,2,4>: 256x256 block tile for single-fp16
//   weights -- 1.5x fewer L2->LDS bytes per FLOP, which is what bounds the 256x128 kernel once the lo MFMAs are gone.
// LNF: row-wide 128x512 tile with the residual add and the LayerNorm in the epilogue (tiled fp16 + 8-bit token stream).
// SPR: plain GEMM -- the next k-tile's DMA pieces are spread over the MFMA schedule (K <= 1024 instances);
//      CONV -- the loader can read the producer's left-out leading rows from a const image (ConvGeom::in_op / const_in).
// Every CONV instance honours ConvGeom::rowmap on its output side (compacted row index, row_full()).
// XE:  plain GEMM instances that also know the implicit-LayerNorm epilogues (GemmArgs::ln_mode, common.h) -- their own
//      instantiations, so that the epilogue code and its registers never reach the instances of the gesture path.
//      XE = 1: consumer (ln_mode 1), XE = 2: producer (ln_mode 2); each is compiled with that one epilogue only (no residual, no
//      tiled operand, no generic path: launch_gemm checks the shapes), which is what keeps the 256x256 instance free of spills.
// C32: conv instance for C == 32 input channels (JEGAL's second audio conv, cnn.3: 32 -> 64 channels, jegal.py:45-47): a 64-wide k-tile
//      is then TWO taps x 32 channels, so a lane's tap depends on which half of the 128-B row its 16-B chunk sits in (one select per
//      piece); with WN = 1 (N = 64) every wave stages ONE weight piece.
template <bool W2, bool CONV, int MI, int WM, int WN, bool LNF = false, bool SPR = false, int XE = 0, bool C32 = false>
__global__ __launch_bounds__(512) void gemm_glds_kernel(GemmArgs a, int n_tiles, int total_tiles, const f16* zeros, int counted_ok, unsigned long long* tl) {
    static_assert(WM * WN == 8, "8 waves");
    static_assert(!C32 || (CONV && !SPR && !W2), "C32: plain conv instance, single fp16 weights");
    static_assert(!LNF || (WM == 1 && !W2), "fused LayerNorm needs a row-wide tile: all 8 waves side by side along n");
    static_assert(!XE || (!CONV && !LNF), "implicit-LayerNorm epilogues: plain GEMM instances only");
    constexpr int BM = 16 * MI * WM, BN = 64 * WN;
    constexpr int XI = BM / 64, WI = BN / 64;            // LDS-DMA instructions (8 rows each) per wave per k-tile
    constexpr int XB = BM * 128, WB = BN * 128;
    constexpr int STAGE = XB + WB * (W2 ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int G = gridDim.x;
    const int frow = lane & 15, fq = lane >> 4;
    const int fsw = (frow >> 1) & 7;
    const int lrow = lane >> 3, pc = lane & 7;

    // CONV with ConvGeom::rowmap: the kernel runs over the COMPACTED rows m' of the launch -- per image the output rows that
    // depend on the position (common.h) -- and rowmap[m'] gives the full output row (the pixel to compute, the row the epilogue
    // stores to) and the image's count s2 in the top byte.  Mrows replaces a.M.
    int Mrows = a.M;
    const int* rmap = nullptr;
    const float inv_per = CONV ? 1.0f / (float)(a.g.OH * a.g.OW) : 0.f, inv_ow = CONV ? 1.0f / (float)a.g.OW : 0.f;
    if constexpr (CONV) {
        if (a.g.rowmap) {
            rmap = a.g.rowmap;
            Mrows = __builtin_amdgcn_readfirstlane(*a.g.rows_total);
            total_tiles = ((Mrows + BM - 1) / BM) * n_tiles;
        }
    }
    // ConvGeom::in_op / const_in (conv instances with SPR, which has no other meaning for CONV: launch_glds picks them when
    // const_in is set): input rows 0..rin-1 of an image were left out by the producer because they do not depend on the
    // position; the loader reads them from the const image of the input instead (same pixel, other base).  rin =
    // conv_skip_decode(s2[img], in_op) is per row (it rides in the low byte of xpix).
    // Its own instances: the compare + select per LDS-DMA piece cost the 256x256 conv kernel 5 % (conv4 / conv5: 580 -> 610 us)
    // when it was compiled into all of them.
    constexpr bool ROWCONST = CONV && SPR;
    const bool rowconst = ROWCONST && rmap && a.g.const_in;
    auto row_full = [&](int m) -> long {
        if (!CONV || !rmap) return m;
        return rmap[m] & 0xffffff;
    };

    // Persistent: one workgroup per CU walks rounds of G tiles.  Within a round the workgroups of one
    // XCD (b, b+8, ...) take consecutive tile ids, so the N tiles of an activation panel share an L2.
    auto tile_of = [&](int round) -> int {
        const int v0 = round * G;
        if (v0 >= total_tiles) return -1;
        const int cnt = total_tiles - v0 < G ? total_tiles - v0 : G;
        const int b = blockIdx.x;
        if (b >= cnt) return -1;
        const int q = cnt / 8, rr = cnt % 8, xcd = b % 8, loc = b / 8;
        return v0 + (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    };

    // per-lane LDS-DMA source state of the current tile: wave-instruction i covers tile rows
    // (wave*R + i)*8 .. +8, lane -> (row, physical 16-B chunk); the XOR swizzle is applied to the SOURCE.
    // CONV: the activation row m is an output pixel; per k-tile every lane turns its chunk's
    // k = (kh,kw,c) into an NHWC address, or into `zeros` (a zero page) for padding taps and the K
    // tail -- LDS-DMA cannot predicate, but it can read zeros.
    // CONV requires C % 64 == 0 (launch_gemm sends other layers to gemm_kernel): a 64-wide k-tile is then 64 channels
    // of ONE tap, so (kh, kw, c0) are wave-uniform and simply advance with the k-tiles (stage() is always called for
    // kt = 0, 1, 2, ... of a tile); per lane only the pointer to its (pixel, chunk) at tap (0,0) -- possibly outside
    // the image for padded layers, dereferenced only when the tap lands inside -- and the pixel coordinates remain.
    // Register budget (the 128x512 and 512x128 instances sit at the 256-VGPR limit): the pixel coordinates of a conv row and
    // its image's const-row count share one register (ih << 20 | (iw & 0xfff) << 8 | rin), and the WI weight-row pointers are two base pointers (even / odd
    // 8-row piece: the swizzled chunk depends on the piece's parity only) plus a wave-uniform multiple of 16 rows.
    const f16* xsrc[XI];
    const f16* xalt[ROWCONST ? XI : 1];          // the same pixel in the const image of the input (ROWCONST)
    int xpix[XI];
    int tidx = 0, tc0 = 0;
    static_assert(WI % 2 == 0 || WI == 1, "weight pieces come in even/odd pairs (or one piece per wave: N = 64)");
    const f16* whb[2];
    const f16* wlb[2];
    int wchunk[2];
    int n0 = 0, m0 = 0;
    auto setup = [&](int bid) {
        n0 = (bid % n_tiles) * BN;
        m0 = (bid / n_tiles) * BM;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int row = (wave * XI + i) * 8 + lrow;
            const int c = pc ^ ((row >> 1) & 7);
            int m = m0 + row;
            m = m < Mrows ? m : Mrows - 1;
            if (CONV) {
                int s2 = 0;
                if (rmap) {
                    const int e = rmap[m];
                    m = e & 0xffffff;
                    s2 = (int)((unsigned)e >> 24);
                }
                // m < 2^24 (launch_gemm): quotients from float reciprocals with a +-1 fix-up instead of two integer divisions
                // (8 pieces x 2 divisions per tile and lane, and their temporaries on top of 128 live accumulators)
                const int per = a.g.OH * a.g.OW;
                int img = (int)((float)m * inv_per);
                int rem = m - img * per;
                img += rem < 0 ? -1 : (rem >= per ? 1 : 0);
                rem += rem < 0 ? per : (rem >= per ? -per : 0);
                int oh = (int)((float)rem * inv_ow);
                int ow = rem - oh * a.g.OW;
                oh += ow < 0 ? -1 : (ow >= a.g.OW ? 1 : 0);
                ow += ow < 0 ? a.g.OW : (ow >= a.g.OW ? -a.g.OW : 0);
                const int ih = oh * a.g.SH - a.g.PH, iw = ow * a.g.SW - a.g.PW;
                const int rin = rowconst ? conv_skip_decode(s2, a.g.in_op) : 0;
                // C32: the low bit says which of the k-tile's two taps this lane's chunk belongs to (chunks 4-7 = the second tap)
                xpix[i] = (ih << 20) | ((iw & 0xfff) << 8) | (C32 ? (c >> 2) : rin);      // |ih|, |iw| < 2048 (checked by launch_gemm), rin <= 19
                xsrc[i] = a.A + (long)img * a.g.H * a.g.W * a.g.C + ((long)ih * a.g.W + iw) * a.g.C + (C32 ? (c & 3) * 8 : c * 8);
                if constexpr (ROWCONST) xalt[i] = a.g.const_in + ((long)ih * a.g.W + iw) * a.g.C + c * 8;
            } else {
                xpix[i] = 0;
xsrc[i] = (!XE && a.a_tiled) ? a.A + (long)(m >> 7) * 65536 + ((m & 127) >> 4) * 1024 + (m & 15) * 16 + (c >> 1) * 256 + (c & 1) * 8
                                            : a.A + (long)m * a.lda + c * 8;
            }
        }
        // N % BN == 0 (launch_gemm sends anything else to gemm_kernel): no row clamp, piece i = base[i & 1] + (i >> 1) * 16 rows
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int row = (wave * WI + par) * 8 + lrow;
            const int c = pc ^ ((row >> 1) & 7);
            const long off = (long)(n0 + row) * a.ldw + c * 8;
            whb[par] = a.Wh + off;
            wlb[par] = W2 ? a.Wl + off : nullptr;
            wchunk[par] = c;
        }
    };
    // One k-tile = NPIECE LDS-DMA instructions per wave (XI activation pieces, then WI weight pieces, then the lo
    // weights).  stage_begin() fixes the k-tile's wave-uniform part, stage_piece<P>() issues one instruction:
    // inside the k loop the pieces are spread over the MFMA schedule (an LDS-DMA issue costs ~60 cycles between
    // MFMAs but 100-185 in a burst of eight in front of them -- MI355X_MICROARCH.md -- and during that burst the
    // matrix pipe of every SIMD idles, because all waves leave the barrier together).
    constexpr int NPIECE = XI + WI * (W2 ? 2 : 1);
    int sk0 = 0;
    long stapoff = 0;
    int stkh = 0, stkw = 0;
    bool skin = true;
    long stapoff2 = 0;           // C32: the k-tile's second tap
    int stkh2 = 0, stkw2 = 0;
    bool skin2 = false;
    auto stage_begin = [&](int kt) __attribute__((always_inline)) {
        sk0 = kt * 64;
        if constexpr (C32) {
            tap_decode(a.g, 2 * kt, stkh, stkw);
            tap_decode(a.g, 2 * kt + 1, stkh2, stkw2);
            stapoff = ((long)stkh * a.g.W + stkw) * a.g.C;
            stapoff2 = ((long)stkh2 * a.g.W + stkw2) * a.g.C;
            skin = sk0 < a.K;
            skin2 = sk0 + 32 < a.K;
        } else if (CONV) {
            if (kt == 0) tidx = tc0 = 0;
            tap_decode(a.g, tidx, stkh, stkw);
            stapoff = ((long)stkh * a.g.W + stkw) * a.g.C + tc0;
            skin = sk0 < a.K;
            tc0 += 64;
            if (tc0 == a.g.C) {
                tc0 = 0;
                ++tidx;
            }
        }
    };
    auto stage_piece = [&](int p, int buf) __attribute__((always_inline)) {      // p is a compile-time constant at every call site (unrolled loops)
        char* base = smem + buf * STAGE;
        if (p < XI) {
            const int i = p;
            const f16* src;
            if constexpr (C32) {
                const bool hb = (xpix[i] & 1) != 0;
                const int ih = (xpix[i] >> 20) + (hb ? stkh2 : stkh), iw = ((xpix[i] << 12) >> 20) + (hb ? stkw2 : stkw);
                const bool ok = (hb ? skin2 : skin) && (unsigned)ih < (unsigned)a.g.H && (unsigned)iw < (unsigned)a.g.W;
                src = ok ? xsrc[i] + (hb ? stapoff2 : stapoff) : zeros;
            } else if (CONV) {
                const int ih = (xpix[i] >> 20) + stkh, iw = ((xpix[i] << 12) >> 20) + stkw;
                const bool ok = skin && (unsigned)ih < (unsigned)a.g.H && (unsigned)iw < (unsigned)a.g.W;
                src = ok ? xsrc[i] + stapoff : zeros;
                if (ROWCONST && ok && ih < (xpix[i] & 0xff)) src = xalt[i] + stapoff;
            } else {
src = xsrc[i] + ((!XE && a.a_tiled) ? (long)sk0 * 128 : (long)sk0);        // tiled plane: a k-tile is 8192 elements on
            }
            // LNF, long K (linear2: 413 MB of hidden activations read once by one tile each): nontemporal (aux = 2) so
            // the stream does not push the weights out of L2.  Measured: linear2+LN 282 -> 266 us; for out_proj (K = 512,
            // its input was written by the attention kernel just before) the same hint costs 10 %.
            if (LNF && a.K >= 1024) __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(base + (wave * XI + i) * 1024), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(base + (wave * XI + i) * 1024), 16, 0, 0);
        } else if (p < XI + WI) {
            const int i = p - XI;
            const bool kok = !CONV || (sk0 + wchunk[i & 1] * 8 < a.K);
            const f16* wsrc = whb[i & 1] + (long)(i >> 1) * 16 * a.ldw + sk0;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(kok ? wsrc : zeros), (lds_ptr_t)(base + XB + (wave * WI + i) * 1024), 16, 0, 0);
        } else if (W2 && p < NPIECE) {
            const int i = p - XI - WI;
            const bool kok = !CONV || (sk0 + wchunk[i & 1] * 8 < a.K);
            const f16* wsrc = wlb[i & 1] + (long)(i >> 1) * 16 * a.ldw + sk0;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(kok ? wsrc : zeros), (lds_ptr_t)(base + XB + WB + (wave * WI + i) * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
        stage_begin(kt);
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) stage_piece(p, buf);
    };

    const int nk = (a.K + 63) / 64;
    // conv launches carry no residual (launch_gemm rejects them): the conv instances are compiled without that path -- the
    // hoisted reciprocal of its `m % res_mod` alone cost the 512x128 instance a spilled register
    const float* const ares = (CONV || XE) ? nullptr : a.res;

    int round = 0;
    int bid = tile_of(0);
    if (bid < 0) return;
    setup(bid);
    stage(0, 0);
    if (const int stagger = counted_ok >> 8) {
        // De-phase the workgroups: left alone all CUs reach their epilogue together and the store burst of a
        // whole round (G tiles) hits HBM at once while the MFMA pipes wait for it.  The highest block ids get
        // the longest delay -- they are the ones that run one tile fewer in the last round.
        const int ph = (int)(((long)blockIdx.x * 4) / G);
        if (ph) {
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < (unsigned long long)(ph * stagger)) __builtin_amdgcn_s_sleep(16);
        }
    }
    counted_ok &= 1;
    // debug timeline (gemm_timeline option): wave 0 of workgroup 0 / G-1 stamps s_memrealtime at the phase edges
    int tli = 0;
    const bool tl_on = tl != nullptr && wave == 0 && (blockIdx.x == 0 || blockIdx.x == G - 1);
    unsigned long long* tlp = tl + (blockIdx.x == 0 ? 0 : 512);
    auto mark = [&]() {
        if (tl_on && tli < 512) {
            const unsigned long long c = wall_clock64();
            if (lane == 0) tlp[tli] = c;
            ++tli;
        }
    };
    // LNF: the accumulators start from the residual.  The token stream is the tiled fp16 + 8-bit pair of common.h
    // (res_dec / res_enc): its raw bits are loaded one tile ahead (4 x 8 B + 16 B per 16-row block and lane, every
    // access a contiguous 512 B / 1 KB per wave instruction) and expanded to fp32 right before the k loop.
    // The raw bits of a 16-row block (4 x 8 B of fp16, 16 B of corrections = 12 registers) are parked IN the 16
    // accumulator registers of that block, which are dead between the block's stores and the next tile's k loop:
    //   acc[0][j] = {x16 i=0 (2 regs), x16 i=3 (2 regs)},  acc[1][j].xy = x16 i=1,  acc[2][j].xy = x16 i=2,  acc[3][j] = d8.
    // (As separate arrays hipcc spilled every loaded value to scratch right behind its load.)
    f32x4 acc[4][MI];
    auto x16t_off = [&](int tm0) -> long { return (long)(tm0 / BM) * 65536 + (long)wn * 8192 + frow * 16 + fq * 4; };     // + j*1024 + i*256
    auto d8t_off = [&](int tm0) -> long { return (long)(tm0 / BM) * 65536 + (long)wn * 8192 + lane * 16; };               // + j*1024
    auto asf = [](unsigned v) -> float { return __builtin_bit_cast(float, v); };
    auto load_stream = [&](int tm0, int j) __attribute__((always_inline)) {
        const f16* xp = a.res16 + x16t_off(tm0) + j * 1024;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        // read-once stream: nontemporal, so it does not push the weights out of L2
        const u32x2 r0 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(xp)), r1 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(xp + 256));
        const u32x2 r2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(xp + 512)), r3 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(xp + 768));
        // res8 == nullptr (option stream_fp16 = 1, the default since round 5): the token stream is the fp16 plane alone -- in a post-norm transformer the
        // LayerNorm output is rounded to fp16 as the next GEMM's operand anyway, and carrying the residual at that precision costs
        // 1-5 % of the feature error (oracle/precision_families.py) for a third fewer stream bytes and no codec arithmetic
        u32x4 dq = {0u, 0u, 0u, 0u};
        if (a.res8) dq = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.res8 + d8t_off(tm0) + j * 1024));
        acc[0][j] = f32x4{asf(r0.x), asf(r0.y), asf(r3.x), asf(r3.y)};
        acc[1][j] = f32x4{asf(r1.x), asf(r1.y), 0.f, 0.f};
        acc[2][j] = f32x4{asf(r2.x), asf(r2.y), 0.f, 0.f};
        acc[3][j] = f32x4{asf(dq.x), asf(dq.y), asf(dq.z), asf(dq.w)};
    };
    if constexpr (LNF) {
#pragma unroll
        for (int j = 0; j < MI; ++j) load_stream(m0, j);
    }
    int pending = 0;           // the k-tile-0 DMA of this tile is older than exactly `pending` epilogue stores (0: unknown)
#if defined(JG_CLOCK_STAMPS) && !defined(JG_BF16)
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (true) {
        // The first k-tile of this tile was issued BEFORE the previous tile's epilogue stores, so it can be
        // retired with a counted wait that leaves those stores in flight: the store burst (and its HBM
        // latency) overlaps this tile's first MFMAs instead of idling the CU.
        switch (pending) {
            case 8: wait_vmcnt<8>(); break;
            case 16: wait_vmcnt<16>(); break;
            case 24: wait_vmcnt<24>(); break;
            case 32: wait_vmcnt<32>(); break;
            case 40: wait_vmcnt<40>(); break;
            case 48: wait_vmcnt<48>(); break;
            case 56: wait_vmcnt<56>(); break;
            case 60: wait_vmcnt<60>(); break;
            default: wait_vmcnt<0>(); break;
        }
        __builtin_amdgcn_s_barrier();
        mark();     // 0: tile start (first k-tile landed)
        const int cn0 = n0, cm0 = m0;
        // interior: the fast epilogues apply.  A tile that is only partial along M (the last row tile: M = 4800 is 18.75 tiles of
        // 256 rows) takes them too, with its stores masked by row (m_full false): the generic per-element path below costs such a
        // tile ~8 us, which is what a one-round launch of the small-M GEMMs then takes (tools/gemm_tiles.py).
        const bool m_full = cm0 + BM <= Mrows;
        const bool interior = cn0 + BN <= a.N;
        const int nb = cn0 + wn * 64 + fq * 4;
        const int mb = cm0 + wm * (16 * MI) + frow;
        // spreading the DMA pieces over the MFMA schedule, measured: -16 % on the 128x512 LN kernel (K = 2048), -5 % on the
        // conv variants, -3 % on the 256x256 Linear tile at K = 512 (SPR instance: qkv, linear1) but +3 % at K = 2048 there
        constexpr bool SPREAD = LNF || CONV || SPR;
        constexpr bool PREFETCH_RES = MI <= 4;        // 128x64 wave tiles have no registers to spare for it

        f32x4 rs[4][PREFETCH_RES ? MI : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                if (!LNF) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PREFETCH_RES) rs[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

        if constexpr (LNF) {
            // expand the parked token-stream bits of this tile into fp32 accumulators (res_dec4, common.h)
            auto asu = [](float v) -> unsigned { return __builtin_bit_cast(unsigned, v); };
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const f32x4 a0 = acc[0][j], a1 = acc[1][j], a2 = acc[2][j], a3 = acc[3][j];
                const unsigned xw[4][2] = {{asu(a0.x), asu(a0.y)}, {asu(a1.x), asu(a1.y)}, {asu(a2.x), asu(a2.y)}, {asu(a0.z), asu(a0.w)}};
                const unsigned dw[4] = {asu(a3.x), asu(a3.y), asu(a3.z), asu(a3.w)};
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = res_dec4(xw[i][0], xw[i][1], dw[i]);
            }
        }
        // LNF: this tile's bias row(s) / gamma / beta -- one element per thread (BN == 512 threads), loaded under the LAST k-tile's MFMAs
        // and parked in LDS right behind the loop (round 5: the loads used to sit between the loop and the first statistics barrier,
        // ~1 us of L2 latency per tile).  Per-clip bias (GemmArgs::bias_clip): a 128-row tile meets at most two clips when rpc >= 128
        // (launch_gemm checks); rows from csplit on take the second clip's vector.
        float pf_b0 = 0.f, pf_b1 = 0.f, pf_g = 0.f, pf_be = 0.f;
        int csplit = 0x7fffffff;
        for (int kt = 0; kt < nk; ++kt) {
            const bool pre = kt + 1 < nk;
            if constexpr (LNF) {
                if (kt == nk - 1) {
                    const float* b0 = a.bias;
                    const float* b1 = a.bias;
                    if (a.bias_clip) {
                        const int c0 = cm0 / a.rpc;
                        b0 = a.bias_clip + (long)(c0 < a.nclips ? c0 : a.nclips - 1) * BN;
                        b1 = a.bias_clip + (long)(c0 + 1 < a.nclips ? c0 + 1 : a.nclips - 1) * BN;
                        csplit = (c0 + 1) * a.rpc;
                    }
                    pf_b0 = b0 ? b0[t] : 0.f;
                    pf_b1 = b1 ? b1[t] : 0.f;
                    pf_g = a.ln_w[t];
                    pf_be = a.ln_b[t];
                }
            }
            if (pre) stage_begin(kt + 1);
            if (pre && !SPREAD) {
#pragma unroll
                for (int p = 0; p < NPIECE; ++p) stage_piece(p, (kt + 1) & 1);
            }
            if (PREFETCH_RES && kt == nk - 1 && interior && ares) {
                // residual prefetch: issued under the last k-tile's MFMAs, consumed in the epilogue
#pragma unroll
                for (int j = 0; j < (PREFETCH_RES ? MI : 1); ++j) {
                    const int m = mb + j * 16 < Mrows ? mb + j * 16 : Mrows - 1;      // rows past M: any valid row (their results are not stored)
                    const int rr = a.res_mod ? (m % a.res_mod) : m;
                    const float* rp = ares + (long)rr * a.ldr + nb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) rs[i][j] = *reinterpret_cast<const f32x4*>(rp + i * 16);
                }
            }
            const char* sX = smem + (kt & 1) * STAGE;
            const char* sWh = sX + XB;
            const char* sWl = sWh + WB;
            // Fragment schedule pinned by hand: the 4 weight fragments of a k-step, then the activation fragments
            // two ahead of the MFMAs that use them.  Left to itself hipcc hoists all 12 reads of a k-step in
            // front of one s_waitcnt lgkmcnt(0), i.e. the MFMA pipe idles for a full LDS latency four times per
            // k-tile and 48 VGPRs of fragments are live (no room to overlap anything).
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int choff = ((kk * 4 + fq) ^ fsw) << 4;
                auto ldx = [&](int j) -> f16x8 {
                    return *reinterpret_cast<const f16x8*>(sX + (wm * (16 * MI) + j * 16 + frow) * 128 + choff);
                };
                f16x8 wf[4], wl[4], xq[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = wn * 64 + i * 16 + frow;
                    wf[i] = *reinterpret_cast<const f16x8*>(sWh + row * 128 + choff);
                    if (W2) wl[i] = *reinterpret_cast<const f16x8*>(sWl + row * 128 + choff);
                }
                xq[0] = ldx(0);
                xq[1] = ldx(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < MI; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][j] = JG_MFMA_16x16x32(wf[i], xq[j % 3], acc[i][j]);
                        if (W2) acc[i][j] = JG_MFMA_16x16x32(wl[i], xq[j % 3], acc[i][j]);
                    }
                    if (j + 2 < MI) xq[(j + 2) % 3] = ldx(j + 2);
                    if (SPREAD && pre) {
                        // this slot's share of the next k-tile's DMA pieces
                        constexpr int SLOTS = 2 * MI, PER = (NPIECE + SLOTS - 1) / SLOTS;
#pragma unroll
                        for (int u = 0; u < PER; ++u) stage_piece((kk * MI + j) * PER + u, (kt + 1) & 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();     // drains the LDS-DMA of tile kt+1 (vmcnt(0)) and fences the reads of tile kt
        }

        mark();     // 1: k loop done
        // fp16-only outputs (qkv, linear1, the conv layers) leave through the row-transposing epilogue below
        constexpr int TP16 = 144;                        // row pitch: 16-B aligned, 36 banks -> conflict-free b64 writes
        constexpr bool ROWS_OK = STAGE / 8 >= 16 * TP16;
        const bool rows16 = ROWS_OK && interior && a.out16 && !a.out32 && !ares && (a.ldc & 7) == 0;
        // implicit LayerNorm (XE instances; launch_gemm checks the shapes): 1 = consumer, through the rows16 path with per-row
        // (rstd, mean) factors; 2 = producer, the two-plane path below
        const int orow_m = cm0 + wm * (16 * MI) + (lane >> 3);
        // ConvGeom::rowmap: the full output rows of this wave's 16*MI tile rows (both store paths below stay inside them) go
        // through a per-wave table in LDS -- 1-2 loads per lane, in front of the next tile's setup and OLDER than its first DMA,
        // so nothing in the epilogue waits for that DMA.  (Sixteen row indices per lane in registers spilled the 512x128 and
        // the ROWCONST instances, and so did two values kept live across setup().)  The table sits behind
        // the transposing epilogue's scratch in LDS stage 1, which is idle until the next tile's k-tile 1 is staged.
        constexpr int WROWS = 16 * MI;
        // (per-lane table addresses are recomputed from a laundered lane id per tile: hoisted out of the persistent loop as
        // invariants they cost the 512x128 instance four spilled registers)
        int lane_t = lane;
        if constexpr (CONV) asm volatile("" : "+v"(lane_t));
        int* rtab = reinterpret_cast<int*>(smem + STAGE + 8 * (16 * TP16)) + wave * 128;
        const bool use_rtab = CONV && rmap && interior;
        if constexpr (CONV) {
            if (use_rtab) {
                const int wrow0 = cm0 + wm * WROWS;
                // (rows past Mrows of an M-partial tile: clamped -- their stores are masked, but the read must stay inside the map)
                const int i0 = wrow0 + lane_t < Mrows ? wrow0 + lane_t : Mrows - 1, i1 = wrow0 + 64 + lane_t < Mrows ? wrow0 + 64 + lane_t : Mrows - 1;
                if (WROWS >= 64 || lane_t < WROWS) rtab[lane_t] = rmap[i0] & 0xffffff;
                if (WROWS > 64) rtab[64 + lane_t] = rmap[i1] & 0xffffff;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // next tile: issue its first k-tile now, BEFORE the epilogue's stores (LNF: inside its epilogue, see there)
        const int nbid = tile_of(++round);
        if (nbid >= 0) setup(nbid);
        __builtin_amdgcn_sched_barrier(0);
        if (nbid >= 0 && !LNF) stage(0, 0);
        __builtin_amdgcn_sched_barrier(0);       // keep the DMA older than the stores (the counted wait relies on it)
        if (!LNF) mark();     // 2: next tile's first DMA issued (LNF: statistics done, see below)

        if constexpr (LNF) {
            // ---- fused residual + LayerNorm epilogue (gestsync.py:20: LN(x + sublayer(x)), eps 1e-5).
            // The tile spans the whole row (BN == N == 512, wave wn owns columns wn*64..+63): row statistics are a
            // shuffle over the 4 lanes of a row inside the wave plus an 8-way exchange through LDS (stage 1 is idle
            // here).  Two-pass (mean, then centred variance) like nn.LayerNorm.
            // Input and output token stream: the tiled fp16 + 8-bit pair (common.h) in this kernel's own fragment
            // order, normally in place.  As soon as the 16-row block j of this tile has been stored, the raw bits of
            // block j of the NEXT tile are loaded (no spare registers are needed for a whole tile of residual, the
            // HBM latency hides behind the rest of the epilogue); that tile's first DMA goes in front of block JD and
            // is retired with a counted wait.  Whole 128-row tiles are written: the planes hold ceil(M/128)*128 rows.
            // bias / gamma / beta come from LDS (3 x 512 floats, re-staged per tile: the k loop owns the LDS).
            constexpr int JD = JG_LNF_JD;                                 // row block in front of which the next tile's DMA goes (default 2)
            float* red = reinterpret_cast<float*>(smem + STAGE);          // [2][8 waves][BM rows]
            float* lnp = red + 2 * 8 * BM;                                // [4][BN]: bias, gamma, beta, bias of the tile's second clip
            static_assert(BN == 512, "one parked element per thread");
            lnp[t] = pf_b0;
            lnp[BN + t] = pf_g;
            lnp[2 * BN + t] = pf_be;
            lnp[3 * BN + t] = pf_b1;
            __syncthreads();
            const int ncol = wn * 64 + fq * 4;                            // + i*16: this lane's 4 columns of block i
            float rsum[MI];
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                float sj = 0.f;
                const float* bsel = lnp + (cm0 + j * 16 + frow >= csplit ? 3 * BN : 0) + ncol;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = acc[i][j] + *reinterpret_cast<const f32x4*>(bsel + i * 16);
                    acc[i][j] = v;
                    sj += (v.x + v.y) + (v.z + v.w);
                }
                sj += __shfl_xor(sj, 16, 64);
                sj += __shfl_xor(sj, 32, 64);
                rsum[j] = sj;
            }
            if (fq == 0) {
#pragma unroll
                for (int j = 0; j < MI; ++j) red[wn * BM + j * 16 + frow] = rsum[j];
            }
            __syncthreads();
            float rsq[MI];
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                float tsum = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) tsum += red[w * BM + j * 16 + frow];
                const float mean = tsum * (1.f / BN);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 d = acc[i][j] - mean;
                    acc[i][j] = d;
                    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                rsq[j] = q;
            }
            if (fq == 0) {
#pragma unroll
                for (int j = 0; j < MI; ++j) red[(8 + wn) * BM + j * 16 + frow] = rsq[j];
            }
            __syncthreads();
            mark();     // 2 (LNF): row statistics done, the store / reload phase starts
            f16* o16 = a.out16 + x16t_off(cm0);
            signed char* o8 = a.out8 ? a.out8 + d8t_off(cm0) : nullptr;
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                if (j == JD) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (nbid >= 0) stage(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float tq = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) tq += red[(8 + w) * BM + j * 16 + frow];
                const float inv = a.ln_flavour == LN_STD ? 1.f / sqrtf(tq * (1.f / BN) + 1e-5f)
                                                         : 1.f / (sqrtf(tq * (1.f / (BN - 1))) + 1e-6f);
                uint4 dq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 y = acc[i][j] * inv * *reinterpret_cast<const f32x4*>(lnp + BN + ncol + i * 16) +
                                    *reinterpret_cast<const f32x4*>(lnp + 2 * BN + ncol + i * 16);
                    const f16x4 hv = {(f16)y.x, (f16)y.y, (f16)y.z, (f16)y.w};
                    *reinterpret_cast<f16x4*>(o16 + j * 1024 + i * 256) = hv;
                    if (o8) {
                        const unsigned dw = res_enc4(y.x, y.y, y.z, y.w, hv[0], hv[1], hv[2], hv[3]);
                        if (i == 0) dq.x = dw; else if (i == 1) dq.y = dw; else if (i == 2) dq.z = dw; else dq.w = dw;
                    }
                }
                if (o8) *reinterpret_cast<uint4*>(o8 + j * 1024) = dq;
                if (nbid >= 0) load_stream(m0, j);                        // m0 is already the next tile's
            }
            mark();     // 3: epilogue issued
            if (nbid < 0) break;
            // the loop-top barrier fences red[] / lnp[] against the next tile's k-tile 1 (staged after it)
            pending = counted_ok ? (MI - JD) * (o8 ? 10 : 8) : 0;
            continue;
        }
        // ---- epilogue: lane holds D[n = 4*fq + r][m = frow] of each 16x16 tile.
        // Interior tiles take a branch-free path: ALL residual / scale / bias loads are issued first, then
        // the math, then all stores.  (Per-element `if (m < M) load` made hipcc branch around every load
        // and wait vmcnt(0) after each one: 16 serial HBM round trips per tile, 1/3 of the Linear time.)
        // fp16-only outputs (qkv, linear1, the conv layers): transpose the tile through LDS so that every store
        // instruction writes 8 rows x 128 contiguous bytes.  In the MFMA fragment layout a store instruction covers
        // 16 rows x 32 B, and the CU's write path retires those at ~17 GB/s: 8-9 us per 256x256 tile, as long as the
        // whole k loop at K = 512 (tools/store_pattern.hip, tools/gemm_timeline.py).  The scratch is this wave's
        // slice of LDS stage 1, idle until the next tile's k-tile 1 is staged (after the barrier at the loop top);
        // a wave's LDS instructions execute in order, so consecutive 16-row blocks reuse the same 2.3 KB.
        constexpr bool xdone = XE == 2;
        if constexpr (XE == 2) {
            {
                // ---- implicit-LayerNorm producer: v = acc + bias + gamma * rstd[m] * (x_prev - mean[m]); v leaves as two fp16 planes
                // (hi = fp16(v), the next GEMM's A operand; lo = fp16(v - hi)) through the row-transposing scratch, one plane after
                // the other, plus the (sum, sum of squares) of this wave's 64 columns per row.  The residual planes are read in
                // fragment order (4 x 8 B per plane and 16-row block: one 128-B line per row over the four i), one block ahead.
                // registers (the 256x256 instance holds 128 accumulators): the bias goes into the accumulators up front, the row
                // statistics ride with the residual prefetch, and where LDS stage 1 has room for two scratch areas per wave the lo plane
                // is written to its own area next to the hi plane instead of waiting in registers
                constexpr bool LO_LDS = STAGE / 8 >= 32 * TP16;
                char* tsc = smem + STAGE + wave * ((LO_LDS ? 32 : 16) * TP16);
                char* tsl = LO_LDS ? tsc + 16 * TP16 : tsc;
                // gamma of this wave's 64 columns waits in LDS (256 B behind the transposing scratch; one ds_read_b128 per use)
                float* gsc = reinterpret_cast<float*>(smem + STAGE + 8 * ((LO_LDS ? 32 : 16) * TP16)) + wave * 64;
                static_assert(STAGE >= 8 * ((LO_LDS ? 32 : 16) * TP16) + 8 * 256, "LDS stage 1 holds the epilogue scratch");
                gsc[lane] = a.scale[cn0 + wn * 64 + lane];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 bi = *reinterpret_cast<const f32x4*>(a.bias + nb + i * 16);
#pragma unroll
                    for (int j = 0; j < MI; ++j) acc[i][j] += bi;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f16x4 rh[2][4], rl[2][4];
                f32x2_t rst[2];
                auto ldres = [&](int j) __attribute__((always_inline)) {
                    const int m = mb + j * 16 < Mrows ? mb + j * 16 : Mrows - 1;
                    const long o = (long)m * a.ldc + nb;
                    rst[j & 1] = *reinterpret_cast<const f32x2_t*>(a.ln_stats + 2 * (long)m);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        rh[j & 1][i] = *reinterpret_cast<const f16x4*>(a.xres_hi + o + i * 16);
                        rl[j & 1][i] = *reinterpret_cast<const f16x4*>(a.xres_lo + o + i * 16);
                    }
                };
                ldres(0);
                const long ocol = cn0 + wn * 64 + (lane & 7) * 8;
                const int sblk = (cn0 >> 6) + wn, nblk = a.N >> 6;
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    if (j + 1 < MI) ldres(j + 1);
                    float s1 = 0.f, s2 = 0.f;
                    f16x4 lo4[LO_LDS ? 1 : 4];
                    const float mu = rst[j & 1].x, rs = rst[j & 1].y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f16x4 xh = rh[j & 1][i], xl = rl[j & 1][i];
                        const f32x4 x = {(float)xh.x + (float)xl.x, (float)xh.y + (float)xl.y, (float)xh.z + (float)xl.z, (float)xh.w + (float)xl.w};
                        const f32x4 v = acc[i][j] + *reinterpret_cast<const f32x4*>(gsc + i * 16 + fq * 4) * ((x - mu) * rs);
                        s1 += (v.x + v.y) + (v.z + v.w);
                        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                        const f16x4 hv = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                        const f16x4 lv = {(f16)(v.x - (float)hv.x), (f16)(v.y - (float)hv.y), (f16)(v.z - (float)hv.z), (f16)(v.w - (float)hv.w)};
                        *reinterpret_cast<f16x4*>(tsc + frow * TP16 + i * 32 + fq * 8) = hv;
                        if (LO_LDS) *reinterpret_cast<f16x4*>(tsl + frow * TP16 + i * 32 + fq * 8) = lv;
                        else lo4[i] = lv;
                    }
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        if (pl == 1 && LO_LDS) break;
                        if (pl == 1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) *reinterpret_cast<f16x4*>(tsc + frow * TP16 + i * 32 + fq * 8) = lo4[LO_LDS ? 0 : i];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int q2 = 0; q2 < (LO_LDS ? 2 : 1); ++q2) {
                            f16* const plane = (LO_LDS ? q2 : pl) ? a.out_lo : a.out16;
                            const char* src = (LO_LDS && q2) ? tsl : tsc;
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const f16x8 o = *reinterpret_cast<const f16x8*>(src + (h2 * 8 + (lane >> 3)) * TP16 + (lane & 7) * 16);
                                const long orow = orow_m + j * 16 + h2 * 8;
                                if (m_full || orow < Mrows) *reinterpret_cast<f16x8*>(plane + orow * a.ldc + ocol) = o;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                    s1 += __shfl_xor(s1, 16, 64);
                    s2 += __shfl_xor(s2, 16, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (fq == 0 && (m_full || mb + j * 16 < Mrows))
                        *reinterpret_cast<f32x2_t*>(a.stat_out + 2 * ((long)(mb + j * 16) * nblk + sblk)) = f32x2_t{s1, s2};
                }
            }
        }
        if constexpr (xdone) {
        } else if (rows16 || XE == 1) {
            char* tsc = smem + STAGE + wave * (16 * TP16);
            f32x4 sc[4], bi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = f32x4{1.f, 1.f, 1.f, 1.f};
                bi[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (XE != 1 && a.scale) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[i] = *reinterpret_cast<const f32x4*>(a.scale + nb + i * 16);
            }
            // per-clip bias (GemmArgs::bias_clip; plain instances): a tile inside one clip just takes that clip's vector; a tile that
            // meets a clip boundary (one in twelve at 3150 rows per clip) picks the vector per 16-row block and lane
            const float* bias_t = a.bias;
            bool straddle = false;
            if constexpr (!CONV && XE == 0) {
                if (a.bias_clip) {
                    const int c0 = cm0 / a.rpc;
                    straddle = (cm0 + BM - 1) / a.rpc != c0;
                    bias_t = a.bias_clip + (long)(c0 < a.nclips ? c0 : a.nclips - 1) * a.N;
                }
            }
            if (bias_t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bi[i] = *reinterpret_cast<const f32x4*>(bias_t + nb + i * 16);
            }
            f16* ocol = a.out16 + cn0 + wn * 64 + (lane & 7) * 8;
            // ln_mode 1: out = rstd[m] * (acc - mean[m] * c1[n]) + bias[n], c1 (column sums of the folded weights) in `sc`
            float xrs[XE ? MI : 1], xrm[XE ? MI : 1];
            // (registers: c1 of this wave's 64 columns waits in LDS behind the transposing scratch instead of in `sc`)
            float* gsc = reinterpret_cast<float*>(smem + STAGE + 8 * (16 * TP16)) + wave * 64;
            if constexpr (XE == 1) {
                static_assert(!XE || STAGE >= 8 * (16 * TP16) + 8 * 256, "LDS stage 1 holds the epilogue scratch");
                gsc[lane] = a.scale[cn0 + wn * 64 + lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                {
#pragma unroll
                    for (int j = 0; j < MI; ++j) {
                        const int m = mb + j * 16 < Mrows ? mb + j * 16 : Mrows - 1;
                        const f32x2_t st = *reinterpret_cast<const f32x2_t*>(a.ln_stats + 2 * (long)m);
                        xrs[j] = st.y;
                        xrm[j] = st.x * st.y;
                    }
                }
            }
            auto store_rows = [&](auto masked_tag, auto straddle_tag) __attribute__((always_inline)) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                constexpr bool STRADDLE = decltype(straddle_tag)::value;
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    const float* bj = nullptr;
                    if constexpr (STRADDLE) {
                        const int cj = (mb + j * 16) / a.rpc;
                        bj = a.bias_clip + (long)(cj < a.nclips ? cj : a.nclips - 1) * a.N + nb;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 v;
                        if constexpr (XE == 1) v = acc[i][j] * xrs[XE ? j : 0] + (bi[i] - *reinterpret_cast<const f32x4*>(gsc + i * 16 + fq * 4) * xrm[XE ? j : 0]);
                        else if constexpr (STRADDLE) v = acc[i][j] * sc[i] + *reinterpret_cast<const f32x4*>(bj + i * 16);
                        else v = acc[i][j] * sc[i] + bi[i];
                        v = act4(v, CONV ? (a.relu != 0) : a.relu);
                        f16x4 hv = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                        *reinterpret_cast<f16x4*>(tsc + frow * TP16 + i * 32 + fq * 8) = hv;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const f16x8 o = *reinterpret_cast<const f16x8*>(tsc + (h2 * 8 + (lane >> 3)) * TP16 + (lane & 7) * 16);
                        const long orow = use_rtab ? rtab[j * 16 + h2 * 8 + (lane_t >> 3)] : orow_m + j * 16 + h2 * 8;
                        if (!MASKED || orow_m + j * 16 + h2 * 8 < Mrows) __builtin_nontemporal_store(o, reinterpret_cast<f16x8*>(ocol + orow * a.ldc));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            };
            if constexpr (!CONV && XE == 0) {
                if (straddle) {
                    if (m_full) store_rows(std::false_type{}, std::true_type{}); else store_rows(std::true_type{}, std::true_type{});
                } else if (m_full) store_rows(std::false_type{}, std::false_type{}); else store_rows(std::true_type{}, std::false_type{});
            } else {
                if (m_full) store_rows(std::false_type{}, std::false_type{}); else store_rows(std::true_type{}, std::false_type{});
            }
        } else if constexpr (XE != 0) {
        } else if (interior) {
            f32x4 sc[4], bi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = f32x4{1.f, 1.f, 1.f, 1.f};
                bi[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (a.scale) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[i] = *reinterpret_cast<const f32x4*>(a.scale + nb + i * 16);
            }
            if (a.bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bi[i] = *reinterpret_cast<const f32x4*>(a.bias + nb + i * 16);
            }
            auto store_frag = [&](auto masked_tag) __attribute__((always_inline)) {
                constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    const long mo = (use_rtab ? (long)rtab[j * 16 + (lane_t & 15)] : (long)(mb + j * 16)) * a.ldc + nb;
                    f32x4 rj[4];
                    if (!PREFETCH_RES) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rj[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (ares) {
                            const int m = !MASKED || mb + j * 16 < Mrows ? mb + j * 16 : Mrows - 1;
                            const int rr = a.res_mod ? (m % a.res_mod) : m;
                            const float* rp = ares + (long)rr * a.ldr + nb;
#pragma unroll
                            for (int i = 0; i < 4; ++i) rj[i] = *reinterpret_cast<const f32x4*>(rp + i * 16);
                        }
                    }
                    if (MASKED && mb + j * 16 >= Mrows) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 v = acc[i][j] * sc[i] + bi[i] + (PREFETCH_RES ? rs[i][j] : rj[i]);
                        v = act4(v, CONV ? (a.relu != 0) : a.relu);
                        if (a.out32) *reinterpret_cast<f32x4*>(a.out32 + mo + i * 16) = v;
                        if (a.out16) {
                            f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                            *reinterpret_cast<f16x4*>(a.out16 + mo + i * 16) = h;
                        }
                    }
                }
            };
            if (m_full) store_frag(std::false_type{}); else store_frag(std::true_type{});
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = cn0 + wn * 64 + i * 16 + fq * 4;
                if (n >= a.N) continue;
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
                if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
                if (a.bias) bi = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    const int m = cm0 + wm * (16 * MI) + j * 16 + frow;
                    if (m >= Mrows) continue;
                    const long mf = row_full(m);
                    f32x4 v = acc[i][j] * sc + bi;
                    if (ares) {
                        const int rr = a.res_mod ? (m % a.res_mod) : m;
                        v += *reinterpret_cast<const f32x4*>(ares + (long)rr * a.ldr + n);
                    }
                    v = act4(v, CONV ? (a.relu != 0) : a.relu);
                    if (a.out32) *reinterpret_cast<f32x4*>(a.out32 + mf * a.ldc + n) = v;
                    if (a.out16) {
                        f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                        *reinterpret_cast<f16x4*>(a.out16 + mf * a.ldc + n) = h;
                    }
                }
            }
        }
        mark();     // 3: epilogue issued
        if (nbid < 0) break;
        pending = !(interior && m_full && counted_ok) || xdone ? 0 : rows16 ? 2 * MI : (a.out32 ? 4 * MI : 0) + (a.out16 ? 4 * MI : 0);
    }
#if defined(JG_CLOCK_STAMPS) && !defined(JG_BF16)
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) {
        jg_clock_stamps_gemm[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - ck0;
        jg_clock_stamps_gemm[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

// ---- host side.  Every tuning switch and per-device resource lives in the caller's EngineOpts (one per jg_handle):
// nothing here is process-global except the "dynamic LDS attribute set" flags, which are per (kernel, device).
constexpr int MAX_DEV = 64;

void engine_opts_set_timeline(EngineOpts& o, bool on) {
    if (on && !o.gemm_tl) {
        if (hipHostMalloc(&o.gemm_tl, 1024 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) o.gemm_tl = nullptr;
    } else if (!on && o.gemm_tl) {
        (void)hipHostFree(o.gemm_tl);
        o.gemm_tl = nullptr;
    }
}
static void dump_timeline(const EngineOpts& o, hipStream_t s, const char* what) {
    (void)hipStreamSynchronize(s);
    for (int b = 0; b < 2; ++b) {
        const unsigned long long* t = o.gemm_tl + b * 512;
        std::fprintf(stderr, "[gemm timeline %s wg %s] (100 MHz ticks -> us)\n", what, b ? "last" : "0");
        for (int i = 0; i + 3 < 512 && t[i + 3]; i += 4) {
            std::fprintf(stderr, "  tile %2d: kloop %6.2f  setup+dma-issue %5.2f  epilogue %5.2f  wait-next %5.2f\n", i / 4, (t[i + 1] - t[i]) * 0.01,
                         (t[i + 2] - t[i + 1]) * 0.01, (t[i + 3] - t[i + 2]) * 0.01, t[i + 4] ? (t[i + 4] - t[i + 3]) * 0.01 : 0.0);
        }
    }
}

hipError_t engine_opts_init(EngineOpts& o, int device) {
    o.device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    o.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    f16* z = nullptr;
    e = hipMalloc(&z, 256);                  // zero page on THIS device: LDS-DMA cannot predicate, but it can read zeros
    if (e != hipSuccess) return e;
    e = hipMemset(z, 0, 256);
    if (e != hipSuccess) return e;
    o.zeros = z;
    return hipSuccess;
}

void engine_opts_release(EngineOpts& o) {
    if (o.zeros) (void)hipFree(const_cast<f16*>(o.zeros));
    o.zeros = nullptr;
    engine_opts_set_timeline(o, false);
}

template <class K>
static hipError_t ensure_lds_attr(K kernel, size_t lds, int device, bool* flags) {
    if (device < 0 || device >= MAX_DEV) return hipErrorInvalidDevice;
    if (!flags[device]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        flags[device] = true;
    }
    return hipSuccess;
}

template <bool W2, bool CONV, int MI, int WM, int WN, bool SPR = false, int XE = 0, bool C32 = false>
static hipError_t launch_glds_cfg(const GemmArgs& a, const EngineOpts& o, hipStream_t s) {
    static bool attr_set[MAX_DEV] = {};
    constexpr int BM = 16 * MI * WM, BN = 64 * WN;
    constexpr size_t lds = 2 * (size_t)(BM * 128 + BN * 128 * (W2 ? 2 : 1));
    hipError_t e = ensure_lds_attr(gemm_glds_kernel<W2, CONV, MI, WM, WN, false, SPR, XE, C32>, lds, o.device, attr_set);
    if (e != hipSuccess) return e;
    if (!o.zeros) return hipErrorInvalidValue;
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const int tiles = mt * nt;
    const int grid = o.gemm_persistent ? (tiles < o.num_cu ? tiles : o.num_cu) : tiles;
    if (o.gemm_tl) {
        (void)hipStreamSynchronize(s);
        std::memset(o.gemm_tl, 0, 1024 * sizeof(unsigned long long));
    }
    // De-phasing the workgroups (4 phases, 2 us apart; the last phase falls on the workgroups that run one tile fewer):
    // once the outputs were nontemporal the epilogues became HBM-write-burst bound (every CU stores its 128 KB at the
    // same moment) and spreading them pays: qkv 157 -> 146 us.  Only for long plain GEMMs (>= 4 rounds); the LN-fused
    // and conv kernels and short launches measured neutral or slower.  Option gemm_stagger: ticks of 10 ns, -1 = off.
    // Round 3: the delay only pays when the last round is less than half full - the highest block ids, delayed longest, then run one
    // tile fewer (ff0, 3.08 rounds: 79 -> 75 us; N = 1024, 6.16 rounds: 129 -> 117 us; qkv 9.23: 175 -> 170; with a nearly full last
    // round it costs 2-3 %), short launches included (round 2: >= 4 rounds).  Inside the two-lane batches the plain GEMMs stay
    // un-staggered: the other lane's kernels already spread the store bursts and the delay only costs (+0.8 % per step measured).
    const int last_round = tiles % o.num_cu;
    const int stagger = o.gemm_stagger < 0 ? 0 : o.gemm_stagger > 0 ? o.gemm_stagger
                        : (!CONV && !o.lanes_active && tiles > o.num_cu && 2 * last_round < o.num_cu ? 300 : 0);
    hipLaunchKernelGGL((gemm_glds_kernel<W2, CONV, MI, WM, WN, false, SPR, XE, C32>), dim3((unsigned)grid), dim3(512), lds, s, a, nt, tiles, o.zeros,
                       o.gemm_counted | (stagger << 8), o.gemm_tl);
    if (o.gemm_tl) dump_timeline(o, s, CONV ? "conv" : "linear");
    return hipGetLastError();
}

template <bool CONV>
static hipError_t launch_glds_ln(const GemmArgs& a, const EngineOpts& o, hipStream_t s) {
    static bool attr_set[MAX_DEV] = {};
    constexpr size_t lds = 2 * (size_t)(128 * 128 + 512 * 128);          // 160 KiB: the whole LDS
    hipError_t e = ensure_lds_attr(gemm_glds_kernel<false, CONV, 8, 1, 8, true>, lds, o.device, attr_set);
    if (e != hipSuccess) return e;
    if (!o.zeros) return hipErrorInvalidValue;
    const int tiles = (a.M + 127) / 128;
    const int grid = tiles < o.num_cu ? tiles : o.num_cu;
    if (o.gemm_tl) {
        (void)hipStreamSynchronize(s);
        std::memset(o.gemm_tl, 0, 1024 * sizeof(unsigned long long));
    }
    // De-phasing (see launch_glds_cfg), round 3: in this kernel every workgroup reaches its store / reload phase at the same moment
    // and that phase is an HBM burst (100 MB per round in ~10 us) while the k loops leave HBM idle.  Four start phases spread it;
    // the delay is free when the last round is less than half full, because the highest block ids - the ones delayed longest -
    // run one tile fewer: out_proj 108 -> 94 us, linear2 265 -> 256 us at M = 100 800 (3.08 rounds), 52 -> 48 us at 1.15 rounds;
    // with a nearly full last round it costs what it delays (1.92 rounds: 58 -> 61 us), so it is off there.
    const int last_round = tiles % o.num_cu;
    const int auto_stagger = tiles > o.num_cu && 2 * last_round < o.num_cu ? (a.K <= 1024 ? 500 : 1400) : 0;
    const int stagger = o.gemm_stagger < 0 ? 0 : o.gemm_stagger > 0 ? o.gemm_stagger : auto_stagger;
    hipLaunchKernelGGL((gemm_glds_kernel<false, CONV, 8, 1, 8, true>), dim3((unsigned)grid), dim3(512), lds, s, a, 1, tiles, o.zeros,
                       o.gemm_counted | (stagger << 8), o.gemm_tl);
    if (o.gemm_tl) dump_timeline(o, s, "linear+LN");
    return hipGetLastError();
}

bool gemm_ln_fusable(const GemmArgs& a) {
    return a.Wl == nullptr && a.N == 512 && a.K % 64 == 0 && a.M >= 1024 && a.lda % 8 == 0 && a.ldw % 8 == 0 && !a.relu && !a.scale &&
           !a.res && !a.out32 && a.res16 && a.out16 && (a.res8 != nullptr) == (a.out8 != nullptr) && !a.a_tiled;
}

template <bool W2, bool CONV>
static hipError_t launch_glds(const GemmArgs& a, const EngineOpts& o, hipStream_t s) {
    // small problems (the JEGAL branch: M = B*T = 4800 tokens) would leave most CUs idle with 256-row tiles:
    // 128x128 tiles (32x64 wave tiles) give 4x the workgroups
    const long tiles256 = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    const long tiles_big = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);        // 256x256 tiles: fewer than CUs -> under-filled
    if constexpr (CONV) {
        if (a.g.rowmap && a.g.const_in) {   // consumer of a row-skipping producer: the instances whose loader can read the const image
            if (o.gemm_small_tile && (tiles256 < 200 || tiles_big < 224)) return launch_glds_cfg<W2, true, 2, 4, 2, true>(a, o, s);
            if constexpr (!W2) {
                if (o.gemm_big_tile && a.N >= 256 && a.N % 256 == 0) return launch_glds_cfg<false, true, 8, 2, 4, true>(a, o, s);
            }
            return launch_glds_cfg<W2, true, 4, 4, 2, true>(a, o, s);
        }
    }
    if constexpr (!CONV) {
        // Plain GEMMs: pick the tile by a cost estimate, rounds of one tile per CU x (k-tiles x time per k-tile + epilogue), with
        // the per-tile figures measured on the box (tools/gemm_tiles.py; us, L2-resident operands): bigger tiles move fewer
        // bytes per FLOP through the CU's LDS-DMA path but fill fewer CUs / leave emptier last rounds.  Every instance
        // accumulates k in the same order, so the choice never changes a bit of the result.  (Round 2 took the 128x128 tile
        // whenever fewer than 224 of the 256x256 tiles existed: M = 4800 x N = 2048 ran three rounds of small tiles instead of
        // one round of 152 big ones, and XLM-R's N = 768 layers likewise.)
        const int nk = (a.K + 63) / 64;
        auto est = [&](int bm, int bn, double kt_us, double epi_us) {
            const long tiles = (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn);
            return (double)((tiles + o.num_cu - 1) / o.num_cu) * (nk * kt_us + epi_us);
        };
        if (a.ln_mode) {
            // implicit-LayerNorm epilogues (XE instances): single fp16 weights -> 256x256 or 128x128, hi+lo -> 256x128 or 128x128, by the
            // same cost estimate
            auto go = [&](auto xe) -> hipError_t {
                constexpr int X = decltype(xe)::value;
                if constexpr (W2) {
                    if (est(128, 128, 1.25, 1.0) < est(256, 128, 1.6, 1.6) && o.gemm_tile != 2) return launch_glds_cfg<true, false, 2, 4, 2, false, X>(a, o, s);
                    return launch_glds_cfg<true, false, 4, 4, 2, false, X>(a, o, s);
                } else {
                    const bool big_ok = a.N % 256 == 0;
                    if (!big_ok || (est(128, 128, 0.95, 1.0) < est(256, 256, 1.45, 2.7) && o.gemm_tile != 3) || o.gemm_tile == 1)
                        return launch_glds_cfg<false, false, 2, 4, 2, false, X>(a, o, s);
                    if (a.K <= 1024) return launch_glds_cfg<false, false, 8, 2, 4, true, X>(a, o, s);
                    return launch_glds_cfg<false, false, 8, 2, 4, false, X>(a, o, s);
                }
            };
            return a.ln_mode == 1 ? go(std::integral_constant<int, 1>{}) : go(std::integral_constant<int, 2>{});
        }
        const bool can_big = !W2 && o.gemm_big_tile && a.N >= 256 && a.N % 256 == 0;
        const double e_small = o.gemm_small_tile ? est(128, 128, W2 ? 1.25 : 0.95, 1.0) : 1e30;
        const double e_mid = est(256, 128, W2 ? 1.6 : 1.2, 1.6);
        const double e_big = can_big ? est(256, 256, 1.45, 2.7) : 1e30;
        int pick = e_big <= e_mid && e_big <= e_small ? 3 : (e_mid <= e_small ? 2 : 1);
        if (o.gemm_tile >= 1 && o.gemm_tile <= 3 && (o.gemm_tile != 3 || can_big)) pick = o.gemm_tile;
        if (pick == 1) return launch_glds_cfg<W2, false, 2, 4, 2>(a, o, s);
        if constexpr (!W2) {
            if (pick == 3) {
                if (a.K <= 1024) return launch_glds_cfg<false, false, 8, 2, 4, true>(a, o, s);       // short k loops: spread DMA issue
                return launch_glds_cfg<false, false, 8, 2, 4>(a, o, s);
            }
        }
        return launch_glds_cfg<W2, false, 4, 4, 2>(a, o, s);
    }
    if (o.gemm_small_tile && (tiles256 < 200 || tiles_big < 224)) return launch_glds_cfg<W2, CONV, 2, 4, 2>(a, o, s);
    if constexpr (!W2) {
        if (o.gemm_big_tile && a.N >= 256 && a.N % 256 == 0) {
            if (!CONV && a.K <= 1024) return launch_glds_cfg<false, CONV, 8, 2, 4, !CONV>(a, o, s);       // short k loops: spread DMA issue
            return launch_glds_cfg<false, CONV, 8, 2, 4>(a, o, s);
        }
        // N = 128 (conv2): 512x128 block tile, the whole 160 KiB of LDS -- the activation side dominates the
        // L2->LDS traffic there, a taller tile halves the weight re-reads per activation byte
        if constexpr (CONV) {
            if (o.gemm_tall_tile && a.N == 128 && a.M >= 512 * 256) return launch_glds_cfg<false, true, 8, 4, 2>(a, o, s);
        }
    }
    return launch_glds_cfg<W2, CONV, 4, 4, 2>(a, o, s);
}

template <int WM, int WN, bool CONV, bool W2>
static hipError_t launch_variant(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const long mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)(BM + BN * (W2 ? 2 : 1)) * 128;
    hipLaunchKernelGGL((gemm_kernel<WM, WN, CONV, W2>), dim3((unsigned)(mt * nt)), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, bool conv, const EngineOpts& o, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (a.ln_w) {
        if (conv || !gemm_ln_fusable(a) || (a.bias_clip && (a.rpc < 128 || a.nclips <= 0))) return hipErrorInvalidValue;
        return launch_glds_ln<false>(a, o, s);
    }
    const bool w2 = a.Wl != nullptr;
    const bool narrow = a.N <= 64;
    if (a.bias_clip) {
        // per-clip bias: only the LDS-DMA kernel's fp16 row epilogue knows it (the LayerNorm-fused launch returned above); rpc >= 256
        // keeps a tile within two clips' reach of the straddle path's per-block lookup
        const bool ok = !conv && !a.ln_mode && o.gemm_glds && a.rpc >= 256 && a.nclips > 0 && a.K % 64 == 0 && a.M >= 128 && a.lda % 8 == 0 && a.ldw % 8 == 0 &&
                        a.N % 128 == 0 && a.out16 && !a.out32 && !a.res && (a.ldc & 7) == 0 && !narrow;
        if (!ok) return hipErrorInvalidValue;
    }
    if (a.ln_mode) {
        // implicit LayerNorm: LDS-DMA instances with the fast epilogues only (whole tiles along n, 16-byte rows); anything else is a
        // caller error -- there is no slow path that would quietly ignore the statistics
        const bool shape_ok = !conv && o.gemm_glds && a.K % 64 == 0 && a.M >= 128 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.N % 128 == 0 && a.ldc % 8 == 0;
        const bool mode1_ok = a.ln_mode == 1 && a.ln_stats && a.scale && a.bias && a.out16 && !a.out32 && !a.res;
        const bool mode2_ok = a.ln_mode == 2 && a.ln_stats && a.scale && a.bias && a.out16 && a.out_lo && a.xres_hi && a.xres_lo && a.stat_out &&
                              !a.out32 && !a.res && !a.relu;
        if (!shape_ok || !(mode1_ok || mode2_ok) || a.ln_w || a.a_tiled) return hipErrorInvalidValue;
        return w2 ? launch_glds<true, false>(a, o, s) : launch_glds<false, false>(a, o, s);
    }
    if (a.a_tiled && (conv || narrow || !o.gemm_glds || a.K != 512 || a.M < 128 || a.N % 128)) return hipErrorInvalidValue;   // LDS-DMA kernel only
    if (conv) {
        if (a.res) return hipErrorInvalidValue;          // the conv instances are compiled without the residual path
        const bool coords_ok = a.g.H + a.g.PH < 2048 && a.g.W + a.g.PW < 2048 && a.M < (1 << 24);      // packed pixel coordinates / rowmap entries
        // C = 32 -> N = 64 (the second audio conv): its own LDS-DMA instance, 256x64 tiles (round 5; before: the register-staged kernel)
        if (o.gemm_glds && !w2 && a.N == 64 && a.g.C == 32 && a.K % 32 == 0 && a.K == a.g.KH * a.g.KW * 32 && !a.g.tap_table && !a.g.rowmap && a.M >= 256 &&
            coords_ok && a.out16 && !a.out32 && (a.ldc & 7) == 0 && (a.ldw & 7) == 0)
            return launch_glds_cfg<false, true, 2, 8, 1, false, 0, true>(a, o, s);
        if (narrow) return w2 ? launch_variant<4, 1, true, true>(a, s) : launch_variant<4, 1, true, false>(a, s);
        if (o.gemm_glds && a.M >= 256 && a.g.C % 64 == 0 && a.N % 128 == 0 && coords_ok) return w2 ? launch_glds<true, true>(a, o, s) : launch_glds<false, true>(a, o, s);
        if (a.g.rowmap) return hipErrorInvalidValue;          // only the LDS-DMA kernel knows the compaction
        return w2 ? launch_variant<2, 2, true, true>(a, s) : launch_variant<2, 2, true, false>(a, s);
    }
    if (narrow) return w2 ? launch_variant<4, 1, false, true>(a, s) : launch_variant<4, 1, false, false>(a, s);
    if (o.gemm_glds && a.K % 64 == 0 && a.M >= 128 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.N % 128 == 0)
        return w2 ? launch_glds<true, false>(a, o, s) : launch_glds<false, false>(a, o, s);
    return w2 ? launch_variant<2, 2, false, true>(a, s) : launch_variant<2, 2, false, false>(a, s);
}

JG_NS_END
