// EXPERIMENT (round 2, not built): see tools/experiments/README.md.  Bit-identical to the implicit GEMM but slower (conv2 2.1 ms vs 1.66 ms).
// Stride-2 NHWC convolution as a "shift GEMM" (gfx950): conv2 of the GestSync stack (gestsync.py:48-55:
// Conv3d(64->128, k(1,5,5), s(1,2,2), p0) + BatchNorm + ReLU on the pooled conv1 map 43x78 -> 20x37).
//
// The implicit GEMM of gemm.hip stages, for EVERY one of the 25 taps, the 64 channels of that tap's input pixel for every
// output pixel of the tile: 64 KB of activations per 8.4 MFLOP k-tile, almost all of them bytes it staged a tap earlier for a
// neighbouring output pixel.  It is LDS-DMA bound (2.0 us per k-tile against 1.2 us of MFMA).  Here the taps are grouped by
// stride-parity class (kh % 2, kw % 2).  All taps of a class read ONE decimated plane of the input, P[r][q] = in[2r + ph][2q + pw],
// at whole-pixel shifts: output (oh, ow), tap (kh, kw) = (2 dh + ph, 2 dw + pw) reads P[oh + dh][ow + dw].  The output is walked
// in the plane's own raster (PC = 39 columns, PR = 22 rows per image; the 2 extra rows and 2 extra columns are computed and
// dropped: 16 % padding), so that "output pixel m, tap (dh, dw)" is plane pixel m + dh * PC + dw: the activation operand of every
// tap of a class is the SAME staged tile read at a row offset.  Per 256-pixel tile the activation DMA drops from 25 x 32 KB to
// 4 x 42 KB, the tile becomes weight-DMA + MFMA bound.
//
//   tile      256 raster pixels x 128 output channels, 8 waves (4 x 2), wave tile 64 x 64 (v_mfma_f32_16x16x32_f16, weights = A operand)
//   LDS       2 plane buffers (336 rows x 128 B, XOR-swizzled by row) + 2 weight stages (128 x 128 B) + epilogue scratch
//   pipeline  one barrier per tap; behind it the next tap's weights and (during the first taps of a class) 2 pieces per wave of the
//             NEXT class's plane -- or the next tile's first plane -- are issued; vmcnt(0) in front of the next barrier
//   epilogue  bias + ReLU, transposed through LDS into whole 128-B row segments, rows outside the image dropped
#include "common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

namespace {
constexpr int BM = 256, BN = 128, CIN = 64;
constexpr int PLANE_ROWS = 384;                       // staged rows per plane buffer (6 pieces of 8 rows per wave); 256 + 2*39 + 2 = 336 needed
constexpr int PLANE_BYTES = PLANE_ROWS * 128;         // 49152
constexpr int W_BYTES = BN * 128;                     // 16384
constexpr int OFF_W = 2 * PLANE_BYTES;                // 98304
constexpr int OFF_SCR = OFF_W + 2 * W_BYTES;          // 131072
constexpr int TP16 = 144;                             // scratch row pitch (16-B aligned, conflict-free b64 writes)
constexpr int LDS_TOTAL = OFF_SCR + 8 * 16 * TP16;    // 149504
constexpr int MAX_TAPS = 32;
}

struct ShiftConvArgs {
    const f16* in;          // [NF][H][W][64]
    const f16* Wt;          // [128][ntaps * 64], taps in parity-class order (api.hip make_conv, reorder = true)
    const float* bias;      // [128]
    f16* out;               // [NF][OH][OW][128]
    int NF, H, W, OH, OW;
    int PR, PC;             // plane rows / columns = raster of one image
    long total;             // NF * PR * PC raster pixels
    int ntaps;
    int tap_off[MAX_TAPS];  // dh * PC + dw
    int tap_cls[MAX_TAPS];  // class index 0..3 (ph * 2 + pw)
    // what is staged behind the barrier of tap t besides the next tap's weights:
    //   tap_stage[t] = -1 nothing, else piece pair p (0..2) | class << 2 | (plane of the NEXT tile's first class) << 4
    int tap_stage[MAX_TAPS];
    const f16* zeros;
    int tiles;
    float inv_raster, inv_pc;
};

template <int N> __device__ __forceinline__ void cs_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512) void conv_shift_kernel(ShiftConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3, pc = lane & 7;
    const int G = gridDim.x;
    const long img_elems = (long)a.H * a.W * CIN;
    const int raster = a.PR * a.PC;
    // raster pixel v (< 2^24, checked by the launcher) -> (image, plane row, plane column) with float reciprocals + one fix-up
    auto split = [&](int v, int& img, int& r, int& q) {
        img = (int)((float)v * a.inv_raster);
        int u = v - img * raster;
        if (u < 0) { --img; u += raster; } else if (u >= raster) { ++img; u -= raster; }
        r = (int)((float)u * a.inv_pc);
        q = u - r * a.PC;
        if (q < 0) { --r; q += a.PC; } else if (q >= a.PC) { ++r; q -= a.PC; }
    };

    // persistent, XCD-aware: the workgroups of one XCD (b, b+8, ...) take consecutive tiles of a round (neighbouring tiles share
    // their 80-row halo and the weights in that L2)
    auto tile_of = [&](int round) -> int {
        const int v0 = round * G;
        if (v0 >= a.tiles) return -1;
        const int cnt = a.tiles - v0 < G ? a.tiles - v0 : G;
        const int b = blockIdx.x;
        if (b >= cnt) return -1;
        const int q = cnt / 8, rr = cnt % 8, xcd = b % 8, loc = b / 8;
        return v0 + (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    };

    // ---- plane staging: wave-instruction i of a wave covers staged rows (wave * 6 + i) * 8 .. +8; per lane the source of its
    // (row, swizzled chunk) in the class (0,0) plane; the other classes add (ph * W + pw) * 64 elements.  `oob` bit 0: the pixel
    // lies outside the tensor for every class (beyond the last image), bit 1: its plane row is the extra row of the odd-row classes.
    const f16* psrc[6];
    int poob[6];
    auto plane_setup = [&](int tile) {
        const long m0 = (long)tile * BM;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int R = (wave * 6 + i) * 8 + lrow;
            const int c = pc ^ ((R >> 1) & 7);
            const long v = m0 + R;
            int img, r, q;
            split((int)(v < a.total ? v : a.total - 1), img, r, q);
            int oob = v >= a.total ? 1 : 0;
            if (2 * r + 1 >= a.H) oob |= 2;                 // row 2r+1 does not exist (r = PR - 1 when H is odd)
            if (2 * r >= a.H) oob |= 1;
            if (2 * q + 1 >= a.W) oob |= 4;                 // column 2q+1 does not exist
            if (2 * q >= a.W) oob |= 1;
            psrc[i] = a.in + (long)img * img_elems + ((long)(2 * r) * a.W + 2 * q) * CIN + c * 8;
            poob[i] = oob;
        }
    };
    auto plane_piece = [&](int i, int cls, int buf) __attribute__((always_inline)) {
        const int ph = cls >> 1, pw = cls & 1;
        const bool bad = (poob[i] & 1) || (ph && (poob[i] & 2)) || (pw && (poob[i] & 4));
        const f16* src = bad ? a.zeros : psrc[i] + ((long)ph * a.W + pw) * CIN;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + buf * PLANE_BYTES + (wave * 6 + i) * 1024), 16, 0, 0);
    };
    // ---- weight staging: 128 rows x 128 B per tap = 16 pieces, 2 per wave
    const f16* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + lrow;
        const int c = pc ^ ((row >> 1) & 7);
        wsrc[i] = a.Wt + (long)row * (a.ntaps * CIN) + c * 8;
    }
    auto w_stage = [&](int tap, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[i] + tap * CIN), (lds_ptr_t)(smem + OFF_W + buf * W_BYTES + (wave * 2 + i) * 1024), 16, 0, 0);
    };

    // bias of this lane's 4 x 4 output channels
    f32x4 bi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bi[i] = *reinterpret_cast<const f32x4*>(a.bias + wn * 64 + i * 16 + fq * 4);

    int round = 0;
    int tile = tile_of(0);
    if (tile < 0) return;
    plane_setup(tile);
    // prologue of the first tile: its first class's plane and the first tap's weights
#pragma unroll
    for (int i = 0; i < 6; ++i) plane_piece(i, a.tap_cls[0], 0);
    w_stage(0, 0);
    int pbuf = 0;                 // plane buffer of the current class
    int wbuf = 0;                 // weight stage of the current tap (alternates across tiles too)

    while (true) {
        const int ntile = tile_of(round + 1);
        const long m0 = (long)tile * BM;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int tap = 0; tap < a.ntaps; ++tap) {
            if (tap > 0 && a.tap_cls[tap] != a.tap_cls[tap - 1]) pbuf ^= 1;
            cs_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // this tap's weights (and plane) have landed; the other stage / plane buffer is free
            // ---- behind the barrier: the next tap's weights and two pieces per wave of the plane that is needed next
            const bool last = tap + 1 == a.ntaps;
            if (!last) w_stage(tap + 1, wbuf ^ 1);
            else if (ntile >= 0) w_stage(0, wbuf ^ 1);
            const int st = a.tap_stage[tap];
            if (st >= 0 && (!(st & 16) || ntile >= 0)) {
                const int p = st & 3, ncls = (st >> 2) & 3;
                if ((st & 16) && p == 0) plane_setup(ntile);       // this tile's planes are all staged: switch to the next tile's pixels
                plane_piece(2 * p, ncls, pbuf ^ 1);
                plane_piece(2 * p + 1, ncls, pbuf ^ 1);
            }
            // ---- this tap: 64 x 64 x 64 per wave
            const char* sP = smem + pbuf * PLANE_BYTES;
            const char* sW = smem + OFF_W + wbuf * W_BYTES;
            const int o = a.tap_off[tap];
            const int fswx = ((frow + o) >> 1) & 7;                 // swizzle of the shifted activation rows (row bases are multiples of 16)
            const int fsww = (frow >> 1) & 7;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f16x8 wf[4], xf[4];
                const int chw = ((kk * 4 + fq) ^ fsww) << 4;
                const int chx = ((kk * 4 + fq) ^ fswx) << 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(sW + (wn * 64 + i * 16 + frow) * 128 + chw);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const f16x8*>(sP + (wm * 64 + j * 16 + frow + o) * 128 + chx);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
            wbuf ^= 1;
        }
        pbuf ^= 1;                // the next tile's first class was staged into the other plane buffer

        // ---- epilogue: bias, ReLU, fp16; 16-row blocks transposed through this wave's LDS scratch into 128-B row segments
        char* tsc = smem + OFF_SCR + wave * (16 * TP16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[i][j] + bi[i];
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                const f16x4 hv = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                *reinterpret_cast<f16x4*>(tsc + frow * TP16 + i * 32 + fq * 8) = hv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int rr = h2 * 8 + (lane >> 3);
                const f16x8 ov = *reinterpret_cast<const f16x8*>(tsc + rr * TP16 + (lane & 7) * 16);
                const long v = m0 + wm * 64 + j * 16 + rr;          // raster pixel of this lane's row
                int img, oh, ow;
                split((int)(v < a.total ? v : a.total - 1), img, oh, ow);
                if (v < a.total && oh < a.OH && ow < a.OW)
                    __builtin_nontemporal_store(ov, reinterpret_cast<f16x8*>(a.out + (((long)img * a.OH + oh) * a.OW + ow) * BN + wn * 64 + (lane & 7) * 8));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (ntile < 0) break;
        tile = ntile;
        ++round;
    }
}

hipError_t launch_conv_shift(const f16* in, int NF, int H, int W, const f16* Wt, const float* bias, int KH, int KW, f16* out,
                             const EngineOpts& o, hipStream_t s) {
    static bool attr_set[64] = {};
    if (o.device < 0 || o.device >= 64) return hipErrorInvalidDevice;
    if (!attr_set[o.device]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_shift_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
        attr_set[o.device] = true;
    }
    if (NF <= 0) return hipSuccess;
    ShiftConvArgs a;
    a.in = in; a.Wt = Wt; a.bias = bias; a.out = out;
    a.NF = NF; a.H = H; a.W = W;
    a.OH = (H - KH) / 2 + 1;
    a.OW = (W - KW) / 2 + 1;
    const int DH = (KH - 1) / 2, DW = (KW - 1) / 2;        // largest row / column shift
    a.PR = a.OH + DH;                                     // plane rows a valid output can touch
    a.PC = a.OW + DW;
    if (a.PR * 2 - 1 > H + 1 || a.PC * 2 - 1 > W + 1) return hipErrorInvalidValue;
    a.total = (long)NF * a.PR * a.PC;
    // taps in parity-class order, exactly as api.hip packs the weights (tap_order(..., reorder = true))
    int n = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw)
            for (int kh = ph; kh < KH; kh += 2)
                for (int kw = pw; kw < KW; kw += 2) {
                    if (n >= MAX_TAPS) return hipErrorInvalidValue;
                    a.tap_off[n] = (kh >> 1) * a.PC + (kw >> 1);
                    a.tap_cls[n] = ph * 2 + pw;
                    ++n;
                }
    a.ntaps = n;
    if (BM + DH * a.PC + DW > PLANE_ROWS || a.total >= (1L << 24)) return hipErrorInvalidValue;
    for (int t = 0; t < n; ++t) a.tap_stage[t] = -1;
    for (int s0 = 0; s0 < n;) {                         // class segments [s0, e0): stage the following class over their first 3 taps
        int e0 = s0;
        while (e0 < n && a.tap_cls[e0] == a.tap_cls[s0]) ++e0;
        if (e0 - s0 < 3) return hipErrorInvalidValue;
        const bool next_tile = e0 == n;
        const int ncls = next_tile ? a.tap_cls[0] : a.tap_cls[e0];
        for (int p = 0; p < 3; ++p) a.tap_stage[s0 + p] = p | (ncls << 2) | (next_tile ? 16 : 0);
        s0 = e0;
    }
    a.inv_raster = 1.0f / (float)(a.PR * a.PC);
    a.inv_pc = 1.0f / (float)a.PC;
    a.zeros = o.zeros;
    a.tiles = (int)((a.total + BM - 1) / BM);
    const int grid = a.tiles < o.num_cu ? a.tiles : o.num_cu;
    hipLaunchKernelGGL(conv_shift_kernel, dim3((unsigned)grid), dim3(512), LDS_TOTAL, s, a);
    return hipGetLastError();
}
