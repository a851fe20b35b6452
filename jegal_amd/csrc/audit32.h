// fp32 "audit" kernels of libjegal_hip (precision mode JG_PREC_FP32, option "audit_stages"): every GEMM / convolution on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation), fp32 activations end to end, fp32 softmax / LayerNorm.
// The reference's CPU path is fp32 (inference_embs.py:497: autocast is a no-op without CUDA): this mode is the on-device stand-in for
// it -- what the default fp16 modes are audited against on a checkpoint the parity tests have never seen.
// Simple kernels on purpose: one GEMM / implicit-GEMM kernel, one attention kernel, a handful of elementwise kernels; nothing here is
// on the timed path.
#pragma once
#include "common.h"

// out[m][n] = act( (sum_k A[m][k] * W[n][k]) * scale[n] + bias[n] + res[m % res_mod][n] ),  everything fp32.
// conv != 0: A is an NHWC fp32 tensor and row m an output pixel (ConvGeom as in common.h; no rowmap / const image), k = (tap, c).
struct Gemm32Args {
    const float* A;
    long lda;
    ConvGeom g;
    int conv;
    const float* W;       // [N][ldw]
    long ldw;
    int M, N, K;
    const float* scale;   // per n or nullptr
    const float* bias;    // per n or nullptr
    const float* res;     // [.][ldr] or nullptr
    long ldr;
    int res_mod;          // residual row = m % res_mod (0: m)
    float* out;           // [M][ldc]
    long ldc;
    int act;              // 0 none, 1 ReLU, 2 exact GELU
};
hipError_t launch_gemm32(const Gemm32Args& a, hipStream_t s);

// softmax(q k^T / sqrt(dk) masked_fill(mask == 0, -1e9)) v per (sequence, head): qkv [B*S][3*H*dk] fp32 (q | k | v), keymask (B,S) fp32 or
// nullptr, out [B*S][H*dk].  dk = 64 or 96.
hipError_t launch_attention32(const float* qkv, const float* keymask, int B, int S, int H, int dk, float* out, hipStream_t s);

// S[b][p][h][w][16] = frame(clamp(p + dt - pad))[h][w][c] (dt < 5, c < 3; slot 15 zero); u8 sources are divided by 255 in fp32 like
// inference_embs.py:282
hipError_t launch_stack_frames32(const void* src, int src_is_u8, long sb, long st, long sh, long sw, long sc, int B, int T, int pad, int H, int W,
                                 float* dst, hipStream_t s);
hipError_t launch_maxpool3x3s2_32(const float* in, float* out, int N, int H, int W, int C, hipStream_t s);
hipError_t launch_group_mean32(const float* in, int groups, int L, int D, float* out, hipStream_t s);
// rows h >= len_b of clip b of an NHWC fp32 tensor [B][H][row_elems] set to zero, len_b = valid[b] halved ((len - 1) / 2 + 1) `halvings` times
hipError_t launch_zero_tail32(float* x, const int* valid, int halvings, int B, int H, long row_elems, hipStream_t s);

// Split-operand GEMM on the fp16 matrix cores (the two ends of the JEGAL gesture branch in the fp16 modes, option jegal_fp32_ends):
//   out[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] + res[m % res_mod][n] ),   A fp32, W = Wh + Wl (fp16 pair), out fp32,
// with A split into hi + lo fp16 IN THE LOADER and three MFMAs per fragment pair (Ah Wh + Ah Wl + Al Wh; the dropped Al Wl term is 2^-22
// of the product): fp32-grade products at 3/16 of the cost of the fp32 MFMA.  K % 256 == 0, N % 128 == 0, lda % 4 == 0.
struct GemmX3Args {
    const float* A;
    long lda;
    const f16* Wh;
    const f16* Wl;
    long ldw;
    int M, N, K;
    const float* bias;
    const float* res;
    long ldr;
    int res_mod;
    float* out;
    long ldc;
    int relu;
};
hipError_t launch_gemm_x3(const GemmX3Args& a, hipStream_t s);

// Test aid (option ws_poison): fill with 0xff bytes (fp16 / fp32 NaN) by a kernel of our own on the stream -- NOT hipMemsetAsync: two 1-GiB
// hipMemsetAsync fills running concurrently on two streams were observed to overlap the kernels enqueued BEHIND them on their own stream
// (tools/experiments/xlmr_race/xl_poison_probe.py, round 6)
hipError_t launch_poison(void* p, size_t bytes, hipStream_t s);
