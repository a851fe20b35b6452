// Shared device/host declarations for libjegal_hip (gfx950 only).
//
// Two builds of the kernels live in the library: the default one with fp16 operands and, for precision mode JG_PREC_BF16, a
// second one of gemm.hip / attention.hip / elementwise.hip compiled with -DJG_BF16: there `f16` -- the 16-bit operand /
// activation type of every kernel -- is __bf16, the MFMA macros below name the bf16 instructions, and everything (this header
// included) sits in namespace bf.  api.hip includes this header twice and dispatches per handle (LAUNCH in api.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#if (defined(JG_BF16) && !defined(JG_COMMON_BF16_INCLUDED)) || (!defined(JG_BF16) && !defined(JG_COMMON_FP16_INCLUDED))
#undef JG_NS_BEGIN
#undef JG_NS_END
#undef JG_MFMA_16x16x32
#undef JG_MFMA_32x32x16
#ifdef JG_BF16
#define JG_COMMON_BF16_INCLUDED
#define JG_NS_BEGIN namespace bf {
#define JG_NS_END }
namespace bf {
typedef __bf16 f16;
#define JG_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define JG_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#else
#define JG_COMMON_FP16_INCLUDED
#define JG_NS_BEGIN
#define JG_NS_END
typedef _Float16 f16;
#define JG_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define JG_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Geometry of an NHWC implicit-GEMM convolution (C must be a power of two >= 8).
struct ConvGeom {
    int H, W, C;          // input spatial size and channels
    int OH, OW;           // output spatial size
    int KH, KW, SH, SW, PH, PW;
    int cshift;           // log2(C)
    // K order of the taps: k = (p, c) with tap p -> (kh, kw) = taps byte p (kh << 4 | kw), or the natural
    // p = kh*KW + kw when tap_table == 0.  Strided layers list their taps by parity class (kh % SH, kw % SW):
    // consecutive taps of a class touch the same input pixels shifted by whole output positions, so the re-reads
    // of a k-loop hit the L2 instead of thrashing it (conv2: 25 taps, 4 classes of 9/6/6/4).  The packed weights
    // use the same order (api.hip:make_conv).
    unsigned long long taps[4];
    int tap_table;
    // ---- position-independent leading rows (LDS-DMA conv kernels ONLY: api.hip sets these for launches that take that path).
    // Behind conv1's zero-band skip (conv1.hip) the first rows of every layer's output do not depend on the position at all:
    // they are what the layer computes from an all-constant input image, and the engine keeps those images per weight load
    // ("const chain", api.hip).  PER POSITION (image) p, conv2 leaves its first s2[p] output rows out and every deeper layer the
    // count conv_skip_decode(s2[p], op) derived from it; a consumer reads the input rows its producer left out from the const
    // image of its input instead.  The computed rows of a launch are COMPACTED: the kernel runs over m' = 0 .. *rows_total - 1 and
    //   rowmap[m'] = (full output row = (img*OH + oh)*OW + ow) | s2[img] << 24
    // gives it the pixel to compute, the row to store to and the image's count (launch_conv_rowmaps builds the maps on the
    // device from conv1_skip_mask_kernel's per-position counts; the host never sees a value, nothing synchronises).
    const int* rowmap;        // nullptr: every row is computed, m' = m
    const int* rows_total;    // device word: number of compacted rows of this launch (read at kernel start)
    int in_op;                // consumer side: input rows < conv_skip_decode(s2[img], in_op) were not computed by the producer ...
    const f16* const_in;      // ... and are read from this const image [H][W][C] of the input instead (nullptr: the input is complete)
};
// op 0: conv2 (the count itself); op 1: conv3 (3x3, stride 2, pad 1: rows whose window ends above s2); op 2: conv4 and op 3: conv5
// (3x3, vertical stride 1, pad 1: one row fewer each)
__host__ __device__ inline int conv_skip_decode(int w, int op) {
    const int s3 = w / 2;
    const int s = op == 0 ? w : s3 - (op - 1);
    return s > 0 ? s : 0;
}
// one layer of the compaction (launch_conv_rowmaps)
struct ConvRowMap {
    int OH, OW, op;           // output geometry of the layer and its conv_skip_decode op
    int* map;                 // [NF*OH*OW] (only the first *total entries are meaningful)
    int* base;                // [NF + 1] scratch: exclusive prefix of the images' computed rows
    int* total;               // device word
};
constexpr int CONV1_ZHDR_WORDS = 64;        // header of conv1's zero-scan scratch: 32 words of zconst, then ...
constexpr int CONV1_ROWSKIP_WORD = 32;      // ... the min over the launch's positions of conv2's position-independent leading rows (debug only)

#ifdef __HIPCC__
__device__ __forceinline__ void tap_decode(const ConvGeom& g, int p, int& kh, int& kw) {
    if (g.tap_table) {
        const unsigned long long w = p < 8 ? g.taps[0] : p < 16 ? g.taps[1] : p < 24 ? g.taps[2] : g.taps[3];
        const int code = (int)(w >> ((p & 7) * 8)) & 0xff;
        kh = code >> 4;
        kw = code & 15;
    } else {
        kh = p / g.KW;
        kw = p - kh * g.KW;
    }
}
#endif

// out[m][n] = epi( sum_k A[m][k] * (Wh[n][k] + Wl[n][k]) )
// epi(v) = relu?( v*scale[n] + bias[n] + res[(m % res_mod)][n] )
struct GemmArgs {
    const f16* A;         // plain: row-major [M][lda]; conv: NHWC input
    long lda;
    ConvGeom g;
    const f16* Wh;        // [N][ldw] fp16 (hi part)
    const f16* Wl;        // [N][ldw] fp16 (lo part, W2 mode) or nullptr
    long ldw;
    int M, N, K;
    const float* scale;   // per-n or nullptr
    const float* bias;    // per-n or nullptr
    const float* res;     // fp32 residual or nullptr
    long ldr;
    int res_mod;          // residual row = m % res_mod (0: m)
    float* out32;         // optional fp32 output
    f16* out16;           // optional fp16 output
    long ldc;
    int relu;
    // fused LayerNorm epilogue (row-wide tiles only, N == 512): out32/out16 receive LN(acc*scale + bias + res)
    const float* ln_w;    // nullptr: no LayerNorm
    const float* ln_b;
    int ln_flavour;       // LN_STD / LN_ANNOTATED
    // Tiled token stream of the fused GestSync transformer (see res_* below): with ln_w the residual comes from
    // (res16, res8) and the LayerNorm output goes to (out16, out8), all in the tiled order; a_tiled: the A operand
    // of a plain GEMM (K == 512) is such a tiled fp16 plane.
    const f16* res16;
    const signed char* res8;
    signed char* out8;
    int a_tiled;
    // Implicit LayerNorm (XLM-RoBERTa's post-norm layers, api.hip:xlmr_encode_impl): the token stream holds the UN-normalised rows
    // x = hi + lo (two fp16 planes; the hi plane is the next GEMM's A operand) and per row (mean, rstd) of x; LN(x) itself is never
    // materialised.  ln_mode 1 (consumer: the Linear behind the LayerNorm, weights pre-multiplied by gamma):
    //     out = rstd[m] * (acc - mean[m] * scale[n]) + bias[n]      scale = column sums of the folded weights, bias = b + W beta
    // ln_mode 2 (producer: the Linear whose output is added to LN(x_prev) and becomes the next x):
    //     v = acc + bias[n] + scale[n] * rstd[m] * (x_prev[m][n] - mean[m])     scale = gamma, bias = b + beta of that LayerNorm
    //     out16 / out_lo = hi / lo planes of v (in place over xres_hi / xres_lo is fine), stat_out[m][n / 64] = (sum, sum of squares) of
    //     the row's 64 columns n .. n + 63 (launch_ln_stats turns them into the next (mean, rstd))
    int ln_mode;
    const float* ln_stats;     // [M][2]: (mean, rstd) of the LayerNorm input rows (mode 1: of A's rows; mode 2: of x_prev's rows)
    const f16* xres_hi;        // mode 2: x_prev planes, row-major [M][ldc]
    const f16* xres_lo;
    f16* out_lo;               // mode 2: lo plane of the output
    float* stat_out;           // mode 2: [M][N / 64][2]
    // Per-clip bias (precision mode JG_PREC_FP16_RC, launch_rc_bias): row m takes bias_clip[min(m / rpc, nclips - 1)][n] instead of
    // bias[n].  LDS-DMA kernel only, through the fp16 row-transposing epilogue (out16 alone, no residual) or the LayerNorm-fused one;
    // launch_gemm rejects anything else -- there is no path that would quietly drop the correction.
    const float* bias_clip;    // [nclips][N] (the layer's bias already inside) or nullptr
    int rpc, nclips;
};

// ---- tiled token stream (N = 512 columns, row tiles of 128) -----------------------------------------------------
// The residual stream of the fused transformer is kept as fp16 + an 8-bit correction instead of fp32 (3 instead of
// 4 bytes per element to read, 3 instead of 6 to write next to the fp16 copy the next GEMM needs anyway):
//   x  =  x16 + c * 2^-13,   c = e4m3( clamp((x - x16) * 2^13, +-448) )   (one OCP fp8 byte, v_cvt_pk_fp8_f32 / v_cvt_pk_f32_fp8)
// |x - x16| <= ulp(x16)/2 and e4m3 keeps 4 significant bits of it: relative error <= 2^-16 per LayerNorm output (rms ~ 5e-6;
// 12 of them per forward pass; the embedding tolerance is 1e-3, the path measures 6e-4 with or without it).  The scale 2^13
// keeps c normal for |x| between 2^-4 and 128 (below: the absolute error is < 2^-23; above: c saturates and the element
// degrades towards plain fp16).  Round 2 stored round((x - x16) * 256/ulp(x16)) as a biased byte (2^-19): three more bits
// that the error budget never saw, for 17 VALU instructions per element in the LayerNorm epilogue instead of 5.
// Both planes are stored in the MFMA fragment order of the 128x512 LN kernel:
//   x16t element (m, n): R*65536 + (n>>6)*8192 + ((m&127)>>4)*1024 + ((n&63)>>4)*256 + (m&15)*16 + (n&15),  R = m>>7
//        (a wave's 8-byte accesses of one (j, i) block are one contiguous 512 B; a 64-wide k-tile of a 128-row
//         panel is one contiguous 16 KB -> the consumer GEMMs' LDS-DMA reads it with a_tiled addressing)
//   d8t  byte    (m, n): R*65536 + (n>>6)*8192 + ((m&127)>>4)*1024 + lane*16 + ((n&63)>>4)*4 + (n&3),
//        lane = ((n&15)>>2)*16 + (m&15)   (one 16-byte access per lane and 16-row block)
#ifdef __HIPCC__
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr float RES_SCALE = 8192.f, RES_INV_SCALE = 1.220703125e-4f;      // 2^13, 2^-13
// h01 / h23: two packed fp16 pairs (elements 0,1 and 2,3), dw: their four corrections
__device__ __forceinline__ f32x4 res_dec4(unsigned h01, unsigned h23, unsigned dw) {
    const f32x2_t c01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)dw, false), c23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)dw, true);
    const f16x2 a = __builtin_bit_cast(f16x2, h01), b = __builtin_bit_cast(f16x2, h23);
    return f32x4{__builtin_fmaf(c01.x, RES_INV_SCALE, (float)a.x), __builtin_fmaf(c01.y, RES_INV_SCALE, (float)a.y),
                 __builtin_fmaf(c23.x, RES_INV_SCALE, (float)b.x), __builtin_fmaf(c23.y, RES_INV_SCALE, (float)b.y)};
}
// four corrections packed into one dword
__device__ __forceinline__ unsigned res_enc4(float y0, float y1, float y2, float y3, f16 h0, f16 h1, f16 h2, f16 h3) {
    auto t = [](float y, f16 h) -> float {        // (y - h) * 2^13: v_mul + v_fma_mix (the fp16 operand is read as such), clamped to e4m3's range
        return __builtin_amdgcn_fmed3f(__builtin_fmaf(-(float)h, RES_SCALE, y * RES_SCALE), -448.f, 448.f);
    };
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(t(y0, h0), t(y1, h1), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(t(y2, h2), t(y3, h3), w, true);
    return (unsigned)w;
}
#endif

enum { LN_STD = 0, LN_ANNOTATED = 1 };


// ---- per-handle engine state shared with the launchers ---------------------------------------------------------
// Tuning / A-B switches and the per-device resources a launch needs.  One instance per jg_handle (api.hip), passed to
// the launchers: two handles -- on one device or on two -- never see each other's settings.
struct EngineOpts {
    int device = 0;
    int num_cu = 256;
    const f16* zeros = nullptr;          // 256-byte zero page on `device` (LDS-DMA source for padding taps / K tails)
    bool gemm_glds = true;               // LDS-DMA GEMM kernels (false: register-staged gemm_kernel everywhere)
    bool gemm_persistent = true;
    bool gemm_big_tile = true, gemm_small_tile = true, gemm_tall_tile = true;
    int gemm_tile = 0;                   // plain GEMMs: 0 = pick by the cost estimate (launch_glds), 1 / 2 / 3 = force the 128x128 / 256x128 / 256x256 tile
    int gemm_counted = 1;                // counted s_waitcnt between a tile's epilogue stores and the next tile's first DMA
    int gemm_stagger = 0;                // 10-ns ticks per phase (0: default policy, -1: off)
    bool lanes_active = false;           // the launch is part of a two-lane batch (api.hip, run_in_lanes): the other lane's kernels already
                                         // spread the store bursts, the default de-phasing only costs time there (12.22 -> 12.16 ms per step)
    unsigned long long* gemm_tl = nullptr;   // debug timeline buffer (option gemm_timeline)
    bool attn_mfma = true;
    bool conv1_zero_skip = true;
    bool conv1_mfma16 = true;            // conv1_direct_kernel's MFMA waves on 16x16x32 MFMAs (false: 32x32x16, the round-1/2 form)
};
hipError_t engine_opts_init(EngineOpts& o, int device);      // queries the CU count, allocates the zero page (current device = `device`)
void engine_opts_release(EngineOpts& o);
void engine_opts_set_timeline(EngineOpts& o, bool on);

// ---- launchers (each returns hipGetLastError()) -----------------------------------------
hipError_t launch_gemm(const GemmArgs& a, bool conv, const EngineOpts& o, hipStream_t s);
bool gemm_ln_fusable(const GemmArgs& a);
// packed_bytes >= 0: frames whose metadata points outside [0, packed_bytes) or is misaligned come out zero instead of being read
hipError_t launch_unpack_masked(const uint8_t* packed, const int* row0, const long long* offs, int n_frames, uint8_t* dst, hipStream_t s,
                                long long packed_bytes = -1);
// offs != nullptr: packed source -- frame f's rows max(mask_y[f] + 1, 0) .. H-1 start at src + offs[f] (src_bytes = size of src)
hipError_t launch_mask_resize(const uint8_t* src, int T, int H, int W, const int* mask_y_dev, uint8_t* dst, hipStream_t s,
                              const long long* offs = nullptr, long long src_bytes = 0);

hipError_t launch_stack_frames(const void* src, int src_is_u8, long sb, long st, long sh, long sw, long sc,
                               int B, int T, int pad, int H, int W, f16* dst, hipStream_t s);
// conv1 from u8 frames = three launches: launch_conv1_scan (zero bands -> skip masks, into zscratch: conv1_zmask_elems words),
// launch_conv1_direct (zscratch == nullptr: nothing is skipped), launch_conv1_edge_fix (pooled columns that straddle two strips)
hipError_t launch_conv1_scan(const uint8_t* src, int nclip, int T, int pad, const f16* Wd, float scale, unsigned* zscratch, hipStream_t s);
hipError_t launch_conv1_direct(const uint8_t* src, int nclip, int T, int pad, const f16* Wd, float scale,
                               f16* out_pooled, f16* edge, const unsigned* zscratch, bool fill_all, const EngineOpts& o, hipStream_t s);
hipError_t launch_conv1_edge_fix(f16* out_pooled, const f16* edge, long positions, hipStream_t s);
// the 64 per-channel values relu(bias) that conv1 produces over an all-zero patch, as the kernel rounds them (-> const chain)
hipError_t launch_conv1_zconst(const f16* Wd, float scale, f16* zconst, hipStream_t s);
size_t conv1_zmask_elems(int nclip, int T);
const int* conv1_s2_counts(const unsigned* zscratch, int nclip, int T, int pad);     // [nclip*P] per position: conv2's position-independent leading rows
// compaction maps of the conv layers behind conv1 from the per-position counts s2 (NF positions = images)
hipError_t launch_conv_rowmaps(const int* s2, int NF, const ConvRowMap* layers, int nlayers, hipStream_t s);
size_t conv1_edge_elems(long positions);
// s2 / in_op / const_in as in ConvGeom: input rows of image n below conv_skip_decode(s2[n], in_op) come from the const image
hipError_t launch_maxpool3x3s2(const f16* in, f16* out, int N, int H, int W, int C, hipStream_t s, const int* s2 = nullptr,
                               int in_op = 0, const f16* const_in = nullptr);
hipError_t launch_window_gather(const float* conv, const float* pe, int B, int P, int Twin, int L, int D, int shift, int tiled,
                                float* x32, f16* x16, hipStream_t s);
hipError_t launch_layernorm(const float* in, const float* w, const float* b, int rows, int D, int flavour,
                            int relu, float* out32, f16* out16, hipStream_t s);
hipError_t launch_attention(const f16* qkv, const float* keymask, int B, int S, int H, int dk, f16* out, const EngineOpts& o, hipStream_t s);
// Window structure of the GestSync clip path for attention straight from per-position projections (attention.hip):
// token j of window (clip c, frame i) comes from conv position clamp(i + j - shift, 0, P - 1) of clip c.
struct AttnGather {
    const f16* pe_qkv;    // [S][3D]: W_qkv pe[j] + b
    int Twin, P, shift;   // windows per clip, conv positions per clip, window_gather's shift
};
hipError_t launch_attention_gather(const f16* qkv_pos, const AttnGather& g, int B, int S, int H, f16* out, hipStream_t s);
// pe_qkv[j][n] = sum_k W[n][k] pe[j][k] + bias[n]  (W = Wh (+ Wl), [N][K] fp16; pe [S][K] fp32): 21 x 1536 outputs
hipError_t launch_pe_project(const float* pe, int S, const f16* Wh, const f16* Wl, const float* bias, int N, int K, f16* out, hipStream_t s);
hipError_t launch_group_mean(const f16* in, int groups, int L, int D, f16* out, hipStream_t s);
hipError_t launch_cast_f32_f16(const float* in, f16* out, long n, hipStream_t s);
hipError_t launch_transpose_tokens(const float* in, int N, int L, int D, float* out, hipStream_t s);
hipError_t launch_l2norm(const float* in, float* out, int rows, int D, hipStream_t s);
// first audio conv (5x5, 1 -> 32 channels, BN folded, ReLU) straight from the mel frames: wh / wl = the packed [32][32] weights (k = tap)
// valid (optional, device [B]): per-clip number of valid mel frames in a zero-padded batch (see zero_tail_kernel, elementwise.hip)
hipError_t launch_audio_conv0(const float* mel, int B, int Tm, int F, const f16* wh, const f16* wl, const float* bias, f16* out, const int* valid,
                              hipStream_t s);
// NHWC [B][H][..row_elems..] fp16: rows h >= len_b of clip b set to zero, len_b = valid[b] halved (len-1)/2+1 `halvings` times
hipError_t launch_zero_tail(f16* x, const int* valid, int halvings, int B, int H, long row_elems, hipStream_t s);
hipError_t launch_segment_mean(const float* seq, int D, const int32_t* seg, int n, f16* dst16, float* dst32,
                               int dst_ld, int dst_col, hipStream_t s);
hipError_t launch_fill_f16(f16* p, long n, hipStream_t s);
hipError_t launch_xlmr_embed(const int32_t* ids, int B, int L, int D, int pad_id, int vocab, int maxpos, const float* word, const float* pos,
                             const float* type, float* out, hipStream_t s);
// implicit-LayerNorm token stream (GemmArgs::ln_mode): embeddings as un-normalised hi / lo planes + per-64-column (sum, sum of squares)
hipError_t launch_xlmr_embed_planes(const int32_t* ids, int B, int L, int D, int pad_id, int vocab, int maxpos, const float* word, const float* pos,
                                    const float* type, f16* hi, f16* lo, float* part, hipStream_t s);
// part [rows][P][2] (sum, sum of squares per 64-column block, P = D / 64) -> stats [rows][2] = (mean, 1 / sqrt(var_biased + 1e-5))
hipError_t launch_ln_stats(const float* part, int rows, int P, float* stats, hipStream_t s);
// out32 = LayerNorm(hi + lo) (nn.LayerNorm, eps 1e-5), D = 768: the explicit LayerNorm at the end of the implicit chain
hipError_t launch_layernorm_planes(const f16* hi, const f16* lo, const float* w, const float* b, int rows, int D, float* out32, hipStream_t s);
hipError_t launch_gelu(const float* in, f16* out, long n, hipStream_t s);
hipError_t launch_mask_i32_f32(const int32_t* in, float* out, long n, hipStream_t s);
hipError_t launch_broadcast_channels(const f16* v, int C, f16* out, long pixels, hipStream_t s);
hipError_t launch_logmel(const float* wav, int B, int n_samples, const float* mel_basis, float* out, hipStream_t s);
// stats != nullptr ([M][2] mean, rstd): sums of the NORMALISED rows (A[m][k] - mean[m]) * rstd[m] (calibration of an implicit-LayerNorm consumer)
hipError_t launch_col_sum(const f16* A, long lda, int M, int K, float* scratch, float* out, hipStream_t s, const float* stats = nullptr);
size_t col_sum_scratch_elems(int K);
// JG_PREC_FP16_RC: out[c][n] = bias[n] + sum_k lo[n][k] * (mean over a fixed sample of clip c's rows of A[.][k]); clip c = rows c*rpc .. +rpc-1 of
// A (row-major [.][lda], or the tiled fp16 token plane when `tiled`: K == 512); scratch: rc_scratch_elems(nclips, K) floats
// valid_rows (optional, device [nclips]): rows of each clip that are its own (the rest of its rpc rows is batch padding)
hipError_t launch_rc_bias(const f16* A, long lda, int tiled, int nclips, int rpc, const int* valid_rows, const f16* lo, const float* bias, int N, int K,
                          float* scratch, float* out, hipStream_t s);
size_t rc_scratch_elems(int nclips, int K);
hipError_t launch_ragged_mean(const float* x, const int32_t* offsets, int n, int D, float* out, hipStream_t s);
hipError_t launch_sim_rank(const float* e1, const float* e2, int n_local, int n_total, int row_offset, int D,
                           int32_t* rank, int32_t* ties, hipStream_t s);
hipError_t launch_spot(const float* g, const float* c, const int32_t* goff, const int32_t* coff, const int32_t* target,
                       int n, int D, float temp, int32_t* pred, float* score, hipStream_t s);
hipError_t launch_asd(const float* q, const float* cand, const int32_t* coff, int n, int D, float temp,
                      int32_t* pred2, hipStream_t s);

#ifdef JG_BF16
}  // namespace bf
#endif
#endif  // this build's declarations
