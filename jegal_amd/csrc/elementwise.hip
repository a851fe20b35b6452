// HBM-bound helper kernels (gfx950): layout changes, pooling, both LayerNorm flavours,
// L2-normalise, ragged means.  One wave (64 lanes) per row for the row reductions; 16-byte
// vector accesses wherever the layout allows (guide G13).
#include "common.h"

JG_NS_BEGIN

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Temporal stacking for conv1 (v0 path): S[b][p][h][w][16] = {frame(p+dt)[h][w][c] : dt<5, c<3} + 0
// frame index = clamp(p + dt - pad, 0, T-1)  (edge padding of inference_embs.py:283; pad=0 for
// the raw-window path).  u8 sources are kept as exact integers (fp16 holds 0..255 exactly); the
// 1/255 is applied in fp32 in the conv1 epilogue.
template <typename SRC>
__global__ void stack_frames_kernel(const SRC* __restrict__ src, long sb, long st, long sh, long sw, long sc,
                                    int B, int T, int pad, int H, int W, f16* __restrict__ dst) {
    const int P = T + 2 * pad - 4;
    const long total = (long)B * P * H * W;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int w = idx % W;
        long r = idx / W;
        const int h = r % H;
        r /= H;
        const int p = r % P;
        const int b = r / P;
        f16 v[16];
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
            int f = p + dt - pad;
            f = f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
            const SRC* s = src + b * sb + f * st + h * sh + w * sw;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[dt * 3 + c] = (f16)(float)s[c * sc];
        }
        v[15] = (f16)0.f;
        uint4* d = reinterpret_cast<uint4*>(dst + idx * 16);
        d[0] = *reinterpret_cast<uint4*>(&v[0]);
        d[1] = *reinterpret_cast<uint4*>(&v[8]);
    }
}

hipError_t launch_stack_frames(const void* src, int src_is_u8, long sb, long st, long sh, long sw, long sc,
                               int B, int T, int pad, int H, int W, f16* dst, hipStream_t s) {
    const long total = (long)B * (T + 2 * pad - 4) * H * W;
    const int grid = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    if (src_is_u8)
        hipLaunchKernelGGL(stack_frames_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (const uint8_t*)src, sb, st, sh, sw, sc, B, T, pad, H, W, dst);
    else
        hipLaunchKernelGGL(stack_frames_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, sb, st, sh, sw, sc, B, T, pad, H, W, dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MaxPool (1,3,3)/(1,2,2), no padding, NHWC fp16, 8 channels per thread (gestsync.py:42-45,74-77).
__global__ void maxpool_kernel(const f16* __restrict__ in, f16* __restrict__ out, int N, int H, int W, int C, int OH, int OW,
                               const int* __restrict__ s2, int in_op, const f16* __restrict__ const_in) {
    const bool skip = s2 && const_in;
    const int cv = C / 8;
    const long total = (long)N * OH * OW * cv;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c8 = idx % cv;
        long r = idx / cv;
        const int ow = r % OW;
        r /= OW;
        const int oh = r % OH;
        const int n = r / OH;
        const int rin = skip ? conv_skip_decode(s2[n], in_op) : 0;      // image n's producer left its first rin rows to the const image
        f16x8 m;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 + kh;
                const f16* img = ih < rin ? const_in : in + (long)n * H * W * C;
                const f16x8 v = *reinterpret_cast<const f16x8*>(img + ((long)ih * W + ow * 2 + kw) * C + c8 * 8);
                if (kh == 0 && kw == 0) m = v;
                else
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        *reinterpret_cast<f16x8*>(out + idx * 8) = m;
    }
}

hipError_t launch_maxpool3x3s2(const f16* in, f16* out, int N, int H, int W, int C, hipStream_t s, const int* s2, int in_op,
                               const f16* const_in) {
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const long total = (long)N * OH * OW * (C / 8);
    const int grid = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid), dim3(256), 0, s, in, out, N, H, W, C, OH, OW, s2, in_op, const_in);
    return hipGetLastError();
}

// Projection of the positional rows through a Linear layer (layer-0 qkv by linearity, attention.hip): one wave per output.
__global__ __launch_bounds__(256) void pe_project_kernel(const float* __restrict__ pe, int S, const f16* __restrict__ Wh, const f16* __restrict__ Wl,
                                                         const float* __restrict__ bias, int N, int K, f16* __restrict__ out) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= S * N) return;
    const int j = o / N, n = o - j * N;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float w = (float)Wh[(long)n * K + k];
        if (Wl) w += (float)Wl[(long)n * K + k];
        acc += w * pe[(long)j * K + k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) out[o] = (f16)(acc + (bias ? bias[n] : 0.f));
}

hipError_t launch_pe_project(const float* pe, int S, const f16* Wh, const f16* Wl, const float* bias, int N, int K, f16* out, hipStream_t s) {
    hipLaunchKernelGGL(pe_project_kernel, dim3((unsigned)((S * N + 3) / 4)), dim3(256), 0, s, pe, S, Wh, Wl, bias, N, K, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Window gather + positional encoding (gestsync.py:152, windowing of inference_embs.py:488-492):
// x[(b,i,j)][:] = conv[b][clamp(i+j-shift, 0, P-1)][:] + pe[j][:]   i < Twin, j < L.  conv is (B,P,D) fp32.
// shift/clamp: with edge padding the first 9 and last 9 padded-clip positions see five copies of the
// same frame, so the conv stack is only evaluated for the T+4 distinct positions (shift = 8).
__global__ void window_gather_kernel(const float* __restrict__ conv, const float* __restrict__ pe, int B, int P, int Twin,
                                     int L, int D, int shift, float* __restrict__ x32, f16* __restrict__ x16) {
    const int dv = D / 4;
    const long total = (long)B * Twin * L * dv;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int d4 = idx % dv;
        long r = idx / dv;
        const int j = r % L;
        r /= L;
        const int i = r % Twin;
        const int b = r / Twin;
        int pp = i + j - shift;
        pp = pp < 0 ? 0 : (pp > P - 1 ? P - 1 : pp);
        f32x4 v = *reinterpret_cast<const f32x4*>(conv + ((long)b * P + pp) * D + d4 * 4);
        v += *reinterpret_cast<const f32x4*>(pe + (long)j * D + d4 * 4);
        const f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
        *reinterpret_cast<f32x4*>(x32 + idx * 4) = v;
        if (x16) *reinterpret_cast<f16x4*>(x16 + idx * 4) = h;      // (nullptr: the fp32 audit path)
    }
}

// Tiled variant (token stream of the fused GEMM+LayerNorm kernel, layouts and res_enc in common.h; x32 is the 8-bit
// correction plane): one wave per 16-row x 64-column block, lane = (column quad ng, row m15) exactly as in the LN epilogue
// of gemm_glds_kernel -- a lane holds columns 16q + 4ng .. +3 (q = 0..3) of its row, so every store instruction of the wave
// covers a contiguous 512 B (fp16 plane, one per q) or 1 KB (correction plane).  (The element-per-thread kernel above wrote
// 32-byte pieces 512 B apart: 67 us for the 155 MB of a 32-clip chunk.)
__global__ __launch_bounds__(256) void window_gather_tiled_kernel(const float* __restrict__ conv, const float* __restrict__ pe, int B, int P, int Twin,
                                                                  int L, int shift, signed char* __restrict__ d8, f16* __restrict__ x16) {
    const long M = (long)B * Twin * L;
    const long nblk = ((M + 15) >> 4) * 8;
    const long blk = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (blk >= nblk) return;
    const int lane = threadIdx.x & 63, m15 = lane & 15, ng = lane >> 4;
    const long rb = blk >> 3;                       // 16-row block
    const int cb = (int)(blk & 7);                  // 64-column block
    long row = rb * 16 + m15;
    const bool live = row < M;
    row = live ? row : M - 1;
    // 32-bit index math (the launcher checks M < 2^31): 64-bit divisions by run-time values cost ~100 VALU instructions each and
    // made this store-bound kernel VALU-bound
    const unsigned r32 = (unsigned)row;
    const unsigned r2 = r32 / (unsigned)L;
    const int j = (int)(r32 - r2 * (unsigned)L);
    const unsigned b = r2 / (unsigned)Twin;
    const int i = (int)(r2 - b * (unsigned)Twin);
    int pp = i + j - shift;
    pp = pp < 0 ? 0 : (pp > P - 1 ? P - 1 : pp);
    const float* src = conv + ((long)b * P + pp) * 512 + cb * 64 + 4 * ng;
    const float* pes = pe + (long)j * 512 + cb * 64 + 4 * ng;
    const long base = (rb >> 3) * 65536 + (long)cb * 8192 + (rb & 7) * 1024;
    unsigned dq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + 16 * q);
        v += *reinterpret_cast<const f32x4*>(pes + 16 * q);
        const f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
        if (live) __builtin_nontemporal_store(h, reinterpret_cast<f16x4*>(x16 + base + q * 256 + m15 * 16 + 4 * ng));
        if (d8) dq[q] = res_enc4(v.x, v.y, v.z, v.w, h[0], h[1], h[2], h[3]);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // (d8 == nullptr: the token stream is the fp16 plane alone, option stream_fp16)
    if (live && d8) __builtin_nontemporal_store(u32x4{dq[0], dq[1], dq[2], dq[3]}, reinterpret_cast<u32x4*>(d8 + base + lane * 16));
}

hipError_t launch_window_gather(const float* conv, const float* pe, int B, int P, int Twin, int L, int D, int shift, int tiled,
                                float* x32, f16* x16, hipStream_t s) {
    if (tiled && D != 512) return hipErrorInvalidValue;
    if (tiled) {
        if ((long)B * Twin * L >= (1L << 31)) return hipErrorInvalidValue;
        const long nblk = (((long)B * Twin * L + 15) >> 4) * 8;
        hipLaunchKernelGGL(window_gather_tiled_kernel, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, s, conv, pe, B, P, Twin, L, shift,
                           tiled == 2 ? nullptr : reinterpret_cast<signed char*>(x32), x16);          // tiled == 2: no correction plane
        return hipGetLastError();
    }
    const long total = (long)B * Twin * L * (D / 4);
    const int grid = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(window_gather_kernel, dim3(grid), dim3(256), 0, s, conv, pe, B, P, Twin, L, D, shift, x32, x16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm, one wave per row, fp32 statistics.
//   LN_STD       : nn.LayerNorm   (x-mean)/sqrt(var_biased+1e-5)*w+b          (gestsync.py:20, jegal.py:26)
//   LN_ANNOTATED : modules.py:32-35  w*(x-mean)/(std_unbiased+1e-6)+b
// D = 64*4*V (V = 2 for 512, 3 for 768).
template <int V>
__global__ void layernorm_kernel(const float* in, const float* __restrict__ w, const float* __restrict__ b,
                                 int rows, int flavour, int relu, float* out32, f16* __restrict__ out16) {
    constexpr int D = 256 * V;
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = in + row * D;
    f32x4 v[V];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(x + (i * 64 + lane) * 4);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        v[i] -= mean;
        sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    sq = wave_sum(sq);
    float inv;
    if (flavour == LN_STD) inv = 1.f / sqrtf(sq * (1.f / D) + 1e-5f);
    else inv = 1.f / (sqrtf(sq * (1.f / (D - 1))) + 1e-6f);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int col = (i * 64 + lane) * 4;
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + col);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + col);
        f32x4 y = v[i] * inv * ww + bb;
        if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        if (out32) *reinterpret_cast<f32x4*>(out32 + row * D + col) = y;
        if (out16) {
            f16x4 h = {(f16)y.x, (f16)y.y, (f16)y.z, (f16)y.w};
            *reinterpret_cast<f16x4*>(out16 + row * D + col) = h;
        }
    }
}

hipError_t launch_layernorm(const float* in, const float* w, const float* b, int rows, int D, int flavour,
                            int relu, float* out32, f16* out16, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int grid = (rows + 3) / 4;
    if (D == 512) hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, s, in, w, b, rows, flavour, relu, out32, out16);
    else if (D == 768) hipLaunchKernelGGL(layernorm_kernel<3>, dim3(grid), dim3(256), 0, s, in, w, b, rows, flavour, relu, out32, out16);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Mean over groups of L consecutive rows (fp16 in, fp32 sum, fp16 out): the mean(-1) of
// inference_embs.py:511 moved in front of ff_vid.2 (it commutes with the Linear).
__global__ void group_mean_kernel(const f16* __restrict__ in, int groups, int L, int D, f16* __restrict__ out) {
    const int dv = D / 8;
    const long total = (long)groups * dv;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int d8 = idx % dv;
        const long g = idx / dv;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < L; ++j) {
            const f16x8 v = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(in + (g * L + j) * D + d8 * 8));      // read once
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
        f16x8 o;
        const float inv = 1.f / L;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(acc[e] * inv);
        *reinterpret_cast<f16x8*>(out + idx * 8) = o;
    }
}

hipError_t launch_group_mean(const f16* in, int groups, int L, int D, f16* out, hipStream_t s) {
    const long total = (long)groups * (D / 8);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(group_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, groups, L, D, out);
    return hipGetLastError();
}

__global__ void cast_kernel(const float* __restrict__ in, f16* __restrict__ out, long n4) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + idx * 4);
        f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
        *reinterpret_cast<f16x4*>(out + idx * 4) = h;
    }
}

hipError_t launch_cast_f32_f16(const float* in, f16* out, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long n4 = n / 4;   // all call sites have n % 4 == 0
    const int grid = (int)((n4 + 255) / 256 < 65536 * 4 ? (n4 + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(cast_kernel, dim3(grid), dim3(256), 0, s, in, out, n4);
    return hipGetLastError();
}

// (N, L, D) -> (N, D, L): the .transpose(1,2) of gestsync.py:156 for the drop-in forward_vid output.
__global__ void transpose_tokens_kernel(const float* __restrict__ in, int L, int D, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const long n = blockIdx.z;
    const int d0 = blockIdx.x * 32, l0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int l = l0 + r, d = d0 + threadIdx.x;
        tile[r][threadIdx.x] = (l < L && d < D) ? in[(n * L + l) * D + d] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int d = d0 + r, l = l0 + threadIdx.x;
        if (d < D && l < L) out[(n * D + d) * L + l] = tile[threadIdx.x][r];
    }
}

hipError_t launch_transpose_tokens(const float* in, int N, int L, int D, float* out, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_tokens_kernel, dim3((D + 31) / 32, (L + 31) / 32, N), dim3(32, 8), 0, s, in, L, D, out);
    return hipGetLastError();
}

// F.normalize(p=2, dim=-1, eps=1e-12): x / max(||x||, eps)  (inference_embs.py:631,635). One wave per row.
__global__ void l2norm_kernel(const float* in, float* out, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = in + row * D;
    float sq = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
        sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(sq)), 1e-12f);
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
        v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
        *reinterpret_cast<f32x4*>(out + row * D + c) = v;
    }
}

hipError_t launch_l2norm(const float* in, float* out, int rows, int D, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in, out, rows, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// First audio conv (jegal.py:42, Conv2d(1,32,5,pad 2) + BN + ReLU) directly (round 3; rounds 1-2: an im2col kernel + a register-staged
// K = 32 GEMM, 0.24 ms of the 0.64 ms audio CNN at B = 64; now 0.10 ms): out[(b,t,f)][32] = relu(sum_taps x16 * w16 + bias), the same operands the GEMM saw (mel and the BN-folded weights
// rounded to the 16-bit operand type, fp32 accumulation; only the summation order differs).  One thread per output pixel: the
// 5 x 5 neighbourhood comes from a mel tile in LDS, the 25 x 32 weights are wave-uniform (fp32 copy in LDS, broadcast reads),
// 800 FMAs per pixel, one 64-byte NHWC store.  W2 modes carry the lo part of the weights in the same fp32 copy.
// valid (optional, [B]): clip b holds only valid[b] mel frames -- frames beyond are read as the zero padding the clip would see alone,
// and output rows beyond are written as zeros (the next layer's zero padding): see zero_tail_kernel.
__global__ __launch_bounds__(256) void audio_conv0_kernel(const float* __restrict__ mel, int B, int Tm, int F, const f16* __restrict__ wh,
                                                          const f16* __restrict__ wl, const float* __restrict__ bias, f16* __restrict__ out,
                                                          const int* __restrict__ valid) {
    constexpr int TR = 3, FW_MAX = 84;                   // 3 output rows x F (<= 80) per block: (TR + 4) x (F + 4) mel values staged
    __shared__ float sm[(TR + 4) * FW_MAX];
    __shared__ float sw[25 * 32];
    __shared__ float sb[32];
    const int tid = threadIdx.x;
    const int tiles_t = (Tm + TR - 1) / TR;
    const int b = blockIdx.x / tiles_t, t0 = (blockIdx.x - b * tiles_t) * TR;
    const int Tv = valid ? min(max(valid[b], 0), Tm) : Tm;
    for (int i = tid; i < 25 * 32; i += 256) {
        const int tap = i >> 5, oc = i & 31;
        sw[i] = (float)wh[oc * 32 + tap] + (wl ? (float)wl[oc * 32 + tap] : 0.f);
    }
    if (tid < 32) sb[tid] = bias[tid];
    const int FW = F + 4;
    for (int i = tid; i < (TR + 4) * FW; i += 256) {
        const int r = i / FW, c = i - r * FW;
        const int tt = t0 + r - 2, ff = c - 2;
        float x = 0.f;
        if (tt >= 0 && tt < Tv && ff >= 0 && ff < F) x = mel[((long)b * Tm + tt) * F + ff];
        sm[r * FW_MAX + c] = (float)(f16)x;             // the operand rounding of the GEMM path
    }
    __syncthreads();
    const int r = tid / F, f = tid - r * F;
    if (r >= TR || t0 + r >= Tm) return;
    uint4* d = reinterpret_cast<uint4*>(out + (((long)b * Tm + t0 + r) * F + f) * 32);
    if (t0 + r >= Tv) {                                  // beyond the clip's own frames: the next layer's zero padding
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    float acc[32];
#pragma unroll
    for (int oc = 0; oc < 32; ++oc) acc[oc] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
            const float x = sm[(r + kh) * FW_MAX + f + kw];
            const float* w = sw + (kh * 5 + kw) * 32;
#pragma unroll
            for (int oc = 0; oc < 32; ++oc) acc[oc] = __builtin_fmaf(x, w[oc], acc[oc]);
        }
    f16 o[32];
#pragma unroll
    for (int oc = 0; oc < 32; ++oc) o[oc] = (f16)fmaxf(acc[oc] + sb[oc], 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<uint4*>(&o[q * 8]);
}

hipError_t launch_audio_conv0(const float* mel, int B, int Tm, int F, const f16* wh, const f16* wl, const float* bias, f16* out, const int* valid,
                              hipStream_t s) {
    if (B <= 0 || Tm <= 0) return hipSuccess;
    if (F < 1 || F > 80 || 3 * F > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(audio_conv0_kernel, dim3((unsigned)(B * ((Tm + 2) / 3))), dim3(256), 0, s, mel, B, Tm, F, wh, wl, bias, out, valid);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Per-clip valid lengths in a zero-padded audio batch (evaluation/extract_jegal_embs.py:141 runs batch_size = 1: a clip never
// sees another clip's length).  Layer outputs are NHWC [b][h][w][c] with h = time; a clip of valid[b] mel frames has
// len = valid[b] rows after the stride-1 layers and (len - 1) / 2 + 1 after each stride-2 layer (3x3, pad 1).  Rows h >= len of
// clip b are set to zero, which is exactly the zero padding the next conv layer would apply to the clip alone -- the rows
// h < len of every layer then do not depend on the batch's longest clip.  One block per (b, h) row.
__global__ __launch_bounds__(256) void zero_tail_kernel(f16* __restrict__ x, const int* __restrict__ valid, int halvings, int H, int row_vec) {
    const int b = blockIdx.x / H, hrow = blockIdx.x - b * H;
    int len = max(valid[b], 0);
    for (int i = 0; i < halvings; ++i) len = len > 0 ? (len - 1) / 2 + 1 : 0;
    if (hrow < len) return;
    uint4* p = reinterpret_cast<uint4*>(x) + (long)blockIdx.x * row_vec;
    for (int i = threadIdx.x; i < row_vec; i += 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

hipError_t launch_zero_tail(f16* x, const int* valid, int halvings, int B, int H, long row_elems, hipStream_t s) {
    if (B <= 0 || H <= 0 || !valid) return hipSuccess;
    if (row_elems % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(zero_tail_kernel, dim3((unsigned)(B * H)), dim3(256), 0, s, x, valid, halvings, H, (int)(row_elems / 8));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Ragged word pooling (jegal.py:174-180,189-195): dst[seg.dst][col..col+D) = mean(seq[seg.start : seg.end]).
// seg = int32 triplets (start_row, end_row_exclusive, dst_row).  One wave per segment.
__global__ void segment_mean_kernel(const float* __restrict__ seq, int D, const int32_t* __restrict__ seg, int n,
                                    f16* __restrict__ dst16, float* __restrict__ dst32, int dst_ld, int dst_col) {
    const int lane = threadIdx.x & 63;
    const int sidx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (sidx >= n) return;
    const int s0 = seg[sidx * 3], s1 = seg[sidx * 3 + 1], dr = seg[sidx * 3 + 2];
    const float inv = 1.f / (float)(s1 - s0);
    for (int c = lane; c < D; c += 64) {
        float acc = 0.f;
        for (int r = s0; r < s1; ++r) acc += seq[(long)r * D + c];
        const float m = (s1 - s0 > 1) ? acc * inv : acc;
        if (dst16) dst16[(long)dr * dst_ld + dst_col + c] = (f16)m;
        if (dst32) dst32[(long)dr * dst_ld + dst_col + c] = m;
    }
}

hipError_t launch_segment_mean(const float* seq, int D, const int32_t* seg, int n, f16* dst16, float* dst32,
                               int dst_ld, int dst_col, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(segment_mean_kernel, dim3((n + 3) / 4), dim3(256), 0, s, seq, D, seg, n, dst16, dst32, dst_ld, dst_col);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// XLM-RoBERTa front end (third-party transformers.XLMRobertaModel, call site jegal.py:116-129).
// Embeddings: word[id] + position[pid] + token_type[0], pid = padding_idx + (number of non-pad tokens up to and including
// this one) for non-pad tokens and padding_idx for pads (create_position_ids_from_input_ids).  One wave per token row.
__global__ __launch_bounds__(256) void xlmr_embed_kernel(const int32_t* __restrict__ ids, int B, int L, int D, int pad_id, int vocab, int maxpos,
                                                         const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
                                                         float* __restrict__ out) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)B * L) return;
    const int b = (int)(row / L), t = (int)(row - (long)b * L);
    int cnt = 0;
    for (int j = lane; j <= t; j += 64) cnt += ids[(long)b * L + j] != pad_id ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    int id = ids[row];
    int pid = id != pad_id ? pad_id + cnt : pad_id;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    pid = pid >= maxpos ? maxpos - 1 : pid;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(word + (long)id * D + c);
        const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (long)pid * D + c);
        const f32x4 ty = *reinterpret_cast<const f32x4*>(type + c);
        *reinterpret_cast<f32x4*>(out + row * D + c) = (w + ty) + p;       // HF: inputs_embeds + token_type_embeddings, then + position
    }
}
hipError_t launch_xlmr_embed(const int32_t* ids, int B, int L, int D, int pad_id, int vocab, int maxpos, const float* word, const float* pos,
                             const float* type, float* out, hipStream_t s) {
    if (D % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(xlmr_embed_kernel, dim3((unsigned)(((long)B * L + 3) / 4)), dim3(256), 0, s, ids, B, L, D, pad_id, vocab, maxpos, word, pos, type, out);
    return hipGetLastError();
}
// Implicit-LayerNorm token stream (GemmArgs::ln_mode, common.h): the same embedding sum, NOT normalised, as two fp16 planes
// (hi = fp16(x), lo = fp16(x - hi)) plus per row and 64-column block the (sum, sum of squares) that launch_ln_stats turns into the
// embedding LayerNorm's (mean, rstd).  One wave per token row; lane l holds columns 256 i + 4 l .. + 3: block = 4 i + l / 16.
__global__ __launch_bounds__(256) void xlmr_embed_planes_kernel(const int32_t* __restrict__ ids, int B, int L, int D, int pad_id, int vocab, int maxpos,
                                                                const float* __restrict__ word, const float* __restrict__ pos,
                                                                const float* __restrict__ type, f16* __restrict__ hi, f16* __restrict__ lo,
                                                                float* __restrict__ part) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)B * L) return;
    const int b = (int)(row / L), t = (int)(row - (long)b * L);
    int cnt = 0;
    for (int j = lane; j <= t; j += 64) cnt += ids[(long)b * L + j] != pad_id ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    int id = ids[row];
    int pid = id != pad_id ? pad_id + cnt : pad_id;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    pid = pid >= maxpos ? maxpos - 1 : pid;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(word + (long)id * D + c);
        const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (long)pid * D + c);
        const f32x4 ty = *reinterpret_cast<const f32x4*>(type + c);
        const f32x4 v = (w + ty) + p;
        const f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
        const f16x4 l = {(f16)(v.x - (float)h.x), (f16)(v.y - (float)h.y), (f16)(v.z - (float)h.z), (f16)(v.w - (float)h.w)};
        *reinterpret_cast<f16x4*>(hi + row * D + c) = h;
        *reinterpret_cast<f16x4*>(lo + row * D + c) = l;
        float s1 = (v.x + v.y) + (v.z + v.w), s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            s1 += __shfl_xor(s1, d, 64);
            s2 += __shfl_xor(s2, d, 64);
        }
        if ((lane & 15) == 0) *reinterpret_cast<f32x2_t*>(part + 2 * (row * (D >> 6) + (c >> 6))) = f32x2_t{s1, s2};
    }
}
hipError_t launch_xlmr_embed_planes(const int32_t* ids, int B, int L, int D, int pad_id, int vocab, int maxpos, const float* word, const float* pos,
                                    const float* type, f16* hi, f16* lo, float* part, hipStream_t s) {
    if (D % 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(xlmr_embed_planes_kernel, dim3((unsigned)(((long)B * L + 3) / 4)), dim3(256), 0, s, ids, B, L, D, pad_id, vocab, maxpos, word, pos, type,
                       hi, lo, part);
    return hipGetLastError();
}
// (mean, rstd) of every row from its per-block partial sums, accumulated in double: var = E[x^2] - mean^2 (biased, nn.LayerNorm), eps 1e-5
__global__ void ln_stats_kernel(const float* __restrict__ part, int rows, int P, float* __restrict__ stats) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s1 = 0.0, s2 = 0.0;
    for (int p = 0; p < P; ++p) {
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(part + 2 * ((long)r * P + p));
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    const double inv_d = 1.0 / (double)(P * 64), mean = s1 * inv_d;
    double var = s2 * inv_d - mean * mean;
    var = var > 0.0 ? var : 0.0;
    *reinterpret_cast<f32x2_t*>(stats + 2 * (long)r) = f32x2_t{(float)mean, 1.f / sqrtf((float)var + 1e-5f)};
}
hipError_t launch_ln_stats(const float* part, int rows, int P, float* stats, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, rows, P, stats);
    return hipGetLastError();
}
// out32 = nn.LayerNorm(hi + lo): the explicit LayerNorm that ends an implicit chain (two-pass statistics like layernorm_kernel)
template <int V>
__global__ void layernorm_planes_kernel(const f16* __restrict__ hi, const f16* __restrict__ lo, const float* __restrict__ w, const float* __restrict__ b,
                                        int rows, float* __restrict__ out32) {
    constexpr int D = 256 * V;
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[V];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const f16x4 h = *reinterpret_cast<const f16x4*>(hi + row * D + (i * 64 + lane) * 4), l = *reinterpret_cast<const f16x4*>(lo + row * D + (i * 64 + lane) * 4);
        v[i] = f32x4{(float)h.x + (float)l.x, (float)h.y + (float)l.y, (float)h.z + (float)l.z, (float)h.w + (float)l.w};
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        v[i] -= mean;
        sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float inv = 1.f / sqrtf(wave_sum(sq) * (1.f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int col = (i * 64 + lane) * 4;
        *reinterpret_cast<f32x4*>(out32 + row * D + col) = v[i] * inv * *reinterpret_cast<const f32x4*>(w + col) + *reinterpret_cast<const f32x4*>(b + col);
    }
}
hipError_t launch_layernorm_planes(const f16* hi, const f16* lo, const float* w, const float* b, int rows, int D, float* out32, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (D != 768) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_planes_kernel<3>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, hi, lo, w, b, rows, out32);
    return hipGetLastError();
}
// exact GELU (erf form, the HF "gelu"): fp32 in -> fp16 out
__global__ void gelu_kernel(const float* __restrict__ in, f16* __restrict__ out, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + i * 4);
        auto g = [](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); };
        const f16x4 h = {(f16)g(v.x), (f16)g(v.y), (f16)g(v.z), (f16)g(v.w)};
        *reinterpret_cast<f16x4*>(out + i * 4) = h;
    }
}
hipError_t launch_gelu(const float* in, f16* out, long n, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    const long n4 = n / 4;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536)), dim3(256), 0, s, in, out, n4);
    return hipGetLastError();
}
// int attention mask (1 = token, 0 = padding) -> the float key mask the attention kernels take
__global__ void mask_i32_f32_kernel(const int32_t* __restrict__ in, float* __restrict__ out, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] != 0 ? 1.f : 0.f;
}
hipError_t launch_mask_i32_f32(const int32_t* in, float* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(mask_i32_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

// out[pixel][c] = v[c]: an image whose every pixel is the same channel vector (the all-constant input of the const chain)
__global__ void broadcast_channels_kernel(const f16* __restrict__ v, int C, f16* __restrict__ out, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) out[i] = v[i % C];
}
hipError_t launch_broadcast_channels(const f16* v, int C, f16* out, long pixels, hipStream_t s) {
    const long total = pixels * C;
    hipLaunchKernelGGL(broadcast_channels_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, v, C, out, total);
    return hipGetLastError();
}

hipError_t launch_fill_f16(f16* p, long n, hipStream_t s) {
    return hipMemsetAsync(p, 0, n * sizeof(f16), s);
}

// Temporal mean of ragged (rows,D) blocks: out[i] = mean(x[offsets[i]:offsets[i+1]])
// (evaluate_retrieval.py:30-31, evaluate_asd.py:31,35).  One wave per (clip, 64-col group).
__global__ void ragged_mean_kernel(const float* __restrict__ x, const int32_t* __restrict__ off, int n, int D, float* __restrict__ out) {
    const int i = blockIdx.x;
    const int s0 = off[i], s1 = off[i + 1];
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float acc = 0.f;
        for (int r = s0; r < s1; ++r) acc += x[(long)r * D + c];
        out[(long)i * D + c] = acc / (float)(s1 - s0);
    }
}

hipError_t launch_ragged_mean(const float* x, const int32_t* offsets, int n, int D, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ragged_mean_kernel, dim3(n), dim3(256), 0, s, x, offsets, n, D, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Column sums of a row-major fp16 matrix (calibration pass of the bias-corrected precision mode):
// out[k] += sum_m A[m][k].  8 columns per thread, rows strided over blockIdx.y, one float atomic per
// (thread, column) at the end.  `out` is zeroed by the caller.
// Deterministic (no atomics): 64 row-strided partial sums per column go to `part` [gy][K], then one thread per column
// adds them to out[] in a fixed order.  out ACCUMULATES across calls (stream order), so a calibration batch may arrive
// in chunks and two handles calibrated on the same data end up with bit-identical bias corrections.
// stats ([M][2] mean, rstd; optional): the rows are normalised first -- what a Linear behind an IMPLICIT LayerNorm effectively sees.
__global__ void col_sum_kernel(const f16* __restrict__ A, long lda, int M, int K, float* __restrict__ part, const float* __restrict__ stats) {
    const int c8 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c8 * 8 >= K) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = blockIdx.y; m < M; m += gridDim.y) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(A + (long)m * lda + c8 * 8);
        const float mu = stats ? stats[2 * (long)m] : 0.f, rs = stats ? stats[2 * (long)m + 1] : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += ((float)v[e] - mu) * rs;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[(long)blockIdx.y * K + c8 * 8 + e] = acc[e];
}
__global__ void col_sum_finish_kernel(const float* __restrict__ part, int gy, int K, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int y = 0; y < gy; ++y) s += part[(long)y * K + k];
    out[k] += s;
}

size_t col_sum_scratch_elems(int K) { return (size_t)64 * K; }

hipError_t launch_col_sum(const f16* A, long lda, int M, int K, float* scratch, float* out, hipStream_t s, const float* stats) {
    if (M <= 0 || K <= 0) return hipSuccess;
    const int cols = K / 8;
    const int gy = M < 64 ? M : 64;
    hipLaunchKernelGGL(col_sum_kernel, dim3((cols + 63) / 64, gy), dim3(64), 0, s, A, lda, M, K, scratch, stats);
    hipLaunchKernelGGL(col_sum_finish_kernel, dim3((K + 255) / 256), dim3(256), 0, s, scratch, gy, K, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Run-time weight-rounding correction (precision mode JG_PREC_FP16_RC).  A Linear that runs on single fp16
// weights wh = fp16(w) loses (w - wh) . x per output; its systematic part is lo . E[x], lo = fp16(w - wh).  JG_PREC_FP16_BC takes
// E[x] from a calibration pass; here it comes from the rows of the clip AT HAND, per GEMM call, so nothing depends on calibration
// data or on the other clips of the batch:
//     bias_clip[c][n] = bias[n] + sum_k lo[n][k] * mean_c[k],   mean_c = mean of a fixed sample of clip c's rows of A
// (rows r = 0 .. rpc-1 relative to the clip's first row; sample: every row when rpc < 1024, otherwise the 16-row runs
// (r >> 4) % 8 == 0 -- the error of a sampled mean is the row spread / sqrt(rows sampled), a few per cent of what the correction
// removes).  Two launches: column means per clip (one workgroup per clip and 128-column slab, fixed summation order), then the skinny
// product on the matrix cores.  Deterministic.
__device__ __forceinline__ int rc_rows_sampled(int rpc) {
    if (rpc < 1024) return rpc;
    int cnt = 0;
    for (int r0 = 0; r0 < rpc; r0 += 128) cnt += rpc - r0 < 16 ? rpc - r0 : 16;
    return cnt;
}
// valid (optional, device [nclips]): clip c's first valid[c] rows are its own (a batch padded to a common length: the rest is padding that
// must not enter the clip's statistics); the sample is then defined on the clip's OWN row count, exactly as if it were alone.
__global__ __launch_bounds__(256) void rc_col_mean_kernel(const f16* __restrict__ A, long lda, int tiled, int rpc_all, const int* __restrict__ valid, int K,
                                                          f16* __restrict__ mean) {
    __shared__ float red[32][8 * 8 + 1];
    const int clip = blockIdx.x, t = threadIdx.x;
    const int cg = t & 7, rl = t >> 3;                          // 8 column groups of 8 (a 64-column slab) x 32 row lanes
    const int n = blockIdx.y * 64 + cg * 8;
    int rpc = rpc_all;                                          // rows of this clip that count
    if (valid) rpc = valid[clip] < 1 ? 1 : (valid[clip] < rpc_all ? valid[clip] : rpc_all);
    const int step = rpc < 1024 ? 16 : 128;                     // 16-row runs: every one, or every eighth
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto row_ptr = [&](int r) -> const f16* {
        const long m = (long)clip * rpc_all + r;
        return tiled ? A + (m >> 7) * 65536 + (long)(n >> 6) * 8192 + ((m & 127) >> 4) * 1024 + ((n & 63) >> 4) * 256 + (m & 15) * 16 + (n & 15)
                     : A + m * lda + n;
    };
    // row lane rl takes row (rl & 15) of every second sampled run (runs of its parity rl >> 4); EIGHT loads in flight per thread (the
    // rows were just written with nontemporal stores: every load is an HBM round trip, and a 150-frame clip gives a thread 13 rows),
    // summed in a fixed order
    const int stride = 2 * step;
    int r = (rl & 15) + (rl >> 4) * step;
    for (; r + 7 * stride < rpc; r += 8 * stride) {
        f16x8 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f16x8*>(row_ptr(r + u * stride));
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
    }
    {   // the tail: up to seven rows, again all in flight
        f16x8 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int ru = r + u * stride;
            v[u] = *reinterpret_cast<const f16x8*>(row_ptr(ru < rpc ? ru : 0));
        }
#pragma unroll
        for (int u = 0; u < 7; ++u)
            if (r + u * stride < rpc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    if (t < 64) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) s += red[q][t];
        mean[(long)clip * K + blockIdx.y * 64 + t] = (f16)(s / (float)rc_rows_sampled(rpc));
    }
}
// out[c][n] = bias[n] + sum_k lo[n][k] * mean[c][k]: a 32-clip x 32-column tile per workgroup on v_mfma_f32_32x32x16 (A = the lo rows,
// B = the clip means as fp16 -- the mean's own rounding, 2^-12, scales a term that is 3e-4 of the output), the four waves split K and
// meet in LDS.  (The VALU form of this product -- a wave-wide reduction per clip and column -- took 17 us per Linear, twice the
// column means; rocprofv3, round 5.)
__global__ __launch_bounds__(256) void rc_gemv_kernel(const f16* __restrict__ mean16, const f16* __restrict__ lo, const float* __restrict__ bias,
                                                      int nclips, int N, int K, float* __restrict__ out) {
    __shared__ float red[3][16][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int cl = c0 + r31 < nclips ? c0 + r31 : nclips - 1;
    const int kq = K >> 2;                                       // this wave's share of K
    const f16* ap = lo + (long)(n0 + r31) * K + wave * kq + 8 * hh;
    const f16* bp = mean16 + (long)cl * K + wave * kq + 8 * hh;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = 0; k < kq; k += 128) {           // kq is 128 or 512: eight k-steps' operands in flight at a time
        f16x8 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = *reinterpret_cast<const f16x8*>(ap + k + 16 * u);
            b[u] = *reinterpret_cast<const f16x8*>(bp + k + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = JG_MFMA_32x32x16(a[u], b[u], acc);
    }
    if (wave) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0 && c0 + r31 < nclips) {
        // register i <-> column n0 + (i & 3) + 8 (i >> 2) + 4 hh, lane <-> clip
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = n0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            out[(long)(c0 + r31) * N + n] = (bias ? bias[n] : 0.f) + (((acc[i] + red[0][i][lane]) + red[1][i][lane]) + red[2][i][lane]);
        }
    }
}

size_t rc_scratch_elems(int nclips, int K) { return (size_t)nclips * K; }

hipError_t launch_rc_bias(const f16* A, long lda, int tiled, int nclips, int rpc, const int* valid_rows, const f16* lo, const float* bias, int N, int K,
                          float* scratch, float* out, hipStream_t s) {
    if (nclips <= 0 || rpc <= 0) return hipSuccess;
    if ((K != 512 && K != 2048) || (tiled && K != 512) || (N & 31)) return hipErrorInvalidValue;
    f16* mean16 = reinterpret_cast<f16*>(scratch);
    hipLaunchKernelGGL(rc_col_mean_kernel, dim3(nclips, K / 64), dim3(256), 0, s, A, lda, tiled, rpc, valid_rows, K, mean16);
    hipLaunchKernelGGL(rc_gemv_kernel, dim3(N / 32, (nclips + 31) / 32), dim3(256), 0, s, mean16, lo, bias, nclips, N, K, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Log-mel front-end (utils/audio_utils.py:28-66): torch.stft(n_fft 512, hop 160, hann(320) centred in the
// 512 window, center=True / reflect padding, onesided) -> drop the last frame -> |X| -> mel_basis (80x257) ->
// log(. + 1e-20).  One block per frame: the windowed 512-sample segment and a 512-entry twiddle table live in
// LDS, thread f computes bin f (and f+256) by direct DFT in fp32 (316 MFLOP per 150-frame clip: not worth an FFT),
// then threads 0..79 do the mel dot products.
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, int n_samples, int n_frames,
                                                     const float* __restrict__ mel_basis, float* __restrict__ out) {
    __shared__ float seg[512];
    __shared__ float tc[512], ts[512];
    __shared__ float mag[257];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* x = wav + (long)b * n_samples;
    for (int n = tid; n < 512; n += 256) {
        float w = 0.f;
        if (n >= 96 && n < 416) w = 0.5f - 0.5f * cosf(6.283185307179586f * (float)(n - 96) / 320.f);
        int i = t * 160 + n - 256;
        if (i < 0) i = -i;
        if (i >= n_samples) i = 2 * (n_samples - 1) - i;
        i = i < 0 ? 0 : i;
        seg[n] = w * x[i];
        float sn, cs;
        sincosf(6.283185307179586f * (float)n / 512.f, &sn, &cs);
        tc[n] = cs;
        ts[n] = sn;
    }
    __syncthreads();
    for (int f = tid; f < 257; f += 256) {
        float re = 0.f, im = 0.f;
        int ph = 0;
        for (int n = 0; n < 512; ++n) {
            re += seg[n] * tc[ph];
            im -= seg[n] * ts[ph];
            ph = (ph + f) & 511;
        }
        mag[f] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    if (tid < 80) {
        const float* mb = mel_basis + tid * 257;
        float acc = 0.f;
        for (int f = 0; f < 257; ++f) acc += mb[f] * mag[f];
        out[((long)b * n_frames + t) * 80 + tid] = logf(acc + 1e-20f);
    }
}

hipError_t launch_logmel(const float* wav, int B, int n_samples, const float* mel_basis, float* out, hipStream_t s) {
    const int n_frames = n_samples / 160;      // 1 + n/160 STFT frames, last one dropped (audio_utils.py:46)
    if (B <= 0 || n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(logmel_kernel, dim3(n_frames, B), dim3(256), 0, s, wav, n_samples, n_frames, mel_basis, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Face-mask + resize pre-step (inference_embs.py:235-276): per frame
//   face found : cv2.rectangle(img,(0,0),(W,y2+15),0,-1) at SOURCE resolution, then cv2.resize(img,(480,270))
//   face None  : cv2.resize first, then cv2.rectangle(img,(0,0),(480,110),0,-1)
// as ONE uint8 -> uint8 kernel (mask_y[f] >= 0: last blanked source row; -1: the face-None case).
// cv2.resize default = INTER_LINEAR on 8-bit: restated from OpenCV's generic fixed-point path
// (modules/imgproc/src/resize.cpp: coefficients cvRound(w*2048) as short, horizontal pass in int,
// vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2).  cv2 is not installed here and the
// pip wheels may dispatch to IPP: PARITY UNPINNED (oracle/jegal_oracle.py:mask_resize_frames is the same
// restatement in numpy).
// Packed source (offs != nullptr): the producer ships only the source rows BELOW each frame's mask -- frame f's rows
// row0 = max(mask_y[f] + 1, 0) .. H-1 start at src + offs[f]; the blanked rows are never read here, so they need not exist.
// A frame whose kept rows would end beyond `src_bytes` (bad metadata) is written as zeros instead of being read.
__global__ void mask_resize_kernel(const uint8_t* __restrict__ src, int T, int H, int W, const int* __restrict__ mask_y,
                                   uint8_t* __restrict__ dst, const long long* __restrict__ offs, long long src_bytes) {
    constexpr int OH = 270, OW = 480;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)T * OH * OW) return;
    const int dx = (int)(idx % OW);
    const int dy = (int)((idx / OW) % OH);
    const int f = (int)(idx / ((long)OW * OH));
    const int my = mask_y[f];
    uint8_t* o = dst + idx * 3;
    if (my < 0 && dy <= 110) {                 // face None: rows 0..110 of the RESIZED frame (rectangle corners inclusive)
        o[0] = o[1] = o[2] = 0;
        return;
    }
    const double scale_x = (double)W / OW, scale_y = (double)H / OH;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx + 1 >= W) { fx = 0.f; sx = W - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    auto sat_short = [](float v) -> int {
        int r = __float2int_rn(v);
        return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
    };
    const int a0 = sat_short((1.f - fx) * 2048.f), a1 = sat_short(fx * 2048.f);
    const int b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
    const int y0 = sy < 0 ? 0 : (sy < H ? sy : H - 1);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 < H ? sy + 1 : H - 1);
    const int x1 = sx + 1 < W ? sx + 1 : W - 1;
    const uint8_t* fr = src + (long)f * H * W * 3;
    bool bad = false;
    if (offs) {
        const int row0 = my >= 0 ? (my + 1 < H ? my + 1 : H) : 0;
        const long long o = offs[f];
        bad = o < 0 || o + (long long)(H - row0) * W * 3 > src_bytes;
        fr = src + o - (long long)row0 * W * 3;
    }
    const bool z0 = bad || (my >= 0 && y0 <= my), z1 = bad || (my >= 0 && y1 <= my);       // blanked source rows
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int p00 = z0 ? 0 : fr[((long)y0 * W + sx) * 3 + c], p01 = z0 ? 0 : fr[((long)y0 * W + x1) * 3 + c];
        const int p10 = z1 ? 0 : fr[((long)y1 * W + sx) * 3 + c], p11 = z1 ? 0 : fr[((long)y1 * W + x1) * 3 + c];
        const int S0 = p00 * a0 + p01 * a1, S1 = p10 * a0 + p11 * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// The same arithmetic, banded (round 4: the streamed source-resolution upload runs this kernel on the upload stream next to the
// compute, so its cost is CU time taken from the extraction): one workgroup = RB output rows of one frame.  The source rows the
// band touches are one contiguous byte span of the frame: staged into LDS with dword loads (the span's start is aligned down, a
// tail of < 4 bytes is loaded bytewise, nothing outside [src, src + src_bytes) is touched), the per-column (sx, a1) and per-row
// (y0, y1, b1) coefficients are computed once per workgroup -- in double / float exactly as in mask_resize_kernel -- and the
// output rows leave through LDS as whole 16-byte pieces (a row is 1440 B).  Results are bit-identical to mask_resize_kernel
// (tests/test_gpu_drivers.py::test_mask_resize_matches_oracle runs both).
constexpr int MR_RB = 6;                       // output rows per workgroup (270 = 45 bands)
constexpr int MR_SPAN_MAX = 40 * 1024;         // LDS bytes for the source span; larger sources take the per-pixel kernel
__global__ __launch_bounds__(256) void mask_resize_band_kernel(const uint8_t* __restrict__ src, int T, int H, int W, const int* __restrict__ mask_y,
                                                               uint8_t* __restrict__ dst, const long long* __restrict__ offs, long long src_bytes) {
    constexpr int OH = 270, OW = 480, RB = MR_RB;
    extern __shared__ __attribute__((aligned(16))) uint8_t mr_smem[];          // [orow RB*1440][span: sized by the launcher]
    uint8_t* const orow = mr_smem;
    uint8_t* const span = mr_smem + RB * OW * 3;
    __shared__ short cx_s[OW], cx_a[OW];       // source column, a1 (a0 from the same float, see below)
    __shared__ short cx_a0[OW];
    __shared__ int ry0[RB], ry1[RB], rb0[RB], rb1[RB];
    const int tid = threadIdx.x;
    const int f = blockIdx.x / (OH / RB), band = blockIdx.x - f * (OH / RB);
    const int dy0 = band * RB;
    const int my = mask_y[f];
    uint4* out16 = reinterpret_cast<uint4*>(dst + ((long)f * OH + dy0) * (OW * 3));
    constexpr int OUT_V = RB * OW * 3 / 16;    // 540 16-byte pieces
    auto sat_short = [](float v) -> int {
        int r = __float2int_rn(v);
        return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
    };
    const double scale_x = (double)W / OW, scale_y = (double)H / OH;
    // per-row coefficients (every thread computes the band's first / last source row itself: wave-uniform, no barrier needed for them)
    auto row_coef = [&](int dy, int& y0, int& y1, int& b0, int& b1) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= sy;
        b0 = sat_short((1.f - fy) * 2048.f);
        b1 = sat_short(fy * 2048.f);
        y0 = sy < 0 ? 0 : (sy < H ? sy : H - 1);
        y1 = sy + 1 < 0 ? 0 : (sy + 1 < H ? sy + 1 : H - 1);
    };
    int ylo, yhi, t0, t1, t2;
    row_coef(dy0, ylo, t0, t1, t2);
    row_coef(dy0 + RB - 1, t0, yhi, t1, t2);
    // rows the band READS: those below the mask (face found) or all of them (face None: the blank rows are cut AFTER the resize)
    const int row0 = my >= 0 ? (my + 1 < H ? my + 1 : H) : 0;
    const int rlo = ylo > row0 ? ylo : row0;
    const uint8_t* fr = src + (long)f * H * W * 3;
    bool bad = false;
    if (offs) {
        const long long o = offs[f];
        bad = o < 0 || o + (long long)(H - row0) * W * 3 > src_bytes;
        fr = src + o - (long long)row0 * W * 3;
    }
    const bool none = bad || rlo > yhi || (my < 0 && dy0 + RB - 1 <= 110);      // nothing of the source reaches this band
    if (none) {
        for (int i = tid; i < OUT_V; i += 256) out16[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    // ---- stage the span [rlo, yhi] of the frame
    const uint8_t* p_lo = fr + (long)rlo * W * 3;
    const long nbytes = (long)(yhi - rlo + 1) * W * 3;
    const int shift = (int)((uintptr_t)p_lo & 3);
    const uint8_t* p_al = p_lo - shift;                        // >= the allocation's start: allocations are at least 4-byte aligned
    const long nfull = (shift + nbytes) >> 2;                  // whole dwords inside [p_al, p_lo + nbytes)
    for (long i = tid; i < nfull; i += 256) reinterpret_cast<uint32_t*>(span)[i] = reinterpret_cast<const uint32_t*>(p_al)[i];
    for (long i = nfull * 4 + tid; i < shift + nbytes; i += 256) span[i] = p_al[i];
    for (int dx = tid; dx < OW; dx += 256) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx + 1 >= W) { fx = 0.f; sx = W - 1; }
        cx_s[dx] = (short)sx;
        cx_a0[dx] = (short)sat_short((1.f - fx) * 2048.f);
        cx_a[dx] = (short)sat_short(fx * 2048.f);
    }
    if (tid < RB) row_coef(dy0 + tid, ry0[tid], ry1[tid], rb0[tid], rb1[tid]);
    __syncthreads();
    const int rowb = W * 3;
    for (int i = tid; i < RB * OW; i += 256) {
        const int r = i / OW, dx = i - r * OW;
        uint8_t* o = orow + i * 3;
        if (my < 0 && dy0 + r <= 110) {                       // face None: rows 0..110 of the RESIZED frame
            o[0] = o[1] = o[2] = 0;
            continue;
        }
        const int y0 = ry0[r], y1 = ry1[r], b0 = rb0[r], b1 = rb1[r];
        const int sx = cx_s[dx], a0 = cx_a0[dx], a1 = cx_a[dx];
        const int x1 = sx + 1 < W ? sx + 1 : W - 1;
        const bool z0 = my >= 0 && y0 <= my, z1 = my >= 0 && y1 <= my;
        const uint8_t* q0 = span + shift + (long)(y0 - rlo) * rowb;
        const uint8_t* q1 = span + shift + (long)(y1 - rlo) * rowb;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p00 = z0 ? 0 : q0[sx * 3 + c], p01 = z0 ? 0 : q0[x1 * 3 + c];
            const int p10 = z1 ? 0 : q1[sx * 3 + c], p11 = z1 ? 0 : q1[x1 * 3 + c];
            const int S0 = p00 * a0 + p01 * a1, S1 = p10 * a0 + p11 * a1;
            const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    __syncthreads();
    for (int i = tid; i < OUT_V; i += 256) out16[i] = reinterpret_cast<const uint4*>(orow)[i];
}

hipError_t launch_mask_resize(const uint8_t* src, int T, int H, int W, const int* mask_y_dev, uint8_t* dst, hipStream_t s,
                              const long long* offs, long long src_bytes) {
    const long n = (long)T * 270 * 480;
    if (n <= 0) return hipSuccess;
    // source rows a band of MR_RB output rows can touch: (MR_RB - 1) * H / 270 + 3
    const long span_rows = (long)(MR_RB - 1) * H / 270 + 3;
    static const bool generic = getenv("JG_MASK_RESIZE_GENERIC") != nullptr;       // A/B and test switch
    // (the banded kernel stages dword-aligned spans: it may read up to 3 bytes in front of a row that does not start on a dword, which
    // is inside the buffer as long as the buffer itself starts on one)
    if (span_rows * W * 3 + 4 <= MR_SPAN_MAX && W <= 32767 && !generic && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
        const size_t lds = (size_t)MR_RB * 480 * 3 + (((size_t)span_rows * W * 3 + 4 + 15) & ~(size_t)15);
        hipLaunchKernelGGL(mask_resize_band_kernel, dim3((unsigned)(T * (270 / MR_RB))), dim3(256), lds, s, src, T, H, W, mask_y_dev, dst, offs, src_bytes);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(mask_resize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, T, H, W, mask_y_dev, dst, offs, src_bytes);
    return hipGetLastError();
}

// ---- masked crops over PCIe in fewer bytes ------------------------------------------------------------------
// The reference blanks rows 0 .. y2+15 of every crop (inference_embs.py:264-270): ~40 % of the bytes of a batch are zeros the
// engine then skips.  A producer ships only the rows BELOW each frame's mask, packed back to back; this kernel rebuilds the
// dense (frames, 270, 480, 3) batch: frame f's rows >= row0[f] come from packed + offs[f], the rows above are zero.
// Bad metadata (row0 outside 0..270, an offset that is negative, not a multiple of 16, or whose rows end beyond packed_bytes) makes
// the frame come out all zero instead of reading out of bounds / misaligned (packed_bytes < 0: unknown size, offsets are trusted).
__global__ __launch_bounds__(256) void unpack_masked_kernel(const uint8_t* __restrict__ packed, const int* __restrict__ row0,
                                                            const long long* __restrict__ offs, uint8_t* __restrict__ dst, long long packed_bytes) {
    constexpr int ROW_B = 480 * 3, ROW_V = ROW_B / 16, ROWS = 30;          // 90 16-byte pieces per row, 30 rows per block (9 blocks per frame)
    const int f = blockIdx.x;
    const int r_begin = blockIdx.y * ROWS;
    int r0 = row0[f];
    const long long of = offs[f];
    if (r0 < 0 || r0 > 270 || of < 0 || (of & 15) || (packed_bytes >= 0 && of + (long long)(270 - r0) * ROW_B > packed_bytes)) r0 = 270;
    const uint8_t* src = packed + of;
    uint8_t* out = dst + (size_t)f * (270 * ROW_B);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    for (int i = threadIdx.x; i < ROWS * ROW_V; i += 256) {
        const int r = r_begin + i / ROW_V, c = i % ROW_V;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r >= r0) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + (size_t)(r - r0) * ROW_B) + c);
        reinterpret_cast<u32x4*>(out + (size_t)r * ROW_B)[c] = v;
    }
}

hipError_t launch_unpack_masked(const uint8_t* packed, const int* row0, const long long* offs, int n_frames, uint8_t* dst, hipStream_t s,
                                long long packed_bytes) {
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_masked_kernel, dim3((unsigned)n_frames, 9), dim3(256), 0, s, packed, row0, offs, dst, packed_bytes);
    return hipGetLastError();
}

JG_NS_END
