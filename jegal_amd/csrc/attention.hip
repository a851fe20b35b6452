// Self-attention for short sequences (gfx950).  softmax(q k^T / sqrt(dk) [masked_fill(mask==0,-1e9)]) v
//   gestsync.py:20-21 (nn.MultiheadAttention, S = 21, no mask), modules.py:61-75 (S = T <= 500 or
//   L text tokens, key-padding mask).
//
// Attention is < 5 % of the path's FLOPs (S = 21: 0.9 of 133 MFLOP per window-layer), so this is a
// VALU kernel organised for the LDS/VALU pipes rather than MFMA: one lane per query row, the head's
// K/V rows staged in LDS as fp16 and read as wave-wide BROADCASTS (every lane of a sequence reads the
// same 16 B), packed v_dot2_f32_f16 for q.k, online softmax over sub-blocks of 8 keys, fp32 state.
// For S <= 32 one 64-lane wave carries floor(64/S) (sequence, head) pairs (3 windows x 21 rows).
#include "common.h"

template <int DK>
__global__ __launch_bounds__(256) void attn_kernel(const f16* __restrict__ qkv, const float* __restrict__ keymask,
                                                   int B, int S, int H, int G, f16* __restrict__ out) {
    constexpr int NV = DK / 8;            // 16-byte vectors per head row
    __shared__ __attribute__((aligned(16))) f16 sK[64 * DK];
    __shared__ __attribute__((aligned(16))) f16 sV[64 * DK];
    __shared__ float sM[64];

    const int D = H * DK;
    const long ld = 3L * D;
    const int tid = threadIdx.x;
    const bool small = G > 1 || blockDim.x == 64;
    const long npairs = (long)B * H;

    long pair;            // (b,h) of this lane's query
    int qi;               // query index inside the sequence
    int kbase;            // first LDS row of this lane's keys (small mode)
    bool active;
    if (small) {
        const int g = tid / S;
        pair = (long)blockIdx.x * G + g;
        qi = tid - g * S;
        kbase = g * S;
        active = g < G && pair < npairs;
    } else {
        pair = blockIdx.x;
        qi = blockIdx.y * blockDim.x + tid;
        kbase = 0;
        active = qi < S;
    }
    const int b = active ? (int)(pair / H) : 0, h = active ? (int)(pair % H) : 0;

    f16x2 q[DK / 2];
    float o[DK];
#pragma unroll
    for (int d = 0; d < DK; ++d) o[d] = 0.f;
    if (active) {
        const f16* qp = qkv + ((long)b * S + qi) * ld + h * DK;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f16x8 t8 = *reinterpret_cast<const f16x8*>(qp + v * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) q[v * 4 + e] = f16x2{t8[2 * e], t8[2 * e + 1]};
        }
    } else {
#pragma unroll
        for (int d = 0; d < DK / 2; ++d) q[d] = f16x2{(f16)0.f, (f16)0.f};
    }
    const float scale = 1.0f / sqrtf((float)DK);
    float mrun = -INFINITY, lrun = 0.f;

    const int nchunks = small ? 1 : (S + 63) / 64;
    for (int ch = 0; ch < nchunks; ++ch) {
        // ---- stage K/V rows (and their mask) of this chunk
        const int rows = small ? G * S : min(64, S - ch * 64);
        __syncthreads();
        for (int idx = tid; idx < rows * NV; idx += blockDim.x) {
            const int r = idx / NV, v = idx - r * NV;
            long p2;
            int j;
            if (small) { const int g = r / S; p2 = (long)blockIdx.x * G + g; j = r - g * S; }
            else { p2 = pair; j = ch * 64 + r; }
            uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
            if (p2 < npairs) {
                const int b2 = (int)(p2 / H), h2 = (int)(p2 % H);
                const f16* base = qkv + ((long)b2 * S + j) * ld + h2 * DK + v * 8;
                kk = *reinterpret_cast<const uint4*>(base + D);
                vv = *reinterpret_cast<const uint4*>(base + 2 * D);
            }
            *reinterpret_cast<uint4*>(&sK[r * DK + v * 8]) = kk;
            *reinterpret_cast<uint4*>(&sV[r * DK + v * 8]) = vv;
        }
        for (int r = tid; r < rows; r += blockDim.x) {
            long p2;
            int j;
            if (small) { const int g = r / S; p2 = (long)blockIdx.x * G + g; j = r - g * S; }
            else { p2 = pair; j = ch * 64 + r; }
            float mk = 1.f;
            if (keymask && p2 < npairs) mk = keymask[(p2 / H) * S + j];
            sM[r] = mk;
        }
        __syncthreads();

        const int nkeys = small ? S : rows;
        for (int j0 = 0; j0 < nkeys; j0 += 8) {
            float sc[8];
            float bmax = -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                float acc = 0.f;
                if (j < nkeys) {
                    const f16* kr = &sK[(kbase + j) * DK];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const f16x8 k8 = *reinterpret_cast<const f16x8*>(kr + v * 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_fdot2(q[v * 4 + e], f16x2{k8[2 * e], k8[2 * e + 1]}, acc, false);
                    }
                    acc *= scale;
                    if (sM[kbase + j] == 0.f) acc = -1e9f;
                } else {
                    acc = -INFINITY;
                }
                sc[u] = acc;
                bmax = fmaxf(bmax, acc);
            }
            const float mnew = fmaxf(mrun, bmax);
            const float alpha = __expf(mrun - mnew);     // first block: exp(-inf) = 0
            lrun *= alpha;
#pragma unroll
            for (int d = 0; d < DK; ++d) o[d] *= alpha;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j < nkeys) {
                    const float p = __expf(sc[u] - mnew);
                    lrun += p;
                    const f16* vr = &sV[(kbase + j) * DK];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const f16x8 v8 = *reinterpret_cast<const f16x8*>(vr + v * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[v * 8 + e] += p * (float)v8[e];
                    }
                }
            }
            mrun = mnew;
        }
    }
    if (active) {
        const float inv = 1.f / lrun;
        f16* op = out + ((long)b * S + qi) * D + h * DK;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            f16x8 t8;
#pragma unroll
            for (int e = 0; e < 8; ++e) t8[e] = (f16)(o[v * 8 + e] * inv);
            *reinterpret_cast<f16x8*>(op + v * 8) = t8;
        }
    }
}

hipError_t launch_attention(const f16* qkv, const float* keymask, int B, int S, int H, int dk, f16* out, hipStream_t s) {
    if (B <= 0 || S <= 0) return hipSuccess;
    const long npairs = (long)B * H;
    dim3 grid, block;
    int G = 1;
    if (S <= 32) {
        G = 64 / S;
        grid = dim3((unsigned)((npairs + G - 1) / G));
        block = dim3(64);
    } else {
        grid = dim3((unsigned)npairs, (S + 255) / 256);
        block = dim3(256);
    }
    if (dk == 64) hipLaunchKernelGGL(attn_kernel<64>, grid, block, 0, s, qkv, keymask, B, S, H, G, out);
    else if (dk == 96) hipLaunchKernelGGL(attn_kernel<96>, grid, block, 0, s, qkv, keymask, B, S, H, G, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
