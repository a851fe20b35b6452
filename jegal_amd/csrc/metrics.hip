// Metric kernels (gfx950): retrieval rank counting on exact-fp32 MFMA, word spotting, ASD.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// Retrieval (evaluate_retrieval.py:38-65).  For each local query row i (global index gi =
// row_offset + i):  d = <e1[i], e2[gi]>,  rank[i] = #{j : <e1[i],e2[j]> > d},
// ties[i] = #{j : <e1[i],e2[j]> == d} (includes j = gi).  The reference's `ind` entries for the row
// are rank .. rank+ties-1.  The N x N similarity matrix is never written to HBM.
//
// v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fmaf chain per element, so an element's value
// depends only on its two vectors -- duplicate gallery rows give bitwise-equal scores, i.e. exact
// ties like the reference.  Block = 4 waves = 64 query rows x 64 gallery rows per tile (each wave a
// 32x32 accumulator), K staged through LDS in 64-wide chunks stored [k][row] so the one-float
// fragment reads (lane -> row l&31, k l>>5) are bank-conflict free.
__global__ __launch_bounds__(256) void sim_rank_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                       int n_local, int n_total, int row_offset, int D,
                                                       int32_t* __restrict__ rank, int32_t* __restrict__ ties) {
    __shared__ float sA[64][64];   // [k][query row]
    __shared__ float sB[64][64];   // [k][gallery row]
    __shared__ float sDiag[64];
    __shared__ int sRank[64], sTies[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    const int i0 = blockIdx.x * 64;
    const int r = tid & 63, kq = tid >> 6;
    if (tid < 64) { sRank[tid] = 0; sTies[tid] = 0; }

    int cnt_gt[16], cnt_eq[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) { cnt_gt[x] = 0; cnt_eq[x] = 0; }

    const int ntiles = (n_total + 63) / 64;
    // tile -1 is the diagonal tile (gallery rows row_offset+i0 ..), then all gallery tiles.
    for (int tile = -1; tile < ntiles; ++tile) {
        const int j0 = tile < 0 ? row_offset + i0 : tile * 64;
        f32x16 acc;
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[x] = 0.f;
        for (int k0 = 0; k0 < D; k0 += 64) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kq * 16 + q * 4;
                f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = va;
                if (i0 + r < n_local) va = *reinterpret_cast<const f32x4*>(e1 + (long)(i0 + r) * D + k0 + k);
                if (j0 + r < n_total) vb = *reinterpret_cast<const f32x4*>(e2 + (long)(j0 + r) * D + k0 + k);
                sA[k][r] = va.x; sA[k + 1][r] = va.y; sA[k + 2][r] = va.z; sA[k + 3][r] = va.w;
                sB[k][r] = vb.x; sB[k + 1][r] = vb.y; sB[k + 2][r] = vb.z; sB[k + 3][r] = vb.w;
            }
            __syncthreads();
#pragma unroll 8
            for (int kk = 0; kk < 64; kk += 2) {
                const float a = sA[kk + (lane >> 5)][wr * 32 + (lane & 31)];
                const float b = sB[kk + (lane >> 5)][wc * 32 + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        // D layout: col = lane&31 (gallery), row = (x&3) + 8*(x>>2) + 4*(lane>>5) (query)
        if (tile < 0) {
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int row = wr * 32 + (x & 3) + 8 * (x >> 2) + 4 * (lane >> 5);
                const int col = wc * 32 + (lane & 31);
                if (row == col) sDiag[row] = acc[x];
            }
            __syncthreads();
        } else {
            const int j = j0 + wc * 32 + (lane & 31);
            const bool jok = j < n_total;
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const int row = wr * 32 + (x & 3) + 8 * (x >> 2) + 4 * (lane >> 5);
                const float d = sDiag[row];
                cnt_gt[x] += (jok && acc[x] > d) ? 1 : 0;
                cnt_eq[x] += (jok && acc[x] == d) ? 1 : 0;
            }
        }
    }
    // reduce over the 32 lanes that share a row, then across the two column waves
#pragma unroll
    for (int x = 0; x < 16; ++x) {
        int g = cnt_gt[x], e = cnt_eq[x];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { g += __shfl_xor(g, o, 64); e += __shfl_xor(e, o, 64); }
        if ((lane & 31) == 0) {
            const int row = wr * 32 + (x & 3) + 8 * (x >> 2) + 4 * (lane >> 5);
            atomicAdd(&sRank[row], g);
            atomicAdd(&sTies[row], e);
        }
    }
    __syncthreads();
    if (tid < 64 && i0 + tid < n_local) {
        rank[i0 + tid] = sRank[tid];
        ties[i0 + tid] = sTies[tid];
    }
}

hipError_t launch_sim_rank(const float* e1, const float* e2, int n_local, int n_total, int row_offset, int D,
                           int32_t* rank, int32_t* ties, hipStream_t s) {
    if (n_local <= 0) return hipSuccess;
    if (D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sim_rank_kernel, dim3((n_local + 63) / 64), dim3(256), 0, s, e1, e2, n_local, n_total, row_offset, D, rank, ties);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Word spotting (evaluate_spotting.py:39-82): per clip A = softmax((G C^T)/temp, dim=1) over words
// with re-normalised rows; pred = first argmax_t A[t][w*], score = A[pred][w*].
// One block per clip, one wave per frame row; the W logits of a row go through a per-wave LDS line.
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Limits: W <= SPOT_MAX_W words and T <= SPOT_MAX_T frames per clip (LDS arrays), 0 <= target < W.  The offsets are device
// arrays, so the host cannot check them without a sync: a clip outside the limits gets pred = -1, score = NaN.
constexpr int SPOT_MAX_W = 1024, SPOT_MAX_T = 8192;

__global__ __launch_bounds__(256) void spot_kernel(const float* __restrict__ g, const float* __restrict__ c,
                                                   const int32_t* __restrict__ goff, const int32_t* __restrict__ coff,
                                                   const int32_t* __restrict__ target, int D, float temp,
                                                   int32_t* __restrict__ pred, float* __restrict__ score) {
    __shared__ float sCn[SPOT_MAX_W];        // 1/max(||c_w||, eps)
    __shared__ float sL[4][SPOT_MAX_W];      // per wave: the logits of the frame it is working on
    __shared__ float sA[SPOT_MAX_T];         // A[t][w*]
    const int clip = blockIdx.x;
    const int t0 = goff[clip], T = goff[clip + 1] - t0;
    const int w0 = coff[clip], W = coff[clip + 1] - w0;
    const int wt = target[clip];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (T <= 0 || T > SPOT_MAX_T || W <= 0 || W > SPOT_MAX_W || wt < 0 || wt >= W) {      // block-uniform
        if (threadIdx.x == 0) { pred[clip] = -1; score[clip] = __builtin_nanf(""); }
        return;
    }
    for (int w = wave; w < W; w += 4) {
        float sq = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = c[(long)(w0 + w) * D + d]; sq += v * v; }
        sq = wsum(sq);
        if (lane == 0) sCn[w] = 1.f / fmaxf(sqrtf(sq), 1e-12f);
    }
    __syncthreads();
    float* L = sL[wave];
    for (int t = wave; t < T; t += 4) {
        const float* gr = g + (long)(t0 + t) * D;
        float sq = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = gr[d]; sq += v * v; }
        const float gn = 1.f / fmaxf(sqrtf(wsum(sq)), 1e-12f);
        for (int w = 0; w < W; ++w) {
            const float* cr = c + (long)(w0 + w) * D;
            float dot = 0.f;
            for (int d = lane; d < D; d += 64) dot += (gr[d] * gn) * (cr[d] * sCn[w]);
            dot = wsum(dot) / temp;
            if (lane == 0) L[w] = dot;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float mx = -INFINITY;
        for (int w = lane; w < W; w += 64) mx = fmaxf(mx, L[w]);
        mx = wmax(mx);
        float e = 0.f;
        for (int w = lane; w < W; w += 64) e += expf(L[w] - mx);
        const float den = wsum(e);
        const float a = expf(L[wt] - mx) / den;
        if (lane == 0) sA[t] = a;
        __builtin_amdgcn_wave_barrier();         // L is rewritten for the next frame
    }
    __syncthreads();
    if (wave == 0) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int t = lane; t < T; t += 64) {
            const float a = sA[t];
            if (a > best) { best = a; bi = t; }      // strided scan keeps the first index per lane
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { pred[clip] = bi; score[clip] = best; }
    }
}

hipError_t launch_spot(const float* g, const float* c, const int32_t* goff, const int32_t* coff, const int32_t* target,
                       int n, int D, float temp, int32_t* pred, float* score, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(spot_kernel, dim3(n), dim3(256), 0, s, g, c, goff, coff, target, D, temp, pred, score);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// ASD (evaluate_asd.py:43-51,94-100): cosine(query content, candidate gestures) (eps 1e-8), softmax
// over the first P in {2,4,6} candidates, argmax.  softmax is monotonic -> argmax of the cosine.
// pred[q*3 + {0,1,2}] = first argmax over the first 2/4/6 candidates (or fewer if the query has fewer).
__global__ void asd_kernel(const float* __restrict__ q, const float* __restrict__ cand, const int32_t* __restrict__ coff,
                           int n, int D, float temp, int32_t* __restrict__ pred) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= n) return;
    const float* qr = q + (long)qi * D;
    float qs = 0.f;
    for (int d = lane; d < D; d += 64) qs += qr[d] * qr[d];
    const float qn = sqrtf(wsum(qs));
    const int c0 = coff[qi];
    int P = coff[qi + 1] - c0;
    P = P < 6 ? P : 6;
    float sim[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        sim[p] = -INFINITY;
        if (p < P) {
            const float* cr = cand + (long)(c0 + p) * D;
            float dot = 0.f, cs = 0.f;
            for (int d = lane; d < D; d += 64) { dot += qr[d] * cr[d]; cs += cr[d] * cr[d]; }
            dot = wsum(dot);
            const float cn = sqrtf(wsum(cs));
            sim[p] = dot / fmaxf(qn * cn, 1e-8f) / temp;
        }
    }
    // the reference takes np.argmax of softmax(sim[:P']) (evaluate_asd.py:47-49,96-100), P' = min(2|4|6, candidates):
    // the softmax is evaluated as torch does (exp(x - max) / sum) so that values it rounds together tie the same way
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int np = (2 * k + 2) < P ? (2 * k + 2) : P;
        float mx = -INFINITY;
#pragma unroll
        for (int p = 0; p < 6; ++p) if (p < np) mx = fmaxf(mx, sim[p]);
        float e[6], den = 0.f;
#pragma unroll
        for (int p = 0; p < 6; ++p) { e[p] = p < np ? expf(sim[p] - mx) : 0.f; den += e[p]; }
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const float sc = e[p] / den;
            if (p < np && sc > best) { best = sc; bi = p; }
        }
        if (lane == 0) pred[qi * 3 + k] = np > 0 ? bi : -1;
    }
}

hipError_t launch_asd(const float* q, const float* cand, const int32_t* coff, int n, int D, float temp,
                      int32_t* pred2, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(asd_kernel, dim3((n + 3) / 4), dim3(256), 0, s, q, cand, coff, n, D, temp, pred2);
    return hipGetLastError();
}
