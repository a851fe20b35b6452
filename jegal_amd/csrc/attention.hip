// Self-attention for short sequences (gfx950).  softmax(q k^T / sqrt(dk) [masked_fill(mask==0,-1e9)]) v
//   gestsync.py:20-21 (nn.MultiheadAttention, S = 21, no mask), modules.py:61-75 (S = T <= 500 or
//   L text tokens, key-padding mask).  Attention is < 5 % of the path's FLOPs and HBM-bound (it reads the
//   packed qkv rows once and writes the context rows once), so all three kernels are organised around the
//   memory and LDS pipes; launch_attention() picks one:
//     attn_mfma_s32_kernel   S <= 32, dk = 64, no mask (the GestSync windows, S = 21): one wave per (window, head),
//                            S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_32x32x16_f16 with K/Q rows as operands straight
//                            from global memory, softmax on the accumulators, P carried as fp16 hi+lo.
//     attn_mfma_kernel<NB>   S <= 160, dk = 64, optional key mask (JEGAL gesture encoder at the dataset's clip lengths):
//                            one workgroup per (clip, head), K and V^T staged in LDS once, one wave per 32 queries.
//     attn_mfma_flash_kernel<DK>  everything else (dk = 64 with S > 160: long clips, XLM-RoBERTa at L > 160; dk = 96: the text
//                            encoder): key chunks of 128 through LDS, online softmax, round 5.
//     attn_kernel<DK>        option attn_mfma = 0 only (A/B, tests): VALU kernel, one lane per
//                            query row, K/V rows in LDS read as wave-wide broadcasts, v_dot2_f32_f16, online softmax
//                            over blocks of 8 keys, fp32 state.  For S <= 32 one wave carries floor(64/S) (sequence, head) pairs.
#include "common.h"

JG_NS_BEGIN

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
#ifdef JG_BF16
                            acc = __builtin_fmaf((float)q[v * 4 + e].x, (float)k8[2 * e], __builtin_fmaf((float)q[v * 4 + e].y, (float)k8[2 * e + 1], acc));
#else
                            acc = __builtin_amdgcn_fdot2(q[v * 4 + e], f16x2{k8[2 * e], k8[2 * e + 1]}, acc, false);
#endif
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


// ---------------------------------------------------------------------------------------------
// MFMA variant for the GestSync layers: S <= 32 keys, dk = 64, no mask.  One wave per (sequence, head) pair,
// four independent waves per workgroup (no workgroup barrier).
//   S^T = K Q^T   (keys x queries, 4 x v_mfma_f32_32x32x16_f16 over d): K rows and Q rows are the A and B
//                 operands straight from global memory (lane = row, 16 B = 8 consecutive d);
//   softmax over the keys of a query = over the 16 accumulator registers of a lane and its partner lane+32;
//   O^T = V^T P^T (d x queries, 2 row blocks x 2 k-steps): P^T is ALREADY in B-operand order in the S^T
//                 accumulators (lane = query; the k-step's 8 keys of lane half h are 4h+{0..3}+16s and
//                 4h+{0..3}+16s+8), V^T comes from a transposed copy of the head's V rows in LDS, read in
//                 the same key order.  P is split into fp16 hi+lo (two MFMAs per block): the probabilities
//                 keep fp32 accuracy, as in the VALU kernel, and the matrix pipe has nothing else to do.
//   The output tile is transposed back through LDS so the stores are whole 128-B rows.
// Rows >= S are clamped on load (finite values), masked to -inf as keys and never stored as queries.
// GATHER (layer 0 of the GestSync transformer, AttnGather in common.h): token j of window (clip c, frame i) is
//   x = conv[c][clamp(i + j - shift)] + pe[j],  so its projection is  W x + b = (W conv[c][p]) + (W pe[j] + b):
// `qkv` then holds ONE projected row per distinct conv position ([clip][P][3D], 154 rows per clip instead of 3150) and
// `g.pe_qkv` the 21 projected positional rows incl. the bias; the operand rows are gathered and summed here.  The windows
// of a clip re-read the same 154 rows, so the blocks are dealt to the XCDs in contiguous ranges (whole clips per L2).
// OR = rows of the K / Q / output image in LDS, >= S: 24 for the GestSync windows (S = 21) -- 8064 B per wave, five
// workgroups (20 waves) per CU, which is also what the 96 VGPRs allow; the kernel is bound by its per-wave latency chain
// (load -> LDS -> MFMA -> softmax -> LDS -> MFMA -> LDS -> store), so the number of waves in flight is what counts.
template <bool GATHER, int OR>
__global__ __launch_bounds__(256, 5) void attn_mfma_s32_kernel(const f16* __restrict__ qkv, int npairs, int S, int H, f16* __restrict__ out, AttnGather g) {
    constexpr int DK = 64, VT_PITCH = 72, O_PITCH = 144;
    __shared__ __attribute__((aligned(16))) char smem[4 * (64 * VT_PITCH + OR * O_PITCH)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wg = blockIdx.x;
    if (GATHER) {
        const int per = ((int)gridDim.x + 7) >> 3;
        wg = (wg & 7) * per + (wg >> 3);
    }
    const int pair = wg * 4 + wave;
    if (pair >= npairs) return;
    char* sVt = smem + wave * (64 * VT_PITCH + OR * O_PITCH);
    const int r31o = (lane & 31) < OR ? (lane & 31) : OR - 1;       // rows >= OR only exist as clamped copies (masked keys, unused queries)
    char* sO = sVt + 64 * VT_PITCH;
    const int b = pair / H, head = pair - b * H;
    const int D = H * DK;
    const long ld = 3L * D;
    const f16* base = qkv + (GATHER ? 0L : (long)b * S * ld) + head * DK;
    const int r31 = lane & 31, hh = lane >> 5;
    // GATHER: source row of token j
    int gc = 0, gi = 0;
    if (GATHER) { gc = b / g.Twin; gi = b - gc * g.Twin - g.shift; }
    auto src_row = [&](int j) -> long {
        if (!GATHER) return j;
        int p = gi + j;
        p = p < 0 ? 0 : (p > g.P - 1 ? g.P - 1 : p);
        return (long)gc * g.P + p;
    };

    // ---- operand loads.  Every row's 128-byte q, k and v slices are read by 8 lanes x 16 B (8 cache lines per load
    // instruction; with lane = row and 16 B per lane a load touches 32 lines and the kernel ends up bound by the CU's
    // address/tag path, not by HBM), and all three operands go through LDS: K, then Q, row-major in the buffer that later
    // holds the output tile (a wave's LDS operations execute in order), V transposed.
    f16x8 qq[4], kk[4], vv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = lane + 64 * r, row = c >> 3, part = c & 7;
        const int rc = row < S ? row : S - 1;
        const f16* p = base + src_row(rc) * ld + part * 8;
        if (!GATHER) {           // read exactly once, by this wave: nontemporal keeps the 310 MB stream from evicting what the other
                                 // kernels of the step (and the other lane) keep in L2 -- attention stage 0.517 -> 0.472 ms per step
            qq[r] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p));
            kk[r] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p + D));
            vv[r] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p + 2 * D));
        } else {
            qq[r] = *reinterpret_cast<const f16x8*>(p);
            kk[r] = *reinterpret_cast<const f16x8*>(p + D);
            vv[r] = *reinterpret_cast<const f16x8*>(p + 2 * D);
        }
    }
    if (GATHER) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = lane + 64 * r, row = c >> 3, part = c & 7;
            const int rc = row < S ? row : S - 1;
            const f16* p = g.pe_qkv + (long)rc * ld + head * DK + part * 8;
            qq[r] += *reinterpret_cast<const f16x8*>(p);
            kk[r] += *reinterpret_cast<const f16x8*>(p + D);
            vv[r] += *reinterpret_cast<const f16x8*>(p + 2 * D);
        }
    }
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    f16x8 kA[4], qB[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = lane + 64 * r, row = c >> 3, part = c & 7;
        if (OR == 32 || row < OR) *reinterpret_cast<f16x8*>(sO + row * O_PITCH + part * 16) = kk[r];
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<f16*>(sVt + (part * 8 + e) * VT_PITCH + row * 2) = vv[r][e];
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < 4; ++s) kA[s] = *reinterpret_cast<const f16x8*>(sO + r31o * O_PITCH + (8 * hh + 16 * s) * 2);
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = lane + 64 * r, row = c >> 3, part = c & 7;
        if (OR == 32 || row < OR) *reinterpret_cast<f16x8*>(sO + row * O_PITCH + part * 16) = qq[r];
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < 4; ++s) qB[s] = *reinterpret_cast<const f16x8*>(sO + r31o * O_PITCH + (8 * hh + 16 * s) * 2);

    // ---- S^T = K Q^T
    f32x16 sc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) sc = JG_MFMA_32x32x16(kA[s], qB[s], sc);

    // ---- softmax over keys: register i <-> key (i&3) + 8(i>>2) + 4*hh
    const float scale = 0.125f;                       // 1/sqrt(64)
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int key = (i & 3) + 8 * (i >> 2) + 4 * hh;
        sc[i] = key < S ? sc[i] * scale : -INFINITY;
        mx = fmaxf(mx, sc[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        sc[i] = __expf(sc[i] - mx);                   // exp(-inf) = 0 for the masked keys
        sum += sc[i];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    f16x8 pH[2], pL[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pv = sc[4 * (2 * s + (j >> 2)) + (j & 3)] * inv;
            const f16 hi = (f16)pv;
            pH[s][j] = hi;
            pL[s][j] = (f16)(pv - (float)hi);
        }

    // ---- O^T = V^T P^T
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x16 o[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) o[blk][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const char* vp = sVt + (r31 + 32 * blk) * VT_PITCH + (16 * s + 4 * hh) * 2;
            const f16x4 v0 = *reinterpret_cast<const f16x4*>(vp), v1 = *reinterpret_cast<const f16x4*>(vp + 16);
            const f16x8 vA = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o[blk] = JG_MFMA_32x32x16(vA, pH[s], o[blk]);
            o[blk] = JG_MFMA_32x32x16(vA, pL[s], o[blk]);
        }
    }
    // ---- O^T (register i <-> d = (i&3) + 8(i>>2) + 4*hh + 32*blk, lane <-> query) -> [query][d] fp16 in LDS -> rows
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f16x4 hv = {(f16)o[blk][4 * g], (f16)o[blk][4 * g + 1], (f16)o[blk][4 * g + 2], (f16)o[blk][4 * g + 3]};
            if (OR == 32 || r31 < OR) *reinterpret_cast<f16x4*>(sO + r31 * O_PITCH + (32 * blk + 8 * g + 4 * hh) * 2) = hv;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f16* obase = out + (long)b * S * D + head * DK;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = lane + 64 * r, row = c >> 3, part = c & 7;
        if (row < S) *reinterpret_cast<f16x8*>(obase + (long)row * D + part * 8) = *reinterpret_cast<const f16x8*>(sO + row * O_PITCH + part * 16);
    }
}


// ---------------------------------------------------------------------------------------------
// MFMA variant for 32 < S <= 160 (JEGAL gesture encoder: S = T = 150 frames), dk = 64, optional key mask.
// One workgroup per (sequence, head) pair, one wave per block of 32 queries (NB = ceil(S/32) waves).  K (row-major)
// and V (transposed) of the head are staged in LDS once per pair; every wave computes its 32 x S score block
// S^T = K Q^T into NB accumulator tiles, softmaxes over its registers and its partner lane, and multiplies by
// V^T straight from the accumulators (same operand trick and fp16 hi+lo probabilities as attn_mfma_s32_kernel).
// masked_fill(mask == 0, -1e9) as in modules.py:66-67; keys >= S get -inf.
template <int NB>
__global__ __launch_bounds__(64 * NB) void attn_mfma_kernel(const f16* __restrict__ qkv, const float* __restrict__ keymask,
                                                            int S, int H, f16* __restrict__ out) {
    constexpr int DK = 64, SP = 32 * NB, K_PITCH = 144, VT_PITCH = SP * 2 + 8, O_PITCH = 144, NT = 64 * NB;
    __shared__ __attribute__((aligned(16))) char sK[SP * K_PITCH];
    __shared__ __attribute__((aligned(16))) char sVt[64 * VT_PITCH];
    __shared__ __attribute__((aligned(16))) float sM[SP];
    static_assert(K_PITCH == O_PITCH, "the output tiles reuse the K image");
    char* sOall = sK;           // after the barrier that follows the score phase
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x;
    const int b = pair / H, head = pair - b * H;
    const int D = H * DK;
    const long ld = 3L * D;
    const f16* base = qkv + (long)b * S * ld + head * DK;
    const int r31 = lane & 31, hh = lane >> 5;

    // ---- stage K (row-major), V (transposed) and the mask row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = tid + NT * r, row = c >> 3, part = c & 7;          // SP rows x 8 chunks = 4 * NT
        const int rc = row < S ? row : S - 1;
        const f16* src = base + (long)rc * ld + part * 8;
        const f16x8 kk = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(src + D));
        const f16x8 vv = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(src + 2 * D));
        *reinterpret_cast<f16x8*>(sK + row * K_PITCH + part * 16) = kk;
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<f16*>(sVt + (part * 8 + e) * VT_PITCH + row * 2) = vv[e];
    }
    for (int j = tid; j < SP; j += NT) sM[j] = j < S ? (keymask ? keymask[(long)b * S + j] : 1.f) : -1.f;   // -1: beyond S
    // this wave's queries: B operand straight from global memory
    const int q0 = 32 * wave;
    const int qr = q0 + r31 < S ? q0 + r31 : S - 1;
    f16x8 qB[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qB[s] = *reinterpret_cast<const f16x8*>(base + (long)qr * ld + 16 * s + 8 * hh);
    __syncthreads();

    // ---- scores: NB tiles of 32 keys x 32 queries
    f32x16 sc[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[kb][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 kA = *reinterpret_cast<const f16x8*>(sK + (32 * kb + r31) * K_PITCH + (16 * s + 8 * hh) * 2);
            sc[kb] = JG_MFMA_32x32x16(kA, qB[s], sc[kb]);
        }
    }
    // ---- softmax over the keys: register i of tile kb <-> key 32kb + (i&3) + 8(i>>2) + 4*hh
    const float scale = 0.125f;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(&sM[32 * kb + 8 * g + 4 * hh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = sc[kb][4 * g + e] * scale;
                v = mk[e] < 0.f ? -INFINITY : (mk[e] == 0.f ? -1e9f : v);
                sc[kb][4 * g + e] = v;
                mx = fmaxf(mx, v);
            }
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sc[kb][i] = __expf(sc[kb][i] - mx);
            sum += sc[kb][i];
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;

    // ---- O^T = V^T P^T
    f32x16 o[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[blk][i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 pH, pL;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = sc[kb][4 * (2 * s + (j >> 2)) + (j & 3)] * inv;
                const f16 hi = (f16)pv;
                pH[j] = hi;
                pL[j] = (f16)(pv - (float)hi);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const char* vp = sVt + (r31 + 32 * blk) * VT_PITCH + (32 * kb + 16 * s + 4 * hh) * 2;
                const f16x4 v0 = *reinterpret_cast<const f16x4*>(vp), v1 = *reinterpret_cast<const f16x4*>(vp + 16);
                const f16x8 vA = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[blk] = JG_MFMA_32x32x16(vA, pH, o[blk]);
                o[blk] = JG_MFMA_32x32x16(vA, pL, o[blk]);
            }
        }
    // ---- back to [query][d] rows through this wave's LDS slice (aliases the K image: every wave is done with K)
    __syncthreads();
    char* sO = sOall + wave * (32 * O_PITCH);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f16x4 hv = {(f16)o[blk][4 * g], (f16)o[blk][4 * g + 1], (f16)o[blk][4 * g + 2], (f16)o[blk][4 * g + 3]};
            *reinterpret_cast<f16x4*>(sO + r31 * O_PITCH + (32 * blk + 8 * g + 4 * hh) * 2) = hv;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f16* obase = out + ((long)b * S + q0) * D + head * DK;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = lane + 64 * r, row = c >> 3, part = c & 7;
        if (q0 + row < S) *reinterpret_cast<f16x8*>(obase + (long)row * D + part * 8) = *reinterpret_cast<const f16x8*>(sO + row * O_PITCH + part * 16);
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA variant for every other instance (VERDICT r4 row k): dk = 64 with 160 < S (JEGAL clips of 161..500 frames, XLM-RoBERTa at
// L > 160) and dk = 96 at any S (JEGAL text encoder: d = 768, h = 8, jegal.py:35-38), optional key mask.  One workgroup per
// (sequence, head, group of up to 256 queries), one wave per 32 queries; the keys go by in chunks of 128: K (row-major) and V
// (transposed) of a chunk are staged in LDS once per workgroup, every wave computes its 32 x 128 score block S^T = K Q^T on
// v_mfma_f32_32x32x16, updates its ONLINE softmax state (running max and sum per query = per lane pair, fp32) and accumulates
// O^T += V^T P^T with the probabilities split into fp16 hi+lo straight from the score accumulators (operand trick of
// attn_mfma_s32_kernel).  The 1 / sum normalisation happens once, in fp32, at the end.  masked_fill(mask == 0, -1e9) as in
// modules.py:66-67; keys >= S get -inf; score tiles entirely beyond S are skipped.
template <int DK>
__global__ __launch_bounds__(512) void attn_mfma_flash_kernel(const f16* __restrict__ qkv, const float* __restrict__ keymask, int S, int H,
                                                              f16* __restrict__ out) {
    constexpr int KC = 128, NV = DK / 8, KS = DK / 16, NBLK = DK / 32;
    constexpr int K_PITCH = DK * 2 + 16, VT_PITCH = KC * 2 + 8, O_PITCH = 80;
    constexpr int K_BYTES = KC * K_PITCH > 8 * 32 * O_PITCH ? KC * K_PITCH : 8 * 32 * O_PITCH;      // the output slices of eight waves alias the K image
    __shared__ __attribute__((aligned(16))) char sK[K_BYTES];
    __shared__ __attribute__((aligned(16))) char sVt[DK * VT_PITCH];
    __shared__ __attribute__((aligned(16))) float sM[KC];
    const int tid = threadIdx.x, lane = tid & 63, NT = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x;
    const int b = pair / H, head = pair - b * H;
    const int D = H * DK;
    const long ld = 3L * D;
    const f16* base = qkv + (long)b * S * ld + head * DK;
    const int r31 = lane & 31, hh = lane >> 5;
    const int q0 = 256 * blockIdx.y + 32 * wave;
    const bool wave_on = q0 < S;                                    // (waves past the last query still help with the staging)
    const int qr = q0 + r31 < S ? q0 + r31 : S - 1;
    f16x8 qB[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qB[s] = *reinterpret_cast<const f16x8*>(base + (long)qr * ld + 16 * s + 8 * hh);
    f32x16 o[NBLK];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[blk][i] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const float scale = DK == 64 ? 0.125f : 0.10206207261596575f;   // 1 / sqrt(dk)

    for (int k0 = 0; k0 < S; k0 += KC) {
        __syncthreads();                                            // every wave is done with the previous chunk
        for (int idx = tid; idx < KC * NV; idx += NT) {
            const int row = idx / NV, part = idx - row * NV;
            const int rc = k0 + row < S ? k0 + row : S - 1;
            const f16* src = base + (long)rc * ld + part * 8;
            const f16x8 kk = *reinterpret_cast<const f16x8*>(src + D);
            const f16x8 vv = *reinterpret_cast<const f16x8*>(src + 2 * D);
            *reinterpret_cast<f16x8*>(sK + row * K_PITCH + part * 16) = kk;
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<f16*>(sVt + (part * 8 + e) * VT_PITCH + row * 2) = vv[e];
        }
        for (int j = tid; j < KC; j += NT) sM[j] = k0 + j < S ? (keymask ? keymask[(long)b * S + k0 + j] : 1.f) : -1.f;   // -1: beyond S
        __syncthreads();
        if (!wave_on) continue;
        const int ntile = (S - k0 + 31) / 32 < KC / 32 ? (S - k0 + 31) / 32 : KC / 32;      // tiles of this chunk that hold a key
        f32x16 sc[KC / 32];
        float cm = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KC / 32; ++kb) {
            if (kb >= ntile) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[kb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f16x8 kA = *reinterpret_cast<const f16x8*>(sK + (32 * kb + r31) * K_PITCH + (16 * s + 8 * hh) * 2);
                sc[kb] = JG_MFMA_32x32x16(kA, qB[s], sc[kb]);
            }
            // register i of tile kb <-> key k0 + 32kb + (i&3) + 8(i>>2) + 4*hh
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(&sM[32 * kb + 8 * g + 4 * hh]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = sc[kb][4 * g + e] * scale;
                    v = mk[e] < 0.f ? -INFINITY : (mk[e] == 0.f ? -1e9f : v);
                    sc[kb][4 * g + e] = v;
                    cm = fmaxf(cm, v);
                }
            }
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float mnew = fmaxf(mrun, cm);                         // finite: the chunk holds at least one key < S
        const float alpha = __expf(mrun - mnew);                    // first chunk: exp(-inf) = 0
        mrun = mnew;
        lrun *= alpha;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[blk][i] *= alpha;
#pragma unroll
        for (int kb = 0; kb < KC / 32; ++kb) {
            if (kb >= ntile) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sc[kb][i] = __expf(sc[kb][i] - mnew);
                lrun += sc[kb][i];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 pH, pL;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = sc[kb][4 * (2 * s + (j >> 2)) + (j & 3)];
                    const f16 hi = (f16)pv;
                    pH[j] = hi;
                    pL[j] = (f16)(pv - (float)hi);
                }
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
                    const char* vp = sVt + (r31 + 32 * blk) * VT_PITCH + (32 * kb + 16 * s + 4 * hh) * 2;
                    const f16x4 v0 = *reinterpret_cast<const f16x4*>(vp), v1 = *reinterpret_cast<const f16x4*>(vp + 16);
                    const f16x8 vA = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[blk] = JG_MFMA_32x32x16(vA, pH, o[blk]);
                    o[blk] = JG_MFMA_32x32x16(vA, pL, o[blk]);
                }
            }
        }
    }
    // ---- normalise, back to [query][d] rows through this wave's LDS slice (aliases the K image), 32 columns at a time
    __syncthreads();
    if (!wave_on) return;
    const float inv = 1.f / (lrun + __shfl_xor(lrun, 32, 64));
    char* sO = sK + wave * (32 * O_PITCH);
    f16* obase = out + ((long)b * S + q0) * D + head * DK;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f16x4 hv = {(f16)(o[blk][4 * g] * inv), (f16)(o[blk][4 * g + 1] * inv), (f16)(o[blk][4 * g + 2] * inv), (f16)(o[blk][4 * g + 3] * inv)};
            *reinterpret_cast<f16x4*>(sO + r31 * O_PITCH + (8 * g + 4 * hh) * 2) = hv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = lane + 64 * r, row = c >> 2, part = c & 3;
            if (q0 + row < S) *reinterpret_cast<f16x8*>(obase + (long)row * D + 32 * blk + part * 8) = *reinterpret_cast<const f16x8*>(sO + row * O_PITCH + part * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int DK>
static hipError_t launch_attn_flash(const f16* qkv, const float* keymask, long npairs, int S, int H, f16* out, hipStream_t s) {
    const int nq = S < 256 ? S : 256;
    hipLaunchKernelGGL((attn_mfma_flash_kernel<DK>), dim3((unsigned)npairs, (unsigned)((S + 255) / 256)), dim3(64 * ((nq + 31) / 32)), 0, s, qkv, keymask, S, H, out);
    return hipGetLastError();
}

template <int NB>
static hipError_t launch_attn_mfma(const f16* qkv, const float* keymask, long npairs, int S, int H, f16* out, hipStream_t s) {
    hipLaunchKernelGGL((attn_mfma_kernel<NB>), dim3((unsigned)npairs), dim3(64 * NB), 0, s, qkv, keymask, S, H, out);
    return hipGetLastError();
}

// Layer-0 attention of the GestSync transformer from per-position projections (see attn_mfma_s32_kernel<true>):
// B = windows (nclip * g.Twin), S <= 32 tokens, dk = 64.
hipError_t launch_attention_gather(const f16* qkv_pos, const AttnGather& g, int B, int S, int H, f16* out, hipStream_t s) {
    if (B <= 0 || S <= 0) return hipSuccess;
    const long npairs = (long)B * H;
    if (S > 32 || npairs >= (1L << 31) || g.Twin <= 0 || g.P <= 0 || !g.pe_qkv) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((npairs + 3) / 4);
    if (S <= 24) hipLaunchKernelGGL((attn_mfma_s32_kernel<true, 24>), dim3((blocks + 7) / 8 * 8), dim3(256), 0, s, qkv_pos, (int)npairs, S, H, out, g);
    else hipLaunchKernelGGL((attn_mfma_s32_kernel<true, 32>), dim3((blocks + 7) / 8 * 8), dim3(256), 0, s, qkv_pos, (int)npairs, S, H, out, g);
    return hipGetLastError();
}

hipError_t launch_attention(const f16* qkv, const float* keymask, int B, int S, int H, int dk, f16* out, const EngineOpts& o, hipStream_t s) {
    if (B <= 0 || S <= 0) return hipSuccess;
    const long npairs = (long)B * H;
    if (o.attn_mfma && S <= 32 && dk == 64 && !keymask && npairs < (1L << 31)) {
        if (S <= 24) hipLaunchKernelGGL((attn_mfma_s32_kernel<false, 24>), dim3((unsigned)((npairs + 3) / 4)), dim3(256), 0, s, qkv, (int)npairs, S, H, out, AttnGather{});
        else hipLaunchKernelGGL((attn_mfma_s32_kernel<false, 32>), dim3((unsigned)((npairs + 3) / 4)), dim3(256), 0, s, qkv, (int)npairs, S, H, out, AttnGather{});
        return hipGetLastError();
    }
    if (o.attn_mfma && S <= 160 && dk == 64 && npairs < (1L << 31)) {      // S <= 32 with a key mask: one query block
        switch ((S + 31) / 32) {
            case 1: return launch_attn_mfma<1>(qkv, keymask, npairs, S, H, out, s);
            case 2: return launch_attn_mfma<2>(qkv, keymask, npairs, S, H, out, s);
            case 3: return launch_attn_mfma<3>(qkv, keymask, npairs, S, H, out, s);
            case 4: return launch_attn_mfma<4>(qkv, keymask, npairs, S, H, out, s);
            default: return launch_attn_mfma<5>(qkv, keymask, npairs, S, H, out, s);
        }
    }
    // everything else on the matrix cores too: dk = 64 beyond 160 keys, dk = 96 (text encoder) at any length
    if (o.attn_mfma && npairs < (1L << 31)) {
        if (dk == 64) return launch_attn_flash<64>(qkv, keymask, npairs, S, H, out, s);
        if (dk == 96) return launch_attn_flash<96>(qkv, keymask, npairs, S, H, out, s);
    }
    // option attn_mfma = 0 (A/B and tests): the VALU kernel
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

JG_NS_END
