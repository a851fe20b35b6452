// conv1 of the GestSync VGG stack, direct from uint8 video frames (gfx950).
//
//   Conv3d(3->64, k(5,7,7), s(1,3,3)) + BatchNorm(eval) + ReLU          (gestsync.py:36-41, 308-316)
//   evaluated ONCE per padded-clip position p (window de-duplication).
//
// Formulation: the 5 input frames of a position are treated as 15 channels ("temporal stack",
// padded to a 16-element = 32-byte pixel slot), so K = 49 (kh,kw) slots x 16 = 784 and one
// v_mfma_f32_32x32x16_f16 k-step is exactly one slot.  Nothing is materialised in HBM: the stack
// exists only as an LDS tile built from the raw u8 frames.
//
// Workgroup = 8 waves on one CU (2 per SIMD):
//   waves 0-3  MFMA waves.  Each keeps the complete K=784 weight panel of 32 output channels in
//              196 VGPRs for the whole kernel (weights are the MFMA A operand, never re-read), and
//              per tile computes two 32-position x 32-channel blocks: per k-step ONE ds_read_b128
//              (the patch fragment, MFMA B operand) feeds one MFMA.
//   waves 4-7  loader waves.  They read the 5 source frames of the next tile straight from the
//              u8 HWC video (12-byte = 4-pixel groups, dword loads), convert u8 -> fp16 exactly
//              (0..255 are exact in fp16; the 1/255 is applied in fp32 in the epilogue) and write the
//              32-byte pixel slots into the other LDS buffer.  Their VALU/VMEM work overlaps the MFMA
//              waves' matrix work on the same SIMDs.
// Tile = 4 conv rows x 32 conv cols of one position: 16 input rows x 100 pixel slots (59,904 B with
// the bank padding), double buffered (119,808 B LDS), one barrier per tile.  Persistent grid, one workgroup per CU,
// tiles of one position are processed by neighbouring workgroups at the same time (its 5 frames =
// 1.9 MB stay in the XCD L2s; a frame is re-read for 5 positions).
#include "common.h"
#include <cstdlib>

struct Conv1Args {
    const uint8_t* src;   // [nclip][T][270][480][3]
    int nclip, T, pad, P;
    const f16* Wd;        // [49][64][16]  BN-folded weights, slot-major
    float scale;          // uniform epilogue scale (1/255 for u8 sources; BN scale is folded into Wd)
    const float* shift;   // [64] BN-folded bias
    f16* out;             // [nclip*P][88][158][64]
    long ntiles;
    int dbg;              // ablation switch (env JG_CONV1_DBG): 1 = loaders idle, 2 = MFMA waves idle, 4 = no epilogue
};

namespace {
constexpr int IH = 270, IW = 480, OH = 88, OW = 158;
constexpr int TROWS = 16, TW = 100, SLOT = 32;
// LDS image of a tile row: pixel slot x at byte 32*x + 16*(x/3) -- a 16-B pad after every 3 slots.
// The MFMA patch reads walk pixels 3r+kw (lane r): 112*r + const, which puts the 16 lanes of every
// ds_read_b128 group on 16 different 16-B bank quads (conflict-free, no swizzle, ONE base register
// and compile-time immediates for all 49 (kh,kw) slots); the loaders' ds_write_b128 of 4-pixel groups
// drop from 8-way (128-B stride) to 2-way conflicts.
constexpr int ROW_PITCH = 3744;                        // >= 32*100 + 16*33, multiple of 16
constexpr int TILE_BYTES = TROWS * ROW_PITCH;          // 59904
__device__ __host__ constexpr int slot_off(int x) { return 32 * x + 16 * (x / 3); }
constexpr int ROW_TILES = 22, COL_TILES = 5, TILES_PER_POS = ROW_TILES * COL_TILES;
}

struct C1Regs { uint32_t w[2][5][3]; };     // two (row, 4-pixel group) items x 5 frames x 12 bytes

__global__ __launch_bounds__(512, 2) void conv1_direct_kernel(Conv1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long G = gridDim.x;
    // accumulator init table: shift[c] / scale, so that relu(acc * scale) = relu(conv * scale + shift)
    // with no global load in the epilogue (an L2 round trip per block otherwise)
    float* sInit = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
    if (tid < 64) sInit[tid] = a.shift[tid] / a.scale;
    __syncthreads();

    if (wave >= 4) {
        // =========================== loader waves ===========================
        // Software pipeline in registers: the global loads of tile t+2 are issued before tile t+1 is
        // converted and written to LDS, so they have a whole tile time (and the barrier) to land.
        const int ltid = tid - 256;
        const int row0 = ltid / 25, g0 = ltid - row0 * 25;                 // item ltid        (< 256 <= 400)
        const int it1 = ltid + 256;
        const bool has1 = it1 < TROWS * 25;                                // item ltid + 256  (< 400)
        const int row1 = has1 ? it1 / 25 : 0, g1 = has1 ? it1 - row1 * 25 : 0;

        auto issue = [&](long tile, C1Regs& R) {
            const long nf = tile / TILES_PER_POS;
            const int rem = (int)(tile - nf * TILES_PER_POS);
            const int rt = rem / COL_TILES, j = rem - rt * COL_TILES;
            const int b = (int)(nf / a.P), p = (int)(nf - (long)b * a.P);
#pragma unroll
            for (int dt = 0; dt < 5; ++dt) {
                int f = p + dt - a.pad;
                f = f < 0 ? 0 : (f > a.T - 1 ? a.T - 1 : f);
                const uint8_t* fb = a.src + ((long)b * a.T + f) * (IH * IW * 3);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = u ? row1 : row0, g = u ? g1 : g0;
                    // UNCONDITIONAL loads from a clamped address: a `cond ? load : 0` select makes hipcc branch
                    // around every load and wait vmcnt(0) after it (10 serial L2 round trips per tile).
                    // Pixels >= 480 (strip 4) only feed conv columns 158/159, which are never stored.
                    int ih = rt * 12 + row, px = j * 96 + 4 * g;
                    ih = ih < IH ? ih : IH - 1;
                    px = px < IW ? px : IW - 4;
                    const uint32_t* s = reinterpret_cast<const uint32_t*>(fb + ((long)ih * IW + px) * 3);
                    R.w[u][dt][0] = s[0];
                    R.w[u][dt][1] = s[1];
                    R.w[u][dt][2] = s[2];
                }
            }
        };
        auto cvt_write = [&](const C1Regs& R, char* buf) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !has1) break;
                const int row = u ? row1 : row0, g = u ? g1 : g0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f16 e[16];
#pragma unroll
                    for (int dt = 0; dt < 5; ++dt)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int bi = 3 * q + c;
                            e[dt * 3 + c] = (f16)(float)((R.w[u][dt][bi >> 2] >> ((bi & 3) * 8)) & 0xffu);
                        }
                    e[15] = (f16)0.f;
                    const int x = 4 * g + q;
                    char* dst = buf + row * ROW_PITCH + slot_off(x);
                    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&e[0]);
                    *reinterpret_cast<uint4*>(dst + 16) = *reinterpret_cast<uint4*>(&e[8]);
                }
            }
        };

        C1Regs RA, RB;
        long tile = blockIdx.x;
        if (a.dbg & 1) {
            for (; tile < a.ntiles; tile += G) __syncthreads();
            return;
        }
        if (tile < a.ntiles) {
            issue(tile, RA);
            cvt_write(RA, smem);
            if (tile + G < a.ntiles) issue(tile + G, RA);
        }
        // iteration `it` (tile): after the barrier the MFMA waves read buf[it&1]; we fill buf[(it+1)&1]
        // with tile+G (already in registers) after issuing the loads of tile+2G into the other set.
        int it = 0;
        while (tile < a.ntiles) {
            __syncthreads();
            if (tile + G < a.ntiles) {
                if (tile + 2 * G < a.ntiles) issue(tile + 2 * G, RB);
                cvt_write(RA, smem + ((it + 1) & 1) * TILE_BYTES);
            }
            tile += G; ++it;
            if (tile >= a.ntiles) break;
            __syncthreads();
            if (tile + G < a.ntiles) {
                if (tile + 2 * G < a.ntiles) issue(tile + 2 * G, RA);
                cvt_write(RB, smem + ((it + 1) & 1) * TILE_BYTES);
            }
            tile += G; ++it;
        }
        return;
    }

    // =========================== MFMA waves ===========================
    const int chalf = wave & 1, mb0 = (wave >> 1) & 1;
    const int r = lane & 31, h = lane >> 5;
    f16x8 wreg[49];
#pragma unroll
    for (int s = 0; s < 49; ++s)
        wreg[s] = *reinterpret_cast<const f16x8*>(a.Wd + ((long)(s * 64 + chalf * 32 + r) * 16 + 8 * h));
    // patch-fragment address of lane (r,h) for slot (kh,kw): pixel x = 3r+kw ->
    //   (3*mb+kh)*ROW_PITCH + 32*x + 16*(x/3) + 16*h = [112*r + 16*h] + [kh*ROW_PITCH + 32*kw + 16*(kw/3)]
    const int lbase = 112 * r + 16 * h;
    const int cb = chalf * 32 + 4 * h;
    constexpr int DEPTH = 4;                             // patch fragments in flight per wave
    int it = 0;
    for (long tile = blockIdx.x; tile < a.ntiles; tile += G, ++it) {
        __syncthreads();
        if (a.dbg & 2) continue;
        const char* cur = smem + (it & 1) * TILE_BYTES;
        const long nf = tile / TILES_PER_POS;
        const int rem = (int)(tile - nf * TILES_PER_POS);
        const int rt = rem / COL_TILES, j = rem - rt * COL_TILES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int mb = mb0 + 2 * q;
            const char* base = cur + 3 * mb * ROW_PITCH + lbase;
            auto frag = [&](int s) -> f16x8 {
                const int kh = s / 7, kw = s - kh * 7;
                return *reinterpret_cast<const f16x8*>(base + kh * ROW_PITCH + 32 * kw + 16 * (kw / 3));
            };
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 iv = *reinterpret_cast<const f32x4*>(sInit + cb + 8 * g);
                acc[4 * g] = iv.x; acc[4 * g + 1] = iv.y; acc[4 * g + 2] = iv.z; acc[4 * g + 3] = iv.w;
            }
            f16x8 fr[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) fr[s] = frag(s);
#pragma unroll
            for (int s = 0; s < 49; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], fr[s % DEPTH], acc, 0, 0, 0);
                if (s + DEPTH < 49) fr[s % DEPTH] = frag(s + DEPTH);
                // pin the order: without this hipcc sinks every ds_read next to its MFMA (one fragment
                // register, lgkmcnt(0) per MFMA) and the LDS latency is exposed 49 times per block
                __builtin_amdgcn_sched_barrier(0);
            }
            // D[i][jj]: jj = lane&31 -> position, i = (x&3) + 8*(x>>2) + 4*h -> channel within the half
            const int oh = rt * 4 + mb, ow = j * 32 + r;
            if (ow < OW && !(a.dbg & 4)) {
                f16* o = a.out + (((long)nf * OH + oh) * OW + ow) * 64 + cb;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    v *= a.scale;
                    f16x4 hv = {(f16)fmaxf(v.x, 0.f), (f16)fmaxf(v.y, 0.f), (f16)fmaxf(v.z, 0.f), (f16)fmaxf(v.w, 0.f)};
                    *reinterpret_cast<f16x4*>(o + 8 * g) = hv;
                }
            }
        }
    }
}

hipError_t launch_conv1_direct(const uint8_t* src, int nclip, int T, int pad, const f16* Wd, float scale,
                               const float* shift, f16* out, hipStream_t s) {
    static int num_cu = 0;
    static bool attr_set = false;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) return e;
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_direct_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES + 256);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    Conv1Args a;
    a.src = src; a.nclip = nclip; a.T = T; a.pad = pad; a.P = T + 2 * pad - 4;
    a.Wd = Wd; a.scale = scale; a.shift = shift; a.out = out;
    a.ntiles = (long)nclip * a.P * TILES_PER_POS;
    static const int dbg = getenv("JG_CONV1_DBG") ? atoi(getenv("JG_CONV1_DBG")) : 0;
    a.dbg = dbg;
    if (a.ntiles <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(a.ntiles < num_cu ? a.ntiles : num_cu);
    hipLaunchKernelGGL(conv1_direct_kernel, dim3(grid), dim3(512), 2 * TILE_BYTES + 256, s, a);
    return hipGetLastError();
}
