// conv1 of the GestSync VGG stack, direct from uint8 video frames, with the max-pool fused (gfx950).
//
//   Conv3d(3->64, k(5,7,7), s(1,3,3)) + BatchNorm(eval) + ReLU + MaxPool3d((1,3,3),(1,2,2))
//   (gestsync.py:36-46, 308-321), evaluated ONCE per distinct padded-clip position
//   (window de-duplication).  Neither the temporal stack nor the 88x158x64
//   pre-pool tensor ever reaches HBM: in = raw u8 frames, out = pooled NHWC fp16 (43x78x64).
//
// Formulation: the 5 input frames of a position are treated as 15 channels ("temporal stack",
// padded to a 16-element = 32-byte pixel slot), so K = 49 (kh,kw) slots x 16 = 784 and one
// v_mfma_f32_32x32x16_f16 k-step is exactly one slot.
//
// Workgroup = 8 waves on one CU (2 per SIMD), persistent, one workgroup per CU:
//   waves 0-3  MFMA waves.  Each keeps the complete K=784 weight panel of 32 output channels in
//              196 VGPRs for the whole kernel (weights are the MFMA A operand, never re-read) and per
//              tile computes two 32-position x 32-channel blocks; per k-step ONE ds_read_b128 (the
//              patch fragment, MFMA B operand) feeds one MFMA, 6 fragments in flight (order pinned
//              with sched_barrier).  Epilogue: scale, ReLU, fp16 -> LDS conv buffer (ds_write_b64).
//   waves 4-7  loader/pool waves.  (a) read the 5 source frames of the tile three ahead straight
//              from the u8 HWC video (inline-asm dwordx3 loads with hand-counted waits, two register
//              sets), (b) pool the previous tile's conv rows from the LDS conv buffer and store the
//              pooled rows (16-B nontemporal stores), (c) turn the next tile's u8 into fp16 by byte
//              placement (n -> the subnormal n * 2^-24, exact; the 2^24/255 rides in the epilogue scale)
//              and write its 32-byte pixel slots.  Their VALU/VMEM/LDS work overlaps the matrix work.
// Zero input bands (the reference blanks the face rows): conv1_zero_scan_kernel + conv1_skip_mask_kernel find
// them before the launch; tiles whose band and upper neighbour are zero are skipped outright (constant
// fill), and the same scan tells conv2 how many of its leading output rows are copies of one row.
// Tile = 4 conv rows x 32 conv cols of one position (16 input rows x 100 pixel slots), double
// buffered, one barrier per tile.  A workgroup marches DOWN a 32-column strip (22 tiles), so the
// vertical 3x3/s2 pooling window that straddles two tiles is served by a one-row carry
// (max of the previous tile's last two conv rows) kept in LDS.  The horizontal window that
// straddles two strips (pooled column 16j+15 needs conv column 32(j+1)) is closed by a tiny
// fix-up kernel from an "edge" side buffer (4 of 78 pooled columns).
// Neighbouring workgroups work on the strips of one position at the same time, so its 5 frames
// (1.9 MB) stay in the XCD L2s; a frame is re-read for 5 positions.
#include "common.h"

// Diagnostic build only (-DJG_CLOCK_STAMPS, tools/inkernel_clock.py; MI355X_MICROARCH.md "DVFS give-back" item 6): MFMA wave 0 of every
// workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its main loop; the differences go to a buffer of
// their own that no kernel reads.  In the product build no stamp executes.
#ifdef JG_CLOCK_STAMPS
__device__ unsigned long long jg_clock_stamps_conv1[2 * 1024];
extern "C" int jg_clock_read_conv1(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jg_clock_stamps_conv1), sizeof(unsigned long long) * (n < 2048 ? n : 2048)) == hipSuccess ? 0 : 1;
}
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Conv1Args {
    const uint8_t* src;   // [nclip][T][270][480][3]
    int nclip, T, pad, P;
    const f16* Wd;        // [49][64][16]  BN-folded weights, slot-major
    float scale;          // uniform epilogue scale (1/255 for u8 sources; BN scale is folded into Wd)
    f16* out;             // pooled [nclip*P][43][78][64]
    f16* edge;            // [nclip*P][43][4][64]: vertically pooled conv column 32*j (j=1..4)
    f16* dump;            // [gridDim.x][256][8]: scratch slots for stores of lanes that have no output (behind `edge`)
    int nstrips;          // nclip * P * 5
    float invP;           // 1/P for the position -> (clip, frame) split
    unsigned long long* tl;   // debug timeline (env JG_CONV1_TL): 100 MHz stamps of workgroup 0, waves 0 and 4
    int zskip;            // 1: all-zero input tiles (the face-mask rows) run only the two bias slots
    const unsigned* zmask;    // [nclip*P] per POSITION: bit rt = input band rt is zero in all five frames of the position
                              // (conv1_zero_scan_kernel + conv1_skip_mask_kernel); tile rt is skipped outright when bits rt and
                              // rt-1 are set; nullptr: no pre-scan, zero tiles are detected from the loaded bytes
    const f16* zconst;        // [64] relu(bias) per channel as fp16: the value of every conv1 output whose patch is all zero
    int fill_partial;         // 1: conv2 honours the per-position row skip (common.h, ConvGeom::rowmap): it reads pooled rows >= 2 s2 of a
                              // position only (s2 from the position's own skip mask), and the constant fill leaves the rows above unwritten
    int dbg;              // ablation switch (env JG_CONV1_DBG): 1 = loaders idle, 2 = MFMA waves idle, 4 = no pooling,
                          // 8 = no u8->fp16 conversion / LDS fill (timing experiments only)
};

namespace {
constexpr int IH = 270, IW = 480;
constexpr int PH = 43, PW = 78;
constexpr int TROWS = 16;
// LDS image of a tile row: pixel slot x at byte 32*x + 16*(x/3) -- a 16-B pad after every 3 slots.
// The MFMA patch reads walk pixels 3r+kw (lane r): 112*r + const, which puts the 16 lanes of every
// ds_read_b128 group on 16 different 16-B bank quads (conflict-free, no swizzle, ONE base register
// and compile-time immediates for all 49 (kh,kw) slots); the loaders' ds_write_b128 of 4-pixel groups
// drop from 8-way (128-B stride) to 2-way conflicts.
constexpr int ROW_PITCH = 3744;                        // >= 32*100 + 16*33, multiple of 16
constexpr int TILE_BYTES = TROWS * ROW_PITCH;          // 59904
__device__ __host__ constexpr int slot_off(int x) { return 32 * x + 16 * (x / 3); }
constexpr int ROW_TILES = 22, COL_TILES = 5;
// conv buffer: [conv row 4][channel group 8][col 32][16 B]; carry: [channel group 8][col 32][16 B]
constexpr int CONV_BYTES = 4 * 8 * 32 * 16;            // 16384
constexpr int CARRY_BYTES = 8 * 32 * 16;               // 4096
constexpr int OFF_CONV = 2 * TILE_BYTES;               // 119808
constexpr int OFF_CARRY = OFF_CONV + 2 * CONV_BYTES;   // 152576
constexpr int OFF_INIT = OFF_CARRY + 2 * CARRY_BYTES;  // 160768: int flags[2][4] -- "loader wave w saw a non-zero byte in tile buffer b"
constexpr int OFF_LIVE = OFF_INIT + 32;                // int live[2] -- "tile buffer b holds a tile" (0 after the workgroup's last tile)
constexpr int OFF_SKIP = OFF_INIT + 256;               // unsigned skip[MAX_WG_STRIPS]: skip mask of this workgroup's k-th strip
constexpr int MAX_WG_STRIPS = 640;
constexpr int LDS_BYTES = OFF_SKIP + 4 * MAX_WG_STRIPS;   // 163584 <= 163840
constexpr int MAX_WGS = 1024;
constexpr int DUMP_HALVES_PER_WG = 256 * 8;            // one 16-B slot per loader thread for stores that have nowhere to go
}

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
struct C1Regs { u32x3 w[2][5]; };           // two (row, 4-pixel group) items x 5 frames x 12 bytes

__device__ __forceinline__ f16x8 max8(f16x8 a, f16x8 b) {
    return __builtin_elementwise_max(a, b);            // 4 x v_pk_max_f16
}

// conv2's position-independent leading output rows of a position from its skip mask: with row tiles 0..L-1 skipped the pooled
// rows 0..2L-2 hold relu(bias) in every column, and conv2 (5x5, stride 2, no padding) output row oh reads pooled rows
// 2oh..2oh+4: rows 0..L-3 are what conv2 computes from an all-constant image.  An all-black position still computes its last row.
__device__ __forceinline__ int conv1_s2_of_mask(unsigned sk) {
    const int L = __builtin_ctz(~sk);                // sk has 22 bits: L <= 22
    constexpr int C2_OH = (PH - 5) / 2 + 1;          // conv2 output rows (20)
    const int rs = L >= 2 ? L - 2 : 0;
    return rs < C2_OH ? rs : C2_OH - 1;
}

// DBG: the timeline stamps (JG_CONV1_TL) and the ablation switches (JG_CONV1_DBG) exist only in the <true> instantiation: in the
// loader waves every extra scalar compare-and-branch per tile is on the critical path (one wave per SIMD, ~9 cycles per
// instruction next to the MFMA wave).
// M16: the MFMA waves run v_mfma_f32_16x16x32_f16 (two pixel slots per k-step) instead of 32x32x16 (one slot): the same FLOPs per
// cycle, but the chip holds a higher clock on the 16x16x32 shape (MI355X_MICROARCH.md "DVFS give-back" item 7; tools/mfma_rate.hip
// on this part, random data, registers only: 19.7 ns per 32x32x16 = 1.70 PFLOP/s against 8.2-9.2 ns per 16x16x32 = 1.8-2.05),
// and the MFMA waves are this kernel's critical role.  See the M16 block below for the k-step pairing and the LDS addressing.
template <bool DBG, bool M16>
__global__ __launch_bounds__(512, 2) void conv1_direct_kernel(Conv1Args a) {
    const int dbg = DBG ? a.dbg : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long G = gridDim.x;
    // The BN-folded bias rides in the GEMM: element 15 of every pixel slot is "1.0" (2^-24, the
    // same scale as the pixel values, see cvt_write) and the weight panel
    // holds shift/scale (hi+lo fp16 pair) at [slot 0][c][15] and [slot 1][c][15] -- no accumulator
    // init read, no global load in the epilogue:  out = relu(acc * scale).

    // Strip assignment, XCD-aware: workgroups b, b+8, b+16, ... share an XCD (and its 4 MB L2), so each
    // XCD gets ONE contiguous range of strips (= whole clips) and its workgroups walk it together --
    // the 5 frames of a position are then fetched into that L2 once instead of into all eight.
    // local tile t of this workgroup -> strip s_lo + (t/22)*GX, row tile t%22
    // All of this index math is 32-bit and incremental: the first version decoded every tile id with 64-bit
    // divisions (t/22, strip/5, nf/P) in both issue() and pool() -- ~1.2 us of VALU per tile on the loader waves,
    // which made THEM the critical path (tools/conv1_ablate.py, JG_CONV1_TL=1).
    const int G32 = (int)G;
    const int xcd = blockIdx.x & 7, xidx = blockIdx.x >> 3;
    const int GX = (G32 + 7 - xcd) >> 3;                                 // workgroups on this XCD
    const int per = (a.nstrips + 7) >> 3;
    const int r_lo = xcd * per, r_hi = (r_lo + per < a.nstrips) ? r_lo + per : a.nstrips;
    const int s_lo = r_lo + xidx;
    // Tiles that are skipped OUTRIGHT (zmask, produced by conv1_zero_scan_kernel from the frames): a tile whose 16 input rows are
    // zero in all 5 frames computes relu(bias) everywhere; if the tile above is such a tile too (or does not exist), both of
    // its pooled rows and its carry are that constant whatever is below, so the tile needs no loads, no barrier slot and no
    // MFMA -- its pooled rows are filled with the constant when the pool waves enter the strip, and the tile below takes the
    // constant as its carry (pool(), carry_const).  A zero tile BELOW a non-zero tile still runs (2 bias slots, see
    // cvt_write): its upper pooled row mixes with real data.
    // The walk below visits the remaining tiles of this workgroup's strips in order; all of it is wave-uniform.
    struct Walk {
        int strip, rt;
        int k;              // strip = s_lo + k * GX: index into the workgroup's skip table in LDS
        unsigned skip;      // bit rt: tile rt of `strip` is skipped
        unsigned z;         // bit rt: input band rt is zero in all five frames of the strip's position (the pre-scan's knowledge)
        bool done;
    };
    // strip -> (position nf, column tile j, clip b, padded-clip position p)
    auto decode = [&](int strip, int& nf, int& j, int& b, int& pp) {
        nf = (int)((unsigned)strip / 5u);
        j = strip - nf * 5;
        b = (int)((float)nf * a.invP);               // nf < 2^24 (checked by the launcher): off by at most one
        pp = nf - b * a.P;
        if (pp < 0) { --b; pp += a.P; }
        else if (pp >= a.P) { ++b; pp -= a.P; }
    };
    // The skip masks of this workgroup's strips sit in LDS (filled below, before the roles split): a global load inside the
    // walks would put an s_waitcnt vmcnt(0) -- hipcc cannot count across the walks' loops -- behind every batch of frame loads.
    const unsigned* skip_tab = reinterpret_cast<const unsigned*>(smem + OFF_SKIP);
    const bool use_skip = a.zmask != nullptr;
    // the table holds the positions' zero-band masks z; a tile is skipped when its band and the band above are zero
    auto strip_zero = [&](int k) -> unsigned {
        return use_skip ? (unsigned)__builtin_amdgcn_readfirstlane((int)skip_tab[k]) : 0u;
    };
    auto skip_of = [](unsigned z) -> unsigned { return z & ((z << 1) | 1u); };
    // on_strip(strip, skip) is called once for every strip the walk enters (the pool walk fills the skipped tiles there)
    auto walk_next = [&](Walk& q, auto&& on_strip) {
        if (q.done) return;
        while (true) {
            if (++q.rt == ROW_TILES) {
                q.rt = 0;
                q.strip += GX;
                ++q.k;
                if (q.strip >= r_hi) { q.done = true; return; }
                q.z = strip_zero(q.k);
                q.skip = skip_of(q.z);
                on_strip(q.strip, q.skip);
            }
            if (!((q.skip >> q.rt) & 1u)) return;
        }
    };
    auto walk_first = [&](auto&& on_strip) -> Walk {
        Walk q = {s_lo, -1, 0, 0u, 0u, !(s_lo < r_hi && GX > 0)};
        if (q.done) return q;
        q.z = strip_zero(0);
        q.skip = skip_of(q.z);
        on_strip(q.strip, q.skip);
        walk_next(q, on_strip);
        return q;
    };
    auto no_fill = [](int, unsigned) {};

    if (!(s_lo < r_hi && GX > 0)) return;      // a workgroup without a strip (launches of fewer strips than CUs): nothing to do, for either role
    if (use_skip) {
        unsigned* tab = reinterpret_cast<unsigned*>(smem + OFF_SKIP);
        for (int k = tid; k < MAX_WG_STRIPS; k += 512) {
            const int strip = s_lo + k * GX;
            tab[k] = strip < r_hi ? a.zmask[(unsigned)strip / 5u] : 0u;
        }
        __syncthreads();
    }
    if (wave >= 4) {
        // =========================== loader / pool waves ===========================
        const int ltid = tid - 256;
        const int row0 = ltid / 25, g0 = ltid - row0 * 25;                 // item ltid        (< 256 <= 400)
        const int it1 = ltid + 256;
        const bool has1 = it1 < TROWS * 25;                                // item ltid + 256  (< 400)
        const int row1 = has1 ? it1 / 25 : 0, g1 = has1 ? it1 - row1 * 25 : 0;

        // ---- frame loads.  Everything that depends only on the STRIP is computed when the issue walk enters it (strip_setup):
        // the wave-uniform base of the position's first frame, the byte distance of its other four frames (frames are clamped
        // at the clip's ends, inference_embs.py:283) and the per-lane byte offset of the two items at row tile 0.  Per tile that
        // leaves one scalar add per frame and one vector add per item.
        // (kept in SGPRs by construction -- readfirstlane at the point of definition -- so that the per-tile address arithmetic
        // is scalar and the loads' base registers are SALU-written: a VALU-written SGPR needs 5 wait states before a VMEM
        // instruction may read it, and hipcc pads nothing for the operands of an asm statement)
        uint32_t fb0_lo = (uint32_t)(uintptr_t)a.src, fb0_hi = (uint32_t)((uintptr_t)a.src >> 32);   // frame clamp(p - pad) of the issue walk's strip
        uint32_t fdelta[5] = {0, 0, 0, 0, 0};       // byte distance of frame dt from it
        uint32_t ioff[2] = {0, 0};                  // per lane: ((row_u) * IW + px_u) * 3 for the strip's column tile
        auto strip_setup = [&](int strip) {
            int nf, j, b, p;
            decode(strip, nf, j, b, p);
            int f0 = p - a.pad;
            f0 = f0 < 0 ? 0 : (f0 > a.T - 1 ? a.T - 1 : f0);
            const uintptr_t fb = (uintptr_t)(a.src + (size_t)(b * a.T + f0) * (size_t)(IH * IW * 3));
            fb0_lo = __builtin_amdgcn_readfirstlane((uint32_t)fb);
            fb0_hi = __builtin_amdgcn_readfirstlane((uint32_t)(fb >> 32));
#pragma unroll
            for (int dt = 0; dt < 5; ++dt) {
                int f = p + dt - a.pad;
                f = f < 0 ? 0 : (f > a.T - 1 ? a.T - 1 : f);
                fdelta[dt] = __builtin_amdgcn_readfirstlane((uint32_t)(f - f0) * (uint32_t)(IH * IW * 3));
            }
            // Pixels >= 480 (strip 4) only feed conv columns 158/159, which are never used: clamp the group to the last one
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = u ? row1 : row0, g = u ? g1 : g0;
                int px = j * 96 + 4 * g;
                px = px < IW ? px : IW - 4;
                ioff[u] = (uint32_t)((row * IW + px) * 3);
            }
        };
        // Input rows 12*rt + row <= 267 < 270: no row clamp.  UNCONDITIONAL loads (a `cond ? load : 0` select makes hipcc
        // branch around every load); inline asm on purpose: hipcc must neither count nor wait for them.  Left to the compiler,
        // the wait in front of cvt_write() is vmcnt(10) -- it cannot count the pool stores behind branches and the other register
        // set's loads across the loop edge -- i.e. every iteration also waited for the pooled-row STORES issued just before the
        // barrier.  wait_frames() below counts by hand.
        auto issue = [&](int rt, C1Regs& R) {
            const uint32_t roff = (uint32_t)rt * (uint32_t)(12 * IW * 3);
            const uint32_t o0 = ioff[0] + roff, o1 = ioff[1] + roff;
#pragma unroll
            for (int dt = 0; dt < 5; ++dt) {
                const uint8_t* fbu = reinterpret_cast<const uint8_t*>(((uintptr_t)fb0_lo | ((uintptr_t)fb0_hi << 32)) + fdelta[dt]);
                // s_nop 4: the base may come straight out of a v_readfirstlane, and hipcc pads no hazard for an operand of an asm
                // statement (VALU-written SGPR -> VMEM address: 5 wait states; cdna_hip_programming.md 5.7 item 2)
                if (DBG && (dbg & 32)) {            // debugging aid: compiler-visible loads
                    R.w[0][dt] = *reinterpret_cast<const u32x3*>(fbu + o0);
                    R.w[1][dt] = *reinterpret_cast<const u32x3*>(fbu + o1);
                    continue;
                }
                asm volatile("s_nop 4\n\tglobal_load_dwordx3 %0, %1, %2" : "=v"(R.w[0][dt]) : "v"(o0), "s"(fbu) : "memory");
                asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(R.w[1][dt]) : "v"(o1), "s"(fbu) : "memory");
            }
        };
        // u8 -> fp16 WITHOUT arithmetic: byte n zero-extended to 16 bits IS the fp16 subnormal n * 2^-24, exact for
        // 0..255, and the MFMA takes fp16 subnormals at full rate.  Every product and partial sum is then the one of
        // the integer-valued formulation times 2^-24 exactly; the epilogue scale carries the 2^24 (a power of two:
        // bit-identical results).  One v_perm_b32 places two bytes -> 8 VALU ops per 16-element pixel slot instead
        // of ~40 (cvt_f32_ubyte + cvt_f16_f32 + pack); the loader waves share their SIMDs with the MFMA waves and
        // their VALU time is what the tile time was waiting for (JG_CONV1_TL=1).
        // slot element k = 3*dt + c (dt = frame 0..4, c = channel), k = 15: the bias lane, 2^-24 (= 1.0 * 2^-24).
        // Zero tiles: the reference blanks the face region of every frame (inference_embs.py:264,270: rows 0..y2+15,
        // ~40 % of the crop).  A tile whose 16 x 100 x 5 source pixels are all zero contributes nothing but the bias
        // slots: the MFMA waves then run 2 of the 49 slots (bit-identical: the other 47 add exact zeros), and a tile
        // buffer that already holds a zero image is not rewritten.  Decided per tile from the bytes just loaded.
        // Counted wait for the frame loads of register set R (issued two iterations ago), by hand (see issue()).  Program order of
        // this wave's VMEM operations per iteration: pooled-row stores (>= 1: pool() issues its output store unconditionally),
        // then the ten loads of one set.  Younger than R's loads are therefore the stores of the last iteration and the OTHER set's
        // loads: vmcnt(11) retires R and leaves those in flight; the stores of THIS iteration are issued behind the wait.
        // Further stores (edge exports, skipped-tile fills) only make the wait retire R earlier than needed, never later than
        // safe.  The first two iterations have no store behind R: vmcnt(10); the prologue's first tile: vmcnt(0).  The empty asm
        // makes the registers opaque so that no consumer is scheduled above the wait (cdna_hip_programming.md 5.7).
        auto wait_frames = [&](C1Regs& R, int level) {
            if (level == 2) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else if (level == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) asm volatile("" : "+v"(R.w[u][dt]));
        };
        bool zero_image[2] = {false, false};       // this wave's part of tile buffer b currently holds an all-zero tile
        int* flags = reinterpret_cast<int*>(smem + OFF_INIT);
        // known_zero: the pre-scan found this tile's input band zero in all five frames (Walk::z) -- then nothing needs to be
        // looked at here.  Without the scan's table (zmask == nullptr) the loaded bytes are OR-ed as in round 1: 30 VALU
        // instructions + a ballot per tile on the role that sets the tile time.
        auto cvt_write = [&](const C1Regs& R, char* buf, int slot, bool known_zero) {
            bool any;
            if (use_skip) {
                any = !known_zero;
            } else {
                uint32_t nz = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int dt = 0; dt < 5; ++dt) nz |= R.w[u][dt][0] | R.w[u][dt][1] | R.w[u][dt][2];
                any = !a.zskip || __builtin_amdgcn_ballot_w64(nz != 0) != 0;       // wave-uniform
            }
            if (lane == 0) flags[slot * 4 + (wave - 4)] = any ? 1 : 0;
            if (!any && zero_image[slot]) return;
            zero_image[slot] = !any;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !has1) break;
                const int row = u ? row1 : row0, g = u ? g1 : g0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t d[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int k0 = 2 * m, k1 = 2 * m + 1;
                        const int b0 = 3 * q + k0 % 3;                               // byte of the 12-byte chunk of frame k0/3
                        const uint32_t lo = R.w[u][k0 / 3][b0 >> 2];
                        uint32_t hi = 1u;                                            // k = 15: the constant 0x0001
                        int hb = 0;
                        if (k1 < 15) {
                            const int b1 = 3 * q + k1 % 3;
                            hi = R.w[u][k1 / 3][b1 >> 2];
                            hb = b1 & 3;
                        }
                        // v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0c = 0x00
                        d[m] = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | ((4u + hb) << 16) | (uint32_t)(b0 & 3));
                    }
                    const int x = 4 * g + q;
                    char* dst = buf + row * ROW_PITCH + slot_off(x);
                    *reinterpret_cast<uint4*>(dst) = uint4{d[0], d[1], d[2], d[3]};
                    *reinterpret_cast<uint4*>(dst + 16) = uint4{d[4], d[5], d[6], d[7]};
                }
            }
        };
        // pooling of local tile t (its conv rows are in conv buffer t&1):
        //   pooled row 2rt-1 = hpool(max(carry, R0)),  pooled row 2rt = hpool(max(R0,R1,R2)),  carry' = max(R2,R3)
        const int pcg = (ltid >> 4) & 7, ppw = ltid & 15;                         // part A: 2 x 8 x 16 threads
        f16* const dump = a.dump + ((long)blockIdx.x * 256 + ltid) * 8;
        // relu(bias) of this thread's 8 channels: what every conv1 output over an all-zero patch is (loaded once: no global
        // load may sit inside the tile loop, see wait_frames)
        const f16x8 cz = use_skip ? *reinterpret_cast<const f16x8*>(a.zconst + pcg * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        const int prow = __builtin_amdgcn_readfirstlane(ltid >> 7);               // wave-uniform: waves 4,5 / 6,7
        const int ccol = ltid & 31, ccg = ltid >> 5;                              // part C: 8 x 32 threads
        // fastz: tile t AND the carry above it come from all-zero input tiles, so every conv value involved is the same
        // per-channel constant relu(bias): one LDS read instead of the 3x3 window
        // per-thread part of the output address (halves): pooled row prow, pooled column ppw, channel group pcg
        const int pthr = (prow * PW + ppw) * 64 + pcg * 8;
        // carry_const: the tile above was skipped outright -- nobody wrote its carry, which is the constant cz in every column
        auto pool = [&](int strip, int rt, int t, bool fastz, bool carry_const) {
            const int nf = (int)((unsigned)strip / 5u);
            const int j = strip - nf * 5;
            const char* cbuf = smem + OFF_CONV + (t & 1) * CONV_BYTES;
            const char* cin = smem + OFF_CARRY + ((rt & 1) ^ 1) * CARRY_BYTES;
            char* cout = smem + OFF_CARRY + (rt & 1) * CARRY_BYTES;
            auto at = [&](const char* base, int row, int cg, int col) {
                return *reinterpret_cast<const f16x8*>(base + ((row * 8 + cg) * 32 + col) * 16);
            };
            // three conv columns of the pooling window; for ppw == 15 column 32 belongs to strip j+1
            // (edge fix-up), so column 31 is simply read twice (max is idempotent) -- no divergent branch
            const int c0 = 2 * ppw, c1 = c0 + 1, c2 = c0 + 2 < 32 ? c0 + 2 : 31;
            f16x8 m, cnext;
            if (fastz) {
                m = at(cbuf, 0, pcg, 0);
                cnext = at(cbuf, 0, ccg, 0);
            } else {
                const f16x8 q2 = at(cbuf, 2, ccg, ccol), q3 = at(cbuf, 3, ccg, ccol);
                cnext = max8(q2, q3);
                if (prow == 0) {            // pooled row 2rt-1: carry (conv rows 4rt-2, 4rt-1) and R0
                    const f16x8 a0 = carry_const ? cz : at(cin, 0, pcg, c0), a1 = carry_const ? cz : at(cin, 0, pcg, c1),
                                a2 = carry_const ? cz : at(cin, 0, pcg, c2);
                    const f16x8 b0 = at(cbuf, 0, pcg, c0), b1 = at(cbuf, 0, pcg, c1), b2 = at(cbuf, 0, pcg, c2);
                    m = max8(max8(max8(a0, a1), max8(a2, b0)), max8(b1, b2));
                } else {                    // pooled row 2rt: R0, R1, R2
                    const f16x8 a0 = at(cbuf, 0, pcg, c0), a1 = at(cbuf, 0, pcg, c1), a2 = at(cbuf, 0, pcg, c2);
                    const f16x8 b0 = at(cbuf, 1, pcg, c0), b1 = at(cbuf, 1, pcg, c1), b2 = at(cbuf, 1, pcg, c2);
                    const f16x8 d0 = at(cbuf, 2, pcg, c0), d1 = at(cbuf, 2, pcg, c1), d2 = at(cbuf, 2, pcg, c2);
                    m = max8(max8(max8(a0, a1), max8(a2, b0)), max8(max8(b1, b2), max8(max8(d0, d1), d2)));
                }
            }
            const int ph = 2 * rt - 1 + prow;
            const int pw = 16 * j + ppw;
            // ONE store per pool() call and wave whatever the lanes hold (wait_frames counts on it): lanes without an output
            // (pooled row -1 of the top tile, pooled columns >= 78 of the last strip) write their slot of a dump area instead.
            // Address = wave-uniform 64-bit part (position, pooled row 2rt-1, column tile) + the per-thread constant.
            {
                const long ub = ((long)nf * PH + (2 * rt - 1)) * (PW * 64) + j * (16 * 64);
                f16* dst = (ph >= 0 && pw < PW) ? a.out + ub + pthr : dump;
                __builtin_nontemporal_store(m, reinterpret_cast<f16x8*>(dst));       // streamed: keep the frames in L2
            }
            if (ph >= 0 && j > 0 && ppw == 0) {     // export conv column 0 (vertically pooled) for strip j-1's last pooled column
                f16x8 e;
                if (fastz) e = m;
                else if (prow == 0) e = max8(carry_const ? cz : at(cin, 0, pcg, 0), at(cbuf, 0, pcg, 0));
                else e = max8(max8(at(cbuf, 0, pcg, 0), at(cbuf, 1, pcg, 0)), at(cbuf, 2, pcg, 0));
                *reinterpret_cast<f16x8*>(a.edge + (((long)nf * PH + ph) * 4 + (j - 1)) * 64 + pcg * 8) = e;
            }
            // carry for the next tile of the strip
            *reinterpret_cast<f16x8*>(cout + (ccg * 32 + ccol) * 16) = cnext;
        };
        // pooled rows (and edge exports) of the SKIPPED tiles of a strip: the per-channel constant relu(bias).  Same thread
        // mapping as part A of pool(): one 16-B store per thread and skipped tile, no LDS, no barrier.
        auto fill_skipped = [&](int strip, unsigned skip) {
            if (!skip) return;
            // first pooled row the fill must write: conv2 of THIS position reads pooled rows >= 2 s2 only (conv1_s2_of_mask)
            const int fill_lo = a.fill_partial ? 2 * conv1_s2_of_mask(skip) : 0;
            const int nf = (int)((unsigned)strip / 5u);
            const int j = strip - nf * 5;
            const int pw = 16 * j + ppw;
            for (unsigned m = skip; m; m &= m - 1) {
                const int rt = __builtin_ctz(m);
                const int ph = 2 * rt - 1 + prow;
                if (ph >= fill_lo && pw < PW)
                    // plain store: as nontemporal stores these seven back-to-back 64-byte half-line writes per thread cost
                    // 1.45x the bytes at the memory side (WRITE_SIZE 3.25 vs 2.22 GB per 32 clips)
                    *reinterpret_cast<f16x8*>(a.out + (((long)nf * PH + ph) * PW + pw) * 64 + pcg * 8) = cz;
                if (ph >= fill_lo && j > 0 && ppw == 0)
                    *reinterpret_cast<f16x8*>(a.edge + (((long)nf * PH + ph) * 4 + (j - 1)) * 64 + pcg * 8) = cz;
            }
        };

        int* live = reinterpret_cast<int*>(smem + OFF_LIVE);
        int tli = 0;
        auto mark = [&]() {
            if (DBG && a.tl && blockIdx.x == 0 && wave == 4 && tli < 2000) {
                const unsigned long long c = wall_clock64();
                if (lane == 0) a.tl[2048 + tli] = c;
                ++tli;
            }
        };
        C1Regs RA, RB;
        // ONE walk, the issue walk (it runs three tiles ahead of the loop); the tiles the other steps of an iteration work on are
        // its last positions, kept in a small history: hA = tile t+2, hB = t+1 (image written this iteration), hC = t (the
        // MFMA waves' tile), hD = t-1 (pooled this iteration).  Entering a strip (on_strip) sets up its frame addressing and
        // fills its skipped tiles with the constant -- those stores touch no other tile's rows, so they can go at any time.
        struct Hist { int strip, rt; unsigned skip; bool zero; bool done; };      // zero: the tile's band is zero in all five frames
        auto on_strip = [&](int strip, unsigned skip) {
            strip_setup(strip);
            fill_skipped(strip, skip);
        };
        Walk qi = walk_first(on_strip);
        auto hist = [](const Walk& q) { return Hist{q.strip, q.rt, q.skip, q.rt >= 0 && ((q.z >> q.rt) & 1u) != 0, q.done}; };
        const Hist none = {0, 0, 0u, false, true};
        Hist hA = none, hB = none, hC = none, hD = none;
        // issue the frame loads of the walk's tile and advance it.  The loads are UNCONDITIONAL (past the end the last tile is
        // simply loaded again: under an `if` the loaded registers become a phi with their old values and hipcc resolves it with
        // copies right behind the loads); the history still records `done`.
        int last_rt = 0;
        auto issue_next = [&](C1Regs& R) {
            hD = hC; hC = hB; hB = hA;
            hA = hist(qi);
            if (!qi.done) last_rt = qi.rt;
            issue(last_rt, R);
            // the strip state (fb0, fdelta, ioff) must stay that of the LAST tile once the walk is done: walk_next() only calls
            // on_strip() for a strip that exists
            walk_next(qi, on_strip);
        };
        if (ltid == 0) { live[0] = qi.done ? 0 : 1; live[1] = 0; }
        if (DBG && (dbg & 1)) {
            Walk q = qi;
            int t = 0;
            while (!q.done) {
                __syncthreads();
                walk_next(q, no_fill);
                if (ltid == 0) live[(t + 1) & 1] = q.done ? 0 : 1;
                ++t;
            }
            __syncthreads();
            return;
        }
        const bool any_tile = !qi.done;
        issue_next(RA);                         // tile 0 (or nothing: the loads are harmless)
        if (any_tile) {
            wait_frames(RA, 0);
            cvt_write(RA, smem, 0, hA.zero);
        }
        issue_next(RB);                         // tile 1
        issue_next(RA);                         // tile 2        now hA = 2, hB = 1, hC = 0
        // iteration t: after the barrier the MFMA waves read tile buffer t&1 and write conv buffer t&1.  We
        //   (1) fill tile buffer (t+1)&1 from the registers loaded TWO ITERATIONS ago,
        //   (2) reload the same registers with tile t+3,
        //   (3) pool tile t-1 (its stores go last).
        // Two register sets, so the loop is unrolled by two (a runtime-selected set would be a phi again).
        // zero status (flags, see cvt_write) of the tiles t-1 and t-2: final once the barrier of their iteration is passed
        bool z1 = false, z2 = false;
        auto tile_is_zero = [&](int tt) -> bool {
            const int4 fl = *reinterpret_cast<const int4*>(smem + OFF_INIT + (tt & 1) * 16);
            return __builtin_amdgcn_readfirstlane(fl.x | fl.y | fl.z | fl.w) == 0;
        };
        // fastz for the tile hD: it is a zero tile and so is the tile above it (the previous tile of the walk, or a skipped
        // one, or there is none) -- every conv value in its pooling windows is then the per-channel constant
        auto pool_step = [&](int tprev) {
            const bool above_skipped = hD.rt > 0 && ((hD.skip >> (hD.rt - 1)) & 1u);
            const bool above_zero = hD.rt == 0 || above_skipped || z2;
            pool(hD.strip, hD.rt, tprev, z1 && above_zero, above_skipped);
        };
        int t = 0;
        while (!hC.done) {                      // hC = tile t
            mark();
            __syncthreads();
            mark();
            const bool zc0 = tile_is_zero(t);          // before cvt_write reuses the other slot; this slot is rewritten at t+1
            if (ltid == 0) live[(t + 1) & 1] = hB.done ? 0 : 1;
            wait_frames(RB, (t >= 2 && !dbg) ? 2 : 1);
            if (!hB.done && !(dbg & 8)) cvt_write(RB, smem + ((t + 1) & 1) * TILE_BYTES, (t + 1) & 1, hB.zero);
            mark();
            if (t > 0 && !(dbg & 4)) pool_step(t - 1);          // hD = tile t-1; its stores go BEFORE the loads (wait_frames)
            mark();
            issue_next(RB);                     // tile t+3; the history shifts: hC = t+1, hD = t
            z2 = z1; z1 = zc0;
            ++t;
            if (hC.done) break;
            mark();
            __syncthreads();
            mark();
            const bool zc1 = tile_is_zero(t);
            if (ltid == 0) live[(t + 1) & 1] = hB.done ? 0 : 1;
            wait_frames(RA, (t >= 2 && !dbg) ? 2 : 1);
            if (!hB.done && !(dbg & 8)) cvt_write(RA, smem + ((t + 1) & 1) * TILE_BYTES, (t + 1) & 1, hB.zero);
            mark();
            if (!(dbg & 4)) pool_step(t - 1);
            mark();
            issue_next(RA);
            z2 = z1; z1 = zc1;
            ++t;
        }
        // The last frame loads (re-loads of the final tile) are never consumed: their registers are dead from here on and hipcc,
        // which does not know the asm loads are in flight, is free to reuse them -- a late return would then overwrite an address
        // of the final pool().  Drain them first.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // the MFMA waves have finished the last tile (hD = tile t-1)
        if (t > 0 && !(dbg & 4)) pool_step(t - 1);
        return;
    }

    // =========================== MFMA waves ===========================
    const int chalf = wave & 1, mb0 = (wave >> 1) & 1;
    if constexpr (M16) {
        // ---- v_mfma_f32_16x16x32_f16: a wave's block (conv row mb, 32 columns, 32 channels) = 2 channel blocks (ob) x 2 column
        // blocks (cb) of 16x16, K = 25 steps of 32 = two pixel slots each.  Lane (n = lane & 15, kq = lane >> 4) holds the k-slice
        //   slot = kq & 1 ? sB(t) : sA(t),   halves 8 * (kq >> 1) .. + 7
        // of step t, for the weights (A operand, registers) and for the patch (B operand, one ds_read_b128 from the tile image).
        // Pairing: (kh, kw) with (kh + 1, kw) for kh = 0, 2, 4 (21 steps, the B slot is one image row below: + ROW_PITCH), then in the
        // last kernel row kw (1,2), (4,5) (+ 32 B), (0,6) (+ 224 B) and kw = 3 alone (the B slot's weights are zero).  One base
        // register per distance and compile-time immediates for all the rest, as in the 32x32x16 path.
        // Bank conflicts: a ds_read_b128 group is {kq = 0 lanes n = 0-3, 12-15} + {kq = 1 lanes n = 4-11} (and its three images), so
        // the two slots of a step sit side by side in one group: the slot distance moves the 16-B bank quad by an EVEN amount in
        // all pairs (3744 / 16 = 234, 32 / 16 = 2, 224 / 16 = 14), and lane n's conv column is permuted so that n = 4..11 take the
        // even columns of the 16-column block and the others the odd ones (quad = 7 * column mod 16): conflict-free.
        const int n16 = lane & 15, kq = lane >> 4;
        const int sel = kq & 1, half = kq >> 1;
        const int colp = (n16 >= 4 && n16 < 12) ? 2 * (n16 - 4) : (n16 < 4 ? 2 * n16 + 1 : 2 * (n16 - 8) + 1);
        auto step_slots = [](int t, int& kha, int& kwa, int& khb, int& kwb) {
            if (t < 21) { kha = 2 * (t / 7); kwa = t % 7; khb = kha + 1; kwb = kwa; }
            else if (t == 21) { kha = khb = 6; kwa = 1; kwb = 2; }
            else if (t == 22) { kha = khb = 6; kwa = 4; kwb = 5; }
            else if (t == 23) { kha = khb = 6; kwa = 0; kwb = 6; }
            else { kha = khb = 6; kwa = 3; kwb = 4; }          // t = 24: slot B is padding: zero weights, and its lanes read slot (6,4) -- a
                                                               // real slot (finite values; the 16-B pads between slots are never written)
        };
        f16x8 wreg[25][2];
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            int kha, kwa, khb, kwb;
            step_slots(t, kha, kwa, khb, kwb);
            const int sl = sel ? khb * 7 + kwb : kha * 7 + kwa;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                wreg[t][ob] = *reinterpret_cast<const f16x8*>(a.Wd + ((long)(sl * 64 + chalf * 32 + ob * 16 + n16) * 16 + 8 * half));
                if (t == 24 && sel) wreg[t][ob] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        // lane part of the patch address: column block cb adds 16 columns = 1792 B, the step its slot-A offset; slot B = + sel * distance
        const int lb = 112 * colp + 16 * half;
        const int lb_row = lb + sel * ROW_PITCH, lb_32 = lb + sel * 32, lb_224 = lb + sel * 224;
        constexpr int DEPTH = 4;                             // patch fragments in flight per wave (2 steps x 2 column blocks; 6 spill)
        int tli = 0;
        auto mark = [&]() {
            if (DBG && a.tl && blockIdx.x == 0 && wave == 0 && tli < 2000) {
                const unsigned long long c = wall_clock64();
                if (lane == 0) a.tl[tli] = c;
                ++tli;
            }
        };
#ifdef JG_CLOCK_STAMPS
        const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int t = 0;; ++t) {
            mark();
            __syncthreads();
            mark();
            if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem + OFF_LIVE + (t & 1) * 4)) == 0) break;
            if (dbg & 2) continue;
            const char* cur = smem + (t & 1) * TILE_BYTES + 3 * mb0 * ROW_PITCH;
            char* cbuf = smem + OFF_CONV + (t & 1) * CONV_BYTES;
            const char* p_row = cur + lb_row;
            const char* p_32 = cur + lb_32;
            const char* p_224 = cur + lb_224;
            // fragment gi of the tile's stream: (block q, step st, column block cb)
            auto frag = [&](int gi) -> f16x8 {
                const int q = gi / 50, st = (gi - q * 50) >> 1, cb = gi & 1;
                int kha, kwa, khb, kwb;
                step_slots(st, kha, kwa, khb, kwb);
                const int off = (6 * q + kha) * ROW_PITCH + 32 * kwa + 16 * (kwa / 3) + 1792 * cb;
                const char* pb = st < 21 ? p_row : (st == 23 ? p_224 : p_32);
                return *reinterpret_cast<const f16x8*>(pb + off);
            };
            // D of a 16x16 block: lane (n, kq) holds channels 4 kq .. 4 kq + 3 of the channel block for column n.
            // conv buffer [row mb][channel group][col][16 B]: group = chalf*4 + ob*2 + (kq >> 1), the lane's 4 channels = 8 B at + 8 (kq & 1)
            auto epilogue = [&](const f32x4 (&acc)[2][2], int mb) {
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const f32x4 v = acc[ob][cb] * a.scale;
                        const f16x4 hv = {(f16)fmaxf(v.x, 0.f), (f16)fmaxf(v.y, 0.f), (f16)fmaxf(v.z, 0.f), (f16)fmaxf(v.w, 0.f)};
                        *reinterpret_cast<f16x4*>(cbuf + ((mb * 8 + chalf * 4 + ob * 2 + (kq >> 1)) * 32 + cb * 16 + colp) * 16 + 8 * (kq & 1)) = hv;
                    }
            };
            const int4 fl = *reinterpret_cast<const int4*>(smem + OFF_INIT + (t & 1) * 16);
            const bool zero_tile = __builtin_amdgcn_readfirstlane(fl.x | fl.y | fl.z | fl.w) == 0;
            if (zero_tile) {
                // all-zero tile (see cvt_write): only slots (0,0) and (0,1) carry anything -- the bias pair on the pad lane: steps 0 and 1
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 acc[2][2];
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) acc[ob][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
                            const f16x8 f = frag(q * 50 + st * 2 + cb);
#pragma unroll
                            for (int ob = 0; ob < 2; ++ob) acc[ob][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[st][ob], f, acc[ob][cb], 0, 0, 0);
                        }
                    epilogue(acc, mb0 + 2 * q);
                }
                continue;
            }
            f16x8 fr[DEPTH];
#pragma unroll
            for (int gi = 0; gi < DEPTH; ++gi) fr[gi] = frag(gi);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 acc[2][2];
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) acc[ob][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 25; ++st)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const int gi = q * 50 + st * 2 + cb;
                        acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[st][0], fr[gi % DEPTH], acc[0][cb], 0, 0, 0);
                        acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[st][1], fr[gi % DEPTH], acc[1][cb], 0, 0, 0);
                        if (gi + DEPTH < 100) fr[gi % DEPTH] = frag(gi + DEPTH);
                        __builtin_amdgcn_sched_barrier(0);       // pin the order (see the 32x32x16 path)
                    }
                epilogue(acc, mb0 + 2 * q);
            }
        }
#ifdef JG_CLOCK_STAMPS
        if (wave == 0 && lane == 0 && blockIdx.x < 1024) {
            jg_clock_stamps_conv1[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - ck0;
            jg_clock_stamps_conv1[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        }
#endif
        return;
    }
    const int r = lane & 31, h = lane >> 5;
    f16x8 wreg[49];
#pragma unroll
    for (int s = 0; s < 49; ++s)
        wreg[s] = *reinterpret_cast<const f16x8*>(a.Wd + ((long)(s * 64 + chalf * 32 + r) * 16 + 8 * h));
    // patch-fragment address of lane (r,h) for slot (kh,kw): pixel x = 3r+kw ->
    //   (3*mb+kh)*ROW_PITCH + 32*x + 16*(x/3) + 16*h = [112*r + 16*h] + [kh*ROW_PITCH + 32*kw + 16*(kw/3)]
    const int lbase = 112 * r + 16 * h;
    constexpr int DEPTH = 6;                             // patch fragments in flight per wave
    int tli = 0;
    auto mark = [&]() {
        if (DBG && a.tl && blockIdx.x == 0 && wave == 0 && tli < 2000) {
            const unsigned long long c = wall_clock64();
            if (lane == 0) a.tl[tli] = c;
            ++tli;
        }
    };
    for (int t = 0;; ++t) {
        mark();
        __syncthreads();
        mark();
        // live[t & 1]: written by the loaders before this barrier; 0 = the workgroup's tiles are done (this was the final barrier)
        if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem + OFF_LIVE + (t & 1) * 4)) == 0) break;
        if (dbg & 2) continue;
        const char* cur = smem + (t & 1) * TILE_BYTES;
        char* cbuf = smem + OFF_CONV + (t & 1) * CONV_BYTES;
        // The two blocks of a tile form ONE stream of 98 (block, slot) steps with DEPTH fragments in
        // flight, so block 1's first fragments are already loading while block 0's epilogue runs.
        const char* base = cur + 3 * mb0 * ROW_PITCH + lbase;
        auto frag = [&](int gi) -> f16x8 {
            const int q = gi / 49, s = gi - q * 49;
            const int kh = s / 7, kw = s - kh * 7;
            return *reinterpret_cast<const f16x8*>(base + (6 * q + kh) * ROW_PITCH + 32 * kw + 16 * (kw / 3));
        };
        // D[i][jj]: jj = lane&31 -> conv column r, i = (x&3) + 8*(x>>2) + 4*h -> channel 32*chalf + 8g + 4h + (x&3).
        // conv buffer [row mb][channel group chalf*4+g][col r][16 B], this lane's 4 channels = 8 B at +8h
        auto epilogue = [&](const f32x16& acc, int mb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                v *= a.scale;
                f16x4 hv = {(f16)fmaxf(v.x, 0.f), (f16)fmaxf(v.y, 0.f), (f16)fmaxf(v.z, 0.f), (f16)fmaxf(v.w, 0.f)};
                *reinterpret_cast<f16x4*>(cbuf + ((mb * 8 + chalf * 4 + g) * 32 + r) * 16 + 8 * h) = hv;
            }
        };
        const int4 fl = *reinterpret_cast<const int4*>(smem + OFF_INIT + (t & 1) * 16);
        if (__builtin_amdgcn_readfirstlane(fl.x | fl.y | fl.z | fl.w) == 0) {
            // all-zero tile (see cvt_write): only slots 0 and 1 carry anything -- the bias pair on the pad lane
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x16 acc;
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[x] = 0.f;
                const f16x8 f0 = frag(q * 49), f1 = frag(q * 49 + 1);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[0], f0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[1], f1, acc, 0, 0, 0);
                epilogue(acc, mb0 + 2 * q);
            }
            continue;
        }
        f16x8 fr[DEPTH];
#pragma unroll
        for (int gi = 0; gi < DEPTH; ++gi) fr[gi] = frag(gi);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int mb = mb0 + 2 * q;
            f32x16 acc;
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[x] = 0.f;
#pragma unroll
            for (int s = 0; s < 49; ++s) {
                const int gi = q * 49 + s;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], fr[gi % DEPTH], acc, 0, 0, 0);
                if (gi + DEPTH < 98) fr[gi % DEPTH] = frag(gi + DEPTH);
                // pin the order: without this hipcc sinks every ds_read next to its MFMA (one fragment
                // register, lgkmcnt(0) per MFMA) and the LDS latency is exposed 49 times per block
                __builtin_amdgcn_sched_barrier(0);
            }
            epilogue(acc, mb);
        }
    }
}

// out[nf][ph][16j+15][:] = max(out[...], edge[nf][ph][j][:])  for j = 0..3 (pooled columns 15,31,47,63)
__global__ void conv1_edge_fix_kernel(f16* __restrict__ out, const f16* __restrict__ edge, long n) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;     // (nf*43+ph, j, cg)
    if (idx >= n) return;
    const int cg = idx & 7;
    const int j = (idx >> 3) & 3;
    const long rowi = idx >> 5;
    f16* o = out + (rowi * PW + 16 * j + 15) * 64 + cg * 8;
    const f16x8 e = *reinterpret_cast<const f16x8*>(edge + (rowi * 4 + j) * 64 + cg * 8);
    *reinterpret_cast<f16x8*>(o) = max8(*reinterpret_cast<const f16x8*>(o), e);
}

// Per source frame: which 16-row input bands (the 22 row tiles of conv1_direct_kernel) are entirely zero.  The reference blanks
// the face region of every crop (inference_embs.py:264,270); such bands need no conv1 work at all.  One workgroup per frame:
//   1. probe -- thread r reads 16 B of row r at a row-dependent column: a natural or noisy row is almost surely non-zero
//      there, so only rows whose probe is zero are read completely (HBM traffic ~ the zero rows + 4 % of the rest),
//   2. full check of the candidate rows, one wave per row, coalesced 16-B loads,
//   3. band flags -> 22-bit mask.
// Block 0 also computes the constant every all-zero patch produces: relu(bias) as the MFMA path rounds it (conv1 bias = the
// hi+lo pair on the pad lane of slots 0 and 1, times 2^-24; both products and their sum are exact in fp32).
// relu(bias) of channel c as the MFMA path rounds it: the bias is the hi+lo pair on the pad lane of slots 0 and 1, times the pad
// lane's "1.0" = 2^-24; both products and their sum are exact in fp32
__device__ __forceinline__ f16 conv1_zero_patch_value(const f16* __restrict__ Wd, float scale, int c) {
    const float hi = (float)Wd[(0 * 64 + c) * 16 + 15], lo = (float)Wd[(1 * 64 + c) * 16 + 15];
    const float acc = hi * 5.9604644775390625e-8f + lo * 5.9604644775390625e-8f;
    return (f16)fmaxf(acc * scale, 0.f);
}
__global__ void conv1_zconst_kernel(const f16* __restrict__ Wd, float scale, f16* __restrict__ zconst) {
    if (threadIdx.x < 64) zconst[threadIdx.x] = conv1_zero_patch_value(Wd, scale, threadIdx.x);
}

__device__ __forceinline__ int* zmask_hdr(f16* zconst) { return reinterpret_cast<int*>(zconst); }   // zconst = word 0 of the header

__global__ __launch_bounds__(256) void conv1_zero_scan_kernel(const uint8_t* __restrict__ src, unsigned* __restrict__ zmask,
                                                              const f16* __restrict__ Wd, float scale, f16* __restrict__ zconst) {
    __shared__ int rowz[IH + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* fb = src + (size_t)blockIdx.x * (size_t)(IH * IW * 3);
    for (int r = tid; r < IH; r += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(fb + (size_t)r * (IW * 3) + ((r * 37) % 90) * 16);
        rowz[r] = (v.x | v.y | v.z | v.w) == 0 ? 1 : 0;
    }
    __syncthreads();
    for (int r = wave; r < IH; r += 4) {
        if (!rowz[r]) continue;                                      // wave-uniform
        const uint8_t* row = fb + (size_t)r * (IW * 3);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        // a row that turns out to be zero is never read again (conv1 skips it): keep it out of the caches
        u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + lane * 16));
        unsigned nz = v.x | v.y | v.z | v.w;
        if (lane < 26) {
            v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + 1024 + lane * 16));
            nz |= v.x | v.y | v.z | v.w;
        }
        if (__builtin_amdgcn_ballot_w64(nz != 0) != 0 && lane == 0) rowz[r] = 0;
    }
    __syncthreads();
    if (wave == 0) {
        bool z = lane < ROW_TILES;
        if (z) {
            for (int r = 12 * lane; r < 12 * lane + TROWS; ++r) z = z && (r >= IH || rowz[r]);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(z);
        if (lane == 0) zmask[blockIdx.x] = (unsigned)m & ((1u << ROW_TILES) - 1u);
    }
    if (blockIdx.x == 0 && tid < 64) zconst[tid] = conv1_zero_patch_value(Wd, scale, tid);
    // reset the launch-wide minimum that conv1_skip_mask_kernel (next on the stream) reduces into
    if (blockIdx.x == 0 && tid == 64) zmask_hdr(zconst)[CONV1_ROWSKIP_WORD] = 0x7fffffff;
}

// position nf = (clip b, padded-clip position p) reads frames clamp(p + dt - pad), dt = 0..4: a tile is all-zero for the position
// when its band is zero in all five; it is SKIPPED when the tile above is all-zero too (or does not exist): both of its pooled
// rows (2rt-1: carry of the tile above and its own row 0; 2rt: its rows 0..2) are then the constant.  Its own carry is the
// constant as well, which the tile below -- if that one runs -- takes from cz instead of LDS (pool(), carry_const).
// It also writes s2[nf], conv2's position-independent leading output rows of the position (conv1_s2_of_mask): the conv2 GEMM
// computes rows >= s2[nf] of that position only, and so on down the stack (ConvGeom::rowmap, the const chain of api.hip).
// *rowskip_min = min over the launch (a debug word: jg_debug_conv2_rowskip).
__global__ void conv1_skip_mask_kernel(const unsigned* __restrict__ fz, int nclip, int T, int pad, int P, unsigned* __restrict__ skip,
                                       int* __restrict__ s2, int* __restrict__ rowskip_min) {
    const int nf = blockIdx.x * blockDim.x + threadIdx.x;
    int rs = 0x7fffffff;
    if (nf < nclip * P) {
        const int b = nf / P, p = nf - b * P;
        unsigned z = (1u << ROW_TILES) - 1u;
        for (int dt = 0; dt < 5; ++dt) {
            int f = p + dt - pad;
            f = f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
            z &= fz[b * T + f];
        }
        const unsigned sk = z & ((z << 1) | 1u);
        skip[nf] = z;                                    // the kernel derives sk itself and uses z for the tiles that still run
        rs = conv1_s2_of_mask(sk);
        s2[nf] = rs;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(rs, d, 64);
        rs = o < rs ? o : rs;
    }
    if ((threadIdx.x & 63) == 0 && rs != 0x7fffffff) atomicMin(rowskip_min, rs);
}

// ---- compaction maps of the conv layers behind conv1 (common.h: ConvGeom::rowmap, ConvRowMap) ----------------------------------
// Image (position) img computes rows >= s = conv_skip_decode(s2[img], op) of a layer's OH x OW output.  Step 1, one workgroup:
// exclusive prefix of the images' computed pixels per layer (base[img], *total).  Step 2, one workgroup per image: its computed
// pixels (one contiguous run of full indices per layer) go to their compacted place, | s2 << 24.
struct RowMapArgs {
    const int* s2;
    int NF, nl;
    ConvRowMap L[4];
};
__global__ __launch_bounds__(1024) void conv_rowmap_scan_kernel(RowMapArgs a) {
    __shared__ int part[4][1024];
    const int tid = threadIdx.x;
    const int per = (a.NF + 1023) / 1024;                  // consecutive images per thread
    const int i0 = tid * per, i1 = i0 + per < a.NF ? i0 + per : a.NF;
    int sum[4] = {0, 0, 0, 0};
    for (int i = i0; i < i1; ++i) {
        const int w = a.s2[i];
#pragma unroll
        for (int l = 0; l < 4; ++l) sum[l] += (a.L[l].OH - conv_skip_decode(w, a.L[l].op)) * a.L[l].OW;
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) part[l][tid] = sum[l];
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                   // Hillis-Steele inclusive scan, the four layers side by side
        int v[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) v[l] = tid >= d ? part[l][tid - d] : 0;
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 4; ++l) part[l][tid] += v[l];
        __syncthreads();
    }
    int run[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) run[l] = part[l][tid] - sum[l];
    for (int i = i0; i < i1; ++i) {
        const int w = a.s2[i];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            if (l < a.nl) a.L[l].base[i] = run[l];
            run[l] += (a.L[l].OH - conv_skip_decode(w, a.L[l].op)) * a.L[l].OW;
        }
    }
    if (tid == 1023) {
#pragma unroll
        for (int l = 0; l < 4; ++l)
            if (l < a.nl) { a.L[l].base[a.NF] = part[l][1023]; *a.L[l].total = part[l][1023]; }
    }
}
// one workgroup per image: its computed pixels of every layer, in order
__global__ __launch_bounds__(256) void conv_rowmap_fill_kernel(RowMapArgs a) {
    const int img = blockIdx.x;
    const int w = a.s2[img];
    for (int l = 0; l < a.nl; ++l) {
        const ConvRowMap& R = a.L[l];
        const int s = conv_skip_decode(w, R.op);
        const int n = (R.OH - s) * R.OW;                   // computed pixels of this image: full rows s*OW .. OH*OW - 1
        const int first = img * R.OH * R.OW + s * R.OW;
        int* dst = R.map + R.base[img];
        for (int i = threadIdx.x; i < n; i += 256) dst[i] = (first + i) | (w << 24);
    }
}

// workspace words: header of CONV1_ZHDR_WORDS (zconst: 64 halves = 32 words; word CONV1_ROWSKIP_WORD: constant leading rows of
// conv2's output) + nclip*T (frame masks) + nclip*(T+2*pad-4) (position skip masks)
// + nclip*(T+2*pad-4) (per-position counts s2); sized for pad <= 12
size_t conv1_zmask_elems(int nclip, int T) { return (size_t)CONV1_ZHDR_WORDS + (size_t)nclip * T + 2 * (size_t)nclip * (T + 20); }
const int* conv1_s2_counts(const unsigned* zscratch, int nclip, int T, int pad) {
    return reinterpret_cast<const int*>(zscratch + CONV1_ZHDR_WORDS + (size_t)nclip * T + (size_t)nclip * (T + 2 * pad - 4));
}

hipError_t launch_conv_rowmaps(const int* s2, int NF, const ConvRowMap* layers, int nlayers, hipStream_t s) {
    if (NF <= 0 || nlayers <= 0 || nlayers > 4) return hipErrorInvalidValue;
    RowMapArgs a;
    a.s2 = s2; a.NF = NF; a.nl = nlayers;
    for (int l = 0; l < 4; ++l) {
        a.L[l] = layers[l < nlayers ? l : nlayers - 1];
        if ((long)NF * a.L[l].OH * a.L[l].OW >= (1L << 24)) return hipErrorInvalidValue;      // 24-bit row index in the map entries
    }
    hipLaunchKernelGGL(conv_rowmap_scan_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(conv_rowmap_fill_kernel, dim3((unsigned)NF), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Zero-band scan + per-position skip masks + the zero-patch constant into `zscratch` (conv1_zmask_elems words).
hipError_t launch_conv1_scan(const uint8_t* src, int nclip, int T, int pad, const f16* Wd, float scale, unsigned* zscratch, hipStream_t s) {
    if (nclip * T <= 0) return hipSuccess;
    const int P = T + 2 * pad - 4;
    unsigned* fz = zscratch + CONV1_ZHDR_WORDS;
    unsigned* sk = fz + (size_t)nclip * T;
    f16* zc = reinterpret_cast<f16*>(zscratch);
    hipLaunchKernelGGL(conv1_zero_scan_kernel, dim3((unsigned)(nclip * T)), dim3(256), 0, s, src, fz, Wd, scale * 16777216.0f, zc);
    hipLaunchKernelGGL(conv1_skip_mask_kernel, dim3((unsigned)((nclip * P + 255) / 256)), dim3(256), 0, s, fz, nclip, T, pad, P, sk,
                       reinterpret_cast<int*>(sk + (size_t)nclip * P), reinterpret_cast<int*>(zscratch) + CONV1_ROWSKIP_WORD);
    return hipGetLastError();
}

// zscratch: filled by launch_conv1_scan (nullptr: no tile is skipped outright).  The 4 pooled columns that straddle two strips
// are closed afterwards by launch_conv1_edge_fix.
hipError_t launch_conv1_zconst(const f16* Wd, float scale, f16* zconst, hipStream_t s) {
    hipLaunchKernelGGL(conv1_zconst_kernel, dim3(1), dim3(64), 0, s, Wd, scale * 16777216.0f, zconst);
    return hipGetLastError();
}

hipError_t launch_conv1_direct(const uint8_t* src, int nclip, int T, int pad, const f16* Wd, float scale,
                               f16* out_pooled, f16* edge, const unsigned* zscratch, bool fill_all, const EngineOpts& o, hipStream_t s) {
    static bool attr_set[64] = {};
    if (o.device < 0 || o.device >= 64) return hipErrorInvalidDevice;
    const int num_cu = o.num_cu;
    if (!attr_set[o.device]) {
        const void* ks[3] = {reinterpret_cast<const void*>(conv1_direct_kernel<false, false>), reinterpret_cast<const void*>(conv1_direct_kernel<true, false>),
                             reinterpret_cast<const void*>(conv1_direct_kernel<false, true>)};
        for (const void* kf : ks) {
            hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set[o.device] = true;
    }
    Conv1Args a;
    a.src = src; a.nclip = nclip; a.T = T; a.pad = pad; a.P = T + 2 * pad - 4;
    a.dump = edge + (size_t)nclip * (T + 2 * pad - 4) * PH * 4 * 64;
    a.Wd = Wd; a.scale = scale * 16777216.0f; a.out = out_pooled; a.edge = edge;      // 2^24: the loaders feed n * 2^-24 (fp16 subnormals)
    if ((long)nclip * a.P >= (1L << 24)) return hipErrorInvalidValue;     // decode() splits positions with a float reciprocal
    a.nstrips = nclip * a.P * COL_TILES;
    a.invP = 1.0f / (float)a.P;
    static const int dbg = getenv("JG_CONV1_DBG") ? atoi(getenv("JG_CONV1_DBG")) : 0;
    a.dbg = dbg;
    a.zskip = o.conv1_zero_skip ? 1 : 0;
    a.zmask = nullptr;
    a.zconst = nullptr;
    a.fill_partial = 0;
    if (num_cu > MAX_WGS) return hipErrorInvalidValue;
    if (a.zskip && zscratch && nclip * T > 0 && (a.nstrips + num_cu - 1) / num_cu + 8 <= MAX_WG_STRIPS) {
        a.zmask = zscratch + CONV1_ZHDR_WORDS + (size_t)nclip * T;
        a.zconst = reinterpret_cast<const f16*>(zscratch);
        if (!fill_all) a.fill_partial = 1;
    }
    static unsigned long long* tl = nullptr;
    static const bool want_tl = getenv("JG_CONV1_TL") != nullptr;
    if (want_tl && !tl && hipHostMalloc(&tl, 4096 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) tl = nullptr;
    a.tl = tl;
    if (tl) {
        (void)hipStreamSynchronize(s);
        std::memset(tl, 0, 4096 * sizeof(unsigned long long));
    }
    if (a.nstrips <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(a.nstrips < num_cu ? a.nstrips : num_cu);
    // (the timeline / ablation build exists for the 32x32x16 form only)
    if (a.dbg || a.tl) hipLaunchKernelGGL((conv1_direct_kernel<true, false>), dim3(grid), dim3(512), LDS_BYTES, s, a);
    else if (o.conv1_mfma16) hipLaunchKernelGGL((conv1_direct_kernel<false, true>), dim3(grid), dim3(512), LDS_BYTES, s, a);
    else hipLaunchKernelGGL((conv1_direct_kernel<false, false>), dim3(grid), dim3(512), LDS_BYTES, s, a);
    if (tl) {
        (void)hipStreamSynchronize(s);
        // MFMA wave 0: per tile [arrive, pass]; loader wave 4: per tile [arrive, pass, loads issued, pooled]
        double wait = 0, work = 0; int n = 0;
        for (int i = 0; i + 2 < 2000 && tl[i + 2]; i += 2, ++n) { wait += (tl[i + 1] - tl[i]) * 0.01; work += (tl[i + 2] - tl[i + 1]) * 0.01; }
        if (n) std::fprintf(stderr, "[conv1 timeline] MFMA wave: %d tiles, barrier wait %.2f us, MFMA+epilogue %.2f us per tile\n", n, wait / n, work / n);
        double w2 = 0, is = 0, po = 0, cv = 0; n = 0;
        const unsigned long long* q = tl + 2048;
        for (int i = 0; i + 4 < 2000 && q[i + 4]; i += 4, ++n) {
            w2 += (q[i + 1] - q[i]) * 0.01; is += (q[i + 2] - q[i + 1]) * 0.01; po += (q[i + 3] - q[i + 2]) * 0.01; cv += (q[i + 4] - q[i + 3]) * 0.01;
        }
        if (n) std::fprintf(stderr, "[conv1 timeline] loader wave: %d tiles, barrier wait %.2f us, convert+fill %.2f us, issue loads %.2f us, pool %.2f us per tile\n", n, w2 / n, is / n, po / n, cv / n);
    }
    return hipGetLastError();
}

hipError_t launch_conv1_edge_fix(f16* out_pooled, const f16* edge, long positions, hipStream_t s) {
    const long n = positions * PH * 4 * 8;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(conv1_edge_fix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out_pooled, edge, n);
    return hipGetLastError();
}

size_t conv1_edge_elems(long positions) { return (size_t)positions * PH * 4 * 64 + (size_t)MAX_WGS * DUMP_HALVES_PER_WG; }
