// Stand-alone check of what tools/experiments/xlmr_race narrowed the XLM-RoBERTa two-stream difference down to:
//   v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]   (low half: a.lo * b.HI + c.lo)
// in one kernel while waves of ANOTHER kernel issue v_mfma_f32_32x32x16_f16 on the same SIMD.  In the library the low-half product came
// out as exactly 0 in lanes 48-63 (result = c.lo) in 1 of ~1e5 executions.
//   victim:    512-thread workgroups, a loop of the packed FMA on known operands, every result compared with two scalar v_fma_f32
//   aggressor: 256-thread workgroups, a loop of 32x32x16 MFMAs (+ optional LDS traffic), launched on a second stream of high priority
// Build: hipcc --offload-arch=gfx950 -O2 -o repro repro.hip
// Run:   ./repro [launches] [aggressor: 0 none, 1 mfma 32x32x16, 2 mfma 16x16x32, 3 valu only] [victim form 0..8, +100: MFMA waves inside the victim's kernel]
//                [victim LDS bytes] [aggressor stream of high priority 1 / 0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void victim(int loops, unsigned* bad, float* rec, int mode) {
    extern __shared__ char lds[];
    const int t = threadIdx.x, lane = t & 63;
    // operands that differ per lane and per iteration, exact in fp32 (small integers / powers of two)
    float a0 = 1.0f + lane, a1 = 2.0f + lane;
    const f32x2 b = {0.25f, 3.0f};          // (lo, hi): op_sel picks b.hi = 3 for BOTH halves
    unsigned nbad = 0;
    if (mode >= 100) {                      // the aggressor INSIDE the victim's kernel: waves 4-7 issue MFMAs, waves 0-3 the packed FMAs
        mode -= 100;
        if ((t >> 8) & 1) {                  // waves 4-7: wave w and w + 4 share SIMD w % 4
            f16x8 pa, pb;
            for (int i = 0; i < 8; ++i) { pa[i] = (_Float16)(0.01f * (lane + i)); pb[i] = (_Float16)(0.02f * (lane - i)); }
            f32x16 acc = {0};
            for (int it = 0; it < loops / 8; ++it) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa, pb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb, pa, acc, 0, 0, 0);
            }
            float s_ = 0.f;
            for (int i = 0; i < 16; ++i) s_ += acc[i];
            if (s_ == 123.456f) rec[0] = s_;
            return;
        }
    }
    for (int it = 0; it < loops; ++it) {
        f32x2 a = {a0, a1};
        f32x2 c = {(float)(it & 1023), (float)((it & 1023) + 7)};
        f32x2 d;
        float e0, e1;
        auto fma1 = [](float x, float y, float z) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); return r; };
        auto mul1 = [](float x, float y) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
        auto add1 = [](float x, float y) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
        switch (mode) {
            case 0: asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); e0 = fma1(a.x, b.y, c.x); e1 = fma1(a.y, b.y, c.y); break;
            case 1: asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(f32x2{3.0f, 3.0f}), "v"(c)); e0 = fma1(a.x, 3.0f, c.x); e1 = fma1(a.y, 3.0f, c.y); break;
            case 2: asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); e0 = fma1(a.x, b.x, c.x); e1 = fma1(a.y, b.x, c.y); break;
            case 3: asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b)); e0 = mul1(a.x, b.y); e1 = mul1(a.y, b.y); break;
            case 4: asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); e0 = fma1(a.y, b.x, c.x); e1 = fma1(a.y, b.y, c.y); break;
            case 5: asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); e0 = fma1(a.x, b.x, c.y); e1 = fma1(a.y, b.y, c.y); break;
            case 6: asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(c)); e0 = add1(a.x, c.y); e1 = add1(a.y, c.y); break;
            case 7: asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); e0 = mul1(a.x, b.y); e1 = mul1(a.y, b.x); break;
            // packed fp16 (one 32-bit register per operand: the selects pick halves of a register, not registers of a pair) and v_fma_mix_f32
            case 20: case 21: case 22: case 23: {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 ha = {(_Float16)(float)(lane & 31), (_Float16)(float)((lane & 31) + 1)}, hb = {(_Float16)0.5f, (_Float16)3.0f};
                const h2 hc = {(_Float16)(float)(it & 63), (_Float16)(float)((it & 63) + 7)};
                h2 hd;
                float m0 = 0.f;
                if (mode == 20) { asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(hd) : "v"(ha), "v"(hb), "v"(hc)); e0 = (float)ha.x * 3.0f + (float)hc.x; e1 = (float)ha.y * 3.0f + (float)hc.y; }
                else if (mode == 21) { asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1]" : "=v"(hd) : "v"(ha), "v"(hb)); e0 = (float)ha.x * 3.0f; e1 = (float)ha.y * 3.0f; }
                else if (mode == 22) { asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1]" : "=v"(hd) : "v"(ha), "v"(hc)); e0 = (float)ha.x + (float)hc.y; e1 = (float)ha.y + (float)hc.y; }
                else { asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(m0) : "v"(ha), "v"(hb), "v"(c.x)); hd = h2{(_Float16)0.f, (_Float16)0.f}; e0 = (float)ha.x * 3.0f + c.x; e1 = 0.f; }
                d = mode == 23 ? f32x2{m0, 0.f} : f32x2{(float)hd.x, (float)hd.y};
                break;
            }
            default: asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); e0 = fma1(a.y, b.y, c.y); e1 = fma1(a.x, b.x, c.x); break;
        }
        if (d.x != e0 || d.y != e1) {
            const unsigned k = atomicAdd(bad, 1u);
            if (k < 64) { rec[8 * k] = (float)lane; rec[8 * k + 1] = d.x; rec[8 * k + 2] = e0; rec[8 * k + 3] = d.y; rec[8 * k + 4] = e1; rec[8 * k + 5] = c.x; rec[8 * k + 6] = (float)it; rec[8 * k + 7] = (float)blockIdx.x; }
            ++nbad;
        }
        a0 += 1.0f; a1 += 1.0f;
        if (a0 > 4096.f) { a0 = 1.0f + lane; a1 = 2.0f + lane; }
    }
    if (nbad == 0xffffffffu) lds[t] = 1;      // keeps the dynamic LDS allocation alive
}

__global__ __launch_bounds__(256) void aggressor(int loops, float* sink, int kind) {
    __shared__ float sm[4 * 1152];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
    f32x16 acc = {0};
    f32x4 acc4 = {0, 0, 0, 0};
    float v = lane * 0.5f;
    for (int it = 0; it < loops; ++it) {
        if (kind == 1) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
        } else if (kind == 2) {
            acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4, 0, 0, 0);
            acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc4, 0, 0, 0);
        } else {
            v = __builtin_fmaf(v, 1.0001f, 0.5f);
        }
        if ((it & 15) == 0) {                       // some LDS traffic, as in the attention kernel
            sm[wave * 1152 + lane * 17 % 1152] = acc[it & 15] + acc4[it & 3] + v;
            __builtin_amdgcn_wave_barrier();
            v += sm[wave * 1152 + (lane * 5 + it) % 1152];
        }
    }
    float s = v;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 4; ++i) s += acc4[i];
    if (s == 123.456f) sink[t] = s;
}

static const char* FORMS[] = {"pk_fma op_sel:[0,1,0]", "pk_fma plain", "pk_fma op_sel_hi:[1,0,1]", "pk_mul op_sel:[0,1]", "pk_fma op_sel:[1,0,0]", "pk_fma op_sel:[0,0,1]",
                              "pk_add op_sel:[0,1]", "pk_mul op_sel:[0,1] op_sel_hi:[1,0]", "pk_fma op_sel:[1,1,1] op_sel_hi:[0,0,0]"};
static const char* F16FORMS[] = {"pk_fma_f16 op_sel:[0,1,0]", "pk_mul_f16 op_sel:[0,1]", "pk_add_f16 op_sel:[0,1]", "fma_mix_f32 op_sel:[0,1,0] op_sel_hi:[1,1,0]"};
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int kind = argc > 2 ? atoi(argv[2]) : 1;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int lds = argc > 4 ? atoi(argv[4]) : 96 * 1024;
    const int high = argc > 5 ? atoi(argv[5]) : 1;          // the aggressor's stream: 1 = highest priority (a hardware queue of its own for sure), 0 = normal
    unsigned* bad; float *rec, *sink;
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&rec, 64 * 8 * 4)); CK(hipMalloc(&sink, 1024));
    CK(hipMemset(bad, 0, 4));
    hipStream_t s1, s2;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, high ? hi : lo > 0 ? 0 : lo));
    CK(hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int it = 0; it < iters; ++it) {
        if (kind) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 0, s2, 20000, sink, kind);
        hipLaunchKernelGGL(victim, dim3(256), dim3(512), lds, s1, 200000, bad, rec, mode);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    }
    unsigned nb; std::vector<float> r(64 * 8);
    CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), rec, 64 * 8 * 4, hipMemcpyDeviceToHost));
    printf("aggressor kind %d (%s priority stream), victim form %s, victim LDS %d KB, %d launches x 256 workgroups x 512 threads x 200000 packed FMAs: %u wrong results\n", kind, high ? "high" : "normal", (mode % 100 >= 20 ? F16FORMS[mode % 100 - 20] : FORMS[mode % 100 < 9 ? mode % 100 : 8]), lds / 1024, iters, nb);
    int lanes[4] = {0, 0, 0, 0}, lo_wrong = 0, hi_wrong = 0;
    for (unsigned k = 0; k < (nb < 64 ? nb : 64); ++k) { ++lanes[(int)r[8 * k] >> 4]; lo_wrong += r[8 * k + 1] != r[8 * k + 2]; hi_wrong += r[8 * k + 3] != r[8 * k + 4]; }
    if (nb) printf("  first %u records: lanes 0-15 %d, 16-31 %d, 32-47 %d, 48-63 %d; low half wrong %d, high half wrong %d\n", nb < 64 ? nb : 64, lanes[0], lanes[1], lanes[2], lanes[3], lo_wrong, hi_wrong);
    for (unsigned k = 0; k < (nb < 3 ? nb : 3); ++k)
        printf("  lane %2.0f: lo %g (expected %g) hi %g (expected %g), c.lo %g, iteration %.0f, workgroup %.0f\n", r[8 * k], r[8 * k + 1], r[8 * k + 2], r[8 * k + 3], r[8 * k + 4], r[8 * k + 5], r[8 * k + 6], r[8 * k + 7]);
    return 0;
}
