// Micro-check: unaligned global_load_dwordx4 (byte offsets that are multiples of 3) returns the right bytes on gfx950,
// both as a compiler load and as the inline-asm saddr form conv1 uses.  Build: hipcc --offload-arch=gfx950 -O3 -o unaligned_load unaligned_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* src, u32x4* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t off = 15u * i + 3u * (i % 7);
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(off), "s"(src) : "memory");
    out[i] = r;
}
int main() {
    const int n = 1 << 16;
    std::vector<uint8_t> h(16 * n + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7);
    uint8_t* d; u32x4* o;
    hipMalloc(&d, h.size()); hipMalloc(&o, n * 16);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    std::vector<uint8_t> r(n * 16);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) { const uint32_t off = 15u * i + 3u * (i % 7); for (int b = 0; b < 16; ++b) bad += r[i * 16 + b] != h[off + b]; }
    printf("unaligned dwordx4: %ld wrong bytes of %d\n", bad, n * 16);
    return bad != 0;
}
