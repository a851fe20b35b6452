// What does v_cvt_pk_u8_f32 do with fractions, negatives and overflow?  (gfx950; candidate for the 8-bit plane of the token stream)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
int main() {
    float h[] = {0.0f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 127.5f, 254.4f, 254.5f, 254.6f, 255.4f, 255.6f, 300.f, -0.4f, -0.6f, -3.f, 1e9f};
    const int n = sizeof(h) / sizeof(float);
    float* d; unsigned* o; unsigned r[32];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4 * n);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    hipMemcpy(r, o, 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%10.2f -> byte %3u  (word %08x)\n", h[i], (r[i] >> 8) & 0xff, r[i]);
    return 0;
}
