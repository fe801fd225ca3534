// Decode table of v_cvt_pk_f32_fp8 on this GPU: all 256 byte values (byte b at lane b, low byte of the word / second byte).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__global__ void k(float* out) {
    const unsigned b = threadIdx.x;                       // 0..255
    const unsigned word = b | ((255u - b) << 8) | (b << 16) | ((255u - b) << 24);
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(word, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(word, true);
    out[b * 4 + 0] = lo[0]; out[b * 4 + 1] = lo[1]; out[b * 4 + 2] = hi[0]; out[b * 4 + 3] = hi[1];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * sizeof(float));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    float h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 256; ++b) printf("%d %.9g %.9g %.9g %.9g\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2], h[b * 4 + 3]);
    return 0;
}
