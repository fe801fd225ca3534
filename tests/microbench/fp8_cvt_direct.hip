// Round 6: v_cvt_scalef32_pk_{bf16,f16}_fp8 (scale 1.0) against the two-step widening (v_cvt_pk_f32_fp8 + f32 -> bf16 / f16) for all 256 e4m3 bytes.
// Prints the number of bytes whose 16-bit results differ (NaN encodings 0x7f / 0xff excepted: reported separately).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__global__ void k(uint32_t* out) {
    const unsigned b = threadIdx.x;                       // 0..255
    const unsigned word = b | ((255u - b) << 8) | (b << 16) | ((255u - b) << 24);
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(word, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(word, true);
    out[b * 8 + 0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2_t));
    out[b * 8 + 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2_t));
    out[b * 8 + 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, 1.0f, false));
    out[b * 8 + 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(word, 1.0f, true));
    out[b * 8 + 4] = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, f16x2_t));
    out[b * 8 + 5] = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, f16x2_t));
    out[b * 8 + 6] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(word, 1.0f, false));
    out[b * 8 + 7] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(word, 1.0f, true));
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * sizeof(uint32_t));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    uint32_t h[2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad_bf = 0, bad_f16 = 0, nan_diff = 0;
    for (int b = 0; b < 256; ++b) {
        const bool nanb = (b & 0x7f) == 0x7f || ((255 - b) & 0x7f) == 0x7f;
        const bool dbf = h[b * 8] != h[b * 8 + 2] || h[b * 8 + 1] != h[b * 8 + 3], df = h[b * 8 + 4] != h[b * 8 + 6] || h[b * 8 + 5] != h[b * 8 + 7];
        if ((dbf || df) && nanb) { ++nan_diff; continue; }
        if (dbf) { ++bad_bf; printf("bf16 differs at byte %d: %08x %08x vs %08x %08x\n", b, h[b * 8], h[b * 8 + 1], h[b * 8 + 2], h[b * 8 + 3]); }
        if (df) { ++bad_f16; printf("f16 differs at byte %d: %08x %08x vs %08x %08x\n", b, h[b * 8 + 4], h[b * 8 + 5], h[b * 8 + 6], h[b * 8 + 7]); }
    }
    printf("direct fp8 -> bf16: %d of 256 words differ; fp8 -> f16: %d differ; words holding a NaN byte that differ: %d\n", bad_bf, bad_f16, nan_diff);
    return 0;
}
