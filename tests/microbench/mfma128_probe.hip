// Probe for the CDNA4 f8f6f4 MFMA (v_mfma[_scale]_f32_16x16x128_f8f6f4), to be run on an MI355X:
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma128_probe tests/microbench/mfma128_probe.hip && /tmp/mfma128_probe
// It checks the operand layout this repo will assume for an fp8 16x16x128 GEMM path (from ck_tile's warp-GEMM attributes for this
// instruction, kAMLane 16 / kABKLane 4 / kABKPerLane 32 / kCMLane 4 / kCM1PerLane 4):
//   A, B: lane l holds row (l & 15), k = 32 (l >> 4) .. + 31, one byte per element, 32 contiguous bytes (8 VGPRs);
//   C/D : lane l holds rows 4 (l >> 4) .. + 3 of column (l & 15)           (as v_mfma_f32_16x16x32_*);
//   scales: one E8M0 byte per lane (its row's 32-element k block), byte `opsel` of the 32-bit scale operand; value 127 = 2^0.
// and prints what the hardware does with (a) zero scale operands (the compiler is expected to select the unscaled opcode),
// (b) scale bytes 127 / 128 (x1 / x2), (c) per-lane different scales (block scaling along k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// OCP e4m3 encodings of the small integers 0..7 (exact): 0, 1 = 0x38, 2 = 0x40, 3 = 0x44, 4 = 0x48, 5 = 0x4a, 6 = 0x4c, 7 = 0x4e
__host__ __device__ inline unsigned char e4m3_of(int v) {
    const unsigned char t[8] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e};
    return t[v & 7];
}

template <int MODE>
__global__ void probe(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, float* __restrict__ C, int sa, int sb)
{
    const int l = threadIdx.x;
    const int row = l & 15, kg = l >> 4;
    i32x8 a, b;
    std::memcpy(&a, A + row * 128 + kg * 32, 32);
    std::memcpy(&b, B + row * 128 + kg * 32, 32);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);             // zero scales
    if (MODE == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);           // uniform scale bytes
    if (MODE == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127 + kg, 0, 127);    // A scaled per k block: 2^kg
    for (int e = 0; e < 4; ++e) C[(4 * kg + e) * 16 + row] = c[e];        // assumed: rows 4 kg + e (from A), column = lane & 15 (from B)
}

int main()
{
    std::vector<unsigned char> A(16 * 128), B(16 * 128);
    std::vector<int> Ai(16 * 128), Bi(16 * 128);
    for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 128; ++k) {
            Ai[r * 128 + k] = (r * 3 + k * 5 + (k >> 5)) % 8;
            Bi[r * 128 + k] = (r * 7 + k + 2 * (k >> 5)) % 8;
            A[r * 128 + k] = e4m3_of(Ai[r * 128 + k]);
            B[r * 128 + k] = e4m3_of(Bi[r * 128 + k]);
        }
    unsigned char *dA, *dB; float* dC;
    HIPCK(hipMalloc(&dA, A.size())); HIPCK(hipMalloc(&dB, B.size())); HIPCK(hipMalloc(&dC, 256 * sizeof(float)));
    HIPCK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); HIPCK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    std::vector<float> C(256);
    auto ref = [&](int m, int n, int mode) {
        double s = 0;
        for (int k = 0; k < 128; ++k) s += (double)Ai[m * 128 + k] * Bi[n * 128 + k] * (mode == 2 ? (double)(1 << (k >> 5)) : 1.0);
        return s;
    };
    auto check = [&](const char* what, int mode, double factor) {
        (void)hipMemcpy(C.data(), dC, 256 * sizeof(float), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n)
                if (C[m * 16 + n] != (float)(ref(m, n, mode) * factor)) ++bad;
        std::printf("%-58s %s (%d of 256 differ; C[0][0] = %g, expected %g)\n", what, bad ? "MISMATCH" : "ok", bad, C[0], ref(0, 0, mode) * factor);
        return bad;
    };
    int bad = 0;
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0, 0); HIPCK(hipDeviceSynchronize());
    bad += check("layout, zero scale operands (unscaled opcode expected)", 0, 1.0);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 127, 127); HIPCK(hipDeviceSynchronize());
    bad += check("scale bytes 127 / 127 (x1)", 1, 1.0);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 128, 127); HIPCK(hipDeviceSynchronize());
    bad += check("scale bytes 128 / 127 (A x2)", 1, 2.0);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 127, 126); HIPCK(hipDeviceSynchronize());
    bad += check("scale bytes 127 / 126 (B x0.5)", 1, 0.5);
    hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0, 0); HIPCK(hipDeviceSynchronize());
    bad += check("per-lane A scale 2^(k block): block scaling along k", 2, 1.0);
    std::printf(bad ? "PROBE: assumptions do NOT all hold\n" : "PROBE: layout and scale semantics as assumed\n");
    return bad ? 1 : 0;
}
