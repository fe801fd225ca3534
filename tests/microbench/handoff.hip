// Kernel-to-kernel activation hand-off cost: chain of N kernels; every block of kernel i reads the whole 64 KB
// buffer written by kernel i-1 (each block wrote one slice) and writes its slice of the other buffer.
// Variants of store / load cache policy.  hipcc --offload-arch=gfx950 -O3 handoff.hip -o handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;

template <int SMODE, int LMODE>   // 0 plain, 1 nontemporal, 2 sc1 (agent-scope relaxed atomic 8B)
__global__ void __launch_bounds__(256) k_chain(const float* __restrict__ in, float* __restrict__ out, int nfloat)
{
    // read all of `in` (nfloat floats), reduce, write slice
    float acc = 0.f;
    for (int i = threadIdx.x * 4; i < nfloat; i += 256 * 4) {
        f4 v;
        if (LMODE == 1) v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(in + i));
        else if (LMODE == 2) {
            unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(in + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(in + i + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[0] = __uint_as_float((unsigned)a); v[1] = __uint_as_float((unsigned)(a >> 32)); v[2] = __uint_as_float((unsigned)b); v[3] = __uint_as_float((unsigned)(b >> 32));
        } else v = *reinterpret_cast<const f4*>(in + i);
        acc += v[0] + v[1] + v[2] + v[3];
    }
    // slice: each block writes nfloat/gridDim.x floats
    const int per = nfloat / gridDim.x;
    for (int j = threadIdx.x; j < per; j += 256) {
        const float val = acc * 1e-9f + (float)j;
        float* p = out + blockIdx.x * per + j;
        if (SMODE == 1) __builtin_nontemporal_store(val, p);
        else if (SMODE == 2) __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = val;
    }
}

template <int SM, int LM>
int run(const char* name, float* a, float* b, hipStream_t st, int grid, int nfloat)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 400;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k_chain<SM, LM>), dim3(grid), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, nfloat);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s grid %4d  %6d floats: %.2f us/kernel\n", name, grid, nfloat, ms * 1000.f / (5 * N));
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return 0;
}

int main()
{
    float *a, *b; CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22)); CK(hipMemset(a, 0, 1 << 22)); CK(hipMemset(b, 0, 1 << 22));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int grid : {1, 8, 80, 240}) {
        const int nf = 16384;   // 64 KB
        run<0, 0>("plain store / plain load", a, b, st, grid, nf);
        run<1, 0>("nt store / plain load", a, b, st, grid, nf);
        run<2, 0>("sc1 store / plain load", a, b, st, grid, nf);
        run<2, 2>("sc1 store / sc1 load", a, b, st, grid, nf);
        run<0, 1>("plain store / nt load", a, b, st, grid, nf);
        run<1, 1>("nt store / nt load", a, b, st, grid, nf);
    }
    run<0, 0>("plain/plain 256 KB", a, b, st, 240, 65536 - 65536 % 240);
    return 0;
}
