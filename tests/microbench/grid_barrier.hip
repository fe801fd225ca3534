// Cost of a software grid barrier (all blocks co-resident) vs a kernel boundary, with a 64 KB dependent
// hand-off between phases as in handoff.hip.   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// MODE 0: barrier only.  MODE 1: barrier + every block reads the 64 KB written by all blocks in the previous phase
// (sc1 stores / sc1 loads so no cache maintenance is needed).
template <int MODE>
__global__ void __launch_bounds__(256) k_persist(unsigned* ctr, float* a, float* b, int nfloat, int phases, unsigned base)
{
    const int per = nfloat / gridDim.x;
    float acc = 0.f;
    for (int ph = 0; ph < phases; ++ph) {
        float* in = (ph & 1) ? b : a; float* out = (ph & 1) ? a : b;
        if (MODE == 1) {
            for (int i = threadIdx.x * 2; i < nfloat; i += 512) {
                const u64 v = __hip_atomic_load(reinterpret_cast<const u64*>(in + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += __uint_as_float((unsigned)v) + __uint_as_float((unsigned)(v >> 32));
            }
            for (int j = threadIdx.x; j < per; j += 256)
                __hip_atomic_store(out + blockIdx.x * per + j, acc * 1e-9f + (float)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_barrier(ctr, base + (unsigned)(ph + 1) * gridDim.x);
    }
    if (acc == 123.456f) a[0] = acc;
}

template <int MODE>
int run(const char* name, int grid, unsigned* ctr, float* a, float* b, hipStream_t st)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 400, nf = 16384 - 16384 % grid;
    CK(hipMemsetAsync(ctr, 0, 4, st));
    hipLaunchKernelGGL((k_persist<MODE>), dim3(grid), dim3(256), 0, st, ctr, a, b, nf, phases, 0u);
    CK(hipStreamSynchronize(st));
    CK(hipMemsetAsync(ctr, 0, 4, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_persist<MODE>), dim3(grid), dim3(256), 0, st, ctr, a, b, nf, phases, 0u);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s grid %4d: %.2f us/phase\n", name, grid, ms * 1000.f / phases);
    return 0;
}

int main()
{
    unsigned* ctr; float *a, *b;
    CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20));
    CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int grid : {8, 64, 128, 240, 256}) {
        run<0>("barrier only", grid, ctr, a, b, st);
        run<1>("barrier + 64 KB all-to-all hand-off", grid, ctr, a, b, st);
    }
    return 0;
}
