// Launch-boundary floor on this box: dependent chain of N kernels, eager vs hipGraph replay,
// trivial vs big-kernarg vs large-dynamic-LDS kernels.  hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { float* p; const float* a; const float* b; const float* c; int v[24]; };
__global__ void k_triv(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_big(Big s) { if (threadIdx.x == 0 && blockIdx.x == 0) s.p[0] += (float)s.v[3]; }
__global__ void k_lds(float* p) { extern __shared__ float sm[]; sm[threadIdx.x] = p[0]; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = sm[1] + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 500;
    Big bs{}; bs.p = d; bs.v[3] = 1;
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int variant = 0; variant < 4; ++variant) {
        auto launch = [&](int grid) {
            if (variant == 0) hipLaunchKernelGGL(k_triv, dim3(grid), dim3(64), 0, st, d);
            else if (variant == 1) hipLaunchKernelGGL(k_big, dim3(grid), dim3(320), 0, st, bs);
            else if (variant == 2) hipLaunchKernelGGL(k_lds, dim3(grid), dim3(320), 97 * 1024, st, d);
            else hipLaunchKernelGGL(k_triv, dim3(grid), dim3(256), 0, st, d);
        };
        const int grid = variant == 3 ? 2048 : (variant == 0 ? 1 : 240);
        for (int i = 0; i < 20; ++i) launch(grid);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) launch(grid);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const float eager = ms * 1000.f / N;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) launch(grid);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d (grid %d): eager %.2f us/kernel, graph %.2f us/kernel\n", variant, grid, eager, ms * 1000.f / (5 * N));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
