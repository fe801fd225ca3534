// Round 4: what a persistent decoder layer would pay per dependency edge on MI355X, measured.
//     hipcc --offload-arch=gfx950 -O3 tests/microbench/xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
//
// One "phase" models one operation of the single-stream decode chain: every block reads the WHOLE token operand the previous
// phase produced (11 rows x 1280 as a bf16 hi/lo pair = 56 KB, written 1/256 per block) and writes its own slice of the next one.
// Three ways to order consecutive phases:
//   launches : one kernel per phase, captured in a hipGraph (what the engine does today; the boundary is the fence)
//   counter  : one persistent kernel, single monotonic counter, relaxed sc1 polling + release / acquire fences
//              (MI355X_MICROARCH.md "barrier-counter"; round 1's grid_barrier.hip polled with ACQUIRE loads — the guide's worst row)
//   xcd      : one persistent kernel, XCD-hierarchical barrier (guide row "barrier-xcd"): per-XCC arrival counter, the LAST arriver of an
//              XCC does the release fence (one L2 write-back per XCD), arrives on the top counter, polls it, acquires, and bumps its
//              XCC's generation word; everybody else polls that word with relaxed loads and then does ONE agent acquire.
// Each with payload 0 (sync only) and with the 56 KB gather.  Output: microseconds per phase.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NF = 14336;              // floats of the operand (56 KB)
struct Bar { unsigned xcc[8][32]; unsigned top[32]; unsigned gen[8][32]; unsigned members[8][32]; unsigned flat[32]; };

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void barrier_counter(Bar* b, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&b->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_relaxed(&b->flat[0]) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// phase p = 1, 2, ...; members[x] = blocks living on XCC x (counted by the kernel itself at start: placement-independent)
__device__ __forceinline__ void barrier_xcd(Bar* b, unsigned p, int x, unsigned mem_x, unsigned n_xcc)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have reached the XCD's L2
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&b->xcc[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == mem_x * p) {                           // last arriver of this XCC: publish the XCD's L2, meet the other leaders
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(&b->top[0]) < n_xcc * p) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(&b->gen[x][0], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(&b->gen[x][0]) < p) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ float gather(const float* in)       // every thread of the block: 14 x 16 B, all issued before the first use
{
    float4 v[NF / 1024];
#pragma unroll
    for (int i = 0; i < NF / 1024; ++i) v[i] = reinterpret_cast<const float4*>(in)[i * 256 + threadIdx.x];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NF / 1024; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    return s;
}
__device__ __forceinline__ void produce(float* out, float s, int ph)
{
    const int per = NF / gridDim.x;
    if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = s * 1e-9f + (float)(ph & 7);
}

template <int MODE, int PAYLOAD>       // MODE 1 counter, 2 xcd
__global__ void __launch_bounds__(256) k_persist(Bar* b, float* a, float* c, int phases, unsigned* err)
{
    const int x = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;       // HW_REG_XCC_ID
    __shared__ unsigned s_mem, s_nx;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&b->members[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&b->flat[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_relaxed(&b->flat[1]) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        unsigned nx = 0;
        for (int i = 0; i < 8; ++i) nx += ld_relaxed(&b->members[i][0]) != 0u;
        s_mem = ld_relaxed(&b->members[x][0]); s_nx = nx;
    }
    __syncthreads();
    const unsigned mem_x = s_mem, n_xcc = s_nx;
    float s = 0.f;
    for (int ph = 0; ph < phases; ++ph) {
        float* in = (ph & 1) ? c : a; float* out = (ph & 1) ? a : c;
        if (PAYLOAD) { s = gather(in); produce(out, s, ph); }
        if (MODE == 1) barrier_counter(b, (unsigned)(ph + 1) * gridDim.x);
        else barrier_xcd(b, (unsigned)(ph + 1), x, mem_x, n_xcc);
        // staleness check: what phase ph wrote must be what phase ph + 1 reads (value of word 0 of every block's slice)
        if (PAYLOAD && threadIdx.x < gridDim.x) {
            const float v = out[threadIdx.x * (NF / gridDim.x)];
            if ((int)v != (ph & 7)) atomicAdd(err, 1u);
        }
    }
    if (s == 123.456f) a[0] = s;
}

template <int PAYLOAD>
__global__ void __launch_bounds__(256) k_phase(const float* in, float* out, int ph)
{
    float s = 0.f;
    if (PAYLOAD) s = gather(in);
    produce(out, s, ph);
}

template <int MODE, int PAYLOAD>
int run_persist(const char* name, int grid, Bar* bar, float* a, float* c, unsigned* err, hipStream_t st)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 400;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(bar, 0, sizeof(Bar), st)); CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((k_persist<MODE, PAYLOAD>), dim3(grid), dim3(256), 0, st, bar, a, c, phases, err);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-46s grid %3d: %6.2f us/phase  stale %u\n", name, grid, best * 1000.f / phases, herr);
    return 0;
}

template <int PAYLOAD>
int run_launches(const char* name, int grid, float* a, float* c, hipStream_t st)
{
    const int phases = 400;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int ph = 0; ph < phases; ++ph)
        hipLaunchKernelGGL((k_phase<PAYLOAD>), dim3(grid), dim3(256), 0, st, (ph & 1) ? c : a, (ph & 1) ? a : c, ph);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-46s grid %3d: %6.2f us/phase\n", name, grid, best * 1000.f / phases);
    return 0;
}

int main()
{
    Bar* bar; float *a, *c; unsigned* err;
    CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMalloc(&a, NF * 4)); CK(hipMalloc(&c, NF * 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(a, 0, NF * 4)); CK(hipMemset(c, 0, NF * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int grid : {128, 256}) {
        run_launches<0>("launches (hipGraph), empty phase", grid, a, c, st);
        run_launches<1>("launches (hipGraph), 56 KB gather", grid, a, c, st);
        run_persist<1, 0>("persistent, counter barrier, sync only", grid, bar, a, c, err, st);
        run_persist<1, 1>("persistent, counter barrier, 56 KB gather", grid, bar, a, c, err, st);
        run_persist<2, 0>("persistent, XCD-hierarchical barrier, sync only", grid, bar, a, c, err, st);
        run_persist<2, 1>("persistent, XCD-hierarchical barrier, 56 KB gather", grid, bar, a, c, err, st);
    }
    return 0;
}
