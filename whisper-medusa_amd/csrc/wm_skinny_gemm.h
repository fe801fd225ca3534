// wm_skinny_gemm.h — weight-streaming MFMA GEMM for the decode step (token rows M <= 16).
//
//   out[m][n] = sum_k X[m][k] * W[n][k]          X: M x K activations, W: N x K weights (packed bf16)
//
// The decode step is HBM-bound: every weight byte is read exactly once per pass, so the kernel is
// organised around the weight stream.  One wavefront owns one 16-row weight tile x one K-slice: it
// fetches 1-KiB packed fragments with fully coalesced non-temporal 16-B loads (the first round is
// issued BEFORE the token operand is prepared, so HBM latency hides under the LayerNorm), multiplies
// them against the M token rows with v_mfma_f32_16x16x32_bf16 (weights = A operand, tokens = B operand:
// the idle token columns of a batch-1 pass cost nothing), and the KSPLIT waves of a row tile reduce
// through LDS in a fixed order (deterministic).
//
// Activations are carried as a bf16 HI/LO PAIR (x = hi + lo, ~17 mantissa bits): two MFMAs per weight
// fragment instead of one — free in a bandwidth-bound kernel — so decoder activations are never rounded
// to 8 bits and the fp32-accumulate results match the oracle to ~1e-6 (no rounding-flip cascades).
//
// Token operand LOADERS:
//   LdPacked  — packed hi/lo planes in global memory (L2-resident), fragments prefetched in registers
//   LdNorm    — fp32 residual rows -> LayerNorm (or identity) -> hi/lo fragments in LDS, fused
// Results leave through a fused EPILOGUE functor (wm_epilogues.h).
#pragma once
#include <cstdlib>
#include "wm_common.h"
#include "wm_epilogues.h"

struct LdPacked {
    const bf16_t* X; int K32; size_t plane;            // lo plane at X + plane
    static constexpr bool kLds = false;
    __host__ __device__ __forceinline__ static size_t lds_bytes(int) { return 0; }
    __device__ __forceinline__ void prepare(char*) const {}
    __device__ __forceinline__ bf16x8_t frag(const char*, int kt, int lane, int pl) const {
        return ld_frag(X + (pl ? plane : 0) + ((size_t)kt * 64 + lane) * 8);
    }
};

__device__ __forceinline__ void hilo8_to_lds(bf16_t* xh, bf16_t* xl, int q, float4 y0, float4 y1)
{
    uint4 h, l;
    h.x = pack_bf2(y0.x, y0.y); h.y = pack_bf2(y0.z, y0.w); h.z = pack_bf2(y1.x, y1.y); h.w = pack_bf2(y1.z, y1.w);
    l.x = pack_bf2(y0.x - __uint_as_float(h.x << 16), y0.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf2(y0.z - __uint_as_float(h.y << 16), y0.w - __uint_as_float(h.y & 0xffff0000u));
    l.z = pack_bf2(y1.x - __uint_as_float(h.z << 16), y1.y - __uint_as_float(h.z & 0xffff0000u));
    l.w = pack_bf2(y1.z - __uint_as_float(h.w << 16), y1.w - __uint_as_float(h.w & 0xffff0000u));
    reinterpret_cast<uint4*>(xh)[q] = h;          // chunk q = 16 B: lane-linear, bank-conflict-free
    reinterpret_cast<uint4*>(xl)[q] = l;
}

// fp32 rows -> (LayerNorm | identity) -> packed hi/lo fragments in LDS.  GEMM row r reads source row
// r*row_mul + row_off (selects the last prompt row per stream on the first base pass).
// Threads walk the 16 x K tile in PACKED order (one 16-B chunk = 8 consecutive k of one row; with a block
// size that is a multiple of 64 every chunk of a thread belongs to the same row r = tid & 15).  All global
// loads of a thread are issued together (ONE memory round trip); LayerNorm statistics (single-pass sum and
// sum of squares) come from the same registers: a 4-row permlane-swap reduction inside the wave, then a fixed-order sum of
// the per-wave partials in LDS (deterministic).  LDS writes are lane-linear ds_write_b128.
struct LdNorm {
    const float* h; const float* gamma; const float* beta;
    int d, K32, M, row_mul, row_off, do_norm;
    static constexpr bool kLds = true;
    static constexpr int UB = 8;                 // packed chunks per thread per batch
    __host__ __device__ __forceinline__ static size_t scratch_bytes(int K32) { return 2048 + (size_t)K32 * 256; }
    __host__ __device__ __forceinline__ static size_t lds_bytes(int K32) { return (size_t)2 * K32 * 1024 + scratch_bytes(K32); }

    // requires K32 * 64 <= UB * blockDim.x (the whole 16 x K tile in one batch; checked by the launcher)
    __device__ __forceinline__ void prepare(char* smem) const {
        bf16_t* xh = reinterpret_cast<bf16_t*>(smem);
        prepare_to(xh, xh + (size_t)K32 * 512, smem + (size_t)2 * K32 * 1024, 0);
    }
    // rows row0 .. row0+15 of the GEMM -> fragments at xh / xl (LDS for the fused GEMM, global memory for the
    // batched path: the arithmetic and its order are the same, so both give bit-identical operands)
    __device__ __forceinline__ void prepare_to(bf16_t* xh, bf16_t* xl, char* scratch, int row0) const {
        float2* part = reinterpret_cast<float2*>(scratch);                              // [waves <= 16][16 rows]
        float* gb = reinterpret_cast<float*>(scratch + 2048);                           // gamma[d] then beta[d]
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        const int nv = d >> 2, nq = K32 * 64, T = blockDim.x;
        const int r = lane & 15;                 // the row of every chunk this thread touches
        const int g8 = (lane >> 4) * 8;
        const bool rvalid = row0 + r < M;
        const float* hrow = h + (size_t)((row0 + r) * row_mul + row_off) * d + g8;
        float4 v0[UB], v1[UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int q = threadIdx.x + i * T;
            if (q < nq && rvalid) {
                const float4* src = reinterpret_cast<const float4*>(hrow + (q >> 6) * 32);
                v0[i] = src[0]; v1[i] = src[1];
            } else { v0[i] = make_float4(0.f, 0.f, 0.f, 0.f); v1[i] = v0[i]; }
        }
        float mean = 0.f, rstd = 1.f;
        if (do_norm) {
            for (int c = threadIdx.x; c < nv; c += T) {
                reinterpret_cast<float4*>(gb)[c] = reinterpret_cast<const float4*>(gamma)[c];
                reinterpret_cast<float4*>(gb + d)[c] = reinterpret_cast<const float4*>(beta)[c];
            }
            float s = 0.f, q2 = 0.f;
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                s += (v0[i].x + v0[i].y) + (v0[i].z + v0[i].w) + (v1[i].x + v1[i].y) + (v1[i].z + v1[i].w);
                q2 += (v0[i].x * v0[i].x + v0[i].y * v0[i].y) + (v0[i].z * v0[i].z + v0[i].w * v0[i].w) +
                      (v1[i].x * v1[i].x + v1[i].y * v1[i].y) + (v1[i].z * v1[i].z + v1[i].w * v1[i].w);
            }
            s = rows4_sum(s); q2 = rows4_sum(q2);
            if (lane < 16) part[wave * 16 + lane] = make_float2(s, q2);
            __syncthreads();
            float ts = 0.f, tq = 0.f;
            for (int w = 0; w < nw; ++w) { const float2 p = part[w * 16 + r]; ts += p.x; tq += p.y; }
            mean = ts / (float)d;
            rstd = rsqrtf(fmaxf(tq / (float)d - mean * mean, 0.f) + 1e-5f);
        }
        const bool norm_row = do_norm && rvalid;
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int q = threadIdx.x + i * T, k0 = (q >> 6) * 32 + g8;
            if (q >= nq) continue;
            float4 y0 = v0[i], y1 = v1[i];
            if (norm_row) {
                const float4 g0 = *reinterpret_cast<const float4*>(gb + k0), g1 = *reinterpret_cast<const float4*>(gb + k0 + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(gb + d + k0), b1 = *reinterpret_cast<const float4*>(gb + d + k0 + 4);
                y0.x = (y0.x - mean) * rstd * g0.x + b0.x; y0.y = (y0.y - mean) * rstd * g0.y + b0.y;
                y0.z = (y0.z - mean) * rstd * g0.z + b0.z; y0.w = (y0.w - mean) * rstd * g0.w + b0.w;
                y1.x = (y1.x - mean) * rstd * g1.x + b1.x; y1.y = (y1.y - mean) * rstd * g1.y + b1.y;
                y1.z = (y1.z - mean) * rstd * g1.z + b1.z; y1.w = (y1.w - mean) * rstd * g1.w + b1.w;
            }
            hilo8_to_lds(xh, xl, q, y0, y1);
        }
        __syncthreads();
    }
    __device__ __forceinline__ bf16x8_t frag(const char* smem, int kt, int lane, int pl) const {
        return ld_frag(reinterpret_cast<const bf16_t*>(smem) + ((size_t)(pl * K32 + kt) * 64 + lane) * 8);
    }
};

template <int U, bool W8, class Ld, class Ep>
__global__ void __launch_bounds__(640)
k_skinny_gemm(const bf16_t* __restrict__ W, const float* __restrict__ wscale, int N16, int K32, int ksplit, int rt_per_wg,
              const int* __restrict__ done, Ld ld, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // done: every stream finished (the rest of this replay is a no-op)
    if (done && *done) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = wave % ksplit, rtl = wave / ksplit;
    const int rt = blockIdx.x * rt_per_wg + rtl;
    float4* red = reinterpret_cast<float4*>(smem + Ld::lds_bytes(K32));
    const int nk = K32 / ksplit, kt0 = ks * nk;
    const bool active = rt < N16;
    const size_t wp = ((size_t)(active ? rt : 0) * K32 + kt0) * 512 + lane * 8;     // element index (bf16: 2 B, fp8: 1 B per element)

    // first round of the weight stream goes out before the token operand exists
    bf16x8_t a[U], xh[U], xl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = ld_wfrag<W8, true>(W, wp + (size_t)u * 512);

    // epilogue operand (residual + bias) of the element this thread will finish: fetched under the weight stream
    float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
    int em = 0, en = 0; bool edo = false;
    if constexpr (Ep::kPre) {
        if (ksplit > 1) {
            if (threadIdx.x < rt_per_wg * 64) {
                const int rt2 = blockIdx.x * rt_per_wg + (threadIdx.x >> 6);
                em = lane & 15; en = rt2 * 16 + 4 * (lane >> 4); edo = rt2 < N16;
            }
        } else { em = lane & 15; en = rt * 16 + 4 * (lane >> 4); edo = active; }
        if (edo) pre = ep.pre4(em, en);
    }

    ld.prepare(smem);

    if (!Ld::kLds) {
#pragma unroll
        for (int u = 0; u < U; ++u) { xh[u] = ld.frag(smem, kt0 + u, lane, 0); xl[u] = ld.frag(smem, kt0 + u, lane, 1); }
    }
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < nk; kk += U) {
        bf16x8_t an[U];
        const bool more = (kk + U) < nk;                      // wave-uniform
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) an[u] = ld_wfrag<W8, true>(W, wp + (size_t)(kk + U + u) * 512);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bf16x8_t bh = Ld::kLds ? ld.frag(smem, kt0 + kk + u, lane, 0) : xh[u];
            const bf16x8_t bl = Ld::kLds ? ld.frag(smem, kt0 + kk + u, lane, 1) : xl[u];
            acc = mfma16(a[u], bh, acc);
            acc = mfma16(a[u], bl, acc);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a[u] = an[u];
                if (!Ld::kLds) { xh[u] = ld.frag(smem, kt0 + kk + U + u, lane, 0); xl[u] = ld.frag(smem, kt0 + kk + U + u, lane, 1); }
            }
        }
    }

    if (ksplit > 1) {
        red[(rtl * ksplit + ks) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (threadIdx.x < rt_per_wg * 64) {              // blockDim >= 64 * rt_per_wg: one output quad per thread
            const int rtl2 = threadIdx.x >> 6, l2 = lane;
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
            for (int k2 = 0; k2 < ksplit; ++k2) {
                const float4 p = red[(rtl2 * ksplit + k2) * 64 + l2];
                s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
            }
            const int rt2 = blockIdx.x * rt_per_wg + rtl2;
            if (rt2 < N16) {
                if constexpr (W8) s = scale4(s, wscale, rt2 * 16 + 4 * (l2 >> 4));
                if constexpr (Ep::kPre) ep.store4p(l2 & 15, rt2 * 16 + 4 * (l2 >> 4), s, pre);
                else ep.store4(l2 & 15, rt2 * 16 + 4 * (l2 >> 4), s);
            }
        }
    } else if (active) {
        if constexpr (W8) acc = scale4(acc, wscale, rt * 16 + 4 * (lane >> 4));
        if constexpr (Ep::kPre) ep.store4p(lane & 15, rt * 16 + 4 * (lane >> 4), acc, pre);
        else ep.store4(lane & 15, rt * 16 + 4 * (lane >> 4), acc);
    }
}

// ---- batched path (token rows R > 16) ----------------------------------------------------------
// LayerNorm of all row tiles to global packed hi/lo planes: the SAME code as the fused loader, run with the
// same block size, so the operand bits equal those of a 16-row launch.
__global__ void __launch_bounds__(640)
k_ln_tiles(LdNorm ld, bf16_t* __restrict__ xg, size_t plane, const int* __restrict__ done)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (done && *done) return;
    const size_t off = (size_t)blockIdx.x * ld.K32 * 512;
    ld.prepare_to(xg + off, xg + plane + off, smem, blockIdx.x * 16);
}

// More than 16 token rows (several streams): a wave owns RT weight row tiles x TT token tiles x ONE K-slice,
// so every weight fragment feeds 2*TT MFMAs and every token fragment RT of them (register blocking cuts the L2
// traffic of the operand re-reads by RT resp. TT); blockIdx.y walks the token-tile groups, so the chip is filled
// by tokens as well as by features and the weights are re-read from L2 / Infinity Cache, not HBM.  Per output the
// accumulation order is the 16-row kernel's (k ascending, hi then lo; K-slices summed in order): bit-identical.
template <int NKR, int RT, int TT, bool W8, class Ep>
__global__ void __launch_bounds__(640)
k_rows_gemm(const bf16_t* __restrict__ W, const float* __restrict__ wscale, int N16, int K32, int ksplit, const int* __restrict__ done,
            const bf16_t* __restrict__ X, size_t plane, int MT, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (done && *done) return;
    const int lane = threadIdx.x & 63;
    const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rt0 = blockIdx.x * RT, mt0 = blockIdx.y * TT;
    const int kt0 = ks * NKR;
    size_t wp[RT]; const bf16_t* xp[TT];
#pragma unroll
    for (int i = 0; i < RT; ++i) wp[i] = ((size_t)min(rt0 + i, N16 - 1) * K32 + kt0) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < TT; ++j) xp[j] = X + ((size_t)min(mt0 + j, MT - 1) * K32 + kt0) * 512 + lane * 8;
    f32x4_t acc[RT][TT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int G = 4;
#pragma unroll
    for (int kg = 0; kg < NKR; kg += G) {
        bf16x8_t a[RT][G], xh[TT][G], xl[TT][G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
#pragma unroll
            for (int j = 0; j < TT; ++j) { xh[j][u] = ld_frag(xp[j] + (size_t)(kg + u) * 512); xl[j][u] = ld_frag(xp[j] + plane + (size_t)(kg + u) * 512); }
#pragma unroll
            for (int i = 0; i < RT; ++i) a[i][u] = ld_wfrag<W8, false>(W, wp[i] + (size_t)(kg + u) * 512);
        }
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j) { acc[i][j] = mfma16(a[i][u], xh[j][u], acc[i][j]); acc[i][j] = mfma16(a[i][u], xl[j][u], acc[i][j]); }
    }
    if (ksplit > 1) {
        float4* red = reinterpret_cast<float4*>(smem);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
                red[((i * TT + j) * ksplit + ks) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        __syncthreads();
        for (int e = threadIdx.x; e < RT * TT * 64; e += blockDim.x) {
            const int t = e >> 6, l2 = e & 63, i = t / TT, j = t - i * TT;
            f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
            for (int k2 = 0; k2 < ksplit; ++k2) {
                const float4 p = red[(t * ksplit + k2) * 64 + l2];
                sacc[0] += p.x; sacc[1] += p.y; sacc[2] += p.z; sacc[3] += p.w;
            }
            if (rt0 + i < N16 && mt0 + j < MT) {
                if constexpr (W8) sacc = scale4(sacc, wscale, (rt0 + i) * 16 + 4 * (l2 >> 4));
                ep.store4((mt0 + j) * 16 + (l2 & 15), (rt0 + i) * 16 + 4 * (l2 >> 4), sacc);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
                if (rt0 + i < N16 && mt0 + j < MT) {
                    if constexpr (W8) acc[i][j] = scale4(acc[i][j], wscale, (rt0 + i) * 16 + 4 * (lane >> 4));
                    ep.store4((mt0 + j) * 16 + (lane & 15), (rt0 + i) * 16 + 4 * (lane >> 4), acc[i][j]);
                }
    }
}

// ---- host-side launch plan -------------------------------------------------------------------
static thread_local const int* g_skinny_done = nullptr;     // device flag checked by every launch of this translation unit
struct SkinnyPlan { int ksplit, rt, U; };

// K-slices of at most 16 fragments and, if possible,
// >= 1024 waves.  The plan depends only on (N16, K32, loader kind): 16-row and batched launches of one GEMM
// share it, which is what makes their results bit-identical.
static inline int skinny_env(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}
static inline SkinnyPlan skinny_plan(int N16, int K32, bool lds_loader) {
    static const int cap = skinny_env("WM_PLAN_WAVE_CAP", 10);          // tuning knobs (bench sweeps); defaults are the shipped plan
    static const int target = skinny_env("WM_PLAN_TARGET_WAVES", 1024);
    static const int nkmax = skinny_env("WM_PLAN_NK_MAX", 16);
    static const int rt2 = skinny_env("WM_PLAN_RT2", 1);
    SkinnyPlan p; p.U = (K32 % 8 == 0) ? 8 : 4;
    const int q = K32 / p.U;                // candidate ksplit must divide q
    int best = 1;
    for (int s = 1; s <= cap && s <= q; ++s) {      // <= 10 waves per block (launch bound 640 threads)
        if (q % s) continue;
        best = s;
        if ((long)N16 * s >= target && K32 / s <= nkmax) break;
    }
    p.ksplit = best;
    p.rt = (best == 1) ? 4 : 1;
    // LDS loaders hold the whole 16 x K operand (one block per CU): when there are more row tiles than CUs,
    // let two tiles share one block (and one LayerNorm) instead of running a second round of blocks
    if (rt2 && lds_loader && best > 1 && best <= 5 && N16 > 256) p.rt = 2;
    return p;
}

// a weight matrix in the packed layout: bf16 (scale == nullptr) or fp8 e4m3 with one fp32 scale per output row
struct WRef {
    const bf16_t* w; const float* scale;
    WRef(const bf16_t* w_, const float* scale_ = nullptr) : w(w_), scale(scale_) {}
};

template <int U, bool W8, class Ld, class Ep>
static inline hipError_t launch_skinny_u(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const Ld& ld, const Ep& ep) {
    const int grid = (N16 + p.rt - 1) / p.rt;
    const int threads = 64 * p.ksplit * p.rt;
    const size_t lds = Ld::lds_bytes(K32) + (p.ksplit > 1 ? (size_t)p.rt * p.ksplit * 1024 : 0);
    auto kern = k_skinny_gemm<U, W8, Ld, Ep>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, W.w, W.scale, N16, K32, p.ksplit, p.rt, g_skinny_done, ld, ep);
    return hipGetLastError();
}

template <class Ld, class Ep>
static inline hipError_t launch_skinny(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const Ld& ld, const Ep& ep) {
    if (W.scale) return p.U == 8 ? launch_skinny_u<8, true>(st, W, N16, K32, p, ld, ep) : launch_skinny_u<4, true>(st, W, N16, K32, p, ld, ep);
    return p.U == 8 ? launch_skinny_u<8, false>(st, W, N16, K32, p, ld, ep) : launch_skinny_u<4, false>(st, W, N16, K32, p, ld, ep);
}

template <int NKR, int RT, bool W8, class Ep>
static inline hipError_t launch_rows_gemm_w(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                            const bf16_t* X, size_t plane, int MT, const Ep& ep) {
    constexpr int TT = 2;
    const dim3 grid((N16 + RT - 1) / RT, (MT + TT - 1) / TT);
    const size_t lds = p.ksplit > 1 ? (size_t)RT * TT * p.ksplit * 1024 : 0;
    auto kern = k_rows_gemm<NKR, RT, TT, W8, Ep>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * p.ksplit), lds, st, W.w, W.scale, N16, K32, p.ksplit, g_skinny_done, X, plane, MT, ep);
    return hipGetLastError();
}

template <int NKR, int RT, class Ep>
static inline hipError_t launch_rows_gemm(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                          const bf16_t* X, size_t plane, int MT, const Ep& ep) {
    if (W.scale) return launch_rows_gemm_w<NKR, RT, true>(st, W, N16, K32, p, X, plane, MT, ep);
    return launch_rows_gemm_w<NKR, RT, false>(st, W, N16, K32, p, X, plane, MT, ep);
}

template <int NKR, class Ep>
static inline hipError_t launch_skinny_mt_nk(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                             const bf16_t* X, size_t plane, int MT, const Ep& ep) {
    static const int min_blocks = skinny_env("WM_ROWS_GEMM_MIN_BLOCKS", 400);
    // weight row tiles per wave: as many as still leave >= 400 blocks (~1.5 per CU; swept 100..800 at 8 and 32 streams) (register blocking divides the L2 re-reads
    // of the token operand; with few token tiles the chip has to be filled by features instead).  Same results.
    const int groups = (MT + 1) / 2;
    if (((N16 + 3) / 4) * groups >= min_blocks) return launch_rows_gemm<NKR, 4>(st, W, N16, K32, p, X, plane, MT, ep);
    if (((N16 + 1) / 2) * groups >= min_blocks) return launch_rows_gemm<NKR, 2>(st, W, N16, K32, p, X, plane, MT, ep);
    return launch_rows_gemm<NKR, 1>(st, W, N16, K32, p, X, plane, MT, ep);
}

template <class Ep>
static inline hipError_t launch_skinny_mt(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                          const bf16_t* X, size_t plane, int MT, const Ep& ep) {
    const int nk = K32 / p.ksplit;
    if (nk == 16) return launch_skinny_mt_nk<16>(st, W, N16, K32, p, X, plane, MT, ep);
    if (nk == 12) return launch_skinny_mt_nk<12>(st, W, N16, K32, p, X, plane, MT, ep);
    if (nk == 8) return launch_skinny_mt_nk<8>(st, W, N16, K32, p, X, plane, MT, ep);
    if (nk == 4) return launch_skinny_mt_nk<4>(st, W, N16, K32, p, X, plane, MT, ep);
    return hipErrorInvalidConfiguration;
}

// out = X (R token rows, packed hi/lo planes in global memory) times W^T (N = 16*N16 features, K = 32*K32)
template <class Ep>
static inline hipError_t launch_skinny_rows(hipStream_t st, WRef W, int N16, int K32, int R, const bf16_t* X, size_t plane,
                                            const Ep& ep) {
    const SkinnyPlan p = skinny_plan(N16, K32, false);
    if (R <= 16) return launch_skinny(st, W, N16, K32, p, LdPacked{X, K32, plane}, ep);
    return launch_skinny_mt(st, W, N16, K32, p, X, plane, (R + 15) / 16, ep);
}

static inline bool skinny_norm_fusable(int N16, int K32) {
    const SkinnyPlan p = skinny_plan(N16, K32, true);
    return K32 <= LdNorm::UB * p.ksplit * p.rt;
}

// LayerNorm-fused GEMM over R token rows.  R <= 16: one fused launch.  R > 16: the same LayerNorm code writes
// the packed hi/lo operand to `xscr` (global), then the register-blocked token-tile kernel runs.  The fused loader
// needs the whole 16 x K tile in one batch of the block's threads (true for every Whisper size).
template <class Ep>
static inline hipError_t launch_skinny_norm(hipStream_t st, WRef W, int N16, int K32, const float* h, const float* gamma,
                                            const float* beta, int d, int R, int row_mul, int row_off, int do_norm, const Ep& ep,
                                            bf16_t* xscr, size_t plane) {
    if (!skinny_norm_fusable(N16, K32)) return hipErrorInvalidConfiguration;
    const SkinnyPlan p = skinny_plan(N16, K32, true);
    const LdNorm ld{h, gamma, beta, d, K32, R, row_mul, row_off, do_norm};
    if (R <= 16) return launch_skinny(st, W, N16, K32, p, ld, ep);
    const int MT = (R + 15) / 16;
    hipLaunchKernelGGL(k_ln_tiles, dim3(MT), dim3(64 * p.ksplit * p.rt), LdNorm::scratch_bytes(K32), st, ld, xscr, plane, g_skinny_done);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_skinny_mt(st, W, N16, K32, p, xscr, plane, MT, ep);
}
