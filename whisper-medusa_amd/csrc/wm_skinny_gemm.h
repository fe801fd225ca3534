// wm_skinny_gemm.h — weight-streaming MFMA GEMM for the decode step (rows M <= 32).
//
//   out[m][n] = sum_k X[m][k] * W[n][k]          X: M x K activations, W: N x K weights (packed bf16)
//
// The decode step is HBM-bound: every weight byte is read exactly once per pass, so the kernel is
// organised around the weight stream.  One wavefront owns one 16-row weight tile x one K-slice:
// it fetches 1-KiB packed fragments with fully coalesced non-temporal 16-B loads (two rounds of U
// fragments in flight), multiplies them against the M (<=16*MT) token rows with
// v_mfma_f32_16x16x32_bf16 (weights = A operand, tokens = B operand, so the 15 idle token columns
// of a batch-1 pass cost nothing extra), and the KSPLIT waves of a row tile reduce through LDS in a
// fixed order (deterministic).  The token operand comes from a pluggable LOADER:
//   LdPacked  — already-packed bf16 rows in global memory (L2-resident)
//   LdNorm    — fp32 residual rows -> LayerNorm (or identity) -> bf16 fragments in LDS, fused
//   LdCombine — cross-attention split-S partials -> softmax-combine -> bf16 fragments in LDS
// and the result leaves through a fused EPILOGUE functor (wm_epilogues.h).
#pragma once
#include "wm_common.h"
#include "wm_epilogues.h"

struct LdPacked {
    const bf16_t* X; int K32;
    static constexpr bool kLds = false;
    __host__ __device__ __forceinline__ static size_t lds_bytes(int, int) { return 0; }
    __device__ __forceinline__ void prepare(char*, int) const {}
    __device__ __forceinline__ bf16x8_t frag(const char*, int mt, int kt, int lane) const {
        return ld_frag(X + ((size_t)(mt * K32 + kt) * 64 + lane) * 8);
    }
};

// fp32 rows -> (LayerNorm | identity) -> packed bf16 in LDS.  Row r of the GEMM reads source row
// r*row_mul + row_off (selects the last prompt row per stream on the first base pass).
// Block 0 optionally writes the normalised fp32 rows (the post-final-LN state the Medusa heads and the
// Block layer consume, model.py:1262,1382).
struct LdNorm {
    const float* h; const float* gamma; const float* beta; float* norm_out;
    int d, K32, M, row_mul, row_off, do_norm;
    static constexpr bool kLds = true;
    __host__ __device__ __forceinline__ static size_t lds_bytes(int MT, int K32) { return (size_t)MT * K32 * 1024; }
    __device__ __forceinline__ void prepare(char* smem, int MT) const {
        bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        const int nv = d >> 2;                                   // float4 per row
        for (int r = wave; r < MT * 16; r += nw) {
            if (r >= M) {                                        // zero the pad rows (keeps LDS finite)
                for (int j = lane; j < nv; j += 64)
                    *reinterpret_cast<uint2*>(xs + packed_index(r, j * 4, K32)) = make_uint2(0u, 0u);
                continue;
            }
            const float4* src = reinterpret_cast<const float4*>(h + (size_t)(r * row_mul + row_off) * d);
            float4 v[8];                                         // d <= 2048
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = lane + 64 * i;
                v[i] = (j < nv) ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                s += v[i].x + v[i].y + v[i].z + v[i].w;
            }
            float mean = 0.f, rstd = 1.f;
            if (do_norm) {
                mean = wave_sum(s) / (float)d;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (lane + 64 * i < nv) {
                        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
                        q += a * a + b * b + c * c + e * e;
                    }
                }
                rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = lane + 64 * i;
                if (j < nv) {
                    float4 y = v[i];
                    if (do_norm) {
                        const float4 g = reinterpret_cast<const float4*>(gamma)[j];
                        const float4 b = reinterpret_cast<const float4*>(beta)[j];
                        y.x = (y.x - mean) * rstd * g.x + b.x; y.y = (y.y - mean) * rstd * g.y + b.y;
                        y.z = (y.z - mean) * rstd * g.z + b.z; y.w = (y.w - mean) * rstd * g.w + b.w;
                    }
                    if (norm_out && blockIdx.x == 0) reinterpret_cast<float4*>(norm_out + (size_t)r * d)[j] = y;
                    uint2 o; o.x = pack_bf2(y.x, y.y); o.y = pack_bf2(y.z, y.w);
                    *reinterpret_cast<uint2*>(xs + packed_index(r, j * 4, K32)) = o;
                }
            }
        }
        __syncthreads();
    }
    __device__ __forceinline__ bf16x8_t frag(const char* smem, int mt, int kt, int lane) const {
        return ld_frag(reinterpret_cast<const bf16_t*>(smem) + ((size_t)(mt * K32 + kt) * 64 + lane) * 8);
    }
};

// Cross-attention partials of NS key-splits: (max, sum) in ml[row][head][split][2], un-normalised
// outputs in o[row][head][split][64]  ->  softmax-combine -> packed bf16 in LDS (K = d).
struct LdCombine {
    const float* ml; const float* o; int H, NS, K32, M;
    static constexpr bool kLds = true;
    __host__ __device__ __forceinline__ static size_t lds_bytes(int MT, int K32) { return (size_t)MT * K32 * 1024; }
    __device__ __forceinline__ void prepare(char* smem, int MT) const {
        bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
        const int d = H * 64, nq = d >> 2;
        for (int e = threadIdx.x; e < MT * 16 * nq; e += blockDim.x) {
            const int r = e / nq, c = (e - r * nq) * 4, hd = c >> 6;
            uint2 outv = make_uint2(0u, 0u);
            if (r < M) {
                const float* mlp = ml + ((size_t)r * H + hd) * NS * 2;
                float mx = -INFINITY;
                for (int s = 0; s < NS; ++s) mx = fmaxf(mx, mlp[2 * s]);
                float L = 0.f; float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int s = 0; s < NS; ++s) {
                    const float w = __expf(mlp[2 * s] - mx);
                    L += mlp[2 * s + 1] * w;
                    const float4 ov = *reinterpret_cast<const float4*>(o + (((size_t)r * H + hd) * NS + s) * 64 + (c & 63));
                    acc.x += ov.x * w; acc.y += ov.y * w; acc.z += ov.z * w; acc.w += ov.w * w;
                }
                const float inv = 1.0f / L;
                outv.x = pack_bf2(acc.x * inv, acc.y * inv); outv.y = pack_bf2(acc.z * inv, acc.w * inv);
            }
            *reinterpret_cast<uint2*>(xs + packed_index(r, c, K32)) = outv;
        }
        __syncthreads();
    }
    __device__ __forceinline__ bf16x8_t frag(const char* smem, int mt, int kt, int lane) const {
        return ld_frag(reinterpret_cast<const bf16_t*>(smem) + ((size_t)(mt * K32 + kt) * 64 + lane) * 8);
    }
};

template <int MT, int U, class Ld, class Ep>
__global__ void __launch_bounds__(1024)
k_skinny_gemm(const bf16_t* __restrict__ W, int N16, int K32, int ksplit, int rt_per_wg, Ld ld, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = wave % ksplit, rtl = wave / ksplit;
    const int rt = blockIdx.x * rt_per_wg + rtl;
    float4* red = reinterpret_cast<float4*>(smem + Ld::lds_bytes(MT, K32));

    ld.prepare(smem, MT);

    const int nk = K32 / ksplit, kt0 = ks * nk;
    f32x4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (rt < N16) {
        const bf16_t* wp = W + ((size_t)rt * K32 + kt0) * 512 + lane * 8;
        bf16x8_t a[U], x[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = ld_frag_nt(wp + (size_t)u * 512);
            if (!Ld::kLds) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) x[u][mt] = ld.frag(smem, mt, kt0 + u, lane);
            }
        }
        for (int kk = 0; kk < nk; kk += U) {
            bf16x8_t an[U], xn[U][MT];
            const bool more = (kk + U) < nk;                      // wave-uniform
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    an[u] = ld_frag_nt(wp + (size_t)(kk + U + u) * 512);
                    if (!Ld::kLds) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) xn[u][mt] = ld.frag(smem, mt, kt0 + kk + U + u, lane);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8_t xb = Ld::kLds ? ld.frag(smem, mt, kt0 + kk + u, lane) : x[u][mt];
                    acc[mt] = mfma16(a[u], xb, acc[mt]);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    a[u] = an[u];
                    if (!Ld::kLds) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) x[u][mt] = xn[u][mt];
                    }
                }
            }
        }
    }

    if (ksplit > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            red[((rtl * ksplit + ks) * MT + mt) * 64 + lane] = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
        __syncthreads();
        for (int e = threadIdx.x; e < rt_per_wg * MT * 64; e += blockDim.x) {
            const int rtl2 = e / (MT * 64), mt = (e >> 6) % MT, l2 = e & 63;
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
            for (int k2 = 0; k2 < ksplit; ++k2) {
                const float4 p = red[((rtl2 * ksplit + k2) * MT + mt) * 64 + l2];
                s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
            }
            const int rt2 = blockIdx.x * rt_per_wg + rtl2;
            if (rt2 < N16) ep.store4(mt * 16 + (l2 & 15), rt2 * 16 + 4 * (l2 >> 4), s);
        }
    } else if (rt < N16) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ep.store4(mt * 16 + (lane & 15), rt * 16 + 4 * (lane >> 4), acc[mt]);
    }
}

// ---- host-side launch plan -------------------------------------------------------------------
struct SkinnyPlan { int ksplit, rt, U; };

static inline SkinnyPlan skinny_plan(int N16, int K32, bool lds_loader) {
    SkinnyPlan p; p.U = (lds_loader && K32 % 8 == 0) ? 8 : 4;
    const int q = K32 / p.U;                // candidate ksplit must divide q
    int best = 1;
    for (int s = 1; s <= 16 && s <= q; ++s) {
        if (q % s) continue;
        best = s;
        if ((long)N16 * s >= 1024) break;   // enough waves to cover 256 CUs x 4 SIMDs
    }
    p.ksplit = best;
    p.rt = (best == 1) ? 4 : 1;
    return p;
}

template <int MT, int U, class Ld, class Ep>
static inline hipError_t launch_skinny_mt(hipStream_t st, const bf16_t* W, int N16, int K32, const SkinnyPlan& p,
                                          const Ld& ld, const Ep& ep) {
    const int grid = (N16 + p.rt - 1) / p.rt;
    const int threads = 64 * p.ksplit * p.rt;
    const size_t lds = Ld::lds_bytes(MT, K32) + (p.ksplit > 1 ? (size_t)p.rt * p.ksplit * MT * 1024 : 0);
    auto kern = k_skinny_gemm<MT, U, Ld, Ep>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, W, N16, K32, p.ksplit, p.rt, ld, ep);
    return hipGetLastError();
}

// out = X (M rows, M <= 32) times W^T (N = 16*N16 features, K = 32*K32)
template <class Ld, class Ep>
static inline hipError_t launch_skinny(hipStream_t st, const bf16_t* W, int N16, int K32, int M, const Ld& ld, const Ep& ep) {
    const SkinnyPlan p = skinny_plan(N16, K32, Ld::kLds);
    if constexpr (Ld::kLds) {        // LDS loaders keep no token fragments in registers: deeper weight prefetch
        if (p.U == 8) {
            if (M <= 16) return launch_skinny_mt<1, 8>(st, W, N16, K32, p, ld, ep);
            return launch_skinny_mt<2, 8>(st, W, N16, K32, p, ld, ep);
        }
    }
    if (M <= 16) return launch_skinny_mt<1, 4>(st, W, N16, K32, p, ld, ep);
    return launch_skinny_mt<2, 4>(st, W, N16, K32, p, ld, ep);
}
