// wm_epilogues.h — fused GEMM epilogues shared by the weight-streaming (decode) GEMM and the
// tiled (encoder / batched) GEMM.  Both kernels compute D[n][m] with the WEIGHT tile as the
// MFMA A operand, so a lane owns 4 consecutive output features n..n+3 of one token row m:
// every functor exposes   store4(m, n, v)   with n % 4 == 0.
#pragma once
#include "wm_common.h"

// Decode epilogues come in two halves so that the weight-streaming GEMM can issue every global load of a launch in ONE
// batch at kernel entry:  pre(m, n) only LOADS the operands the element needs (bias, residual, cache position) — no
// arithmetic, so nothing waits on them while the weight stream is in flight —, fin(m, n, v, pre) does the arithmetic and the
// store once the accumulator exists.  store4(m, n, v) = fin(m, n, v, pre(m, n)) for the kernels that do not prefetch.
struct EpPre { float4 a, b; int i; float4 c; };      // (c: only the folding residual epilogue; a field nobody reads costs no register)

// h[m][n] = (h[m][n] + bias[n]) + v      (out_proj / fc2 + residual, HF:modeling_whisper.py:396-413)
// FOLD (round 6, wm_common.h "LayerNorm folded into the GEMM it feeds"): the row this launch completes is the input of the NEXT LayerNorm-fed
// GEMM, so the epilogue also writes that GEMM's operand gamma_next o h as packed hi / lo planes (`xo`) and the tile's partial (sum, sum of
// squares) of h into the statistics table — the three LayerNorm launches per layer of a batched pass and the statistics prologue of the
// single-stream launches disappear.  fin() must then be called by all 64 lanes of a wave together (the partial crosses the four 16-lane rows).
template <bool FOLD>
struct EpResidualT {
    float* h; const float* bias; int ld; int M;
    const float* gnext = nullptr; bf16_t* xo = nullptr; size_t xplane = 0; float2* stats = nullptr; int K32 = 0, sld = 0;
    __device__ __forceinline__ EpPre pre(int m, int n) const {
        // rows >= M read row M - 1 (never stored): no exec-masked loads in the launch's request batch (wm_skinny_gemm.h LdPacked::issue)
        EpPre p; p.i = 0;
        p.a = *reinterpret_cast<const float4*>(h + (size_t)min(m, M - 1) * ld + n); p.b = *reinterpret_cast<const float4*>(bias + n);
        if constexpr (FOLD) p.c = *reinterpret_cast<const float4*>(gnext + n);
        return p;
    }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        const float4 y = make_float4((p.a.x + p.b.x) + v[0], (p.a.y + p.b.y) + v[1], (p.a.z + p.b.z) + v[2], (p.a.w + p.b.w) + v[3]);
        if constexpr (!FOLD) {
            if (m < M) *reinterpret_cast<float4*>(h + (size_t)m * ld + n) = y;
        } else {
            // lane (row = lane & 15, g = lane >> 4) holds features n .. n + 3 = 16 tile + 4 g ..: the tile's partial of a row is the sum over its 4 g-lanes
            const float s = rows4_sum(sum4(y));
            const float q = rows4_sum(sumsq4(y));
            if (m < M) {
                *reinterpret_cast<float4*>(h + (size_t)m * ld + n) = y;
                const size_t o = packed_index(m, n, K32);
                act_st4(xo + o, xo + xplane + o, make_float4(y.x * p.c.x, y.y * p.c.y, y.z * p.c.z, y.w * p.c.w));
                if ((n & 15) == 0) stats[(size_t)(n >> 4) * sld + m] = make_float2(s, q);
            }
        }
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
};
typedef EpResidualT<false> EpResidual;
typedef EpResidualT<true> EpResidualFold;

template <bool BIAS>            // out[m][n] = (v + bias[n]) * scale   (BIAS: cross-attn q;  !BIAS: vocabulary logits, scale 1)
struct EpF32T {                 // two TYPES, not a run-time null test: a branch around the bias load in the launch's request batch breaks the
    float* out; const float* bias; int ld; int M; float scale;      // compiler's counted waits apart (wm_skinny_gemm.h LdNormT::issue)
    __device__ __forceinline__ EpPre pre(int m, int n) const {
        EpPre p; p.i = 0; p.a = make_float4(0.f, 0.f, 0.f, 0.f); p.b = p.a;
        if constexpr (BIAS) p.a = *reinterpret_cast<const float4*>(bias + n);
        return p;
    }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        if (m >= M) return;
        *reinterpret_cast<float4*>(out + (size_t)m * ld + n) =
            make_float4((v[0] + p.a.x) * scale, (v[1] + p.a.y) * scale, (v[2] + p.a.z) * scale, (v[3] + p.a.w) * scale);
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
};
typedef EpF32T<true> EpF32;
typedef EpF32T<false> EpLogits;

template <int ACT>             // packed bf16 out = act(v + bias)   (fc1 + GELU -> next GEMM's operand); ACT 1: decoder (erff), 2: encoder (gelu_phi)
struct EpPackedAct {           // out_lo != nullptr: decoder path, value kept as a bf16 hi/lo pair
    bf16_t* out; bf16_t* out_lo; const float* bias; int K32out; int M;
    __device__ __forceinline__ EpPre pre(int m, int n) const {
        EpPre p; p.i = 0; p.b = make_float4(0.f, 0.f, 0.f, 0.f);
        p.a = *reinterpret_cast<const float4*>(bias + n);
        return p;
    }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        if (m >= M) return;
        float x0 = v[0] + p.a.x, x1 = v[1] + p.a.y, x2 = v[2] + p.a.z, x3 = v[3] + p.a.w;
        if (ACT == 1) { x0 = gelu_erf(x0); x1 = gelu_erf(x1); x2 = gelu_erf(x2); x3 = gelu_erf(x3); }
        if (ACT == 2) { x0 = gelu_phi(x0); x1 = gelu_phi(x1); x2 = gelu_phi(x2); x3 = gelu_phi(x3); }
        const size_t o = packed_index(m, n, K32out);
        if (out_lo) { act_st4(out + o, out_lo + o, make_float4(x0, x1, x2, x3)); return; }      // decoder: the next GEMM's operand (wm_common.h ActFrag)
        uint2 u; u.x = pack_bf2(x0, x1); u.y = pack_bf2(x2, x3);
        *reinterpret_cast<uint2*>(out + o) = u;
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
};

// Decoder self-attention projections: q (scaled, fp32) to a row buffer, k/v rows (bf16) straight
// into the contiguous KV cache at position base[stream] + r   (HF:modeling_whisper.py:288-318; the
// reference's per-iteration cat-compaction, model.py:378-402, becomes "overwrite rows >= kv_len").
// DENSE (a type, not a run-time branch: a conditional load in pre() splits the launch's request batch — the one-stream QKV launch
// measured 7.05 -> 7.76 us with an `if (rowinfo)` here): the merged-step schedule's dense rows, row m -> rowinfo[m] (k_step_begin).
template <bool DENSE>
struct EpQKVDecT {             // K cache [s][h][pos][64]; V cache [s][h] as V^T MFMA fragments (vfrag_index)
    float* q; bf16_t* kc; bf16_t* vc; const float* bias; const int* base;
    int Mper, d, H, Tal, M;
    const int4* rowinfo = nullptr; // DENSE: {stream, index inside the stream, position of its row 0, kind 0 none / 1 base / 2 verify}
    __device__ __forceinline__ EpPre pre(int m, int n) const {
        EpPre p; p.b = make_float4(0.f, 0.f, 0.f, 0.f);
        p.a = *reinterpret_cast<const float4*>(bias + n);
        if constexpr (DENSE) {                                                // one 16-byte load, requested with the rest of the launch's batch
            const int4 ri = rowinfo[min(m, M - 1)];
            p.i = ri.z; p.b.x = __int_as_float(ri.x); p.b.y = __int_as_float(ri.y); p.b.z = __int_as_float(ri.w == 2 ? Mper : ri.w);   // kind 0 / 1 / 2 -> 0 / 1 / Mper rows
        } else {
            const int sm = min(m, M - 1) / Mper;                              // rows >= M: the last row's stream (never stored)
            p.i = base[sm];
            p.b.x = __int_as_float(sm); p.b.y = __int_as_float(min(m, M - 1) - sm * Mper); p.b.z = __int_as_float(Mper);
        }
        return p;
    }
    // NOTE: must be called by all 64 lanes of a wave together (the V branch exchanges values between lanes)
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        const bool inb = m < M;
        const float x0 = v[0] + p.a.x, x1 = v[1] + p.a.y, x2 = v[2] + p.a.z, x3 = v[3] + p.a.w;
        if (n < d) {                                   // n is wave-uniform up to the 16-feature tile: the branch is uniform
            if (inb) *reinterpret_cast<float4*>(q + (size_t)m * d + n) = make_float4(x0 * 0.125f, x1 * 0.125f, x2 * 0.125f, x3 * 0.125f);
            return;
        }
        const int s = __float_as_int(p.b.x), r = __float_as_int(p.b.y), cnt = __float_as_int(p.b.z);
        const bool valid = inb && r < cnt;
        int pos = p.i + r; if (pos > Tal - 1) pos = Tal - 1;
        if (n < 2 * d) {
            const int c = n - d;
            uint2 o; o.x = pack_bf2(x0, x1); o.y = pack_bf2(x2, x3);
            if (valid) *reinterpret_cast<uint2*>(kc + (((size_t)s * H + (c >> 6)) * Tal + pos) * 64 + (c & 63)) = o;
        } else {
            const int c = n - 2 * d;
            // leader: position multiple of 4 whose next three rows exist, belong to the same stream and stay inside the cache
            const bool lead = valid && (pos & 3) == 0 && (m & 15) <= 12 && m + 3 < M && r + 3 < cnt && p.i + r + 3 <= Tal - 1;
            const int back = pos & 3;                  // my group's leader sits `back` lanes below me (if it is in my row group)
            const int lane = (int)(threadIdx.x & 63);
            const int src = lane - back;
            // lead flag of the lane `back` (1..3) below me inside my 16-lane row: three DPP row shifts (row_shr:n — lane i reads i-n)
            const int lf = (int)lead;
            const int l1 = __builtin_amdgcn_update_dpp(0, lf, 0x111, 0xf, 0xf, true);
            const int l2 = __builtin_amdgcn_update_dpp(0, lf, 0x112, 0xf, 0xf, true);
            const int l3 = __builtin_amdgcn_update_dpp(0, lf, 0x113, 0xf, 0xf, true);
            const int lflag = back == 1 ? l1 : (back == 2 ? l2 : l3);
            const bool covered = valid && back > 0 && src >= (lane & ~15) && lflag != 0;
            vt_store4(vc + ((size_t)s * H + (c >> 6)) * 64 * Tal, pos, c & 63, x0, x1, x2, x3, valid, lead, covered);
        }
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
};
typedef EpQKVDecT<false> EpQKVDec;
typedef EpQKVDecT<true> EpQKVDecDense;

// Medusa residual heads: y = x + SiLU(W x + b) (model.py:180-210) for head k = n / d, written as
// packed bf16 row  m*row_mul + row_off + k  of the vocabulary-projection operand.
struct EpHead {
    bf16_t* y; bf16_t* y_lo; const float* hf; const float* bias; int d, K32, row_mul, row_off, M, src_mul, src_off;
    __device__ __forceinline__ EpPre pre(int m, int n) const {
        EpPre p; p.i = 0;
        const int k = n / d, c = n - k * d;
        p.a = *reinterpret_cast<const float4*>(bias + n);
        p.b = *reinterpret_cast<const float4*>(hf + (size_t)(min(m, M - 1) * src_mul + src_off) * d + c);
        return p;
    }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        if (m >= M) return;
        const int k = n / d, c = n - k * d;
        const size_t o = packed_index(m * row_mul + row_off + k, c, K32);
        act_st4(y + o, y_lo + o, make_float4(p.b.x + silu(v[0] + p.a.x), p.b.y + silu(v[1] + p.a.y),
                                              p.b.z + silu(v[2] + p.a.z), p.b.w + silu(v[3] + p.a.w)));
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
};

// ---- encoder -----------------------------------------------------------------------------
struct EpConv1 {               // a1[b][t][n] = gelu(conv1) as row-major bf16 (input of the conv2 im2col)
    bf16_t* a1; const float* bias; int T, Tpad, d;
    __device__ __forceinline__ EpPre pre(int, int n) const {
        EpPre p; p.i = 0; p.b = make_float4(0.f, 0.f, 0.f, 0.f); p.a = *reinterpret_cast<const float4*>(bias + n);
        return p;
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        const int b = m / Tpad, t = m - b * Tpad;
        if (t >= T) return;
        const float4 bb = p.a;
        uint2 o;
        o.x = pack_bf2(gelu_phi(v[0] + bb.x), gelu_phi(v[1] + bb.y));
        o.y = pack_bf2(gelu_phi(v[2] + bb.z), gelu_phi(v[3] + bb.w));
        *reinterpret_cast<uint2*>(a1 + ((size_t)b * T + t) * d + n) = o;
    }
};

struct EpConv2 {               // h = gelu(conv2) + embed_positions   (HF:modeling_whisper.py:626-632)
    float* h; const float* bias; const float* pos; int S, Spad, d;
    __device__ __forceinline__ EpPre pre(int m, int n) const {
        EpPre p; p.i = 0; p.a = make_float4(0.f, 0.f, 0.f, 0.f); p.b = p.a;
        const int s = m % Spad;
        if (s < S) { p.a = *reinterpret_cast<const float4*>(bias + n); p.b = *reinterpret_cast<const float4*>(pos + (size_t)s * d + n); }
        return p;
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        const int s = m % Spad;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < S) {
            const float4 bb = p.a;
            const float4 pp = p.b;
            o = make_float4(gelu_phi(v[0] + bb.x) + pp.x, gelu_phi(v[1] + bb.y) + pp.y,
                            gelu_phi(v[2] + bb.z) + pp.z, gelu_phi(v[3] + bb.w) + pp.w);
        }
        *reinterpret_cast<float4*>(h + (size_t)m * d + n) = o;
    }
};

struct EpQKVEnc {              // q (x 64^-1/2), k as [b][h][s][64]; v as V^T MFMA fragments per [b][h] (vfrag_index)
    bf16_t* q; bf16_t* k; bf16_t* vt; const float* bias; int Spad, H, d;
    __device__ __forceinline__ EpPre pre(int, int n) const {
        EpPre p; p.i = 0; p.b = make_float4(0.f, 0.f, 0.f, 0.f); p.a = *reinterpret_cast<const float4*>(bias + n);
        return p;
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        const int b = m / Spad, s = m - b * Spad;
        const float4 bb = p.a;
        float x0 = v[0] + bb.x, x1 = v[1] + bb.y, x2 = v[2] + bb.z, x3 = v[3] + bb.w;
        if (n < 2 * d) {
            const bool isq = n < d;
            const int c = isq ? n : n - d;
            if (isq) { x0 *= 0.125f; x1 *= 0.125f; x2 *= 0.125f; x3 *= 0.125f; }
            uint2 o; o.x = pack_bf2(x0, x1); o.y = pack_bf2(x2, x3);
            *reinterpret_cast<uint2*>((isq ? q : k) + (((size_t)b * H + (c >> 6)) * Spad + s) * 64 + (c & 63)) = o;
        } else {
            const int c = n - 2 * d;          // rows of a 16-lane group are 16 consecutive, 16-aligned positions of one clip
            const bool lead = (s & 3) == 0;
            vt_store4(vt + ((size_t)b * H + (c >> 6)) * 64 * Spad, s, c & 63, x0, x1, x2, x3, true, lead, !lead);
        }
    }
};

struct EpCrossKV {             // K_x [kvl][b][h][s][64], V_x [kvl][b][h] as V^T MFMA fragments (vfrag_index)   (HF:modeling_whisper.py:322-335)
    bf16_t* kx; bf16_t* vx; const float* bias; int Spad, H, d, B;
    __device__ __forceinline__ EpPre pre(int, int n) const {
        EpPre p; p.i = 0; p.b = make_float4(0.f, 0.f, 0.f, 0.f); p.a = *reinterpret_cast<const float4*>(bias + n);
        return p;
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        const int b = m / Spad, s = m - b * Spad;
        const int kvl = n / (2 * d), rem = n - kvl * 2 * d;
        const bool isv = rem >= d;
        const int c = isv ? rem - d : rem;
        const float4 bb = p.a;
        const size_t slab = (((size_t)kvl * B + b) * H + (c >> 6)) * Spad * 64;
        if (!isv) {
            uint2 o; o.x = pack_bf2(v[0] + bb.x, v[1] + bb.y); o.y = pack_bf2(v[2] + bb.z, v[3] + bb.w);
            *reinterpret_cast<uint2*>(kx + slab + (size_t)s * 64 + (c & 63)) = o;
        } else {
            const bool lead = (s & 3) == 0;
            vt_store4(vx + slab, s, c & 63, v[0] + bb.x, v[1] + bb.y, v[2] + bb.z, v[3] + bb.w, true, lead, !lead);
        }
    }
};
