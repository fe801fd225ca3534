// wm_decoder.hip — decode-step kernels and the per-iteration schedule (F3..F14 of SURVEY.md §8a).
//
// One Medusa iteration = base pass (1 token per stream; P on the first) + verify pass (K+1 candidate
// tokens per stream) + accept.  All lengths live on the device (L, kvlen, finished), every kernel has a
// static shape, so the whole iteration is captured once as a hipGraph and replayed (wm_engine.hip).
#include "wm_internal.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "wm_skinny_gemm.h"

// ---------------------------------------------------------------------------------------------
// embed: h[row] = embed_tokens[tok] + embed_positions[pos]        (HF:modeling_whisper.py:745-765)
// ---------------------------------------------------------------------------------------------
// rowinfo (merged-step schedule, k_step_begin): the pass runs over DENSE rows; a stream in base mode contributes ONE row — token
// ids[s][kvlen[s]] at position kvlen[s] — a stream in verify mode its K + 1 candidate rows at L[s] ..
__global__ void k_embed(float* __restrict__ h, const bf16_t* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                        const int* __restrict__ base, const int* __restrict__ tok_src, int tok_stride, int use_base_off,
                        int Mper, int d, int V, int Tmax, const int* __restrict__ depth,
                        const int4* __restrict__ rowinfo = nullptr, const int* __restrict__ ids = nullptr, int ids_stride = 0,
                        // LayerNorm fold (wm_common.h): operand gamma o h of layer 0's QKV GEMM + the row's statistics partials
                        const float* __restrict__ gnext = nullptr, bf16_t* __restrict__ xo = nullptr, size_t xplane = 0,
                        float2* __restrict__ stats = nullptr, int sld = 0)
{
    const int row = blockIdx.x;
    int s = row / Mper, r = row - s * Mper;
    bool base_row = false;
    int b0;
    if (rowinfo) {                      // merged-step schedule: dense rows (k_step_begin); a stream with ONE row contributes its base token
        const int4 ri = rowinfo[row];
        if (ri.w == 0) return;          // no row here in this step
        s = ri.x; r = ri.y; b0 = ri.z; base_row = ri.w == 1;
    } else {
        b0 = base[s];
    }
    int tok = base_row ? ids[(size_t)s * ids_stride + b0] : tok_src[(size_t)s * tok_stride + (use_base_off ? b0 : 0) + r];
    tok = min(max(tok, 0), V - 1);
    // candidate tree: node r sits at position L + depth(r) (medusa_position_ids, medusa_utils.py:360-363)
    const int pos = min(b0 + (base_row ? 0 : (depth ? depth[r] : r)), Tmax - 1);
    const bf16_t* te = tok_emb + (size_t)tok * d;
    const float* pe = pos_emb + (size_t)pos * d;
    for (int j = threadIdx.x * 4; j < d; j += blockDim.x * 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(te + j);
        const float4 p = *reinterpret_cast<const float4*>(pe + j);
        const float4 y = make_float4(bf2f((bf16_t)(t.x & 0xffff)) + p.x, bf2f((bf16_t)(t.x >> 16)) + p.y,
                                     bf2f((bf16_t)(t.y & 0xffff)) + p.z, bf2f((bf16_t)(t.y >> 16)) + p.w);
        *reinterpret_cast<float4*>(h + (size_t)row * d + j) = y;
        if (gnext) {            // (kernel-uniform; a 16-feature tile = 4 adjacent lanes, all inside the loop together: d % 16 == 0)
            const float4 g = *reinterpret_cast<const float4*>(gnext + j);
            const size_t o = packed_index(row, j, d >> 5);
            act_st4(xo + o, xo + xplane + o, make_float4(y.x * g.x, y.y * g.y, y.z * g.z, y.w * g.w));
            float sm = sum4(y), q = sumsq4(y);
            sm += __shfl_xor(sm, 1, 64); q += __shfl_xor(q, 1, 64);
            sm += __shfl_xor(sm, 2, 64); q += __shfl_xor(q, 2, 64);
            if ((j & 15) == 0) stats[(size_t)(j >> 4) * sld + row] = make_float2(sm, q);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rows_norm: one wave per row. out row m <- LayerNorm (or identity) of src row m*src_mul+src_off.
// Optional outputs: fp32 rows (two copies) and packed bf16 hi/lo rows at m*p_mul + p_off.
// ---------------------------------------------------------------------------------------------
__global__ void k_rows_norm(const float* __restrict__ src, int src_mul, int src_off, const float* __restrict__ gamma,
                            const float* __restrict__ beta, int do_norm, float* __restrict__ out_a, float* __restrict__ out_b,
                            bf16_t* __restrict__ out_p, size_t p_plane, int K32, int p_mul, int p_off, int d, int M,
                            const int* __restrict__ carry, const float* __restrict__ hf_keep,
                            // LayerNorm fold (Medusa-Block: the extra layer's LN1 reads the rows written to out_b): operand gamma o y + partials
                            const float* __restrict__ gnext = nullptr, bf16_t* __restrict__ xo = nullptr, size_t xplane = 0,
                            float2* __restrict__ stats = nullptr, int sld = 0)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nv = d >> 2;
    const float4* sp = reinterpret_cast<const float4*>(src + (size_t)(m * src_mul + src_off) * d);
    if (carry && carry[m]) {          // hidden-state carry: the post-LN row was saved by k_accept; the layers were skipped
        sp = reinterpret_cast<const float4*>(hf_keep + (size_t)m * d);
        do_norm = 0;
    }
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        v[i] = (j < nv) ? sp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    float mean = 0.f, rstd = 1.f;
    if (do_norm) {
        mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (lane + 64 * i < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
                q += a * a + b * b + c * c + e * e;
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
            if (out_a) reinterpret_cast<float4*>(out_a + (size_t)m * d)[j] = y;
            if (out_b) reinterpret_cast<float4*>(out_b + (size_t)m * d)[j] = y;
            if (gnext) {        // (kernel-uniform; features 4 j .. 4 j + 3: a 16-feature tile = 4 adjacent lanes, all of them below nv: d % 16 == 0)
                const float4 g = reinterpret_cast<const float4*>(gnext)[j];
                const size_t o = packed_index(m, j * 4, K32);
                act_st4(xo + o, xo + xplane + o, make_float4(y.x * g.x, y.y * g.y, y.z * g.z, y.w * g.w));
                float sm = (y.x + y.y) + (y.z + y.w), q2 = (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
                sm += __shfl_xor(sm, 1, 64); q2 += __shfl_xor(q2, 1, 64);
                sm += __shfl_xor(sm, 2, 64); q2 += __shfl_xor(q2, 2, 64);
                if ((j & 3) == 0) stats[(size_t)(j >> 2) * sld + m] = make_float2(sm, q2);
            }
            if (out_p) {
                const size_t o = packed_index(m * p_mul + p_off, j * 4, K32);
                act_st4(out_p + o, out_p + p_plane + o, y);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Decode attention on the matrix cores for M (<=16) query rows of one (stream, head).
//   S^T = K Q^T  : A = 16 keys x 32 dims of the bf16 K cache (row-major rows = 128-B lines),
//                  B = Q^T as a bf16 hi/lo pair (q keeps ~17 bits) -> a lane owns 8 scores of ONE query,
//   O^T = V^T P^T: A = V^T (the V cache is stored transposed [64][rows]), B = P hi/lo straight from the
//                  score registers (the MFMA k-slot <-> key permutation is shared by both operands).
// fp32 online softmax; 4 waves split the keys and merge through LDS in a fixed order.
//  CROSS = false: causal over the contiguous self-KV cache (query r sees keys <= base + r); wave w takes the
//                 32-key steps w, w+4, ...; writes normalised output as packed hi/lo rows (out_proj operand).
//  CROSS = true : grid.x = key split; the block owns keys [256*split, +256), wave w 64 of them; it publishes
//                 its un-normalised partial (max, sum, o[64]) with an agent-scope release and takes a ticket;
//                 the LAST block of a (stream, head) acquires, merges the NS partials in split order
//                 (deterministic) and writes the packed hi/lo rows — no separate combine launch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_hilo8(const float* x, bf16x8_t& hi, bf16x8_t& lo)
{
    uint4 h, l;
    h.x = pack_bf2(x[0], x[1]); h.y = pack_bf2(x[2], x[3]); h.z = pack_bf2(x[4], x[5]); h.w = pack_bf2(x[6], x[7]);
    l.x = pack_bf2(x[0] - __uint_as_float(h.x << 16), x[1] - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf2(x[2] - __uint_as_float(h.y << 16), x[3] - __uint_as_float(h.y & 0xffff0000u));
    l.z = pack_bf2(x[4] - __uint_as_float(h.z << 16), x[5] - __uint_as_float(h.z & 0xffff0000u));
    l.w = pack_bf2(x[6] - __uint_as_float(h.w << 16), x[7] - __uint_as_float(h.w & 0xffff0000u));
    hi = __builtin_bit_cast(bf16x8_t, h); lo = __builtin_bit_cast(bf16x8_t, l);
}

#define WM_XATTN_NS_MAX 8           // key splits of 256 encoder frames per (stream, head): n_ctx <= 2048 (Whisper: 1500 -> 6)
#define WM_XATTN_SPB_MAX WM_XATTN_NS_MAX   // a block may walk all of them (LDS copy of its partials)

struct KVStep { bf16x8_t k00, k01, k10, k11, v[4]; };        // one 32-key step: K rows as A fragments, V^T fragments
struct AttnAcc { float m_run, l_run; f32x4_t o[4]; };
// the same step from the fp8 e4m3 copy of the cross-K/V (wm_config.cross_kv_fp8; layouts: wm_encoder.hip k_xkv_quant): four 16-byte loads
// instead of eight — lane (key c, g) holds dims 16 g .. + 15 of keys kb + c and kb + 16 + c (k0, k1: two MFMA fragments each, the query is loaded
// in the same dim order) and its 8 keys of the dim-tile pairs (0, 1) and (2, 3) of V^T (v01, v23) —, widened to bf16 right before the MFMAs
struct KVStep8 { u32x4_t k0, k1, v01, v23; };
__device__ __forceinline__ void kv_load8(KVStep8& t, const unsigned char* kp, const unsigned char* vp, int kb, int c, int g, int lane)
{
    const unsigned char* kr = kp + (size_t)(kb + c) * 64 + g * 16;
    const unsigned char* vr = vp + ((size_t)(kb >> 5) * 128 + lane) * 16;
    t.k0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kr));
    t.k1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kr + 16 * 64));
    t.v01 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vr));
    t.v23 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vr + 1024));
}
__device__ __forceinline__ KVStep kv_widen(const KVStep8& r)
{
    KVStep t;
    t.k00 = fp8x8_to_bf16(r.k0[0], r.k0[1]); t.k01 = fp8x8_to_bf16(r.k0[2], r.k0[3]);
    t.k10 = fp8x8_to_bf16(r.k1[0], r.k1[1]); t.k11 = fp8x8_to_bf16(r.k1[2], r.k1[3]);
    t.v[0] = fp8x8_to_bf16(r.v01[0], r.v01[1]); t.v[1] = fp8x8_to_bf16(r.v01[2], r.v01[3]);
    t.v[2] = fp8x8_to_bf16(r.v23[0], r.v23[1]); t.v[3] = fp8x8_to_bf16(r.v23[2], r.v23[3]);
    return t;
}
__device__ __forceinline__ const KVStep& kv_widen(const KVStep& r) { return r; }

// 8 fully coalesced 16-B-per-lane loads: K rows are 128-B lines, the V^T fragments of a step are 4 contiguous KiB
// NT: the cross K/V of a (stream, head) is read by one block once per pass — stream it past the caches (nt policy)
template <bool NT>
__device__ __forceinline__ void kv_load(KVStep& t, const bf16_t* kp, const bf16_t* vp, int kb, int c, int g, int lane)
{
    const bf16_t* kr = kp + (size_t)(kb + c) * 64 + g * 8;
    const bf16_t* vr = vp + ((size_t)(kb >> 5) * 256 + lane) * 8;
    if (NT) {
        t.k00 = ld_frag_nt(kr); t.k01 = ld_frag_nt(kr + 32); t.k10 = ld_frag_nt(kr + 16 * 64); t.k11 = ld_frag_nt(kr + 16 * 64 + 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) t.v[dt] = ld_frag_nt(vr + dt * 512);
    } else {
        t.k00 = ld_frag(kr); t.k01 = ld_frag(kr + 32); t.k10 = ld_frag(kr + 16 * 64); t.k11 = ld_frag(kr + 16 * 64 + 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) t.v[dt] = ld_frag(vr + dt * 512);
    }
}

// scores of 32 keys x 16 queries, online softmax, O^T += V^T P^T
// Visibility of key `kidx` for this lane's query: keys below `limit` (chain: history + causal prefix; tree: history only)
// plus, for a candidate tree, the provisional rows b0 + n of the query node's ancestors (bit n of `anc`; 0 for the chain).
__device__ __forceinline__ void attn_step(AttnAcc& st, const KVStep& t, const bf16x8_t (&qhi)[2], const bf16x8_t (&qlo)[2],
                                          int kb, int g, int limit, int b0 = 0, unsigned long long anc = 0ull)
{
    f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    s0 = mfma16(t.k00, qhi[0], s0); s0 = mfma16(t.k01, qhi[1], s0); s0 = mfma16(t.k00, qlo[0], s0); s0 = mfma16(t.k01, qlo[1], s0);
    s1 = mfma16(t.k10, qhi[0], s1); s1 = mfma16(t.k11, qhi[1], s1); s1 = mfma16(t.k10, qlo[0], s1); s1 = mfma16(t.k11, qlo[1], s1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k0 = kb + 4 * g + r, k1 = k0 + 16;
        const unsigned o0 = (unsigned)(k0 - b0), o1 = (unsigned)(k1 - b0);
        if (k0 >= limit && !(o0 < 64u && ((anc >> o0) & 1ull))) s0[r] = -INFINITY;
        if (k1 >= limit && !(o1 < 64u && ((anc >> o1) & 1ull))) s1[r] = -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
    mx = rows4_max(mx);
    const float m_new = fmaxf(st.m_run, mx);
    const float alpha = (m_new == -INFINITY) ? 1.f : __expf(st.m_run - m_new);
    float p[8], rs = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        p[r] = (m_new == -INFINITY) ? 0.f : __expf(s0[r] - m_new);
        p[4 + r] = (m_new == -INFINITY) ? 0.f : __expf(s1[r] - m_new);
        rs += p[r] + p[4 + r];
    }
    rs = rows4_sum(rs);
    st.l_run = st.l_run * alpha + rs;
    st.m_run = m_new;
    bf16x8_t phi, plo;
    split_hilo8(p, phi, plo);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        st.o[dt][0] *= alpha; st.o[dt][1] *= alpha; st.o[dt][2] *= alpha; st.o[dt][3] *= alpha;
        st.o[dt] = mfma16(t.v[dt], phi, st.o[dt]);
        st.o[dt] = mfma16(t.v[dt], plo, st.o[dt]);
    }
}

// Cross-attention: a wave's two 32-key steps of a split as ONE 64-key block — all 16 score MFMAs first, one max / one sum,
// no running rescale (a split starts from scratch), then all 16 PV MFMAs: half the dependent softmax rounds of two
// successive attn_step calls.  `has1`: the second step exists (wave-uniform).
__device__ __forceinline__ void attn_block64(AttnAcc& st, const KVStep& t0, const KVStep& t1, const bf16x8_t (&qhi)[2], const bf16x8_t (&qlo)[2],
                                             int kb, int g, int limit, bool has1)
{
    f32x4_t s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    s[0] = mfma16(t0.k00, qhi[0], s[0]); s[0] = mfma16(t0.k01, qhi[1], s[0]); s[0] = mfma16(t0.k00, qlo[0], s[0]); s[0] = mfma16(t0.k01, qlo[1], s[0]);
    s[1] = mfma16(t0.k10, qhi[0], s[1]); s[1] = mfma16(t0.k11, qhi[1], s[1]); s[1] = mfma16(t0.k10, qlo[0], s[1]); s[1] = mfma16(t0.k11, qlo[1], s[1]);
    if (has1) {
        s[2] = mfma16(t1.k00, qhi[0], s[2]); s[2] = mfma16(t1.k01, qhi[1], s[2]); s[2] = mfma16(t1.k00, qlo[0], s[2]); s[2] = mfma16(t1.k01, qlo[1], s[2]);
        s[3] = mfma16(t1.k10, qhi[0], s[3]); s[3] = mfma16(t1.k11, qhi[1], s[3]); s[3] = mfma16(t1.k10, qlo[0], s[3]); s[3] = mfma16(t1.k11, qlo[1], s[3]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (kb + 16 * i + 4 * g + r >= limit || (i >= 2 && !has1)) s[i][r] = -INFINITY;
            mx = fmaxf(mx, s[i][r]);
        }
    mx = rows4_max(mx);
    float p[16], rs = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[4 * i + r] = (mx == -INFINITY) ? 0.f : __expf(s[i][r] - mx); rs += p[4 * i + r]; }
    rs = rows4_sum(rs);
    st.m_run = mx; st.l_run = rs;
    bf16x8_t ph0, pl0, ph1, pl1;
    split_hilo8(p, ph0, pl0);
    split_hilo8(p + 8, ph1, pl1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        st.o[dt] = mfma16(t0.v[dt], ph0, st.o[dt]);
        st.o[dt] = mfma16(t0.v[dt], pl0, st.o[dt]);
        if (has1) {
            st.o[dt] = mfma16(t1.v[dt], ph1, st.o[dt]);
            st.o[dt] = mfma16(t1.v[dt], pl1, st.o[dt]);
        }
    }
}

// LDS of the attention kernels (dynamic, so that the fused variant can put its LayerNorm / projection staging next to it)
template <int NSP>
struct AttnLds {
    float s_m[4][16], s_l[4][16];
    int s_last; int pad_[3];
    float s_o[4][16][68];
    float s_part[NSP][16][68];          // [..][64] = max, [..][65] = sum
};

// Fused cross-attention query (single-tile passes): the LN2 + cross-q GEMM launch disappears — every cross-attention
// block computes the 64 query features of ITS head itself: LayerNorm of the token rows (K-slice waves, the LdNorm code and
// slicing of the stand-alone GEMM), normalised hi/lo fragments through LDS, 4 waves x one 16-feature weight row tile x all
// K-slices with the stand-alone kernel's accumulation order (slice partials from zero, summed in slice order) -> q is
// bit-identical to the two-launch path (which the token-tile passes of several streams still use).  The blocks of the 6 key
// splits of a head repeat the projection (160 KB of weights each, L2 hits after the first): a launch boundary and a
// dependent weight round trip cost more.
template <int NK_, int KS_>
struct FuseQ {
    static constexpr bool kOn = true;
    static constexpr int NK = NK_, KS = KS_;
    LdNorm ln; const bf16_t* Wq; const float* bq;
};
struct NoFuseQ { static constexpr bool kOn = false; static constexpr int NK = 1, KS = 1; };

template <bool CROSS, bool NT, class FQ, bool KV8 = false>
__global__ void __launch_bounds__(FQ::kOn ? (FQ::KS > 4 ? 64 * FQ::KS : 256) : 256)
k_attn_mfma(const bf16_t* __restrict__ kmat, const bf16_t* __restrict__ vtmat, const float* __restrict__ q,
            const int4* __restrict__ sinfo, const int* __restrict__ sskip, int mper_nbz, int h_ns, int rows_alloc, int S,
            // ---- the 14 dwords above are preloaded into SGPRs (-amdgpu-kernarg-preload-count=16 = kernarg pointer + 14 dwords):
            // everything a block needs to put its K/V and q loads on the wire.  What follows is fetched from the kernarg
            // segment (a dependent scalar load, ~1 us on a cold launch: the in-kernel timeline showed the attention launches
            // entering 1.3 us later than the GEMMs, whose early arguments already sat in the preloaded range) and is first
            // touched after those loads are in flight.  Round 5: `sinfo` (read BEFORE the loads: a stream without rows in this step leaves) took
            // the preloaded slot of `base` (the cache length, first needed after the loads) — its pointer was a kernarg load and its entry a
            // second dependent one in front of every K/V request of a merged step —, and gx, a HIDDEN kernel argument (one more
            // scalar round trip before the first request of every cross-attention launch), rides in bits 24-31 of h_ns.
            const int* __restrict__ done, int K32, bf16_t* __restrict__ xout, size_t xplane, float* __restrict__ ml, float* __restrict__ po,
            int* __restrict__ ticket, PfJob pf, FQ fq, const unsigned long long* __restrict__ anc_tab, const int* __restrict__ base,
            // KV8 (cross_kv_fp8): kmat / vtmat are the e4m3 copies, one byte per element; the scales of the launch's streams [stream][head]
            const float* __restrict__ kscale, const float* __restrict__ vscale TL_ARG)
{
    static_assert(!KV8 || (CROSS && !FQ::kOn), "the fp8 K/V copy exists for the cross-attention only");
    // Mper query rows per stream as nqt tiles of <= 16 (a candidate tree of more than 16 nodes; the chain and every base pass: one tile);
    // blockIdx.z = stream * nqt + query tile
    const int Mper = mper_nbz & 0xff, nbz = mper_nbz >> 8, H = h_ns & 0xff, NS = (h_ns >> 8) & 0xff, nqt = max((h_ns >> 16) & 0xff, 1);
    const int gx = (h_ns >> 24) & 0xff;             // == gx
    extern __shared__ __attribute__((aligned(16))) char smem_attn[];
    if ((int)blockIdx.z >= nbz) {          // prefetch-only blocks (extra z slices): wm_skinny_gemm.h, PfJob
        const int main_total = gx * gridDim.y * nbz;
        pf_block(pf, main_total + (int)(blockIdx.x + gx * (blockIdx.y + gridDim.y * (blockIdx.z - nbz))) - pf_round8(main_total));
        return;
    }
    TL_BEGIN
    const int s = nqt > 1 ? (int)blockIdx.z / nqt : (int)blockIdx.z, r0 = ((int)blockIdx.z - s * nqt) * 16;
    // per-stream skip: the stream carried its hidden state, this base-pass row is not used (its K/V reads are saved)
    if (sskip && sskip[s]) return;
    // merged-step schedule: the stream's rows of this pass are the dense rows row_s .. + cnt_s (k_step_begin); otherwise s * Mper .. + Mper
    int row_s = s * Mper, cnt_s = Mper, pos_s = -1;
    if (sinfo) { const int4 si = sinfo[s]; row_s = si.x; cnt_s = si.y; pos_s = si.z; }
    if (r0 >= cnt_s) return;
    const int rows = min(16, cnt_s - r0);
    typedef AttnLds<CROSS ? WM_XATTN_SPB_MAX : 1> Lds;
    Lds& A = *reinterpret_cast<Lds*>(smem_attn);
    float (&s_m)[4][16] = A.s_m; float (&s_l)[4][16] = A.s_l; int& s_last = A.s_last;
    float (&s_o)[4][16][68] = A.s_o; auto& s_part = A.s_part;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const bool aw = w < 4;                  // the four attention waves (the fused variant may carry more waves for its LayerNorm)
    const int hd = blockIdx.y, d = H * 64;
    // CROSS: the block walks key splits [sp0, sp1); with gx == NS that is one split per block (single stream:
    // all CUs busy), with fewer blocks per (stream, head) each walks several (large batches: fewer partial hand-offs).
    // The arithmetic per split and the merge order over splits do not depend on the grouping: bit-identical outputs.
    const int spb = CROSS ? (NS + gx - 1) / gx : 1;
    const int sp0 = CROSS ? blockIdx.x * spb : 0, sp1 = CROSS ? min(NS, sp0 + spb) : 1;

    const bf16_t* kp = kmat + ((size_t)s * H + hd) * rows_alloc * 64;
    const bf16_t* vp = vtmat + ((size_t)s * H + hd) * 64 * rows_alloc;
    const unsigned char* kp8 = reinterpret_cast<const unsigned char*>(kmat) + ((size_t)s * H + hd) * rows_alloc * 64;       // KV8: the same element index, one byte each
    const unsigned char* vp8 = reinterpret_cast<const unsigned char*>(vtmat) + ((size_t)s * H + hd) * rows_alloc * 64;

    // The launch's memory batch goes out before anything is waited for: K/V do not depend on q nor on the cache length.
    // Self: wave w walks the 32-key steps w, w+4, ... with the next step in flight; its first step is fetched whatever
    // the length turns out to be (the rows exist: rows_alloc is a multiple of 32).  Cross: a wave owns 64 keys (two steps)
    // of every split and keeps the same two steps of the NEXT split in flight while it works on this one (16 KiB per wave,
    // 3 blocks per CU): bytes in flight are what the cross-K/V stream needs.  Then q (written by the previous launch).
    typename std::conditional<KV8, KVStep8, KVStep>::type c0 = {}, c1 = {};
    int kb = CROSS ? sp0 * 256 + w * 64 : 32 * w;
    if constexpr (KV8) {
        if (aw && kb < S) kv_load8(c0, kp8, vp8, kb, c, g, lane);
        if (aw && kb + 32 < S) kv_load8(c1, kp8, vp8, kb + 32, c, g, lane);
    } else {
        if (aw && (CROSS ? (kb < S) : (kb < rows_alloc))) kv_load<NT>(c0, kp, vp, kb, c, g, lane);
        if (aw && CROSS && kb + 32 < S) kv_load<NT>(c1, kp, vp, kb + 32, c, g, lane);
    }
    float ksc = 1.f, vsc = 1.f;
    if constexpr (KV8) { ksc = kscale[s * H + hd]; vsc = vscale[s * H + hd]; }
    float4 qraw[2][2];
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) { qraw[ds][0] = make_float4(0.f, 0.f, 0.f, 0.f); qraw[ds][1] = qraw[ds][0]; }
    if constexpr (FQ::kOn) {
        constexpr int NK = FQ::NK, KS = FQ::KS, KT = NK * KS;          // K32 == KT
        // LDS: [normalised hi | lo fragments, 2 KT KiB — dead before the attention state A (which aliases it) is first written]
        //      [gamma | beta | LayerNorm statistics: the LdNorm layout][q tile 16 rows x 64 features fp32]
        constexpr size_t kFragBytes = (size_t)2 * KT * 1024;
        constexpr size_t kLnOff = kFragBytes > sizeof(Lds) ? kFragBytes : ((sizeof(Lds) + 15) & ~(size_t)15);
        char* lnb = smem_attn + kLnOff;
        float* qt = reinterpret_cast<float*>(lnb + fq.ln.lds_bytes());
        typename LdNorm::template Regs<NK> xr;
        fq.ln.template issue<NK>(xr, lnb, w * NK, lane, 0, w < KS);     // K-slice w of the token rows; every wave helps stage gamma / beta
        // projection weights of this wave's 16-feature row tile: the first slices now, the rest once the LayerNorm registers are free
        constexpr int KT0 = ((KS + 1) / 2) * NK;
        const bf16_t* wq = fq.Wq + ((size_t)(hd * 4 + (aw ? w : 0)) * KT * 64 + lane) * 8;
        u32x4_t a[KT];
#pragma unroll
        for (int u = 0; u < KT0; ++u) a[u] = aw ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wq + (size_t)u * 512)) : u32x4_t{0u, 0u, 0u, 0u};
        const float4 qb = aw ? *reinterpret_cast<const float4*>(fq.bq + hd * 64 + 16 * w + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        fq.ln.template stage<NK>(xr, lnb);                             // gamma | beta registers -> LDS (behind the last request)
        if (done && *done) return;
        fq.ln.template stats<NK>(xr, lnb, w, KS, w < KS, lane);         // block barrier inside
        if (w < KS) {
#pragma unroll
            for (int u = 0; u < NK; ++u) {
                bf16x8_t bh, bl;
                fq.ln.template frag_hilo<NK>(xr, lnb, u, w * NK + u, lane, bh, bl);
                const size_t o = ((size_t)(w * NK + u) * 64 + lane) * 16;
                *reinterpret_cast<uint4*>(smem_attn + o) = __builtin_bit_cast(uint4, bh);
                *reinterpret_cast<uint4*>(smem_attn + (size_t)KT * 1024 + o) = __builtin_bit_cast(uint4, bl);
            }
        }
#pragma unroll
        for (int u = KT0; u < KT; ++u) a[u] = aw ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wq + (size_t)u * 512)) : u32x4_t{0u, 0u, 0u, 0u};
        __syncthreads();
        if (aw) {
            f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NK; ++u) {
                    const int kt = sl * NK + u;
                    const size_t o = ((size_t)kt * 64 + lane) * 16;
                    const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem_attn + o));
                    const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem_attn + (size_t)KT * 1024 + o));
                    const bf16x8_t av = __builtin_bit_cast(bf16x8_t, a[kt]);
                    acc = mfma16(av, bh, acc);
                    acc = mfma16(av, bl, acc);
                }
                tot[0] += acc[0]; tot[1] += acc[1]; tot[2] += acc[2]; tot[3] += acc[3];
            }
            // (v + bias) * 0.125 exactly like EpF32 of the stand-alone projection; lane holds features 16 w + 4 g .. + 3 of token row c
            *reinterpret_cast<float4*>(qt + c * 64 + 16 * w + 4 * g) =
                make_float4((tot[0] + qb.x) * 0.125f, (tot[1] + qb.y) * 0.125f, (tot[2] + qb.z) * 0.125f, (tot[3] + qb.w) * 0.125f);
        }
        __syncthreads();                    // q tile complete; the fragment staging is dead: A may be written from here on
        if (aw && c < rows) {
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                const float4* qp = reinterpret_cast<const float4*>(qt + (s * Mper + c) * 64 + ds * 32 + g * 8);      // (fused variant: single-tile passes only)
                qraw[ds][0] = qp[0]; qraw[ds][1] = qp[1];
            }
        }
    } else {
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
            if (c < rows) {
                // (KV8: the fp8 K rows are loaded 16 consecutive dims per lane; fragment ds of lane g then holds dims 16 g + 8 ds .. + 7)
                const float4* qp = reinterpret_cast<const float4*>(q + (size_t)(row_s + r0 + c) * d + hd * 64 + (KV8 ? g * 16 + ds * 8 : ds * 32 + g * 8));
                qraw[ds][0] = qp[0]; qraw[ds][1] = qp[1];
            }
        }
    }
    const int b0 = CROSS ? 0 : (pos_s >= 0 ? pos_s : base[s]);
    if (done && *done) return;          // all streams finished: checked after the batch went out (off the critical path)
    // keys < limit are visible to query c; a candidate-tree node sees the history and (through `anc`) its ancestors' rows
    const unsigned long long anc = (!CROSS && anc_tab && c < rows) ? anc_tab[r0 + c] : 0ull;
    const int limit = CROSS ? S : ((!CROSS && anc_tab) ? min(b0, rows_alloc) : min(b0 + r0 + c + 1, rows_alloc));
    int kend = CROSS ? min(S, kb + 64) : min(b0 + cnt_s, rows_alloc);
    bf16x8_t qhi[2], qlo[2];
    if constexpr (KV8) {            // K's scale rides on the query: (q s_k) . k8 = q . (s_k k8)
#pragma unroll
        for (int ds = 0; ds < 2; ++ds)
#pragma unroll
            for (int e = 0; e < 2; ++e) { qraw[ds][e].x *= ksc; qraw[ds][e].y *= ksc; qraw[ds][e].z *= ksc; qraw[ds][e].w *= ksc; }
    }
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) split_hilo8(qraw[ds][0], qraw[ds][1], qhi[ds], qlo[ds]);
    TL_PREP
    const int qr = threadIdx.x >> 4, ch = (threadIdx.x & 15) * 4;
    const int row = row_s + r0 + qr;
    typedef unsigned long long u64;
    float M = -INFINITY, L = 0.f; float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int sp = sp0; sp < sp1; ++sp) {
        AttnAcc st;
        st.m_run = -INFINITY; st.l_run = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st.o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (CROSS) {
            // each step's registers are refilled with the same step of the NEXT split as soon as its math has issued
            const int nkb = (sp + 1) * 256 + w * 64, nkend = (sp + 1 < sp1) ? min(S, nkb + 64) : 0;
            if (aw && kb < kend) attn_block64(st, kv_widen(c0), kv_widen(c1), qhi, qlo, kb, g, limit, kb + 32 < kend);
            if constexpr (KV8) {
                if (aw && nkb < nkend) kv_load8(c0, kp8, vp8, nkb, c, g, lane);
                if (aw && nkb + 32 < nkend) kv_load8(c1, kp8, vp8, nkb + 32, c, g, lane);
            } else {
                if (aw && nkb < nkend) kv_load<NT>(c0, kp, vp, nkb, c, g, lane);
                if (aw && nkb + 32 < nkend) kv_load<NT>(c1, kp, vp, nkb + 32, c, g, lane);
            }
            kb = nkb; kend = nkend;
        } else {
            for (; kb < kend; kb += 128) {
                const bool second = CROSS ? false : ((kb >> 7) & 1);        // two register sets alternate: the next step is in flight
                if constexpr (!KV8) {
                    if (kb + 128 < kend) { if (second) kv_load<NT>(c0, kp, vp, kb + 128, c, g, lane); else kv_load<NT>(c1, kp, vp, kb + 128, c, g, lane); }
                    if (second) attn_step(st, c1, qhi, qlo, kb, g, limit, b0, anc); else attn_step(st, c0, qhi, qlo, kb, g, limit, b0, anc);
                }
            }
        }
        const float m_run = st.m_run, l_run = st.l_run;
        const f32x4_t (&o)[4] = st.o;
        // merge the 4 waves' partials of this split (fixed order)
        if (aw) {
            if (g == 0) { s_m[w][c] = m_run; s_l[w][c] = l_run; }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4*>(&s_o[w][c][dt * 16 + 4 * g]) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
        }
        __syncthreads();
        M = -INFINITY; L = 0.f; acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qr < rows) {
            M = fmaxf(fmaxf(s_m[0][qr], s_m[1][qr]), fmaxf(s_m[2][qr], s_m[3][qr]));
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const float e = (s_m[ww][qr] == -INFINITY) ? 0.f : __expf(s_m[ww][qr] - M);
                L += s_l[ww][qr] * e;
                const float4 ov = *reinterpret_cast<const float4*>(&s_o[ww][qr][ch]);
                acc.x += ov.x * e; acc.y += ov.y * e; acc.z += ov.z * e; acc.w += ov.w * e;
            }
        }
        if (!CROSS) break;
        if (qr < rows) {
            // keep the partial for this block's own merge (LDS) and, when other blocks share the (stream, head),
            // publish it: relaxed agent-scope atomics (write-through sc1), see the hand-off note below
            *reinterpret_cast<float4*>(&s_part[sp - sp0][qr][ch]) = acc;
            if (ch == 0) { s_part[sp - sp0][qr][64] = M; s_part[sp - sp0][qr][65] = L; }
            if (gx > 1) {
                u64* pd = reinterpret_cast<u64*>(po + (((size_t)row * H + hd) * NS + sp) * 64 + ch);
                __hip_atomic_store(pd, ((u64)__float_as_uint(acc.y) << 32) | __float_as_uint(acc.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pd + 1, ((u64)__float_as_uint(acc.w) << 32) | __float_as_uint(acc.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ch == 0)
                    __hip_atomic_store(reinterpret_cast<u64*>(ml + (((size_t)row * H + hd) * NS + sp) * 2),
                                       ((u64)__float_as_uint(L) << 32) | __float_as_uint(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (sp + 1 < sp1) __syncthreads();                 // s_o / s_m / s_l are rewritten by the next split
    }
    TL_MID
    if (!CROSS) {
        if (qr < rows) {
            const float inv = 1.0f / L;
            const size_t oi = packed_index(row, hd * 64 + ch, K32);
            act_st4(xout + oi, xout + xplane + oi, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
        }
        TL_END
        return;
    }
    // ---- the last block of the (stream, head) merges all NS partials ----
    // Hand-off in the "8-byte agent-scope atomics on both sides" form (cdna_hip_programming.md G16): payload
    // stores are relaxed agent-scope atomics (write-through sc1), every storing wave drains vmcnt, one lane takes
    // a relaxed ticket; the last arriver reads the other partials with relaxed agent-scope atomic loads (L1 bypass).
    if (gx > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(ticket + blockIdx.z * H + hd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == gx - 1);
            if (last) __hip_atomic_store(ticket + blockIdx.z * H + hd, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch / replay
            s_last = last;
        }
        __syncthreads();
        if (!s_last) { TL_END return; }
    } else __syncthreads();
    if (qr < rows) {
        // merge the NS partials in split order (the same arithmetic whether they come from LDS or from other blocks)
        float ms[WM_XATTN_NS_MAX], ls[WM_XATTN_NS_MAX]; float4 ov[WM_XATTN_NS_MAX];
        if (gx > 1) {
            // every partial (this block's own included: they were published above) is fetched in ONE batch of relaxed
            // agent-scope loads (L1 bypass) — one memory round trip for the last-arriving block instead of one per split
            const u64* mlp = reinterpret_cast<const u64*>(ml + ((size_t)row * H + hd) * NS * 2);
            const u64* op = reinterpret_cast<const u64*>(po + ((size_t)row * H + hd) * NS * 64 + ch);
            u64 rml[WM_XATTN_NS_MAX], ro0[WM_XATTN_NS_MAX], ro1[WM_XATTN_NS_MAX];
#pragma unroll
            for (int sp = 0; sp < WM_XATTN_NS_MAX; ++sp) {
                const int spc = min(sp, NS - 1);            // clamped: loads stay unconditional (straight-line, one batch)
                rml[sp] = __hip_atomic_load(mlp + spc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ro0[sp] = __hip_atomic_load(op + (size_t)spc * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ro1[sp] = __hip_atomic_load(op + (size_t)spc * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int sp = 0; sp < WM_XATTN_NS_MAX; ++sp) {
                ms[sp] = __uint_as_float((unsigned)rml[sp]); ls[sp] = __uint_as_float((unsigned)(rml[sp] >> 32));
                ov[sp] = make_float4(__uint_as_float((unsigned)ro0[sp]), __uint_as_float((unsigned)(ro0[sp] >> 32)),
                                     __uint_as_float((unsigned)ro1[sp]), __uint_as_float((unsigned)(ro1[sp] >> 32)));
            }
        } else {
#pragma unroll
            for (int sp = 0; sp < WM_XATTN_NS_MAX; ++sp) {
                const int spc = min(sp, NS - 1);
                ms[sp] = s_part[spc][qr][64]; ls[sp] = s_part[spc][qr][65];
                ov[sp] = *reinterpret_cast<const float4*>(&s_part[spc][qr][ch]);
            }
        }
        float Mx = -INFINITY;
#pragma unroll
        for (int sp = 0; sp < WM_XATTN_NS_MAX; ++sp) if (sp < NS) Mx = fmaxf(Mx, ms[sp]);
        float Lt = 0.f; float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int sp = 0; sp < WM_XATTN_NS_MAX; ++sp) {
            if (sp < NS) {
                const float e = (ms[sp] == -INFINITY) ? 0.f : __expf(ms[sp] - Mx);
                Lt += ls[sp] * e;
                o4.x += ov[sp].x * e; o4.y += ov[sp].y * e; o4.z += ov[sp].z * e; o4.w += ov[sp].w * e;
            }
        }
        const float inv = vsc / Lt;             // (KV8: V's scale rides on the normalised output)
        const size_t oi = packed_index(row, hd * 64 + ch, K32);
        act_st4(xout + oi, xout + xplane + oi, make_float4(o4.x * inv, o4.y * inv, o4.z * inv, o4.w * inv));
    }
    TL_END
}

// ---------------------------------------------------------------------------------------------
// select: logits processors (F7) + argmax (F9) + typical-acceptance statistics (F11), split over
// SEL_SP blocks per logits row.
//   row -> stream s = row / rps, slot i = row % rps;  cur_len = L[s] (same for every row, model.py:689-694)
//   k_select1: per-slice (max, first argmax, sum exp(x - max_slice))        -> part1[row][sp][3]
//   k_select2: global max / Z / argmax from the slices, then the slice's share of
//              -sum p log(p + 1e-5) (medusa_utils.py:566-568)               -> part2[row][sp]; slice 0 also
//              writes argmax and p(candidate_{i+1}).
// ---------------------------------------------------------------------------------------------
#define SEL_SP 16

__device__ __forceinline__ float proc_logit(float x, int n, int cur_len, const GenDev& gp, const unsigned char* mask,
                                            const float* exppen)
{
    if (n == gp.eos && gp.exp_start >= 0 && cur_len > gp.exp_start) x += fabsf(x) * exppen[cur_len];
    const unsigned char mk = mask[n];
    if ((mk & 1) || ((mk & 2) && cur_len == gp.begin)) x = -INFINITY;
    return x;
}

// Sibling rows (wm_config.sibling_rows; one stream): the tokens of nodes K+1 .. K+S of the verify pass = head 1's top-2 .. top-(S+1) processed
// logits (descending, lower index first on equal values), leaves under the root at position L + 1.  They never enter the acceptance rule
// (medusa_utils.py:526-641 runs on the chain); k_accept only asks whether the next root — argmax v_0 after an accept length of 0 — is one of them.
// Selection rides on the candidate stage's own launches: the SEL_SP slice blocks of k_select1 that score head 1's row also keep a per-thread sorted
// top-6 and pop their slice's six winners (block-wide arg-max rounds) into `sibpart`; k_cand_fin — the one-block kernel that turns the slice
// partials into the chain's candidates — merges the SEL_SP x 6 partials the same way.  (Rounds of this: one block over the whole row, 131 us per
// iteration — more than the base passes it saved; 32 slice blocks + last-arriver merge behind two fences, 18.7 us; this form adds no launch.)
__device__ __forceinline__ void sib_block_pop(float (&tv)[6], int (&ti)[6], float* sv, int* si, int tid, float& mx, int& mi)
{
    const int lane = tid & 63, w = tid >> 6;
    mx = tv[0]; mi = ti[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    __syncthreads();                                    // (sv / si may still be read by the caller's previous round)
    if (lane == 0) { sv[w] = mx; si[w] = mi; }
    __syncthreads();
    mx = sv[0]; mi = si[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) if (sv[q] > mx || (sv[q] == mx && si[q] < mi)) { mx = sv[q]; mi = si[q]; }
    if (ti[0] == mi && mi != 0x7fffffff) {              // token indices are distinct: exactly one thread owns the winner
#pragma unroll
        for (int q = 0; q < 5; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
        tv[5] = -INFINITY; ti[5] = 0x7fffffff;
    }
}
__device__ __forceinline__ void sib_insert(float (&tv)[6], int (&ti)[6], float v, int vi)
{
    if (v > tv[5] || (v == tv[5] && vi < ti[5])) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (v > tv[j] || (v == tv[j] && vi < ti[j])) { const float fv = tv[j]; const int fi = ti[j]; tv[j] = v; ti[j] = vi; v = fv; vi = fi; }
    }
}

// NEED_Z = false (candidate stage: only the arg-max is consumed) skips the second sweep, the slice's softmax denominator
template <bool NEED_Z = true>
__global__ void __launch_bounds__(256)
k_select1(const float* __restrict__ logits, GenDev gp, const unsigned char* __restrict__ mask, const float* __restrict__ exppen,
          const int* __restrict__ L, int rps, float* __restrict__ part1, const int4* __restrict__ rowinfo = nullptr,
          float2* __restrict__ sibpart = nullptr)
{
    __shared__ float sv[4]; __shared__ int si[4]; __shared__ float sz[4];
    const int row = blockIdx.y, sp = blockIdx.x;
    int s = row / rps;
    const bool sibrow = sibpart != nullptr && row - s * rps == 1;       // head 1's row of a stream whose verify pass carries sibling rows
    if (rowinfo) {                          // merged-step schedule: dense rows; only a stream's verify rows are scored
        const int4 ri = rowinfo[row];
        if (ri.w != 2) return;
        s = ri.x;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cur_len = L[s];
    const float* x = logits + (size_t)row * gp.Vpad;
    const int per = (gp.V + SEL_SP - 1) / SEL_SP, n0 = sp * per, n1 = min(gp.V, n0 + per);
    float mx = -INFINITY; int mi = 0x7fffffff;
    if (sibrow) {
        // the slice's six best (the first of them is the slice arg-max the chain needs)
        float tv[6]; int ti[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
        for (int n = n0 + tid; n < n1; n += 256) sib_insert(tv, ti, proc_logit(x[n], n, cur_len, gp, mask, exppen), n);
        float2* mine = sibpart + ((size_t)s * SEL_SP + sp) * 6;
        float m0 = -INFINITY; int i0 = 0x7fffffff;
        for (int j = 0; j < 6; ++j) {
            float pm; int pi;
            sib_block_pop(tv, ti, sv, si, tid, pm, pi);
            if (j == 0) { m0 = pm; i0 = pi; }
            if (tid == 0) mine[j] = make_float2(pm, __int_as_float(pi));
        }
        __syncthreads();
        mx = m0; mi = i0;
    } else {
        for (int n = n0 + tid; n < n1; n += 256) {
            const float v = proc_logit(x[n], n, cur_len, gp, mask, exppen);
            if (v > mx || (v == mx && n < mi)) { mx = v; mi = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(mi, o, 64);
            if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
        }
        if (lane == 0) { sv[w] = mx; si[w] = mi; }
        __syncthreads();
        mx = sv[0]; mi = si[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (sv[k] > mx || (sv[k] == mx && si[k] < mi)) { mx = sv[k]; mi = si[k]; }
    }
    float z = 0.f;
    if (NEED_Z && mx != -INFINITY)
        for (int n = n0 + tid; n < n1; n += 256) {
            const float v = proc_logit(x[n], n, cur_len, gp, mask, exppen);
            z += (v == -INFINITY) ? 0.f : expf((v - mx) * gp.inv_temp);
        }
    z = wave_sum(z);
    if (lane == 0) sz[w] = z;
    __syncthreads();
    if (tid == 0) {
        float* o = part1 + ((size_t)row * SEL_SP + sp) * 4;
        o[0] = mx; o[1] = __int_as_float(mi); o[2] = (sz[0] + sz[1]) + (sz[2] + sz[3]);
    }
}

__global__ void __launch_bounds__(256)
k_select2(const float* __restrict__ logits, GenDev gp, const unsigned char* __restrict__ mask, const float* __restrict__ exppen,
          const int* __restrict__ L, const int* __restrict__ cand, int rps, int out_row0, const float* __restrict__ part1,
          float* __restrict__ part2, int* __restrict__ amax, float* __restrict__ pc, const TreeDev* __restrict__ tree,
          const int4* __restrict__ rowinfo = nullptr)
{
    __shared__ float sh[4];
    const int row = blockIdx.y, sp = blockIdx.x;
    int s = row / rps, i = row - s * rps;
    if (rowinfo) {
        const int4 ri = rowinfo[row];
        if (ri.w != 2) return;
        s = ri.x; i = ri.y;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cur_len = L[s];
    const float* x = logits + (size_t)row * gp.Vpad;
    const float* p1 = part1 + (size_t)row * SEL_SP * 4;
    float mx = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < SEL_SP; ++k) {
        const float v = p1[4 * k]; const int idx = __float_as_int(p1[4 * k + 1]);
        if (v > mx || (v == mx && idx < mi)) { mx = v; mi = idx; }
    }
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < SEL_SP; ++k) {
        const float v = p1[4 * k];
        z += (v == -INFINITY) ? 0.f : p1[4 * k + 2] * expf((v - mx) * gp.inv_temp);
    }
    const float invz = 1.0f / z;
    const int per = (gp.V + SEL_SP - 1) / SEL_SP, n0 = sp * per, n1 = min(gp.V, n0 + per);
    float hs = 0.f;
    for (int n = n0 + tid; n < n1; n += 256) {
        const float v = proc_logit(x[n], n, cur_len, gp, mask, exppen);
        const float p = (v == -INFINITY) ? 0.f : expf((v - mx) * gp.inv_temp) * invz;
        hs += p * logf(p + 1e-5f);
    }
    hs = wave_sum(hs);
    if (lane == 0) sh[w] = hs;
    __syncthreads();
    if (tid == 0) {
        const int orow = out_row0 + row;
        part2[(size_t)orow * SEL_SP + sp] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (sp == 0) {
            amax[orow] = mi;
            if (tree) {
                // candidate tree: this row is node i's distribution; every child's token is scored under it (pc indexed by child)
                if (i == 0) pc[out_row0 + s * rps] = 0.f;
                for (int j = 0; j < 4; ++j) {
                    const int ch = tree->children[i][j];
                    if (ch < 0) break;
                    const int c = cand[s * WM_CAND_STRIDE + ch];
                    const float vc = proc_logit(x[c], c, cur_len, gp, mask, exppen);
                    pc[out_row0 + s * rps + ch] = (vc == -INFINITY) ? 0.f : expf((vc - mx) * gp.inv_temp) * invz;
                }
            } else {
                float pcv = 0.f;
                if (i + 1 < rps) {
                    const int c = cand[s * WM_CAND_STRIDE + i + 1];
                    const float vc = proc_logit(x[c], c, cur_len, gp, mask, exppen);
                    pcv = (vc == -INFINITY) ? 0.f : expf((vc - mx) * gp.inv_temp) * invz;
                }
                pc[orow] = pcv;
            }
        }
    }
}

// argmax of each row from the slice partials (base pass candidates / vanilla token)
__global__ void k_select_argmax(const float* __restrict__ part1, int nrows, int out_row0, int* __restrict__ amax)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    const float* p1 = part1 + (size_t)row * SEL_SP * 4;
    float mx = -INFINITY; int mi = 0x7fffffff;
    for (int k = 0; k < SEL_SP; ++k) {
        const float v = p1[4 * k]; const int idx = __float_as_int(p1[4 * k + 1]);
        if (v > mx || (v == mx && idx < mi)) { mx = v; mi = idx; }
    }
    amax[out_row0 + row] = mi;
}

// batched hidden-state carry, Medusa-Block: rows of carrying streams take the saved block-layer output
__global__ void k_rows_take_carried(float* __restrict__ dst, const float* __restrict__ keep, const int* __restrict__ carry, int d, int M,
                                    const int* __restrict__ done)
{
    if (done && *done) return;
    const int m = blockIdx.x;
    if (m >= M || !carry[m]) return;
    const float4* sp = reinterpret_cast<const float4*>(keep + (size_t)m * d);
    float4* dp = reinterpret_cast<float4*>(dst + (size_t)m * d);
    for (int j = threadIdx.x; j < (d >> 2); j += blockDim.x) dp[j] = sp[j];
}

// Candidates of the base pass: cand[s][i] = argmax of head i (medusa_utils.py:446-458, top-1 chain), from the slice partials in ONE launch (one
// block per stream): row i's arg-max over its SEL_SP slices -> amax, cand[s][i] (rounds 1-5: k_select_argmax + a k_set_cand launch); with sibling rows, the S + 1 best of head 1's SEL_SP x 6 slice winners -> cand[s][K+1 ..].
__global__ void __launch_bounds__(256)
k_cand_fin(const float* __restrict__ part1, GenDev gp, int* __restrict__ amax, int* __restrict__ cand, const float2* __restrict__ sibpart)
{
    __shared__ float sv[4]; __shared__ int si[4];
    const int s = blockIdx.x, tid = threadIdx.x, rps = gp.K + 1;
    if (tid < rps) {
        const float* p1 = part1 + (size_t)(s * rps + tid) * SEL_SP * 4;
        float mx = -INFINITY; int mi = 0x7fffffff;
        for (int k = 0; k < SEL_SP; ++k) {
            const float v = p1[4 * k]; const int idx = __float_as_int(p1[4 * k + 1]);
            if (v > mx || (v == mx && idx < mi)) { mx = v; mi = idx; }
        }
        amax[s * rps + tid] = mi;
        cand[s * WM_CAND_STRIDE + tid] = mi;
    }
    if (gp.sib <= 0 || sibpart == nullptr) return;
    float tv[6]; int ti[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    if (tid < SEL_SP * 6) {
        const float2 p = sibpart[(size_t)s * SEL_SP * 6 + tid];
        tv[0] = p.x; ti[0] = __float_as_int(p.y);
    }
    for (int j = 0; j <= gp.sib; ++j) {
        float mx; int mi;
        sib_block_pop(tv, ti, sv, si, tid, mx, mi);
        if (tid == 0 && j > 0) cand[s * WM_CAND_STRIDE + gp.K + j] = (mi == 0x7fffffff) ? -1 : mi;      // (-1: fewer than j + 1 unsuppressed tokens — never a hit)
    }
}

// merged-step schedule: what every stream contributes to this step's pass, as DENSE rows.  carry[s] (k_accept of the previous step /
// iteration): 1 = the post-LN state of the stream's next base token is in hf_keep -> its K + 1 candidates are verified now (rows at L ..);
// 0 = the stream accepted nothing: its next token ids[kvlen] needs the base pass first -> ONE row at position kvlen.
// rowinfo[row] = {stream, index inside the stream, position of the stream's row 0, kind: 0 no row / 1 base row / 2 verify row} (.w doubles as
// "row count class": the consumers need cnt = 1 or rps, derived from it), sinfo[s] = {first dense row, row count, position of row 0, mode},
// steprows = {rows, 16-row tiles, passes that carried rows since wm_decode_begin}.  A finished stream contributes no rows.  One block;
// streams <= 1024.
__global__ void __launch_bounds__(256)
k_step_begin(const int* __restrict__ carry, const int* __restrict__ L, const int* __restrict__ kvlen, const int* __restrict__ finished,
             int4* __restrict__ rowinfo, int4* __restrict__ sinfo, int* __restrict__ steprows, int rps, int B, int Rmax)
{
    __shared__ int s_start[1025];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int s = 0; s < B; ++s) { s_start[s] = acc; acc += finished[s] ? 0 : (carry[s] ? rps : 1); }
        s_start[B] = acc;
        steprows[0] = acc; steprows[1] = (acc + 15) >> 4;
        if (acc > 0) steprows[2] += 1;          // passes that carried rows (wm_stats.schedule_steps)
    }
    __syncthreads();
    const int nrows = s_start[B];
    for (int s = threadIdx.x; s < B; s += blockDim.x) {
        const int m = carry[s] ? 1 : 0, cnt = finished[s] ? 0 : (m ? rps : 1), pb = m ? L[s] : kvlen[s], r0 = s_start[s];
        sinfo[s] = make_int4(r0, cnt, pb, m);
        for (int r = 0; r < cnt; ++r) rowinfo[r0 + r] = make_int4(s, r, pb, m ? 2 : 1);
    }
    for (int row = nrows + threadIdx.x; row < Rmax; row += blockDim.x) rowinfo[row] = make_int4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// accept: one wavefront per stream.  Lane i (< K) evaluates candidate i+1; the leading-true count of
// the wave ballot is the accept length a (medusa_utils.py:573-577 cumprod-sum for one candidate path).
// Then emit tokens, keep the accepted provisional KV rows by advancing kvlen (model.py:378-402), apply
// the stop rules (model.py:774-793).
// ---------------------------------------------------------------------------------------------
__global__ void k_accept(GenDev gp, const int* __restrict__ cand, const int* __restrict__ amax, const float* __restrict__ pc,
                         const float* __restrict__ part2, int* __restrict__ ids, int* __restrict__ L, int* __restrict__ kvlen,
                         int* __restrict__ finished, int* __restrict__ niter, long long* __restrict__ hist, int* __restrict__ done, int B,
                         int* __restrict__ carry, const float* __restrict__ hf, float* __restrict__ hf_keep, int d,
                         int* __restrict__ hostflags, const float* __restrict__ hb, float* __restrict__ hb_keep,
                         const int4* __restrict__ sinfo = nullptr, int* __restrict__ sel_src = nullptr, int* __restrict__ sel_n = nullptr,
                         int* __restrict__ sel_base = nullptr)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (finished[s]) { if (sel_n && lane == 0) sel_n[s] = 0; return; }
    const int K = gp.K, rps = K + 1;
    const int row0 = sinfo ? sinfo[s].x : s * rps;            // merged-step schedule: the stream's first dense row of this pass
    if (sinfo != nullptr && sinfo[s].w == 0) {
        // merged-step schedule: this stream's one row was its base pass (token ids[kvlen] at position kvlen; its K / V row is in the cache):
        // keep the post-LN state for the heads of the next step, nothing is verified or emitted
        const float4* srcp = reinterpret_cast<const float4*>(hf + (size_t)row0 * d);
        float4* dstp = reinterpret_cast<float4*>(hf_keep + (size_t)s * d);
        for (int j = lane; j < (d >> 2); j += 64) dstp[j] = srcp[j];
        if (hb) {
            const float4* bs = reinterpret_cast<const float4*>(hb + (size_t)row0 * d);
            float4* bd = reinterpret_cast<float4*>(hb_keep + (size_t)s * d);
            for (int j = lane; j < (d >> 2); j += 64) bd[j] = bs[j];
        }
        if (lane == 0) { kvlen[s] = L[s]; carry[s] = 1; }
        return;
    }
    bool ok = false;
    if (lane < K) {
        if (gp.accept_mode == WM_ACCEPT_GREEDY) ok = (cand[s * WM_CAND_STRIDE + lane + 1] == amax[row0 + lane]);
        else {
            const float* hp = part2 + (size_t)(row0 + lane) * SEL_SP;
            float hsum = 0.f;
#pragma unroll
            for (int k = 0; k < SEL_SP; ++k) hsum += hp[k];
            const float thr = fminf(gp.thr, gp.alpha * expf(hsum));            // H = -hsum
            ok = pc[row0 + lane] > thr;
        }
    }
    const unsigned long long m = __ballot(ok);
    const unsigned long long inv = ~m;
    int a = (inv == 0ull) ? 64 : (__ffsll((long long)inv) - 1);
    if (a > K) a = K;
    if (gp.force_accept >= 0) a = min(gp.force_accept, K);        // benchmark knob (wm.h): iteration cost at a given acceptance
    const int Lcur = L[s];
    const int n_emit = (a == 0) ? 2 : a + 1;
    int tok = -1;
    if (lane < n_emit) {
        tok = (a == 0 && lane == 1) ? amax[row0] : cand[s * WM_CAND_STRIDE + lane];
        if (Lcur + lane < gp.Tids) ids[(size_t)s * gp.Tids + Lcur + lane] = tok;
    }
    const bool hit_eos = __ballot(lane < n_emit && tok == gp.eos) != 0ull;
    // hidden-state carry (a > 0): row a of the verify pass saw exactly the accepted prefix, so its post-LN state IS
    // what the next base pass would recompute for token c_a, and its K/V row is already in the cache: keep a+1 rows,
    // save the row, and let the next (redundant) base pass skip its layers.  Bit-identical tokens, half the passes.
    // sibling rows (gp.sib > 0: one stream): nothing accepted, but the next root — argmax v_0, emitted above — was in the pass as leaf K + 1 + j under
    // the root: that row saw the history, the root and itself at position L + 1, i.e. exactly what the next base pass would compute.  Its state is
    // carried like row a's, its K/V rows move from provisional row K + 1 + j to row 1 (k_kv_compact), the base pass is skipped.
    int sib_row = -1;
    if (gp.sib > 0 && carry != nullptr && a == 0 && sel_n != nullptr && gp.force_accept < 0) {     // (a forced accept length prices the iteration AT that length: no hit)
        const int nxt = amax[row0];
        const unsigned long long hit = __ballot(lane < gp.sib && cand[s * WM_CAND_STRIDE + rps + lane] == nxt);
        if (hit) sib_row = rps + (__ffsll((long long)hit) - 1);
    }
    const bool do_carry = carry != nullptr && (a > 0 || sib_row >= 0);
    const int crow = (a > 0) ? a : sib_row;          // the pass row whose post-LN state is the next iteration's base state
    if (do_carry) {
        const float4* srcp = reinterpret_cast<const float4*>(hf + (size_t)(row0 + crow) * d);
        float4* dstp = reinterpret_cast<float4*>(hf_keep + (size_t)s * d);
        for (int j = lane; j < (d >> 2); j += 64) dstp[j] = srcp[j];
        if (hb) {                  // Medusa-Block: the heads read the extra layer's output of that row (model.py:1414-1417)
            const float4* bs = reinterpret_cast<const float4*>(hb + (size_t)(row0 + crow) * d);
            float4* bd = reinterpret_cast<float4*>(hb_keep + (size_t)s * d);
            for (int j = lane; j < (d >> 2); j += 64) bd[j] = bs[j];
        }
    }
    if (sel_n != nullptr && lane < 16) sel_src[s * 16 + lane] = (lane == 1 && sib_row >= 0) ? sib_row : lane;
    if (lane == 0) {
        const int Ln = Lcur + n_emit;
        L[s] = Ln;
        kvlen[s] = (a == 0 && sib_row < 0) ? Lcur + 1 : (do_carry ? Ln : Lcur + a);
        if (carry) carry[s] = do_carry ? 1 : 0;
        if (sel_n != nullptr) { sel_n[s] = sib_row >= 0 ? 2 : 0; sel_base[s] = Lcur; }
        if (sib_row >= 0) atomicAdd(reinterpret_cast<unsigned long long*>(hist + 17), 1ull);
        niter[s] += 1;
        atomicAdd(reinterpret_cast<unsigned long long*>(hist + a), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(hist + 16), (unsigned long long)n_emit);
        const bool fin = hit_eos || Ln >= gp.max_length || Ln + K >= gp.hard_max_length;
        if (fin) {
            finished[s] = 1;
            if (atomicAdd(done + 1, 1) == B - 1) done[0] = 1;           // done[1] counts finished streams
        }
        if (hostflags) { hostflags[0] = do_carry ? 1 : 0; hostflags[1] = fin ? 1 : 0; }   // host-mapped (single-stream runs)
    }
}

// ---------------------------------------------------------------------------------------------
// Candidate trees (medusa_choices with top-k > 1; generate_candidates, medusa_utils.py:424-458).
// k_tree_cand: one block per (stream, logits row k of the base pass): the top-c_k tokens of the processed row in
// descending order (ties: lower index first), written as the tokens of every node at depth k:
// node (k, j) <- top[j % c_k]  (tree_indices).  Row 0 (base head) has c = 1: the argmax.
// Every thread keeps a sorted top-4 of its strided share; c rounds of block-wide argmax over the threads' heads then pop
// the winners (the global top-c is inside the union of the per-thread top-c lists).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tree_cand(const float* __restrict__ logits, GenDev gp, const unsigned char* __restrict__ mask, const float* __restrict__ exppen,
            const int* __restrict__ L, const TreeDev* __restrict__ tree, int* __restrict__ cand, const int* __restrict__ done)
{
    __shared__ float sv[4]; __shared__ int si[4]; __shared__ int s_top[4];
    if (done && *done) return;
    const int rps = gp.K + 1, row = blockIdx.x, s = row / rps, k = row - s * rps;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cur_len = L[s], c = tree->topk[k];
    const float* x = logits + (size_t)row * gp.Vpad;
    float t0 = -INFINITY, t1 = -INFINITY, t2 = -INFINITY, t3 = -INFINITY;
    int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff;
    for (int n = tid; n < gp.V; n += 256) {
        const float v = proc_logit(x[n], n, cur_len, gp, mask, exppen);
        if (v > t3) {                                   // n ascends inside a thread: an equal value never displaces an earlier index
            if (v > t0)      { t3 = t2; i3 = i2; t2 = t1; i2 = i1; t1 = t0; i1 = i0; t0 = v; i0 = n; }
            else if (v > t1) { t3 = t2; i3 = i2; t2 = t1; i2 = i1; t1 = v; i1 = n; }
            else if (v > t2) { t3 = t2; i3 = i2; t2 = v; i2 = n; }
            else             { t3 = v; i3 = n; }
        }
    }
    for (int j = 0; j < c; ++j) {
        float mx = t0; int mi = i0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(mi, o, 64);
            if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
        }
        if (lane == 0) { sv[w] = mx; si[w] = mi; }
        __syncthreads();
        mx = sv[0]; mi = si[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) if (sv[q] > mx || (sv[q] == mx && si[q] < mi)) { mx = sv[q]; mi = si[q]; }
        if (tid == 0) s_top[j] = (mi == 0x7fffffff) ? 0 : mi;
        if (i0 == mi && mi != 0x7fffffff) { t0 = t1; i0 = i1; t1 = t2; i1 = i2; t2 = t3; i2 = i3; t3 = -INFINITY; i3 = 0x7fffffff; }
        __syncthreads();
    }
    if (tid < tree->cumprod[k]) cand[s * WM_CAND_STRIDE + tree->start[k] + tid] = s_top[tid % c];
}

// ---------------------------------------------------------------------------------------------
// accept over a candidate tree: one wavefront per stream, lane p = candidate path p (evaluate_posterior with several
// candidates, medusa_utils.py:526-588).  Per path the leading run of accepted nodes; the accept length is the maximum over
// paths; the chosen path is the first of maximal length (greedy) / the one of maximal length with the largest
// sum log p_c over its accepted nodes, first on ties (typical).  Emits the tokens, records which provisional K/V rows
// (tree nodes) form the accepted path (k_kv_compact moves them to rows L.., model.py:378-392) and carries the accepted
// node's post-LN state like k_accept.
// amax / pc / part2 rows are indexed [stream][node]: node n's logits row is ITS distribution (for its children); pc[n] is
// p_{parent(n)}(token_n).
// ---------------------------------------------------------------------------------------------
__global__ void k_accept_tree(GenDev gp, const TreeDev* __restrict__ tree, const int* __restrict__ cand, const int* __restrict__ amax,
                              const float* __restrict__ pc, const float* __restrict__ part2, int* __restrict__ ids, int* __restrict__ L,
                              int* __restrict__ kvlen, int* __restrict__ finished, int* __restrict__ niter, long long* __restrict__ hist,
                              int* __restrict__ done, int B, int* __restrict__ carry, const float* __restrict__ hf, float* __restrict__ hf_keep,
                              int d, int* __restrict__ hostflags, const float* __restrict__ hb, float* __restrict__ hb_keep,
                              int* __restrict__ sel_src, int* __restrict__ sel_n, int* __restrict__ sel_base)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (finished[s]) { if (lane == 0) sel_n[s] = 0; return; }
    const int K = gp.K, tn = tree->n_nodes, tp = tree->n_paths;
    // threshold of every node's distribution (typical mode): lane n < tn
    float thr_n = 0.f;
    if (gp.accept_mode != WM_ACCEPT_GREEDY && lane < tn) {
        const float* hp = part2 + (size_t)(s * tn + lane) * SEL_SP;
        float hsum = 0.f;
#pragma unroll
        for (int k = 0; k < SEL_SP; ++k) hsum += hp[k];
        thr_n = fminf(gp.thr, gp.alpha * expf(hsum));
    }
    const int p = min(lane, tp - 1);
    int a_p = 0; float lik = 0.f; bool alive = true;
    for (int i = 1; i <= K; ++i) {
        const int node = tree->retrieve[p][i], par = tree->retrieve[p][i - 1];
        const float thr = __shfl(thr_n, par, 64);
        bool ok; float pcv = 1.f;
        if (gp.accept_mode == WM_ACCEPT_GREEDY) ok = cand[s * WM_CAND_STRIDE + node] == amax[s * tn + par];
        else { pcv = pc[s * tn + node]; ok = pcv > thr; }
        if (alive && ok) { a_p += 1; lik += logf(pcv); } else alive = false;
    }
    if (lane >= tp) { a_p = -1; lik = -INFINITY; }
    int a = a_p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = max(a, __shfl_xor(a, o, 64));
    int best = 0;
    if (a > 0) {
        float bl = (a_p == a) ? lik : -INFINITY; int bi = (a_p == a) ? lane : 64;
        if (gp.accept_mode == WM_ACCEPT_GREEDY) bl = (a_p == a) ? 0.f : -INFINITY;          // first path of maximal length
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ol = __shfl_xor(bl, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ol > bl || (ol == bl && oi < bi)) { bl = ol; bi = oi; }
        }
        best = min(bi, tp - 1);
    }
    if (gp.force_accept >= 0) { a = min(gp.force_accept, K); best = 0; }
    const int Lcur = L[s];
    const int n_emit = (a == 0) ? 2 : a + 1;
    int tok = -1;
    if (lane < n_emit) {
        tok = (a == 0 && lane == 1) ? amax[s * tn + 0] : cand[s * WM_CAND_STRIDE + tree->retrieve[best][lane]];
        if (Lcur + lane < gp.Tids) ids[(size_t)s * gp.Tids + Lcur + lane] = tok;
    }
    const bool hit_eos = __ballot(lane < n_emit && tok == gp.eos) != 0ull;
    const bool do_carry = carry != nullptr && a > 0;
    const int n_keep = (a == 0) ? 1 : (do_carry ? a + 1 : a);           // provisional rows that stay: path nodes 0 .. n_keep-1
    const int my_src = (lane < n_keep) ? tree->retrieve[best][lane] : lane;
    if (lane < 16) sel_src[s * 16 + lane] = my_src;
    const bool moved = __ballot(lane < n_keep && my_src != lane) != 0ull;
    if (do_carry) {
        const int nd = tree->retrieve[best][a];
        const float4* srcp = reinterpret_cast<const float4*>(hf + (size_t)(s * tn + nd) * d);
        float4* dstp = reinterpret_cast<float4*>(hf_keep + (size_t)s * d);
        for (int j = lane; j < (d >> 2); j += 64) dstp[j] = srcp[j];
        if (hb) {
            const float4* bs = reinterpret_cast<const float4*>(hb + (size_t)(s * tn + nd) * d);
            float4* bd = reinterpret_cast<float4*>(hb_keep + (size_t)s * d);
            for (int j = lane; j < (d >> 2); j += 64) bd[j] = bs[j];
        }
    }
    if (lane == 0) {
        const int Ln = Lcur + n_emit;
        sel_n[s] = moved ? n_keep : 0; sel_base[s] = Lcur;
        L[s] = Ln;
        kvlen[s] = Lcur + n_keep;
        if (carry) carry[s] = do_carry ? 1 : 0;
        niter[s] += 1;
        atomicAdd(reinterpret_cast<unsigned long long*>(hist + a), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(hist + 16), (unsigned long long)n_emit);
        const bool fin = hit_eos || Ln >= gp.max_length || Ln + K >= gp.hard_max_length;
        if (fin) {
            finished[s] = 1;
            if (atomicAdd(done + 1, 1) == B - 1) done[0] = 1;
        }
        if (hostflags) { hostflags[0] = do_carry ? 1 : 0; hostflags[1] = fin ? 1 : 0; }
    }
}

// Moves the K / V^T rows of the accepted path's nodes (provisional rows base + src[i]) to rows base + i of every KV slot:
// one block per (head, stream, slot); all reads happen before the first write (src[i] >= i, but src[i] may be another
// destination).  K rows are 128-B lines; V is stored as V^T MFMA fragments (vfrag_index).
__global__ void __launch_bounds__(256)
k_kv_compact(bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, const int* __restrict__ sel_src, const int* __restrict__ sel_n,
             const int* __restrict__ sel_base, int H, int Tal, int maxB)
{
    const int hd = blockIdx.x, s = blockIdx.y, slot = blockIdx.z;
    const int n = sel_n[s];
    if (n == 0) return;
    const int base = sel_base[s];
    const int i = threadIdx.x >> 4, q = threadIdx.x & 15;
    const size_t off = (((size_t)slot * maxB + s) * H + hd) * (size_t)Tal * 64;
    bf16_t* kp = kc + off; bf16_t* vp = vc + off;
    uint2 kv = make_uint2(0u, 0u); bf16_t vv[4] = {0, 0, 0, 0};
    const bool act = i < n;
    if (act) {
        const int src = base + sel_src[s * 16 + i];
        kv = *reinterpret_cast<const uint2*>(kp + (size_t)src * 64 + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) vv[j] = vp[vfrag_index(src, q * 4 + j)];
    }
    __syncthreads();
    if (act) {
        const int dst = base + i;
        *reinterpret_cast<uint2*>(kp + (size_t)dst * 64 + q * 4) = kv;
#pragma unroll
        for (int j = 0; j < 4; ++j) vp[vfrag_index(dst, q * 4 + j)] = vv[j];
    }
}

__global__ void k_accept_vanilla1(GenDev gp, int B, const int* __restrict__ amax, int* __restrict__ ids, int* __restrict__ L,
                                  int* __restrict__ kvlen, int* __restrict__ finished, int* __restrict__ niter,
                                  long long* __restrict__ hist, int* __restrict__ done)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B || finished[s]) return;
    const int Lcur = L[s], tok = amax[s];
    if (Lcur < gp.Tids) ids[(size_t)s * gp.Tids + Lcur] = tok;
    L[s] = Lcur + 1; kvlen[s] = Lcur; niter[s] += 1;
    atomicAdd(reinterpret_cast<unsigned long long*>(hist + 16), 1ull);
    if (tok == gp.eos || Lcur + 1 >= gp.max_length) {
        finished[s] = 1;
        if (atomicAdd(done + 1, 1) == B - 1) done[0] = 1;
    }
}

// blocks per (stream, head) of the cross-attention: one per 256-key split while that is what fills the chip
// (single stream), fewer — each walking several splits — once streams x heads alone do.  Results do not depend on it.
static inline int xattn_blocks_per_head(int NS, int heads_total)
{
    static const int target = [] { const char* v = std::getenv("WM_XATTN_TARGET_BLOCKS"); return v ? std::atoi(v) : 768; }();
    int gx = (target + heads_total / 2) / heads_total;       // ~768 resident block slots (3 per CU): one round when possible
    gx = std::max(1, std::min(gx, NS));
    int spb = std::min((NS + gx - 1) / gx, WM_XATTN_SPB_MAX);
    return (NS + spb - 1) / spb;
}

// =============================================================================================
// host side
// =============================================================================================
static int dec_layer(wm_ctx* ctx, const DecLayerW& w, int slot, float* h, int b0, int nb, int Mper, const int* base, bool kv_only,
                     const int* sskip = nullptr, const DecLayerW* next = nullptr, const int4* rowinfo = nullptr, const int4* sinfo = nullptr)
{
    hipStream_t st = ctx->stream;
    const int d = ctx->d, H = ctx->H, K32 = d / 32, R = nb * Mper, F32 = ctx->ffn / 32;
    const size_t xpl = (size_t)ctx->Rcap * d, fpl = (size_t)ctx->Rcap * ctx->ffn;
    bf16_t* kc = ctx->kc + ((size_t)slot * ctx->maxB + b0) * H * ctx->Tal * 64;
    bf16_t* vc = ctx->vc + ((size_t)slot * ctx->maxB + b0) * H * ctx->Tal * 64;
    const size_t xoff = ((size_t)slot * ctx->Benc + b0) * H * ctx->Spad * 64;
    // cross_kv_fp8: the decode loop streams the e4m3 copy (one byte per element: half the bytes) + one scale per (stream, head) for K and V
    const bool x8 = ctx->xkv8;
    const bf16_t* kx = x8 ? reinterpret_cast<const bf16_t*>(ctx->kx8 + xoff) : ctx->kx + xoff;
    const bf16_t* vx = x8 ? reinterpret_cast<const bf16_t*>(ctx->vx8 + xoff) : ctx->vx + xoff;
    const float* kxs = x8 ? ctx->kxs + ((size_t)slot * ctx->Benc + b0) * H : nullptr;
    const float* vxs = x8 ? ctx->vxs + ((size_t)slot * ctx->Benc + b0) * H : nullptr;
    const unsigned xkb = x8 ? 64u : 128u;        // bytes of one key row of a head
    // In-launch prefetch (single-tile passes, WM_PREFETCH != 0): launch k carries extra blocks that pull the operand of launch
    // k+1 into L2 (wm_skinny_gemm.h PfJob).  The self-attention launch is tiny, so LN1+QKV fetches for the launch after it.
    const bool pf = ctx->prefetch && R <= 16 && !kv_only;
    const bool f8 = w.qkv_s != nullptr;
    static const int skip_div = [] { const char* v = std::getenv("WM_XATTN_SKIP_DIV"); return v ? std::max(1, std::atoi(v)) : 3; }();
    const int nqt = (Mper + 15) / 16;            // query tiles per stream (a candidate tree of more than 16 nodes; otherwise 1)
    const int nz = nb * nqt;                     // (stream, query tile) pairs = z-blocks of the attention launches
    const int xheads = sskip ? std::max(1, H * nb / skip_div) : H * nz;
    const int xgrid = xattn_blocks_per_head(ctx->NS, xheads);
    // LayerNorm fold (round 6, wm_common.h): the three LayerNorm-fed GEMMs read the operand gamma o h (ctx->xn) and the rows' statistics
    // partials (ctx->lnstats) that the launch producing h wrote (embed / final LayerNorm for LN1 of the first layer, FC2 of the layer below,
    // out-proj for LN2, cross-out for LN3) and apply mean / rstd to their accumulators: no LayerNorm launch, no statistics prologue.
    const bool fold = ctx->ln_fold;
    const FoldIn fin_base{ctx->lnstats, nullptr, d / 16, ctx->Rcap, 1.0f / (float)d};
    auto fold_in = [&](const float* c) { FoldIn f = fin_base; f.c = c; return f; };
    auto res_fold = [&](const float* bias, const float* gnext) {
        EpResidualFold e{h, bias, d, R}; e.gnext = gnext; e.xo = ctx->xn; e.xplane = xpl; e.stats = ctx->lnstats; e.K32 = K32; e.sld = ctx->Rcap;
        return e;
    };
    // 1. LN1 + QKV; k rows / transposed v rows straight into the cache
    if (pf) g_pf_job = pf_for_gemm(w.out_w, f8, d / 16, K32, false);
    TL_SET(slot * 16 + 1 + 8192 * Mper);
    if (fold && rowinfo)
        WM_HIP(launch_skinny_fold(st, WRef{w.qkv_w, w.qkv_s}, 3 * d / 16, K32, R, ctx->xn, xpl, fold_in(w.qkv_c),
                                  EpQKVDecDense{ctx->qbuf, kc, vc, w.qkv_bf, base, Mper, d, H, ctx->Tal, R, rowinfo}));
    else if (fold)
        WM_HIP(launch_skinny_fold(st, WRef{w.qkv_w, w.qkv_s}, 3 * d / 16, K32, R, ctx->xn, xpl, fold_in(w.qkv_c),
                                  EpQKVDec{ctx->qbuf, kc, vc, w.qkv_bf, base, Mper, d, H, ctx->Tal, R}));
    else if (rowinfo)
        WM_HIP(launch_skinny_norm(st, WRef{w.qkv_w, w.qkv_s}, 3 * d / 16, K32, h, w.ln1_w, w.ln1_b, d, R, 1, 0, 1,
                                  EpQKVDecDense{ctx->qbuf, kc, vc, w.qkv_b, base, Mper, d, H, ctx->Tal, R, rowinfo}, ctx->xbuf, xpl));
    else
        WM_HIP(launch_skinny_norm(st, WRef{w.qkv_w, w.qkv_s}, 3 * d / 16, K32, h, w.ln1_w, w.ln1_b, d, R, 1, 0, 1,
                                  EpQKVDec{ctx->qbuf, kc, vc, w.qkv_b, base, Mper, d, H, ctx->Tal, R}, ctx->xbuf, xpl));
    if (kv_only) return WM_OK;
    // fused cross-attention query (FuseQ above): single-tile passes with bf16 weights whose projection plan has a compiled instance
    // OFF by default: measured 14.6 us for the fused launch against 5.6 (LN2 + cross-q) + 2.2 (boundary) + 6.5 (cross-attention)
    // = 14.3 us for the two it replaces (profiles/r02_timeline_fused_cross_q.md): the 6 key-split blocks of a head each pull the
    // head's 160 KB of projection weights through one CU (~47 GB/s) and run LayerNorm -> projection -> attention back to back.
    // Kept (WM_FUSE_CQ=1) because it is bit-identical and the cheapest way to re-measure the trade-off on other shapes.
    static const bool fuse_env = [] { const char* v = std::getenv("WM_FUSE_CQ"); return v && std::atoi(v) != 0; }();
    const SkinnyPlan cqp = skinny_plan(d / 16, K32, true);
    const bool fuse_shape = (cqp.nk == 8 && cqp.ksplit >= 1 && cqp.ksplit <= 5) || (cqp.nk == 4 && (cqp.ksplit == 1 || cqp.ksplit == 3));
    // (not under the merged-step schedule's dense rows: the fused instance reads q at stream * Mper + row)
    // (hi / lo builds only: the fused kernel multiplies bf16 weights by the hi / lo pair itself)
    const bool fuse_cq = WM_ACT_PLANES == 2 && fuse_env && !x8 && !fold && R <= 16 && !f8 && fuse_shape && H * 4 == d / 16 && rowinfo == nullptr;       // single tile: nqt == 1
    const PfJob kvjob = (pf && nqt == 1 && ctx->NS % xgrid == 0 && ctx->Spad == ctx->NS * 256)
        ? PfJob{reinterpret_cast<const char*>(kx), reinterpret_cast<const char*>(vx), (unsigned)(ctx->NS / xgrid) * 256 * xkb,
                (unsigned)(xgrid * H * nb), (unsigned long long)H * nb * ctx->Spad * xkb}
        : PfJob{nullptr, nullptr, 0u, 0u, 0ull};
    // 2. causal self-attention over the contiguous cache (20 blocks: its spare CUs fetch this layer's cross K/V when the
    //    fused cross-attention follows two launches later; otherwise LN2 + cross-q carries that job)
    {
        // batched passes (R > 16): the self-attention launch pulls the out-proj weights towards the chip (Infinity Cache), one job per
        // 64 KB; single-tile passes: LN1 + QKV carried that job (the self-attention launch is tiny there)
        const bool pf_big = ctx->prefetch && R > 16;
        const unsigned long long ow = (unsigned long long)d * d * (f8 ? 1 : 2);
        const PfJob spf = fuse_cq ? kvjob : (pf_big ? PfJob{reinterpret_cast<const char*>(w.out_w), nullptr, 65536u, (unsigned)((ow + 65535) / 65536), ow}
                                                    : PfJob{nullptr, nullptr, 0u, 0u, 0ull});
        const int main_total = H * nz;
        const int zs = spf.n_jobs ? nz + (pf_round8(main_total) - main_total + (int)spf.n_jobs + H - 1) / H : nz;
        TL_SET(slot * 16 + 2 + 8192 * Mper);
        hipLaunchKernelGGL((k_attn_mfma<false, false, NoFuseQ>), dim3(1, H, zs), dim3(256), sizeof(AttnLds<1>), st, kc, vc, ctx->qbuf, sinfo, sskip,
                           Mper | (nz << 8), H | (1 << 8) | (nqt << 16) | (1 << 24), ctx->Tal, 0, g_skinny_done, K32, ctx->xbuf, xpl, nullptr, nullptr, nullptr, spf, NoFuseQ{}, ctx->cur_anc, base, (const float*)nullptr, (const float*)nullptr TL_PASS);
        WM_HIP(hipGetLastError());
    }
    // 3. out_proj + residual
    if (pf) g_pf_job = pf_for_gemm(w.cq_w, f8, d / 16, K32, true);
    TL_SET(slot * 16 + 3 + 8192 * Mper);
    // batched passes with the LayerNorm folded: no LayerNorm launch is left to pull the next GEMM's weights towards the chip, the launch that
    // produces the residual rows carries that job (wm_skinny_gemm.h pf_set_batched); WM_PREFETCH=0 turns it off with the others
    const bool pfb = fold && ctx->prefetch && R > 16;
    const int MTp = (R + 15) / 16;
    if (pfb) pf_set_batched(w.cq_w, nullptr, f8, d / 16, K32, MTp);
    if (fold) WM_HIP(launch_skinny_rows(st, WRef{w.out_w, w.out_s}, d / 16, K32, R, ctx->xbuf, xpl, res_fold(w.out_b, w.ln2_w)));
    else WM_HIP(launch_skinny_rows(st, WRef{w.out_w, w.out_s}, d / 16, K32, R, ctx->xbuf, xpl, EpResidual{h, w.out_b, d, R}));
    // 4. LN2 + cross-attention q (its own launch unless fused into 5.)
    if (!fuse_cq) {
        if (pf) g_pf_job = kvjob;
        TL_SET(slot * 16 + 4 + 8192 * Mper);
        if (fold)
            WM_HIP(launch_skinny_fold(st, WRef{w.cq_w, w.cq_s}, d / 16, K32, R, ctx->xn, xpl, fold_in(w.cq_c), EpF32{ctx->qbuf, w.cq_bf, d, R, 0.125f}));
        else
            WM_HIP(launch_skinny_norm(st, WRef{w.cq_w, w.cq_s}, d / 16, K32, h, w.ln2_w, w.ln2_b, d, R, 1, 0, 1, EpF32{ctx->qbuf, w.cq_b, d, R, 0.125f},
                                      ctx->xbuf, xpl));
    }
    // 5. cross-attention over the encoder K/V, 256 keys per block
    // a base pass with per-stream carry skips the blocks of carrying streams (about half of them at the measured acceptance
    // mix): the key-split grouping is sized for the blocks that actually run
    {
        PfJob xpf{nullptr, nullptr, 0u, 0u, 0ull};
        int zs = nz;
        if (pf || (ctx->prefetch && R > 16)) {
            xpf = pf_for_gemm(w.cout_w, f8, d / 16, K32, false);
            const int main_total = xgrid * H * nz;
            zs = nz + (pf_round8(main_total) - main_total + (int)xpf.n_jobs + xgrid * H - 1) / (xgrid * H);
        }
        TL_SET(slot * 16 + 5 + 8192 * Mper);
        static const bool xattn_nt = [] { const char* v = std::getenv("WM_XATTN_NT"); return v ? std::atoi(v) != 0 : true; }();
#if WM_ACT_PLANES == 2
        if (fuse_cq) {
            const LdNorm ln{h, w.ln2_w, w.ln2_b, d, K32, R, 1, 0};
#define WM_XFUSE(NKv, KSv)                                                                                                    \
            do {                                                                                                              \
                typedef FuseQ<NKv, KSv> FQ;                                                                                   \
                typedef AttnLds<WM_XATTN_SPB_MAX> LdsT;                                                                       \
                const size_t fragb = (size_t)2 * NKv * KSv * 1024;                                                            \
                const size_t lds = (fragb > sizeof(LdsT) ? fragb : ((sizeof(LdsT) + 15) & ~(size_t)15)) + ln.lds_bytes() + 16 * 64 * sizeof(float); \
                auto kern = k_attn_mfma<true, true, FQ>;                                                                      \
                WM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL(kern, dim3(xgrid, H, zs), dim3(64 * (KSv > 4 ? KSv : 4)), lds, st, kx, vx, ctx->qbuf, (const int4*)nullptr, sskip, \
                                   Mper | (nz << 8), H | (ctx->NS << 8) | (nqt << 16) | (xgrid << 24), ctx->Spad, ctx->S, g_skinny_done, K32, ctx->xbuf, xpl, ctx->cml, ctx->co, ctx->ticket, xpf, \
                                   FQ{ln, w.cq_w, w.cq_b}, nullptr, base, (const float*)nullptr, (const float*)nullptr TL_PASS);                                                    \
            } while (0)
            if (cqp.nk == 8 && cqp.ksplit == 5) WM_XFUSE(8, 5);
            else if (cqp.nk == 8 && cqp.ksplit == 4) WM_XFUSE(8, 4);
            else if (cqp.nk == 8 && cqp.ksplit == 3) WM_XFUSE(8, 3);
            else if (cqp.nk == 8 && cqp.ksplit == 2) WM_XFUSE(8, 2);
            else if (cqp.nk == 8 && cqp.ksplit == 1) WM_XFUSE(8, 1);
            else if (cqp.nk == 4 && cqp.ksplit == 3) WM_XFUSE(4, 3);
            else WM_XFUSE(4, 1);
#undef WM_XFUSE
        } else
#endif
        if (x8)
            hipLaunchKernelGGL((k_attn_mfma<true, true, NoFuseQ, true>), dim3(xgrid, H, zs), dim3(256), sizeof(AttnLds<WM_XATTN_SPB_MAX>), st, kx, vx, ctx->qbuf, sinfo, sskip,
                               Mper | (nz << 8), H | (ctx->NS << 8) | (nqt << 16) | (xgrid << 24), ctx->Spad, ctx->S, g_skinny_done, K32, ctx->xbuf, xpl, ctx->cml, ctx->co, ctx->ticket, xpf, NoFuseQ{}, nullptr, base,
                               kxs, vxs TL_PASS);
        else if (xattn_nt)
            hipLaunchKernelGGL((k_attn_mfma<true, true, NoFuseQ>), dim3(xgrid, H, zs), dim3(256), sizeof(AttnLds<WM_XATTN_SPB_MAX>), st, kx, vx, ctx->qbuf, sinfo, sskip,
                               Mper | (nz << 8), H | (ctx->NS << 8) | (nqt << 16) | (xgrid << 24), ctx->Spad, ctx->S, g_skinny_done, K32, ctx->xbuf, xpl, ctx->cml, ctx->co, ctx->ticket, xpf, NoFuseQ{}, nullptr, base, (const float*)nullptr, (const float*)nullptr TL_PASS);
        else
            hipLaunchKernelGGL((k_attn_mfma<true, false, NoFuseQ>), dim3(xgrid, H, zs), dim3(256), sizeof(AttnLds<WM_XATTN_SPB_MAX>), st, kx, vx, ctx->qbuf, sinfo, sskip,
                               Mper | (nz << 8), H | (ctx->NS << 8) | (nqt << 16) | (xgrid << 24), ctx->Spad, ctx->S, g_skinny_done, K32, ctx->xbuf, xpl, ctx->cml, ctx->co, ctx->ticket, xpf, NoFuseQ{}, nullptr, base, (const float*)nullptr, (const float*)nullptr TL_PASS);
        WM_HIP(hipGetLastError());
    }
    // 6. out_proj + residual
    if (pf) g_pf_job = pf_for_gemm(w.fc1_w, f8, ctx->ffn / 16, K32, true);
    TL_SET(slot * 16 + 6 + 8192 * Mper);
    if (pfb) pf_set_batched(w.fc1_w, nullptr, f8, ctx->ffn / 16, K32, MTp);
    if (fold) WM_HIP(launch_skinny_rows(st, WRef{w.cout_w, w.cout_s}, d / 16, K32, R, ctx->xbuf, xpl, res_fold(w.cout_b, w.ln3_w)));
    else WM_HIP(launch_skinny_rows(st, WRef{w.cout_w, w.cout_s}, d / 16, K32, R, ctx->xbuf, xpl, EpResidual{h, w.cout_b, d, R}));
    // 7. LN3 + fc1 + GELU
    if (pf) g_pf_job = pf_for_gemm(w.fc2_w, f8, d / 16, F32, false);
    if (ctx->prefetch && R > 16 && !fold) g_ln_pf_extra = w.fc2_w;     // batched passes: the LayerNorm launch pulls FC1 and FC2 (same size)
    TL_SET(slot * 16 + 7 + 8192 * Mper);
    if (pfb) pf_set_batched(w.fc2_w, nullptr, f8, d / 16, F32, MTp);
    if (fold)
        WM_HIP(launch_skinny_fold(st, WRef{w.fc1_w, w.fc1_s}, ctx->ffn / 16, K32, R, ctx->xn, xpl, fold_in(w.fc1_c),
                                  EpPackedAct<1>{ctx->fbuf, ctx->fbuf + fpl, w.fc1_bf, F32, R}));
    else
        WM_HIP(launch_skinny_norm(st, WRef{w.fc1_w, w.fc1_s}, ctx->ffn / 16, K32, h, w.ln3_w, w.ln3_b, d, R, 1, 0, 1,
                                  EpPackedAct<1>{ctx->fbuf, ctx->fbuf + fpl, w.fc1_b, F32, R}, ctx->xbuf, xpl));
    if (pfb && next) pf_set_batched(next->qkv_w, nullptr, f8, 3 * d / 16, K32, MTp);
    // 8. fc2 + residual (fold: + the operand of the NEXT layer's LN1 + QKV; the last layer's rows go to the final LayerNorm launch instead)
    if (pf && next) g_pf_job = pf_for_gemm(next->qkv_w, f8, 3 * d / 16, K32, true);
    TL_SET(slot * 16 + 8 + 8192 * Mper);
    if (fold && next) WM_HIP(launch_skinny_rows(st, WRef{w.fc2_w, w.fc2_s}, d / 16, F32, R, ctx->fbuf, fpl, res_fold(w.fc2_b, next->ln1_w)));
    else WM_HIP(launch_skinny_rows(st, WRef{w.fc2_w, w.fc2_s}, d / 16, F32, R, ctx->fbuf, fpl, EpResidual{h, w.fc2_b, d, R}));
    return WM_OK;
}

// c[n] = sum_k W[n][k] gamma[k], bf[n] = b[n] + sum_k W[n][k] beta[k] of one LayerNorm-fed matrix in the packed layout (bf16, or fp8 e4m3 with one
// scale per row): one wave per output row, fp64 sums (once per context, wm_create).  (wm_common.h "LayerNorm folded into the GEMM it feeds")
__global__ void k_fold_vectors(const bf16_t* __restrict__ W, const float* __restrict__ wscale, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ bias, int N, int K, float* __restrict__ c, float* __restrict__ bf)
{
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    double sc = 0.0, sb = 0.0;
    for (int k = lane; k < K; k += 64) {
        const size_t i = packed_index(n, k, K >> 5);
        float wv;
        if (wscale) {
            const unsigned char b8 = reinterpret_cast<const unsigned char*>(W)[i];
            wv = __builtin_amdgcn_cvt_f32_fp8((int)b8, 0) * wscale[n];
        } else wv = w16_to_f32(W[i]);        // bf16, or fp16 in an f16 build
        sc += (double)wv * (double)gamma[k]; sb += (double)wv * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sc += __shfl_xor(sc, o, 64); sb += __shfl_xor(sb, o, 64); }
    if (lane == 0) { c[n] = (float)sc; bf[n] = (float)((double)bias[n] + sb); }
}

int wm_dec_fold_init(wm_ctx* ctx)
{
    hipStream_t st = ctx->stream;
    const int d = ctx->d, ffn = ctx->ffn;
    float* p = ctx->foldv;
    for (auto& w : ctx->dec) {
        float* v[6];
        const int len[6] = {3 * d, 3 * d, d, d, ffn, ffn};
        for (int i = 0; i < 6; ++i) { v[i] = p; p += len[i]; }
        hipLaunchKernelGGL(k_fold_vectors, dim3((3 * d + 3) / 4), dim3(256), 0, st, w.qkv_w, w.qkv_s, w.ln1_w, w.ln1_b, w.qkv_b, 3 * d, d, v[0], v[1]);
        hipLaunchKernelGGL(k_fold_vectors, dim3((d + 3) / 4), dim3(256), 0, st, w.cq_w, w.cq_s, w.ln2_w, w.ln2_b, w.cq_b, d, d, v[2], v[3]);
        hipLaunchKernelGGL(k_fold_vectors, dim3((ffn + 3) / 4), dim3(256), 0, st, w.fc1_w, w.fc1_s, w.ln3_w, w.ln3_b, w.fc1_b, ffn, d, v[4], v[5]);
        WM_HIP(hipGetLastError());
        w.qkv_c = v[0]; w.qkv_bf = v[1]; w.cq_c = v[2]; w.cq_bf = v[3]; w.fc1_c = v[4]; w.fc1_bf = v[5];
    }
    return WM_OK;
}

// ---- stage 1: embed + the L decoder layers for streams [b0, b0+nb), Mper tokens each -> ctx->h --------
// mode 0 = base pass (tokens ids[kvlen..], positions kvlen..), mode 1 = verify pass (tokens cand[0..Mper),
// positions L..).
int wm_dec_stage_layers(wm_ctx* ctx, int b0, int nb, int Mper, int mode)
{
    hipStream_t st = ctx->stream;
    g_skinny_done = ctx->use_done ? ctx->done : nullptr;
    const int d = ctx->d, R = nb * Mper;
    // mode 2 = merged step: per stream either its verify rows (as mode 1) or its one base row (k_step_begin: rowinfo / sinfo)
    const int* base = (mode == 0 ? ctx->kvlen : ctx->L) + b0;          // mode 2: positions come from rowinfo / sinfo
    if (R > ctx->Rcap || Mper > ctx->Mmax) { ctx->err = "decode pass exceeds the row capacity of the context"; return WM_ERR_ARG; }
    // LayerNorm fold: the embed launch also writes layer 0's QKV operand (gamma_ln1 o h) and the rows' statistics partials
    const float* eg = ctx->ln_fold ? ctx->dec[0].ln1_w : nullptr;
    const size_t expl = (size_t)ctx->Rcap * d;
    // node tables of a verify pass: the candidate tree, or the chain + sibling leaves of a single-stream pass that carries them (Mper > K + 1)
    const TreeDev* ptree = ctx->tn ? ctx->tree : ((mode == 1 && ctx->gp.sib > 0 && Mper == ctx->K + 1 + ctx->gp.sib) ? ctx->sibtree : nullptr);
    if (mode == 2)
        hipLaunchKernelGGL(k_embed, dim3(R), dim3(256), 0, st, ctx->h, ctx->tok_emb, ctx->dec_pos, base,
                           ctx->cand + (size_t)b0 * WM_CAND_STRIDE, WM_CAND_STRIDE, 0, Mper, d, ctx->V, ctx->Tmax, (const int*)nullptr,
                           ctx->rowinfo, ctx->ids, ctx->gp.Tids, eg, ctx->xn, expl, ctx->lnstats, ctx->Rcap);
    else if (mode == 0)
        hipLaunchKernelGGL(k_embed, dim3(R), dim3(256), 0, st, ctx->h, ctx->tok_emb, ctx->dec_pos, base,
                           ctx->ids + (size_t)b0 * ctx->gp.Tids, ctx->gp.Tids, 1, Mper, d, ctx->V, ctx->Tmax, (const int*)nullptr,
                           (const int4*)nullptr, (const int*)nullptr, 0, eg, ctx->xn, expl, ctx->lnstats, ctx->Rcap);
    else
        hipLaunchKernelGGL(k_embed, dim3(R), dim3(256), 0, st, ctx->h, ctx->tok_emb, ctx->dec_pos, base,
                           ctx->cand + (size_t)b0 * WM_CAND_STRIDE, WM_CAND_STRIDE, 0, Mper, d, ctx->V, ctx->Tmax, ptree ? ptree->depth : nullptr,
                           (const int4*)nullptr, (const int*)nullptr, 0, eg, ctx->xn, expl, ctx->lnstats, ctx->Rcap);
    WM_HIP(hipGetLastError());
    ctx->cur_anc = (mode == 1 && ptree) ? ptree->anc : nullptr;
    // batched hidden-state carry: a stream whose previous verify pass accepted a > 0 candidates already has the state
    // of its base token (k_accept saved it, k_rows_norm below picks it up); its rows still ride through the GEMMs
    // (the weights are streamed once for everybody) but its attention blocks exit, saving their K/V reads
    const int* sskip = (mode == 0 && Mper == 1 && ctx->dev_carry) ? ctx->carry + b0 : nullptr;
    for (int l = 0; l < ctx->cfg.dec_layers; ++l) {
        int rc = dec_layer(ctx, ctx->dec[l], l, ctx->h, b0, nb, Mper, base, false, sskip, l + 1 < ctx->cfg.dec_layers ? &ctx->dec[l + 1] : nullptr,
                           mode == 2 ? ctx->rowinfo : nullptr, mode == 2 ? ctx->sinfo : nullptr);
        if (rc) return rc;
    }
    return WM_OK;
}

// ---- stage 2: final LayerNorm for all rows (-> hf); Medusa-Block: the extra decoder layer on the
// post-LN state with its own KV slot (model.py:1382-1417) — complete when its output feeds the heads,
// K/V side effect only when medusa is disabled (verify pass, model.py:1410-1413) -----------------------
int wm_dec_stage_final(wm_ctx* ctx, int b0, int nb, int Mper, int mode, int medusa)
{
    hipStream_t st = ctx->stream;
    const int d = ctx->d, K32 = d / 32, R = nb * Mper;
    const int* base = (mode == 0 ? ctx->kvlen : ctx->L) + b0;          // mode 2: positions come from rowinfo / sinfo
    float* hf = ctx->hf + (size_t)b0 * Mper * d;          // rows of this chunk; persists until the stream's next pass
    // (Medusa-Block + LayerNorm fold: the rows copied to hblk are the extra layer's LN1 input — its QKV operand and statistics partials go out here)
    const bool bfold = ctx->block && ctx->ln_fold && !ctx->gp.vanilla;
    hipLaunchKernelGGL(k_rows_norm, dim3((R + 3) / 4), dim3(256), 0, st, ctx->h, 1, 0, ctx->dec_lnf_w, ctx->dec_lnf_b, 1,
                       hf, ctx->block ? ctx->hblk : nullptr, nullptr, (size_t)0, K32, 1, 0, d, R,
                       (mode == 0 && Mper == 1 && ctx->dev_carry) ? ctx->carry + b0 : nullptr, ctx->hf_keep + (size_t)b0 * d,
                       bfold ? ctx->dec[ctx->nkv - 1].ln1_w : nullptr, ctx->xn, (size_t)ctx->Rcap * d, ctx->lnstats, ctx->Rcap);
    WM_HIP(hipGetLastError());
    ctx->hf_cur = hf;
    if (ctx->block && !ctx->gp.vanilla) {
        // verify pass (medusa disabled): the reference needs only the layer's K/V (model.py:1410-1413); with the hidden-state
        // carry on, the layer runs completely so that row a's output can serve the next iteration's heads
        const bool carrying = ctx->host_carry || ctx->dev_carry;
        const bool kv_only = !medusa && !carrying;
        const int* sskip = (mode == 0 && Mper == 1 && ctx->dev_carry) ? ctx->carry + b0 : nullptr;
        int rc = dec_layer(ctx, ctx->dec[ctx->nkv - 1], ctx->nkv - 1, ctx->hblk, b0, nb, Mper, base, kv_only, sskip, nullptr,
                           mode == 2 ? ctx->rowinfo : nullptr, mode == 2 ? ctx->sinfo : nullptr);
        if (rc) return rc;
        if (sskip) {
            hipLaunchKernelGGL(k_rows_take_carried, dim3(R), dim3(256), 0, st, ctx->hblk + (size_t)b0 * d, ctx->hb_keep + (size_t)b0 * d,
                               sskip, d, R, g_skinny_done);
            WM_HIP(hipGetLastError());
        }
    }
    return WM_OK;
}

// ---- stage 3: Medusa heads + shared vocabulary projection for nsel selected rows (row m of the
// selection = pass row m*sel_mul + sel_off) -> ctx->logits rows [m][head] -------------------------------
int wm_dec_stage_heads(wm_ctx* ctx, int nsel, int sel_mul, int sel_off, int medusa)
{
    hipStream_t st = ctx->stream;
    TL_SET(1000);
    const int d = ctx->d, K32 = d / 32, K = ctx->K;
    const int nout = medusa ? K + 1 : 1;
    const size_t ypl = (size_t)ctx->Rcap * d;
    if (nsel * nout > ctx->Rcap) { ctx->err = "head stage exceeds the row capacity of the context"; return WM_ERR_ARG; }
    if (!ctx->block) {
        // Medusa-Linear: every head (incl. base head 0) = x + SiLU(W_k x + b_k), then proj_out (model.py:1274-1284)
        WM_HIP(launch_skinny_norm(st, ctx->heads_w, nout * d / 16, K32, ctx->hf_cur, nullptr, nullptr, d, nsel, sel_mul, sel_off, 0,
                                  EpHead{ctx->ybuf, ctx->ybuf + ypl, ctx->hf_cur, ctx->heads_b, d, K32, nout, 0, nsel, sel_mul, sel_off},
                                  ctx->xbuf, ypl));
    } else {
        // Medusa-Block: base logits = proj_out(hf) (model.py:1287); K heads on the block output (model.py:1414-1417)
        hipLaunchKernelGGL(k_rows_norm, dim3((nsel + 3) / 4), dim3(256), 0, st, ctx->hf_cur, sel_mul, sel_off, nullptr, nullptr, 0,
                           nullptr, nullptr, ctx->ybuf, ypl, K32, nout, 0, d, nsel, nullptr, nullptr);
        WM_HIP(hipGetLastError());
        if (medusa)
            WM_HIP(launch_skinny_norm(st, ctx->heads_w, K * d / 16, K32, ctx->hblk, nullptr, nullptr, d, nsel, sel_mul, sel_off, 0,
                                      EpHead{ctx->ybuf, ctx->ybuf + ypl, ctx->hblk, ctx->heads_b, d, K32, nout, 1, nsel, sel_mul, sel_off},
                                      ctx->xbuf, ypl));
    }
    // shared vocabulary projection (tied proj_out, model.py:1277)
    TL_SET(1001);
    WM_HIP(launch_skinny_rows(st, ctx->vocab_w, ctx->Vpad / 16, K32, nsel * nout, ctx->ybuf, ypl,
                              EpLogits{ctx->logits, nullptr, ctx->Vpad, nsel * nout, 1.0f}));
    return WM_OK;
}

int wm_dec_pass(wm_ctx* ctx, int b0, int nb, int Mper, int mode, int medusa, int all_rows)
{
    int rc = wm_dec_stage_layers(ctx, b0, nb, Mper, mode);
    if (rc) return rc;
    rc = wm_dec_stage_final(ctx, b0, nb, Mper, mode, medusa);
    if (rc) return rc;
    if (all_rows) return wm_dec_stage_heads(ctx, nb * Mper, 1, 0, medusa);
    return wm_dec_stage_heads(ctx, nb, Mper, Mper - 1, medusa);
}

__global__ void k_advance_kvlen(int* __restrict__ kvlen, int n, int B)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < B) kvlen[s] += n;
}

// Long prompts (`prompt_ids` conditioning, model.py:1519-1529 -> HF _prepare_decoder_input_ids): all but the last <= 16 prompt
// tokens go through the layers in 16-row chunks for their K/V only (Medusa-Block: through the extra layer as well); the
// first iteration proper then starts from the last chunk.  Returns the number of prompt tokens left for it.
int wm_dec_prompt_prefix(wm_ctx* ctx, int P)
{
    hipStream_t st = ctx->stream;
    const int B = ctx->Bdec;
    int left = P;
    while (left > 16) {
        const int n = std::min(16, left - 1);                 // keep >= 1 token for the iteration that produces logits
        int rc = wm_dec_stage_layers(ctx, 0, B, n, 0);
        if (rc) return -1;
        if (ctx->block) { rc = wm_dec_stage_final(ctx, 0, B, n, 0, 1); if (rc) return -1; }
        hipLaunchKernelGGL(k_advance_kvlen, dim3((B + 63) / 64), dim3(64), 0, st, ctx->kvlen, n, B);
        if (hipGetLastError() != hipSuccess) { ctx->err = "k_advance_kvlen launch failed"; return -1; }
        left -= n;
    }
    return left;
}

// One full Medusa iteration (or one vanilla step) over all Bdec streams, chunked to <= 16 token rows.
int wm_dec_iteration(wm_ctx* ctx, int Mper_base)
{
    hipStream_t st = ctx->stream;
    const int B = ctx->Bdec, K = ctx->K, rps = K + 1;
    const GenDev gp = ctx->gp;
    if (gp.vanilla) {
        const int chunk = B;
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = min(chunk, B - b0);
            int rc = wm_dec_pass(ctx, b0, nb, Mper_base, 0, 0, 0);
            if (rc) return rc;
            hipLaunchKernelGGL(k_select1<false>, dim3(SEL_SP, nb), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen,
                               ctx->L + b0, 1, ctx->part1);
            WM_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_select_argmax, dim3((nb + 63) / 64), dim3(64), 0, st, ctx->part1, nb, b0, ctx->amax);
            WM_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(k_accept_vanilla1, dim3((B + 63) / 64), dim3(64), 0, st, gp, B, ctx->amax, ctx->ids, ctx->L,
                           ctx->kvlen, ctx->finished, ctx->niter, ctx->hist, ctx->done);
        WM_HIP(hipGetLastError());
        return WM_OK;
    }
    int rc = wm_dec_iter_base(ctx, Mper_base);
    if (rc) return rc;
    return wm_dec_iter_rest(ctx, Mper_base);
}

// (a1) base pass layers + final LayerNorm -> hf rows (skipped by the host when the hidden state was carried)
int wm_dec_iter_base(wm_ctx* ctx, int Mper_base)
{
    int rc = wm_dec_stage_layers(ctx, 0, ctx->Bdec, Mper_base, 0);
    if (rc) return rc;
    return wm_dec_stage_final(ctx, 0, ctx->Bdec, Mper_base, 0, 1);
}

// (a2) heads + candidates, (d) verify pass + posterior statistics, (f)-(j) accept
int wm_dec_iter_rest(wm_ctx* ctx, int Mper_base)
{
    hipStream_t st = ctx->stream;
    const int B = ctx->Bdec, K = ctx->K, rps = K + 1, nb = B;
    const GenDev gp = ctx->gp;
    const bool carry = ctx->host_carry;
    ctx->hf_cur = ctx->hf;
    g_skinny_done = ctx->use_done ? ctx->done : nullptr;
    int rc = wm_dec_stage_heads(ctx, nb, Mper_base, Mper_base - 1, 1);
    if (rc) return rc;
    const int vr = ctx->tn ? ctx->tn : rps;                  // rows per stream of the verify pass: tree nodes / chain candidates
    if (ctx->tn) {
        hipLaunchKernelGGL(k_tree_cand, dim3(nb * rps), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L, ctx->tree,
                           ctx->cand, g_skinny_done);
        WM_HIP(hipGetLastError());
    } else {
        // slice partials of every head's row (head 1's slices also keep their six best when the verify pass carries sibling rows), then ONE
        // block per stream turns them into the chain's candidates (+ the sibling tokens)
        hipLaunchKernelGGL(k_select1<false>, dim3(SEL_SP, nb * rps), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L, rps, ctx->part1,
                           (const int4*)nullptr, gp.sib > 0 ? ctx->sibpart : nullptr);
        WM_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_cand_fin, dim3(nb), dim3(256), 0, st, ctx->part1, gp, ctx->amax, ctx->cand, gp.sib > 0 ? ctx->sibpart : nullptr);
        WM_HIP(hipGetLastError());
    }
    // (d) verify pass over the candidates (chain: positions L..L+K; tree: node n at L + depth(n), ancestor-masked), then
    //     posterior statistics of every row
    if (gp.sib > 0 && !ctx->tn) {
        // the layers and the final LayerNorm (Medusa-Block: the extra layer) run over K + 1 + S rows; logits only for the chain's K + 1 (rows 0 .. K of the
        // one stream): the sibling rows are wanted for their hidden state and their K/V rows
        rc = wm_dec_stage_layers(ctx, 0, nb, rps + gp.sib, 1);
        if (rc) return rc;
        rc = wm_dec_stage_final(ctx, 0, nb, rps + gp.sib, 1, 0);
        if (rc) return rc;
        rc = wm_dec_stage_heads(ctx, rps, 1, 0, 0);
    } else
        rc = wm_dec_pass(ctx, 0, nb, vr, 1, 0, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_select1<true>, dim3(SEL_SP, nb * vr), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L, vr, ctx->part1);
    WM_HIP(hipGetLastError());
    if (gp.accept_mode == WM_ACCEPT_TYPICAL)
        hipLaunchKernelGGL(k_select2, dim3(SEL_SP, nb * vr), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L,
                           ctx->cand, vr, 0, ctx->part1, ctx->part2, ctx->amax, ctx->pc, ctx->tn ? ctx->tree : nullptr);
    else
        hipLaunchKernelGGL(k_select_argmax, dim3((nb * vr + 63) / 64), dim3(64), 0, st, ctx->part1, nb * vr, 0, ctx->amax);
    WM_HIP(hipGetLastError());
    // (f)-(j) accept / emit / compact / stop.  With host_carry the carried post-LN row goes straight to hf row 0
    // (where the skipped base pass would have put it) and the carry / finished flags to host-mapped memory; with
    // dev_carry (several streams) it goes to hf_keep[stream] and the next base pass's final LayerNorm selects it.
    if (ctx->tn) {
        hipLaunchKernelGGL(k_accept_tree, dim3(B), dim3(64), 0, st, gp, ctx->tree, ctx->cand, ctx->amax, ctx->pc, ctx->part2, ctx->ids, ctx->L,
                           ctx->kvlen, ctx->finished, ctx->niter, ctx->hist, ctx->done, B, (carry || ctx->dev_carry) ? ctx->carry : nullptr, ctx->hf,
                           carry ? ctx->hf : ctx->hf_keep, ctx->d, carry ? ctx->hostflags_dev : nullptr,
                           ctx->block ? ctx->hblk : nullptr, carry ? ctx->hblk : ctx->hb_keep, ctx->sel_src, ctx->sel_n, ctx->sel_base);
        WM_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_kv_compact, dim3(ctx->H, B, ctx->nkv), dim3(256), 0, st, ctx->kc, ctx->vc, ctx->sel_src, ctx->sel_n, ctx->sel_base,
                           ctx->H, ctx->Tal, ctx->maxB);
        WM_HIP(hipGetLastError());
        return WM_OK;
    }
    hipLaunchKernelGGL(k_accept, dim3(B), dim3(64), 0, st, gp, ctx->cand, ctx->amax, ctx->pc, ctx->part2, ctx->ids, ctx->L,
                       ctx->kvlen, ctx->finished, ctx->niter, ctx->hist, ctx->done, B, (carry || ctx->dev_carry) ? ctx->carry : nullptr, ctx->hf,
                       carry ? ctx->hf : ctx->hf_keep, ctx->d, carry ? ctx->hostflags_dev : nullptr,
                       ctx->block ? ctx->hblk : nullptr, carry ? ctx->hblk : ctx->hb_keep, (const int4*)nullptr,
                       gp.sib > 0 ? ctx->sel_src : nullptr, gp.sib > 0 ? ctx->sel_n : nullptr, gp.sib > 0 ? ctx->sel_base : nullptr);
    WM_HIP(hipGetLastError());
    if (gp.sib > 0) {           // a sibling hit: its K / V rows go from provisional row K + 1 + j to row 1 of every self-KV slot (no-op otherwise: sel_n = 0)
        hipLaunchKernelGGL(k_kv_compact, dim3(ctx->H, B, ctx->nkv), dim3(256), 0, st, ctx->kc, ctx->vc, ctx->sel_src, ctx->sel_n, ctx->sel_base,
                           ctx->H, ctx->Tal, ctx->maxB);
        WM_HIP(hipGetLastError());
    }
    return WM_OK;
}

// One STEP of the merged-step schedule (several streams, chain candidates, per-stream hidden-state carry; wm_engine.hip picks it).
// The lock-step iteration above runs a base pass for every stream (the rows of carrying streams ride along) and then the verify pass:
// two weight streams and two launch chains per iteration although only the streams that accepted nothing need the base pass.  Here
// every stream contributes to ONE pass per step what it needs next: its K + 1 candidate rows (state carried: heads -> candidates ->
// verify -> accept, exactly the iteration's second half) or its single base row (then its post-LN state is kept and it verifies in the
// next step).  Per stream the arithmetic, the order of its passes and therefore its tokens are those of the lock-step schedule (and of a
// single-stream run); only which other streams share a launch changes.  Reference loop: model.py:634-793 (one stream).
int wm_dec_step(wm_ctx* ctx, int)
{
    hipStream_t st = ctx->stream;
    const int B = ctx->Bdec, K = ctx->K, rps = K + 1;
    const GenDev gp = ctx->gp;
    g_skinny_done = ctx->use_done ? ctx->done : nullptr;
    g_skinny_ntiles = nullptr;
    if (B > 1024) { ctx->err = "merged-step schedule: more than 1024 streams"; return WM_ERR_ARG; }
    hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(256), 0, st, ctx->carry, ctx->L, ctx->kvlen, ctx->finished, ctx->rowinfo, ctx->sinfo, ctx->steprows, rps, B, B * rps);
    WM_HIP(hipGetLastError());
    // (a2) heads + candidates from the kept states (streams in base mode: computed and ignored)
    ctx->hf_cur = ctx->hf_keep;
    float* hblk_rows = ctx->hblk;
    if (ctx->block) ctx->hblk = ctx->hb_keep;                 // Medusa-Block: the heads read the kept block-layer outputs
    int rc = wm_dec_stage_heads(ctx, B, 1, 0, 1);
    ctx->hblk = hblk_rows;
    if (rc) return rc;
    hipLaunchKernelGGL(k_select1<false>, dim3(SEL_SP, B * rps), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L, rps, ctx->part1, (const int4*)nullptr, (float2*)nullptr);
    WM_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_cand_fin, dim3(B), dim3(256), 0, st, ctx->part1, gp, ctx->amax, ctx->cand, (const float2*)nullptr);       // arg-max of every row -> cand, one launch
    WM_HIP(hipGetLastError());
    // (d) ONE pass over the dense rows: verify rows and base rows; the launches are sized for B * rps rows, token tiles beyond the step's
    //     rows exit at once (g_skinny_ntiles)
    g_skinny_ntiles = ctx->steprows + 1;
    rc = wm_dec_stage_layers(ctx, 0, B, rps, 2);
    if (rc == WM_OK) rc = wm_dec_stage_final(ctx, 0, B, rps, 2, 0);
    if (rc == WM_OK) rc = wm_dec_stage_heads(ctx, B * rps, 1, 0, 0);
    g_skinny_ntiles = nullptr;
    if (rc) return rc;
    hipLaunchKernelGGL(k_select1<true>, dim3(SEL_SP, B * rps), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L, rps, ctx->part1, (const int4*)ctx->rowinfo, (float2*)nullptr);
    WM_HIP(hipGetLastError());
    if (gp.accept_mode == WM_ACCEPT_TYPICAL)
        hipLaunchKernelGGL(k_select2, dim3(SEL_SP, B * rps), dim3(256), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L,
                           ctx->cand, rps, 0, ctx->part1, ctx->part2, ctx->amax, ctx->pc, (const TreeDev*)nullptr, (const int4*)ctx->rowinfo);
    else
        hipLaunchKernelGGL(k_select_argmax, dim3((B * rps + 63) / 64), dim3(64), 0, st, ctx->part1, B * rps, 0, ctx->amax);
    WM_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_accept, dim3(B), dim3(64), 0, st, gp, ctx->cand, ctx->amax, ctx->pc, ctx->part2, ctx->ids, ctx->L,
                       ctx->kvlen, ctx->finished, ctx->niter, ctx->hist, ctx->done, B, ctx->carry, ctx->hf, ctx->hf_keep, ctx->d,
                       (int*)nullptr, ctx->block ? ctx->hblk : nullptr, ctx->hb_keep, (const int4*)ctx->sinfo);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

// Times decode-path GEMMs of decoder layer 0 for `rows` token rows (bench.py roofline leg; tests/microbench).
// kernel 0: the six weight-streaming GEMMs of a layer back to back (LayerNorm-fused / LN launch + token-tile GEMM as the pass
// would run them); 1..6: one of them alone (1 LN1+QKV, 2 out-proj, 3 LN2+cross-q, 4 cross-out, 5 LN3+FC1+GELU, 6 FC2);
// 7: the shared vocabulary projection.  *bytes = the weight bytes the timed launches stream.
int wm_dec_profile(wm_ctx* ctx, int kernel, int rows, int reps, float* ms, double* bytes)
{
    if (kernel < 0 || kernel > 7 || rows < 1 || rows > ctx->Rcap) { ctx->err = "wm_profile_kernel: bad arguments"; return WM_ERR_ARG; }
    g_skinny_done = nullptr;
    hipStream_t st = ctx->stream;
    const int d = ctx->d, K32 = d / 32, R = rows, H = ctx->H, F32 = ctx->ffn / 32;
    const size_t xpl = (size_t)ctx->Rcap * d, fpl = (size_t)ctx->Rcap * ctx->ffn;
    const DecLayerW& w = ctx->dec[0];
    WM_HIP(hipMemsetAsync(ctx->h, 0, (size_t)ctx->Rcap * d * sizeof(float), st));
    WM_HIP(hipMemsetAsync(ctx->kvlen, 0, sizeof(int) * ctx->maxB, st));
    const bool all = kernel == 0;
    // the QKV epilogue scatters K / V rows into the self-attention cache: the rows go in as the verify pass would put them, Rcap / maxB
    // rows per stream from position 0 (<= 64 <= Tal; all of them on one stream would run past a head's slab once rows > Tal)
    const int mper = ctx->Rcap / ctx->maxB;
    const bool fold = ctx->ln_fold;       // the launches as dec_layer issues them (LayerNorm folded: wm_common.h)
    const FoldIn fin_base{ctx->lnstats, nullptr, d / 16, ctx->Rcap, 1.0f / (float)d};
    auto fold_in = [&](const float* c) { FoldIn f = fin_base; f.c = c; return f; };
    auto res_fold = [&](const float* bias, const float* gnext) {
        EpResidualFold e{ctx->h, bias, d, R}; e.gnext = gnext; e.xo = ctx->xn; e.xplane = xpl; e.stats = ctx->lnstats; e.K32 = K32; e.sld = ctx->Rcap;
        return e;
    };
    auto body = [&]() -> int {
        if ((all || kernel == 1) && fold)
            WM_HIP(launch_skinny_fold(st, WRef{w.qkv_w, w.qkv_s}, 3 * d / 16, K32, R, ctx->xn, xpl, fold_in(w.qkv_c),
                                      EpQKVDec{ctx->qbuf, ctx->kc, ctx->vc, w.qkv_bf, ctx->kvlen, mper, d, H, ctx->Tal, R}));
        else if (all || kernel == 1)
            WM_HIP(launch_skinny_norm(st, WRef{w.qkv_w, w.qkv_s}, 3 * d / 16, K32, ctx->h, w.ln1_w, w.ln1_b, d, R, 1, 0, 1,
                                      EpQKVDec{ctx->qbuf, ctx->kc, ctx->vc, w.qkv_b, ctx->kvlen, mper, d, H, ctx->Tal, R}, ctx->xbuf, xpl));
        if ((all || kernel == 2) && fold)
            WM_HIP(launch_skinny_rows(st, WRef{w.out_w, w.out_s}, d / 16, K32, R, ctx->xbuf, xpl, res_fold(w.out_b, w.ln2_w)));
        else if (all || kernel == 2)
            WM_HIP(launch_skinny_rows(st, WRef{w.out_w, w.out_s}, d / 16, K32, R, ctx->xbuf, xpl, EpResidual{ctx->h, w.out_b, d, R}));
        if ((all || kernel == 3) && fold)
            WM_HIP(launch_skinny_fold(st, WRef{w.cq_w, w.cq_s}, d / 16, K32, R, ctx->xn, xpl, fold_in(w.cq_c), EpF32{ctx->qbuf, w.cq_bf, d, R, 0.125f}));
        else if (all || kernel == 3)
            WM_HIP(launch_skinny_norm(st, WRef{w.cq_w, w.cq_s}, d / 16, K32, ctx->h, w.ln2_w, w.ln2_b, d, R, 1, 0, 1, EpF32{ctx->qbuf, w.cq_b, d, R, 0.125f},
                                      ctx->xbuf, xpl));
        if ((all || kernel == 4) && fold)
            WM_HIP(launch_skinny_rows(st, WRef{w.cout_w, w.cout_s}, d / 16, K32, R, ctx->xbuf, xpl, res_fold(w.cout_b, w.ln3_w)));
        else if (all || kernel == 4)
            WM_HIP(launch_skinny_rows(st, WRef{w.cout_w, w.cout_s}, d / 16, K32, R, ctx->xbuf, xpl, EpResidual{ctx->h, w.cout_b, d, R}));
        if ((all || kernel == 5) && fold)
            WM_HIP(launch_skinny_fold(st, WRef{w.fc1_w, w.fc1_s}, ctx->ffn / 16, K32, R, ctx->xn, xpl, fold_in(w.fc1_c),
                                      EpPackedAct<1>{ctx->fbuf, ctx->fbuf + fpl, w.fc1_bf, F32, R}));
        else if (all || kernel == 5)
            WM_HIP(launch_skinny_norm(st, WRef{w.fc1_w, w.fc1_s}, ctx->ffn / 16, K32, ctx->h, w.ln3_w, w.ln3_b, d, R, 1, 0, 1,
                                      EpPackedAct<1>{ctx->fbuf, ctx->fbuf + fpl, w.fc1_b, F32, R}, ctx->xbuf, xpl));
        if ((all || kernel == 6) && fold)
            WM_HIP(launch_skinny_rows(st, WRef{w.fc2_w, w.fc2_s}, d / 16, F32, R, ctx->fbuf, fpl, res_fold(w.fc2_b, w.ln1_w)));
        else if (all || kernel == 6)
            WM_HIP(launch_skinny_rows(st, WRef{w.fc2_w, w.fc2_s}, d / 16, F32, R, ctx->fbuf, fpl, EpResidual{ctx->h, w.fc2_b, d, R}));
        if (kernel == 7)
            WM_HIP(launch_skinny_rows(st, ctx->vocab_w, ctx->Vpad / 16, K32, R, ctx->ybuf, xpl, EpLogits{ctx->logits, nullptr, ctx->Vpad, R, 1.0f}));
        return WM_OK;
    };
    int rc = body();
    if (rc) return rc;
    WM_HIP(hipEventRecord(ctx->ev0, st));
    for (int i = 0; i < reps; ++i) { rc = body(); if (rc) return rc; }
    WM_HIP(hipEventRecord(ctx->ev1, st));
    WM_HIP(hipEventSynchronize(ctx->ev1));
    float t = 0.f;
    WM_HIP(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
    *ms = t / reps;
    const double dd = (double)d * d, df = (double)d * ctx->ffn;
    const double wb[8] = {2.0 * (6 * dd + 2 * df), 6 * dd, 2 * dd, 2 * dd, 2 * dd, 2 * df, 2 * df, 2.0 * ctx->Vpad * d};
    *bytes = wb[kernel];
    return WM_OK;
}

#ifdef WM_TIMELINE
// debug build only (libwm_tl.so): point the in-kernel timeline probes at a record buffer (DEV) and its append index (DEV)
extern "C" int wm_debug_timeline(wm_ctx* ctx, void* buf, unsigned* idx, unsigned cap)
{
    WM_HIP(hipSetDevice(ctx->device));
    TlRec* b = reinterpret_cast<TlRec*>(buf);
    WM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &b, sizeof(b)));
    WM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tl_idx), &idx, sizeof(idx)));
    WM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tl_cap), &cap, sizeof(cap)));
    return WM_OK;
}
#endif
