// wm_common.h — shared device helpers for the gfx950 Whisper-Medusa engine.
//
// "Packed" layout (used for EVERY bf16 GEMM operand, weights and activations alike):
// a row-major [R][K] matrix is stored as [R/16][K/32][64 lanes][8 elems]; lane l of a
// wavefront holds row (l & 15), columns 8*(l >> 4) .. +8 of the 16x32 tile — exactly the
// A (and, transposed, B) operand fragment of v_mfma_f32_16x16x32_bf16.  One tile = 1 KiB
// contiguous, so a wave fetches a fragment with one fully coalesced 16-B-per-lane load and
// an LDS image of it is read back conflict-free with ds_read_b128 (lane-linear).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                       // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;         // one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;           // one MFMA C/D fragment

#define WM_HEAD_DIM 64
#define WM_MAX_ROWS_SKINNY 16      // token rows (streams x tokens) the weight-streaming GEMM handles per launch
#define WM_TREE_MAX_NODES 64       // candidate-tree nodes per stream: the verify pass of a stream is up to four 16-row query tiles
#define WM_TREE_MAX_PATHS 32
#define WM_CAND_STRIDE 64          // ints per stream in the candidate-token buffer (chain: K + 1 <= 16 used)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// round-to-nearest-even (== torch) on the gfx950 converter: one v_cvt_pk_bf16_f32 per pair
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// element index of (row, k) inside a packed [R][K] matrix with K32 = K/32 k-tiles
__device__ __host__ __forceinline__ size_t packed_index(int row, int k, int K32) {
    return ((size_t)(row >> 4) * K32 + (k >> 5)) * 512 + (size_t)(((row & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7));
}

// V operand of the decode attention (self V cache and cross V): per (stream, head) the [rows][64] matrix is stored
// as MFMA A fragments of V^T, [rows/32][4 dim tiles][64 lanes][8]: lane (c = dim & 15, g) of fragment (t, dt) holds
// V[32t + 4g .. +3][16dt + c] and V[32t + 16 + 4g .. +3][16dt + c] — the k-slot <-> key permutation of the P
// operand the score MFMAs leave in registers.  A wave fetches a 32-key step of V as four contiguous 1 KiB loads.
__device__ __host__ __forceinline__ size_t vfrag_index(int key, int dim) {
    const int r = key & 31, rr = r & 15;
    return ((size_t)((key >> 5) * 4 + (dim >> 4)) * 64 + (dim & 15) + 16 * (rr >> 2)) * 8 + (r >> 4) * 4 + (rr & 3);
}

// V^T fragment store for MFMA epilogues.  In the C/D layout a lane (row = lane & 15, g = lane >> 4) holds 4 consecutive
// features (dims) of ONE row (key position); the V^T layout wants, per dim, 4 consecutive KEYS adjacent (8 bytes).  The 16
// lanes of a row group hold 16 consecutive rows, so a lane whose position is a multiple of 4 (`lead`) collects the values of
// the next three lanes (3 DPP row shifts per dim inside the 16-lane group) and writes 8 bytes per dim; a lane not `covered` by such
// a leader falls back to 2-byte stores.  All 64 lanes must call this convergently.  One 8-B store replaces four 2-B stores:
// the scattered 2-B stores were what made the launches that write V drain slowly (QKV -> self-attention boundary 3.7 us vs
// 1.9 us elsewhere) and they are a quarter of the cross-KV projection's epilogue traffic.
__device__ __forceinline__ void vt_store4(bf16_t* __restrict__ head, int pos, int dim0, float x0, float x1, float x2, float x3,
                                           bool valid, bool lead, bool covered)
{
    const float x[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // row_shl:n — lane i of a 16-lane row reads lane i+n (DPP: in the VALU, no LDS round trip like ds_bpermute)
        const int xi = __float_as_int(x[j]);
        const float y1 = __int_as_float(__builtin_amdgcn_update_dpp(0, xi, 0x101, 0xf, 0xf, true));
        const float y2 = __int_as_float(__builtin_amdgcn_update_dpp(0, xi, 0x102, 0xf, 0xf, true));
        const float y3 = __int_as_float(__builtin_amdgcn_update_dpp(0, xi, 0x103, 0xf, 0xf, true));
        bf16_t* p = head + vfrag_index(pos, dim0 + j);
        if (lead) {
            uint2 o; o.x = pack_bf2(x[j], y1); o.y = pack_bf2(y2, y3);
            *reinterpret_cast<uint2*>(p) = o;
        } else if (valid && !covered) {
            *p = f2bf(x[j]);
        }
    }
}

// Reductions over the four 16-lane rows of a wavefront (lanes c, c+16, c+32, c+48), every lane gets the result:
// gfx950's v_permlane16_swap / v_permlane32_swap exchange rows in the VALU (a few cycles) where __shfl_xor(.,16|32)
// goes through the LDS crossbar (ds_bpermute, ~100 cycles of latency on the critical path of every softmax step and
// LayerNorm).  Same operand pairing as the xor-16-then-xor-32 butterfly: bit-identical results.
__device__ __forceinline__ float rows4_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows4_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The encoder's GELU (exact erf form, HF:activations.py GELUActivation) = x * Phi(x), Phi(x) = erfc(-x / sqrt 2) / 2, with erfc from
// the Chebyshev fit  erfc(z) = t exp(-z^2 + P(t)), t = 1 / (1 + z / 2), z >= 0  (fractional error < 1.2e-7 everywhere): one
// branch-free path of ~21 VALU instructions (two of them quarter-rate: v_rcp_f32, v_exp_f32) where the library erff costs ~44 per
// element with both of its branches executed — the FC1 epilogue of the big-batch GEMM spent 5.8 k VALU instructions per wave and
// tile on it, as long as its MFMAs take (profiles/r03_encoder_gemm_parts.md).  Against the float64 value: |error| < 4e-7 over
// the whole line and relatively accurate in the negative tail, where 0.5 x (1 + erf) in fp32 cancels (tests/test_host.py checks
// the numpy restatement of this function).  The coefficients are P's times log2(e): the exponential is v_exp_f32 = 2^u.
__device__ __forceinline__ float gelu_phi(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.0f));
    float p = 0.17087277f * 1.4426950408889634f;
    p = fmaf(p, t, -0.82215223f * 1.4426950408889634f);
    p = fmaf(p, t, 1.48851587f * 1.4426950408889634f);
    p = fmaf(p, t, -1.13520398f * 1.4426950408889634f);
    p = fmaf(p, t, 0.27886807f * 1.4426950408889634f);
    p = fmaf(p, t, -0.18628806f * 1.4426950408889634f);
    p = fmaf(p, t, 0.09678418f * 1.4426950408889634f);
    p = fmaf(p, t, 0.37409196f * 1.4426950408889634f);
    p = fmaf(p, t, 1.00002368f * 1.4426950408889634f);
    p = fmaf(p, t, -1.26551223f * 1.4426950408889634f);
    const float u = fmaf(z * -1.4426950408889634f, z, p);
    const float he = 0.5f * t * __builtin_amdgcn_exp2f(u);       // erfc(z) / 2
    return x * (x < 0.f ? he : 1.0f - he);
}
// Two values at a time on the packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32: one issue slot for two lanes' worth of work per
// thread): the same Horner chain, 23 instructions per PAIR instead of 21 per element.  The FC1 epilogue of the big-batch encoder GEMMs is
// bound by VALU instruction issue (128 GELUs per lane and block tile), not by memory.
__device__ __forceinline__ f32x2_t gelu_phi2(f32x2_t x) {
    const f32x2_t z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const f32x2_t dn = __builtin_elementwise_fma(f32x2_t{0.5f, 0.5f}, z, f32x2_t{1.0f, 1.0f});
    const f32x2_t t = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
    constexpr float L2E = 1.4426950408889634f;
    constexpr float c[10] = {0.17087277f * L2E, -0.82215223f * L2E, 1.48851587f * L2E, -1.13520398f * L2E, 0.27886807f * L2E,
                             -0.18628806f * L2E, 0.09678418f * L2E, 0.37409196f * L2E, 1.00002368f * L2E, -1.26551223f * L2E};
    f32x2_t p = {c[0], c[0]};
#pragma unroll
    for (int i = 1; i < 10; ++i) p = __builtin_elementwise_fma(p, t, f32x2_t{c[i], c[i]});
    const f32x2_t u = __builtin_elementwise_fma(z * -L2E, z, p);
    const f32x2_t he = (t * 0.5f) * f32x2_t{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};      // erfc(z) / 2
    return f32x2_t{x[0] * (x[0] < 0.f ? he[0] : 1.0f - he[0]), x[1] * (x[1] < 0.f ? he[1] : 1.0f - he[1])};
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) {       // 16-B aligned
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p));
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ bf16x8_t ld_frag_nt(const bf16_t* p) {    // streamed-once weights: global_load_dwordx4 ... nt
    return __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)));
}

// Weight fragment of the weight-streaming GEMMs: bf16 (16 B per lane) or fp8 e4m3 (8 B per lane, the matrix's per-row
// scale is applied to the fp32 accumulator afterwards).  The fp8 tile has the packed layout of the bf16 one — element
// index = the same number, one byte per element — and is widened to bf16 exactly (every e4m3 value is a bf16 value), so
// the MFMA and the hi/lo token operand are those of the bf16 path.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
// NT: streamed once per pass (single-stream GEMMs) -> nt policy; re-read by other token-tile groups (batched GEMM) -> default
// 8 e4m3 values -> a bf16 MFMA fragment, whatever the build's decoder-weight type (attention operands are bf16 in both contracts)
// (round 6: v_cvt_scalef32_pk_bf16_fp8 / _pk_f16_fp8 with scale 1.0 — two e4m3 bytes to two 16-bit values in ONE instruction; rounds 4-5 went through
//  fp32, v_cvt_pk_f32_fp8 + a pack: twice the VALU issue in front of the cross-attention MFMAs of the fp8 K/V cache and of every fp8 weight fragment.
//  Same values for all 256 bytes: tests/microbench/fp8_cvt_direct.hip.)
__device__ __forceinline__ bf16x8_t fp8x8_to_bf16(unsigned v0, unsigned v1) {
    uint4 r;
    r.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v0, 1.0f, false)); r.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v0, 1.0f, true));
    r.z = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v1, 1.0f, false)); r.w = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v1, 1.0f, true));
    return __builtin_bit_cast(bf16x8_t, r);
}
// A product that must stay a product.  hipcc contracts a * b + c into one fma wherever a multiply meets an add after inlining, and WHETHER it does
// depends on the code around it: the same epilogue arithmetic came out as fma(x, rstd, bias) in one GEMM kernel and as mul + add in another
// (round 6: the cross-q projection of a two-tile pass differed from the 16-row launch's in the last bit of 28 % of its outputs; behind the fp16
// operand rounding of the next GEMM that is a token now and then).  Batch invariance needs every kernel that can finish a given output to round it
// the same way: products that feed an epilogue's add go through this (no instruction: the value just becomes opaque to the combiner).
__device__ __forceinline__ float nofuse(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ f32x4_t scale4v(f32x4_t v, float4 s) {
    return f32x4_t{nofuse(v[0] * s.x), nofuse(v[1] * s.y), nofuse(v[2] * s.z), nofuse(v[3] * s.w)};
}
__device__ __forceinline__ f32x4_t scale4(f32x4_t v, const float* wscale, int n) {       // n % 4 == 0
    return scale4v(v, *reinterpret_cast<const float4*>(wscale + n));
}

// store 4 consecutive values as a bf16 hi/lo pair (x = hi + lo keeps ~17 mantissa bits)
__device__ __forceinline__ void st_hilo4(bf16_t* hi, bf16_t* lo, float4 y)
{
    uint2 a; a.x = pack_bf2(y.x, y.y); a.y = pack_bf2(y.z, y.w);
    *reinterpret_cast<uint2*>(hi) = a;
    uint2 b;
    b.x = pack_bf2(y.x - __uint_as_float(a.x << 16), y.y - __uint_as_float(a.x & 0xffff0000u));
    b.y = pack_bf2(y.z - __uint_as_float(a.y << 16), y.w - __uint_as_float(a.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(lo) = b;
}

// ---- the decoder GEMMs' activation operand (the decode numerics contract) ----------------------------------------------
// Two contracts, one source: the library is built for one of them (build.py: libwm.so = hi / lo, libwm_f16.so = -DWM_ACT_F16; wm_create
// refuses a wm_config.act_fp16 that is not the build's).
//   hi / lo (rounds 1-5): an activation is a bf16 PAIR x = hi + lo (~17 mantissa bits) in two planes; two v_mfma_f32_16x16x32_bf16 per weight
//     fragment; decoder matrices bf16.
//   f16 (round 6): ONE fp16 plane (11 mantissa bits — the precision the reference's own fp16 / bf16 inference carries); one
//     v_mfma_f32_16x16x32_f16 per weight fragment; decoder matrices held as fp16 (exact from bf16 for |w| >= 2^-17, e4m3 exactly).  Half the
//     L2 -> CU bytes and half the MFMAs of every token operand: the 32-stream step is bound by exactly that (profiles/r06_upper_bounds.md).
// Storage stays a raw 16-bit container (bf16_t) either way: same packed layout, same plane arithmetic (plane unused with one plane).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {        // round-to-nearest-even, like torch's .half()
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
#ifdef WM_ACT_F16
#define WM_ACT_PLANES 1
typedef f16x8_t wfrag_t;                       // a weight fragment as the MFMA takes it
struct ActFrag { f16x8_t h; };
__device__ __forceinline__ f32x4_t mfma_act(wfrag_t a, const ActFrag& x, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, x.h, c, 0, 0, 0); }
__device__ __forceinline__ ActFrag act_ld(const bf16_t* p, size_t) { ActFrag f; f.h = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(p)); return f; }
__device__ __forceinline__ void act_st_frag(bf16_t* p, size_t, const ActFrag& f) { *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, f.h); }
__device__ __forceinline__ ActFrag act_split8(float4 y0, float4 y1) {
    uint4 h; h.x = pack_h2(y0.x, y0.y); h.y = pack_h2(y0.z, y0.w); h.z = pack_h2(y1.x, y1.y); h.w = pack_h2(y1.z, y1.w);
    ActFrag f; f.h = __builtin_bit_cast(f16x8_t, h); return f;
}
__device__ __forceinline__ void act_st4(bf16_t* p, bf16_t*, float4 y) {      // 4 consecutive k of one row
    uint2 a; a.x = pack_h2(y.x, y.y); a.y = pack_h2(y.z, y.w);
    *reinterpret_cast<uint2*>(p) = a;
}
__device__ __host__ __forceinline__ float w16_to_f32(bf16_t v) {            // a decoder-matrix element of the blob (fp16 in this build)
    return (float)__builtin_bit_cast(_Float16, v);
}
#else
#define WM_ACT_PLANES 2
typedef bf16x8_t wfrag_t;
struct ActFrag { bf16x8_t h, l; };
__device__ __forceinline__ f32x4_t mfma_act(wfrag_t a, const ActFrag& x, f32x4_t c) { c = mfma16(a, x.h, c); return mfma16(a, x.l, c); }
__device__ __forceinline__ ActFrag act_ld(const bf16_t* p, size_t plane) { ActFrag f; f.h = ld_frag(p); f.l = ld_frag(p + plane); return f; }
__device__ __forceinline__ void act_st_frag(bf16_t* p, size_t plane, const ActFrag& f) {
    *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, f.h); *reinterpret_cast<uint4*>(p + plane) = __builtin_bit_cast(uint4, f.l);
}
__device__ __forceinline__ ActFrag act_split8(float4 y0, float4 y1) {
    uint4 h, l;
    h.x = pack_bf2(y0.x, y0.y); h.y = pack_bf2(y0.z, y0.w); h.z = pack_bf2(y1.x, y1.y); h.w = pack_bf2(y1.z, y1.w);
    l.x = pack_bf2(y0.x - __uint_as_float(h.x << 16), y0.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf2(y0.z - __uint_as_float(h.y << 16), y0.w - __uint_as_float(h.y & 0xffff0000u));
    l.z = pack_bf2(y1.x - __uint_as_float(h.z << 16), y1.y - __uint_as_float(h.z & 0xffff0000u));
    l.w = pack_bf2(y1.z - __uint_as_float(h.w << 16), y1.w - __uint_as_float(h.w & 0xffff0000u));
    ActFrag f; f.h = __builtin_bit_cast(bf16x8_t, h); f.l = __builtin_bit_cast(bf16x8_t, l); return f;
}
__device__ __forceinline__ void act_st4(bf16_t* p, bf16_t* plo, float4 y) { st_hilo4(p, plo, y); }
__device__ __host__ __forceinline__ float w16_to_f32(bf16_t v) {
    const uint32_t u = ((uint32_t)v) << 16; return __builtin_bit_cast(float, u);
}
#endif

// 8 e4m3 values (two dwords) -> the MFMA's weight fragment type of this build (bf16 or fp16: both hold every e4m3 value exactly)
__device__ __forceinline__ uint4 fp8x8_to_w16(unsigned v0, unsigned v1) {
    uint4 r;
#if WM_ACT_PLANES == 1
    r.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v0, 1.0f, false)); r.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v0, 1.0f, true));
    r.z = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v1, 1.0f, false)); r.w = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v1, 1.0f, true));
#else
    r.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v0, 1.0f, false)); r.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v0, 1.0f, true));
    r.z = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v1, 1.0f, false)); r.w = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v1, 1.0f, true));
#endif
    return r;
}
template <bool W8, bool NT>
__device__ __forceinline__ wfrag_t ld_wfrag(const bf16_t* W, size_t elem) {
    if constexpr (W8) {
        const u32x2_t* p8 = reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(W) + elem);
        const u32x2_t v = NT ? __builtin_nontemporal_load(p8) : *p8;
        return __builtin_bit_cast(wfrag_t, fp8x8_to_w16(v[0], v[1]));
    } else {
        const u32x4_t* p = reinterpret_cast<const u32x4_t*>(W + elem);
        return __builtin_bit_cast(wfrag_t, NT ? __builtin_nontemporal_load(p) : *p);
    }
}

// ---- LayerNorm folded into the GEMM it feeds (round 6) -------------------------------------------------------------------------
//   W LN(x) + b  =  rstd * ( W (gamma o x)  -  mean * c )  +  b',      c[n] = sum_k W[n][k] gamma[k],   b'[n] = b[n] + sum_k W[n][k] beta[k]
// (exact algebra: HF LayerNorm, modeling_whisper.py:416-505 pre-LN layers).  The launch that PRODUCES the residual row x (out-proj / FC2 epilogue,
// embed, final LayerNorm for the Medusa-Block layer) also writes the operand gamma o x as packed hi / lo planes and, per (row, 16-feature tile),
// the partial (sum x, sum x^2); the consuming GEMM is then a plain packed-operand GEMM whose block sums the row's partials in a FIXED order
// (groups of 8 tiles in tile order, then the groups in order: the same in the 16-row, two-tile and token-tile kernels => a B-stream run stays
// bit-identical to B single-stream runs) and applies mean / rstd to the accumulator.  No LayerNorm launch (three per decoder layer in a batched
// pass), no statistics -> barrier -> normalise prologue in the single-stream launches.  c and b' are computed once in wm_create (fp64 sums).
// Partial table layout: stats[tile][row] (float2; row stride `ld` = row capacity, a multiple of 16): a finishing wave stores 16 rows x 8 B = one
// 128-B line per tile.
struct FoldIn {
    const float2* stats; const float* c; int T16; int ld; float inv_d;
};

// the thread's share of a block's row statistics: partial group `grp` (tiles 8 grp .. 8 grp + 7, summed in order) of row `row`
struct FoldPart { float2 v[8]; };
__device__ __forceinline__ void fold_part_load(FoldPart& fp, const FoldIn& f, int row, int grp) {
    const float2* p = f.stats + (size_t)(grp * 8) * f.ld + row;
#pragma unroll
    for (int j = 0; j < 8; ++j) fp.v[j] = p[(size_t)j * f.ld];
}
__device__ __forceinline__ float2 fold_part_sum(const FoldPart& fp) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s += fp.v[j].x; q += fp.v[j].y; }
    return make_float2(s, q);
}
// mean / rstd of a row from its G group partials in LDS (part[grp * nrows + r]), summed in group order
__device__ __forceinline__ float2 fold_row_stat(const float2* part, int r, int nrows, int G, float inv_d) {
    float s = 0.f, q = 0.f;
    for (int g2 = 0; g2 < G; ++g2) { const float2 p = part[g2 * nrows + r]; s += p.x; q += p.y; }
    // (explicit fma / nofuse: every kernel rounds these the same way whatever the compiler would contract around them — see nofuse above)
    const float mean = s * inv_d;
    return make_float2(mean, rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, q * inv_d), 0.f) + 1e-5f));
}
__device__ __forceinline__ f32x4_t fold_apply(f32x4_t v, float2 mr, float4 c) {
    return f32x4_t{nofuse(mr.y * __builtin_fmaf(-mr.x, c.x, v[0])), nofuse(mr.y * __builtin_fmaf(-mr.x, c.y, v[1])),
                   nofuse(mr.y * __builtin_fmaf(-mr.x, c.z, v[2])), nofuse(mr.y * __builtin_fmaf(-mr.x, c.w, v[3]))};
}
// a 4-feature share of a row's statistics partial (sum, sum of squares): fixed rounding points
__device__ __forceinline__ float sum4(float4 y) { return (y.x + y.y) + (y.z + y.w); }
__device__ __forceinline__ float sumsq4(float4 y) { return __builtin_fmaf(y.x, y.x, y.y * y.y) + __builtin_fmaf(y.z, y.z, y.w * y.w); }
