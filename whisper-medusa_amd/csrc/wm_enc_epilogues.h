// wm_enc_epilogues.h — wave-level epilogues of the encoder's tile GEMMs (wm_encoder.hip only): how a wave's NI x NJ output tiles
// leave the accumulators.  The per-element functors (bias, GELU, residual, head-split scatter) are in wm_epilogues.h, shared with the
// decode GEMMs; nothing here is on the decode path.
#pragma once
#include "wm_epilogues.h"

// Epilogue of a wave's NI x NJ output tiles (encoder GEMMs): per feature tile the operands of all NJ token tiles (bias, residual,
// positions) are requested in ONE batch, then the arithmetic and the stores run.  Calling store4 tile by tile made every tile a
// dependent memory round trip — the compiler cannot hoist loads over the stores of the previous tile (possible aliasing) — and
// 32 round trips per wave were a third of the big-batch GEMM time (tests/microbench/r03_call3.sh: 117 -> 89 ms per 32-clip encoder
// pass with the epilogue switched off).
template <int NI, int NJ, class Ep>
__device__ __forceinline__ void ep_tiles_generic(const Ep& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        EpPre pre[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) pre[j] = ep.pre(m0 + j * 16, n0 + i * 16);
        // loads of the NEXT group must not be hoisted above this group's stores: loads and stores share one counter (vmcnt) and
        // complete out of order with respect to each other, so a load consumed after a store has been issued costs s_waitcnt vmcnt(0),
        // i.e. the completion of that store — per tile
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) ep.fin(m0 + j * 16, n0 + i * 16, acc[i][j], pre[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NI, int NJ, class Ep>
__device__ __forceinline__ void ep_tiles(const Ep& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ]) { ep_tiles_generic<NI, NJ>(ep, m0, n0, acc); }

// ---- wave-level epilogues of the encoder's attention projections -----------------------------------------------------------
// The per-tile functors above recompute, for each of a wave's 32 output tiles, what is the same for all of them: which clip
// the token belongs to (an integer division by a run-time Spad per tile), which of q / k / v and which head the feature belongs
// to, 64-bit slab addresses — ~120 VALU instructions per tile, 3.9 k per wave and 256 x 256 tile, during which the CU's matrix
// pipes idle (tests/microbench/r03_call10.sh).  A wave's NJ x 16 <= 128 tokens lie in ONE clip (Spad % 128 == 0 and the span is
// aligned to its size) and its NI x 16 = 64 features in ONE head of ONE of q / k / v (d % 64 == 0, span 64-aligned): the overloads
// below derive clip, head and slab once per wave and leave two adds and a store per tile.  Same values, same stores: results are
// bit-identical to the per-tile functors (which the fp8 wrappers and the unit tests still use).
//
// `swapped` tiles: the pipelined 256 x 256 kernel computes the feature tiles of V (a launch of their own) with the MFMA operands exchanged
// (tokens as the A operand), so a lane owns 4 consecutive TOKENS of one feature — exactly the 8-byte unit of the V^T fragment
// layout (vfrag_index): one plain store per tile where the feature-major orientation needs 12 DPP row shifts, four packs and four
// masked stores per tile to transpose inside the wave.  a x b and b x a give the same products in the same k order: bit-identical.

template <int NI, int NJ>      // [s][64] slab of one (clip, head): lane (r, g) stores features 4g .. 4g+3 of token r of each tile
__device__ __forceinline__ void wave_store_rows(bf16_t* __restrict__ row0, const float* __restrict__ bias_w, float scale, bool scaled,
                                                f32x4_t (&acc)[NI][NJ])
{
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float4 bb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bb[i] = *reinterpret_cast<const float4*>(bias_w + i * 16 + 4 * g);
    bf16_t* lp = row0 + r * 64 + 4 * g;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float x0 = acc[i][j][0] + bb[i].x, x1 = acc[i][j][1] + bb[i].y, x2 = acc[i][j][2] + bb[i].z, x3 = acc[i][j][3] + bb[i].w;
            if (scaled) { x0 *= scale; x1 *= scale; x2 *= scale; x3 *= scale; }
            uint2 o; o.x = pack_bf2(x0, x1); o.y = pack_bf2(x2, x3);
            *reinterpret_cast<uint2*>(lp + j * 1024 + i * 16) = o;
        }
}

template <int NI, int NJ>      // V^T fragments from the feature-major orientation (DPP transpose inside the wave, vt_store4)
__device__ __forceinline__ void wave_store_vt(bf16_t* __restrict__ slab, int s0, const float* __restrict__ bias_w, f32x4_t (&acc)[NI][NJ])
{
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const bool lead = (r & 3) == 0;              // s0 is a multiple of 16
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float4 bb = *reinterpret_cast<const float4*>(bias_w + i * 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            vt_store4(slab, s0 + j * 16 + r, i * 16 + 4 * g, acc[i][j][0] + bb.x, acc[i][j][1] + bb.y, acc[i][j][2] + bb.z, acc[i][j][3] + bb.w,
                      true, lead, !lead);
    }
}

template <int NI, int NJ>      // V^T fragments from the token-major (`swapped`) orientation: lane (r, g) holds tokens 4g .. 4g+3 of feature r
__device__ __forceinline__ void wave_store_vt_swapped(bf16_t* __restrict__ slab, int s0, const float* __restrict__ bias_w, f32x4_t (&acc)[NI][NJ])
{
    const int lane = threadIdx.x & 63, r = lane & 15;
    float bb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bb[i] = bias_w[i * 16 + r];
    bf16_t* lp = slab + lane * 8;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int sj = s0 + j * 16;              // vfrag_index(sj + 4g + e, 16 i + r) = ((sj >> 5) * 4 + i) * 512 + lane * 8 + ((sj >> 4) & 1) * 4 + e
        bf16_t* pj = lp + (sj >> 5) * 2048 + ((sj >> 4) & 1) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            uint2 o;
            o.x = pack_bf2(acc[i][j][0] + bb[i], acc[i][j][1] + bb[i]);
            o.y = pack_bf2(acc[i][j][2] + bb[i], acc[i][j][3] + bb[i]);
            *reinterpret_cast<uint2*>(pj + i * 512) = o;
        }
    }
}

struct WaveTile { int b, s0, nt0; };             // clip, first position in the clip and first feature of a wave's tiles (all wave-uniform)
__device__ __forceinline__ WaveTile wave_tile(int m0, int n0, int Spad) {
    const int lane = threadIdx.x & 63;
    const int mt0 = __builtin_amdgcn_readfirstlane(m0 - (lane & 15));
    WaveTile w; w.nt0 = __builtin_amdgcn_readfirstlane(n0 - 4 * (lane >> 4));
    w.b = mt0 / Spad; w.s0 = mt0 - w.b * Spad;
    return w;
}

// Which feature tiles (of `width` features) an epilogue wants token-major: of every `period` consecutive tiles the last period - plain.
// Host side: the launcher of the pipelined 256 x 256 kernel splits such a GEMM into two launches (EpWantsSwap<Ep>::value).
template <class Ep> struct EpWantsSwap { static constexpr bool value = false; };
template <> struct EpWantsSwap<EpQKVEnc> { static constexpr bool value = true; };
template <> struct EpWantsSwap<EpCrossKV> { static constexpr bool value = true; };
static inline bool ep_swap_split(const EpQKVEnc& ep, int width, int& period, int& plain) {
    if (ep.d % width) return false;
    period = 3 * ep.d / width; plain = 2 * ep.d / width; return true;
}
static inline bool ep_swap_split(const EpCrossKV& ep, int width, int& period, int& plain) {
    if (ep.d % width) return false;
    period = 2 * ep.d / width; plain = ep.d / width; return true;
}
template <int NI, int NJ, class Ep>              // only reached for the epilogues that ask for swapped tiles
__device__ __forceinline__ void ep_tiles_swapped(const Ep&, int, int, f32x4_t (&)[NI][NJ]) {}

template <int NI, int NJ>
__device__ __forceinline__ void ep_tiles(const EpQKVEnc& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    static_assert(NI == 4, "a wave's features are one 64-wide head");
    const WaveTile w = wave_tile(m0, n0, ep.Spad);
    const int seg = w.nt0 / ep.d, c0 = w.nt0 - seg * ep.d;                     // 0: q, 1: k, 2: v;  c0 = 64 * head
    const size_t head = (size_t)w.b * ep.H + (c0 >> 6);
    if (seg < 2) wave_store_rows<NI, NJ>((seg == 0 ? ep.q : ep.k) + (head * ep.Spad + w.s0) * 64, ep.bias + w.nt0, 0.125f, seg == 0, acc);
    else wave_store_vt<NI, NJ>(ep.vt + head * 64 * ep.Spad, w.s0, ep.bias + w.nt0, acc);
}
template <int NI, int NJ>
__device__ __forceinline__ void ep_tiles_swapped(const EpQKVEnc& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    const WaveTile w = wave_tile(m0, n0, ep.Spad);
    const int c0 = w.nt0 - 2 * ep.d;
    wave_store_vt_swapped<NI, NJ>(ep.vt + ((size_t)w.b * ep.H + (c0 >> 6)) * 64 * ep.Spad, w.s0, ep.bias + w.nt0, acc);
}

template <int NI, int NJ>
__device__ __forceinline__ void ep_tiles(const EpCrossKV& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    static_assert(NI == 4, "a wave's features are one 64-wide head");
    const WaveTile w = wave_tile(m0, n0, ep.Spad);
    const int kvl = w.nt0 / (2 * ep.d), rem = w.nt0 - kvl * 2 * ep.d;
    const bool isv = rem >= ep.d;
    const int c0 = isv ? rem - ep.d : rem;
    const size_t slab = (((size_t)kvl * ep.B + w.b) * ep.H + (c0 >> 6)) * ep.Spad * 64;
    if (!isv) wave_store_rows<NI, NJ>(ep.kx + slab + (size_t)w.s0 * 64, ep.bias + w.nt0, 1.0f, false, acc);
    else wave_store_vt<NI, NJ>(ep.vx + slab, w.s0, ep.bias + w.nt0, acc);
}
template <int NI, int NJ>
__device__ __forceinline__ void ep_tiles_swapped(const EpCrossKV& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    const WaveTile w = wave_tile(m0, n0, ep.Spad);
    const int kvl = w.nt0 / (2 * ep.d), c0 = w.nt0 - kvl * 2 * ep.d - ep.d;
    wave_store_vt_swapped<NI, NJ>(ep.vx + (((size_t)kvl * ep.B + w.b) * ep.H + (c0 >> 6)) * ep.Spad * 64, w.s0, ep.bias + w.nt0, acc);
}

// Residual GEMMs (out-proj, FC2: h += bias + X W^T) in the pipelined 256 x 256 kernel: the accumulators START as the residual tile.
// The generic epilogue reads h in NI batches of NJ loads per wave, 64 KiB in flight per CU, each batch a memory round trip (and the
// acknowledgement of the previous batch's stores: loads and stores share one counter) with the CU's matrix pipes idle: 34 us per
// block tile, half of the K = 1280 GEMM's time (596 TFLOP/s against 840-930 for the other GEMMs, profiles/r03_kernel_trace_encoder_b32.md;
// starting every other block late did not help — it is latency x bytes-in-flight per CU, not HBM contention).  More loads in flight
// need registers the epilogue does not have (128 accumulators; batches of 16 loads made the allocator spill the accumulators).  Loading
// the tile INTO the accumulators before the K loop costs no register, puts all 32 loads of a wave in flight at once, and leaves an
// epilogue of one add and one store per tile, behind no load.  Numerics: h + (p_1 + ... + p_n) + bias becomes ((h + p_1) + ... + p_n)
// + bias — the same fp32 terms in another order (one rounding per MFMA at the magnitude of h instead of one at the end; the residual
// stream is O(1-100), so <= 1e-5 absolute per layer against bf16-rounded operands downstream) — inside the encoder tolerance of the
// parity tests, and the few-clip kernels keep the classic order (big-batch and few-clip outputs never were bit-identical: different
// tile shapes).
template <class Ep> struct EpAccInit { static constexpr bool value = false; };
template <> struct EpAccInit<EpResidual> { static constexpr bool value = true; };

template <int NI, int NJ, class Ep>
__device__ __forceinline__ void ep_acc_init(const Ep&, int, int, f32x4_t (&)[NI][NJ]) {}
template <int NI, int NJ, class Ep>
__device__ __forceinline__ void ep_tiles_init(const Ep&, int, int, f32x4_t (&)[NI][NJ]) {}

template <int NI, int NJ>
__device__ __forceinline__ void ep_acc_init(const EpResidual& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int mt0 = __builtin_amdgcn_readfirstlane(m0 - r), nt0 = __builtin_amdgcn_readfirstlane(n0 - 4 * g);
    const char* hb = reinterpret_cast<const char*>(ep.h + (size_t)mt0 * ep.ld + nt0);      // wave-uniform base, 32-bit lane offsets
    // the lane offsets are recomputed per tile on purpose (opaque copy of the lane's row): as loop invariants they were hoisted out of
    // the tile loop, spilled across the 256-VGPR K loop, and every reload's s_waitcnt vmcnt(0) cut the 32 loads into 8 round trips
    int rv = r;
    asm volatile("" : "+v"(rv));
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned off = (unsigned)(((j * 16 + rv) * ep.ld + 4 * g) * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = *reinterpret_cast<const f32x4_t*>(hb + off + i * 64);
    }
}
template <int NI, int NJ>
__device__ __forceinline__ void ep_tiles_init(const EpResidual& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int mt0 = __builtin_amdgcn_readfirstlane(m0 - r), nt0 = __builtin_amdgcn_readfirstlane(n0 - 4 * g);
    char* hb = reinterpret_cast<char*>(ep.h + (size_t)mt0 * ep.ld + nt0);
    float4 bb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bb[i] = *reinterpret_cast<const float4*>(ep.bias + nt0 + 4 * g + i * 16);
    int rv = r;
    asm volatile("" : "+v"(rv));                   // (see ep_acc_init)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned off = (unsigned)(((j * 16 + rv) * ep.ld + 4 * g) * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            *reinterpret_cast<float4*>(hb + off + i * 64) = make_float4(acc[i][j][0] + bb[i].x, acc[i][j][1] + bb[i].y, acc[i][j][2] + bb[i].z, acc[i][j][3] + bb[i].w);
    }
}

// Residual epilogue of the few-clip kernels (a wave holds <= 16 tiles): all residual loads in one batch, the arithmetic, then all stores —
// the generic form's row guards (m < M) put every tile in a basic block of its own and the compiler opened each with s_waitcnt vmcnt(0),
// i.e. waited for the previous tile's STORE to be acknowledged: 8-16 dependent memory round trips per wave at the end of a 17 us GEMM.
// The encoder's row count is a multiple of every tile height, so the guards are only kept as a wave-uniform fallback.
template <int NI, int NJ>
__device__ __forceinline__ void ep_tiles(const EpResidual& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int mt0 = __builtin_amdgcn_readfirstlane(m0 - r), nt0 = __builtin_amdgcn_readfirstlane(n0 - 4 * g);
    if constexpr (NI * NJ > 16) {
        ep_tiles_generic<NI, NJ>(ep, m0, n0, acc);
    } else {
        if (mt0 + NJ * 16 > ep.M) { ep_tiles_generic<NI, NJ>(ep, m0, n0, acc); return; }
        char* hb = reinterpret_cast<char*>(ep.h + (size_t)mt0 * ep.ld + nt0);
        float4 bb[NI], rr[NI][NJ];
        unsigned off[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) off[j] = (unsigned)(((j * 16 + r) * ep.ld + 4 * g) * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            bb[i] = *reinterpret_cast<const float4*>(ep.bias + nt0 + 4 * g + i * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) rr[i][j] = *reinterpret_cast<const float4*>(hb + off[j] + i * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *reinterpret_cast<float4*>(hb + off[j] + i * 64) =
                    make_float4((rr[i][j].x + bb[i].x) + acc[i][j][0], (rr[i][j].y + bb[i].y) + acc[i][j][1],
                                (rr[i][j].z + bb[i].z) + acc[i][j][2], (rr[i][j].w + bb[i].w) + acc[i][j][3]);
    }
}

// FC1 epilogue of a wave's NI x NJ tiles: bias + GELU + packed bf16 store.  The per-tile functor guards its bias load and its store by
// row validity, so every tile is a basic block of its own that opens with s_waitcnt vmcnt(0) — i.e. waits for the previous tile's STORE
// to be acknowledged: 25-32 dependent round trips per wave and block tile in the largest GEMM of the encoder (ISA of round 3).  The
// encoder's row count is a multiple of every tile height (and its FC1 output is plain bf16): four bias loads, then NI x NJ x (GELU, pack,
// store) with nothing to wait for.  ACT = 2 is the encoder's functor only.
template <int NI, int NJ>
__device__ __forceinline__ void ep_tiles(const EpPackedAct<2>& ep, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    float4 bb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bb[i] = *reinterpret_cast<const float4*>(ep.bias + n0 + i * 16);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x2_t g0 = gelu_phi2(f32x2_t{acc[i][j][0] + bb[i].x, acc[i][j][1] + bb[i].y});
            const f32x2_t g1 = gelu_phi2(f32x2_t{acc[i][j][2] + bb[i].z, acc[i][j][3] + bb[i].w});
            uint2 u; u.x = pack_bf2(g0[0], g0[1]); u.y = pack_bf2(g1[0], g1[1]);
            *reinterpret_cast<uint2*>(ep.out + packed_index(m0 + j * 16, n0 + i * 16, ep.K32out)) = u;
        }
}
