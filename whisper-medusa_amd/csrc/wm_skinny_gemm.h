// wm_skinny_gemm.h — weight-streaming MFMA GEMM for the decode step (token rows M <= 16 per launch; more rows through
// the token-tile kernel below).
//
//   out[m][n] = sum_k X[m][k] * W[n][k]          X: M x K activations, W: N x K weights (packed bf16 or fp8 e4m3)
//
// The decode step is HBM-bound and, at one stream, LATENCY-bound: a pass is a chain of ~270 dependent launches, each of
// which moves 3-13 MB.  The kernel is therefore organised around ONE memory round trip per launch:
//   * every global load of the launch is issued in one batch at kernel entry, in the order the data is needed — the
//     wave's whole K-slice of the weight stream (non-temporal 16-B loads, 1 KiB packed fragments), its slice of the token
//     operand, the LayerNorm parameters (LDS-DMA, no registers) and the operands of the epilogue (bias, residual, cache
//     position: wm_epilogues.h `pre`) — and nothing touches a loaded value before all of it is in flight;
//   * one wavefront owns one 16-row weight tile x one K-slice (NK fragments, a template parameter: straight-line code, so
//     the compiler's vmcnt bookkeeping is exact); weights are the MFMA A operand, tokens the B operand;
//   * the token operand of a LayerNorm-fused GEMM never goes through LDS: a wave normalises exactly the fragments of its own
//     K-slice, in registers, in MFMA B layout (only the per-row statistics cross waves, 8 bytes per row and slice);
//   * rows >= M of the token tile are never loaded (exec-masked lanes): a 1-row base pass reads 1/16 of the operand;
//   * the KSPLIT waves of a row tile reduce through LDS in a fixed order (deterministic).
//
// Activations are carried as a bf16 HI/LO PAIR (x = hi + lo, ~17 mantissa bits): two MFMAs per weight fragment instead
// of one — free in a bandwidth-bound kernel — so decoder activations are never rounded to 8 bits and the fp32-accumulate
// results match the oracle to ~1e-6 (no rounding-flip cascades).
//
// Per output element the accumulation order is: K-slices in order, inside a slice k ascending, hi then lo.  The token-tile
// kernel (R > 16 rows) keeps that order and shares the LayerNorm code, so a B-stream run is bit-identical to B
// single-stream runs.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "wm_common.h"
#include "wm_epilogues.h"

// ---- optional in-kernel timeline (build with -DWM_TIMELINE -> libwm_tl.so; tests/microbench/timeline.py) ----------
// Thread 0 of every block appends one record {tag, block, realtime at entry / exit (100 MHz, chip-wide), shader cycles from
// entry to "operands ready", "products done" and exit}.  Not compiled into the product library.
#ifdef WM_TIMELINE
struct TlRec { unsigned tag, block; unsigned long long rt0, rt1; unsigned c_prep, c_mid, c_end, pad; };
static __device__ TlRec* g_tl_buf = nullptr;
static __device__ unsigned* g_tl_idx = nullptr;
static __device__ unsigned g_tl_cap = 0;
static thread_local int g_tl_tag = 0;              // host: tag of the next launch
struct TlProbe {
    unsigned long long rt0, c0; unsigned c_prep, c_mid; int tag;
    __device__ __forceinline__ void begin(int t) { tag = t; rt0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_readcyclecounter(); c_mid = 0; c_prep = 0; }
    __device__ __forceinline__ void prep() { c_prep = (unsigned)(__builtin_readcyclecounter() - c0); }
    __device__ __forceinline__ void mid() { c_mid = (unsigned)(__builtin_readcyclecounter() - c0); }
    __device__ __forceinline__ void end() {
        if (threadIdx.x != 0 || !g_tl_buf) return;
        const unsigned c_end = (unsigned)(__builtin_readcyclecounter() - c0);
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        const unsigned i = atomicAdd(g_tl_idx, 1u);
        if (i < g_tl_cap) g_tl_buf[i] = TlRec{(unsigned)tag, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), rt0, rt1, c_prep, c_mid, c_end, 0u};
    }
};
#define TL_ARG , int tl_tag
#define TL_PASS , g_tl_tag
#define TL_BEGIN TlProbe tl_; tl_.begin(tl_tag);
#define TL_PREP tl_.prep();
#define TL_MID tl_.mid();
#define TL_END tl_.end();
#define TL_SET(t) (g_tl_tag = (t))
#else
#define TL_ARG
#define TL_PASS
#define TL_BEGIN
#define TL_PREP
#define TL_MID
#define TL_END
#define TL_SET(t) ((void)0)
#endif

// ---- weight fragments: raw load now, widen later (an fp8 fragment must not be converted before the batch is in flight) ----
template <bool W8> struct WRaw { typedef u32x4_t type; };
template <> struct WRaw<true> { typedef u32x2_t type; };

template <bool W8, bool NT>
__device__ __forceinline__ typename WRaw<W8>::type ld_wraw(const bf16_t* W, size_t elem) {
    if constexpr (W8) {
        const u32x2_t* p8 = reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(W) + elem);
        return NT ? __builtin_nontemporal_load(p8) : *p8;
    } else {
        const u32x4_t* p = reinterpret_cast<const u32x4_t*>(W + elem);
        return NT ? __builtin_nontemporal_load(p) : *p;
    }
}
template <bool W8>
__device__ __forceinline__ wfrag_t w_expand(typename WRaw<W8>::type v) {
    if constexpr (W8) return __builtin_bit_cast(wfrag_t, fp8x8_to_w16(v[0], v[1]));      // e4m3 -> bf16 / fp16 is exact
    else return __builtin_bit_cast(wfrag_t, v);
}

// 8 fp32 values -> bf16 hi / lo MFMA fragments
__device__ __forceinline__ void split_hilo8(float4 y0, float4 y1, bf16x8_t& hi, bf16x8_t& lo)
{
    uint4 h, l;
    h.x = pack_bf2(y0.x, y0.y); h.y = pack_bf2(y0.z, y0.w); h.z = pack_bf2(y1.x, y1.y); h.w = pack_bf2(y1.z, y1.w);
    l.x = pack_bf2(y0.x - __uint_as_float(h.x << 16), y0.y - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf2(y0.z - __uint_as_float(h.y << 16), y0.w - __uint_as_float(h.y & 0xffff0000u));
    l.z = pack_bf2(y1.x - __uint_as_float(h.z << 16), y1.y - __uint_as_float(h.z & 0xffff0000u));
    l.w = pack_bf2(y1.z - __uint_as_float(h.w << 16), y1.w - __uint_as_float(h.w & 0xffff0000u));
    hi = __builtin_bit_cast(bf16x8_t, h); lo = __builtin_bit_cast(bf16x8_t, l);
}

__device__ __forceinline__ void glds16_f(const float* gsrc, char* lds_wave_base)     // LDS-DMA, 16 B per lane
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- next-launch prefetch riding on this launch --------------------------------------------------------------------
// The single-stream chain is latency-bound and HBM idles most of the time.  A launch may therefore carry extra blocks that
// do nothing but touch the bytes the NEXT launch's blocks will read (its weights, or the cross K/V), so that those find them
// in L2: job j = what block j of the consumer reads, fetched by the extra block whose global id is congruent to j modulo 8
// — the dispatcher places block b on XCD b % 8 (observed, MI355X_MICROARCH.md), i.e. into the L2 the consumer reads through;
// another placement only turns the L2 hit into an Infinity-Cache hit.  A pure hint: results cannot change, nothing waits.
// (Tried first as a second stream inside the hipGraph: the fork/join event nodes raised every launch boundary from 1.9 to
// 12-14 us — profiles/r02_timeline_sidestream_prefetch.md.)
struct PfJob { const char* p0; const char* p1; unsigned job_bytes; unsigned n_jobs; unsigned long long total; };
static thread_local PfJob g_pf_job = {nullptr, nullptr, 0u, 0u, 0ull};     // host: rides on the next launch, which clears it
// host: a second matrix OF THE SAME SIZE the next batched LayerNorm launch (R > 16 rows) should also pull in (LN3 + FC1: the FC2 weights, which
// no launch in between can carry); cleared by that launch
static thread_local const void* g_ln_pf_extra = nullptr;
// host: the prefetch job of a BATCHED launch (k_rows_gemm / k_skinny2_gemm; round 6: with the LayerNorm folded there is no LayerNorm launch left to
// carry the next GEMM's weights, so the launch that produces the residual row carries them); sliced = pf_block_sliced (token-tile consumer)
static thread_local PfJob g_pf_batched = {nullptr, nullptr, 0u, 0u, 0ull};
static thread_local int g_pf_batched_sliced = 0;

__device__ __forceinline__ void pf_block(const PfJob& pf, int job)
{
    if (job < 0 || job >= (int)pf.n_jobs) return;
    const unsigned long long lo = (unsigned long long)job * pf.job_bytes;
    const unsigned long long hi = min(lo + pf.job_bytes, pf.total);
    // The loads are issued from inline asm so that nothing waits for them one by one; their destination is ONE register quad
    // that stays allocated ("+v" keeps it live) until the final wait — a destination the compiler believed dead would be
    // reused while loads are still in flight and be overwritten when they land (cdna_hip_programming.md §5.7 item 1).
    u32x4_t sink = {0u, 0u, 0u, 0u};
    for (unsigned long long i = lo + (unsigned long long)threadIdx.x * 16; i < hi; i += (unsigned long long)blockDim.x * 16) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(pf.p0 + i) : "memory");
        if (pf.p1) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(pf.p1 + i) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) :: "memory");
}
__host__ __device__ __forceinline__ int pf_round8(int n) { return (n + 7) & ~7; }

// XCD-sliced variant for the token-tile GEMM (k_tile_gemm): that kernel gives XCD x the x-th eighth of the weight row tiles, so job j
// (run by a block on XCD j % 8) fetches piece j / 8 of the x-th eighth of the matrix: n_jobs = 8 * pieces per slice.
__device__ __forceinline__ void pf_block_sliced(const PfJob& pf, int job)
{
    if (job < 0 || job >= (int)pf.n_jobs) return;
    const unsigned long long slice = ((pf.total + 7) / 8 + 1023) & ~1023ull;
    const unsigned long long base = (unsigned long long)(job & 7) * slice;
    const unsigned long long lo = base + (unsigned long long)(job >> 3) * pf.job_bytes;
    const unsigned long long hi = min(min(lo + pf.job_bytes, base + slice), pf.total);
    u32x4_t sink = {0u, 0u, 0u, 0u};
    for (unsigned long long i = lo + (unsigned long long)threadIdx.x * 16; i < hi; i += (unsigned long long)blockDim.x * 16) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(pf.p0 + i) : "memory");
        if (pf.p1) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(pf.p1 + i) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) :: "memory");
}

// ---- token operand: packed hi/lo planes in global memory (written by the previous kernel's epilogue) ------------
// Kernel-argument form of a loader (round 5): three pointers and two ints passed as SCALAR kernel arguments right behind W, so that the kernarg
// preload (the first 16 dwords) covers everything the launch's request batch needs.  A loader passed as a struct is never preloaded (LLVM
// preloads scalar arguments only): the ISA of rounds 2-4 had an s_load + s_waitcnt lgkmcnt(0) — a scalar-memory round trip — between the first
// four weight requests and the rest of the batch in the plain launches, and in front of EVERY request in the LayerNorm-fused ones.
struct LdArgs { const void* a; const void* b; const void* c; int i0, i1; };
struct LdPacked {
    const bf16_t* X; int K32; size_t plane; int M;           // lo plane at X + plane; rows >= M are not read
    static constexpr bool kNorm = false;
    LdArgs pack() const { return LdArgs{X, reinterpret_cast<const void*>(plane), nullptr, M, 0}; }
    bool packs() const { return true; }
    __device__ __forceinline__ static LdPacked make(const void* a, const void* b, const void*, int i0, int, int K32_) {
        return LdPacked{reinterpret_cast<const bf16_t*>(a), K32_, reinterpret_cast<size_t>(b), i0};
    }
    template <int NB> struct Regs { ActFrag x[NB]; };
    __host__ __device__ int lds_bytes() const { return 0; }
    // Lanes of rows >= M read row M - 1 again (the same 16 bytes as that row's lane: no extra traffic) instead of being switched off: an
    // exec-masked load into a zero-initialised register made the compiler copy the loaded value inside the masked region — an
    // s_waitcnt vmcnt(0) in the MIDDLE of the launch's request batch, a full memory round trip (ISA, profiles/r04_ln_prologue.md).  What the
    // MFMA computes for those token columns is never stored, and a column of the product depends on its own token column only.
    template <int NB>
    __device__ __forceinline__ void issue(Regs<NB>& r, char*, int kt0, int lane, int = 0) const {
        const int ln = min(lane & 15, M - 1) + (lane & 48);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const bf16_t* p = X + ((size_t)(kt0 + u) * 64 + ln) * 8;
            r.x[u] = act_ld(p, plane);
        }
    }
    template <int NB> __device__ __forceinline__ void stats(Regs<NB>&, char*, int, int, bool, int, int = 0) const {}
    template <int NB> __device__ __forceinline__ void stage(const Regs<NB>&, char*) const {}
    template <int NB>
    __device__ __forceinline__ ActFrag frag(const Regs<NB>& r, const char*, int u, int, int) const { return r.x[u]; }
};

// ---- token operand: fp32 residual rows -> LayerNorm (or identity) -> hi/lo fragments, in registers -------------------
// GEMM row r reads source row r*row_mul + row_off (selects the last prompt row per stream on the first base pass).
// A wave owns the k-tiles of ITS K-slice: lane (r = lane & 15, g = lane >> 4) holds h[r][32 kt + 8 g .. + 8] for each of
// them — exactly the B fragment the MFMA wants once normalised.  Statistics (single-pass sum and sum of squares): per lane
// over its k-tiles in order, 4-group permlane-swap butterfly inside the wave, then the K-slices' partials are summed in
// slice order through LDS.  gamma / beta are staged once per block by LDS-DMA.  The token-tile path (k_ln_tiles) runs the
// same code with the same slicing: bit-identical operands.
template <bool NORM>           // NORM = false: identity (the Medusa heads read the post-LN rows as they are)
struct LdNormT {
    const float* h; const float* gamma; const float* beta;
    int d, K32, M, row_mul, row_off;
    static constexpr bool kNorm = true;
    static constexpr bool do_norm = NORM;
    // (M < 65536 rows, row_mul < 256, row_off < 128 — prompt chunks of <= 16 tokens, candidate trees of <= 64 nodes: the launchers check packs())
    LdArgs pack() const { return LdArgs{h, gamma, beta, d, M | (row_mul << 16) | (row_off << 24)}; }
    bool packs() const { return M >= 0 && M < 65536 && row_mul >= 0 && row_mul < 256 && row_off >= 0 && row_off < 128; }
    __device__ __forceinline__ static LdNormT make(const void* a, const void* b, const void* c, int i0, int i1, int K32_) {
        return LdNormT{reinterpret_cast<const float*>(a), reinterpret_cast<const float*>(b), reinterpret_cast<const float*>(c),
                       i0, K32_, i1 & 65535, (i1 >> 16) & 255, i1 >> 24};
    }
    // gb: the thread's share of gamma | beta on its way to LDS.  A block has >= 64 K32 / NB threads (one wave per K-slice of NB k-tiles, times
    // the row-tile groups), gamma | beta are d / 2 = 16 K32 float4: ceil(NB / 4) per thread always suffice.
    template <int NB> struct Regs { float4 v0[NB], v1[NB]; float mean, rstd; float4 gb[(NB + 3) / 4]; int ngb; int nthr; };
    __host__ __device__ int lds_bytes() const { return 2 * d * (int)sizeof(float) + 2048; }   // gamma, beta, statistics

    // Order of the requests: gamma / beta FIRST (ordinary loads into registers, written to LDS in stats()), the token rows after them; the
    // caller requests its weight stream AFTER this call.  The vector-memory counter retires in order: the statistics wait for the
    // token rows only, and run — like the normalisation — while the weights are still arriving.
    // Rounds 1-3 staged gamma / beta by LDS-DMA (global_load_lds) after the weights and the rows.  Two things followed (ISA and timeline:
    // profiles/r04_ln_prologue.md): (1) with an LDS-DMA pending next to ordinary loads the compiler's wait-count pass treats the counter as
    // out of order and turns EVERY wait of the kernel into s_waitcnt vmcnt(0) (cdna_hip_programming.md §5, trap (b): "beside glds it
    // waits vmcnt(0) for any ordinary VGPR-destination load"), so each LayerNorm-fused launch of the single-stream chain waited for its
    // complete weight batch before it computed a mean; (2) rows >= M were skipped with exec-masked loads into zero-initialised
    // registers, which made the compiler copy a loaded value inside the masked region: a full s_waitcnt in the MIDDLE of the request batch.
    // `rows`: false for waves that only help staging gamma / beta (the fused cross-attention kernel has more waves than K-slices)
    template <int NB>
    // nthr: the block's thread count when the caller has it in a register (k_skinny_gemm: from its preloaded plan word); 0 = blockDim.x, which
    // is a hidden kernel argument — a scalar load and its wait in front of the first request of the launch
    __device__ __forceinline__ void issue(Regs<NB>& r, char* smem, int kt0, int lane, int row0 = 0, bool rows = true, bool dma = true) const {
        issue_w<NB, false, false>(r, smem, kt0, lane, row0, rows, dma, 0, 15, 0);
    }
    template <int NB>          // the caller knows the block's thread count (a run-time test "nthr ? nthr : blockDim.x" keeps the hidden-argument load in the code)
    __device__ __forceinline__ void issue_n(Regs<NB>& r, char* smem, int kt0, int lane, int nthr) const {
        issue_w<NB, false, true>(r, smem, kt0, lane, 0, true, true, 0, 15, nthr);
    }
    // WIN: [rlo, rhi] = the rows of the tile this block works on (k_ln_tiles with row sub-blocks); lanes of other rows read the nearest row of
    // the window again — the same addresses as that row's lanes, no extra traffic, no exec-masked load — and their results are not stored
    template <int NB, bool WIN, bool NTHR>
    __device__ __forceinline__ void issue_w(Regs<NB>& r, char* smem, int kt0, int lane, int row0, bool rows, bool dma, int rlo, int rhi, int nthr_) const {
        const int rr = WIN ? min(max(lane & 15, rlo), rhi) : (lane & 15), g8 = (lane >> 4) * 8;
        // rows >= M: the last row again (LdPacked::issue: no exec-masked loads); `rows` is wave-uniform
        const float* hrow = h + (size_t)(min(row0 + rr, M - 1) * row_mul + row_off) * d + (size_t)kt0 * 32 + g8;
        r.ngb = dma ? 1 : 0;                                          // (a literal at every call site)
        if constexpr (NORM) {
            if (dma) {
                // always ceil(NB / 4) float4 per thread, indices clamped (a thread beyond the end re-reads the last float4 and re-writes
                // its LDS slot): no branch and no run-time count in the request batch — a conditional load here went through scratch
                const int nf4 = d >> 1, nthr = NTHR ? nthr_ : (int)blockDim.x;            // float4 of gamma | beta; d % 4 == 0
                r.nthr = nthr;
#pragma unroll
                for (int i = 0; i < (NB + 3) / 4; ++i) {
                    const int f = min((int)threadIdx.x + i * nthr, nf4 - 1);
                    r.gb[i] = reinterpret_cast<const float4*>(f < (d >> 2) ? gamma : beta)[f < (d >> 2) ? f : f - (d >> 2)];
                }
            }
        }
        if (rows) {
#pragma unroll
            for (int u = 0; u < NB; ++u) { const float4* src = reinterpret_cast<const float4*>(hrow + u * 32); r.v0[u] = src[0]; r.v1[u] = src[1]; }
        } else {
#pragma unroll
            for (int u = 0; u < NB; ++u) { r.v0[u] = make_float4(0.f, 0.f, 0.f, 0.f); r.v1[u] = r.v0[u]; }
        }
    }
    // gamma | beta registers -> LDS.  Call it right behind the LAST request of the launch's batch: the parameter loads are the oldest
    // entries of the in-order vector-memory counter, so the wait in front of these LDS writes (vmcnt(number of younger requests)) leaves
    // rows and weights in flight and costs an L2 round trip that the weight stream's latency covers anyway.  (Left to stats(), the
    // scheduler sank one of the parameter loads below the weight stream to save registers — and the LDS write then waited for the
    // YOUNGEST request of the batch, i.e. for everything.)
    template <int NB>
    __device__ __forceinline__ void stage(const Regs<NB>& r, char* smem) const {
        if constexpr (NORM) {
            if (r.ngb) {
                const int nf4 = d >> 1, nthr = r.nthr;
#pragma unroll
                for (int i = 0; i < (NB + 3) / 4; ++i) reinterpret_cast<float4*>(smem)[min((int)threadIdx.x + i * nthr, nf4 - 1)] = r.gb[i];
            }
        }
    }
    // per-row statistics over the whole row: needs every K-slice -> LDS exchange (contains the block barrier, which also
    // makes the staged gamma / beta visible).  `leader`: this wave publishes its slice's partial (one wave per slice).
    // `slot`: which half of the 2 KiB statistics area is used (the two-tile kernel normalises its tiles one after the other: the
    // second tile's partials must not overwrite the first's while a slower wave still reads them; slots hold <= 8 K-slices)
    template <int NB>
    __device__ __forceinline__ void stats(Regs<NB>& r, char* smem, int ks, int ksplit, bool leader, int lane, int slot = 0) const {
        r.mean = 0.f; r.rstd = 1.f;
        if constexpr (!NORM) return;
        float2* part = reinterpret_cast<float2*>(smem + (size_t)2 * d * sizeof(float)) + slot * 128;      // [ksplit <= 16][16 rows]
        float s = 0.f, q2 = 0.f;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const float4 a = r.v0[u], b = r.v1[u];
            s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
            q2 += ((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w));
        }
        s = rows4_sum(s); q2 = rows4_sum(q2);
        if (leader && lane < 16) part[ks * 16 + lane] = make_float2(s, q2);
        __syncthreads();
        float ts = 0.f, tq = 0.f;
        for (int k2 = 0; k2 < ksplit; ++k2) { const float2 p = part[k2 * 16 + (lane & 15)]; ts += p.x; tq += p.y; }
        r.mean = ts / (float)d;
        r.rstd = rsqrtf(fmaxf(tq / (float)d - r.mean * r.mean, 0.f) + 1e-5f);
    }
    template <int NB>
    __device__ __forceinline__ ActFrag frag(const Regs<NB>& r, const char* smem, int u, int kt, int lane) const {
        float4 y0 = r.v0[u], y1 = r.v1[u];
        if constexpr (NORM) {
            const float* gb = reinterpret_cast<const float*>(smem) + kt * 32 + (lane >> 4) * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(gb), g1 = *reinterpret_cast<const float4*>(gb + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(gb + d), b1 = *reinterpret_cast<const float4*>(gb + d + 4);
            const float m = r.mean, rs = r.rstd;
            y0.x = __builtin_fmaf((y0.x - m) * rs, g0.x, b0.x); y0.y = __builtin_fmaf((y0.y - m) * rs, g0.y, b0.y);
            y0.z = __builtin_fmaf((y0.z - m) * rs, g0.z, b0.z); y0.w = __builtin_fmaf((y0.w - m) * rs, g0.w, b0.w);
            y1.x = __builtin_fmaf((y1.x - m) * rs, g1.x, b1.x); y1.y = __builtin_fmaf((y1.y - m) * rs, g1.y, b1.y);
            y1.z = __builtin_fmaf((y1.z - m) * rs, g1.z, b1.z); y1.w = __builtin_fmaf((y1.w - m) * rs, g1.w, b1.w);
        }
        return act_split8(y0, y1);
    }
    // (the fused cross-attention query of wm_decoder.hip — hi / lo builds only — stages the two planes through LDS itself)
    template <int NB>
    __device__ __forceinline__ void frag_hilo(const Regs<NB>& r, const char* smem, int u, int kt, int lane, bf16x8_t& bh, bf16x8_t& bl) const {
        float4 y0 = r.v0[u], y1 = r.v1[u];
        if constexpr (NORM) {
            const float* gb = reinterpret_cast<const float*>(smem) + kt * 32 + (lane >> 4) * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(gb), g1 = *reinterpret_cast<const float4*>(gb + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(gb + d), b1 = *reinterpret_cast<const float4*>(gb + d + 4);
            const float m = r.mean, rs = r.rstd;
            y0.x = __builtin_fmaf((y0.x - m) * rs, g0.x, b0.x); y0.y = __builtin_fmaf((y0.y - m) * rs, g0.y, b0.y);
            y0.z = __builtin_fmaf((y0.z - m) * rs, g0.z, b0.z); y0.w = __builtin_fmaf((y0.w - m) * rs, g0.w, b0.w);
            y1.x = __builtin_fmaf((y1.x - m) * rs, g1.x, b1.x); y1.y = __builtin_fmaf((y1.y - m) * rs, g1.y, b1.y);
            y1.z = __builtin_fmaf((y1.z - m) * rs, g1.z, b1.z); y1.w = __builtin_fmaf((y1.w - m) * rs, g1.w, b1.w);
        }
        split_hilo8(y0, y1, bh, bl);
    }
};
typedef LdNormT<true> LdNorm;
typedef LdNormT<false> LdIdent;

// NK = fragments of a wave's K-slice; ksplit * NK == K32.  XB = token-operand fragments held at a time (NK, or NK / 2
// when the slice is long: the second half is issued as soon as the first has been consumed).  RT = weight row tiles per
// wave (register blocking: the token fragment — and, for LdNorm, its normalisation — is shared by RT tiles; used where one
// tile per block would put more blocks than CUs on the chip).  RT > 1 needs ksplit >= RT.
// FOLD (round 6): the operand is gamma o x written by the producing launch, the block sums the rows' statistics partials and the accumulator
// becomes rstd (acc - mean c) before the epilogue (wm_common.h "LayerNorm folded into the GEMM it feeds"); Ld = LdPacked.
template <int NK, int RT, bool W8, class Ld, class Ep, bool FOLD = false>
__global__ void __launch_bounds__(640)
k_skinny_gemm(const bf16_t* __restrict__ W, const void* __restrict__ la, const void* __restrict__ lb, const void* __restrict__ lc, int li0, int li1,
              int N16, int K32, int plan, int nmain, const float* __restrict__ wscale, const int* __restrict__ done, Ep ep, PfJob pf, FoldIn fold TL_ARG)
{
    // W .. nmain are 14 dwords: exactly what the kernarg preload delivers in SGPRs with the wave (-amdgpu-kernarg-preload-count=16 = the kernarg
    // pointer + 14 dwords); wscale, done and the structs behind them are s_loaded at the top of the main path and consumed after the request batch.
    // plan = ksplit | rt_per_wg << 8 | ks_magic << 16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= nmain) { pf_block(pf, (int)blockIdx.x - pf_round8(nmain)); return; }      // prefetch-only blocks
    const Ld ld = Ld::make(la, lb, lc, li0, li1, K32);
    const int ksplit = plan & 255, rt_per_wg = (plan >> 8) & 255, ks_magic = plan >> 16;
    TL_BEGIN
    constexpr int XB = (NK > 8 && !Ld::kNorm) ? NK / 2 : NK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rtl = (wave * ks_magic) >> 8, ks = wave - rtl * ksplit;          // wave / ksplit, wave % ksplit (waves <= 16)
    const int tile0 = (blockIdx.x * rt_per_wg + rtl) * RT;                      // first of this wave's RT row tiles
    const int kt0 = ks * NK;

    // ---- the launch's memory batch: weights, token operand, LayerNorm parameters, epilogue operands ----
    // LayerNorm-fused loaders: parameters and token rows are requested BEFORE the weight stream (LdNormT::issue), so that the statistics do
    // not wait for the weights; plain loaders: weights first (their latency is the long one and the MFMAs need both anyway).
    typename WRaw<W8>::type a[RT][NK];
    typename Ld::template Regs<XB> xr;
    if constexpr (Ld::kNorm) {
        ld.template issue_n<XB>(xr, smem, kt0, lane, 64 * ksplit * rt_per_wg);
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const size_t wp = ((size_t)min(tile0 + i, N16 - 1) * K32 + kt0) * 512 + lane * 8;   // element index (bf16: 2 B, fp8: 1 B per element)
#pragma unroll
        for (int u = 0; u < NK; ++u) a[i][u] = ld_wraw<W8, true>(W, wp + (size_t)u * 512);
    }
    if constexpr (!Ld::kNorm) ld.template issue<XB>(xr, smem, kt0, lane);
    EpPre pre; pre.i = 0; pre.a = make_float4(0.f, 0.f, 0.f, 0.f); pre.b = pre.a;
    float4 wsc = make_float4(1.f, 1.f, 1.f, 1.f), fc4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // the element this thread will finish: with K-slices, wave f < rt_per_wg * RT finishes the block's f-th row tile; without,
    // every wave finishes its own tile (RT == 1 there)
    const int tf = (ksplit > 1) ? blockIdx.x * rt_per_wg * RT + wave : tile0;
    const int em = lane & 15, en = tf * 16 + 4 * (lane >> 4);
    const bool edo = ((ksplit > 1) ? (wave < rt_per_wg * RT) : true) && tf < N16;
    // (the epilogue's fields are the only kernel arguments of the batch that are not preloaded: their s_load goes out at the top of the main path.
    // A sched_barrier here — to keep their s_waitcnt out of the batch — sinks that s_load below the barrier as well, the epilogue operands are
    // then requested a scalar round trip later, and since the LAST request's arrival ends a launch of this chain, the iteration took 2.64 instead
    // of 2.52 ms: measured, profiles/r05_kernarg_preload.md)
    {   // every wave requests them (a wave that finishes nothing: the last tile's; a few hundred bytes): no branch in the request batch
        const int enc = min(tf, N16 - 1) * 16 + 4 * (lane >> 4);
        pre = ep.pre(em, enc);
        if constexpr (W8) wsc = *reinterpret_cast<const float4*>(wscale + enc);
        if constexpr (FOLD) fc4 = *reinterpret_cast<const float4*>(fold.c + enc);
    }
    // FOLD: the thread's share of the 16 rows' statistics partials (group f_grp of row f_row; threads beyond 16 G repeat the last share)
    FoldPart fpl; int f_slot = 0;
    if constexpr (FOLD) {
        const int idx = min((int)threadIdx.x, 2 * fold.T16 - 1);            // 16 rows x G = T16 / 8 groups
        f_slot = idx;
        fold_part_load(fpl, fold, min(idx & 15, li0 - 1), idx >> 4);        // li0 = M (LdPacked): rows >= M take the last row's (never stored)
    }
    ld.template stage<XB>(xr, smem);
    // done: every stream finished (the rest of this replay is a no-op).  Checked after the batch went out: the flag's
    // latency overlaps the stream instead of heading every launch of the dependent chain.
    if (done && *done) return;

    ld.template stats<XB>(xr, smem, ks, ksplit, rtl == 0, lane);
    TL_PREP

    f32x4_t acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < XB; ++u) {
        const ActFrag xf = ld.template frag<XB>(xr, smem, u, kt0 + u, lane);
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i] = mfma_act(w_expand<W8>(a[i][u]), xf, acc[i]);
    }
    if constexpr (XB < NK) {
        ld.template issue<XB>(xr, smem, kt0 + XB, lane);
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const ActFrag xf = ld.template frag<XB>(xr, smem, u, kt0 + XB + u, lane);
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[i] = mfma_act(w_expand<W8>(a[i][XB + u]), xf, acc[i]);
        }
    }
    TL_MID

    // FOLD: group partials of the rows' statistics -> LDS (behind the K-slice partials), visible after the block barrier below
    const float2* fpart = reinterpret_cast<const float2*>(smem + ld.lds_bytes() + (ksplit > 1 ? (size_t)rt_per_wg * RT * ksplit * 1024 : 0));
    if constexpr (FOLD) const_cast<float2*>(fpart)[f_slot] = fold_part_sum(fpl);
    if (ksplit > 1) {
        float4* red = reinterpret_cast<float4*>(smem + ld.lds_bytes());
#pragma unroll
        for (int i = 0; i < RT; ++i)
            red[((rtl * RT + i) * ksplit + ks) * 64 + lane] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        __syncthreads();
        if (edo) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
            for (int k2 = 0; k2 < ksplit; ++k2) {
                const float4 p = red[(wave * ksplit + k2) * 64 + lane];
                s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
            }
            if constexpr (W8) s = scale4v(s, wsc);
            if constexpr (FOLD) s = fold_apply(s, fold_row_stat(fpart, em, 16, fold.T16 >> 3, fold.inv_d), fc4);
            ep.fin(em, en, s, pre);
        }
    } else {
        if constexpr (FOLD) __syncthreads();
        if (edo) {
            if constexpr (W8) acc[0] = scale4v(acc[0], wsc);
            if constexpr (FOLD) acc[0] = fold_apply(acc[0], fold_row_stat(fpart, em, 16, fold.T16 >> 3, fold.inv_d), fc4);
            ep.fin(em, en, acc[0], pre);
        }
    }
    TL_END
}

// ---- batched path (token rows R > 16) ----------------------------------------------------------
// LayerNorm of all row tiles to global packed hi/lo planes: the SAME code as the fused loader with the same K-slicing
// (one wave per slice), so the operand bits equal those of a 16-row launch.
// Row sub-blocks (round 5): a 16-row tile is worked on by `sub` blocks of 16 / sub rows each (nmain = tiles x sub).  The launch has 2..22 tiles
// and part of its time is what ONE CU needs to pull a tile's 80 KB of fp32 rows in and push 80 KB of hi / lo fragments out (~55 GB/s per CU: 1.5 us
// each way); a row's statistics and normalisation involve no other row, so half tiles on two CUs move half the bytes each.  Every wave still
// runs the full-tile instruction stream — lanes outside the block's row window re-read a row of the window and do not store —, so the operand
// bits are those of the whole-tile form.  Measured at 32 streams (profiles/r05_ln_row_subblocks.md): 1 / 2 / 4 / 8 blocks per tile = 7.87 / 7.72 /
// 7.74 / 7.91 ms per Medusa iteration, vanilla step 3.70 -> 3.63 ms; the launch stays a ~5 us latency structure, so 2 is the default.
template <int NK, class Ld>
__global__ void __launch_bounds__(640)
k_ln_tiles(const void* __restrict__ la, const void* __restrict__ lb, const void* __restrict__ lc, int li0, int li1, int K32, int ks_sub, int nmain,
           const int* __restrict__ ntiles, bf16_t* __restrict__ xg, size_t plane, const int* __restrict__ done, PfJob pf, int pf_sliced)
{
    // la .. ntiles are 13 dwords, preloaded with the wave (LdArgs; as a struct in front, the loader kept EVERY argument of this launch out of the
    // preload: two scalar round trips — arguments, then *ntiles — in front of the first request); ks_sub = ksplit | sub << 8
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Ld ld = Ld::make(la, lb, lc, li0, li1, K32);
    const int ksplit = ks_sub & 255, sub = ks_sub >> 8;
    const int tile = (int)blockIdx.x / sub;
    // blocks beyond the token tiles: the launch has 2..22 blocks of work — the rest of the chip pulls the weight matrix of the GEMM that
    // follows towards the CUs that will read it (per consumer block / XCD for the two-tile kernel, whose block j runs on XCD j % 8; in
    // eighths for the token-tile kernels, where every XCD ends up reading the whole matrix and the point is the Infinity Cache)
    if ((int)blockIdx.x >= nmain) {
        if (!(done && *done)) {
            const int job = (int)blockIdx.x - pf_round8(nmain);
            if (pf_sliced) pf_block_sliced(pf, job); else pf_block(pf, job);
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kt0 = ks * NK;
    typename Ld::template Regs<NK> xr;
    const int nrow = 16 / sub, rlo = ((int)blockIdx.x - tile * sub) * nrow, rhi = rlo + nrow - 1;
    ld.template issue_w<NK, true, true>(xr, smem, kt0, lane, tile * 16, true, true, rlo, rhi, 64 * ksplit);
    ld.template stage<NK>(xr, smem);
    // `done` (every stream finished) and the step's tile count (merged-step schedule: no rows in this tile in this step) only decide whether
    // the block STORES: as tests in front of the requests they were dependent scalar round trips heading every launch of the chain, and as an
    // early exit behind the requests the compiler sank the row loads below the exit's branch (ISA) — the same round trips again.  A block
    // without rows normalises whatever lies in its tile of the scratch (sized for the full pass) and stores nothing.
    const int* dp = done ? done : reinterpret_cast<const int*>(la);
    const int* np = ntiles ? ntiles : reinterpret_cast<const int*>(la);
    int dv = *dp, nv = *np;
    asm volatile("" : "+s"(dv), "+s"(nv));           // requested HERE, behind the row requests and under their latency (left alone, the two s_loads sink to the stores)
    ld.template stats<NK>(xr, smem, ks, ksplit, true, lane);
    bf16_t* dst = xg + (size_t)tile * ld.K32 * 512;
    const bool live = !((done && dv) || (ntiles && tile >= nv));
    const bool mine = live && (lane & 15) >= rlo && (lane & 15) <= rhi;
#pragma unroll
    for (int u = 0; u < NK; ++u) {
        const ActFrag f = ld.template frag<NK>(xr, smem, u, kt0 + u, lane);
        const size_t o = ((size_t)(kt0 + u) * 64 + lane) * 8;
        if (mine) act_st_frag(dst + o, plane, f);
    }
}

// More than 16 token rows (several streams): a wave owns RT weight row tiles x TT token tiles x ONE K-slice,
// so every weight fragment feeds 2*TT MFMAs and every token fragment RT of them (register blocking cuts the L2
// traffic of the operand re-reads by RT resp. TT); blockIdx.y walks the token-tile groups, so the chip is filled
// by tokens as well as by features and the weights are re-read from L2 / Infinity Cache, not HBM.  Per output the
// accumulation order is the 16-row kernel's (k ascending, hi then lo; K-slices summed in order): bit-identical.
// (Round 4's measured-negative variants of this kernel — four token tiles per wave, an explicit two-deep request pipeline, two feature
// groups per block sharing token fragments through the L1, a 96-register budget — are kept as tests/microbench/r04_rows_variants.patch;
// round 5's LDS-shared token tiles with register-direct weights (k_rows_lds) and the LayerNorm in the producing GEMM's tail with a
// write-through hand-off as tests/microbench/r05_rows_lds_ln_tail.patch: measured, profiles/r05_rows_lds.md.)
template <int NKR, int RT, int TT, bool W8, class Ep, bool FOLD = false>
__global__ void __launch_bounds__(640)
k_rows_gemm(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X, size_t plane, const int* __restrict__ done, const int* __restrict__ ntiles,
            int N16, int K32, int ksplit, int MT, const float* __restrict__ wscale, Ep ep, FoldIn fold, PfJob pf, int pf_sliced TL_ARG)
{
    // W .. MT are 14 dwords: preloaded (ntiles used to follow the epilogue struct: its pointer, then *ntiles — two scalar round trips at entry)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {   // extra rows of blocks behind the token-tile groups: prefetch-only (the next GEMM's weights towards the Infinity Cache / this XCD's L2)
        const int ny = (MT + TT - 1) / TT;
        if ((int)blockIdx.y >= ny) {
            if (!(done && *done)) {
                const int job = ((int)blockIdx.y - ny) * (int)gridDim.x + (int)blockIdx.x;
                if (pf_sliced) pf_block_sliced(pf, job); else pf_block(pf, job);
            }
            return;
        }
    }
    TL_BEGIN
    // (checked first: moving the flag behind the first group of loads — a mid-loop exit — cost the 352-row launches ~3 us each, the
    //  compiler no longer overlapped the load groups across it: tests/microbench/r03_call4.sh)
    {   // both flags are requested before either is waited for (address select instead of a null test: a branch would serialise the two scalar loads)
        const int* dp = done ? done : reinterpret_cast<const int*>(W);
        const int* np = ntiles ? ntiles : reinterpret_cast<const int*>(W);
        int dv = *dp, nv = *np;
        asm volatile("" : "+s"(dv), "+s"(nv));            // both values exist here: the two s_loads go out together, one wait
        if ((done && dv) || (ntiles && (int)blockIdx.y * TT >= nv)) return;          // merged-step schedule: no rows in this token-tile group in this step
    }
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
    // FOLD: the thread's share of the TT x 16 rows' statistics partials (wm_common.h) is requested ahead of the operand stream and summed into LDS
    // as soon as the first operand group is on the wire: the 16 registers are free again before the MFMAs (held across the K loop they pushed the
    // QKV instance from 127 to 144 registers: three waves per SIMD instead of four, two resident blocks instead of three)
    const float2* fpart = reinterpret_cast<const float2*>(smem + (ksplit > 1 ? (size_t)RT * TT * ksplit * 1024 : 0));
    FoldPart fpl; int f_slot = 0;
    if constexpr (FOLD) {
        const int idx = min((int)threadIdx.x, TT * 2 * fold.T16 - 1);          // TT x 16 rows x G = T16 / 8 groups
        f_slot = idx;
        const int r = idx % (TT * 16);
        fold_part_load(fpl, fold, min(mt0 + (r >> 4), MT - 1) * 16 + (r & 15), idx / (TT * 16));
    }
    constexpr int G = 4;          // k-tiles per load group
#pragma unroll
    for (int kg = 0; kg < NKR; kg += G) {
        wfrag_t a[RT][G]; ActFrag x[TT][G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
#pragma unroll
            for (int j = 0; j < TT; ++j) x[j][u] = act_ld(xp[j] + (size_t)(kg + u) * 512, plane);
#pragma unroll
            for (int i = 0; i < RT; ++i) a[i][u] = ld_wfrag<W8, false>(W, wp[i] + (size_t)(kg + u) * 512);
        }
        if constexpr (FOLD) {
            if (kg == 0) const_cast<float2*>(fpart)[f_slot] = fold_part_sum(fpl);      // (the partial loads are the oldest requests: their wait leaves the operand group in flight)
        }

#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < TT; ++j) acc[i][j] = mfma_act(a[i][u], x[j][u], acc[i][j]);
    }
    TL_PREP          // timeline build: the wave's MFMAs have issued
    if (ksplit > 1) {
        float4* red = reinterpret_cast<float4*>(smem);
        const int e0 = threadIdx.x, estep = blockDim.x;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
                red[((i * TT + j) * ksplit + ks) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        __syncthreads();
        TL_MID       // timeline build: K-slice partials exchanged
        for (int e = e0; e < RT * TT * 64; e += estep) {
            const int t = e >> 6, l2 = e & 63, i = t / TT, j = t - i * TT;
            f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
            for (int k2 = 0; k2 < ksplit; ++k2) {
                const float4 p = red[(t * ksplit + k2) * 64 + l2];
                sacc[0] += p.x; sacc[1] += p.y; sacc[2] += p.z; sacc[3] += p.w;
            }
            if (rt0 + i < N16 && mt0 + j < MT) {
                if constexpr (W8) sacc = scale4(sacc, wscale, (rt0 + i) * 16 + 4 * (l2 >> 4));
                if constexpr (FOLD)
                    sacc = fold_apply(sacc, fold_row_stat(fpart, j * 16 + (l2 & 15), TT * 16, fold.T16 >> 3, fold.inv_d),
                                      *reinterpret_cast<const float4*>(fold.c + (rt0 + i) * 16 + 4 * (l2 >> 4)));
                ep.store4((mt0 + j) * 16 + (l2 & 15), (rt0 + i) * 16 + 4 * (l2 >> 4), sacc);
            }
        }
    } else if constexpr (FOLD) {
        // (one K-slice per block: d_model <= 256 only — the folded GEMMs keep <= 8 fragments per slice.  Tile by tile: the batched operand form
        //  below would set the register budget of the instances every real model runs through the K-slice path above)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
                if (rt0 + i < N16 && mt0 + j < MT) {
                    if constexpr (W8) acc[i][j] = scale4(acc[i][j], wscale, (rt0 + i) * 16 + 4 * (lane >> 4));
                    acc[i][j] = fold_apply(acc[i][j], fold_row_stat(fpart, j * 16 + (lane & 15), TT * 16, fold.T16 >> 3, fold.inv_d),
                                           *reinterpret_cast<const float4*>(fold.c + (rt0 + i) * 16 + 4 * (lane >> 4)));
                    ep.store4((mt0 + j) * 16 + (lane & 15), (rt0 + i) * 16 + 4 * (lane >> 4), acc[i][j]);
                }
    } else {
        // operands of every tile's epilogue in one batch, then the stores (tile-by-tile store4 calls are dependent memory round trips:
        // loads cannot be hoisted over the previous tile's stores)
        EpPre pre[RT][TT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                pre[i][j].i = 0; pre[i][j].a = make_float4(0.f, 0.f, 0.f, 0.f); pre[i][j].b = pre[i][j].a;
                if (rt0 + i < N16 && mt0 + j < MT) pre[i][j] = ep.pre((mt0 + j) * 16 + (lane & 15), (rt0 + i) * 16 + 4 * (lane >> 4));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
                if (rt0 + i < N16 && mt0 + j < MT) {
                    if constexpr (W8) acc[i][j] = scale4(acc[i][j], wscale, (rt0 + i) * 16 + 4 * (lane >> 4));
                    ep.fin((mt0 + j) * 16 + (lane & 15), (rt0 + i) * 16 + 4 * (lane >> 4), acc[i][j], pre[i][j]);
                }
    }
    TL_END
}

// ---- 17..32 token rows (two token tiles): the weight-streaming kernel with a second token tile ---------------------
// The base pass of up to 32 streams (and every vanilla step) is still a weight-streaming problem: 3-13 MB of weights against
// <= 64 KB of token operand.  Same organisation as k_skinny_gemm — the wave's whole K-slice of the weight stream, both token
// tiles' hi/lo fragments and the epilogue operands are requested in one batch at entry — with every weight fragment feeding
// four MFMAs.  (The register-blocked kernel walked its K-slice in dependent groups of 4 fragments: FC2 at 32 rows took 15 us
// against 7 us for the single-tile launch.)  Same plan, same accumulation order: bit-identical to single-stream runs.
template <int NK, bool W8, class Ep, bool FOLD = false>
__global__ void __launch_bounds__(640)
k_skinny2_gemm(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X, size_t plane, const int* __restrict__ done, int N16, int K32, int plan, int M,
               const float* __restrict__ wscale, Ep ep, FoldIn fold, PfJob pf)
{
    // W .. M are 12 dwords: preloaded (plane and M used to lie behind the 14th dword: the token requests waited for a scalar load);
    // plan = ksplit | rt_per_wg << 8 | ks_magic << 16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ksplit = plan & 255, rt_per_wg = (plan >> 8) & 255, ks_magic = plan >> 16;
    {   // blocks behind the row-tile groups: prefetch-only (one job per block of the NEXT two-tile GEMM, same XCD by construction)
        const int nmain = (N16 + rt_per_wg - 1) / rt_per_wg;
        if ((int)blockIdx.x >= nmain) { if (!(done && *done)) pf_block(pf, (int)blockIdx.x - pf_round8(nmain)); return; }
    }
    // token fragments held at a time: 4 k-tiles x 2 token tiles x hi/lo = 64 registers next to the wave's NK weight fragments (a block
    // of 10 waves leaves 168 registers per lane); the next round is requested as soon as the MFMAs of this one have issued
    constexpr int XB = 4;
    static_assert(NK % XB == 0, "K-slice is a multiple of 4 k-tiles");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rtl = (wave * ks_magic) >> 8, ks = wave - rtl * ksplit;
    const int tile0 = blockIdx.x * rt_per_wg + rtl;
    const int kt0 = ks * NK;
    typename WRaw<W8>::type a[NK];
    {
        const size_t wp = ((size_t)min(tile0, N16 - 1) * K32 + kt0) * 512 + lane * 8;
#pragma unroll
        for (int u = 0; u < NK; ++u) a[u] = ld_wraw<W8, true>(W, wp + (size_t)u * 512);
    }
    // rows >= M of the second tile read its last valid row again (16 < M <= 32; LdPacked::issue: no exec-masked loads)
    const int ln1 = min(lane & 15, M - 17) + (lane & 48);
    ActFrag x0[XB], x1[XB];
    auto issue = [&](int kt) {
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const bf16_t* p0 = X + ((size_t)(kt + u) * 64 + lane) * 8;
            const bf16_t* p1 = X + ((size_t)K32 * 64 + (size_t)(kt + u) * 64 + ln1) * 8;
            x0[u] = act_ld(p0, plane);
            x1[u] = act_ld(p1, plane);
        }
    };
    issue(kt0);
    // the element this thread finishes: with K-slices, wave f < 2 * rt_per_wg finishes (row tile f >> 1, token tile f & 1) of the block
    const int nfin = 2 * rt_per_wg;
    const int tf = (ksplit > 1) ? blockIdx.x * rt_per_wg + (wave >> 1) : tile0;
    const int en = tf * 16 + 4 * (lane >> 4);
    const bool edo = ((ksplit > 1) ? (wave < nfin) : true) && tf < N16;
    EpPre pre0, pre1; pre0.i = 0; pre0.a = make_float4(0.f, 0.f, 0.f, 0.f); pre0.b = pre0.a; pre1 = pre0;
    float4 wsc = make_float4(1.f, 1.f, 1.f, 1.f), fc4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (edo) {
        if (ksplit > 1) pre0 = ep.pre((wave & 1) * 16 + (lane & 15), en);
        else { pre0 = ep.pre(lane & 15, en); pre1 = ep.pre(16 + (lane & 15), en); }
        if constexpr (W8) wsc = *reinterpret_cast<const float4*>(wscale + en);
        if constexpr (FOLD) fc4 = *reinterpret_cast<const float4*>(fold.c + en);
    }
    // FOLD: the thread's share of the 32 rows' statistics partials (wm_common.h)
    FoldPart fpl; int f_slot = 0;
    if constexpr (FOLD) {
        const int idx = min((int)threadIdx.x, 4 * fold.T16 - 1);              // 32 rows x G = T16 / 8 groups
        f_slot = idx;
        fold_part_load(fpl, fold, min(idx & 31, M - 1), idx >> 5);
    }
    if (done && *done) return;

    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NK / XB; ++r) {
        if (r > 0) issue(kt0 + r * XB);
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const wfrag_t av = w_expand<W8>(a[r * XB + u]);
            acc0 = mfma_act(av, x0[u], acc0);
            acc1 = mfma_act(av, x1[u], acc1);
        }
    }
    const float2* fpart = reinterpret_cast<const float2*>(smem + (ksplit > 1 ? (size_t)rt_per_wg * 2 * ksplit * 1024 : 0));
    if constexpr (FOLD) const_cast<float2*>(fpart)[f_slot] = fold_part_sum(fpl);
    if (ksplit > 1) {
        float4* red = reinterpret_cast<float4*>(smem);
        red[((rtl * 2 + 0) * ksplit + ks) * 64 + lane] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        red[((rtl * 2 + 1) * ksplit + ks) * 64 + lane] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        __syncthreads();
        if (edo) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
            for (int k2 = 0; k2 < ksplit; ++k2) {
                const float4 p = red[(wave * ksplit + k2) * 64 + lane];
                s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
            }
            if constexpr (W8) s = scale4v(s, wsc);
            if constexpr (FOLD) s = fold_apply(s, fold_row_stat(fpart, (wave & 1) * 16 + (lane & 15), 32, fold.T16 >> 3, fold.inv_d), fc4);
            ep.fin((wave & 1) * 16 + (lane & 15), en, s, pre0);
        }
    } else {
        if constexpr (FOLD) __syncthreads();
        if (edo) {
            if constexpr (W8) {
                acc0 = scale4v(acc0, wsc);
                acc1 = scale4v(acc1, wsc);
            }
            if constexpr (FOLD) {
                acc0 = fold_apply(acc0, fold_row_stat(fpart, lane & 15, 32, fold.T16 >> 3, fold.inv_d), fc4);
                acc1 = fold_apply(acc1, fold_row_stat(fpart, 16 + (lane & 15), 32, fold.T16 >> 3, fold.inv_d), fc4);
            }
            ep.fin(lane & 15, en, acc0, pre0);
            ep.fin(16 + (lane & 15), en, acc1, pre1);
        }
    }
}

// The LayerNorm-fused form for two token tiles: a wave normalises its K-slice of tile 0 and of tile 1 one after the other with the
// single-tile kernel's code (same statistics, same slicing: bit-identical operands) against the same weight registers — the three
// k_ln_tiles launches per decoder layer of a 17..32-row pass (~7 us each, two blocks of work) disappear.  Blocks of <= 6 waves
// (2 per SIMD: 256 registers per lane hold both tiles' fp32 rows).
template <int NK, bool W8, class Ld, class Ep>
__global__ void __launch_bounds__(384)
k_skinny2_norm(const bf16_t* __restrict__ W, const float* __restrict__ wscale, int N16, int K32, int ksplit, int rt_per_wg, int ks_magic,
               const int* __restrict__ done, Ld ld, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rtl = (wave * ks_magic) >> 8, ks = wave - rtl * ksplit;
    const int tile0 = blockIdx.x * rt_per_wg + rtl;
    const int kt0 = ks * NK;
    typename WRaw<W8>::type a[NK];
    {
        const size_t wp = ((size_t)min(tile0, N16 - 1) * K32 + kt0) * 512 + lane * 8;
#pragma unroll
        for (int u = 0; u < NK; ++u) a[u] = ld_wraw<W8, true>(W, wp + (size_t)u * 512);
    }
    typename Ld::template Regs<NK> x0, x1;
    ld.template issue<NK>(x0, smem, kt0, lane, 0);
    ld.template issue<NK>(x1, smem, kt0, lane, 16, true, false);
    const int nfin = 2 * rt_per_wg;
    const int tf = (ksplit > 1) ? blockIdx.x * rt_per_wg + (wave >> 1) : tile0;
    const int en = tf * 16 + 4 * (lane >> 4);
    const bool edo = ((ksplit > 1) ? (wave < nfin) : true) && tf < N16;
    EpPre pre0, pre1; pre0.i = 0; pre0.a = make_float4(0.f, 0.f, 0.f, 0.f); pre0.b = pre0.a; pre1 = pre0;
    float4 wsc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (edo) {
        if (ksplit > 1) pre0 = ep.pre((wave & 1) * 16 + (lane & 15), en);
        else { pre0 = ep.pre(lane & 15, en); pre1 = ep.pre(16 + (lane & 15), en); }
        if constexpr (W8) wsc = *reinterpret_cast<const float4*>(wscale + en);
    }
    ld.template stage<NK>(x0, smem);
    if (done && *done) return;

    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    ld.template stats<NK>(x0, smem, ks, ksplit, rtl == 0, lane, 0);
#pragma unroll
    for (int u = 0; u < NK; ++u) acc0 = mfma_act(w_expand<W8>(a[u]), ld.template frag<NK>(x0, smem, u, kt0 + u, lane), acc0);
    ld.template stats<NK>(x1, smem, ks, ksplit, rtl == 0, lane, 1);
#pragma unroll
    for (int u = 0; u < NK; ++u) acc1 = mfma_act(w_expand<W8>(a[u]), ld.template frag<NK>(x1, smem, u, kt0 + u, lane), acc1);
    if (ksplit > 1) {
        float4* red = reinterpret_cast<float4*>(smem + ld.lds_bytes());
        red[((rtl * 2 + 0) * ksplit + ks) * 64 + lane] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        red[((rtl * 2 + 1) * ksplit + ks) * 64 + lane] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
        __syncthreads();
        if (edo) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
            for (int k2 = 0; k2 < ksplit; ++k2) {
                const float4 p = red[(wave * ksplit + k2) * 64 + lane];
                s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
            }
            if constexpr (W8) s = scale4v(s, wsc);
            ep.fin((wave & 1) * 16 + (lane & 15), en, s, pre0);
        }
    } else if (edo) {
        if constexpr (W8) {
            acc0 = scale4v(acc0, wsc);
            acc1 = scale4v(acc1, wsc);
        }
        ep.fin(lane & 15, en, acc0, pre0);
        ep.fin(16 + (lane & 15), en, acc1, pre1);
    }
}

// ---- token-tile GEMM through an LDS ring (R >= 48 token rows: the verify pass of several streams) -----------------
// At 352 rows (32 streams x 11 candidates) a decoder GEMM is ~20 output tiles per CU: neither weight-streaming (the
// register-blocked kernel above re-reads every operand from L2 once per wave: 8 KB per 16 MFMAs, L2 -> CU bound) nor big
// enough for the encoder's 256 x 256 tiles (22 token tiles).  Here a block of 8 waves owns 4F weight row tiles x TT token
// tiles and walks ALL of K: every operand byte crosses L2 -> CU once per block, through a ring of LDS stages filled by
// LDS-DMA (global_load_lds, 1 KiB packed fragments, lane-linear: conflict-free ds_read_b128), R - 1 stages in flight.
//   wave (w = wave & 3, hh = wave >> 2): row tiles w*F .. +F of the block, token tiles hh*TH .. +TH (TH = TT / 2): two
//   waves per SIMD, so one wave's fragment reads hide under the other's MFMAs (no register double buffering needed);
//   stage = 2 k-tiles (64 k): [W 4F x 2 KiB][X hi TT x 2 KiB][X lo TT x 2 KiB], pieces dealt round-robin to the 8 waves;
//   sync per stage: counted s_waitcnt vmcnt (the wave's own pieces of the NEXT stage; younger stages stay in flight) ->
//   s_barrier (raw: __syncthreads() would drain vmcnt(0), LDS-DMA counts as a pending LDS write) -> refill of the stage that
//   just moved to registers -> fragment requests of the next stage -> MFMAs of this one (fragments double-buffered in registers).
// Accumulation order per output = the 16-row kernel's: a fresh accumulator per K-slice of nk k-tiles (k ascending, hi then
// lo), slice partials added in slice order to a running total that starts from +0 => bit-identical to single-stream runs.
// blockIdx -> tile: XCD-aware (block b runs on XCD b % 8): an XCD owns a contiguous range of weight row tiles and walks
// the token groups inside it, so each weight byte enters ONE L2 (not eight) and is re-read there by the token groups.
template <int N> __device__ __forceinline__ void wm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int LPW>
__device__ __forceinline__ void wm_wait_younger(int younger)          // block-uniform: `younger` stages issued after the one needed
{
    static_assert(8 * LPW <= 63, "vmcnt is a 6-bit counter");
    switch (younger) {
        case 0: wm_wait_vmcnt<0>(); break;
        case 1: wm_wait_vmcnt<LPW>(); break;
        case 2: wm_wait_vmcnt<2 * LPW>(); break;
        case 3: wm_wait_vmcnt<3 * LPW>(); break;
        case 4: wm_wait_vmcnt<4 * LPW>(); break;
        case 5: wm_wait_vmcnt<5 * LPW>(); break;
        case 6: wm_wait_vmcnt<6 * LPW>(); break;
        case 7: wm_wait_vmcnt<7 * LPW>(); break;
        default: wm_wait_vmcnt<8 * LPW>(); break;
    }
}

template <int F, int TT>
struct TileGemmCfg {
    static constexpr int TH = TT / 2;
    static constexpr int NPW = 8 * F, NPX = 2 * TT * WM_ACT_PLANES, NP = NPW + NPX;   // 1 KiB pieces of a stage (2 k-tiles; token tiles: one plane or hi + lo)
    static constexpr int LPW = NP / 8;                                // pieces per wave per stage
    static constexpr int STAGE = NP * 1024;
    static constexpr int R_FIT = (160 * 1024) / STAGE;
    static constexpr int R = R_FIT > 9 ? 9 : R_FIT;                   // ring stages (<= 8 younger stages in flight)
    static_assert(TT % 2 == 0 && NP % 8 == 0 && R >= 3, "tile shape");
};

template <int F, int TT, class Ep>
__global__ void __launch_bounds__(512)
k_tile_gemm(const bf16_t* __restrict__ W, int N16, int K32, int nk, const bf16_t* __restrict__ X, size_t plane, int MT,
            int n_fb, int n_tg, const int* __restrict__ done, Ep ep, const int* __restrict__ ntiles)
{
    typedef TileGemmCfg<F, TT> C;
    constexpr int TH = C::TH, LPW = C::LPW, R = C::R, STAGE = C::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = wave & 3, hh = wave >> 2;
    // XCD-aware tile order (bijective for any grid size, cdna_hip_programming.md §5 template)
    const int nwg = n_fb * n_tg;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int fb = bid / n_tg, tg = bid - fb * n_tg;
    const int rt0 = fb * 4 * F, mt0 = tg * TT;
    const int NS = K32 >> 1;                   // stages

    // per-piece source pointers of this wave (stage 0); a stage advances every piece by 2 KiB
    const char* src[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int p = i * 8 + wave;            // piece index inside the stage = LDS position
        if (p < C::NPW) {
            const int t = min(rt0 + (p >> 1), N16 - 1), kk = p & 1;
            src[i] = reinterpret_cast<const char*>(W) + ((size_t)t * K32 + kk) * 1024 + lane * 16;
        } else {
            const int q = p - C::NPW, pl = q / (2 * TT), rr = q - pl * 2 * TT;
            const int t = min(mt0 + (rr >> 1), MT - 1), kk = rr & 1;
            src[i] = reinterpret_cast<const char*>(X + (pl ? plane : 0)) + ((size_t)t * K32 + kk) * 1024 + lane * 16;
        }
    }
    auto issue = [&](int s) {
        char* sb = smem + (s % R) * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)s * 2048),
                                             (__attribute__((address_space(3))) void*)(sb + i * 8192), 16, 0, 0);
    };
    // fragments of one stage (both k-tiles): a[kk][i], token hi / lo [kk][j]
    struct Frags { wfrag_t a[2][F]; ActFrag x[2][TH]; };
    auto frag_load = [&](int s, Frags& f) {
        const char* sb = smem + (s % R) * STAGE + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < F; ++i) f.a[kk][i] = __builtin_bit_cast(wfrag_t, *reinterpret_cast<const uint4*>(sb + ((w * F + i) * 2 + kk) * 1024));
#pragma unroll
            for (int j = 0; j < TH; ++j)        // (LDS image: the hi plane's 2 TT pieces, then — two-plane builds — the lo plane's, 2 TT KiB further)
                f.x[kk][j] = act_ld(reinterpret_cast<const bf16_t*>(sb + C::NPW * 1024 + ((hh * TH + j) * 2 + kk) * 1024), (size_t)TT * 1024);
        }
    };
#pragma unroll
    for (int s = 0; s < R; ++s)
        if (s < NS) issue(s);
    if ((done && *done) || (ntiles && mt0 >= *ntiles)) {   // every stream finished / no rows in this token group in this step: nothing may
        wm_wait_vmcnt<0>();                                // stay in flight into a released LDS allocation
        return;
    }

    f32x4_t acc[F][TH], tot[F][TH];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < TH; ++j) { acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; tot[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    int left = nk;                             // k-tiles until the current slice is complete (nk is even)

    // stage 0 -> registers
    wm_wait_younger<LPW>(min(NS, R) - 1);
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0) as the builtin: the compiler's wait-count pass sees it (wm_encoder.hip ring_barrier)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");             // compiler-only: the fragment reads below must not be hoisted above the barrier
    Frags f0, f1;
    frag_load(0, f0);

    // One step: stage s sits in `cur`; wait for stage s+1, barrier, refill the buffer of stage s (it lives in registers now), request
    // the fragments of stage s+1 into `nxt`, then issue the MFMAs of stage s — the LDS round trip of a stage hides under the MFMAs of
    // the stage before it.  The request is unconditional (after the last stage it re-reads a stale buffer, unused): a branch there
    // makes the compiler wait for those very reads in front of the MFMAs.
    auto step = [&](int s, const Frags& cur, Frags& nxt) {
        wm_wait_younger<LPW>(max(min(NS - 1, s + R - 1) - (s + 1), 0));
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");         // compiler-only: the fragment reads below must not be hoisted above the barrier
        if (s + R < NS) issue(s + R);
        frag_load(s + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < F; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][j] = mfma_act(cur.a[kk][i], cur.x[kk][j], acc[i][j]);      // (per output: hi then lo, k ascending)
        }
        left -= 2;
        if (left == 0) {                       // slice complete: add its partial to the running total, start a fresh accumulator
            left = nk;
#pragma unroll
            for (int i = 0; i < F; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) {
                    tot[i][j][0] += acc[i][j][0]; tot[i][j][1] += acc[i][j][1]; tot[i][j][2] += acc[i][j][2]; tot[i][j][3] += acc[i][j][3];
                    acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
        }
    };
    for (int s = 0; s < NS; s += 2) {          // NS is even (d_model and ffn_dim are multiples of 128)
        step(s, f0, f1);
        step(s + 1, f1, f0);
    }
    // epilogue: the operands of every tile (bias, residual, cache position) are requested in one batch, then the stores go out
    EpPre pre[F][TH];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < TH; ++j) {
            const int rt = rt0 + w * F + i, mt = mt0 + hh * TH + j;
            pre[i][j].i = 0; pre[i][j].a = make_float4(0.f, 0.f, 0.f, 0.f); pre[i][j].b = pre[i][j].a;
            if (rt < N16 && mt < MT) pre[i][j] = ep.pre(mt * 16 + (lane & 15), rt * 16 + 4 * (lane >> 4));
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < TH; ++j) {
            const int rt = rt0 + w * F + i, mt = mt0 + hh * TH + j;
            if (rt < N16 && mt < MT) ep.fin(mt * 16 + (lane & 15), rt * 16 + 4 * (lane >> 4), tot[i][j], pre[i][j]);
        }
}

// ---- host-side launch plan -------------------------------------------------------------------
static thread_local const int* g_skinny_done = nullptr;     // device flag checked by every launch of this translation unit
// merged-step schedule (wm_decoder.hip wm_dec_step): the pass's rows are dense and their number changes from step to step while the launches
// (a captured graph) are sized for the maximum: device word = 16-row token tiles that hold rows in this step; the batched kernels' blocks
// of tiles beyond it exit at once.  nullptr (every other pass): all tiles of the launch.
static thread_local const int* g_skinny_ntiles = nullptr;
struct SkinnyPlan { int ksplit, rt, nk, RT; };      // rt: row-tile groups per block (waves), RT: row tiles per wave (registers)

// K-slices of at most 16 fragments (8 when the token operand is normalised in registers) and, if possible, >= 1024 waves.
// The plan depends only on (N16, K32, loader kind): 16-row and batched launches of one GEMM share it, which is what makes
// their results bit-identical.  nk = K32 / ksplit is always one of {4, 8, 12, 16}.
static inline int skinny_env(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}
static inline SkinnyPlan skinny_plan(int N16, int K32, bool norm_loader) {
    static const int cap = skinny_env("WM_PLAN_WAVE_CAP", 10);          // tuning knobs (bench sweeps); defaults are the shipped plan
    static const int target = skinny_env("WM_PLAN_TARGET_WAVES", 1024);
    static const int nkmax_env = skinny_env("WM_PLAN_NK_MAX", 16);
    const int nkmax = norm_loader ? std::min(nkmax_env, 8) : nkmax_env;
    SkinnyPlan p;
    const int U = (K32 % 8 == 0) ? 8 : 4;
    const int q = K32 / U;                  // candidate ksplit must divide q
    int best = 1;
    for (int s = 1; s <= cap && s <= q; ++s) {      // <= 10 waves per block (launch bound 640 threads)
        if (q % s) continue;
        best = s;
        if ((long)N16 * s >= target && K32 / s <= nkmax) break;
    }
    p.ksplit = best;
    p.nk = K32 / best;
    p.rt = (best == 1) ? 4 : 1;
    // more row tiles than CUs: a second block on some CUs doubles their share of the launch's memory batch (the launch then
    // waits for those CUs) — let a wave own two row tiles instead (the token fragment and its LayerNorm are shared)
    static const int rt2 = skinny_env("WM_PLAN_RT2", 1);
    p.RT = (rt2 && best >= 2 && p.nk <= 8 && N16 > 256 && N16 <= 1024) ? 2 : 1;
    return p;
}

static inline PfJob pf_for_gemm(const bf16_t* W, bool fp8, int N16, int K32, bool norm_loader) {
    const SkinnyPlan p = skinny_plan(N16, K32, norm_loader);
    const int per_block = p.rt * p.RT, grid = (N16 + per_block - 1) / per_block;
    const unsigned long long tile = (unsigned long long)K32 * 512 * (fp8 ? 1 : 2);
    // one job per consumer block (same XCD by construction).  Cutting the matrix into 240 / 480 equal jobs instead — more, smaller
    // extra blocks, no affinity — measured 2.73 / 2.77 ms per iteration against 2.57 (tests/microbench/r02_call14.sh)
    return PfJob{reinterpret_cast<const char*>(W), nullptr, (unsigned)(tile * per_block), (unsigned)grid, tile * N16};
}

// The job a BATCHED launch (R > 16 rows) carries for the LayerNorm-folded GEMM (W, N16, K32) that follows it, `extra` = a second matrix of the
// same size: MT == 2 -> consumer k_skinny2_gemm, one job per consumer block; MT >= 3 -> token-tile consumer, 64 KB pieces of XCD eighths.
static inline void pf_set_batched(const bf16_t* W, const void* extra, bool fp8, int N16, int K32, int MT) {
    const SkinnyPlan p = skinny_plan(N16, K32, true);
    const unsigned long long wbytes = (unsigned long long)N16 * K32 * (fp8 ? 512 : 1024);
    if (MT == 2 && skinny_env("WM_SKINNY2", 1) && p.ksplit * p.rt <= 10) {
        g_pf_batched = PfJob{reinterpret_cast<const char*>(W), reinterpret_cast<const char*>(extra), (unsigned)((unsigned long long)p.rt * K32 * (fp8 ? 512 : 1024)),
                             (unsigned)((N16 + p.rt - 1) / p.rt), wbytes};
        g_pf_batched_sliced = 0;
    } else {
        const unsigned long long slice = ((wbytes + 7) / 8 + 1023) & ~1023ull;
        g_pf_batched = PfJob{reinterpret_cast<const char*>(W), reinterpret_cast<const char*>(extra), 65536u, (unsigned)(8 * ((slice + 65535) / 65536)), wbytes};
        g_pf_batched_sliced = 1;
    }
}

// a weight matrix in the packed layout: bf16 (scale == nullptr) or fp8 e4m3 with one fp32 scale per output row
struct WRef {
    const bf16_t* w; const float* scale;
    WRef(const bf16_t* w_, const float* scale_ = nullptr) : w(w_), scale(scale_) {}
};

template <int NK, int RT, bool W8, class Ld, class Ep, bool FOLD = false>
static inline hipError_t launch_skinny_nk_rt(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const Ld& ld, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    const int per_block = p.rt * RT;
    const int grid = (N16 + per_block - 1) / per_block;
    const int threads = 64 * p.ksplit * p.rt;
    const size_t lds = (size_t)ld.lds_bytes() + (p.ksplit > 1 ? (size_t)per_block * p.ksplit * 1024 : 0) + (FOLD ? (size_t)2 * fold.T16 * sizeof(float2) : 0);
    if (FOLD && 2 * fold.T16 > threads) return hipErrorInvalidConfiguration;      // 16 rows x T16 / 8 partial groups, one per thread
    const int magic = (256 + p.ksplit - 1) / p.ksplit;         // (wave * magic) >> 8 == wave / ksplit for wave < 16
    auto kern = k_skinny_gemm<NK, RT, W8, Ld, Ep, FOLD>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const PfJob pf = g_pf_job;
    g_pf_job = PfJob{nullptr, nullptr, 0u, 0u, 0ull};
    const int grid_all = pf.n_jobs ? pf_round8(grid) + (int)pf.n_jobs : grid;
    const LdArgs la = ld.pack();
    if (p.ksplit > 255 || p.rt > 255 || magic > 32767 || !ld.packs()) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(kern, dim3(grid_all), dim3(threads), lds, st, W.w, la.a, la.b, la.c, la.i0, la.i1, N16, K32, p.ksplit | (p.rt << 8) | (magic << 16), grid,
                       W.scale, g_skinny_done, ep, pf, fold TL_PASS);
    return hipGetLastError();
}
// the bytes block j of the skinny GEMM (W, N16, K32, loader kind) reads: one prefetch job per consumer block
static inline PfJob pf_for_gemm(const bf16_t* W, bool fp8, int N16, int K32, bool norm_loader);
template <int NK, bool W8, class Ld, class Ep, bool FOLD = false>
static inline hipError_t launch_skinny_nk(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const Ld& ld, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    if constexpr (NK <= 8) {
        if (p.RT == 2 && p.ksplit >= 2) return launch_skinny_nk_rt<NK, 2, W8, Ld, Ep, FOLD>(st, W, N16, K32, p, ld, ep, fold);
    }
    return launch_skinny_nk_rt<NK, 1, W8, Ld, Ep, FOLD>(st, W, N16, K32, p, ld, ep, fold);
}

template <bool W8, class Ld, class Ep, bool FOLD = false>
static inline hipError_t launch_skinny_w(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const Ld& ld, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    if (p.ksplit * p.rt > 10 || p.ksplit * p.nk != K32 || (Ld::kNorm && p.nk > 8)) return hipErrorInvalidConfiguration;
    if (Ld::kNorm && K32 * 16 > ((p.nk + 3) / 4) * 64 * p.ksplit * p.rt) return hipErrorInvalidConfiguration;      // gamma | beta float4 per thread (LdNormT::issue)
    switch (p.nk) {
        case 4: return launch_skinny_nk<4, W8, Ld, Ep, FOLD>(st, W, N16, K32, p, ld, ep, fold);
        case 8: return launch_skinny_nk<8, W8, Ld, Ep, FOLD>(st, W, N16, K32, p, ld, ep, fold);
        case 12: if constexpr (!Ld::kNorm && !FOLD) return launch_skinny_nk<12, W8>(st, W, N16, K32, p, ld, ep); break;      // (a folded GEMM keeps the LayerNorm plan: <= 8 fragments per K-slice)
        case 16: if constexpr (!Ld::kNorm && !FOLD) return launch_skinny_nk<16, W8>(st, W, N16, K32, p, ld, ep); break;
        default: break;
    }
    return hipErrorInvalidConfiguration;
}

template <class Ld, class Ep, bool FOLD = false>
static inline hipError_t launch_skinny(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const Ld& ld, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    if (W.scale) return launch_skinny_w<true, Ld, Ep, FOLD>(st, W, N16, K32, p, ld, ep, fold);
    return launch_skinny_w<false, Ld, Ep, FOLD>(st, W, N16, K32, p, ld, ep, fold);
}

template <int NKR, int RT, bool W8, class Ep, bool FOLD = false, int TT = 2>
static inline hipError_t launch_rows_gemm_w(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                            const bf16_t* X, size_t plane, int MT, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    const dim3 grid((N16 + RT - 1) / RT, (MT + TT - 1) / TT);
    const size_t lds = (p.ksplit > 1 ? (size_t)RT * TT * p.ksplit * 1024 : 0) + (FOLD ? (size_t)TT * 2 * fold.T16 * sizeof(float2) : 0);
    if (FOLD && TT * 2 * fold.T16 > 64 * p.ksplit) return hipErrorInvalidConfiguration;      // TT x 16 rows x T16 / 8 partial groups, one per thread
    auto kern = k_rows_gemm<NKR, RT, TT, W8, Ep, FOLD>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const PfJob pf = g_pf_batched; const int pfs = g_pf_batched_sliced;
    g_pf_batched = PfJob{nullptr, nullptr, 0u, 0u, 0ull};
    const dim3 grid_all(grid.x, grid.y + (pf.n_jobs ? (pf.n_jobs + grid.x - 1) / grid.x : 0));
    hipLaunchKernelGGL(kern, grid_all, dim3(64 * p.ksplit), lds, st, W.w, X, plane, g_skinny_done, g_skinny_ntiles, N16, K32, p.ksplit, MT, W.scale, ep, fold, pf, pfs TL_PASS);
    return hipGetLastError();
}

template <int NKR, int RT, class Ep, bool FOLD = false>
static inline hipError_t launch_rows_gemm(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                          const bf16_t* X, size_t plane, int MT, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    if (W.scale) return launch_rows_gemm_w<NKR, RT, true, Ep, FOLD>(st, W, N16, K32, p, X, plane, MT, ep, fold);
    return launch_rows_gemm_w<NKR, RT, false, Ep, FOLD>(st, W, N16, K32, p, X, plane, MT, ep, fold);
}

template <int NKR, class Ep, bool FOLD = false>
static inline hipError_t launch_skinny_mt_nk(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                             const bf16_t* X, size_t plane, int MT, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    static const int min_blocks = skinny_env("WM_ROWS_GEMM_MIN_BLOCKS", 400);
    // weight row tiles per wave: as many as still leave >= 400 blocks (~1.5 per CU; swept 100..800 at 8 and 32 streams) (register blocking divides the L2 re-reads
    // of the token operand; with few token tiles the chip has to be filled by features instead).  Same results.
    const int groups = (MT + 1) / 2;
    if (((N16 + 3) / 4) * groups >= min_blocks) return launch_rows_gemm<NKR, 4, Ep, FOLD>(st, W, N16, K32, p, X, plane, MT, ep, fold);
    if (((N16 + 1) / 2) * groups >= min_blocks) return launch_rows_gemm<NKR, 2, Ep, FOLD>(st, W, N16, K32, p, X, plane, MT, ep, fold);
    return launch_rows_gemm<NKR, 1, Ep, FOLD>(st, W, N16, K32, p, X, plane, MT, ep, fold);
}

// ---- LDS-ring token-tile GEMM: tile shape and launch ----
// One round of blocks (<= 256, one per CU) costs what ONE block costs, and a block's time is its L2 -> LDS bytes,
// K x (128 F + 64 TT) [weight rows 2 B, token rows 4 B as a hi/lo pair]: pick the shape that minimises rounds x bytes; ties go
// to two row tiles per wave (each token fragment read from LDS then feeds two MFMAs).  WM_TILE_F / WM_TILE_TT pin it (sweeps).
struct TilePlan { int F, TT; };
static inline TilePlan tile_plan(int N16, int MT) {
    const int envF = skinny_env("WM_TILE_F", 0), envTT = skinny_env("WM_TILE_TT", 0);       // read per launch: sweeps flip them inside one process
    TilePlan best{1, 2};
    double bc = 1e30;
    for (int F = 2; F >= 1; --F)
        for (int TT = 2; TT <= 8; TT += 2) {
            if ((envF && F != envF) || (envTT && TT != envTT)) continue;
            if ((8 * F + 2 * TT * WM_ACT_PLANES) % 8) continue;          // a stage's pieces are dealt evenly to the 8 waves
            const int blocks = ((N16 + 4 * F - 1) / (4 * F)) * ((MT + TT - 1) / TT);
            const double rounds = blocks <= 256 ? 1.0 : blocks / 256.0;
            const double cost = rounds * (128.0 * F + 64.0 * TT);
            if (cost < bc - 1e-9) { bc = cost; best = TilePlan{F, TT}; }
        }
    return best;
}

template <int F, int TT, class Ep>
static inline hipError_t launch_tile_gemm_ft(hipStream_t st, const bf16_t* W, int N16, int K32, int nk, const bf16_t* X, size_t plane, int MT, const Ep& ep) {
    typedef TileGemmCfg<F, TT> C;
    const int n_fb = (N16 + 4 * F - 1) / (4 * F), n_tg = (MT + TT - 1) / TT;
    constexpr int lds = C::R * C::STAGE;
    auto kern = k_tile_gemm<F, TT, Ep>;
    static thread_local bool attr_done = false;        // per instantiation and host thread
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_fb * n_tg), dim3(512), lds, st, W, N16, K32, nk, X, plane, MT, n_fb, n_tg, g_skinny_done, ep, g_skinny_ntiles);
    return hipGetLastError();
}

template <class Ep>
static inline hipError_t launch_tile_gemm(hipStream_t st, const bf16_t* W, int N16, int K32, int nk, const bf16_t* X, size_t plane, int MT, const Ep& ep) {
    const TilePlan t = tile_plan(N16, MT);
#define WM_TG(Fv, TTv) if constexpr ((8 * Fv + 2 * TTv * WM_ACT_PLANES) % 8 == 0) if (t.F == Fv && t.TT == TTv) return launch_tile_gemm_ft<Fv, TTv>(st, W, N16, K32, nk, X, plane, MT, ep)
    WM_TG(1, 2); WM_TG(1, 4); WM_TG(1, 6); WM_TG(1, 8);
    WM_TG(2, 2); WM_TG(2, 4); WM_TG(2, 6); WM_TG(2, 8);
#undef WM_TG
    return hipErrorInvalidConfiguration;
}

// Which kernel takes a GEMM of MT >= 2 token tiles.  Measured on MI355X, large-v2, 352 rows (tests/microbench/r03_sweep.py,
// profiles/r03_tile_gemm_sweep.md): both kernels are bound by the L2 -> CU fill rate (~31 B/clk per CU on L2 hits): the LDS-ring
// tile kernel moves fewer bytes but pays ~4.5 us of ring fill / drain per launch, so it wins where the grid is many rounds of blocks
// deep (vocabulary projection: 211 -> 116 us at 352 rows, 63 -> 39 us at 88) and loses on the one-round layer GEMMs (out-proj
// 6.6 -> 9.4 us, QKV 19 -> 18 us, FC2 23 -> 29 us).  WM_TILE_GEMM_MIN_N16 / _MIN_MT move the boundary (sweeps; 0 = tile kernel off).
static inline bool use_tile_gemm(int N16, int K32, int MT, bool w8, int nk) {
    const int min_mt = skinny_env("WM_TILE_GEMM_MIN_MT", 3), min_n16 = skinny_env("WM_TILE_GEMM_MIN_N16", 1024);
    return min_mt > 0 && MT >= min_mt && N16 >= min_n16 && !w8 && (nk & 1) == 0 && (K32 & 1) == 0;
}

template <int NK, bool W8, class Ep, bool FOLD = false>
static inline hipError_t launch_skinny2_nk(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const bf16_t* X, size_t plane, int R, const Ep& ep,
                                           const FoldIn& fold = FoldIn{}) {
    const int grid = (N16 + p.rt - 1) / p.rt, threads = 64 * p.ksplit * p.rt;
    const size_t lds = (p.ksplit > 1 ? (size_t)p.rt * 2 * p.ksplit * 1024 : 0) + (FOLD ? (size_t)4 * fold.T16 * sizeof(float2) : 0);
    if (FOLD && 4 * fold.T16 > threads) return hipErrorInvalidConfiguration;      // 32 rows x T16 / 8 partial groups, one per thread
    const int magic = (256 + p.ksplit - 1) / p.ksplit;
    PfJob pf = g_pf_batched;
    g_pf_batched = PfJob{nullptr, nullptr, 0u, 0u, 0ull};
    if (g_pf_batched_sliced) pf.n_jobs = 0;          // (a job cut for the token-tile consumer: not this kernel's kind)
    const int grid_all = pf.n_jobs ? pf_round8(grid) + (int)pf.n_jobs : grid;
    hipLaunchKernelGGL((k_skinny2_gemm<NK, W8, Ep, FOLD>), dim3(grid_all), dim3(threads), lds, st, W.w, X, plane, g_skinny_done, N16, K32, p.ksplit | (p.rt << 8) | (magic << 16), R,
                       W.scale, ep, fold, pf);
    return hipGetLastError();
}
template <bool W8, class Ep, bool FOLD = false>
static inline hipError_t launch_skinny2_w(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const bf16_t* X, size_t plane, int R, const Ep& ep,
                                          const FoldIn& fold = FoldIn{}) {
    switch (p.nk) {
        case 4: return launch_skinny2_nk<4, W8, Ep, FOLD>(st, W, N16, K32, p, X, plane, R, ep, fold);
        case 8: return launch_skinny2_nk<8, W8, Ep, FOLD>(st, W, N16, K32, p, X, plane, R, ep, fold);
        case 12: if constexpr (!FOLD) return launch_skinny2_nk<12, W8>(st, W, N16, K32, p, X, plane, R, ep); break;
        case 16: if constexpr (!FOLD) return launch_skinny2_nk<16, W8>(st, W, N16, K32, p, X, plane, R, ep); break;
        default: break;
    }
    return hipErrorInvalidConfiguration;
}
template <class Ep, bool FOLD = false>
static inline hipError_t launch_skinny2(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p, const bf16_t* X, size_t plane, int R, const Ep& ep,
                                        const FoldIn& fold = FoldIn{}) {
    if (W.scale) return launch_skinny2_w<true, Ep, FOLD>(st, W, N16, K32, p, X, plane, R, ep, fold);
    return launch_skinny2_w<false, Ep, FOLD>(st, W, N16, K32, p, X, plane, R, ep, fold);
}

template <class Ep, bool FOLD = false>
static inline hipError_t launch_skinny_mt(hipStream_t st, WRef W, int N16, int K32, const SkinnyPlan& p,
                                          const bf16_t* X, size_t plane, int MT, int R, const Ep& ep, const FoldIn& fold = FoldIn{}) {
    // the LDS-ring tile kernel (same accumulation order, bit-identical results)
    if constexpr (!FOLD) {
        if (use_tile_gemm(N16, K32, MT, W.scale != nullptr, p.nk))
            return launch_tile_gemm(st, W.w, N16, K32, p.nk, X, plane, MT, ep);
    }
    // two token tiles: the weight-streaming kernel with a second token tile (WM_SKINNY2=0: the register-blocked kernel)
    const bool s2_dbg = (FOLD ? skinny_env("WM_SKINNY2_FOLD", 1) : 1) && (std::is_same<Ep, EpResidualFold>::value ? skinny_env("WM_SKINNY2_RESFOLD", 1) : 1);
    if (MT == 2 && skinny_env("WM_SKINNY2", 1) && s2_dbg && p.ksplit * p.rt <= 10 && p.ksplit * p.nk == K32)
        return launch_skinny2<Ep, FOLD>(st, W, N16, K32, p, X, plane, R, ep, fold);
    if constexpr (!FOLD) {
        if (p.nk == 16) return launch_skinny_mt_nk<16>(st, W, N16, K32, p, X, plane, MT, ep);
        if (p.nk == 12) return launch_skinny_mt_nk<12>(st, W, N16, K32, p, X, plane, MT, ep);
    }
    if (p.nk == 8) return launch_skinny_mt_nk<8, Ep, FOLD>(st, W, N16, K32, p, X, plane, MT, ep, fold);
    if (p.nk == 4) return launch_skinny_mt_nk<4, Ep, FOLD>(st, W, N16, K32, p, X, plane, MT, ep, fold);
    return hipErrorInvalidConfiguration;
}

// LayerNorm-fed GEMM with the LayerNorm folded (wm_common.h): X = gamma o x as packed hi / lo planes (written by the launch that produced x),
// fold = the rows' statistics partials + c = W gamma; the epilogue's bias is b' = b + W beta.  The plan is the LayerNorm plan (<= 8 fragments per
// K-slice: 64 ksplit threads >= 32 d / 128 partial groups, whatever d_model), shared by the 16-row, two-tile and token-tile kernels.
template <class Ep>
static inline hipError_t launch_skinny_fold(hipStream_t st, WRef W, int N16, int K32, int R, const bf16_t* X, size_t plane, const FoldIn& fold, const Ep& ep) {
    const SkinnyPlan p = skinny_plan(N16, K32, true);
    if (p.nk > 8 || fold.T16 * 16 != K32 * 32) return hipErrorInvalidConfiguration;
    if (R <= 16) return launch_skinny<LdPacked, Ep, true>(st, W, N16, K32, p, LdPacked{X, K32, plane, R}, ep, fold);
    return launch_skinny_mt<Ep, true>(st, W, N16, K32, p, X, plane, (R + 15) / 16, R, ep, fold);
}

// out = X (R token rows, packed hi/lo planes in global memory) times W^T (N = 16*N16 features, K = 32*K32)
template <class Ep>
static inline hipError_t launch_skinny_rows(hipStream_t st, WRef W, int N16, int K32, int R, const bf16_t* X, size_t plane,
                                            const Ep& ep) {
    const SkinnyPlan p = skinny_plan(N16, K32, false);
    if (R <= 16) return launch_skinny(st, W, N16, K32, p, LdPacked{X, K32, plane, R}, ep);
    return launch_skinny_mt(st, W, N16, K32, p, X, plane, (R + 15) / 16, R, ep);
}

template <class Ld, class Ep>
static inline hipError_t launch_skinny_norm_t(hipStream_t st, WRef W, int N16, int K32, const Ld& ld, const Ep& ep, bf16_t* xscr, size_t plane) {
    const SkinnyPlan p = skinny_plan(N16, K32, true);
    const int R = ld.M;
    if (p.nk > 8 || K32 * 32 != ld.d || K32 * 16 > ((p.nk + 3) / 4) * 64 * p.ksplit) return hipErrorInvalidConfiguration;
    if (R <= 16) return launch_skinny(st, W, N16, K32, p, ld, ep);
    const int MT = (R + 15) / 16;
    // two token tiles: LayerNorm fused into the two-tile weight-streaming kernel.  OFF by default (WM_SKINNY2_NORM=1 turns it on): measured
    // slower than the LayerNorm launch + unfused kernel (vanilla step at 32 streams 4.02 vs 3.68 ms, tests/microbench/r03_call4.sh) — every
    // one of the 80..320 blocks of a GEMM pulls the 32 fp32 rows (160 KB) through its CU and normalises them again; at two tiles that
    // L2 -> CU traffic costs more than the three ~7 us launches it removes.  Kept: bit-identical, re-measurable on other shapes.
    if (MT == 2 && p.ksplit * p.rt <= 6 && p.ksplit <= 8 && p.ksplit * p.nk == K32 && skinny_env("WM_SKINNY2", 1) && skinny_env("WM_SKINNY2_NORM", 0)) {
        const int grid = (N16 + p.rt - 1) / p.rt, threads = 64 * p.ksplit * p.rt;
        const size_t lds = (size_t)ld.lds_bytes() + (p.ksplit > 1 ? (size_t)p.rt * 2 * p.ksplit * 1024 : 0);
        const int magic = (256 + p.ksplit - 1) / p.ksplit;
        if (W.scale) {
            if (p.nk == 8) hipLaunchKernelGGL((k_skinny2_norm<8, true, Ld, Ep>), dim3(grid), dim3(threads), lds, st, W.w, W.scale, N16, K32, p.ksplit, p.rt, magic, g_skinny_done, ld, ep);
            else hipLaunchKernelGGL((k_skinny2_norm<4, true, Ld, Ep>), dim3(grid), dim3(threads), lds, st, W.w, W.scale, N16, K32, p.ksplit, p.rt, magic, g_skinny_done, ld, ep);
        } else {
            if (p.nk == 8) hipLaunchKernelGGL((k_skinny2_norm<8, false, Ld, Ep>), dim3(grid), dim3(threads), lds, st, W.w, W.scale, N16, K32, p.ksplit, p.rt, magic, g_skinny_done, ld, ep);
            else hipLaunchKernelGGL((k_skinny2_norm<4, false, Ld, Ep>), dim3(grid), dim3(threads), lds, st, W.w, W.scale, N16, K32, p.ksplit, p.rt, magic, g_skinny_done, ld, ep);
        }
        return hipGetLastError();
    }
    // weight prefetch riding on the LayerNorm launch (token-tile path with bf16 weights only; WM_LN_PREFETCH=0 turns it off)
    const int ln_pf = skinny_env("WM_LN_PREFETCH", 1);
    PfJob pf{nullptr, nullptr, 0u, 0u, 0ull};
    int grid = MT, pf_sliced = 1;
    const unsigned long long wbytes = (unsigned long long)N16 * K32 * (W.scale ? 512 : 1024);
    const char* extra = reinterpret_cast<const char*>(g_ln_pf_extra);
    g_ln_pf_extra = nullptr;
    if (ln_pf && MT == 2 && skinny_env("WM_SKINNY2", 1) && p.ksplit * p.rt <= 10) {
        // consumer = k_skinny2_gemm: block j reads the p.rt row tiles j*p.rt.. — one job per consumer block
        const unsigned job_bytes = (unsigned)((unsigned long long)p.rt * K32 * (W.scale ? 512 : 1024));
        pf = PfJob{reinterpret_cast<const char*>(W.w), extra, job_bytes, (unsigned)((N16 + p.rt - 1) / p.rt), wbytes};
        pf_sliced = 0;
    } else if (ln_pf && MT >= 3) {
        const unsigned long long slice = ((wbytes + 7) / 8 + 1023) & ~1023ull;
        const unsigned job_bytes = 64 * 1024;
        pf = PfJob{reinterpret_cast<const char*>(W.w), extra, job_bytes, (unsigned)(8 * ((slice + job_bytes - 1) / job_bytes)), wbytes};
    }
    // row sub-blocks per token tile (WM_LN_SUB = 1 / 2 / 4 / 8; 1 = the whole-tile form of rounds 1-4)
    int sub = skinny_env("WM_LN_SUB", 2);
    if (sub != 1 && sub != 2 && sub != 4 && sub != 8) sub = 1;
    const int nmain = MT * sub;
    grid = nmain;
    if (pf.n_jobs) grid = pf_round8(nmain) + (int)pf.n_jobs;
    const LdArgs la = ld.pack();
    if (!ld.packs() || p.ksplit > 255) return hipErrorInvalidConfiguration;
    if (p.nk == 8) hipLaunchKernelGGL((k_ln_tiles<8, Ld>), dim3(grid), dim3(64 * p.ksplit), ld.lds_bytes(), st, la.a, la.b, la.c, la.i0, la.i1, K32, p.ksplit | (sub << 8), nmain,
                                      g_skinny_ntiles, xscr, plane, g_skinny_done, pf, pf_sliced);
    else if (p.nk == 4) hipLaunchKernelGGL((k_ln_tiles<4, Ld>), dim3(grid), dim3(64 * p.ksplit), ld.lds_bytes(), st, la.a, la.b, la.c, la.i0, la.i1, K32, p.ksplit | (sub << 8), nmain,
                                           g_skinny_ntiles, xscr, plane, g_skinny_done, pf, pf_sliced);
    else return hipErrorInvalidConfiguration;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_skinny_mt(st, W, N16, K32, p, xscr, plane, MT, R, ep);
}

// LayerNorm-fused GEMM over R token rows.  R <= 16: one fused launch.  R > 16: the same LayerNorm code writes
// the packed hi/lo operand to `xscr` (global), then the register-blocked token-tile kernel runs.
template <class Ep>
static inline hipError_t launch_skinny_norm(hipStream_t st, WRef W, int N16, int K32, const float* h, const float* gamma,
                                            const float* beta, int d, int R, int row_mul, int row_off, int do_norm, const Ep& ep,
                                            bf16_t* xscr, size_t plane) {
    if (do_norm) return launch_skinny_norm_t(st, W, N16, K32, LdNorm{h, gamma, beta, d, K32, R, row_mul, row_off}, ep, xscr, plane);
    return launch_skinny_norm_t(st, W, N16, K32, LdIdent{h, gamma, beta, d, K32, R, row_mul, row_off}, ep, xscr, plane);
}
