// wm_encoder.hip — log-mel front end (F0), Whisper encoder (F1) and cross-K/V projection (F2).
// MFMA-bound prefill: every matmul runs on v_mfma_f32_16x16x32_bf16 from packed operands.
#include "wm_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include "wm_enc_epilogues.h"

// =============================================================================================
// Tiled GEMM  out[m][n] = sum_k X[m][k] W[n][k],  X/W packed bf16.  128x128 tile, BK=64, 4 waves (2x2),
// each wave 64 features x 64 tokens = 4x4 MFMA tiles.  Operands are staged with global_load_lds
// (16 B/lane, one 1-KiB packed fragment per wave-instruction): the LDS image is fragment-major, so
// every ds_read_b128 is lane-linear and bank-conflict-free without any swizzle.  2-stage LDS ring.
// =============================================================================================
#define GT_BN 128

__device__ __forceinline__ void glds16(const bf16_t* gsrc, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#ifndef WM_FLASH_VALU
#define WM_FLASH_VALU 2      // k_flash_enc softmax arithmetic (see the kernel); A/B builds: -DWM_FLASH_VALU=1
#endif
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// max of values known not to be NaN.  fmaxf under IEEE mode canonicalises every operand that is not "known canonical" with a v_max x, x first — MFMA
// results and permlane outputs never are: 15 instructions per 8-score tile where 7 do.  This file is compiled with -fno-honor-nans (build.py: its
// softmax / LayerNorm / GELU arithmetic never produces or tests a NaN; -inf masks are infinities, not NaNs), which puts `nnan` on the calls and lets
// the backend emit v_max_f32 / v_max3_f32 as written.
// (NOT inline asm: an asm statement reading MFMA results gets none of the wait states the compiler inserts between a matrix instruction and the VALU
// instruction that reads its result — round 6, call 17: a first form, `asm("v_max3_f32 ...")`, made the encoder output of the 96-frame test shapes,
// whose groups are one query tile, differ from run to run.)
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float rows4_vmax(float v) { return rows4_max(v); }
#ifndef WM_GEMM_SCHED
#define WM_GEMM_SCHED 3      // k_gemm_256p K-loop schedule (see the kernel); A/B builds: build.py --variant NAME -DWM_GEMM_SCHED=1|2
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Block barrier of the LDS-DMA ring kernels.  NOT __syncthreads(): an LDS-DMA in flight is a pending LDS write on the vector
// memory counter, so the fence inside __syncthreads() emits s_waitcnt vmcnt(0) and drains the whole ring at every step (the
// round-2 ISA of every ring kernel here had exactly that in front of its s_barrier: the "partial" waits never took effect).
// The wave's own fragment reads of the stage about to be refilled are retired first (lgkmcnt), the DMA pieces it needs were
// waited for with a counted vmcnt by the caller.
// The lgkmcnt wait is the BUILTIN, not inline asm: the compiler's wait-count pass then knows that every LDS read and scalar
// load issued so far is complete.  With an opaque asm wait it kept the kernel-argument s_loads of the epilogue "pending" through
// the whole K loop and, SMEM returning out of order, turned every counted LDS wait of the loop into lgkmcnt(0) — i.e. the MFMAs of
// a step waited for the fragment reads of the NEXT step that had just been issued.
__device__ __forceinline__ void ring_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0); vmcnt / expcnt fields left at their maxima (no wait)
    __builtin_amdgcn_s_barrier();
    // Both builtins are "no memory" intrinsics for the optimiser: without this (instruction-free) compiler barrier the plain LDS reads of
    // the stage that other waves' LDS-DMA has just landed could legally be hoisted above the s_barrier.
    asm volatile("" ::: "memory");
}

// BM = token rows per tile: 128 (default) or 64 (doubles the block count of the N=d GEMMs).
// NST = LDS ring stages: a stage (64 k of the X tile and of the 128-feature W tile) is filled by LDS-DMA; NST-1 stages
// are in flight while one is consumed, and the consumer waits with a partial vmcnt for ITS stage only.
// KS = K-split groups inside the block: group g (4 waves, its own LDS ring) walks the g-th part of the K loop and the
// partial tiles are added in group order through LDS at the end (deterministic).  One clip gives the N = d GEMMs one
// 64-row block per CU at most: KS = 2 doubles the waves per CU and halves the dependent K loop (FC2: 80 -> 40 steps).
// Everywhere else two resident blocks x 2 stages measured faster than one block x 3-4 stages
// (tests/microbench/enc_sweep.sh: occupancy beats ring depth).
template <int BM, int NST, int KS, class Ep>
__global__ void __launch_bounds__(256 * KS)
k_gemm_tiled(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, int K32, int tiles_m, int tiles_n, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    constexpr int XB = BM / 16 * 2;            // X fragments per stage (m-tiles x 2 k-tiles)
    constexpr int NB = XB + 16;                // + 16 W fragments (8 n-tiles x 2 k-tiles)
    constexpr int STAGE = NB * 1024;
    constexpr int LPW = NB / 4;                // LDS-DMA loads per wave per stage
    constexpr int MJ = BM / 32;                // token tiles per wave
    const int lane = threadIdx.x & 63;
    const int wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wa >> 2, w = wa & 3;
    const int wn = w >> 1, wm = w & 1;
    char* smem = smem_all + grp * (NST * STAGE);
    // XCD-aware remap: consecutive tiles of one weight panel stay on one XCD's L2
    int bid = blockIdx.x;
    const int nwg = tiles_m * tiles_n;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int tn = bid / tiles_m, tm = bid - tn * tiles_m;

    const int nkt = (K32 >> 1) / KS;           // k-steps of this group (the launcher checks divisibility)
    const int k0 = grp * nkt;
    const bf16_t* xg = X + (size_t)tm * (BM / 16) * K32 * 512 + lane * 8;
    const bf16_t* wg = W + (size_t)tn * 8 * K32 * 512 + lane * 8;

    auto stage_load = [&](int stage, int kt2) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int blk = w * LPW + i;               // 0..XB-1 X fragments, then 16 W fragments
            const bool isx = blk < XB;
            const int bb = isx ? blk : blk - XB;
            const int t = bb >> 1, kk = bb & 1;
            const bf16_t* src = (isx ? xg : wg) + ((size_t)t * K32 + (k0 + kt2) * 2 + kk) * 512;
            glds16(src, sb + blk * 1024);
        }
    };

    f32x4_t acc[4][MJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nkt) stage_load(st, st);

    for (int kt2 = 0; kt2 < nkt; ++kt2) {
        // stages kt2 .. min(nkt-1, kt2+NST-2) are outstanding, in order: wait until only the younger ones remain
        const int younger = min(nkt - 1, kt2 + NST - 2) - kt2;
        if (NST >= 4 && younger >= 2) wait_vmcnt<2 * LPW>();
        else if (NST >= 3 && younger >= 1) wait_vmcnt<LPW>();
        else wait_vmcnt<0>();
        ring_barrier();                        // every wave's part of stage kt2 landed; everyone is done with stage kt2-1
        if (kt2 + NST - 1 < nkt) stage_load((kt2 + NST - 1) % NST, kt2 + NST - 1);
        const bf16_t* xs = reinterpret_cast<const bf16_t*>(smem + (kt2 % NST) * STAGE);
        const bf16_t* ws = xs + XB * 512;
        bf16x8_t a[2][4], b[2][MJ];          // both halves' fragment reads first (see k_gemm_256)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[kk][i] = ld_frag(ws + (((wn * 4 + i) * 2 + kk) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < MJ; ++j) b[kk][j] = ld_frag(xs + (((wm * MJ + j) * 2 + kk) * 64 + lane) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16(a[kk][i], b[kk][j], acc[i][j]);
    }

    if (KS > 1) {                              // partial tiles of groups 1.. go through LDS, group 0 adds them in order
        __syncthreads();                       // the rings are no longer read
        float4* red = reinterpret_cast<float4*>(smem_all);
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j)
                    red[((((grp - 1) * 4 + w) * 4 + i) * MJ + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
        if (grp > 0) return;
        for (int g2 = 1; g2 < KS; ++g2)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j) {
                    const float4 p = red[((((g2 - 1) * 4 + w) * 4 + i) * MJ + j) * 64 + lane];
                    acc[i][j][0] += p.x; acc[i][j][1] += p.y; acc[i][j][2] += p.z; acc[i][j][3] += p.w;
                }
    }

    const int m0 = tm * BM + wm * (BM / 2) + (lane & 15), n0 = tn * GT_BN + wn * 64 + 4 * (lane >> 4);
    ep_tiles<4, MJ>(ep, m0, n0, acc);
}

template <int BM, int NST, int KS, class Ep>
static inline hipError_t launch_gemm_tiled_bm(hipStream_t st, const bf16_t* X, const bf16_t* W, int Mrows, int N, int K32, const Ep& ep)
{
    const int tiles_m = Mrows / BM, tiles_n = N / GT_BN;
    constexpr int lds = KS * NST * (BM / 16 * 2 + 16) * 1024;
    auto kern = k_gemm_tiled<BM, NST, KS, Ep>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256 * KS), lds, st, X, W, K32, tiles_m, tiles_n, ep);
    return hipGetLastError();
}

// =============================================================================================
// 256 x 256 tile for big batches (>= ~200 tiles): 8 waves as 2 (token halves) x 4 (feature quarters), a wave owns 128 tokens
// x 64 features = 8 x 4 MFMA tiles (128 accumulator registers).  Per 64-deep k-step a wave reads 24 fragments from LDS for 64
// MFMAs (the 128 x 128 kernel: 16 for 32) and the block brings 64 KiB in for 4x the math of a 128 x 128 tile, so each
// operand byte that crosses L2 -> LDS -> registers feeds twice the MFMAs.  Two 64 KiB stages filled by LDS-DMA: the loads of
// step t+1 go out right after the barrier that opens step t and have its 64 MFMAs per wave (~1 us with two waves per SIMD) to
// land.  Same packed operands, same k order, one accumulator per output: bit-identical to k_gemm_tiled<128, ., 1>.
// SQ counters that motivated it (profiles/r02_pmc_sq_bench_b32_before.md): the 128 x 128 kernel's waves sit in s_waitcnt /
// s_barrier 44-58 % of their cycles, LDS issue stalls 1 %, no bank conflicts — latency, not LDS bandwidth.
// =============================================================================================
template <class Ep>
__global__ void __launch_bounds__(512)
k_gemm_256(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, int K32, int tiles_m, int tiles_n, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 64 * 1024;            // 32 X fragments (16 token tiles x 2 k-tiles) then 32 W fragments
    const int lane = threadIdx.x & 63;
    const int wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wa >> 2, wn = wa & 3;
    int bid = blockIdx.x;
    const int nwg = tiles_m * tiles_n;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);       // XCD-aware: consecutive tiles of one weight panel share an L2
    const int tn = bid / tiles_m, tm = bid - tn * tiles_m;
    const int nkt = K32 >> 1;
    const bf16_t* xg = X + (size_t)tm * 16 * K32 * 512 + lane * 8;
    const bf16_t* wg = W + (size_t)tn * 16 * K32 * 512 + lane * 8;

    auto stage_load = [&](int stage, int kt2) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int blk = wa * 8 + i;                // 0..31 X fragments, 32..63 W fragments
            const bool isx = blk < 32;
            const int bb = isx ? blk : blk - 32;
            const int t = bb >> 1, kk = bb & 1;
            glds16((isx ? xg : wg) + ((size_t)t * K32 + kt2 * 2 + kk) * 512, sb + blk * 1024);
        }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    stage_load(0, 0);
    for (int kt2 = 0; kt2 < nkt; ++kt2) {
        wait_vmcnt<0>();
        ring_barrier();                        // stage kt2 landed for everyone; everyone is done reading the other buffer
        if (kt2 + 1 < nkt) stage_load((kt2 + 1) & 1, kt2 + 1);
        const bf16_t* xs = reinterpret_cast<const bf16_t*>(smem + (kt2 & 1) * STAGE);
        const bf16_t* ws = xs + 32 * 512;
        // both 32-deep halves of the step are requested from LDS before the first MFMA: the second half's reads land under
        // the first half's 32 MFMAs (PMC: the pipes were 30 % busy — with one wave pair per SIMD every fragment read was
        // an exposed LDS round trip in front of the MFMAs that needed it)
        bf16x8_t a[2][4], b[2][8];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[kk][i] = ld_frag(ws + (((wn * 4 + i) * 2 + kk) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) b[kk][j] = ld_frag(xs + (((wm * 8 + j) * 2 + kk) * 64 + lane) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the reads above the MFMAs: left alone, the scheduler sinks each ds_read to just in
                                               // front of its first use and waits lgkmcnt(0) there (ISA: R1 w0 M4 R1 w0 M4 ...)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma16(a[kk][i], b[kk][j], acc[i][j]);
    }
    const int m0 = tm * 256 + wm * 128 + (lane & 15), n0 = tn * 256 + wn * 64 + 4 * (lane >> 4);
    ep_tiles<4, 8>(ep, m0, n0, acc);
}

// =============================================================================================
// 256 x 256 tile, software-pipelined (round 3).  Same tile, wave layout, operands, k order and accumulators as k_gemm_256
// (bit-identical results); what changes is WHEN things happen:
//   * stages are 32 k deep (32 KiB: 16 X + 16 W fragments), NST of them in a ring; a stage is refilled as soon as its
//     fragments sit in registers and only a counted `s_waitcnt vmcnt` + raw s_barrier stands between steps, so NST-1 stages
//     (96 KiB) stay in flight across the barrier (k_gemm_256 drained the queue, vmcnt(0), at every step);
//   * fragments are double-buffered in registers: right after the barrier a wave requests the 12 fragments of step t+1
//     and then issues the 32 MFMAs of step t from the registers filled one step earlier — LDS latency and the LDS pipe's
//     transfer time (8 waves x 12 KiB per step) sit under the matrix pipe instead of in front of it.  PMC of the round-2
//     kernel: matrix pipes 29-33 % busy, waves parked in s_waitcnt / s_barrier half of their cycles.
//   * blockIdx -> tile: an XCD (32 CUs, one private L2) walks the tiles strip-major, so the ~32 blocks resident on it work on a
//     compact patch and share ~6 token panels and ~5 weight panels through its L2 (with the panel-major order every resident block
//     streamed its own token panel from the Infinity Cache).
// =============================================================================================
// WN = feature-wave columns: 4 -> 256 x 256 tile, 8 waves, one block per CU; 2 -> 256 tokens x 128 features, 4 waves (one per SIMD),
// 24 KiB stages, TWO blocks per CU: the epilogue of one block (HBM stores, GELU: ~a third of the one-block-per-CU kernel's time,
// during which its CU's matrix pipes idle) overlaps the K loop of the other, at 1.5x the L2 -> LDS bytes per flop.
// SW: the launch computes its tiles with the MFMA operands exchanged (tokens as the A operand): a lane then owns 4 consecutive TOKENS
// of one feature, the store unit of the V^T fragment layout (wm_epilogues.h: ep_tiles_swapped).  The launcher splits a GEMM whose
// epilogue wants that for some of its feature tiles into two launches over disjoint tile subsets: tile column index c of a launch
// is the matrix's feature tile (c / tn_take) * tn_period + tn_off + c % tn_take.  (One kernel with both K loops behind a block-uniform
// branch spilled accumulators inside the loops.)
template <int NST, int WN, bool SW, bool DBGK, class Ep>
__global__ void __launch_bounds__(128 * WN, 2)
k_gemm_256p(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, int K32, int tiles_m, int tiles_n, int PN, int persistent, int dbg_arg,
            int tn_period, int tn_take, int tn_off, Ep ep)
{
    // DBGK = false (every production launch): the measurement switches below are compile-time zero — as run-time tests they cut each K step
    // into ~8 basic blocks with a scalar branch in front of the DMA issue, the fragment reads and the MFMAs
    const int dbg = DBGK ? dbg_arg : 0;
    // dbg (WM_ENC_GEMM_DBG, measurement only): bit 0 skips the MFMAs, bit 1 the ring refills inside the K loop, bit 2 the epilogue,
    // bit 3 the fragment reads of the K loop (results are then wrong) — what is left shows which part bounds the kernel; bit 4
    // raises the wave priority around the MFMAs, bit 5 fills the ring of a persistent block's next tile AFTER the epilogue (results stay
    // right); bit 6 (launcher): V tiles feature-major like every other tile, one launch (results stay right); bit 7: residual GEMMs
    // with zeroed accumulators and the read-modify-write epilogue (results right, summation order of the few-clip kernels)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPIECE = 16 + 4 * WN;         // 16 X fragments (token tiles) then 4 WN W fragments (row tiles), one k-tile
    constexpr int STAGE = NPIECE * 1024;
    constexpr int LPW = NPIECE / (2 * WN);      // LDS-DMA pieces per wave per stage
    static_assert(NPIECE % (2 * WN) == 0, "pieces split evenly over the waves");
    const int lane = threadIdx.x & 63;
    const int wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wa / WN, wn = wa - wm * WN;
    const int NT = K32;                         // stages (even; the launcher checks)
    // Tile order.  Tiles are numbered strip-major: the feature tiles are cut into strips of PN columns and, inside a strip, tile
    // (tm, tn) has the number tm * PN + tn % PN — any 32 consecutive numbers are ~32 / PN token panels x PN weight panels, a compact
    // patch.  XCD x = blockIdx % 8 (observed placement; only speed depends on it) owns a contiguous eighth of the numbers:
    //  persistent (grid = 8 XCDs x 32 CUs x blocks per CU): the S blocks of an XCD take the numbers start + slot, + S, + 2 S, ...: at any
    //   time they work on S consecutive numbers and share ~32 / PN + PN operand panels through the XCD's L2;
    //  otherwise one tile per block, XCD x walking its eighth in order (bijective for any grid).
    // (Until call 12 of round 3 a patch was PM x PN tiles with PM * PN as close to 32 as the divisors allowed: for 5 feature tiles that
    // is 32 x 1 — 33 panels per 32 tiles, every token panel read 5 times from beyond the L2: the residual and QKV GEMMs pulled 4.1-4.5
    // TB/s through the fabric at L2 hit rates of 0.53 / 0.69, profiles/r03_pmc_l2_encoder_gemms.md.)
    const int T = tiles_m * tiles_n;
    int id, id_end, id_step;
    {
        const int xcd = blockIdx.x & 7, q = T >> 3, r = T & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        if (persistent) { id = start + (blockIdx.x >> 3); id_end = start + q + (xcd < r ? 1 : 0); id_step = gridDim.x >> 3; }
        else { id = start + (blockIdx.x >> 3); id_end = id + 1; id_step = 1; }
    }
    if (id >= id_end) return;
    // measurement (dbg >> 8 = n): every other block of an XCD starts n x 3.4 us late, so that the blocks are not all in their
    // epilogue at the same time (no gain: profiles/r03_encoder_gemm_parts.md)
    if ((dbg >> 8) && ((blockIdx.x >> 3) & 1))
        for (int i = 0; i < (dbg >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    const int strip = tiles_m * PN;
    auto tile_of = [&](int id_, int& tm_, int& tn_) {
        const int sb = id_ / strip, rem = id_ - sb * strip;
        tm_ = rem / PN;
        const int c = sb * PN + (rem - tm_ * PN);
        tn_ = (c / tn_take) * tn_period + tn_off + c % tn_take;
    };
    auto stage_load = [&](const bf16_t* xg, const bf16_t* wg, int kt) {
        char* sb = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int blk = wa * LPW + i;              // 0..15 X fragments, then the W fragments
            const bool isx = blk < 16;
            const int t = isx ? blk : blk - 16;
            glds16((isx ? xg : wg) + ((size_t)t * K32 + kt) * 512, sb + blk * 1024);
        }
    };
    auto frag_load = [&](int kt, bf16x8_t (&a)[4], bf16x8_t (&b)[8]) {
        const bf16_t* xs = reinterpret_cast<const bf16_t*>(smem + (kt % NST) * STAGE);
        const bf16_t* ws = xs + 16 * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = ld_frag(ws + ((wn * 4 + i) * 64 + lane) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = ld_frag(xs + ((wm * 8 + j) * 64 + lane) * 8);
    };

    int tm, tn;
    tile_of(id, tm, tn);
    const bf16_t* xg = X + (size_t)tm * 16 * K32 * 512 + lane * 8;
    const bf16_t* wg = W + (size_t)tn * (4 * WN) * K32 * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < NST; ++s)
        if (s < NT) stage_load(xg, wg, s);

    // residual GEMMs: the accumulators start as the residual tile (wm_epilogues.h: EpAccInit); dbg bit 7 keeps the classic epilogue
    constexpr bool kAccInit = EpAccInit<Ep>::value && !SW;
    const bool acc_init = kAccInit && !(dbg & 128);
    for (;;) {
        f32x4_t acc[4][8];
        const int m0 = tm * 256 + wm * 128 + (lane & 15), n0 = tn * (64 * WN) + wn * 64 + 4 * (lane >> 4);
        if (acc_init) {
            ep_acc_init<4, 8>(ep, m0, n0, acc);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }

        // stage 0 -> registers.  (A later tile of a persistent block: its first NST stages were requested before the previous tile's
        // epilogue; the stores of that epilogue are younger entries of the same counter, which only makes this wait conservative.)
        if (NT >= NST) wait_vmcnt<(NST - 1) * LPW>(); else wait_vmcnt<0>();
        ring_barrier();
        bf16x8_t a0[4], b0[8], a1[4], b1[8];
        frag_load(0, a0, b0);

        // one step: stage t is in (ac, bc); request stage t+1 into (an, bn), then the MFMAs of stage t
        auto step = [&](int t, bf16x8_t (&ac)[4], bf16x8_t (&bc)[8], bf16x8_t (&an)[4], bf16x8_t (&bn)[8]) {
            // stages issued so far: 0 .. min(NT-1, t+NST-1); stage t+1 must have landed: only the stages after it may be outstanding
            const int younger = min(NT - 1, t + NST - 1) - (t + 1);
            if (younger >= 3) wait_vmcnt<3 * LPW>();
            else if (younger == 2) wait_vmcnt<2 * LPW>();
            else if (younger == 1) wait_vmcnt<LPW>();
            else wait_vmcnt<0>();
            ring_barrier();                        // everyone's pieces of stage t+1 landed; everyone's reads of stage t (and older) are complete
            if (t + NST < NT && !(dbg & 2)) stage_load(xg, wg, t + NST);   // into the buffer of stage t: it lives in registers now
            if (!(dbg & 8)) frag_load(t + 1, an, bn);   // unconditional (after the last step it re-reads a stale buffer, unused): a branch here makes
                                                   // the compiler wait lgkmcnt(0) at the join, i.e. for THESE reads, in front of the MFMAs below
            __builtin_amdgcn_sched_barrier(0);     // keep the requests above the MFMAs
            if (!(dbg & 1)) {
                if (dbg & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = SW ? mfma16(bc[j], ac[i], acc[i][j]) : mfma16(ac[i], bc[j], acc[i][j]);
                if (dbg & 16) __builtin_amdgcn_s_setprio(0);
            }
        };
        static_assert(NST >= 3 && NST <= 5, "ring depth");
        if constexpr (NST == 4 && !DBGK) {
            // production form: the steady state (NST - 2 stages in flight behind the one needed, one refill per step) is a loop without a test
            // inside, the last four steps (nothing left to refill, the ring drains) are peeled with their waits as constants.  The generic
            // step() chooses its wait and its refill at run time: three scalar branches per step in front of the barrier.
            // WM_GEMM_SCHED (build-time; round 6).  The ISA of the form below without its trailing sched_barrier (rounds 3-5): the compiler sank 31 of a
            // step's 32 MFMAs below the NEXT step's wait + s_barrier (MFMAs touch no memory: nothing ordered them against the barrier), so every other
            // barrier interval held ONE MFMA and an exposed lgkmcnt(0) on the 12 fragment reads just issued, the following one 63 MFMAs.
            //  1: a sched_barrier closes each step — its MFMAs stay between its own barrier and the next;
            //  2: as 1, and the step's requests are dealt out in front of four groups of 8 MFMAs (one LDS-DMA piece + three fragment reads each)
            //     instead of 16 requests ahead of the first MFMA: the issue of the requests (vector-memory and LDS queues shared by 8 waves) runs
            //     in the shadow of the previous group's MFMAs.  Same operands, same accumulators, same order per accumulator: bit-identical.
            //  3: as 2 with the requests in front of the first three groups only (2 + 1 + 1 LDS-DMA pieces, 4 fragment reads each): the last group's
            //     8 MFMAs stand between the step's last fragment read and the lgkmcnt(0) of the next barrier;
            //  4: 2 with the wave priority raised over the MFMA groups (the other wave of the SIMD issues its requests around them).
            auto quarter = [&](int t, int q, bool refill, bf16x8_t (&ac)[4], bf16x8_t (&bc)[8], bf16x8_t (&an)[4], bf16x8_t (&bn)[8]) {
                const bf16_t* xs = reinterpret_cast<const bf16_t*>(smem + ((t + 1) % NST) * STAGE);
                const bf16_t* ws = xs + 16 * 512;
                auto piece = [&](int i) {
                    const int blk = wa * LPW + i;
                    const bool isx = blk < 16;
                    glds16((isx ? xg : wg) + ((size_t)(isx ? blk : blk - 16) * K32 + t + NST) * 512, smem + ((t + NST) % NST) * STAGE + blk * 1024);
                };
                if constexpr (WM_GEMM_SCHED == 3) {
                    if (refill) { if (q == 0) { piece(0); piece(1); } else if (q < 3) piece(q + 1); }
                    if (q == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) an[i] = ld_frag(ws + ((wn * 4 + i) * 64 + lane) * 8);
                    } else if (q < 3) {
#pragma unroll
                        for (int j = 4 * (q - 1); j < 4 * q; ++j) bn[j] = ld_frag(xs + ((wm * 8 + j) * 64 + lane) * 8);
                    }
                } else {
                    if (refill) piece(q);
                    an[q] = ld_frag(ws + ((wn * 4 + q) * 64 + lane) * 8);
                    bn[2 * q] = ld_frag(xs + ((wm * 8 + 2 * q) * 64 + lane) * 8);
                    bn[2 * q + 1] = ld_frag(xs + ((wm * 8 + 2 * q + 1) * 64 + lane) * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (WM_GEMM_SCHED == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 2 * q; j < 2 * q + 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = SW ? mfma16(bc[j], ac[i], acc[i][j]) : mfma16(ac[i], bc[j], acc[i][j]);
                if constexpr (WM_GEMM_SCHED == 4) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto steady = [&](int t, bf16x8_t (&ac)[4], bf16x8_t (&bc)[8], bf16x8_t (&an)[4], bf16x8_t (&bn)[8]) {
                wait_vmcnt<2 * LPW>();
                ring_barrier();
                if constexpr (WM_GEMM_SCHED >= 2 && LPW == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) quarter(t, q, true, ac, bc, an, bn);
                } else {
                    stage_load(xg, wg, t + NST);
                    frag_load(t + 1, an, bn);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][j] = SW ? mfma16(bc[j], ac[i], acc[i][j]) : mfma16(ac[i], bc[j], acc[i][j]);
                    if constexpr (WM_GEMM_SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto drain = [&](int t, auto younger, bf16x8_t (&ac)[4], bf16x8_t (&bc)[8], bf16x8_t (&an)[4], bf16x8_t (&bn)[8]) {
                wait_vmcnt<decltype(younger)::value * LPW>();
                ring_barrier();
                if constexpr (WM_GEMM_SCHED >= 2 && LPW == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) quarter(t, q, false, ac, bc, an, bn);      // (after the last step: a stale buffer, unused)
                } else {
                    frag_load(t + 1, an, bn);              // (after the last step: a stale buffer, unused)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][j] = SW ? mfma16(bc[j], ac[i], acc[i][j]) : mfma16(ac[i], bc[j], acc[i][j]);
                    if constexpr (WM_GEMM_SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
                }
            };
            int t = 0;
            for (; t < NT - NST; t += 2) {             // NT is even and >= NST (the launcher checks)
                steady(t, a0, b0, a1, b1);
                steady(t + 1, a1, b1, a0, b0);
            }
            drain(t, std::integral_constant<int, 2>{}, a0, b0, a1, b1);
            drain(t + 1, std::integral_constant<int, 1>{}, a1, b1, a0, b0);
            drain(t + 2, std::integral_constant<int, 0>{}, a0, b0, a1, b1);
            drain(t + 3, std::integral_constant<int, 0>{}, a1, b1, a0, b0);
        } else {
            for (int t = 0; t < NT; t += 2) {
                step(t, a0, b0, a1, b1);
                step(t + 1, a1, b1, a0, b0);
            }
        }
        // next tile of a persistent block: its first stages go out BEFORE this tile's epilogue (every wave has passed the last barrier
        // of the K loop: nobody reads the ring any more, except for the unused trailing request), so the ring fill — and the memory
        // latency in front of it — hides behind the epilogue's loads and stores
        const int next = id + id_step;
        const bool more = next < id_end;
        int tm2 = tm, tn2 = tn;
        if (more && !(dbg & 32)) {
            tile_of(next, tm2, tn2);
            xg = X + (size_t)tm2 * 16 * K32 * 512 + lane * 8;
            wg = W + (size_t)tn2 * (4 * WN) * K32 * 512 + lane * 8;
#pragma unroll
            for (int s = 0; s < NST; ++s)
                if (s < NT) stage_load(xg, wg, s);
        }
        if (!(dbg & 4)) {
            if constexpr (SW) ep_tiles_swapped<4, 8>(ep, m0, n0, acc);
            else if (acc_init) ep_tiles_init<4, 8>(ep, m0, n0, acc);
            else ep_tiles<4, 8>(ep, m0, n0, acc);
        } else if (acc[0][0][0] == 12345.678f) ep.store4(m0, n0, acc[1][1]);      // keeps the accumulators alive
        if (!more) break;
        if (dbg & 32) {                             // measurement: ring fill after the epilogue (the round-3 call-2 form)
            tile_of(next, tm2, tn2);
            xg = X + (size_t)tm2 * 16 * K32 * 512 + lane * 8;
            wg = W + (size_t)tn2 * (4 * WN) * K32 * 512 + lane * 8;
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int s = 0; s < NST; ++s)
                if (s < NT) stage_load(xg, wg, s);
        }
        id = next; tm = tm2; tn = tn2;
    }
}

// strip width of the tile order: the divisor PN of tiles_n that minimises the operand panels per 32 consecutive tiles, 32 / PN + PN
static inline int gemm256_strip(int tiles_n)
{
    const int forced = [] { const char* v = std::getenv("WM_ENC_GEMM_PN"); return v ? std::atoi(v) : 0; }();      // measurement (1 = the 32 x 1 patches of before)
    if (forced > 0 && tiles_n % forced == 0) return forced;
    int best = 1; float best_cost = 33.f;
    for (int pn = 1; pn <= tiles_n && pn <= 32; ++pn) {
        if (tiles_n % pn) continue;
        const float cost = 32.f / pn + pn;
        if (cost < best_cost) { best_cost = cost; best = pn; }
    }
    return best;
}

template <int NST, int WN, bool SW, class Ep>
static inline hipError_t launch_gemm_256p_sub(hipStream_t st, const bf16_t* X, const bf16_t* W, int K32, int tiles_m, int tiles_n, int tn_period, int tn_take,
                                              int tn_off, int dbg, const Ep& ep)
{
    constexpr int per_cu = WN == 4 ? 1 : 2;          // resident blocks per CU (LDS: NST stages of 16 + 4 WN KiB)
    constexpr int lds = NST * (16 + 4 * WN) * 1024;
    const int PN = gemm256_strip(tiles_n);
    // persistent grid (8 XCDs x 32 CUs x blocks per CU) once there are more tiles than that; WM_ENC_GEMM_PERSIST=0: one tile per block
    const int persist_env = [] { const char* v = std::getenv("WM_ENC_GEMM_PERSIST"); return v ? std::atoi(v) : 1; }();
    const int persistent = (persist_env && tiles_m * tiles_n > 256 * per_cu) ? 1 : 0;
    auto kern = dbg ? k_gemm_256p<NST, WN, SW, true, Ep> : k_gemm_256p<NST, WN, SW, false, Ep>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(persistent ? 256 * per_cu : tiles_m * tiles_n), dim3(128 * WN), lds, st, X, W, K32, tiles_m, tiles_n, PN, persistent, dbg,
                       tn_period, tn_take, tn_off, ep);
    return hipGetLastError();
}

template <int NST, int WN, class Ep>
static inline hipError_t launch_gemm_256p_nst(hipStream_t st, const bf16_t* X, const bf16_t* W, int K32, int tiles_m, int tiles_n, const Ep& ep)
{
    const int dbg = [] { const char* v = std::getenv("WM_ENC_GEMM_DBG"); return v ? std::atoi(v) : 0; }();
    // feature tiles the epilogue wants token-major (V of the attention projections): every `period` tiles, the last `period - plain`
    if constexpr (EpWantsSwap<Ep>::value) {
        int period = 0, plain = 0;
        if (!(dbg & 64) && ep_swap_split(ep, 64 * WN, period, plain) && tiles_n % period == 0) {
            const int reps = tiles_n / period;
            hipError_t e = launch_gemm_256p_sub<NST, WN, false>(st, X, W, K32, tiles_m, reps * plain, period, plain, 0, dbg, ep);
            if (e != hipSuccess) return e;
            return launch_gemm_256p_sub<NST, WN, true>(st, X, W, K32, tiles_m, reps * (period - plain), period, period - plain, plain, dbg, ep);
        }
    }
    return launch_gemm_256p_sub<NST, WN, false>(st, X, W, K32, tiles_m, tiles_n, tiles_n, tiles_n, 0, dbg, ep);
}

template <class Ep>
static inline hipError_t launch_gemm_256p(hipStream_t st, const bf16_t* X, const bf16_t* W, int Mrows, int N, int K32, const Ep& ep)
{
    const int nst = [] { const char* v = std::getenv("WM_ENC_GEMM_RING"); return v ? std::atoi(v) : 0; }();      // read per launch (sweeps); 0 = default
    const int wn = [] { const char* v = std::getenv("WM_ENC_GEMM_WN"); return v ? std::atoi(v) : 4; }();         // 2: 256 x 128 tiles, two blocks per CU
    if (wn == 2) {
        const int tiles_m = Mrows / 256, tiles_n = N / 128;
        if (nst == 2) return launch_gemm_256p_nst<3, 2>(st, X, W, K32, tiles_m, tiles_n, ep);     // (3 is the minimum ring)
        return launch_gemm_256p_nst<3, 2>(st, X, W, K32, tiles_m, tiles_n, ep);
    }
    const int tiles_m = Mrows / 256, tiles_n = N / 256;
    if (nst == 5) return launch_gemm_256p_nst<5, 4>(st, X, W, K32, tiles_m, tiles_n, ep);
    if (nst == 3) return launch_gemm_256p_nst<3, 4>(st, X, W, K32, tiles_m, tiles_n, ep);
    return launch_gemm_256p_nst<4, 4>(st, X, W, K32, tiles_m, tiles_n, ep);
}

template <class Ep>
static inline hipError_t launch_gemm_256(hipStream_t st, const bf16_t* X, const bf16_t* W, int Mrows, int N, int K32, const Ep& ep)
{
    const int tiles_m = Mrows / 256, tiles_n = N / 256;
    auto kern = k_gemm_256<Ep>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), 128 * 1024, st, X, W, K32, tiles_m, tiles_n, ep);
    return hipGetLastError();
}

template <class Ep>
static inline hipError_t launch_gemm_tiled(hipStream_t st, const bf16_t* X, const bf16_t* W, int Mrows, int N, int K32, const Ep& ep)
{
    const int big_nst = [] { const char* v = std::getenv("WM_ENC_GEMM_STAGES"); return v ? std::atoi(v) : 2; }();          // read per launch (sweeps)
    static const int small_ks = [] { const char* v = std::getenv("WM_ENC_GEMM_KSPLIT"); return v ? std::atoi(v) : 2; }();
    const int blocks128 = (Mrows / 128) * (N / GT_BN);
    static const int use256 = [] { const char* v = std::getenv("WM_ENC_GEMM_256"); return v ? std::atoi(v) : 1; }();
    const int use256p = [] { const char* v = std::getenv("WM_ENC_GEMM_256P"); return v ? std::atoi(v) : 1; }();   // 0: the round-2 two-stage kernel (read per launch: A/B inside one process)
    if (use256 && Mrows % 256 == 0 && N % 256 == 0 && K32 % 2 == 0 && K32 >= 4 && (Mrows / 256) * (N / 256) >= 200)
        return use256p ? launch_gemm_256p(st, X, W, Mrows, N, K32, ep) : launch_gemm_256(st, X, W, Mrows, N, K32, ep);
    // fewer than ~one block per CU with 128-row tiles: halve the tile to fill the chip and split K inside the block
    // tuning knob; 600 / 1000 (64-row tiles for the one-clip QKV / FC1 GEMMs too) measured 7.9 ms per encoder pass against 6.4
    static const int bm64_below = [] { const char* v = std::getenv("WM_ENC_BM64_BELOW"); return v ? std::atoi(v) : 200; }();
    if (blocks128 < bm64_below) {
        if (small_ks == 2 && (K32 >> 1) % 2 == 0 && (K32 >> 1) >= 8) return launch_gemm_tiled_bm<64, 3, 2>(st, X, W, Mrows, N, K32, ep);
        return launch_gemm_tiled_bm<64, 4, 1>(st, X, W, Mrows, N, K32, ep);
    }
    // experiment knob (off): 256-token tiles — a wave owns 128 tokens x 64 features (8 x 4 MFMA tiles, 128 accumulator
    // registers), a quarter fewer LDS reads per MFMA, but one wave per SIMD: measured 511 vs 591 TFLOP/s at 32 clips
    static const int bm256 = [] { const char* v = std::getenv("WM_ENC_GEMM_BM256"); return v ? std::atoi(v) : 0; }();
    if (bm256 && Mrows % 256 == 0 && (Mrows / 256) * (N / GT_BN) >= 512) {
        if (bm256 == 3) return launch_gemm_tiled_bm<256, 3, 1>(st, X, W, Mrows, N, K32, ep);
        return launch_gemm_tiled_bm<256, 2, 1>(st, X, W, Mrows, N, K32, ep);
    }
    if (big_nst == 3) return launch_gemm_tiled_bm<128, 3, 1>(st, X, W, Mrows, N, K32, ep);
    return launch_gemm_tiled_bm<128, 2, 1>(st, X, W, Mrows, N, K32, ep);
}

// =============================================================================================
// fp8 (OCP e4m3) MFMA GEMMs of the encoder (BASELINE configs[4]): out[m][n] = xs[m] * ws[n] * sum_k X8[m][k] W8[n][k] on
// v_mfma_f32_16x16x128_f8f6f4 — the CDNA4 fp8 form that runs at TWICE the bf16 MFMA rate (the 16x16x32 fp8 form of rounds 2-3 runs
// at the bf16 rate) — with zero scale operands (the compiler then emits the unscaled opcode; layout and semantics confirmed on the
// GPU by tests/microbench/mfma128_probe.hip, profiles/r04_call1_patches.md): half the operand bytes through L2 -> LDS -> registers
// AND half the matrix-pipe time per flop.
//   X8: the LayerNorm output of a token row, quantised to e4m3 with ONE fp32 scale per row (xs[m] = max|row| / 448): the
//       LayerNorm kernel has the whole row in one wave, so the scale costs one wave reduction.  Only the GEMMs whose operand
//       IS a LayerNorm output take this path (QKV, FC1, cross-K/V projection: 7/12 of the encoder's GEMM flops + the
//       projection); out-proj and FC2 read attention / GELU outputs whose row maximum is spread over blocks and stay bf16.
//   W8: e4m3 with one fp32 scale per output row (weights.quantize_rows_e4m3).
// Packed fp8 layout (both operands, ABI layout 7): [R/16][Kp/128][2 halves][64 lanes][16 B], Kp = K rounded up to 128 (zero-filled).
// Lane l of the MFMA holds row l & 15 and the 32 consecutive k = 32 (l >> 4) .. + 31 of a 128-k unit; half h of the unit is the
// [64 lanes][16 B] image of bytes 16 h .. + 15 of every lane.  One half = 1 KiB = one LDS-DMA instruction, lane-linear in LDS: a
// fragment is two conflict-free ds_read_b128.
// =============================================================================================
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
__device__ __forceinline__ f32x4_t mfma128_f8(i32x8_t a, i32x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);       // cbsz = blgp = 0: e4m3 x e4m3; no scales
}

__device__ __host__ __forceinline__ size_t f8k_index(int row, int k, int K128) {       // byte index of (row, k) in the packed fp8 layout
    const int kk = k & 127, b = kk & 31;
    return ((size_t)(row >> 4) * K128 + (k >> 7)) * 2048 + (size_t)(b >> 4) * 1024 + (size_t)(((row & 15) + 16 * (kk >> 5)) * 16 + (b & 15));
}

template <class Ep>
struct EpScaled {              // fp8 GEMM: dequantise the accumulator (token-row scale, then weight-row scale) in front of the epilogue
    Ep ep; const float* xs; const float* ws;
    // the wrapped epilogues (EpQKVEnc, EpPackedAct, EpCrossKV) only use pre.a: the weight scales ride in pre.b, the token scale in pre.i
    __device__ __forceinline__ EpPre pre(int m, int n) const {
        EpPre p = ep.pre(m, n);
        p.b = *reinterpret_cast<const float4*>(ws + n);
        p.i = __float_as_int(xs[m]);
        return p;
    }
    __device__ __forceinline__ void fin(int m, int n, f32x4_t v, const EpPre& p) const {
        const float s = __int_as_float(p.i);
        ep.fin(m, n, f32x4_t{(v[0] * s) * p.b.x, (v[1] * s) * p.b.y, (v[2] * s) * p.b.z, (v[3] * s) * p.b.w}, p);
    }
    __device__ __forceinline__ void store4(int m, int n, f32x4_t v) const { fin(m, n, v, pre(m, n)); }
};

// A wave's tiles of an fp8 GEMM: dequantise the accumulators in place (token-row scale, then weight-row scale: the order of
// EpScaled::fin), then the wave-level epilogue of the wrapped functor (wm_enc_epilogues.h) — bit-identical to the per-tile path.
template <int NI, int NJ, class E>
__device__ __forceinline__ void ep_tiles(const EpScaled<E>& e, int m0, int n0, f32x4_t (&acc)[NI][NJ])
{
    float xs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) xs[j] = e.xs[m0 + j * 16];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float4 w = *reinterpret_cast<const float4*>(e.ws + n0 + i * 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4_t& a = acc[i][j];
            a[0] = (a[0] * xs[j]) * w.x; a[1] = (a[1] * xs[j]) * w.y; a[2] = (a[2] * xs[j]) * w.z; a[3] = (a[3] * xs[j]) * w.w;
        }
    }
    ep_tiles<NI, NJ>(e.ep, m0, n0, acc);
}

__device__ __forceinline__ i32x8_t ld_f8k(const unsigned char* unit_lane) {      // unit base + lane * 16
    const uint4 lo = *reinterpret_cast<const uint4*>(unit_lane), hi = *reinterpret_cast<const uint4*>(unit_lane + 1024);
    return i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}

// (64 or 128 tokens) x 128 features, 4 waves (wn = wave >> 1: 64 features, wm = wave & 1: BM / 2 tokens); a stage = one 128-k unit of
// every operand tile (BM / 16 + 8 units of 2 KiB), NST-deep LDS-DMA ring, one barrier per 128 k.
template <int BM, int NST, class Ep>
__global__ void __launch_bounds__(256)
k_gemm_f8k(const unsigned char* __restrict__ X, const unsigned char* __restrict__ W, int K128, int tiles_m, int tiles_n, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XB = BM / 16;                // X units per stage
    constexpr int NB = XB + 8;                 // + 8 W units (128 features)
    constexpr int STAGE = NB * 2048;
    constexpr int LPW = NB * 2 / 4;            // 1-KiB pieces per wave per stage
    constexpr int MJ = BM / 32;
    static_assert((NB * 2) % 4 == 0 && NST >= 2 && NST <= 4, "stage must split evenly over 4 waves");
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = w >> 1, wm = w & 1;
    int bid = blockIdx.x;
    const int nwg = tiles_m * tiles_n;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int tn = bid / tiles_m, tm = bid - tn * tiles_m;
    const unsigned char* xg = X + (size_t)tm * XB * K128 * 2048 + lane * 16;
    const unsigned char* wg = W + (size_t)tn * 8 * K128 * 2048 + lane * 16;

    auto stage_load = [&](int stage, int kt) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int pc = w * LPW + i;                // piece = (unit, half)
            const int u = pc >> 1, h = pc & 1;
            const bool isx = u < XB;
            const int t = isx ? u : u - XB;
            const unsigned char* src = (isx ? xg : wg) + ((size_t)t * K128 + kt) * 2048 + h * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + pc * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[4][MJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < K128) stage_load(st, st);

    for (int kt = 0; kt < K128; ++kt) {
        const int younger = min(K128 - 1, kt + NST - 2) - kt;
        if (NST >= 4 && younger >= 2) wait_vmcnt<2 * LPW>();
        else if (NST >= 3 && younger >= 1) wait_vmcnt<LPW>();
        else wait_vmcnt<0>();
        ring_barrier();
        if (kt + NST - 1 < K128) stage_load((kt + NST - 1) % NST, kt + NST - 1);
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(smem + (kt % NST) * STAGE) + lane * 16;
        const unsigned char* ws = xs + XB * 2048;
        i32x8_t a[4], b[MJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = ld_f8k(ws + (wn * 4 + i) * 2048);
#pragma unroll
        for (int j = 0; j < MJ; ++j) b[j] = ld_f8k(xs + (wm * MJ + j) * 2048);
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mfma128_f8(a[i], b[j], acc[i][j]);
    }
    const int m0 = tm * BM + wm * (BM / 2) + (lane & 15), n0 = tn * GT_BN + wn * 64 + 4 * (lane >> 4);
    ep_tiles<4, MJ>(ep, m0, n0, acc);
}

// 256 x 256 tile, 8 waves (2 x 4), wave = 128 tokens x 64 features.  A 128-k stage is 64 KiB (16 X + 16 W units): TWO stages — stage
// t + 1 lands in the buffer stage t - 1 was read from while the 32 MFMAs per wave of stage t run (1024 matrix-pipe cycles per wave, two
// waves per SIMD: 2048 cycles per step against ~64 KiB of L2 -> LDS fill per CU).  The token fragments of a step are read in two halves,
// the second behind the first half's MFMAs.  Strip-major XCD tile order as k_gemm_256p.
template <class Ep>
__global__ void __launch_bounds__(512)
k_gemm_f8k_256(const unsigned char* __restrict__ X, const unsigned char* __restrict__ W, int K128, int tiles_m, int tiles_n, int PN, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 64 * 1024, LPW = 8;
    const int lane = threadIdx.x & 63;
    const int wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wa >> 2, wn = wa & 3;
    int tm, tn;
    {
        const int nwg = tiles_m * tiles_n, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        const int strip = tiles_m * PN, sb = id / strip, rem = id - sb * strip;
        tm = rem / PN; tn = sb * PN + (rem - tm * PN);
    }
    const unsigned char* xg = X + (size_t)tm * 16 * K128 * 2048 + lane * 16;
    const unsigned char* wg = W + (size_t)tn * 16 * K128 * 2048 + lane * 16;

    auto stage_load = [&](int stage, int kt) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int pc = wa * LPW + i;               // pieces 0..31: the 16 X units' halves, 32..63: the W units'
            const int u = pc >> 1, h = pc & 1;
            const bool isx = u < 16;
            const int t = isx ? u : u - 16;
            const unsigned char* src = (isx ? xg : wg) + ((size_t)t * K128 + kt) * 2048 + h * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + pc * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    stage_load(0, 0);
    for (int kt = 0; kt < K128; ++kt) {
        wait_vmcnt<0>();                               // this wave's pieces of stage kt (the only stage in flight)
        ring_barrier();                                // everyone's pieces landed; everyone's reads of stage kt - 1 are complete
        if (kt + 1 < K128) stage_load((kt + 1) & 1, kt + 1);
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(smem + (kt & 1) * STAGE) + lane * 16;
        const unsigned char* ws = xs + 16 * 2048;
        i32x8_t a[4], b0[4], b1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = ld_f8k(ws + (wn * 4 + i) * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = ld_f8k(xs + (wm * 8 + j) * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) b1[j] = ld_f8k(xs + (wm * 8 + 4 + j) * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mfma128_f8(a[i], b0[j], acc[i][j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][4 + j] = mfma128_f8(a[i], b1[j], acc[i][4 + j]);
    }
    const int m0 = tm * 256 + wm * 128 + (lane & 15), n0 = tn * 256 + wn * 64 + 4 * (lane >> 4);
    ep_tiles<4, 8>(ep, m0, n0, acc);
}

// The same tile with the two token halves of the block (waves 0-3: wm = 0, waves 4-7: wm = 1; every SIMD hosts one wave of each) running
// HALF A STEP APART: a step is two phases, each closed by a block barrier; a wave reads the 12 fragments of its stage into registers in one
// phase and issues its 32 MFMAs in the next, and the other half does the opposite — while one wave of a SIMD waits for the LDS (the lock-step
// kernel above leaves the matrix pipes idle for the ~770 LDS cycles of every step: 192 KiB of fragment reads per block and step) the other
// keeps that SIMD's matrix pipe busy.  Phase p:  wm = 0 reads stage p / 2 (p even) and multiplies it (p odd); wm = 1 reads in the odd phases
// and multiplies in the even ones.  LDS-DMA: the pieces of stage s are issued by every wave in phase 2 s - 2 (its buffer held stage s - 2,
// last read by wm = 1 in phase 2 s - 3) and waited for (vmcnt(0)) before the barrier that closes phase 2 s - 1; wm = 0 reads them in phase
// 2 s, wm = 1 in phase 2 s + 1.  Both halves pass 2 K128 + 2 barriers.
template <class Ep>
__global__ void __launch_bounds__(512)
k_gemm_f8k_256s(const unsigned char* __restrict__ X, const unsigned char* __restrict__ W, int K128, int tiles_m, int tiles_n, int PN, Ep ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 64 * 1024, LPW = 8;
    const int lane = threadIdx.x & 63;
    const int wa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wa >> 2, wn = wa & 3;
    int tm, tn;
    {
        const int nwg = tiles_m * tiles_n, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        const int strip = tiles_m * PN, sb = id / strip, rem = id - sb * strip;
        tm = rem / PN; tn = sb * PN + (rem - tm * PN);
    }
    const unsigned char* xg = X + (size_t)tm * 16 * K128 * 2048 + lane * 16;
    const unsigned char* wg = W + (size_t)tn * 16 * K128 * 2048 + lane * 16;
    auto stage_load = [&](int kt) {
        char* sb = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int pc = wa * LPW + i;
            const int u = pc >> 1, h = pc & 1;
            const bool isx = u < 16;
            const int t = isx ? u : u - 16;
            const unsigned char* src = (isx ? xg : wg) + ((size_t)t * K128 + kt) * 2048 + h * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + pc * 1024), 16, 0, 0);
        }
    };
    i32x8_t a[4], b[8];
    auto read_frags = [&](int kt) {
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(smem + (kt & 1) * STAGE) + lane * 16;
        const unsigned char* ws = xs + 16 * 2048;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = ld_f8k(ws + (wn * 4 + i) * 2048);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = ld_f8k(xs + (wm * 8 + j) * 2048);
    };
    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mfma128_f8(a[i], b[j], acc[i][j]);
    };
    const int NT = K128;
    stage_load(0);
    wait_vmcnt<0>();
    ring_barrier();                                    // stage 0 has landed
    if (wm == 0) {
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) stage_load(t + 1);         // phase 2 t
            read_frags(t);
            ring_barrier();                            // (lgkmcnt(0) inside: the fragments are in registers)
            mfmas();                                   // phase 2 t + 1
            wait_vmcnt<0>();
            ring_barrier();
        }
        ring_barrier();                                // phase 2 NT: the other half's last products
    } else {
        if (NT > 1) stage_load(1);                     // phase 0
        ring_barrier();
        for (int t = 0; t < NT; ++t) {
            read_frags(t);                             // phase 2 t + 1
            wait_vmcnt<0>();
            ring_barrier();
            if (t + 2 < NT) stage_load(t + 2);         // phase 2 t + 2
            mfmas();
            ring_barrier();
        }
    }
    const int m0 = tm * 256 + wm * 128 + (lane & 15), n0 = tn * 256 + wn * 64 + 4 * (lane >> 4);
    ep_tiles<4, 8>(ep, m0, n0, acc);
}

template <int BM, int NST, class Ep>
static inline hipError_t launch_gemm_f8_bm(hipStream_t st, const unsigned char* X, const unsigned char* W, int Mrows, int N, int K128, const Ep& ep)
{
    const int tiles_m = Mrows / BM, tiles_n = N / GT_BN;
    constexpr int lds = NST * (BM / 16 + 8) * 2048;
    auto kern = k_gemm_f8k<BM, NST, Ep>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, st, X, W, K128, tiles_m, tiles_n, ep);
    return hipGetLastError();
}

// K128 = 128-k units per row (K rounded up to 128)
template <class Ep>
static inline hipError_t launch_gemm_f8(hipStream_t st, const unsigned char* X, const unsigned char* W, int Mrows, int N, int K128, const Ep& ep)
{
    static const int use256 = [] { const char* v = std::getenv("WM_ENC_GEMM_256"); return v ? std::atoi(v) : 1; }();
    if (use256 && Mrows % 256 == 0 && N % 256 == 0 && (Mrows / 256) * (N / 256) >= 200) {
        // WM_F8_STAGGER=1: the form with the two token halves half a step apart.  Measured SLOWER (profiles/r04_fp8_encoder.md: FC1 531 vs
        // 519 us, QKV 443 vs 406 us per launch at 32 clips): the K loop is bound by the L2 -> LDS fill (64 KiB per step and CU at ~66 GB/s
        // = one step's MFMA time), not by the exposed fragment reads the stagger hides; the extra barrier per step only costs.  Kept as a knob.
        const int stagger = [] { const char* v = std::getenv("WM_F8_STAGGER"); return v ? std::atoi(v) : 0; }();
        auto kern = stagger ? k_gemm_f8k_256s<Ep> : k_gemm_f8k_256<Ep>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((Mrows / 256) * (N / 256)), dim3(512), 128 * 1024, st, X, W, K128, Mrows / 256, N / 256, gemm256_strip(N / 256), ep);
        return hipGetLastError();
    }
    // few clips: 24 KiB (64 tokens) / 32 KiB (128 tokens) stages, two resident blocks per CU
    if ((Mrows / 128) * (N / GT_BN) < 200) return launch_gemm_f8_bm<64, 3>(st, X, W, Mrows, N, K128, ep);
    return launch_gemm_f8_bm<128, 2>(st, X, W, Mrows, N, K128, ep);
}

// =============================================================================================
// Encoder self-attention (non-causal, S keys): flash-style, one wave = 16 queries, 32 keys per step.
//   S^T = K Q^T   (A = K rows, B = Q^T; a lane then owns 8 scores of ONE query -> softmax reduces over
//                  2 xor-shuffles), P stays in registers and feeds O^T = V^T P^T directly: the MFMA
//   k-slot <-> key assignment is permuted identically in the V^T fragment and the P fragment, so no
//   cross-lane movement is needed.  q is pre-scaled; fp32 softmax; P rounded to bf16.
// =============================================================================================
// QT x 16 queries per wave: every K / V^T fragment is reused by QT query tiles; G = query tiles per softmax phase group; OCC = waves per
// SIMD the register allocation aims at (3: <= 168 VGPRs, three resident blocks per CU)
template <int QT, int G = (QT >= 2 ? 2 : 1), int OCC = 2>
__global__ void __launch_bounds__(256, OCC)
k_flash_enc(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kt, const bf16_t* __restrict__ Vf,
            bf16_t* __restrict__ out, int S, int Spad, int H, int K32out)
{
    // The 4 waves of a block share the K / V of their (batch, head): a 64-key step = 8 K fragments + 8 V^T fragments
    // (16 KiB) is brought in ONCE per block by LDS-DMA (each wave issues 4 of the 16 KiB-sized loads) into a 3-stage
    // ring — two steps are in flight while one is consumed — and read back with lane-linear ds_read_b128.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = 3, STAGE = 16 * 1024;
    constexpr float kLog2e = 1.4426950408889634f;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int hd = blockIdx.y, b = blockIdx.z;
    const size_t bh = (size_t)b * H + hd;
    const int q0 = (blockIdx.x * 4 + w) * 16 * QT;
    const bf16_t* kbase = Kt + bh * Spad * 64;
    const bf16_t* vbase = Vf + bh * 64 * Spad;
    const int nsteps = (S + 63) >> 6;

    auto stage_load = [&](int stage, int step) {
        char* sb = smem + stage * STAGE;
        const int kb = step * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = 2 * w + i;                                   // K fragment f: key tile f >> 1, dims (f & 1) * 32 ..
            glds16(kbase + (size_t)(kb + (f >> 1) * 16 + c) * 64 + (f & 1) * 32 + g * 8, sb + f * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = 2 * w + i;                                   // V^T fragment f of the step: contiguous in vfrag layout
            glds16(vbase + ((size_t)(kb >> 5) * 256 + f * 64 + lane) * 8, sb + 8192 + f * 1024);
        }
    };
    stage_load(0, 0);
    if (nsteps > 1) stage_load(1, 1);

    bf16x8_t qb[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const bf16_t* qp = Q + (bh * Spad + q0 + t * 16 + c) * 64 + g * 8;
        qb[t][0] = ld_frag(qp); qb[t][1] = ld_frag(qp + 32);
    }
    float m_run[QT], l_run[QT];
    f32x4_t o[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -INFINITY; l_run[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // one 32-key half step of the wave's QT query tiles: 4 K fragments (two key tiles x two 32-dim halves) and 4 V^T fragments
    struct KF { bf16x8_t a00, a01, a10, a11; };
    struct VF { bf16x8_t va[4]; };
    auto read_k = [&](KF& f, int i, int hh) {
        const bf16_t* sb = reinterpret_cast<const bf16_t*>(smem + (i % NST) * STAGE);
        f.a00 = ld_frag(sb + ((4 * hh + 0) * 64 + lane) * 8); f.a01 = ld_frag(sb + ((4 * hh + 1) * 64 + lane) * 8);
        f.a10 = ld_frag(sb + ((4 * hh + 2) * 64 + lane) * 8); f.a11 = ld_frag(sb + ((4 * hh + 3) * 64 + lane) * 8);
    };
    auto read_v = [&](VF& f, int i, int hh) {
        const bf16_t* sb = reinterpret_cast<const bf16_t*>(smem + (i % NST) * STAGE);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) f.va[dt] = ld_frag(sb + 4096 + ((4 * hh + dt) * 64 + lane) * 8);
    };
    struct SC { f32x4_t s0[G], s1[G]; };            // the scores of one group of G query tiles against one half step's 32 keys
    auto scores = [&](const KF& f, int t0, SC& sc) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            sc.s0[u] = f32x4_t{0.f, 0.f, 0.f, 0.f}; sc.s1[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            sc.s0[u] = mfma16(f.a00, qb[t0 + u][0], sc.s0[u]); sc.s0[u] = mfma16(f.a01, qb[t0 + u][1], sc.s0[u]);
            sc.s1[u] = mfma16(f.a10, qb[t0 + u][0], sc.s1[u]); sc.s1[u] = mfma16(f.a11, qb[t0 + u][1], sc.s1[u]);
        }
    };
    // mask (last keys), softmax arithmetic, ONE rescale decision, PV MFMAs of the group
    auto soft_pv = [&](const VF& f, int t0, int kb, SC& sc) {
        const bool tail = kb + 32 > S;
        if (tail) {
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kb + 4 * g + r >= S) sc.s0[u][r] = -INFINITY;
                    if (kb + 16 + 4 * g + r >= S) sc.s1[u][r] = -INFINITY;
                }
        }
        float alpha[G];
        bf16x8_t pb[G];
        bool any_rescale = false;
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int t = t0 + u;
#if WM_FLASH_VALU
            // Round 6 (ISA of the form in the #else branch: 120 max instructions per 64-key step and wave, 15 per tile and half — fmaxf under
            // IEEE mode canonicalises each operand with a v_max x, x first, and MFMA results / permlane outputs are never "known canonical").
            // The scores are finite or -inf, never NaN: with -fno-honor-nans the maxima are 3 v_max3_f32 + v_max_f32 (+ 3 across lanes and against the running maximum), same values.
            float mx = vmax2(vmax2(vmax2(sc.s0[u][0], sc.s0[u][1]), vmax2(sc.s0[u][2], sc.s0[u][3])), vmax2(vmax2(sc.s1[u][0], sc.s1[u][1]), vmax2(sc.s1[u][2], sc.s1[u][3])));
            mx = rows4_vmax(mx);
            const float m_new = vmax2(m_run[t], mx);
            // exp(s - m) = 2^(s log2e - m log2e): one fma + v_exp_f32 per score; the fmas as v_pk_fma_f32 (two scores per instruction: an
            // accumulator's 4 values sit in an aligned register quad) — the same fma per score, half the issue slots
            const float nml = -m_new * kLog2e;
            alpha[u] = __builtin_amdgcn_exp2f(fmaf(m_run[t], kLog2e, nml));
            float p0[4], p1[4], rs = 0.f;
            const f32x2_t l2 = {kLog2e, kLog2e}, n2 = {nml, nml};
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2_t e0 = __builtin_elementwise_fma(f32x2_t{sc.s0[u][r], sc.s0[u][r + 1]}, l2, n2);
                const f32x2_t e1 = __builtin_elementwise_fma(f32x2_t{sc.s1[u][r], sc.s1[u][r + 1]}, l2, n2);
                p0[r] = __builtin_amdgcn_exp2f(e0[0]); p0[r + 1] = __builtin_amdgcn_exp2f(e0[1]);
                p1[r] = __builtin_amdgcn_exp2f(e1[0]); p1[r + 1] = __builtin_amdgcn_exp2f(e1[1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) rs += p0[r] + p1[r];
#else
            float mx = fmaxf(fmaxf(fmaxf(sc.s0[u][0], sc.s0[u][1]), fmaxf(sc.s0[u][2], sc.s0[u][3])), fmaxf(fmaxf(sc.s1[u][0], sc.s1[u][1]), fmaxf(sc.s1[u][2], sc.s1[u][3])));
            mx = rows4_max(mx);
            const float m_new = fmaxf(m_run[t], mx);
            // exp(s - m) = 2^(s log2e - m log2e): one fma + v_exp_f32 per score (the sub / mul / exp form was a fifth more VALU issue
            // in a kernel bound by it: ~500 VALU + 72 exp against 64 MFMAs per 64-key step and wave)
            const float nml = -m_new * kLog2e;
            alpha[u] = __builtin_amdgcn_exp2f(fmaf(m_run[t], kLog2e, nml));
            float p0[4], p1[4], rs = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p0[r] = __builtin_amdgcn_exp2f(fmaf(sc.s0[u][r], kLog2e, nml)); p1[r] = __builtin_amdgcn_exp2f(fmaf(sc.s1[u][r], kLog2e, nml));
                rs += p0[r] + p1[r];
            }
#endif
            l_run[t] = fmaf(l_run[t], alpha[u], rs);          // this lane's 8 keys per step only: the four g-lanes of a row meet once, at the end
            m_run[t] = m_new;
            uint4 pw;
            pw.x = pack_bf2(p0[0], p0[1]); pw.y = pack_bf2(p0[2], p0[3]);
            pw.z = pack_bf2(p1[0], p1[1]); pw.w = pack_bf2(p1[2], p1[3]);
            pb[u] = __builtin_bit_cast(bf16x8_t, pw);
            any_rescale = any_rescale || (alpha[u] != 1.0f);
        }
        // lazy rescale: once the running maxima have settled alpha is exactly 1 for every query of the wave and the 16 multiplies
        // per tile are skipped (one wave-uniform branch per group; x * 1.0f == x, results unchanged)
        if (__builtin_amdgcn_ballot_w64(any_rescale) != 0ull) {
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { o[t0 + u][dt][0] *= alpha[u]; o[t0 + u][dt][1] *= alpha[u]; o[t0 + u][dt][2] *= alpha[u]; o[t0 + u][dt][3] *= alpha[u]; }
        }
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[t0 + u][dt] = mfma16(f.va[dt], pb[u], o[t0 + u][dt]);
    };
    auto half = [&](const KF& kf, const VF& vf, int kb) {
        // The query tiles of the wave go through each phase in GROUPS — the score MFMAs of the group, then its softmax arithmetic,
        // ONE rescale decision, then its PV MFMAs — so the tiles' dependent chains (MFMA -> row max across lanes -> exp -> row sum
        // -> convert -> MFMA) interleave.  Tile by tile with a rescale branch inside (round 2), every tile was its own scheduling
        // region and its chain latency was exposed: the kernel sat at ~37 % of its VALU bound.
#pragma unroll
        for (int t0 = 0; t0 < QT; t0 += G) {
            SC sc;
            scores(kf, t0, sc);
            soft_pv(vf, t0, kb, sc);
        }
    };
#if WM_FLASH_VALU >= 2
    // Round 6.  K fragments one half step ahead in a second register set (the GEMM's trick), V^T fragments requested at the start of their half step
    // (first use behind the group's score MFMAs and softmax), ONE barrier per step at its middle:
    //   read K (i, 1), V (i, 0) | arithmetic (i, 0) | wait stage i+1, barrier | refill stage i+2, read K (i+1, 0), V (i, 1) | arithmetic (i, 1)
    // The barrier of step i says: stage i+1 landed for everyone, and every wave is done with stage i-1 (its last reads, V (i-1, 1), retired by the
    // lgkmcnt(0) of ring_barrier at the latest) — the buffer refilled next, stage (i+2) % 3.  Before: every half step began with 8 fragment reads
    // and an lgkmcnt(0) in front of its first MFMA (an exposed LDS round trip, twice per step and wave): 613 -> 521 us per launch at 32 clips
    // (profiles/r06_encoder_prefill.md; measured on top and not kept: the next group's score MFMAs issued ahead of the current group's softmax, +0.5 %).
    {
        if (nsteps > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();
        KF k0, k1;
        VF v;
        read_k(k0, 0, 0);
        // every step but the last has both halves (its second half starts below S): no test between the requests and the arithmetic
        for (int i = 0; i + 1 < nsteps; ++i) {
            read_k(k1, i, 1); read_v(v, i, 0);
            __builtin_amdgcn_sched_barrier(0);                        // (left alone the reads sink towards their first use)
            half(k0, v, i * 64);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // stage i+1, requested a step ago (stage i+2 goes out below)
            ring_barrier();
            if (i + 2 < nsteps) stage_load((i + 2) % NST, i + 2);
            read_k(k0, i + 1, 0); read_v(v, i, 1);
            __builtin_amdgcn_sched_barrier(0);
            half(k1, v, i * 64 + 32);
        }
        const int il = nsteps - 1;
        const bool second = il * 64 + 32 < S;                         // block-uniform
        if (second) read_k(k1, il, 1);
        read_v(v, il, 0);
        half(k0, v, il * 64);
        if (second) {
            read_v(v, il, 1);
            half(k1, v, il * 64 + 32);
        }
    }
#else
    for (int i = 0; i < nsteps; ++i) {
        // stage i has landed when at most the 4 loads of stage i+1 are still outstanding (vmcnt counts in order)
        if (i + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();                                                // everyone's part landed; everyone is done with stage i-1
        if (i + 2 < nsteps) stage_load((i + 2) % NST, i + 2);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int kb = i * 64 + hh * 32;
            if (kb >= S) break;                                        // block-uniform
            KF kf; VF vf;
            read_k(kf, i, hh); read_v(vf, i, hh);
            half(kf, vf, kb);
        }
    }
#endif
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float inv = 1.0f / rows4_sum(l_run[t]);
        const int row = b * Spad + q0 + t * 16 + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 u;
            u.x = pack_bf2(o[t][dt][0] * inv, o[t][dt][1] * inv);
            u.y = pack_bf2(o[t][dt][2] * inv, o[t][dt][3] * inv);
            *reinterpret_cast<uint2*>(out + packed_index(row, hd * 64 + dt * 16 + 4 * g, K32out)) = u;
        }
    }
}

// =============================================================================================
// LayerNorm over many rows -> packed bf16.  One block per 16-row group of the packed layout; lane (r, g) of wave w holds, for the
// k-tiles f = w, w + 4, ..., the 8 values k = 32 f + 8 g .. + 7 of row r — the lane layout of the MFMA fragment itself, so the
// normalised tile leaves as ONE lane-linear 16-byte store per lane: a wave instruction writes a whole 1-KiB fragment.  (The
// one-wave-per-row form of rounds 1-2 scattered a row over 40 fragments, 16 bytes at a 256-byte stride and 32 requests per store
// instruction: 119 us per launch at 32 clips = 3.1 TB/s for a kernel that only streams, profiles/r03_kernel_trace_bench_b32.md.)
// Row statistics: each lane sums its <= 80 values, rows4_sum adds the four g-lanes of a row inside the wave, the four waves' partials
// meet in LDS (two block barriers: mean, then the centred second moment — the two-pass form of the reference LayerNorm).
// =============================================================================================
template <int LN16_MAXF>                           // k-tiles per wave: d <= 4 * 32 * LN16_MAXF
__global__ void __launch_bounds__(256)
k_enc_ln(const float* __restrict__ src, const float* __restrict__ gamma, const float* __restrict__ beta,
         bf16_t* __restrict__ out_p, int K32, int d, int M)
{
    __shared__ float part[2][4][16];
    // gamma / beta staged in LDS (one coalesced pass per block, under the row loads): read from global inside the output loop, every
    // k-tile's parameter loads put an s_waitcnt vmcnt(0) in front of its store — which also waits for the previous k-tile's store to be
    // acknowledged: 10 dependent round trips per wave (ISA of round 3)
    __shared__ __attribute__((aligned(16))) float gbs[2][LN16_MAXF * 128];
    for (int t = threadIdx.x; t < (d >> 2); t += 256) {
        reinterpret_cast<float4*>(gbs[0])[t] = reinterpret_cast<const float4*>(gamma)[t];
        reinterpret_cast<float4*>(gbs[1])[t] = reinterpret_cast<const float4*>(beta)[t];
    }
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int w = threadIdx.x >> 6;
    const int row = blockIdx.x * 16 + r;            // M % 16 == 0 (rows are clips x Spad)
    const float* sp = src + (size_t)row * d + 8 * g;
    float4 v[LN16_MAXF][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN16_MAXF; ++i) {
        const int f = w + 4 * i;
        if (f < K32) {
            v[i][0] = *reinterpret_cast<const float4*>(sp + f * 32);
            v[i][1] = *reinterpret_cast<const float4*>(sp + f * 32 + 4);
        } else {
            v[i][0] = make_float4(0.f, 0.f, 0.f, 0.f); v[i][1] = v[i][0];
        }
    }
#pragma unroll
    for (int i = 0; i < LN16_MAXF; ++i)
        s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    s = rows4_sum(s);
    if (g == 0) part[0][w][r] = s;
    __syncthreads();
    const float mean = ((part[0][0][r] + part[0][1][r]) + (part[0][2][r] + part[0][3][r])) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN16_MAXF; ++i)
        if (w + 4 * i < K32) {
            const float a0 = v[i][0].x - mean, a1 = v[i][0].y - mean, a2 = v[i][0].z - mean, a3 = v[i][0].w - mean;
            const float a4 = v[i][1].x - mean, a5 = v[i][1].y - mean, a6 = v[i][1].z - mean, a7 = v[i][1].w - mean;
            q += ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((a4 * a4 + a5 * a5) + (a6 * a6 + a7 * a7));
        }
    q = rows4_sum(q);
    if (g == 0) part[1][w][r] = q;
    __syncthreads();
    const float rstd = rsqrtf(((part[1][0][r] + part[1][1][r]) + (part[1][2][r] + part[1][3][r])) / (float)d + 1e-5f);
    bf16_t* op = out_p + (size_t)blockIdx.x * K32 * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < LN16_MAXF; ++i) {
        const int f = w + 4 * i;
        if (f < K32) {
            const float4 g0 = *reinterpret_cast<const float4*>(gbs[0] + f * 32 + 8 * g), g1 = *reinterpret_cast<const float4*>(gbs[0] + f * 32 + 8 * g + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(gbs[1] + f * 32 + 8 * g), b1 = *reinterpret_cast<const float4*>(gbs[1] + f * 32 + 8 * g + 4);
            uint4 o;
            o.x = pack_bf2((v[i][0].x - mean) * rstd * g0.x + b0.x, (v[i][0].y - mean) * rstd * g0.y + b0.y);
            o.y = pack_bf2((v[i][0].z - mean) * rstd * g0.z + b0.z, (v[i][0].w - mean) * rstd * g0.w + b0.w);
            o.z = pack_bf2((v[i][1].x - mean) * rstd * g1.x + b1.x, (v[i][1].y - mean) * rstd * g1.y + b1.y);
            o.w = pack_bf2((v[i][1].z - mean) * rstd * g1.z + b1.z, (v[i][1].w - mean) * rstd * g1.w + b1.w);
            *reinterpret_cast<uint4*>(op + (size_t)f * 512) = o;
        }
    }
}

// One wave per row (the form of rounds 1-2): with one or two clips the 16-row-group kernel is only 96-192 blocks of serial phases
// (10.3 us per launch against 6.8 for this one, profiles/r03_kernel_trace_bench_b1.md); its scattered 16-byte stores do not matter at
// that size.  Same two-pass statistics, another summation order (big-batch and few-clip encoder outputs are not bit-identical anyway).
__global__ void __launch_bounds__(256)
k_enc_ln_rows(const float* __restrict__ src, const float* __restrict__ gamma, const float* __restrict__ beta,
              bf16_t* __restrict__ out_p, int K32, int d, int M)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nv = d >> 2;
    const float4* sp = reinterpret_cast<const float4*>(src + (size_t)m * d);
    float4 v[8], gv[8], bv[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        v[i] = (j < nv) ? sp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // gamma / beta requested with the row (one batch): loaded inside the output loop, every iteration's parameter loads put an
    // s_waitcnt vmcnt(0) in front of its store, which also waits for the previous iteration's store — 5 dependent round trips per row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        gv[i] = (j < nv) ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[i] = (j < nv) ? reinterpret_cast<const float4*>(beta)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
            q += a * a + b * b + c * c + e * e;
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) {
            const float4 g = gv[i], b = bv[i];
            uint2 o;
            o.x = pack_bf2((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y);
            o.y = pack_bf2((v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
            *reinterpret_cast<uint2*>(out_p + packed_index(m, j * 4, K32)) = o;
        }
    }
}

static inline void launch_enc_ln(hipStream_t st, const float* src, const float* gamma, const float* beta, bf16_t* out_p, int K32, int d, int M)
{
    static const int rows_below = [] { const char* v = std::getenv("WM_ENC_LN_ROWS_BELOW"); return v ? std::atoi(v) : 256; }();   // 16-row groups
    if (M / 16 < rows_below) { hipLaunchKernelGGL(k_enc_ln_rows, dim3((M + 3) / 4), dim3(256), 0, st, src, gamma, beta, out_p, K32, d, M); return; }
    if (K32 <= 40) hipLaunchKernelGGL(k_enc_ln<10>, dim3(M / 16), dim3(256), 0, st, src, gamma, beta, out_p, K32, d, M);
    else hipLaunchKernelGGL(k_enc_ln<16>, dim3(M / 16), dim3(256), 0, st, src, gamma, beta, out_p, K32, d, M);     // d <= 2048 (wm_create)
}

// LayerNorm -> fp8 e4m3 operand + one fp32 scale per row (fp8 MFMA GEMMs above).  RB: the LayerNorm output is first rounded to
// bf16 and also stored packed (the encoder output: stored bf16 by contract, and the cross-K/V projection quantises exactly
// those stored values, so that an oracle given the stored encoder output reproduces the fp8 codes bit for bit).
// scale = max|row| / 448 (1 for an all-zero row); q = y / scale, round-to-nearest-even e4m3 (v_cvt_pk_fp8_f32; |q| <= 448).
template <bool RB>
__global__ void __launch_bounds__(256)
k_enc_ln_f8(const float* __restrict__ src, const float* __restrict__ gamma, const float* __restrict__ beta,
            unsigned char* __restrict__ out8, float* __restrict__ xs, bf16_t* __restrict__ out_p, int K32, int d, int M)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nv = d >> 2;
    const float4* sp = reinterpret_cast<const float4*>(src + (size_t)m * d);
    float4 v[8], gv[8], bv[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        v[i] = (j < nv) ? sp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // gamma / beta requested with the row (one batch): loaded inside the output loop, every iteration's parameter loads put an
    // s_waitcnt vmcnt(0) in front of its store, which also waits for the previous iteration's store — 5 dependent round trips per row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        gv[i] = (j < nv) ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[i] = (j < nv) ? reinterpret_cast<const float4*>(beta)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
            q += a * a + b * b + c * c + e * e;
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[j];
            const float4 b = reinterpret_cast<const float4*>(beta)[j];
            float4 y = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                   (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
            if (RB) {
                uint2 o; o.x = pack_bf2(y.x, y.y); o.y = pack_bf2(y.z, y.w);
                *reinterpret_cast<uint2*>(out_p + packed_index(m, j * 4, K32)) = o;
                y = make_float4(__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u));
            }
            v[i] = y;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(y.x), fabsf(y.y))), fmaxf(fabsf(y.z), fabsf(y.w)));
        }
    }
    amax = wave_max(amax);
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    if (lane == 0) xs[m] = scale;
    const int K128 = (K32 + 3) >> 2;            // 128-k units per row; the bytes k >= d of the last unit stay zero (allocation-time memset)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) {
            const float q0 = fminf(fmaxf(v[i].x / scale, -448.f), 448.f), q1 = fminf(fmaxf(v[i].y / scale, -448.f), 448.f);
            const float q2 = fminf(fmaxf(v[i].z / scale, -448.f), 448.f), q3 = fminf(fmaxf(v[i].w / scale, -448.f), 448.f);
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(q2, q3, pk, true);
            *reinterpret_cast<int*>(out8 + f8k_index(m, j * 4, K128)) = pk;
        }
    }
}

// =============================================================================================
// conv front end as implicit GEMM: im2col gathers written directly in packed-operand order
// (HF:modeling_whisper.py:626-627: conv1 k3 p1, conv2 k3 s2 p1).  One thread = one 16-B chunk.
// =============================================================================================
__global__ void k_im2col1(const float* __restrict__ feats, bf16_t* __restrict__ A1, int n_mels, int Tm, int Tmpad, int K32, long nchunks)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nchunks) return;
    const long tile = q >> 6; const int ln = (int)(q & 63);
    const int row = (int)(tile / K32) * 16 + (ln & 15), k0 = (int)(tile % K32) * 32 + (ln >> 4) * 8;
    const int b = row / Tmpad, t = row - b * Tmpad;
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + 2 * e + h;
            const int kw = k / n_mels, c = k - kw * n_mels, tt = t - 1 + kw;
            v2[h] = (kw < 3 && t < Tm && tt >= 0 && tt < Tm) ? feats[((size_t)b * n_mels + c) * Tm + tt] : 0.f;
        }
        o[e] = pack_bf2(v2[0], v2[1]);
    }
    *reinterpret_cast<uint4*>(A1 + q * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

__global__ void k_im2col2(const bf16_t* __restrict__ a1, bf16_t* __restrict__ A2, int d, int Tm, int S, int Spad, int K32, long nchunks)
{
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nchunks) return;
    const long tile = q >> 6; const int ln = (int)(q & 63);
    const int row = (int)(tile / K32) * 16 + (ln & 15), k0 = (int)(tile % K32) * 32 + (ln >> 4) * 8;
    const int b = row / Spad, s = row - b * Spad;
    const int kw = k0 / d, c = k0 - kw * d, tt = 2 * s - 1 + kw;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (s < S && tt >= 0 && tt < Tm) o = *reinterpret_cast<const uint4*>(a1 + ((size_t)b * Tm + tt) * d + c);
    *reinterpret_cast<uint4*>(A2 + q * 8) = o;
}

// =============================================================================================
// F0 log-mel: reflect-padded 400-point Hann STFT (hop 160) as a direct DFT in fp32, power, Slaney mel,
// log10, then per-clip max clamp and (x+4)/4   (HF:feature_extraction_whisper.py:105-133)
// =============================================================================================
__device__ __forceinline__ int f2ord(float x) { const int i = __float_as_int(x); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void __launch_bounds__(256)
k_logmel_stft(const float* __restrict__ wav, const float* __restrict__ win, const float* __restrict__ tw,
              const float* __restrict__ melfb, float* __restrict__ feats, int* __restrict__ clipmax,
              int n_samples, int Tm, int n_mels)
{
    __shared__ float xw[4][400];
    __shared__ float tws[800];
    __shared__ float pw[4][208];
    const int b = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + w;
    for (int i = threadIdx.x; i < 800; i += 256) tws[i] = tw[i];
    const float* x = wav + (size_t)b * n_samples;
    if (f < Tm) {
        for (int i = lane; i < 400; i += 64) {
            int n = f * 160 + i - 200;
            if (n < 0) n = -n;
            if (n >= n_samples) n = 2 * (n_samples - 1) - n;
            xw[w][i] = x[n] * win[i];
        }
    }
    __syncthreads();
    if (f < Tm) {
        for (int k = lane; k < 201; k += 64) {
            float re = 0.f, im = 0.f;
            int idx = 0;
            for (int i = 0; i < 400; ++i) {
                const float v = xw[w][i];
                re += v * tws[2 * idx]; im -= v * tws[2 * idx + 1];
                idx += k; if (idx >= 400) idx -= 400;
            }
            pw[w][k] = re * re + im * im;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    if (f < Tm) {
        for (int m = lane; m < n_mels; m += 64) {
            float acc = 0.f;
            for (int k = 0; k < 201; ++k) acc += melfb[k * n_mels + m] * pw[w][k];
            const float lg = log10f(fmaxf(acc, 1e-10f));
            feats[((size_t)b * n_mels + m) * Tm + f] = lg;
            mx = fmaxf(mx, lg);
        }
    }
    mx = wave_max(mx);
    if (lane == 0 && f < Tm) atomicMax(clipmax + b, f2ord(mx));
}

__global__ void k_logmel_norm(float* __restrict__ feats, const int* __restrict__ clipmax, long per_clip, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float mx = ord2f(clipmax[i / per_clip]);
    feats[i] = (fmaxf(feats[i], mx - 8.0f) + 4.0f) * 0.25f;
}

// =============================================================================================
// Audio front door: channel-mean downmix + windowed-sinc polyphase resampling to the model
// rate — what the reference's callers do with torchaudio before the feature extractor (README.md:120-125,
// eval_whisper_medusa.py:41-45; algorithm: torchaudio==2.2.2 functional.py _get_sinc_resample_kernel /
// _apply_sinc_resample_kernel, sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99).
//   out[f*nw + j] = sum_k table[k][j] * xpad[f*og + k],   xpad[p] = mean_c in[c][p - width]  (0 outside the clip)
// One block = one output frame f (nw samples): the og + 2*width inputs it needs are downmixed into LDS once,
// threads walk j (table is stored [k][j]: coalesced; the LDS read is a broadcast).
// =============================================================================================
__global__ void __launch_bounds__(256)
k_resample(const float* __restrict__ in, const float* __restrict__ table, float* __restrict__ out, int channels, int n_in,
           int og, int nw, int width, int K, int n_out)
{
    extern __shared__ float s_x[];
    const int f = blockIdx.x, b = blockIdx.y;
    const float* src = in + (size_t)b * channels * n_in;
    const float inv = 1.0f / (float)channels;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const long p = (long)f * og + k - width;
        float v = 0.f;
        if (p >= 0 && p < n_in) {
            for (int c = 0; c < channels; ++c) v += src[(size_t)c * n_in + p];
            if (channels > 1) v *= inv;
        }
        s_x[k] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nw; j += blockDim.x) {
        const long o = (long)f * nw + j;
        if (o >= n_out) break;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(table[(size_t)k * nw + j], s_x[k], acc);
        out[(size_t)b * n_out + o] = acc;
    }
}

__global__ void k_downmix(const float* __restrict__ in, float* __restrict__ out, int channels, int n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    float v = 0.f;
    for (int c = 0; c < channels; ++c) v += in[((size_t)b * channels + c) * n + i];
    out[(size_t)b * n + i] = channels > 1 ? v * (1.0f / (float)channels) : v;
}

// =============================================================================================
// host side
// =============================================================================================
static int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

long wm_enc_resample_len(long n_in, int sr_in, int sr_out)
{
    if (n_in < 0 || sr_in < 1 || sr_out < 1) return -1;
    const int g = gcd_int(sr_in, sr_out);
    const long og = sr_in / g, nw = sr_out / g;
    return (nw * n_in + og - 1) / og;                 // ceil(new_freq * length / orig_freq)
}

// filter bank of torchaudio's Resample(sr_in, sr_out) defaults, computed like its float64 path (the phase offset
// j / new_freq is a float32 quotient there: `torch.arange(0, -new, -1) / new` is float32 before it meets the float64
// index grid), rounded to float32 at the end; stored [k][j].
static void build_resample_table(int og, int nw, int& width, std::vector<float>& tab)
{
    const double lowpass_filter_width = 6.0, rolloff = 0.99, pi = 3.14159265358979323846;
    const double base = (double)std::min(og, nw) * rolloff;
    width = (int)std::ceil(lowpass_filter_width * og / base);
    const int K = 2 * width + og;
    tab.assign((size_t)K * nw, 0.f);
    for (int j = 0; j < nw; ++j) {
        const double off = (double)((float)(-j) / (float)nw);
        for (int k = 0; k < K; ++k) {
            double t = (off + (double)(k - width) / (double)og) * base;
            t = std::min(std::max(t, -lowpass_filter_width), lowpass_filter_width);
            const double c = std::cos(t * pi / lowpass_filter_width / 2.0);
            const double window = c * c;
            t *= pi;
            const double sinc = (t == 0.0) ? 1.0 : std::sin(t) / t;
            tab[(size_t)k * nw + j] = (float)(sinc * (window * (base / og)));
        }
    }
}

int wm_enc_resample(wm_ctx* ctx, const float* in, int B, int channels, int n_in, int sr_in, int sr_out, float* out)
{
    hipStream_t st = ctx->stream;
    if (B < 1 || channels < 1 || n_in < 1 || sr_in < 1 || sr_out < 1) { ctx->err = "wm_resample: bad arguments"; return WM_ERR_ARG; }
    const int g = gcd_int(sr_in, sr_out), og = sr_in / g, nw = sr_out / g;
    if (og == nw) {                                   // same rate: torchaudio returns the input; only the downmix is left
        hipLaunchKernelGGL(k_downmix, dim3((unsigned)((n_in + 255) / 256), B), dim3(256), 0, st, in, out, channels, n_in);
        WM_HIP(hipGetLastError());
        WM_HIP(hipStreamSynchronize(st));             // like wm_logmel: the result is complete when the call returns
        return WM_OK;
    }
    if (ctx->rs_og != og || ctx->rs_nw != nw) {
        std::vector<float> tab;
        int width = 0;
        build_resample_table(og, nw, width, tab);
        if ((size_t)(2 * width + og) * sizeof(float) > 60 * 1024) { ctx->err = "wm_resample: rate ratio needs a filter longer than 15360 taps"; return WM_ERR_ARG; }
        WM_HIP(hipStreamSynchronize(st));
        if (ctx->rs_table) { WM_HIP(hipFree(ctx->rs_table)); ctx->rs_table = nullptr; }
        WM_HIP(hipMalloc(&ctx->rs_table, tab.size() * sizeof(float)));
        WM_HIP(hipMemcpy(ctx->rs_table, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
        ctx->rs_og = og; ctx->rs_nw = nw; ctx->rs_width = width;
    }
    const int K = 2 * ctx->rs_width + og;
    const long n_out = wm_enc_resample_len(n_in, sr_in, sr_out);
    const long frames = (n_out + nw - 1) / nw;
    hipLaunchKernelGGL(k_resample, dim3((unsigned)frames, B), dim3(256), K * sizeof(float), st, in, ctx->rs_table, out, channels, n_in,
                       og, nw, ctx->rs_width, K, (int)n_out);
    WM_HIP(hipGetLastError());
    WM_HIP(hipStreamSynchronize(st));                 // like wm_logmel: the result is complete when the call returns
    return WM_OK;
}

int wm_enc_logmel(wm_ctx* ctx, const float* wav, int B, int n_samples, float* feats)
{
    hipStream_t st = ctx->stream;
    if (B < 1 || B > ctx->maxB || n_samples != 160 * ctx->Tm) { ctx->err = "wm_logmel: bad B or n_samples (must be 320*n_ctx)"; return WM_ERR_ARG; }
    WM_HIP(hipEventRecord(ctx->ev0, st));
    WM_HIP(hipMemsetAsync(ctx->clipmax, 0x80, sizeof(int) * B, st));
    hipLaunchKernelGGL(k_logmel_stft, dim3((ctx->Tm + 3) / 4, B), dim3(256), 0, st, wav, ctx->win, ctx->twiddle, ctx->melfb, feats,
                       reinterpret_cast<int*>(ctx->clipmax), n_samples, ctx->Tm, ctx->cfg.n_mels);
    WM_HIP(hipGetLastError());
    const long per = (long)ctx->cfg.n_mels * ctx->Tm, total = per * B;
    hipLaunchKernelGGL(k_logmel_norm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feats,
                       reinterpret_cast<const int*>(ctx->clipmax), per, total);
    WM_HIP(hipGetLastError());
    WM_HIP(hipEventRecord(ctx->ev1, st));
    WM_HIP(hipEventSynchronize(ctx->ev1));
    WM_HIP(hipEventElapsedTime(&ctx->ms_logmel, ctx->ev0, ctx->ev1));
    return WM_OK;
}

// forward(encoder_outputs=...) (reference model.py:1223-1243 -> HF WhisperModel: a caller-supplied encoder_outputs[0] replaces the
// encoder pass): fp32 [B][S][d] rows -> the packed bf16 operand the encoder's last LayerNorm would have written (rows S..Spad of a
// clip are padding: zero), 8 consecutive k of one row per thread.
__global__ void __launch_bounds__(256)
k_pack_hidden(const float* __restrict__ hid, bf16_t* __restrict__ out_p, int S, int Spad, int d, long n8)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int d8 = d >> 3;
    const long row = i / d8; const int k = (int)(i - row * d8) * 8;
    const int b = (int)(row / Spad), s = (int)(row - (long)b * Spad);
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (s < S) {
        const float4* p = reinterpret_cast<const float4*>(hid + ((size_t)b * S + s) * d + k);
        const float4 a = p[0], c = p[1];
        o.x = pack_bf2(a.x, a.y); o.y = pack_bf2(a.z, a.w); o.z = pack_bf2(c.x, c.y); o.w = pack_bf2(c.z, c.w);
    }
    *reinterpret_cast<uint4*>(out_p + packed_index((int)row, k, d / 32)) = o;
}

// ---------------------------------------------------------------------------------------------
// wm_config.cross_kv_fp8: e4m3 copy of the projected cross-K/V (the decode loop's dominant HBM stream at several streams).
// One block per (head, stream, kv layer): max|x| over the head's S x 64 keys (padding rows excluded) for K and for V -> scale = max / 448
// (1 for an all-zero slab), then q = rne_e4m3(x / scale).  K keeps its row-major [Spad][64] order (one byte per element: a decode lane
// (key c, g) then loads the 16 consecutive dims 16 g .. + 15 of its key with ONE 16-byte load = two MFMA fragments; the query is
// permuted to match, wm_decoder.hip k_attn_mfma).  V: the V^T fragment layout [Spad/32][4 dim tiles][64 lanes][8] becomes
// [Spad/32][2][64 lanes][16]: a lane's 8 keys of dim tiles 2 j and 2 j + 1 adjacent (one 16-byte load per pair).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned bf2_absmax(unsigned w, unsigned m) {      // max of the two bf16 magnitudes of a dword and m
    return max(m, max(w & 0x7fffu, (w >> 16) & 0x7fffu));
}
__device__ __forceinline__ uint2 bf8_to_fp8(uint4 v, float inv_unused, float scale) {
    const float x0 = __uint_as_float(v.x << 16), x1 = __uint_as_float(v.x & 0xffff0000u), x2 = __uint_as_float(v.y << 16), x3 = __uint_as_float(v.y & 0xffff0000u);
    const float x4 = __uint_as_float(v.z << 16), x5 = __uint_as_float(v.z & 0xffff0000u), x6 = __uint_as_float(v.w << 16), x7 = __uint_as_float(v.w & 0xffff0000u);
    auto q = [&](float x) { return fminf(fmaxf(x / scale, -448.f), 448.f); };
    int a = __builtin_amdgcn_cvt_pk_fp8_f32(q(x0), q(x1), 0, false); a = __builtin_amdgcn_cvt_pk_fp8_f32(q(x2), q(x3), a, true);
    int b = __builtin_amdgcn_cvt_pk_fp8_f32(q(x4), q(x5), 0, false); b = __builtin_amdgcn_cvt_pk_fp8_f32(q(x6), q(x7), b, true);
    return make_uint2((unsigned)a, (unsigned)b);
}
__global__ void __launch_bounds__(256)
k_xkv_quant(const bf16_t* __restrict__ kx, const bf16_t* __restrict__ vx, unsigned char* __restrict__ kx8, unsigned char* __restrict__ vx8,
            float* __restrict__ kxs, float* __restrict__ vxs, int S, int Spad)
{
    __shared__ unsigned s_k[4], s_v[4];
    const size_t head = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;          // [kv layer][stream][head]
    const size_t slab = (size_t)Spad * 64;
    const uint4* kp = reinterpret_cast<const uint4*>(kx + head * slab);
    const uint4* vp = reinterpret_cast<const uint4*>(vx + head * slab);
    const int n16 = Spad * 8;                   // 16-byte units (8 bf16) of a slab
    unsigned mk = 0u, mv = 0u;
    for (int i = threadIdx.x; i < n16; i += 256) {
        if ((i >> 3) < S) { const uint4 a = kp[i]; mk = bf2_absmax(a.x, bf2_absmax(a.y, bf2_absmax(a.z, bf2_absmax(a.w, mk)))); }      // K row = i / 8
        // V^T fragment unit i = (t * 4 + dt) * 64 + lane: element e of lane (c, g) is key 32 t + 16 (e >> 2) + 4 g + (e & 3)
        const int t = i >> 8, g = (i & 63) >> 4, k0 = 32 * t + 4 * g;
        const uint4 b = vp[i];
        const unsigned h0 = (b.x & 0x7fffu), h1 = (b.x >> 16) & 0x7fffu, h2 = (b.y & 0x7fffu), h3 = (b.y >> 16) & 0x7fffu;
        const unsigned h4 = (b.z & 0x7fffu), h5 = (b.z >> 16) & 0x7fffu, h6 = (b.w & 0x7fffu), h7 = (b.w >> 16) & 0x7fffu;
        if (k0 + 0 < S) mv = max(mv, h0); if (k0 + 1 < S) mv = max(mv, h1); if (k0 + 2 < S) mv = max(mv, h2); if (k0 + 3 < S) mv = max(mv, h3);
        if (k0 + 16 < S) mv = max(mv, h4); if (k0 + 17 < S) mv = max(mv, h5); if (k0 + 18 < S) mv = max(mv, h6); if (k0 + 19 < S) mv = max(mv, h7);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mk = max(mk, (unsigned)__shfl_xor((int)mk, o, 64)); mv = max(mv, (unsigned)__shfl_xor((int)mv, o, 64)); }
    if ((threadIdx.x & 63) == 0) { s_k[threadIdx.x >> 6] = mk; s_v[threadIdx.x >> 6] = mv; }
    __syncthreads();
    mk = max(max(s_k[0], s_k[1]), max(s_k[2], s_k[3])); mv = max(max(s_v[0], s_v[1]), max(s_v[2], s_v[3]));
    const float ak = __uint_as_float(mk << 16), av = __uint_as_float(mv << 16);
    const float sk = ak > 0.f ? ak / 448.0f : 1.0f, sv = av > 0.f ? av / 448.0f : 1.0f;
    if (threadIdx.x == 0) { kxs[head] = sk; vxs[head] = sv; }
    unsigned char* k8 = kx8 + head * slab;
    unsigned char* v8 = vx8 + head * slab;
    for (int i = threadIdx.x; i < n16; i += 256) {
        *reinterpret_cast<uint2*>(k8 + (size_t)i * 8) = bf8_to_fp8(kp[i], 0.f, sk);
        const int t = i >> 8, dt = (i >> 6) & 3, lane = i & 63;
        *reinterpret_cast<uint2*>(v8 + (((size_t)(t * 2 + (dt >> 1)) * 64 + lane) * 16 + (dt & 1) * 8)) = bf8_to_fp8(vp[i], 0.f, sv);
    }
}

int wm_enc_quant_cross_kv(wm_ctx* ctx, int B)
{
    if (!ctx->xkv8) return WM_OK;
    // (slabs are indexed [kv layer][B][H]: the B of THIS encode, like kx / vx)
    hipLaunchKernelGGL(k_xkv_quant, dim3(ctx->H, B, ctx->nkv), dim3(256), 0, ctx->stream, ctx->kx, ctx->vx, ctx->kx8, ctx->vx8, ctx->kxs, ctx->vxs,
                       ctx->S, ctx->Spad);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

int wm_enc_set_output(wm_ctx* ctx, const float* hidden, int B)
{
    hipStream_t st = ctx->stream;
    if (B < 1 || B > ctx->maxB) { ctx->err = "wm_set_encoder_output: B out of range"; return WM_ERR_ARG; }
    if (ctx->enc_f8) { ctx->err = "wm_set_encoder_output: not available on an enc_fp8 context (its cross-K/V projection reads the fp8 LayerNorm output)"; return WM_ERR_ARG; }
    const int d = ctx->d, H = ctx->H, S = ctx->S, Spad = ctx->Spad, K32 = d / 32, M = B * Spad;
    WM_HIP(hipEventRecord(ctx->ev0, st));
    const long n8 = (long)M * d / 8;
    hipLaunchKernelGGL(k_pack_hidden, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, hidden, ctx->enc_out, S, Spad, d, n8);
    WM_HIP(hipGetLastError());
    WM_HIP(launch_gemm_tiled(st, ctx->enc_out, ctx->ckv_w, M, ctx->nkv * 2 * d, K32, EpCrossKV{ctx->kx, ctx->vx, ctx->ckv_b, Spad, H, d, B}));
    { const int rq = wm_enc_quant_cross_kv(ctx, B); if (rq) return rq; }
    ctx->Benc = B;
    WM_HIP(hipEventRecord(ctx->ev1, st));
    WM_HIP(hipEventSynchronize(ctx->ev1));
    WM_HIP(hipEventElapsedTime(&ctx->ms_encode, ctx->ev0, ctx->ev1));
    return WM_OK;
}

int wm_enc_encode(wm_ctx* ctx, const float* feats, int B)
{
    hipStream_t st = ctx->stream;
    if (B < 1 || B > ctx->maxB) { ctx->err = "wm_encode: B out of range"; return WM_ERR_ARG; }
    const int d = ctx->d, H = ctx->H, ffn = ctx->ffn, S = ctx->S, Spad = ctx->Spad, Tm = ctx->Tm, Tmpad = ctx->Tmpad;
    const int K32 = d / 32, M = B * Spad;
    const bool f8 = ctx->enc_f8;               // fp8 MFMA for the LayerNorm-fed GEMMs (QKV, FC1, cross-K/V projection)
    WM_HIP(hipEventRecord(ctx->ev0, st));
    // conv1 + GELU
    {
        const int K32c = ctx->K1pad / 32;
        const long nch = (long)B * Tmpad * ctx->K1pad / 8;
        hipLaunchKernelGGL(k_im2col1, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, st, feats, ctx->A1, ctx->cfg.n_mels, Tm, Tmpad, K32c, nch);
        WM_HIP(hipGetLastError());
        WM_HIP(launch_gemm_tiled(st, ctx->A1, ctx->conv1_w, B * Tmpad, d, K32c, EpConv1{ctx->a1, ctx->conv1_b, Tm, Tmpad, d}));
    }
    // conv2 (stride 2) + GELU + positions
    {
        const int K32c = 3 * d / 32;
        const long nch = (long)M * 3 * d / 8;
        hipLaunchKernelGGL(k_im2col2, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, st, ctx->a1, ctx->A2, d, Tm, S, Spad, K32c, nch);
        WM_HIP(hipGetLastError());
        WM_HIP(launch_gemm_tiled(st, ctx->A2, ctx->conv2_w, M, d, K32c, EpConv2{ctx->eh, ctx->conv2_b, ctx->enc_pos, S, Spad, d}));
    }
    for (int l = 0; l < ctx->cfg.enc_layers; ++l) {
        const EncLayerW& w = ctx->enc[l];
        if (f8) {
            hipLaunchKernelGGL(k_enc_ln_f8<false>, dim3((M + 3) / 4), dim3(256), 0, st, ctx->eh, w.ln1_w, w.ln1_b, ctx->exn8, ctx->exs, nullptr, K32, d, M);
            WM_HIP(hipGetLastError());
            WM_HIP(launch_gemm_f8(st, ctx->exn8, w.qkv_w8, M, 3 * d, (K32 + 3) / 4,
                                  EpScaled<EpQKVEnc>{EpQKVEnc{ctx->eq, ctx->ek, ctx->evt, w.qkv_b, Spad, H, d}, ctx->exs, w.qkv_ws}));
        } else {
            launch_enc_ln(st, ctx->eh, w.ln1_w, w.ln1_b, ctx->exn, K32, d, M);
            WM_HIP(hipGetLastError());
            WM_HIP(launch_gemm_tiled(st, ctx->exn, w.qkv_w, M, 3 * d, K32, EpQKVEnc{ctx->eq, ctx->ek, ctx->evt, w.qkv_b, Spad, H, d}));
        }
        // queries per block: 256 (QT = 4: every K / V fragment read from LDS serves 4 query tiles) once that still gives >= 256
        // blocks; with one or two clips 64 (QT = 1): twice the blocks of the 128-query form, two or three resident per CU, so
        // one block's softmax (VALU) overlaps another's MFMAs — at one wave per SIMD nothing did
        static const int qt1_below = [] { const char* v = std::getenv("WM_FLASH_QT1_BELOW"); return v ? std::atoi(v) : 256; }();     // one clip: 6.63 -> 6.42 ms per encoder pass
        const int flash_var = [] { const char* v = std::getenv("WM_FLASH_VARIANT"); return v ? std::atoi(v) : 0; }();   // measurement: 1 = groups of 4, 2 = three blocks per CU, 3 = both
        if (B * (Spad / 128) * H >= 256 && Spad % 256 == 0) {
            const dim3 grid(Spad / 256, H, B);
            if (flash_var == 1) hipLaunchKernelGGL((k_flash_enc<4, 4, 2>), grid, dim3(256), 3 * 16 * 1024, st, ctx->eq, ctx->ek, ctx->evt, ctx->exn, S, Spad, H, K32);
            else if (flash_var == 2) hipLaunchKernelGGL((k_flash_enc<4, 2, 3>), grid, dim3(256), 3 * 16 * 1024, st, ctx->eq, ctx->ek, ctx->evt, ctx->exn, S, Spad, H, K32);
            else if (flash_var == 3) hipLaunchKernelGGL((k_flash_enc<4, 1, 3>), grid, dim3(256), 3 * 16 * 1024, st, ctx->eq, ctx->ek, ctx->evt, ctx->exn, S, Spad, H, K32);
            else hipLaunchKernelGGL(k_flash_enc<4>, grid, dim3(256), 3 * 16 * 1024, st, ctx->eq, ctx->ek, ctx->evt, ctx->exn, S, Spad, H, K32);
        }
        else if (B * (Spad / 128) * H < qt1_below)
            hipLaunchKernelGGL(k_flash_enc<1>, dim3(Spad / 64, H, B), dim3(256), 3 * 16 * 1024, st, ctx->eq, ctx->ek, ctx->evt, ctx->exn, S, Spad, H, K32);
        else
            hipLaunchKernelGGL(k_flash_enc<2>, dim3(Spad / 128, H, B), dim3(256), 3 * 16 * 1024, st, ctx->eq, ctx->ek, ctx->evt, ctx->exn, S, Spad, H, K32);
        WM_HIP(hipGetLastError());
        WM_HIP(launch_gemm_tiled(st, ctx->exn, w.out_w, M, d, K32, EpResidual{ctx->eh, w.out_b, d, M}));
        if (f8) {
            hipLaunchKernelGGL(k_enc_ln_f8<false>, dim3((M + 3) / 4), dim3(256), 0, st, ctx->eh, w.ln2_w, w.ln2_b, ctx->exn8, ctx->exs, nullptr, K32, d, M);
            WM_HIP(hipGetLastError());
            WM_HIP(launch_gemm_f8(st, ctx->exn8, w.fc1_w8, M, ffn, (K32 + 3) / 4,
                                  EpScaled<EpPackedAct<2>>{EpPackedAct<2>{ctx->eff, nullptr, w.fc1_b, ffn / 32, M}, ctx->exs, w.fc1_ws}));
        } else {
            launch_enc_ln(st, ctx->eh, w.ln2_w, w.ln2_b, ctx->exn, K32, d, M);
            WM_HIP(hipGetLastError());
            WM_HIP(launch_gemm_tiled(st, ctx->exn, w.fc1_w, M, ffn, K32, EpPackedAct<2>{ctx->eff, nullptr, w.fc1_b, ffn / 32, M}));
        }
        WM_HIP(launch_gemm_tiled(st, ctx->eff, w.fc2_w, M, d, ffn / 32, EpResidual{ctx->eh, w.fc2_b, d, M}));
    }
    // cross K/V of every decoder layer (+ the Medusa block) in one GEMM
    if (f8) {
        hipLaunchKernelGGL(k_enc_ln_f8<true>, dim3((M + 3) / 4), dim3(256), 0, st, ctx->eh, ctx->enc_lnf_w, ctx->enc_lnf_b, ctx->exn8, ctx->exs,
                           ctx->enc_out, K32, d, M);
        WM_HIP(hipGetLastError());
        WM_HIP(launch_gemm_f8(st, ctx->exn8, ctx->ckv_w8, M, ctx->nkv * 2 * d, (K32 + 3) / 4,
                              EpScaled<EpCrossKV>{EpCrossKV{ctx->kx, ctx->vx, ctx->ckv_b, Spad, H, d, B}, ctx->exs, ctx->ckv_ws}));
    } else {
        launch_enc_ln(st, ctx->eh, ctx->enc_lnf_w, ctx->enc_lnf_b, ctx->enc_out, K32, d, M);
        WM_HIP(hipGetLastError());
        WM_HIP(launch_gemm_tiled(st, ctx->enc_out, ctx->ckv_w, M, ctx->nkv * 2 * d, K32,
                                 EpCrossKV{ctx->kx, ctx->vx, ctx->ckv_b, Spad, H, d, B}));
    }
    { const int rq = wm_enc_quant_cross_kv(ctx, B); if (rq) return rq; }
    ctx->Benc = B;
    WM_HIP(hipEventRecord(ctx->ev1, st));
    WM_HIP(hipEventSynchronize(ctx->ev1));
    WM_HIP(hipEventElapsedTime(&ctx->ms_encode, ctx->ev0, ctx->ev1));
    return WM_OK;
}
