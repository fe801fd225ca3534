// wm_decoder.hip — decode-step kernels and the per-iteration schedule (F3..F14 of SURVEY.md §8a).
//
// One Medusa iteration = base pass (1 token per stream; P on the first) + verify pass (K+1 candidate
// tokens per stream) + accept.  All lengths live on the device (L, kvlen, finished), every kernel has a
// static shape, so the whole iteration is captured once as a hipGraph and replayed (wm_engine.hip).
#include "wm_internal.h"
#include "wm_skinny_gemm.h"

// ---------------------------------------------------------------------------------------------
// embed: h[row] = embed_tokens[tok] + embed_positions[pos]        (HF:modeling_whisper.py:745-765)
// ---------------------------------------------------------------------------------------------
__global__ void k_embed(float* __restrict__ h, const bf16_t* __restrict__ tok_emb, const float* __restrict__ pos_emb,
                        const int* __restrict__ base, const int* __restrict__ tok_src, int tok_stride, int use_base_off,
                        int Mper, int d, int V, int Tmax)
{
    const int row = blockIdx.x, s = row / Mper, r = row - s * Mper;
    const int b0 = base[s];
    int tok = tok_src[(size_t)s * tok_stride + (use_base_off ? b0 : 0) + r];
    tok = min(max(tok, 0), V - 1);
    const int pos = min(b0 + r, Tmax - 1);
    const bf16_t* te = tok_emb + (size_t)tok * d;
    const float* pe = pos_emb + (size_t)pos * d;
    for (int j = threadIdx.x * 4; j < d; j += blockDim.x * 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(te + j);
        const float4 p = *reinterpret_cast<const float4*>(pe + j);
        *reinterpret_cast<float4*>(h + (size_t)row * d + j) =
            make_float4(bf2f((bf16_t)(t.x & 0xffff)) + p.x, bf2f((bf16_t)(t.x >> 16)) + p.y,
                        bf2f((bf16_t)(t.y & 0xffff)) + p.z, bf2f((bf16_t)(t.y >> 16)) + p.w);
    }
}

// ---------------------------------------------------------------------------------------------
// rows_norm: one wave per row. out row m <- LayerNorm (or identity) of src row m*src_mul+src_off.
// Optional outputs: fp32 rows (two copies) and packed-bf16 rows at m*p_mul + p_off.
// ---------------------------------------------------------------------------------------------
__global__ void k_rows_norm(const float* __restrict__ src, int src_mul, int src_off, const float* __restrict__ gamma,
                            const float* __restrict__ beta, int do_norm, float* __restrict__ out_a, float* __restrict__ out_b,
                            bf16_t* __restrict__ out_p, int K32, int p_mul, int p_off, int d, int M)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nv = d >> 2;
    const float4* sp = reinterpret_cast<const float4*>(src + (size_t)(m * src_mul + src_off) * d);
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
            if (out_p) {
                uint2 o; o.x = pack_bf2(y.x, y.y); o.y = pack_bf2(y.z, y.w);
                *reinterpret_cast<uint2*>(out_p + packed_index(m * p_mul + p_off, j * 4, K32)) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Decode attention for M (<=16) query rows of one (stream, head).
//  CROSS = false: causal self-attention over the contiguous KV cache (keys 0 .. base+r), output
//                 written as packed bf16 rows (operand of out_proj).
//  CROSS = true : keys [split*Ck, ...) of the encoder cross K/V; writes un-normalised partials
//                 (max, sum, o[64]) that the out_proj loader combines (LdCombine).
// fp32 scores / softmax, bf16 K/V (HF:modeling_whisper.py:214-238, 288-340).
// ---------------------------------------------------------------------------------------------
template <bool CROSS>
__global__ void __launch_bounds__(256)
k_attn_decode(const float* __restrict__ q, const bf16_t* __restrict__ kmat, const bf16_t* __restrict__ vmat,
              const int* __restrict__ base, bf16_t* __restrict__ xout, float* __restrict__ ml, float* __restrict__ po,
              int Mper, int H, int rows_alloc /* Tal or Spad */, int S, int Ck, int NS, int K32)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int hd = blockIdx.y, s = blockIdx.z, split = blockIdx.x;
    const int tid = threadIdx.x, d = H * 64;
    const int ldp = CROSS ? Ck : rows_alloc;                // ps row stride
    float* qs = reinterpret_cast<float*>(smem);             // [16][64]
    float* ps = qs + 16 * 64;                               // [16][ldp]
    float* red = ps + 16 * ldp;                             // [4][16][64]

    int k0, k1;
    if (CROSS) { k0 = split * Ck; k1 = min(S, k0 + Ck); }
    else { k0 = 0; k1 = min(base[s] + Mper, rows_alloc); }
    const int nk = max(k1 - k0, 0);
    const int b0 = CROSS ? 0 : base[s];
    const bf16_t* kp = kmat + ((size_t)s * H + hd) * rows_alloc * 64;
    const bf16_t* vp = vmat + ((size_t)s * H + hd) * rows_alloc * 64;

    for (int e = tid; e < Mper * 64; e += 256) qs[e] = q[(size_t)(s * Mper + (e >> 6)) * d + hd * 64 + (e & 63)];
    __syncthreads();

    // ---- scores ----
    for (int j = tid; j < nk; j += 256) {
        const uint4* kr = reinterpret_cast<const uint4*>(kp + (size_t)(k0 + j) * 64);
        float kf[64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 u = kr[i];
            kf[8 * i + 0] = bf2f((bf16_t)(u.x & 0xffff)); kf[8 * i + 1] = bf2f((bf16_t)(u.x >> 16));
            kf[8 * i + 2] = bf2f((bf16_t)(u.y & 0xffff)); kf[8 * i + 3] = bf2f((bf16_t)(u.y >> 16));
            kf[8 * i + 4] = bf2f((bf16_t)(u.z & 0xffff)); kf[8 * i + 5] = bf2f((bf16_t)(u.z >> 16));
            kf[8 * i + 6] = bf2f((bf16_t)(u.w & 0xffff)); kf[8 * i + 7] = bf2f((bf16_t)(u.w >> 16));
        }
        for (int r = 0; r < Mper; ++r) {
            const float4* qr = reinterpret_cast<const float4*>(qs + r * 64);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4 qq = qr[i];
                a0 += qq.x * kf[4 * i]; a1 += qq.y * kf[4 * i + 1]; a2 += qq.z * kf[4 * i + 2]; a3 += qq.w * kf[4 * i + 3];
            }
            float sc = (a0 + a1) + (a2 + a3);
            if (!CROSS && (k0 + j) > b0 + r) sc = -INFINITY;      // causal: row r sees keys <= base + r
            ps[r * ldp + j] = sc;
        }
    }
    __syncthreads();

    // ---- softmax statistics: wave w owns rows w, w+4, ... ----
    {
        const int lane = tid & 63, w = tid >> 6;
        for (int r = w; r < Mper; r += 4) {
            float mx = -INFINITY;
            for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, ps[r * ldp + j]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = lane; j < nk; j += 64) {
                const float e = (mx == -INFINITY) ? 0.f : __expf(ps[r * ldp + j] - mx);
                ps[r * ldp + j] = e; sum += e;
            }
            sum = wave_sum(sum);
            if (CROSS) {
                if (lane == 0) {
                    float* o = ml + (((size_t)(s * Mper + r) * H + hd) * NS + split) * 2;
                    o[0] = mx; o[1] = sum;
                }
            } else {
                const float inv = 1.0f / sum;
                for (int j = lane; j < nk; j += 64) ps[r * ldp + j] *= inv;
            }
        }
    }
    __syncthreads();

    // ---- P.V: thread = (dd, key-group g); fixed-order reduction over the 4 groups ----
    {
        const int dd = tid & 63, g = tid >> 6;
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int j = g; j < nk; j += 4) {
            const float vv = bf2f(vp[(size_t)(k0 + j) * 64 + dd]);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r < Mper) acc[r] += ps[r * ldp + j] * vv;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r < Mper) red[(g * 16 + r) * 64 + dd] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < Mper * 16; e += 256) {
        const int r = e >> 4, c = (e & 15) * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 p = *reinterpret_cast<const float4*>(red + (g * 16 + r) * 64 + c);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        const int row = s * Mper + r;
        if (CROSS) {
            *reinterpret_cast<float4*>(po + (((size_t)row * H + hd) * NS + split) * 64 + c) = o;
        } else {
            uint2 u; u.x = pack_bf2(o.x, o.y); u.y = pack_bf2(o.z, o.w);
            *reinterpret_cast<uint2*>(xout + packed_index(row, hd * 64 + c, K32)) = u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// select: logits processors (F7) + argmax (F9) + typical-acceptance statistics (F11) for one row.
//   row -> stream s = row / rps, slot i = row % rps;  cur_len = L[s] (same for every row, model.py:689-694)
//   mode 0: argmax only.  mode 1: also p(candidate_{i+1}) and entropy of softmax(x / T).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float proc_logit(float x, int n, int cur_len, const GenDev& gp, const unsigned char* mask,
                                            const float* exppen)
{
    if (n == gp.eos && gp.exp_start >= 0 && cur_len > gp.exp_start) x += fabsf(x) * exppen[cur_len];
    const unsigned char mk = mask[n];
    if ((mk & 1) || ((mk & 2) && cur_len == gp.P)) x = -INFINITY;
    return x;
}

__global__ void __launch_bounds__(512)
k_select(const float* __restrict__ logits, GenDev gp, const unsigned char* __restrict__ mask, const float* __restrict__ exppen,
         const int* __restrict__ L, const int* __restrict__ cand, int rps, int mode, int out_row0,
         int* __restrict__ amax, float* __restrict__ pc, float* __restrict__ ent)
{
    __shared__ float sv[8]; __shared__ int si[8]; __shared__ float sb[2];
    const int row = blockIdx.x, s = row / rps, i = row - s * rps;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cur_len = L[s];
    const float* x = logits + (size_t)row * gp.Vpad;
    // pass 1: max / first argmax
    float mx = -INFINITY; int mi = 0x7fffffff;
    for (int n = tid; n < gp.V; n += 512) {
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
    for (int k = 1; k < 8; ++k) if (sv[k] > mx || (sv[k] == mx && si[k] < mi)) { mx = sv[k]; mi = si[k]; }
    const int orow = out_row0 + row;
    if (tid == 0) amax[orow] = mi;
    if (mode == 0 || i + 1 >= rps) return;
    // pass 2: Z = sum exp((x - max)/T)
    float z = 0.f;
    for (int n = tid; n < gp.V; n += 512) {
        const float v = proc_logit(x[n], n, cur_len, gp, mask, exppen);
        z += (v == -INFINITY) ? 0.f : expf((v - mx) * gp.inv_temp);
    }
    z = wave_sum(z);
    __syncthreads();
    if (lane == 0) sv[w] = z;
    __syncthreads();
    z = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) z += sv[k];
    // pass 3: H = -sum p log(p + 1e-5)      (medusa_utils.py:566-568)
    float hsum = 0.f;
    const float invz = 1.0f / z;
    for (int n = tid; n < gp.V; n += 512) {
        const float v = proc_logit(x[n], n, cur_len, gp, mask, exppen);
        const float p = (v == -INFINITY) ? 0.f : expf((v - mx) * gp.inv_temp) * invz;
        hsum += p * logf(p + 1e-5f);
    }
    hsum = wave_sum(hsum);
    __syncthreads();
    if (lane == 0) sv[w] = hsum;
    __syncthreads();
    if (tid == 0) {
        float hh = 0.f;
        for (int k = 0; k < 8; ++k) hh += sv[k];
        const int c = cand[s * 16 + i + 1];
        const float vc = proc_logit(x[c], c, cur_len, gp, mask, exppen);
        pc[orow] = (vc == -INFINITY) ? 0.f : expf((vc - mx) * gp.inv_temp) * invz;
        ent[orow] = -hh;
    }
    (void)sb;
}

// candidates of the base pass: cand[s][i] = argmax of head i   (medusa_utils.py:446-458, top-1 chain)
__global__ void k_set_cand(const int* __restrict__ amax, int* __restrict__ cand, int rps, int n)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) cand[(e / rps) * 16 + (e % rps)] = amax[e];
}

// ---------------------------------------------------------------------------------------------
// accept: one wavefront per stream.  Lane i (< K) evaluates candidate i+1; the leading-true count of
// the wave ballot is the accept length a (medusa_utils.py:573-577 cumprod-sum for one candidate path).
// Then emit tokens, keep the accepted provisional KV rows by advancing kvlen (model.py:378-402), apply
// the stop rules (model.py:774-793).
// ---------------------------------------------------------------------------------------------
__global__ void k_accept(GenDev gp, const int* __restrict__ cand, const int* __restrict__ amax, const float* __restrict__ pc,
                         const float* __restrict__ ent, int* __restrict__ ids, int* __restrict__ L, int* __restrict__ kvlen,
                         int* __restrict__ finished, int* __restrict__ niter, long long* __restrict__ hist)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (finished[s]) return;
    const int K = gp.K, rps = K + 1;
    bool ok = false;
    if (lane < K) {
        if (gp.accept_mode == WM_ACCEPT_GREEDY) ok = (cand[s * 16 + lane + 1] == amax[s * rps + lane]);
        else {
            const float thr = fminf(gp.thr, gp.alpha * expf(-ent[s * rps + lane]));
            ok = pc[s * rps + lane] > thr;
        }
    }
    const unsigned long long m = __ballot(ok);
    const unsigned long long inv = ~m;
    int a = (inv == 0ull) ? 64 : (__ffsll((long long)inv) - 1);
    if (a > K) a = K;
    const int Lcur = L[s];
    const int n_emit = (a == 0) ? 2 : a + 1;
    int tok = -1;
    if (lane < n_emit) {
        tok = (a == 0 && lane == 1) ? amax[s * rps + 0] : cand[s * 16 + lane];
        if (Lcur + lane < gp.Tids) ids[(size_t)s * gp.Tids + Lcur + lane] = tok;
    }
    const bool hit_eos = __ballot(lane < n_emit && tok == gp.eos) != 0ull;
    if (lane == 0) {
        const int Ln = Lcur + n_emit;
        L[s] = Ln;
        kvlen[s] = (a == 0) ? Lcur + 1 : Lcur + a;
        niter[s] += 1;
        atomicAdd(reinterpret_cast<unsigned long long*>(hist + a), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(hist + 16), (unsigned long long)n_emit);
        if (hit_eos || Ln >= gp.max_length || Ln + K >= gp.hard_max_length) finished[s] = 1;
    }
}

__global__ void k_accept_vanilla1(GenDev gp, int B, const int* __restrict__ amax, int* __restrict__ ids, int* __restrict__ L,
                                  int* __restrict__ kvlen, int* __restrict__ finished, int* __restrict__ niter,
                                  long long* __restrict__ hist)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B || finished[s]) return;
    const int Lcur = L[s], tok = amax[s];
    if (Lcur < gp.Tids) ids[(size_t)s * gp.Tids + Lcur] = tok;
    L[s] = Lcur + 1; kvlen[s] = Lcur; niter[s] += 1;
    atomicAdd(reinterpret_cast<unsigned long long*>(hist + 16), 1ull);
    if (tok == gp.eos || Lcur + 1 >= gp.max_length) finished[s] = 1;
}

// =============================================================================================
// host side
// =============================================================================================
static inline int r16(int x) { return (x + 15) & ~15; }

static int dec_layer(wm_ctx* ctx, const DecLayerW& w, int slot, float* h, int b0, int nb, int Mper, const int* base, bool kv_only)
{
    hipStream_t st = ctx->stream;
    const int d = ctx->d, H = ctx->H, K32 = d / 32, R = nb * Mper;
    bf16_t* kc = ctx->kc + ((size_t)slot * ctx->maxB + b0) * H * ctx->Tal * 64;
    bf16_t* vc = ctx->vc + ((size_t)slot * ctx->maxB + b0) * H * ctx->Tal * 64;
    const bf16_t* kx = ctx->kx + ((size_t)slot * ctx->Benc + b0) * H * ctx->Spad * 64;
    const bf16_t* vx = ctx->vx + ((size_t)slot * ctx->Benc + b0) * H * ctx->Spad * 64;
    // 1. LN1 + QKV, k/v rows straight into the cache
    WM_HIP(launch_skinny(st, w.qkv_w, 3 * d / 16, K32, R,
                         LdNorm{h, w.ln1_w, w.ln1_b, nullptr, d, K32, R, 1, 0, 1},
                         EpQKVDec{ctx->qbuf, kc, vc, w.qkv_b, base, Mper, d, H, ctx->Tal, R}));
    if (kv_only) return WM_OK;
    // 2. causal self-attention over the contiguous cache
    {
        const size_t lds = (size_t)(16 * 64 + 16 * ctx->Tal + 4 * 16 * 64) * sizeof(float);
        hipLaunchKernelGGL(k_attn_decode<false>, dim3(1, H, nb), dim3(256), lds, st, ctx->qbuf, kc, vc, base, ctx->xbuf,
                           nullptr, nullptr, Mper, H, ctx->Tal, 0, 0, 1, K32);
        WM_HIP(hipGetLastError());
    }
    // 3. out_proj + residual
    WM_HIP(launch_skinny(st, w.out_w, d / 16, K32, R, LdPacked{ctx->xbuf, K32}, EpResidual{h, w.out_b, d, R}));
    // 4. LN2 + cross-attention q
    WM_HIP(launch_skinny(st, w.cq_w, d / 16, K32, R, LdNorm{h, w.ln2_w, w.ln2_b, nullptr, d, K32, R, 1, 0, 1},
                         EpF32{ctx->qbuf, w.cq_b, d, R, 0.125f}));
    // 5. cross-attention over the encoder K/V, split over keys
    {
        const size_t lds = (size_t)(16 * 64 + 16 * ctx->Ck + 4 * 16 * 64) * sizeof(float);
        hipLaunchKernelGGL(k_attn_decode<true>, dim3(ctx->NS, H, nb), dim3(256), lds, st, ctx->qbuf, kx, vx, base, nullptr,
                           ctx->cml, ctx->co, Mper, H, ctx->Spad, ctx->S, ctx->Ck, ctx->NS, K32);
        WM_HIP(hipGetLastError());
    }
    // 6. combine splits + out_proj + residual
    WM_HIP(launch_skinny(st, w.cout_w, d / 16, K32, R, LdCombine{ctx->cml, ctx->co, H, ctx->NS, K32, R},
                         EpResidual{h, w.cout_b, d, R}));
    // 7. LN3 + fc1 + GELU
    WM_HIP(launch_skinny(st, w.fc1_w, ctx->ffn / 16, K32, R, LdNorm{h, w.ln3_w, w.ln3_b, nullptr, d, K32, R, 1, 0, 1},
                         EpPackedAct<1>{ctx->fbuf, w.fc1_b, ctx->ffn / 32, R}));
    // 8. fc2 + residual
    WM_HIP(launch_skinny(st, w.fc2_w, d / 16, ctx->ffn / 32, R, LdPacked{ctx->fbuf, ctx->ffn / 32},
                         EpResidual{h, w.fc2_b, d, R}));
    return WM_OK;
}

// ---- stage 1: embed + the L decoder layers for streams [b0, b0+nb), Mper tokens each -> ctx->h --------
// mode 0 = base pass (tokens ids[kvlen..], positions kvlen..), mode 1 = verify pass (tokens cand[0..Mper),
// positions L..).
int wm_dec_stage_layers(wm_ctx* ctx, int b0, int nb, int Mper, int mode)
{
    hipStream_t st = ctx->stream;
    const int d = ctx->d, R = nb * Mper;
    const int* base = (mode == 0 ? ctx->kvlen : ctx->L) + b0;
    if (R > WM_MAX_ROWS_SKINNY || Mper > 16) { ctx->err = "decode chunk exceeds 32 rows / 16 tokens per stream"; return WM_ERR_ARG; }
    if (mode == 0)
        hipLaunchKernelGGL(k_embed, dim3(R), dim3(256), 0, st, ctx->h, ctx->tok_emb, ctx->dec_pos, base,
                           ctx->ids + (size_t)b0 * ctx->gp.Tids, ctx->gp.Tids, 1, Mper, d, ctx->V, ctx->Tmax);
    else
        hipLaunchKernelGGL(k_embed, dim3(R), dim3(256), 0, st, ctx->h, ctx->tok_emb, ctx->dec_pos, base,
                           ctx->cand + (size_t)b0 * 16, 16, 0, Mper, d, ctx->V, ctx->Tmax);
    WM_HIP(hipGetLastError());
    for (int l = 0; l < ctx->cfg.dec_layers; ++l) {
        int rc = dec_layer(ctx, ctx->dec[l], l, ctx->h, b0, nb, Mper, base, false);
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
    const int* base = (mode == 0 ? ctx->kvlen : ctx->L) + b0;
    hipLaunchKernelGGL(k_rows_norm, dim3((R + 3) / 4), dim3(256), 0, st, ctx->h, 1, 0, ctx->dec_lnf_w, ctx->dec_lnf_b, 1,
                       ctx->hf, ctx->block ? ctx->hblk : nullptr, nullptr, K32, 1, 0, d, R);
    WM_HIP(hipGetLastError());
    if (ctx->block && !ctx->gp.vanilla) {
        int rc = dec_layer(ctx, ctx->dec[ctx->nkv - 1], ctx->nkv - 1, ctx->hblk, b0, nb, Mper, base, !medusa);
        if (rc) return rc;
    }
    return WM_OK;
}

// ---- stage 3: Medusa heads + shared vocabulary projection for nsel selected rows (row m of the
// selection = pass row m*sel_mul + sel_off) -> ctx->logits rows [m][head] -------------------------------
int wm_dec_stage_heads(wm_ctx* ctx, int nsel, int sel_mul, int sel_off, int medusa)
{
    hipStream_t st = ctx->stream;
    const int d = ctx->d, K32 = d / 32, K = ctx->K;
    const int nout = medusa ? K + 1 : 1;
    if (nsel * nout > WM_MAX_ROWS_SKINNY) { ctx->err = "head stage exceeds 32 logit rows"; return WM_ERR_ARG; }
    if (!ctx->block) {
        // Medusa-Linear: every head (incl. base head 0) = x + SiLU(W_k x + b_k), then proj_out (model.py:1274-1284)
        WM_HIP(launch_skinny(st, ctx->heads_w, nout * d / 16, K32, nsel,
                             LdNorm{ctx->hf, nullptr, nullptr, nullptr, d, K32, nsel, sel_mul, sel_off, 0},
                             EpHead{ctx->ybuf, ctx->hf, ctx->heads_b, d, K32, nout, 0, nsel, sel_mul, sel_off}));
    } else {
        // Medusa-Block: base logits = proj_out(hf) (model.py:1287); K heads on the block output (model.py:1414-1417)
        hipLaunchKernelGGL(k_rows_norm, dim3((nsel + 3) / 4), dim3(256), 0, st, ctx->hf, sel_mul, sel_off, nullptr, nullptr, 0,
                           nullptr, nullptr, ctx->ybuf, K32, nout, 0, d, nsel);
        WM_HIP(hipGetLastError());
        if (medusa)
            WM_HIP(launch_skinny(st, ctx->heads_w, K * d / 16, K32, nsel,
                                 LdNorm{ctx->hblk, nullptr, nullptr, nullptr, d, K32, nsel, sel_mul, sel_off, 0},
                                 EpHead{ctx->ybuf, ctx->hblk, ctx->heads_b, d, K32, nout, 1, nsel, sel_mul, sel_off}));
    }
    // shared vocabulary projection (tied proj_out, model.py:1277)
    WM_HIP(launch_skinny(st, ctx->vocab_w, ctx->Vpad / 16, K32, nsel * nout, LdPacked{ctx->ybuf, K32},
                         EpF32{ctx->logits, nullptr, ctx->Vpad, nsel * nout, 1.0f}));
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

// One full Medusa iteration (or one vanilla step) over all Bdec streams, chunked to <= 32 rows.
int wm_dec_iteration(wm_ctx* ctx, int Mper_base)
{
    hipStream_t st = ctx->stream;
    const int B = ctx->Bdec, K = ctx->K, rps = K + 1;
    const GenDev gp = ctx->gp;
    if (gp.vanilla) {
        const int chunk = max(1, WM_MAX_ROWS_SKINNY / Mper_base);
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = min(chunk, B - b0);
            int rc = wm_dec_pass(ctx, b0, nb, Mper_base, 0, 0, 0);
            if (rc) return rc;
            hipLaunchKernelGGL(k_select, dim3(nb), dim3(512), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L + b0,
                               ctx->cand + b0 * 16, 1, 0, b0, ctx->amax, ctx->pc, ctx->ent);
            WM_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(k_accept_vanilla1, dim3((B + 63) / 64), dim3(64), 0, st, gp, B, ctx->amax, ctx->ids, ctx->L,
                           ctx->kvlen, ctx->finished, ctx->niter, ctx->hist);
        WM_HIP(hipGetLastError());
        return WM_OK;
    }
    const int chunk = max(1, WM_MAX_ROWS_SKINNY / max(Mper_base, rps));
    // (a) base pass -> K+1 candidates per stream
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = min(chunk, B - b0);
        int rc = wm_dec_pass(ctx, b0, nb, Mper_base, 0, 1, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_select, dim3(nb * rps), dim3(512), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L + b0,
                           ctx->cand + b0 * 16, rps, 0, b0 * rps, ctx->amax, ctx->pc, ctx->ent);
        WM_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_set_cand, dim3((nb * rps + 63) / 64), dim3(64), 0, st, ctx->amax + b0 * rps, ctx->cand + b0 * 16,
                           rps, nb * rps);
        WM_HIP(hipGetLastError());
    }
    // (d) verify pass over the candidates at positions L..L+K, then posterior statistics
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = min(chunk, B - b0);
        int rc = wm_dec_pass(ctx, b0, nb, rps, 1, 0, 1);
        if (rc) return rc;
        hipLaunchKernelGGL(k_select, dim3(nb * rps), dim3(512), 0, st, ctx->logits, gp, ctx->supmask, ctx->exppen, ctx->L + b0,
                           ctx->cand + b0 * 16, rps, gp.accept_mode == WM_ACCEPT_TYPICAL ? 1 : 0, b0 * rps, ctx->amax, ctx->pc,
                           ctx->ent);
        WM_HIP(hipGetLastError());
    }
    // (f)-(j) accept / emit / compact / stop
    hipLaunchKernelGGL(k_accept, dim3(B), dim3(64), 0, st, gp, ctx->cand, ctx->amax, ctx->pc, ctx->ent, ctx->ids, ctx->L,
                       ctx->kvlen, ctx->finished, ctx->niter, ctx->hist);
    WM_HIP(hipGetLastError());
    return WM_OK;
}

// Times the six weight-streaming GEMMs of one decoder layer for `rows` token rows (bench.py roofline leg).
int wm_dec_profile(wm_ctx* ctx, int kernel, int rows, int reps, float* ms, double* bytes)
{
    if (kernel != 0 || rows < 1 || rows > WM_MAX_ROWS_SKINNY) { ctx->err = "wm_profile_kernel: bad arguments"; return WM_ERR_ARG; }
    hipStream_t st = ctx->stream;
    const int d = ctx->d, K32 = d / 32, R = rows, H = ctx->H;
    const DecLayerW& w = ctx->dec[0];
    WM_HIP(hipMemsetAsync(ctx->h, 0, (size_t)WM_MAX_ROWS_SKINNY * d * sizeof(float), st));
    WM_HIP(hipMemsetAsync(ctx->kvlen, 0, sizeof(int) * ctx->maxB, st));
    auto body = [&]() -> int {
        WM_HIP(launch_skinny(st, w.qkv_w, 3 * d / 16, K32, R, LdNorm{ctx->h, w.ln1_w, w.ln1_b, nullptr, d, K32, R, 1, 0, 1},
                             EpQKVDec{ctx->qbuf, ctx->kc, ctx->vc, w.qkv_b, ctx->kvlen, R, d, H, ctx->Tal, R}));
        WM_HIP(launch_skinny(st, w.out_w, d / 16, K32, R, LdPacked{ctx->xbuf, K32}, EpResidual{ctx->h, w.out_b, d, R}));
        WM_HIP(launch_skinny(st, w.cq_w, d / 16, K32, R, LdNorm{ctx->h, w.ln2_w, w.ln2_b, nullptr, d, K32, R, 1, 0, 1},
                             EpF32{ctx->qbuf, w.cq_b, d, R, 0.125f}));
        WM_HIP(launch_skinny(st, w.cout_w, d / 16, K32, R, LdPacked{ctx->xbuf, K32}, EpResidual{ctx->h, w.cout_b, d, R}));
        WM_HIP(launch_skinny(st, w.fc1_w, ctx->ffn / 16, K32, R, LdNorm{ctx->h, w.ln3_w, w.ln3_b, nullptr, d, K32, R, 1, 0, 1},
                             EpPackedAct<1>{ctx->fbuf, w.fc1_b, ctx->ffn / 32, R}));
        WM_HIP(launch_skinny(st, w.fc2_w, d / 16, ctx->ffn / 32, R, LdPacked{ctx->fbuf, ctx->ffn / 32},
                             EpResidual{ctx->h, w.fc2_b, d, R}));
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
    *bytes = 2.0 * ((double)(3 + 1 + 1 + 1) * d * d + 2.0 * (double)d * ctx->ffn);
    return WM_OK;
}
