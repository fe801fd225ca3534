// wm_engine.hip — C-ABI entry points of libwm.so (include/wm.h): context, parameter table,
// HBM allocation, decode-loop driver with hipGraph replay, parity taps.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "wm_internal.h"

static thread_local std::string g_create_err;

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

template <class T>
static hipError_t dev_alloc(T** p, size_t n, hipStream_t st)
{
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, 0, std::max<size_t>(n, 1) * sizeof(T), st);
}

extern "C" int wm_abi_version(void) { return WM_ABI_VERSION; }
extern "C" int wm_build_act_fp16(void) { return WM_ACT_PLANES == 1 ? 1 : 0; }

extern "C" const char* wm_last_error(const wm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" void wm_destroy(wm_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->graph) hipGraphExecDestroy(ctx->graph);
    if (ctx->graph_base) hipGraphExecDestroy(ctx->graph_base);
    if (ctx->hostflags) hipHostFree(ctx->hostflags);
    void* bufs[] = {ctx->feats_own, ctx->clipmax, ctx->A1, ctx->a1, ctx->A2, ctx->eh, ctx->exn, ctx->eq, ctx->ek, ctx->evt, ctx->eff,
                    ctx->enc_out, ctx->kx, ctx->vx, ctx->kc, ctx->vc, ctx->h, ctx->hblk, ctx->hf, ctx->qbuf, ctx->xbuf, ctx->fbuf,
                    ctx->ybuf, ctx->cml, ctx->co, ctx->ticket, ctx->logits, ctx->amax, ctx->pc, ctx->part1, ctx->part2, ctx->ids, ctx->L, ctx->kvlen,
                    ctx->finished, ctx->cand, ctx->niter, ctx->hist, ctx->supmask, ctx->exppen, ctx->tap_tok, ctx->done,
                    ctx->hf_keep, ctx->hb_keep, ctx->carry, ctx->rowinfo, ctx->sinfo, ctx->steprows, ctx->rs_table, ctx->tree, ctx->sibtree, ctx->sibpart, ctx->sel_src, ctx->sel_n, ctx->sel_base, ctx->exn8, ctx->exs,
                    ctx->xn, ctx->lnstats, ctx->foldv, ctx->kx8, ctx->vx8, ctx->kxs, ctx->vxs};
    for (void* b : bufs) if (b) hipFree(b);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int wm_create(const wm_config* cfg, const wm_weights* w, int device, void* hip_stream, wm_ctx** out)
{
    g_create_err.clear();
    if (!cfg || !w || !out) { g_create_err = "wm_create: null argument"; return WM_ERR_ARG; }
    if (cfg->abi_version != WM_ABI_VERSION) { g_create_err = "wm_create: ABI version mismatch"; return WM_ERR_ARG; }
    if ((cfg->act_fp16 != 0) != (WM_ACT_PLANES == 1)) {
        g_create_err = std::string("wm_create: this library is built for the ") + (WM_ACT_PLANES == 1 ? "fp16 single-plane" : "bf16 hi / lo") +
                       " decode contract (wm_build_act_fp16), wm_config.act_fp16 asks for the other"; return WM_ERR_ARG; }
    if (cfg->d_model <= 0 || cfg->d_model % 128 || cfg->n_heads * WM_HEAD_DIM != cfg->d_model || cfg->d_model > 2048) {
        g_create_err = "wm_create: d_model must be a multiple of 128 (<= 2048) with 64-wide heads"; return WM_ERR_ARG; }
    if (cfg->ffn_dim % 128 || cfg->medusa_heads < 1 || cfg->medusa_heads > 15 || cfg->max_batch < 1 || cfg->n_mels * 3 > 256 ||
        (cfg->heads_type != WM_HEADS_LINEAR && cfg->heads_type != WM_HEADS_BLOCK) || cfg->n_tgt < 8 || cfg->n_ctx < 8) {
        g_create_err = "wm_create: unsupported configuration"; return WM_ERR_ARG; }
    // candidate tree (medusa_choices): zero / all-one entries = the chain
    TreeDev tree{};
    int tn = 0, tp = 0;
    {
        const int K = cfg->medusa_heads;
        bool chain = true, zero = true;
        for (int i = 0; i < 16; ++i) { zero = zero && cfg->medusa_choices[i] == 0; }
        for (int i = 0; i <= K; ++i) chain = chain && cfg->medusa_choices[i] == 1;
        if (!zero && !chain) {
            const int32_t* c = cfg->medusa_choices;
            bool ok = c[0] == 1;
            for (int i = 0; i <= K; ++i) ok = ok && c[i] >= 1 && c[i] <= 4;
            for (int i = K + 1; i < 16; ++i) ok = ok && c[i] == 0;
            long nodes = 0, prod = 1;
            for (int i = 0; ok && i <= K; ++i) { prod *= c[i]; nodes += prod; if (prod > WM_TREE_MAX_PATHS || nodes > WM_TREE_MAX_NODES) ok = false; }
            if (!ok) { g_create_err = "wm_create: medusa_choices must be [1, c_1..c_K] with c_k in 1..4, <= 64 tree nodes and <= 32 paths"; return WM_ERR_ARG; }
            // generate_medusa_buffers (medusa_utils.py:305-421): node (depth i, index j) has parent (i-1, j / c_i) and takes
            // the (j % c_i)-th of head i's top-c_i tokens; path p visits node (i, p / (n_paths / cumprod_i)) at depth i
            tree.K = K; tree.n_paths = (int)prod; tree.n_nodes = (int)nodes;
            int cp = 1;
            tree.start[0] = 0;
            for (int i = 0; i <= K; ++i) { cp *= c[i]; tree.cumprod[i] = cp; tree.topk[i] = c[i]; tree.start[i + 1] = tree.start[i] + cp; }
            for (int n = 0; n < WM_TREE_MAX_NODES; ++n) { tree.parent[n] = -1; for (int j = 0; j < 4; ++j) tree.children[n][j] = -1; }
            for (int i = 0; i <= K; ++i)
                for (int j = 0; j < tree.cumprod[i]; ++j) {
                    const int n = tree.start[i] + j;
                    tree.depth[n] = i;
                    if (i > 0) {
                        const int par = tree.start[i - 1] + j / c[i];
                        tree.parent[n] = par;
                        tree.children[par][j % c[i]] = n;
                    }
                    tree.anc[n] = (1ull << n) | (i > 0 ? tree.anc[tree.parent[n]] : 0ull);
                }
            for (int p = 0; p < tree.n_paths; ++p)
                for (int i = 0; i <= K; ++i) tree.retrieve[p][i] = tree.start[i] + p / (tree.n_paths / tree.cumprod[i]);
            tn = tree.n_nodes; tp = tree.n_paths;
        }
    }
    wm_ctx* ctx = new wm_ctx();
    ctx->cfg = *cfg; ctx->device = device;
    ctx->tn = tn; ctx->tp = tp; ctx->tree_host = tree;
#define CREATE_HIP(expr)                                                                       \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                       \
        g_create_err = std::string(#expr) + ": " + hipGetErrorString(_e); wm_destroy(ctx); return WM_ERR_HIP; } } while (0)
    CREATE_HIP(hipSetDevice(device));
    if (hip_stream) ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    else { CREATE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    CREATE_HIP(hipEventCreate(&ctx->ev0));
    CREATE_HIP(hipEventCreate(&ctx->ev1));
    { const char* v = std::getenv("WM_PREFETCH"); ctx->prefetch = !(v && std::atoi(v) == 0); }

    const int d = ctx->d = cfg->d_model;
    ctx->H = cfg->n_heads; ctx->ffn = cfg->ffn_dim; ctx->V = cfg->vocab; ctx->Vpad = rup(cfg->vocab, 128);
    ctx->S = cfg->n_ctx; ctx->Spad = rup(cfg->n_ctx, 128); ctx->Tm = 2 * cfg->n_ctx; ctx->Tmpad = rup(ctx->Tm, 128);
    ctx->Mmax = std::max(16, rup(tn, 16));      // rows per stream of a pass: the chain's K + 1 <= 16, or the tree's nodes in 16-row query tiles
    ctx->Tmax = cfg->n_tgt; ctx->Tal = rup(cfg->n_tgt + ctx->Mmax, 32); ctx->K = cfg->medusa_heads;
    ctx->block = cfg->heads_type == WM_HEADS_BLOCK;
    ctx->nkv = cfg->dec_layers + (ctx->block ? 1 : 0);
    ctx->nres = ctx->K + (ctx->block ? 0 : 1);
    ctx->maxB = cfg->max_batch;
    ctx->K1pad = rup(3 * cfg->n_mels, 128);
    ctx->NS = (ctx->Spad + 255) / 256;
    if (ctx->NS > 8 || ctx->H > 32) { g_create_err = "wm_create: n_ctx > 2048 or more than 32 heads unsupported"; wm_destroy(ctx); return WM_ERR_ARG; }

    // ---- parameter table ----
    const bool w8 = cfg->dec_weight_fp8 != 0;
    const bool e8 = cfg->enc_fp8 != 0;
    ctx->enc_f8 = e8;
    if (e8 && (cfg->d_model % 64 || cfg->ffn_dim % 128)) { g_create_err = "wm_create: enc_fp8 needs d_model % 64 == 0 and ffn_dim % 128 == 0"; wm_destroy(ctx); return WM_ERR_ARG; }
    const int n_expected = 19 + 12 * cfg->enc_layers + 18 * ctx->nkv + (w8 ? 6 * ctx->nkv : 0) + (e8 ? 4 * cfg->enc_layers + 2 : 0);
    if (w->n_offsets != n_expected || !w->blob || !w->offsets) {
        g_create_err = "wm_create: weight table has " + std::to_string(w->n_offsets) + " entries, expected " + std::to_string(n_expected);
        wm_destroy(ctx); return WM_ERR_ARG;
    }
    const char* base = reinterpret_cast<const char*>(w->blob);
    for (int i = 0; i < w->n_offsets; ++i)
        if (w->offsets[i] >= w->blob_bytes || (w->offsets[i] & 15)) { g_create_err = "wm_create: bad weight offset"; wm_destroy(ctx); return WM_ERR_ARG; }
    int t = 0;
    auto F = [&]() { return reinterpret_cast<const float*>(base + w->offsets[t++]); };
    auto Hh = [&]() { return reinterpret_cast<const bf16_t*>(base + w->offsets[t++]); };
    ctx->win = F(); ctx->twiddle = F(); ctx->melfb = F();
    ctx->conv1_w = Hh(); ctx->conv1_b = F(); ctx->conv2_w = Hh(); ctx->conv2_b = F();
    ctx->enc_pos = F(); ctx->enc_lnf_w = F(); ctx->enc_lnf_b = F();
    ctx->tok_emb = Hh(); ctx->vocab_w = Hh(); ctx->dec_pos = F(); ctx->dec_lnf_w = F(); ctx->dec_lnf_b = F();
    ctx->heads_w = Hh(); ctx->heads_b = F(); ctx->ckv_w = Hh(); ctx->ckv_b = F();
    ctx->enc.resize(cfg->enc_layers);
    for (auto& e : ctx->enc) {
        e.ln1_w = F(); e.ln1_b = F(); e.qkv_w = Hh(); e.qkv_b = F(); e.out_w = Hh(); e.out_b = F();
        e.ln2_w = F(); e.ln2_b = F(); e.fc1_w = Hh(); e.fc1_b = F(); e.fc2_w = Hh(); e.fc2_b = F();
    }
    ctx->dec.resize(ctx->nkv);
    for (auto& e : ctx->dec) {
        e.ln1_w = F(); e.ln1_b = F(); e.qkv_w = Hh(); e.qkv_b = F(); e.out_w = Hh(); e.out_b = F();
        e.ln2_w = F(); e.ln2_b = F(); e.cq_w = Hh(); e.cq_b = F(); e.cout_w = Hh(); e.cout_b = F();
        e.ln3_w = F(); e.ln3_b = F(); e.fc1_w = Hh(); e.fc1_b = F(); e.fc2_w = Hh(); e.fc2_b = F();
    }
    if (w8)                                    // fp8 e4m3 decoder-layer matrices: one fp32 scale per output row, appended to the table
        for (auto& e : ctx->dec) { e.qkv_s = F(); e.out_s = F(); e.cq_s = F(); e.cout_s = F(); e.fc1_s = F(); e.fc2_s = F(); }
    if (e8) {                                  // fp8 MFMA encoder: e4m3 matrices (128-k unit layout, K zero-padded to 128) + per-row scales, appended last
        auto U8 = [&]() { return reinterpret_cast<const unsigned char*>(base + w->offsets[t++]); };
        for (auto& e : ctx->enc) { e.qkv_w8 = U8(); e.qkv_ws = F(); e.fc1_w8 = U8(); e.fc1_ws = F(); }
        ctx->ckv_w8 = U8(); ctx->ckv_ws = F();
    }

    // ---- HBM allocation ----
    hipStream_t st = ctx->stream;
    const size_t B = ctx->maxB, Spad = ctx->Spad, H = ctx->H, Tal = ctx->Tal;
    const size_t Menc = B * Spad;
    CREATE_HIP(dev_alloc(&ctx->feats_own, B * cfg->n_mels * ctx->Tm, st));
    CREATE_HIP(dev_alloc(&ctx->clipmax, B, st));
    CREATE_HIP(dev_alloc(&ctx->A1, B * ctx->Tmpad * ctx->K1pad, st));
    CREATE_HIP(dev_alloc(&ctx->a1, B * ctx->Tm * d, st));
    CREATE_HIP(dev_alloc(&ctx->A2, Menc * 3 * d, st));
    CREATE_HIP(dev_alloc(&ctx->eh, Menc * d, st));
    CREATE_HIP(dev_alloc(&ctx->exn, Menc * d, st));
    if (e8) { CREATE_HIP(dev_alloc(&ctx->exn8, Menc * (size_t)((d + 127) / 128 * 128), st)); CREATE_HIP(dev_alloc(&ctx->exs, Menc, st)); }
    CREATE_HIP(dev_alloc(&ctx->eq, Menc * d, st));
    CREATE_HIP(dev_alloc(&ctx->ek, Menc * d, st));
    CREATE_HIP(dev_alloc(&ctx->evt, Menc * d, st));
    CREATE_HIP(dev_alloc(&ctx->eff, Menc * ctx->ffn, st));
    CREATE_HIP(dev_alloc(&ctx->enc_out, Menc * d, st));
    CREATE_HIP(dev_alloc(&ctx->kx, (size_t)ctx->nkv * Menc * d, st));
    CREATE_HIP(dev_alloc(&ctx->vx, (size_t)ctx->nkv * Menc * d, st));
    ctx->xkv8 = cfg->cross_kv_fp8 != 0;
    if (ctx->xkv8) {
        CREATE_HIP(dev_alloc(&ctx->kx8, (size_t)ctx->nkv * Menc * d, st));
        CREATE_HIP(dev_alloc(&ctx->vx8, (size_t)ctx->nkv * Menc * d, st));
        CREATE_HIP(dev_alloc(&ctx->kxs, (size_t)ctx->nkv * B * H, st));
        CREATE_HIP(dev_alloc(&ctx->vxs, (size_t)ctx->nkv * B * H, st));
    }
    CREATE_HIP(dev_alloc(&ctx->kc, (size_t)ctx->nkv * B * H * Tal * 64, st));
    CREATE_HIP(dev_alloc(&ctx->vc, (size_t)ctx->nkv * B * H * Tal * 64, st));
    ctx->Rcap = ctx->Mmax * ctx->maxB;
    const size_t RW = ctx->Rcap;
    CREATE_HIP(dev_alloc(&ctx->h, RW * d, st));
    CREATE_HIP(dev_alloc(&ctx->hblk, RW * d, st));
    CREATE_HIP(dev_alloc(&ctx->hf, RW * d, st));
    CREATE_HIP(dev_alloc(&ctx->hf_keep, B * d, st));
    CREATE_HIP(dev_alloc(&ctx->hb_keep, B * d, st));
    CREATE_HIP(dev_alloc(&ctx->carry, B, st));
    CREATE_HIP(dev_alloc(&ctx->rowinfo, (size_t)ctx->Mmax * B, st)); CREATE_HIP(dev_alloc(&ctx->sinfo, B, st)); CREATE_HIP(dev_alloc(&ctx->steprows, 4, st));
    CREATE_HIP(dev_alloc(&ctx->sel_src, B * 16, st));
    CREATE_HIP(dev_alloc(&ctx->sel_n, B, st));
    CREATE_HIP(dev_alloc(&ctx->sel_base, B, st));
    if (ctx->tn) {
        CREATE_HIP(dev_alloc(&ctx->tree, 1, st));
        CREATE_HIP(hipMemcpyAsync(ctx->tree, &ctx->tree_host, sizeof(TreeDev), hipMemcpyHostToDevice, st));
    }
    // sibling rows (wm_config.sibling_rows): the chain's K + 1 nodes, then S leaves under the root — node K + 1 + j sits at depth 1 and sees the
    // history, the root and itself.  Only the depth / ancestor tables are used (k_embed, k_attn_mfma); candidates and acceptance stay the chain's.
    ctx->sib_cfg = 0;
    if (ctx->tn == 0 && cfg->sibling_rows > 0) {
        const int K = cfg->medusa_heads, S = std::min({(int)cfg->sibling_rows, 15 - K, 5});
        if (S > 0) {
            TreeDev sib{};               // (host staging: the copy below is awaited before the block ends)
            sib.K = K; sib.n_nodes = K + 1 + S; sib.n_paths = 1;
            for (int n = 0; n <= K; ++n) { sib.depth[n] = n; sib.parent[n] = n - 1; sib.anc[n] = (2ull << n) - 1ull; }
            for (int j = 0; j < S; ++j) { const int n = K + 1 + j; sib.depth[n] = 1; sib.parent[n] = 0; sib.anc[n] = 1ull | (1ull << n); }
            CREATE_HIP(dev_alloc(&ctx->sibtree, 1, st));
            CREATE_HIP(dev_alloc(&ctx->sibpart, (size_t)cfg->max_batch * 16 * 6, st));      // SEL_SP = 16 slices
            CREATE_HIP(hipMemcpyAsync(ctx->sibtree, &sib, sizeof(TreeDev), hipMemcpyHostToDevice, st));
            CREATE_HIP(hipStreamSynchronize(st));
            ctx->sib_cfg = S;
        }
    }
    CREATE_HIP(dev_alloc(&ctx->qbuf, RW * d, st));
    CREATE_HIP(dev_alloc(&ctx->xbuf, 2 * RW * d, st));          // hi + lo planes
    { const char* v = std::getenv("WM_LN_FOLD"); ctx->ln_fold = !(v && std::atoi(v) == 0); }
    if (ctx->ln_fold) {
        CREATE_HIP(dev_alloc(&ctx->xn, 2 * RW * d, st));
        CREATE_HIP(dev_alloc(&ctx->lnstats, RW * (d / 16), st));
        CREATE_HIP(dev_alloc(&ctx->foldv, (size_t)ctx->nkv * 2 * (4 * d + ctx->ffn), st));
    }
    CREATE_HIP(dev_alloc(&ctx->fbuf, 2 * RW * ctx->ffn, st));
    CREATE_HIP(dev_alloc(&ctx->ybuf, 2 * RW * d, st));
    CREATE_HIP(dev_alloc(&ctx->cml, RW * H * ctx->NS * 2, st));
    CREATE_HIP(dev_alloc(&ctx->co, RW * H * ctx->NS * 64, st));
    CREATE_HIP(dev_alloc(&ctx->ticket, B * 32 * (WM_TREE_MAX_NODES / 16), st));
    CREATE_HIP(dev_alloc(&ctx->logits, RW * ctx->Vpad, st));
    CREATE_HIP(dev_alloc(&ctx->amax, B * WM_TREE_MAX_NODES, st));
    CREATE_HIP(dev_alloc(&ctx->pc, B * WM_TREE_MAX_NODES, st));
    CREATE_HIP(dev_alloc(&ctx->part1, RW * 16 * 4, st));
    CREATE_HIP(dev_alloc(&ctx->part2, B * WM_TREE_MAX_NODES * 16, st));
    const size_t Tids = ctx->Tal;
    CREATE_HIP(dev_alloc(&ctx->ids, B * Tids, st));
    CREATE_HIP(dev_alloc(&ctx->L, B, st));
    CREATE_HIP(dev_alloc(&ctx->kvlen, B, st));
    CREATE_HIP(dev_alloc(&ctx->finished, B, st));
    CREATE_HIP(dev_alloc(&ctx->cand, B * WM_CAND_STRIDE, st));
    CREATE_HIP(dev_alloc(&ctx->niter, B, st));
    CREATE_HIP(dev_alloc(&ctx->hist, 32, st));
    CREATE_HIP(dev_alloc(&ctx->supmask, (size_t)ctx->Vpad, st));
    CREATE_HIP(dev_alloc(&ctx->exppen, Tids + 1, st));
    CREATE_HIP(dev_alloc(&ctx->tap_tok, 16, st));
    CREATE_HIP(dev_alloc(&ctx->done, 4, st));
    CREATE_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->hostflags), 64, hipHostMallocMapped));
    CREATE_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->hostflags_dev), ctx->hostflags, 0));
    ctx->hostflags[0] = ctx->hostflags[1] = 0;
    if (ctx->ln_fold && wm_dec_fold_init(ctx) != WM_OK) { g_create_err = "wm_create: " + ctx->err; wm_destroy(ctx); return WM_ERR_HIP; }
    CREATE_HIP(hipStreamSynchronize(st));
#undef CREATE_HIP
    *out = ctx;
    return WM_OK;
}

extern "C" int wm_sync(wm_ctx* ctx) { WM_HIP(hipSetDevice(ctx->device)); WM_HIP(hipStreamSynchronize(ctx->stream)); return WM_OK; }

extern "C" int64_t wm_resample_len(int64_t n_in, int sr_in, int sr_out) { return wm_enc_resample_len((long)n_in, sr_in, sr_out); }

extern "C" int wm_resample(wm_ctx* ctx, const float* in, int B, int channels, int n_in, int sr_in, int sr_out, float* out)
{
    if (!ctx || !in || !out) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    return wm_enc_resample(ctx, in, B, channels, n_in, sr_in, sr_out, out);
}

extern "C" int wm_logmel(wm_ctx* ctx, const float* wav, int B, int n_samples, float* feats)
{
    if (!ctx || !wav || !feats) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    return wm_enc_logmel(ctx, wav, B, n_samples, feats);
}

extern "C" int wm_encode(wm_ctx* ctx, const float* feats, int B)
{
    if (!ctx || !feats) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    ctx->began = false;
    return wm_enc_encode(ctx, feats, B);
}

extern "C" int wm_set_encoder_output(wm_ctx* ctx, const float* hidden, int B)
{
    if (!ctx || !hidden) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    ctx->began = false;
    return wm_enc_set_output(ctx, hidden, B);
}

extern "C" int wm_decode_begin(wm_ctx* ctx, const wm_gen_params* gp, int B)
{
    if (!ctx || !gp) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    if (ctx->Benc < 1 || B != ctx->Benc) { ctx->err = "wm_decode_begin: call wm_encode with the same B first"; return WM_ERR_STATE; }
    const int P = gp->prompt_len, K = ctx->K, Tids = ctx->Tal;
    if (P < 1 || P >= ctx->Tmax - K - 1) {
        // same condition class as the reference's over-long prompt ValueError (model.py:1526-1529)
        ctx->err = "wm_decode_begin: prompt length must be in [1, n_tgt - K - 2]"; return WM_ERR_ARG; }
    if (gp->eos_token_id < 0 || gp->eos_token_id >= ctx->V) { ctx->err = "wm_decode_begin: eos out of range"; return WM_ERR_ARG; }
    if (!gp->vanilla && gp->accept_mode == WM_ACCEPT_TYPICAL && !(gp->temperature > 0.f)) {
        ctx->err = "wm_decode_begin: typical acceptance needs temperature > 0"; return WM_ERR_ARG; }
    hipStream_t st = ctx->stream;
    GenDev g{};
    g.P = P; g.eos = gp->eos_token_id; g.pad = gp->pad_token_id;
    g.max_length = std::min(gp->max_length, ctx->Tmax);
    g.hard_max_length = std::min(gp->hard_max_length, ctx->Tmax);
    g.exp_start = gp->exp_decay_start >= 0 ? gp->exp_decay_start + P : -1;
    g.thr = gp->posterior_threshold; g.alpha = gp->posterior_alpha;
    g.inv_temp = (gp->accept_mode == WM_ACCEPT_TYPICAL && gp->temperature > 0.f) ? 1.0f / gp->temperature : 1.0f;
    g.force_accept = gp->force_accept;
    g.begin = gp->begin_index >= 0 ? gp->begin_index : P;
    g.accept_mode = gp->accept_mode; g.vanilla = gp->vanilla; g.K = K; g.V = ctx->V; g.Vpad = ctx->Vpad; g.Tids = Tids;
    ctx->fuse = std::getenv("WM_NO_CARRY") == nullptr;
    ctx->host_carry = ctx->fuse && B == 1 && !gp->vanilla;
    ctx->dev_carry = ctx->fuse && B > 1 && !gp->vanilla;
    // several streams with the per-stream carry, chain candidates: the merged-step schedule (a stream that needs a base pass gets its one
    // row inside the other streams' verify pass: wm_dec_step).  WM_NO_STEP=1: the lock-step iteration (base pass + verify pass for everybody).
    ctx->step_flow = ctx->dev_carry && ctx->tn == 0 && std::getenv("WM_NO_STEP") == nullptr;
    g.fuse = ctx->host_carry ? 1 : (ctx->dev_carry ? (ctx->step_flow ? 3 : 2) : 0);
    g.sib = (ctx->host_carry && ctx->tn == 0 && ctx->sib_cfg > 0 && std::getenv("WM_NO_SIBLINGS") == nullptr) ? ctx->sib_cfg : 0;
    const bool same = ctx->graph && ctx->graph_B == B && std::memcmp(&g, &ctx->gp, sizeof(GenDev)) == 0;
    if (!same && ctx->graph) { hipGraphExecDestroy(ctx->graph); ctx->graph = nullptr; }
    if (!same && ctx->graph_base) { hipGraphExecDestroy(ctx->graph_base); ctx->graph_base = nullptr; }
    ctx->gp = g; ctx->Bdec = B;

    std::vector<int> ids((size_t)B * Tids, gp->pad_token_id), L(B, P), zero(B, 0);
    for (int b = 0; b < B; ++b) for (int i = 0; i < P; ++i) ids[(size_t)b * Tids + i] = gp->prompt[i];
    std::vector<unsigned char> mask(ctx->Vpad, 0);
    for (int i = 0; i < gp->n_suppress; ++i) if (gp->suppress[i] >= 0 && gp->suppress[i] < ctx->V) mask[gp->suppress[i]] |= 1;
    for (int i = 0; i < gp->n_begin_suppress; ++i)
        if (gp->begin_suppress[i] >= 0 && gp->begin_suppress[i] < ctx->V) mask[gp->begin_suppress[i]] |= 2;
    std::vector<float> pen(Tids + 1, 0.f);
    if (g.exp_start >= 0)
        for (int t = 0; t <= Tids; ++t)     // (factor^(cur_len - start) - 1) evaluated in double like the Python scalar
            pen[t] = t > g.exp_start ? (float)(std::pow((double)gp->exp_decay_factor, (double)(t - g.exp_start)) - 1.0) : 0.f;
    WM_HIP(hipMemcpyAsync(ctx->ids, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, st));
    WM_HIP(hipMemcpyAsync(ctx->L, L.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
    WM_HIP(hipMemcpyAsync(ctx->kvlen, zero.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
    WM_HIP(hipMemcpyAsync(ctx->finished, zero.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
    WM_HIP(hipMemcpyAsync(ctx->niter, zero.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
    WM_HIP(hipMemsetAsync(ctx->hist, 0, 32 * sizeof(long long), st));
    WM_HIP(hipMemsetAsync(ctx->done, 0, 4 * sizeof(int), st));
    WM_HIP(hipMemsetAsync(ctx->carry, 0, ctx->maxB * sizeof(int), st));
    WM_HIP(hipMemsetAsync(ctx->steprows, 0, 4 * sizeof(int), st));
    ctx->use_done = true;
    WM_HIP(hipMemcpyAsync(ctx->supmask, mask.data(), mask.size(), hipMemcpyHostToDevice, st));
    WM_HIP(hipMemcpyAsync(ctx->exppen, pen.data(), pen.size() * sizeof(float), hipMemcpyHostToDevice, st));
    WM_HIP(hipStreamSynchronize(st));      // host vectors go out of scope
    ctx->hostflags[0] = 0; ctx->hostflags[1] = 0;
    ctx->began = true; ctx->first_done = false; ctx->iters = 0; ctx->ms_decode = 0.f; ctx->graph_replays = 0;
    return WM_OK;
}

static int count_unfinished(wm_ctx* ctx, int* n)
{
    std::vector<int> f(ctx->Bdec);
    WM_HIP(hipMemcpyAsync(f.data(), ctx->finished, ctx->Bdec * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    int c = 0;
    for (int v : f) c += v ? 0 : 1;
    *n = c;
    return WM_OK;
}

static int capture_graph(wm_ctx* ctx, hipGraphExec_t* out, int (*body)(wm_ctx*, int))
{
    hipStream_t st = ctx->stream;
    hipGraph_t gr = nullptr;
    *out = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return WM_OK; }
    const int rc = body(ctx, 1);
    const hipError_t e2 = hipStreamEndCapture(st, &gr);
    if (rc == WM_OK && e2 == hipSuccess && gr) {
        hipGraphExec_t ex = nullptr;
        if (hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) == hipSuccess) *out = ex;
    }
    if (gr) hipGraphDestroy(gr);
    (void)hipGetLastError();
    return WM_OK;
}

extern "C" int wm_decode_run(wm_ctx* ctx, int max_iters, int* n_unfinished)
{
    if (!ctx) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    if (!ctx->began) { ctx->err = "wm_decode_run: call wm_decode_begin first"; return WM_ERR_STATE; }
    hipStream_t st = ctx->stream;
    const bool use_graph = std::getenv("WM_NO_GRAPH") == nullptr;
    int left = 0, done = 0;
    int rc = count_unfinished(ctx, &left);
    if (rc) return rc;
    WM_HIP(hipEventRecord(ctx->ev0, st));
    if (ctx->host_carry) {
        // Single stream: the host reads {carry, finished} (host-mapped, written by k_accept) after every iteration and
        // launches the base-pass graph only when the hidden state was NOT carried over from the verify pass.
        while (left > 0 && done < max_iters) {
            if (!ctx->first_done) {
                const int p_last = wm_dec_prompt_prefix(ctx, ctx->gp.P);      // long prompts: leading chunks for their K/V only
                if (p_last < 1) return WM_ERR_HIP;
                rc = wm_dec_iteration(ctx, p_last);               // iteration 1: the base pass consumes the (last <= 16) prompt tokens
                if (rc) return rc;
                ctx->first_done = true;
            } else {
                const bool need_base = ctx->hostflags[0] == 0;
                const bool warm = ctx->iters >= 2;
                if (need_base) {
                    if (use_graph && warm && !ctx->graph_base) { rc = capture_graph(ctx, &ctx->graph_base, wm_dec_iter_base); if (rc) return rc; }
                    if (use_graph && ctx->graph_base) WM_HIP(hipGraphLaunch(ctx->graph_base, st));
                    else { rc = wm_dec_iter_base(ctx, 1); if (rc) return rc; }
                }
                if (use_graph && warm && !ctx->graph) { rc = capture_graph(ctx, &ctx->graph, wm_dec_iter_rest); if (rc) return rc; ctx->graph_B = ctx->Bdec; }
                if (use_graph && ctx->graph) { WM_HIP(hipGraphLaunch(ctx->graph, st)); ctx->graph_replays++; }
                else { rc = wm_dec_iter_rest(ctx, 1); if (rc) return rc; }
            }
            WM_HIP(hipStreamSynchronize(st));
            ctx->iters++; done++;
            left = ctx->hostflags[1] ? 0 : 1;
        }
    } else {
        const int poll = 4;
        while (left > 0 && done < max_iters) {
            const int burst = std::min(poll, max_iters - done);
            for (int i = 0; i < burst; ++i) {
                if (!ctx->first_done) {                       // iteration 1: the base pass consumes the P prompt tokens
                    const int p_last = wm_dec_prompt_prefix(ctx, ctx->gp.P);
                    if (p_last < 1) return WM_ERR_HIP;
                    rc = wm_dec_iteration(ctx, p_last);
                    if (rc) return rc;
                    ctx->first_done = true;
                } else {
                    int (*body)(wm_ctx*, int) = ctx->step_flow ? wm_dec_step : wm_dec_iteration;
                    if (use_graph && ctx->iters >= 2 && !ctx->graph) {   // shapes are warm: capture one steady-state iteration / step
                        rc = capture_graph(ctx, &ctx->graph, body);
                        if (rc) return rc;
                        ctx->graph_B = ctx->Bdec;
                    }
                    if (use_graph && ctx->graph) { WM_HIP(hipGraphLaunch(ctx->graph, st)); ctx->graph_replays++; }
                    else { rc = body(ctx, 1); if (rc) return rc; }
                }
                ctx->iters++; done++;
            }
            rc = count_unfinished(ctx, &left);
            if (rc) return rc;
        }
    }
    WM_HIP(hipEventRecord(ctx->ev1, st));
    WM_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    WM_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->ms_decode += ms;
    if (n_unfinished) *n_unfinished = left;
    return WM_OK;
}

extern "C" int wm_get_tokens(wm_ctx* ctx, int stream, int32_t* out, int cap, int* n)
{
    if (!ctx || !out || !n || stream < 0 || stream >= ctx->Bdec) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    int L = 0;
    WM_HIP(hipMemcpyAsync(&L, ctx->L + stream, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    L = std::min(L, ctx->gp.Tids);
    std::vector<int> ids(L);
    WM_HIP(hipMemcpyAsync(ids.data(), ctx->ids + (size_t)stream * ctx->gp.Tids, L * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    bool seen = false;                       // post-EOS overwrite, model.py:798-810
    for (int i = 0; i < L; ++i) { if (seen) ids[i] = ctx->gp.eos; else if (ids[i] == ctx->gp.eos) seen = true; }
    const int m = std::min(L, cap);
    std::memcpy(out, ids.data(), m * sizeof(int));
    *n = L;
    return WM_OK;
}

extern "C" int wm_get_stats(wm_ctx* ctx, wm_stats* out)
{
    if (!ctx || !out) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    long long h[32];
    std::vector<int> ni(std::max(ctx->Bdec, 1), 0);
    WM_HIP(hipMemcpyAsync(h, ctx->hist, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->Bdec > 0) WM_HIP(hipMemcpyAsync(ni.data(), ctx->niter, ctx->Bdec * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    std::memset(out, 0, sizeof(*out));
    out->iterations = *std::max_element(ni.begin(), ni.end());
    out->iterations_launched = ctx->iters;
    for (int i = 0; i < 16; ++i) out->accept_hist[i] = h[i];
    out->tokens_emitted = h[16];
    out->ms_logmel = ctx->ms_logmel; out->ms_encode = ctx->ms_encode; out->ms_decode = ctx->ms_decode;
    out->graph_replays = ctx->graph_replays;
    out->sibling_hits = (int32_t)h[17];
    if (ctx->step_flow && ctx->steprows) {
        int sr[4] = {0, 0, 0, 0};
        WM_HIP(hipMemcpyAsync(sr, ctx->steprows, sizeof(sr), hipMemcpyDeviceToHost, ctx->stream));       // on the context's stream like every other read
        WM_HIP(hipStreamSynchronize(ctx->stream));
        out->schedule_steps = sr[2];
    }
    return WM_OK;
}

// ---- parity taps ---------------------------------------------------------------------------
static void unpack_rows(const std::vector<bf16_t>& p, int rows, int K, float* out)
{
    const int K32 = K / 32;
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            const uint32_t u = ((uint32_t)p[packed_index(r, k, K32)]) << 16;
            float f; std::memcpy(&f, &u, 4);
            out[(size_t)r * K + k] = f;
        }
}

extern "C" int wm_get_encoder_output(wm_ctx* ctx, int B, float* out)
{
    if (!ctx || !out || B < 1 || B > ctx->Benc) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)B * ctx->Spad * ctx->d;
    std::vector<bf16_t> p(n);
    WM_HIP(hipMemcpyAsync(p.data(), ctx->enc_out, n * sizeof(bf16_t), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<float> full(n);
    unpack_rows(p, B * ctx->Spad, ctx->d, full.data());
    for (int b = 0; b < B; ++b)
        std::memcpy(out + (size_t)b * ctx->S * ctx->d, full.data() + (size_t)b * ctx->Spad * ctx->d, (size_t)ctx->S * ctx->d * sizeof(float));
    return WM_OK;
}

extern "C" int wm_get_cross_kv(wm_ctx* ctx, int kv_layer, int stream, int head, float* k_out, float* v_out)
{
    if (!ctx || kv_layer < 0 || kv_layer >= ctx->nkv || stream < 0 || stream >= ctx->Benc || head < 0 || head >= ctx->H) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    const size_t off = (((size_t)kv_layer * ctx->Benc + stream) * ctx->H + head) * ctx->Spad * 64, n = (size_t)ctx->S * 64;
    const size_t nv = (size_t)ctx->Spad * 64;
    std::vector<bf16_t> kb(n), vb(nv);
    WM_HIP(hipMemcpyAsync(kb.data(), ctx->kx + off, n * sizeof(bf16_t), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipMemcpyAsync(vb.data(), ctx->vx + off, nv * sizeof(bf16_t), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i) { const uint32_t u = ((uint32_t)kb[i]) << 16; std::memcpy(k_out + i, &u, 4); }
    for (int s = 0; s < ctx->S; ++s)            // V is stored as V^T MFMA fragments
        for (int dd = 0; dd < 64; ++dd) { const uint32_t u = ((uint32_t)vb[vfrag_index(s, dd)]) << 16; std::memcpy(v_out + (size_t)s * 64 + dd, &u, 4); }
    return WM_OK;
}

// forward(): one decoder pass over T tokens per stream at positions pos0.., K/V appended at row pos0.
// Uses (and overwrites) the decode-loop state: call it before wm_decode_begin, or begin again afterwards.
extern "C" int wm_forward_logits(wm_ctx* ctx, int B, const int32_t* tokens, int T, int pos0, int disable_medusa, float* logits_out)
{
    if (!ctx || !tokens || !logits_out || B < 1 || B > ctx->Benc || T < 1 || T > 16 || pos0 < 0 || pos0 + T > ctx->Tmax) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int Tids = ctx->Tal, K = ctx->K, V = ctx->V, nout = disable_medusa ? 1 : K + 1;
    GenDev g = ctx->gp;
    g.K = K; g.V = V; g.Vpad = ctx->Vpad; g.Tids = Tids; g.vanilla = 0;
    ctx->gp = g; ctx->began = false; ctx->use_done = false; ctx->host_carry = false; ctx->dev_carry = false;
    ctx->step_flow = false;                 // (wm_get_stats must not report the previous decode's schedule_steps for this pass)
    if (ctx->graph) { hipGraphExecDestroy(ctx->graph); ctx->graph = nullptr; }
    if (ctx->graph_base) { hipGraphExecDestroy(ctx->graph_base); ctx->graph_base = nullptr; }
    std::vector<float> rowbuf((size_t)nout * ctx->Vpad);
    for (int b = 0; b < B; ++b) {
        std::vector<int> v(T);
        for (int i = 0; i < T; ++i) v[i] = tokens[(size_t)b * T + i];
        const int kv = pos0, L = pos0 + T;
        WM_HIP(hipMemcpyAsync(ctx->ids + (size_t)b * Tids + pos0, v.data(), T * sizeof(int), hipMemcpyHostToDevice, st));
        WM_HIP(hipMemcpyAsync(ctx->kvlen + b, &kv, sizeof(int), hipMemcpyHostToDevice, st));
        WM_HIP(hipMemcpyAsync(ctx->L + b, &L, sizeof(int), hipMemcpyHostToDevice, st));
        WM_HIP(hipStreamSynchronize(st));
        int rc = wm_dec_stage_layers(ctx, b, 1, T, 0);
        if (rc) return rc;
        rc = wm_dec_stage_final(ctx, b, 1, T, 0, !disable_medusa);
        if (rc) return rc;
        for (int r = 0; r < T; ++r) {
            rc = wm_dec_stage_heads(ctx, 1, 1, r, !disable_medusa);
            if (rc) return rc;
            WM_HIP(hipMemcpyAsync(rowbuf.data(), ctx->logits, rowbuf.size() * sizeof(float), hipMemcpyDeviceToHost, st));
            WM_HIP(hipStreamSynchronize(st));
            for (int k = 0; k < nout; ++k)      // out layout [n_out][B][T][V]
                std::memcpy(logits_out + (((size_t)k * B + b) * T + r) * V, rowbuf.data() + (size_t)k * ctx->Vpad, (size_t)V * sizeof(float));
        }
    }
    return WM_OK;
}

extern "C" int wm_profile_kernel(wm_ctx* ctx, int kernel, int rows, int reps, float* ms, double* bytes)
{
    if (!ctx || !ms || !bytes || reps < 1) return WM_ERR_ARG;
    WM_HIP(hipSetDevice(ctx->device));
    return wm_dec_profile(ctx, kernel, rows, reps, ms, bytes);
}
