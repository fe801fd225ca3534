// wm_internal.h — engine context shared by the three translation units of libwm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/wm.h"
#include "wm_common.h"

#define WM_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e);                   \
            return WM_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

struct EncLayerW {
    const float *ln1_w, *ln1_b, *qkv_b, *out_b, *ln2_w, *ln2_b, *fc1_b, *fc2_b;
    const bf16_t *qkv_w, *out_w, *fc1_w, *fc2_w;
    // fp8 MFMA encoder (wm_config.enc_fp8): e4m3 copies of the LayerNorm-fed matrices in the 64-k unit layout + per-row scales
    const unsigned char *qkv_w8 = nullptr, *fc1_w8 = nullptr;
    const float *qkv_ws = nullptr, *fc1_ws = nullptr;
};
struct DecLayerW {
    const float *ln1_w, *ln1_b, *qkv_b, *out_b, *ln2_w, *ln2_b, *cq_b, *cout_b, *ln3_w, *ln3_b, *fc1_b, *fc2_b;
    const bf16_t *qkv_w, *out_w, *cq_w, *cout_w, *fc1_w, *fc2_w;     // packed bf16 — or packed fp8 e4m3 when the scales below are set
    const float *qkv_s = nullptr, *out_s = nullptr, *cq_s = nullptr, *cout_s = nullptr, *fc1_s = nullptr, *fc2_s = nullptr;
    // LayerNorm folded into the three LayerNorm-fed GEMMs (wm_common.h; computed in wm_create by wm_dec_fold_init): c = W gamma, bf = b + W beta
    const float *qkv_c = nullptr, *qkv_bf = nullptr, *cq_c = nullptr, *cq_bf = nullptr, *fc1_c = nullptr, *fc1_bf = nullptr;
};

// scalars every decode kernel may need (passed by value)
struct GenDev {
    int P, eos, pad, max_length, hard_max_length, exp_start /* absolute: start + P, or -1 */;
    float thr, alpha, inv_temp;
    int accept_mode, vanilla, K, V, Vpad, Tids, fuse, force_accept;
    int begin;      // sequence length at which the begin-suppress list applies (wm_gen_params.begin_index; P when that is < 0)
    int sib;        // sibling rows of this decode's verify pass (wm_config.sibling_rows; 0 unless one stream, chain candidates, hidden-state carry)
};

// Static tables of the candidate tree (device memory; medusa_utils.py:305-421).  Nodes are numbered depth by depth.
struct TreeDev {
    int n_nodes, n_paths, K, pad_;
    int topk[16];           // c_k of logits row k (row 0 = base head: 1)
    int start[17];          // first node of depth i
    int cumprod[16];        // nodes at depth i
    int depth[WM_TREE_MAX_NODES];                   // medusa_position_ids
    unsigned long long anc[WM_TREE_MAX_NODES];      // bit n of anc[m]: node n is m or one of its ancestors (rows of medusa_attn_mask)
    int parent[WM_TREE_MAX_NODES];
    int children[WM_TREE_MAX_NODES][4];             // -1 padded
    int retrieve[WM_TREE_MAX_PATHS][16];            // [path][depth] -> node (retrieve_indices)
};

struct wm_ctx {
    wm_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // derived sizes
    int d = 0, H = 0, ffn = 0, V = 0, Vpad = 0, S = 0, Spad = 0, Tm = 0 /* mel frames 2S */, Tmpad = 0;
    int Tmax = 0 /* n_tgt */, Tal = 0 /* cache rows allocated */, K = 0, nkv = 0, nres = 0, maxB = 0;
    bool block = false;
    int NS = 1;         // cross-attention key splits (256 keys per block)
    int Rcap = 16;      // token-row capacity of the decode scratch (16 rows per stream; a candidate tree of more than 16 nodes: its node count rounded up to 16)
    int Mmax = 16;      // rows per stream a pass may carry

    // ---- parameters (pointers into the caller's blob) ----
    const float *win = nullptr, *twiddle = nullptr, *melfb = nullptr;
    const bf16_t *conv1_w = nullptr, *conv2_w = nullptr;
    const float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *enc_lnf_w = nullptr, *enc_lnf_b = nullptr;
    const bf16_t *tok_emb = nullptr, *vocab_w = nullptr, *heads_w = nullptr, *ckv_w = nullptr;
    const float *dec_pos = nullptr, *dec_lnf_w = nullptr, *dec_lnf_b = nullptr, *heads_b = nullptr, *ckv_b = nullptr;
    std::vector<EncLayerW> enc;
    std::vector<DecLayerW> dec;       // nkv entries (last = medusa_block for Block)

    // ---- encoder scratch (sized for maxB clips) ----
    float* feats_own = nullptr;       // [B][n_mels][Tm]
    float* clipmax = nullptr;         // [B] ordered-int encoded
    bf16_t* A1 = nullptr;             // packed [B*Tmpad][K1pad]
    bf16_t* a1 = nullptr;             // [B][Tm][d] bf16
    bf16_t* A2 = nullptr;             // packed [B*Spad][3d]
    float* eh = nullptr;              // [B*Spad][d] fp32 residual
    bf16_t* exn = nullptr;            // packed [B*Spad][d]
    bf16_t *eq = nullptr, *ek = nullptr, *evt = nullptr;   // [B][H][Spad][64] / vt [B][H][64][Spad]
    bf16_t* eff = nullptr;            // packed [B*Spad][ffn]
    bf16_t* enc_out = nullptr;        // packed [B*Spad][d]
    int K1pad = 0;
    bool enc_f8 = false;              // fp8 MFMA encoder path
    unsigned char* exn8 = nullptr;    // packed fp8 [B*Spad][d] (LayerNorm output as e4m3)
    float* exs = nullptr;             // [B*Spad] row scales of exn8
    const unsigned char* ckv_w8 = nullptr; const float* ckv_ws = nullptr;

    // ---- caches ----
    bf16_t *kx = nullptr, *vx = nullptr;   // cross K/V [nkv][Benc][H][Spad][64]
    // wm_config.cross_kv_fp8: e4m3 copies (K row-major [Spad][64] bytes; V^T fragments with the dim-tile pairs of a lane adjacent:
    // [Spad/32][2][64 lanes][16 B]) + one fp32 scale per (kv layer, stream, head) each; written by wm_enc_quant_cross_kv
    unsigned char *kx8 = nullptr, *vx8 = nullptr; float *kxs = nullptr, *vxs = nullptr;
    bool xkv8 = false;
    bf16_t *kc = nullptr, *vc = nullptr;   // self K/V  [nkv][maxB][H][Tal][64]
    int Benc = 0;                          // batch of the last wm_encode

    // ---- decode row scratch (<= WM_MAX_ROWS_SKINNY rows per chunk) ----
    float *h = nullptr, *hblk = nullptr, *hf = nullptr, *qbuf = nullptr;    // hf: [maxB*16][d] post-final-LN rows
    float *hb_keep = nullptr;                                               // Medusa-Block: carried block-layer output row per stream
    float *hf_cur = nullptr, *hf_keep = nullptr;                            // current chunk's rows; carried row per stream
    int* carry = nullptr;                                                   // [maxB] next base pass is redundant
    // merged-step schedule (several streams, chain candidates; wm_decoder.hip wm_dec_step): per step and stream, written by k_step_begin
    int4* rowinfo = nullptr;                                                // [Mmax * maxB] dense rows of the step's pass: {stream, index inside the stream, position of its row 0, kind 0 none / 1 base / 2 verify}
    int4* sinfo = nullptr;                                                  // [maxB] {first dense row, row count, position of row 0, mode}
    int* steprows = nullptr;                                                // [4] {dense rows of the pass, their 16-row tiles, 0, 0}
    bool step_flow = false;
    // candidate tree (medusa_choices with top-k > 1); tn == 0: the chain
    int tn = 0, tp = 0;
    TreeDev tree_host{};
    TreeDev* tree = nullptr;
    TreeDev* sibtree = nullptr;     // wm_config.sibling_rows: depth / ancestor tables of the chain + S leaves under the root (nodes K+1 .. K+S)
    int sib_cfg = 0;                // S the context was created for (0: off)
    float2* sibpart = nullptr;      // [maxB][SEL_SP slices][6] (value, token): the slice winners of head 1's row (k_select1 -> k_cand_fin)
    const unsigned long long* cur_anc = nullptr;                            // ancestor masks of the pass being enqueued (verify pass of a tree)
    int *sel_src = nullptr, *sel_n = nullptr, *sel_base = nullptr;          // K/V rows of the chosen path to move: [maxB*16], [maxB], [maxB]
    bool fuse = true;
    bool host_carry = false;                                                // single-stream runs: the host skips the base pass
    float* rs_table = nullptr; int rs_og = 0, rs_nw = 0, rs_width = 0;     // cached resampling filter bank [taps][phases]
    bool dev_carry = false;                                                 // several streams: per-stream carry flags on the device
    int *hostflags = nullptr, *hostflags_dev = nullptr;                     // host-mapped {carry, finished}
    hipGraphExec_t graph_base = nullptr;
    bf16_t *xbuf = nullptr, *fbuf = nullptr, *ybuf = nullptr;
    // LayerNorm fold (round 6): operand gamma o x of the next LayerNorm-fed GEMM (hi / lo planes, written by the launch that produced x), the
    // rows' statistics partials [d / 16 tiles][Rcap rows] and the fold vectors of every decoder layer; WM_LN_FOLD=0 keeps the LayerNorm launches
    bf16_t* xn = nullptr; float2* lnstats = nullptr; float* foldv = nullptr;
    bool ln_fold = false;
    float *cml = nullptr, *co = nullptr;   // cross-attention partials
    int* ticket = nullptr;                 // [16 streams][H] arrival tickets of the cross-attention key splits
    float* logits = nullptr;               // [32][Vpad]
    int* amax = nullptr; float *pc = nullptr;                   // select outputs, [maxB*16]
    float *part1 = nullptr, *part2 = nullptr;                   // select slice partials [16][SEL_SP][4] / [maxB*16][SEL_SP]

    // ---- decode state ----
    int *ids = nullptr, *L = nullptr, *kvlen = nullptr, *finished = nullptr, *cand = nullptr, *niter = nullptr;
    long long* hist = nullptr;             // [16] accept histogram + [16]=tokens emitted
    unsigned char* supmask = nullptr;      // [Vpad] bit0 suppress, bit1 begin-suppress
    float* exppen = nullptr;               // [Tids+1] (factor^(t-start) - 1) as float
    int* tap_tok = nullptr;
    int* done = nullptr;                   // [0] = all streams finished, [1] = number of finished streams
    bool use_done = false;
    GenDev gp{};
    int Bdec = 0;
    bool began = false, first_done = false;
    long long iters = 0;

    hipGraphExec_t graph = nullptr;
    int graph_B = 0;
    int graph_replays = 0;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool prefetch = true;           // in-launch next-operand prefetch blocks (WM_PREFETCH=0 turns them off)
    float ms_logmel = 0.f, ms_encode = 0.f, ms_decode = 0.f;
};

// implemented in wm_encoder.hip
long wm_enc_resample_len(long n_in, int sr_in, int sr_out);
int wm_enc_resample(wm_ctx* ctx, const float* in, int B, int channels, int n_in, int sr_in, int sr_out, float* out);
int wm_enc_logmel(wm_ctx* ctx, const float* wav, int B, int n_samples, float* feats);
int wm_enc_encode(wm_ctx* ctx, const float* feats, int B);
int wm_enc_set_output(wm_ctx* ctx, const float* hidden, int B);
int wm_enc_quant_cross_kv(wm_ctx* ctx, int B);      // cross_kv_fp8: the e4m3 copy of the projected cross-K/V of B streams
// implemented in wm_decoder.hip
int wm_dec_stage_layers(wm_ctx* ctx, int b0, int nb, int Mper, int mode /*0 base, 1 verify*/);
int wm_dec_stage_final(wm_ctx* ctx, int b0, int nb, int Mper, int mode, int medusa);
int wm_dec_stage_heads(wm_ctx* ctx, int nsel, int sel_mul, int sel_off, int medusa);
int wm_dec_pass(wm_ctx* ctx, int b0, int nb, int Mper, int mode, int medusa, int all_rows);
int wm_dec_iteration(wm_ctx* ctx, int Mper_base);   // one full iteration over all streams
int wm_dec_prompt_prefix(wm_ctx* ctx, int P);       // K/V of a long prompt's leading chunks; returns the tokens left (<= 16), < 0 on error
int wm_dec_iter_base(wm_ctx* ctx, int Mper_base);   // base pass layers + final LN
int wm_dec_iter_rest(wm_ctx* ctx, int Mper_base);
int wm_dec_step(wm_ctx* ctx, int);   // heads, candidates, verify pass, accept
int wm_dec_profile(wm_ctx* ctx, int kernel, int rows, int reps, float* ms, double* bytes);
int wm_dec_fold_init(wm_ctx* ctx);   // c = W gamma, b' = b + W beta of every LayerNorm-fed decoder GEMM (needs ctx->foldv)
