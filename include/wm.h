/* wm.h — C-ABI of libwm.so, the MI355X (gfx950) Whisper-Medusa inference engine.
 *
 * The reference (aiola-lab/whisper-medusa) is pure Python and has no FFI; its boundary for
 * this path is the Python API `WhisperMedusaModel.from_pretrained()/generate()/forward()`
 * (whisper_medusa/models/model.py:265-291, :1419-1779, :1223-1347).  The entry points below
 * are what a native back end for that API binds; every declaration cites the reference
 * interface it replaces.  The Python drop-in (whisper-medusa_amd/whisper_medusa/api.py)
 * calls them through ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: all functions return 0 on success, <0 on error (wm_last_error(ctx) gives the
 * text; WM_ERR_* below).  Handles are opaque.  Pointers marked DEV are device (HBM)
 * pointers, HOST are host pointers.  One context = one GPU + one HIP stream; a context is
 * not thread-safe, distinct contexts are independent (and may run concurrently from different host threads).
 * Ordering: the work runs on the context's own stream; device inputs must be complete before the call (the caller
 * synchronises whatever produced them) and every entry point returns after its work has finished, so outputs can be
 * read from any stream.  Nothing here takes a torch type.
 */
#ifndef WM_H_
#define WM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_ABI_VERSION 9

#define WM_OK 0
#define WM_ERR_ARG (-1)      /* bad argument / unsupported configuration (reference: ValueError, model.py:225-229) */
#define WM_ERR_HIP (-2)      /* HIP runtime failure */
#define WM_ERR_STATE (-3)    /* call sequence violated (e.g. decode before encode) */
#define WM_ERR_NOMEM (-4)

#define WM_HEADS_LINEAR 0    /* medusa_heads_type="base_head"   (model.py:235-246) */
#define WM_HEADS_BLOCK 1     /* medusa_heads_type="medusa_block" (model.py:248-256) */

#define WM_ACCEPT_GREEDY 0   /* temperature==0 branch, medusa_utils.py:547-560 */
#define WM_ACCEPT_TYPICAL 1  /* temperature!=0 branch, medusa_utils.py:562-588 (what generate() runs, model.py:1877-1881) */

typedef struct wm_ctx wm_ctx;

/* Mirrors MedusaConfig + the Whisper dims it inherits (utils/config_and_args.py:17-62). */
typedef struct wm_config {
    int32_t abi_version;        /* = WM_ABI_VERSION */
    int32_t d_model;            /* multiple of 64; head_dim is 64 */
    int32_t enc_layers, dec_layers;
    int32_t n_heads;            /* d_model / 64 (encoder == decoder) */
    int32_t ffn_dim;            /* encoder_ffn_dim == decoder_ffn_dim */
    int32_t vocab;
    int32_t n_mels;             /* 80 */
    int32_t n_ctx;              /* max_source_positions (1500): encoder frames per clip */
    int32_t n_tgt;              /* max_target_positions (448) */
    int32_t medusa_heads;       /* K = medusa_num_heads, <= 15 */
    int32_t heads_type;         /* WM_HEADS_* */
    int32_t max_batch;          /* streams the context is sized for */
    int32_t dec_weight_fp8;     /* 1: the six matrices of every decoder layer are fp8 e4m3 (OCP) in the packed layout, with one fp32
                                 * scale per output row appended to the table (BASELINE.json configs[4]); 0: bf16 */
    int32_t medusa_choices[16]; /* MedusaConfig.medusa_choices = [1, c_1, .., c_K] (utils/config_and_args.py:17-62; consumed by
                                 * generate_medusa_buffers / generate_candidates, medusa_utils.py:305-458): head k contributes its
                                 * top-c_k tokens, the candidate tree is their cartesian product.  All zero or all one = the chain
                                 * [1]*(K+1) every shipped checkpoint uses.  Limits: sum_i prod_{l<=i} c_l <= 64 nodes (the verify
                                 * pass of a stream is up to four 16-row query tiles), prod c_l <= 32 paths, c_k <= 4 — e.g. K = 10
                                 * with top-2 on the first two heads: [1,2,2,1,1,1,1,1,1,1,1] = 39 nodes.  Each node attends
                                 * to the history and to its own ancestors and sits at position L + depth — the mask / position
                                 * ids the reference builds (medusa_utils.py:343-363) and then never hands to its decoder. */
    int32_t enc_fp8;            /* 1: the encoder GEMMs fed by a LayerNorm (QKV, FC1) and the cross-K/V projection run on the CDNA4
                                 * fp8 MFMA v_mfma_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate; BASELINE.json configs[4]), operands in
                                 * the 128-k unit layout [R/16][Kp/128][2][64 lanes][16 B] (K zero-padded to Kp = 128 ceil(K / 128);
                                 * whisper_medusa/weights.py pack_matrix_fp8_k128): e4m3 weights with one fp32 scale per output row — 4 table
                                 * entries per encoder layer (qkv, qkv scales, fc1, fc1 scales) + 2 (cross-K/V) appended after the
                                 * decoder scales, the bf16 entries of those matrices may then be 16-byte placeholders — and the
                                 * LayerNorm output quantised to e4m3 with one scale per token row; 0: bf16 */
    int32_t act_fp16;           /* the decode numerics contract (ABI v8).  0: every decoder GEMM operand is a bf16 hi / lo pair
                                 * (~17 mantissa bits, two planes, two MFMAs per weight fragment), decoder matrices bf16.  1: ONE fp16 plane
                                 * (11 bits: the precision of the reference's own half-precision inference, model.py:1223-1347 under
                                 * torch.float16), one v_mfma_f32_16x16x32_f16 per weight fragment; the blob then holds the decoder-layer
                                 * matrices, the Medusa heads and the packed vocabulary projection as fp16 (exact from bf16 for |w| >= 2^-17;
                                 * whisper_medusa/weights.py build_blob(act_fp16=True)).  A library is BUILT for one contract
                                 * (wm_build_act_fp16(): libwm.so 0, libwm_f16.so 1); wm_create refuses the other. */
    int32_t cross_kv_fp8;       /* 1 (BASELINE.json configs[4]; ABI v8): the decode loop reads the encoder cross-K/V — the dominant HBM stream of a
                                 * batched step, 245.8 MB per stream and pass in bf16 — from an fp8 e4m3 copy with ONE fp32 scale per (kv layer,
                                 * stream, head) for K and one for V (scale = max|x| / 448 over the head's 1500 x 64 values), written by a
                                 * quantisation pass behind wm_encode's cross-K/V projection (HF WhisperAttention cross branch,
                                 * modeling_whisper.py:322-335); K's scale rides on the query, V's on the normalised output.  The bf16 projection
                                 * stays in HBM (wm_get_cross_kv returns it).  0: the bf16 cache. */
    int32_t sibling_rows;       /* ABI v9.  S > 0 (one stream, candidate chain): the verify pass carries, in the spare rows of its 16-row tile, head 1's
                                 * top-2 .. top-(S+1) tokens as LEAVES under the root (position L + 1, attending the history, the root and
                                 * themselves; S <= 15 - medusa_heads, <= 5).  Acceptance is evaluated on the chain exactly as without them
                                 * (medusa_utils.py:526-641: the emitted ids are the reference's); when the chain accepts nothing (a = 0) and
                                 * argmax v_0 — the next root, model.py:710-713 — is one of the siblings, that row's post-LN state IS the
                                 * state the next base pass would compute for it: its K/V rows move to position L + 1 and the base pass is
                                 * skipped (the hidden-state carry of an accept length > 0, extended to a = 0).  0: off. */
} wm_config;

/* Packed parameter blob (layout: whisper_medusa/weights.py).  The blob
 * stays owned by the caller and must outlive the context.  Replaces the state-dict load of
 * model.py:273-278. */
typedef struct wm_weights {
    const void* blob;           /* DEV */
    uint64_t blob_bytes;
    const uint64_t* offsets;    /* HOST: byte offset of every tensor, canonical order */
    int32_t n_offsets;
} wm_weights;

/* Per-call generation parameters: what generate() derives from generation_config
 * (model.py:1168-1207 processors, :1635-1639 lengths, :774-793 stop rules,
 * medusa_utils.py:14-18 posterior constants). */
typedef struct wm_gen_params {
    const int32_t* prompt;          /* HOST, decoder prompt ids (model.py:1519-1537) */
    int32_t prompt_len;             /* P = begin_index */
    int32_t eos_token_id, pad_token_id;
    const int32_t* suppress;        /* HOST, SuppressTokensLogitsProcessor list */
    int32_t n_suppress;
    const int32_t* begin_suppress;  /* HOST, SuppressTokensAtBeginLogitsProcessor list */
    int32_t n_begin_suppress;
    int32_t max_length;             /* MaxLengthCriteria: stop when L >= max_length */
    int32_t hard_max_length;        /* model.py:789-793: stop when L + K >= this */
    int32_t exp_decay_start;        /* ExponentialDecayLengthPenalty start (relative to P); <0 = off */
    float exp_decay_factor;
    float posterior_threshold, posterior_alpha;   /* 0.09 / 0.3 */
    float temperature;              /* divides verify logits in typical mode (1.0 via generate()) */
    int32_t accept_mode;            /* WM_ACCEPT_* */
    int32_t vanilla;                /* 1 = plain greedy decoding on the base head (anchor measurement) */
    int32_t begin_index;            /* sequence length at which the begin-suppress list applies (SuppressTokensAtBeginLogitsProcessor
                                     * begin_index).  < 0: prompt_len.  With `prompt_ids` the reference hands HF the number of init tokens
                                     * only (model.py:1537 `begin_index = init_tokens.shape[1]`, :1640-1644 set_begin_index), not the
                                     * length of the whole decoder prompt — the caller passes that number here */
    int32_t force_accept;           /* measurement knob (bench.py acceptance-sensitivity rows): >= 0 forces every iteration's accept
                                     * length to min(force_accept, K) whatever the posterior says — the tokens are then meaningless,
                                     * the cost of an iteration at that acceptance is not; < 0 = off (always, outside benchmarks) */
} wm_gen_params;

typedef struct wm_stats {
    int64_t iterations;             /* decode iterations the slowest stream needed since wm_decode_begin */
    int64_t iterations_launched;    /* iterations enqueued (polling granularity; extra ones are device-side no-ops) */
    int64_t tokens_emitted;         /* sum over streams of tokens appended after the prompt */
    int64_t accept_hist[16];        /* histogram of accept length a (0..K), all streams */
    float ms_logmel, ms_encode, ms_decode;   /* hipEvent-timed on the context's stream, last call of each */
    int32_t graph_replays;          /* decode iterations that ran as hipGraph replays */
    int32_t schedule_steps;         /* merged-step schedule (several streams, candidate chain): passes that carried rows since
                                     * wm_decode_begin (a stream's iteration takes one pass, two after an accept length of 0);
                                     * 0 = lock-step schedule (base pass + verify pass per iteration) */
    int32_t sibling_hits;           /* wm_config.sibling_rows: iterations with accept length 0 whose next root was a sibling row (base pass skipped) */
} wm_stats;

/* ---- lifecycle (replaces WhisperMedusaModel.from_pretrained / .to(device), model.py:265-291) ---- */
int wm_create(const wm_config* cfg, const wm_weights* w, int device, void* hip_stream /* hipStream_t or NULL */,
              wm_ctx** out);
void wm_destroy(wm_ctx* ctx);
const char* wm_last_error(const wm_ctx* ctx);     /* ctx may be NULL: last create error */
int wm_abi_version(void);
/* the decode numerics contract this library was compiled for (wm_config.act_fp16 must equal it) */
int wm_build_act_fp16(void);

/* ---- audio front door (replaces what the reference's callers do with torchaudio before the
 * feature extractor: `input_speech.mean(dim=0)` and `torchaudio.transforms.Resample(sr, 16000)`, README.md:120-125,
 * eval_whisper_medusa.py:41-45).  Channel-mean downmix + windowed-sinc polyphase resampling with torchaudio 2.2.2's
 * default filter (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99).
 * in: DEV float32 [B][channels][n_in]; out: DEV float32 [B][wm_resample_len(n_in, sr_in, sr_out)].
 * sr_in == sr_out: downmix only. */
int64_t wm_resample_len(int64_t n_in, int sr_in, int sr_out);    /* ceil(n_in * sr_out / sr_in) */
int wm_resample(wm_ctx* ctx, const float* in, int B, int channels, int n_in, int sr_in, int sr_out, float* out);

/* ---- F0 log-mel (replaces WhisperProcessor.__call__, eval_whisper_medusa.py:46-50) ----
 * wav: DEV float32 [B][n_samples] already padded/trimmed to n_samples = 320*n_ctx (480000).
 * feats: DEV float32 [B][n_mels][2*n_ctx]. */
int wm_logmel(wm_ctx* ctx, const float* wav, int B, int n_samples, float* feats);

/* ---- F1+F2 encoder and cross-KV projection (replaces the encoder call of
 * _prepare_encoder_decoder_kwargs_for_generation, model.py:1005-1011, and the cross K/V
 * projection hidden in the first decoder pass).  feats: DEV float32 [B][n_mels][2*n_ctx]. */
int wm_encode(wm_ctx* ctx, const float* feats, int B);
/* forward(encoder_outputs=...) (model.py:1223-1243, :1232: a caller-supplied `encoder_outputs[0]` replaces the encoder pass, HF
 * WhisperModel.forward): hidden = DEV float32 [B][n_ctx][d_model] last hidden state (after the encoder's final LayerNorm).  It is
 * stored bf16 like wm_encode's own output and the cross K/V of every decoder layer are projected from it; afterwards the context
 * is in the state wm_encode leaves.  Not available on an enc_fp8 context (WM_ERR_ARG). */
int wm_set_encoder_output(wm_ctx* ctx, const float* hidden, int B);

/* ---- F3..F14 the Medusa decode loop (replaces _medusa_greedy_search, model.py:404-835) ---- */
int wm_decode_begin(wm_ctx* ctx, const wm_gen_params* gp, int B);
/* Runs up to max_iters iterations (each = base pass + verify pass + accept), replayed from a
 * hipGraph after the first; returns the number of unfinished streams in *n_unfinished.
 * Several streams with candidate chains run the merged-step schedule: max_iters then counts STEPS (one pass each:
 * a stream whose last accept length was 0 spends one step on its base row and verifies in the next; the emitted
 * tokens, accept histogram and per-stream iteration counts are those of the lock-step schedule).
 * Environment: WM_NO_STEP=1 lock-step schedule, WM_NO_CARRY=1 no hidden-state carry, WM_NO_GRAPH=1 eager launches. */
int wm_decode_run(wm_ctx* ctx, int max_iters, int* n_unfinished);
/* ids (prompt + generated, post-EOS overwrite of model.py:798-810 applied) of one stream. */
int wm_get_tokens(wm_ctx* ctx, int stream, int32_t* out /* HOST */, int cap, int* n);
int wm_get_stats(wm_ctx* ctx, wm_stats* out /* HOST */);
int wm_sync(wm_ctx* ctx);

/* ---- parity taps (test-only views of intermediate state; no reference equivalent except
 * forward(), model.py:1223-1347) ---- */
/* encoder output [B][n_ctx][d_model] as float32 to HOST */
int wm_get_encoder_output(wm_ctx* ctx, int B, float* out /* HOST */);
/* One decoder pass for stream 0..B-1 over T (<=16) tokens each at positions pos0.., appending
 * K/V at kv row pos0; logits_out HOST float32 [n_out][B][T][vocab], n_out = 1 if disable_medusa
 * else K+1 (all T rows, as forward() returns).  Uses (and overwrites) the decode-loop state: call it
 * before wm_decode_begin, or begin again afterwards. */
int wm_forward_logits(wm_ctx* ctx, int B, const int32_t* tokens /* HOST [B][T] */, int T, int pos0,
                      int disable_medusa, float* logits_out);
/* cross K/V of one kv-layer/stream/head: HOST float32 [n_ctx][64] each */
int wm_get_cross_kv(wm_ctx* ctx, int kv_layer, int stream, int head, float* k_out, float* v_out);
/* Times `reps` launches of one decode-path kernel class in its current shape with hipEvents on
 * the context stream (bench.py roofline leg).  kernel: 0 = the weight-streaming GEMMs of one decoder
 * layer pass (all 6), 1..6 = one of them (LN1+QKV, out-proj, LN2+cross-q, cross-out, LN3+FC1+GELU, FC2),
 * 7 = the shared vocabulary projection; rows = token rows (<= 16 x max_batch).  Returns avg ms per rep
 * in *ms and the weight bytes those launches stream in *bytes.  No reference counterpart (measurement). */
int wm_profile_kernel(wm_ctx* ctx, int kernel, int rows, int reps, float* ms, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* WM_H_ */
