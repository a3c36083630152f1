/* C ABI of libkangaroo_hip.so: the MI355X (gfx950) decode path of Kangaroo
 * (double-early-exit self-speculative greedy decoding, batch size 1).
 *
 * The reference is pure Python on eager PyTorch and has no FFI of its own; each entry
 * point below names the reference code it stands in for.  All pointers marked "device"
 * are HIP device pointers owned by the caller (PyTorch allocates them); the library
 * never frees caller memory.  Every function returns 0 on success and a negative code
 * on failure (-1 bad argument, -2 HIP error, -3 does not fit on chip, -4 an opt-in persistent pass
 * gave up at a grid barrier and was switched off for the session: call again, -5 a tensor-parallel peer did not
 * deliver its partial sums in time);
 * kg_last_error() gives the text.  Nothing throws.  `stream` is a hipStream_t passed
 * as void* (torch.cuda.current_stream().cuda_stream).
 *
 * Weight layout ("packed"): a Linear weight W[N][K] (row-major, as HF stores it) is
 * padded to ceil16(N) rows and stored as 1 KiB tiles of 16 rows x 32 k; tile (rg, kt)
 * sits at element offset (rg * K/32 + kt) * 512, and inside a tile element (g, r, j) at
 * g*128 + r*8 + j is W[16*rg + r][32*kt + 8*g + j].  kangaroo_amd/packing.py produces it.
 * Row order before packing:
 *   wqkv   rows = [q heads | k heads | v heads]; inside every head the rows are
 *          interleaved (j, j + hd/2) -> (2j, 2j+1) so a RoPE pair is adjacent
 *   wgu    rows (2i, 2i+1) = (gate_proj row i, up_proj row i)
 *   wo, wdown, lm_head   natural order
 */
#ifndef KANGAROO_HIP_H
#define KANGAROO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define KG_DTYPE_F16 0
#define KG_DTYPE_BF16 1
#define KG_MAX_ROWS 16

typedef struct kg_session kg_session;

typedef struct {
    int dtype;                 /* KG_DTYPE_* : storage and MAC input type (fp32 accumulate) */
    int hidden_size, num_layers, num_heads, num_kv_heads, head_dim;
    int intermediate_size, vocab_size;
    float rms_eps;
    int max_positions;         /* rows of the RoPE tables */
    int exit_layer;            /* EARLY_STOP_LAYER: layers [0, exit) draft, [exit, N) verify; == num_layers: a draft-only
                                  session (pass-by-pass entry points only) whose remaining layers run elsewhere */
} kg_shape;

typedef struct {
    const void* wqkv;          /* packed [(H + 2 Hkv) hd, d] */
    const void* wo;            /* packed [d, H hd] */
    const void* wgu;           /* packed [2 F, d]; NULL for the adapter layer */
    const void* wdown;         /* packed [d, F];   NULL for the adapter layer */
    const void* norm1;         /* [d] input_layernorm */
    const void* norm2;         /* [d] post_attention_layernorm; NULL for the adapter */
} kg_layer_weights;

typedef struct {
    const void* embed;         /* [V, d] row-major */
    const void* final_norm;    /* [d] model.norm */
    const void* lm_head;       /* packed [V, d] (the reference keeps a separate copy as head_model) */
    const kg_layer_weights* layers;   /* host array of num_layers entries (copied) */
    kg_layer_weights adapter;  /* the adapter's single attention-only layer */
    const void* adapter_norm;  /* [d] adapter final norm */
    const float* rope_cos;     /* [max_positions, hd/2] fp32 */
    const float* rope_sin;
} kg_weights;

typedef struct {
    void* k_cache;             /* [num_layers][Hkv][t_max][hd] */
    void* v_cache;             /* same, or [num_layers][Hkv][hd][t_max] when head_dim >= 32 and t_max % 32 == 0 */
    void* ak_cache;            /* adapter: [Hkv][t_max][hd] */
    void* av_cache;
    void* workspace;           /* >= kg_workspace_bytes(), zero-filled by the caller */
    size_t workspace_bytes;
    int t_max;                 /* cache rows = max_length of the loop (2048 in the reference) */
} kg_buffers;

typedef struct {
    int n_tokens;              /* len(output_ids) = start_index + 1 */
    int new_tokens;            /* start_index - context_length + 1 */
    int rounds;                /* verify rounds (= idx of the reference's return) */
    int drafted_total;         /* adapter steps over all rounds */
    int stopped_on_eos;
    float prefill_ms;          /* device time of the prefill */
    float decode_ms;           /* device time of the draft/verify loop */
} kg_gen_result;

int kg_version(void);
const char* kg_last_error(void);

/* ---- kernel-level entry points (unit parity tests call these) -------------------- */

/* y = x W^T for n_rows <= 128 rows with the fused prologue/epilogue of the decode path.
 * Stands in for torch.nn.functional.linear as the reference's layers call it
 * (kangaroo/adapter.py:170-172,222,242; kangaroo/kangaroo_model.py:34).
 *   prologue 0: x used as is; 1: x = RMSNorm(x; norm_w, eps) (adapter.py:254-259)
 *   epilogue 0: out = y; 1: out = resid + y (adapter.py:304); 3: out = silu(y_gate) * y_up
 *            (adapter.py:242, rows interleaved as wgu); 5: out = y as unrounded fp32 (out is float*,
 *            out_stride in floats; <= 16 rows, no prologue): the partial sums of a row-split projection
 *            of a tensor-parallel layer, which the caller adds up across ranks
 * n_rows == -1 is ONE row computed by the MFMA kernel of a multi-row pass (bit-identical to that row
 * inside a verify pass; kg_qkv_rope_append takes it too).
 * n_rows == 1 runs the wave-reduced GEMV, 2..16 the MFMA decode kernel, 17..128 the prefill GEMM
 * kernel (the base model's prompt pass, evaluation/inference_kangaroo.py:30; nblocks / nwaves unused;
 * with prologue 1 the rows are normalised into the scratch buffer first, once per pass).
 * The RMSNorm keeps the reference's rounding points: x * rsqrt(mean x^2 + eps) is rounded to the
 * model dtype before the product with norm_w is rounded again (adapter.py:257-259). */
int kg_linear(int dtype, const void* w_packed, int N, int K, const void* x, int n_rows,
              int prologue, const void* norm_w, float eps,
              int epilogue, const void* resid, void* out, int out_stride,
              void* scratch /* device, >= kg_linear_scratch_bytes(n_rows, K) */, int nblocks, int nwaves, void* stream);
/* 256 B, plus room for the normalised rows of a pass of more than 16 rows */
size_t kg_linear_scratch_bytes(int n_rows, int K);

/* pos0 of kg_qkv_rope_append / kg_attention: the first position is not in the call but in device memory, the int32
 * at scratch + 128, read when the pass runs.  For callers that capture a pass into a graph once and replay it at
 * changing positions (they update that int between replays); a value outside [0, t_max - n_rows] turns the pass
 * into a no-op. */
#define KG_POS_IN_SCRATCH (-2)

/* Fused RMSNorm + QKV projection + RoPE + in-place KV append for n_rows (<= 128) rows at
 * positions pos0.. (adapter.py:170-189 with the cache growing by torch.cat there).
 * K cache [Hkv][t_max][hd]; V cache [Hkv][t_max][hd] (v_transposed 0) or [Hkv][hd][t_max]
 * (v_transposed 1: the layout the MFMA attention kernel reads; head_dim 32/64/128 and
 * t_max % 32 == 0). */
int kg_qkv_rope_append(int dtype, const void* wqkv_packed, int d, int n_heads, int n_kv_heads, int head_dim,
                       const void* x, int n_rows, int pos0, const void* norm_w, float eps,
                       const float* rope_cos, const float* rope_sin,
                       void* q_out, void* k_cache, void* v_cache, int t_max, int v_transposed,
                       void* scratch /* device, >= kg_linear_scratch_bytes(n_rows, d) */, int nblocks, int nwaves, void* stream);

/* Causal attention of n_rows query rows (<= 16; <= 128 with v_transposed) at positions pos0..
 * over the cache rows [0, pos0 + n_rows) (adapter.py:191-221; the mask of :38-67 is the causal bound). */
int kg_attention(int dtype, const void* q, int n_rows, int pos0, const void* k_cache, const void* v_cache,
                 int n_heads, int n_kv_heads, int head_dim, int t_max, int v_transposed, int nsplit,
                 void* out, void* scratch /* device, kg_attention_scratch_bytes() */, void* stream);
size_t kg_attention_scratch_bytes(int n_heads, int head_dim, int nsplit);

/* lm_head + per-row argmax + softmax max (inference_kangaroo.py:79-83, 91-92).
 * logits_out may be NULL.  prologue as kg_linear. */
int kg_head(int dtype, const void* w_packed, int vocab, int d, const void* x, int n_rows,
            int prologue, const void* norm_w, float eps,
            void* logits_out, int* argmax_out, float* pmax_out,
            void* scratch /* device, kg_head_scratch_bytes() */, int nblocks, int nwaves, void* stream);
size_t kg_head_scratch_bytes(int vocab);

/* Accept / reject of one verify round on its own (evaluation/inference_kangaroo.py:92-104 and the
 * cache-length rollback of :108-119; rules 1 and 2: evaluation/inference_kangaroo_MTBench.py:377-389 and
 * :356-374).  Runs the accept kernel of the decode loop (and its rank / nucleus kernels) on a loop state
 * laid out in `scratch`.  logits: device, [q, vocab] stored 16-bit verify logits of rows start_index ..
 * start_index+q-1 (q <= 15); drafted_host: q-1 tokens drafted for positions start_index+1 ..; rule 0 greedy,
 * 1 top-k, 2 top-p on softmax(logits / temperature), 3 "typical" (:293-314: a posterior test against
 * min(epsilon, exp(-entropy) * delta) at every row that ends a run of matches; moves start_index without
 * writing a token and rebinds the draft threshold, as the reference does); steps / max_infer: the loop bound
 * of :46 for out[5].  Token slots beyond the drafted ones hold eos_token_id, as the loop's buffer does (:20).
 * out_host[10] = accept length, token at the new start index, eos / stop flag, new start index,
 * adapter cache length after the rollback, next-round-runs flag, drafted count, rounds done, and the draft
 * loop's confidence threshold after the round as the two halves of a double (0.6 going in). */
int kg_accept(int dtype, const void* logits, int vocab, int q, const int32_t* drafted_host, int start_index,
              int eos_token_id, int rule, int top_k, float top_p, float temperature, float epsilon, float delta,
              int steps, int max_infer, int adapter_len, int32_t* out_host,
              void* scratch /* device, kg_control_scratch_bytes() */, void* stream);
/* Control of one draft step on its own (evaluation/inference_kangaroo.py:55-61 adapter rows, :79-83 token and
 * confidence, :74 the second early exit as it will apply to the next step).  logits: device, [1, vocab] draft
 * logits of this step, or NULL to stop after the adapter-row decision.  out_host[8] = adapter rows of the step
 * (2 = the missing-row patch), their first position, drafted token, float bits of the confidence, end_index
 * after the step, next-step-drafts flag, adapter cache length, rows of the coming verify pass. */
int kg_draft_step(int dtype, const void* logits, int vocab, int step, int steps, double threshold,
                  int start_index, int end_index, int adapter_len, int32_t* out_host,
                  void* scratch /* device, kg_control_scratch_bytes() */, void* stream);
size_t kg_control_scratch_bytes(void);

/* kg_head without a scratch argument, for callers that cannot allocate device memory (code under HIP graph
 * capture, serving loops and harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * per-block softmax records and the pass parameters live in a device array built into the library (1 MiB
 * + 256 B per GPU, held until the process ends), so vocab is limited to kg_head_noalloc_max_vocab() = 65536;
 * larger values are refused with -1.  Neither call allocates or frees.  Calls on one GPU share that array:
 * calls that overlap on one GPU (different streams or threads) race; calls on one stream do not.
 * kg_head_noalloc_init() looks the array's address up for the current device (done on first use otherwise). */
int kg_head_noalloc_init(void);
size_t kg_head_noalloc_max_vocab(void);
int kg_head_noalloc(int dtype, const void* w_packed, int vocab, int d, const void* x, int n_rows,
                    int prologue, const void* norm_w, float eps,
                    void* logits_out, int* argmax_out, float* pmax_out, int nblocks, int nwaves, void* stream);

/* ---- model-level entry points ---------------------------------------------------- */

size_t kg_workspace_bytes(const kg_shape* shape, int t_max);
int kg_session_create(const kg_shape* shape, const kg_weights* weights, const kg_buffers* buffers, kg_session** out);
int kg_session_destroy(kg_session* s);
/* kg_session_get: "v_transposed" (layout of buffers.v_cache / av_cache the session expects:
 * 1 when head_dim >= 32 and t_max % 32 == 0), "prefill_rows", or a tuning key; -1 if unknown. */
int kg_session_get(kg_session* s, const char* key);
/* tuning knobs: "attn_nsplit" (number of fixed key chunks: fixes the boundaries of the attention
 * reduction), "attn_blocks" (blocks that share one head's chunks) and "attn_teams" (1 or 2 four-wave teams per
 * block working on different chunks at once): geometry only, results do not change;
 * "attn_prefetch" (1: off, 1 + n: attention blocks whose chunks hold no keys yet read the first n/8 of every
 * o_proj row group into the L2 their XCD shares with the o_proj blocks that follow; default: the share that
 * comes to about 24 MB of o_proj, 7 at the 7B shape),
 * "attn_prefetch_min" (chunks in use from which they do so; default 6) and "attn_prefetch_delay" (1 + cycles / 64
 * they wait first; default 17): timing only, the values read are discarded;
 * "<mode>_waves" and "<mode>_blocks_per_cu" for all linear ops of a mode, or "<mode>.<op>.waves" /
 * "<mode>.<op>.blocks_per_cu" for one, with mode in {vec (1 row), mfma (2..16 rows)} and op in
 * {qkv, o, gateup, down, head}: with at most cap = CUs * blocks_per_cu blocks a block walks
 * k = ceil(G / cap) of the op's G 16-row groups, and the launch uses ceil(G / k) blocks so that every
 * block walks k or k - 1; "<mode>.<op>.blocks" sets the block count outright (blocks_per_cu resets it); "persist" (2: run the verify pass and the deep layers of plain
 * decoding as one persistent kernel with grid barriers, 1: one kernel per op, the default) and
 * "persist.<op>.waves"; "accept_top_k" (1: greedy acceptance, evaluation/inference_kangaroo.py:95-104;
 * k > 1: a drafted token is kept while it is among the k largest verify logits of its row,
 * evaluation/inference_kangaroo_MTBench.py:377-389).  Block counts never change results; wave counts set how K is split
 * inside a block and therefore the fp32 summation order.  Values must be >= 1.
 * kg_session_get reads the per-op keys and "persist" (0 / 1) back.
 * "vanilla_gemv" (2: plain decoding, kg_generate mode 1, runs every layer and the head through the 1-row
 * GEMV kernels — faster by a few per cent, but no longer the arithmetic of a verify row, so not the
 * comparator of the token-identity tests; 1: off, the default).
 * "bench_layers": kg_bench_op cycles through this many layers' weights (all by default; 1 measures
 * the kernels on cache-warm weights). */
int kg_session_set(kg_session* s, const char* key, int value);
/* Real-valued settings: "accept_top_p" (0: off; p in (0, 1]: top-p verification, a drafted token is
 * kept while it lies in the pooled nuclei of the round's verify rows,
 * evaluation/inference_kangaroo_MTBench.py:356-374; takes precedence over "accept_top_k") and
 * "temperature" (> 0; divides the verify logits before that rule's softmax, :358); "typical_epsilon" and
 * "typical_delta" (the posterior threshold min(epsilon, exp(-entropy) * delta) of the typical rule, :279-304,
 * which the integer key "accept_typical" = 2 switches on and 1 off; it takes precedence over the other two).
 * kg_session_getf returns the value, or -1 for an unknown key. */
int kg_session_setf(kg_session* s, const char* key, float value);
float kg_session_getf(kg_session* s, const char* key);

/* Device address of a named region of the workspace ("tokens": int32[t_max], "state":
 * int32 loop state, "hs_early": the exit layer's input rows kept by the last prefill
 * [t_max, d], "head_tok" int32[16], "head_pmax" float[16]); NULL for an unknown name. */
void* kg_session_ptr(kg_session* s, const char* name);

/* Copy token ids into the session's device token buffer at [pos0, pos0 + n); ids outside
 * [0, vocab_size) are refused with -1. */
int kg_set_tokens(kg_session* s, const int32_t* tokens_host, int pos0, int n, void* stream);

/* Layers [layer_begin, layer_end) of the base model on n_rows <= 16 rows at positions
 * pos0.., appending their K/V at cache slots pos0...  x_in == NULL embeds
 * tokens[pos0 + i] first (EarlyExitLlamaForCausalLM.forward_draft_or_large_model,
 * kangaroo/earlyexit.py:41-66, and the prefill call inference_kangaroo.py:30). */
int kg_forward_layers(kg_session* s, int layer_begin, int layer_end, const void* x_in, int pos0, int n_rows,
                      void* h_out, void* stream);
/* model.norm (earlyexit.py:69) */
int kg_final_norm(kg_session* s, const void* x, int n_rows, void* out, void* stream);
/* The prefill call of the loop (evaluation/inference_kangaroo.py:30-35: `model.base_model(input_ids, position_ids,
 * output_hidden_states=True)`) as ONE pass over the whole prompt, every projection a single matrix product: tokens [0, n_tokens)
 * (kg_set_tokens) through all layers with cache append; early_out [n_tokens][d] = hidden_states[exit_layer], final_out
 * [n_tokens][d] = hidden_states[-1] (after the final norm), logits_out [n_tokens][vocab] (16 bit; NULL: not computed).
 * Returns -3 when the session cannot run it (see kg_last_error); the caller then makes 16-row kg_forward_layers calls. */
int kg_prefill_rows(kg_session* s, int n_tokens, void* early_out, void* final_out, void* logits_out, void* stream);
/* AdapterModel.forward_early_stop (kangaroo/adapter.py:378-460) on a whole prompt's rows at positions 0 .. n_rows - 1 (the
 * adapter prefill of evaluation/inference_kangaroo.py:36-38), as one pass. */
int kg_adapter_forward_rows(kg_session* s, const void* x_in, int n_rows, void* h_out, void* stream);
/* AdapterModel.forward_early_stop (kangaroo/adapter.py:378-460) on n_rows rows at pos0.. */
int kg_adapter_forward(kg_session* s, const void* x_in, int pos0, int n_rows, void* h_out, void* stream);
/* head_model (kangaroo/kangaroo_model.py:34-44) on already-normalised rows; n_rows == -1: one row in a
 * multi-row pass's arithmetic, as for kg_linear */
int kg_head_logits(kg_session* s, const void* x, int n_rows, void* logits_out, int* argmax_out, float* pmax_out, void* stream);

/* The whole kangaroo_forward (evaluation/inference_kangaroo.py:15-129): prefill, then
 * draft / verify rounds replayed from a captured graph; the early-exit test, accept /
 * reject and cache rollback run on the device.  mode 0 = Kangaroo, 1 = plain greedy
 * decoding on the same kernels (the comparator, inference_baseline.py:14-25).
 * out_tokens_host receives n_tokens ids; accept_log_host (may be NULL) receives per
 * round (accept_length, drafted) pairs.  The work runs on a stream the session owns, ordered after
 * everything queued on `stream` at the time of the call; the call returns after that work has finished.
 * Token ids outside [0, vocab_size) are refused with -1. */
int kg_generate(kg_session* s, int mode, const int32_t* prompt_host, int context_length, int max_new_tokens,
                int max_length, int steps, double threshold, int eos_token_id,
                int32_t* out_tokens_host, int32_t* accept_log_host, int accept_log_capacity,
                kg_gen_result* result, void* stream);

/* ---- tensor-parallel remaining layers (the >= 33B-class configuration) --------------
 * The reference only places whole layers on GPUs one after the other (device_map="auto",
 * kangaroo/kangaroo_model.py:22).  Here the layers [exit_layer, N) of a session can be SHARDS: q/k/v and gate/up are
 * split by output rows (whole heads, slices of the MLP width), o_proj and down_proj by input columns, so that every
 * rank produces fp32 partial sums [rows, d] which are added up in rank order on every rank (identical bits
 * everywhere); layers [0, exit_layer), adapter and head stay replicated.  After kg_session_set_tp every entry point
 * of the session (kg_generate's prefill and captured rounds, kg_forward_layers) runs the remaining layers that way.
 *
 * The exchange is one kernel per row-split projection: each rank copies its partial into its slot of every peer's
 * exchange buffer over the direct links, raises a flag there, waits for the peers' flags in its own buffer and sums
 * the slots (+ residual): one hop, no library collective, captured in the round graph like any other kernel.  The
 * buffer lives in uncached device memory that peers map through an inter-process handle (kg_tp_xbuf_*); ranks must
 * run the same passes in the same order.  A wait that outlasts about ten seconds raises an abort flag, after which no
 * wait blocks and kg_generate / kg_forward_layers return -5.
 * n_local == tp_size rehearses every rank on ONE GPU, shard after shard (no peers, no flags): bit-identical to a
 * job with one rank per GPU, because every partial is produced by the same kernel on the same data either way. */
#define KG_TP_MAX 8
typedef struct {
    int tp_size;                              /* ranks that share the remaining layers (2..8) */
    int n_local;                              /* shards this session runs: 1 (one rank per GPU) or tp_size (rehearsal) */
    int local_rank[KG_TP_MAX];                /* rank of each local shard */
    int heads_local, kv_heads_local;          /* heads per shard (a head count tp_size does not divide is padded with zero heads) */
    int ffn_local;                            /* intermediate_size / tp_size, a multiple of 32 */
    const kg_layer_weights* layers[KG_TP_MAX];/* per local shard: host array of num_layers - exit_layer entries (copied);
                                                 wqkv packed [(heads_local + 2 kv_heads_local) hd, d], wo [d, heads_local hd],
                                                 wgu [2 ffn_local, d], wdown [d, ffn_local], norms replicated */
    void* k_cache[KG_TP_MAX];                 /* per local shard: [num_layers - exit_layer][kv_heads_local][t_max][hd] */
    void* v_cache[KG_TP_MAX];                 /* same, transposed as the session's v_cache is */
    void* xbuf;                               /* this GPU's exchange buffer (kg_tp_xbuf_create) */
    void* peer_xbuf[KG_TP_MAX];               /* by rank: that rank's exchange buffer as mapped into this process
                                                 (kg_tp_xbuf_open); ignored for local ranks */
} kg_tp_config;
size_t kg_tp_xbuf_bytes(int tp_size, int hidden_size);
/* Zero-filled uncached device memory on the current GPU + its 64-byte inter-process handle (may be NULL). */
int kg_tp_xbuf_create(size_t bytes, void** out, void* ipc_handle_64);
int kg_tp_xbuf_open(const void* ipc_handle_64, void** out);     /* a peer's buffer, mapped into this process */
int kg_tp_xbuf_close(void* mapped);
int kg_tp_xbuf_destroy(void* p);
/* weights->layers[l] for l >= exit_layer and buffers->k_cache / v_cache beyond exit_layer layers are not used by a
 * session in this mode (they may be NULL / absent). */
int kg_session_set_tp(kg_session* s, const kg_tp_config* cfg);

/* Measurement aid (no counterpart in the reference): replays `iters` launches of one of
 * the session's linear kernels, cycling over the layers' weights so that no launch finds
 * its weights in the Infinity Cache, from a captured graph on the session's own stream,
 * and returns the mean microseconds per launch measured with HIP events on that stream.
 * op: 0 qkv(+RMSNorm, RoPE, KV append) 1 o_proj(+residual) 2 gate/up(+RMSNorm, SiLU*mul)
 * 3 down_proj(+residual) 4 lm_head(+RMSNorm, argmax partials).  rows == 1 times the GEMV,
 * rows > 1 the MFMA kernel as the verify pass launches it.  Overwrites KV slots
 * [0, rows) and the activation workspace: call it between generations only. */
int kg_bench_op(kg_session* s, int op, int rows, int iters, float* us_per_launch, size_t* weight_bytes_per_launch);

/* ---- dense matrix products of the compute-bound passes -------------------------------
 * C[M][N] = alpha * A[M][K] . B[N][K]^T with fp32 accumulation on the matrix cores: torch.nn.functional.linear as the
 * reference's layers and training loop call it on many rows at once (kangaroo/adapter.py:170-172,222 on a padded batch,
 * the frozen head of train.py:265-273, and every product of their backward pass after a transpose of the operand whose
 * contraction index is not contiguous).  A, B: 16-bit (dtype), row strides lda / ldb in elements, multiples of 8, K a
 * multiple of 8; C: 16-bit, or fp32 when out_f32.  batch > 1: independent products at element strides sA / sB / sC
 * (attention heads).  b_packed 1: B is a decode-path weight in the packed tile layout (K % 32 == 0, batch 1);
 * b_packed 2: B is given K-major, [K][N] with row stride ldb (C = alpha * A . B; N % 8 == 0): a product that contracts
 * over the rows of a stored matrix needs no transposed copy of it.
 * causal 1: tiles of C entirely above the diagonal are neither computed nor written (scores of a causal attention);
 * causal 2: A is lower triangular, so K is walked only up to each tile's last row (probabilities times values);
 * causal 3: A is upper triangular, K is walked from each tile's first row on (transposed probabilities);
 * causal 4 (K-major B only): row k of B has no entry beyond column k, K is walked from each tile's first column on
 * (dV^T = dO^T . P over the queries at or after each key).
 * kg_matmul_nt_heads: the batch index z = zo * batch_inner + zi addresses A at zo * sA + zi * sA2, C likewise and B at
 * zo * sB + (zi / b_inner_div) * sB2: the heads of [batch, seq, heads, head_dim] tensors read in place, b_inner_div
 * query heads sharing a key / value head (kangaroo/adapter.py:69-78). */
int kg_matmul_nt(int dtype, const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                 int batch, long long sA, long long sB, long long sC, int b_packed, int out_f32, int causal, float alpha, void* stream);
int kg_matmul_nt_heads(int dtype, const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                       int batch, long long sA, long long sB, long long sC, int batch_inner, long long sA2, long long sB2, long long sC2, int b_inner_div,
                       int b_packed, int out_f32, int causal, float alpha, void* stream);
/* The same product with K cut into n_ranges ranges of whole 64-steps that run side by side: range r leaves its fp32 partial
 * product at parts + r * range_stride (elements; >= M * ldc).  For products with few output tiles (the d-column projections
 * of a short prompt, evaluation/inference_kangaroo.py:30 through kangaroo/adapter.py:222,242): the caller adds the ranges in
 * range order, which keeps the result independent of how the ranges were scheduled.  b_packed 0 or 1. */
int kg_matmul_nt_ranges(int dtype, const void* A, const void* B, float* parts, int M, int N, int K, long long lda, long long ldb, long long ldc,
                        int b_packed, int n_ranges, long long range_stride, void* stream);
/* out[b][c][r] = in[b][r][c] for 16-bit elements (row strides ld_in / ld_out, batch strides s_in / s_out, in elements). */
int kg_transpose16(const void* in, void* out, int rows, int cols, long long ld_in, long long ld_out, int batch, long long s_in, long long s_out, void* stream);

/* ---- adapter training step (train.py:258-296): the fused, memory-bound pieces ------
 * The step's matrix products are plain GEMMs over batch x sequence rows and stay with the BLAS library
 * (kangaroo_amd/train.py); these kernels do everything between them.  Activations are 16-bit (dtype),
 * statistics / parameters / optimizer state fp32.  All pointers are device pointers. */

/* LlamaRMSNorm forward (kangaroo/adapter.py:254-259) on rows x d: y = round16(gamma * x * rstd); rstd[rows] is kept
 * for the backward pass. */
int kg_train_rmsnorm_fwd(int dtype, const void* x, const float* gamma, float eps, int rows, int d, void* y, float* rstd, void* stream);
/* Its backward: dx (may be NULL) and dgamma[d] = sum over rows; scratch of kg_train_rmsnorm_bwd_scratch(rows, d) bytes
 * holds per-block shares that are added in a fixed order.  d <= 8192. */
size_t kg_train_rmsnorm_bwd_scratch(int rows, int d);
int kg_train_rmsnorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* rstd, int rows, int d,
                         void* dx, float* dgamma, void* scratch, void* stream);
/* apply_rotary_pos_emb (kangaroo/adapter.py:87-95) in place on [rows][n_heads][head_dim], position = row % S,
 * tables [S][head_dim/2] fp32; inverse != 0 applies the adjoint (backward pass). */
int kg_train_rope(int dtype, void* x, long long rows, int S, int n_heads, int head_dim, const float* cos_t, const float* sin_t, int inverse, void* stream);
/* Attention probabilities in place over the scores [B][H][S][S] (kangaroo/adapter.py:195-211): softmax(scale * s) over
 * the keys j <= i with key_mask[b][j] != 0 (causal + padding masks of :38-67,327-348), zeros for the other keys up to the
 * end of the 256-column block that holds the diagonal; the columns beyond that block are NOT written (the causal products
 * of kg_matmul_nt_heads never read them).  S <= 4096. */
int kg_train_softmax_fwd(int dtype, void* scores, const int* key_mask, int B, int H, int S, float scale, void* stream);
/* dscores = scale * p * (dp - sum(dp * p)) written over dprobs, over the same columns as kg_train_softmax_fwd writes. */
int kg_train_softmax_bwd(int dtype, const void* probs, void* dprobs, int B, int H, int S, float scale, void* stream);
/* train.py:265-279 for rows of out-logits and target-logits [rows][vocab]: row_loss = -sum_v softmax(t)_v log_softmax(o)_v,
 * row_acc = sum_v min(softmax(t)_v, softmax(o)_v), and the gradient grad_scale * (softmax(o) - softmax(t)) written over
 * logits_out; rows with loss_mask 0 get zeros.  row_rank (may be NULL): how many out-logits lie above the out-logit of the
 * target's argmax token, i.e. the target is among the top k predictions iff rank < k (train.py:298-305); -1 outside the mask.
 * row_stride >= vocab: elements between rows (the products that make the logits want rows of whole 16-byte pieces; columns
 * beyond vocab are left alone). */
int kg_train_softce(int dtype, void* logits_out, const void* logits_tgt, const int* loss_mask, int rows, int vocab, long long row_stride, float grad_scale,
                    float* row_loss, float* row_acc, int* row_rank, void* stream);
/* y = y + x on n 16-bit elements (n a multiple of 8, 16-byte aligned), rounded once: the residual add of
 * kangaroo/adapter.py:304 after the o_proj product. */
int kg_train_add(int dtype, void* y, const void* x, long long n, void* stream);
/* Raises *flag to 1 when any of the n fp32 values is an infinity or a NaN, and leaves it alone otherwise (several tensors
 * share one flag; the caller clears it).  The flag is what kg_train_adamw's `skip` reads: torch.cuda.amp.GradScaler drops a
 * step whose gradients overflowed, which accelerate does for the reference's fp16 runs (train.py:46-47,285-287). */
int kg_train_nonfinite(const float* g, long long n, int* flag, void* stream);
/* accelerator.clip_grad_value_(clip) + torch.optim.AdamW step `step` (1-based) on n fp32 parameters (train.py:227,285-286);
 * grad_dtype 0 fp16, 1 bf16, 2 fp32, multiplied by inv_scale first; p16 (may be NULL) receives the 16-bit copy
 * (copy_dtype) that the GEMMs read.  skip (device, may be NULL): a non-zero word makes the call a no-op, read when the kernel
 * runs: the caller raises it when any gradient of the step is not finite, which is how the reference's GradScaler drops a step
 * whose fp16 gradients overflowed (accelerate's optimizer wrapper around train.py:286). */
int kg_train_adamw(float* p, float* m, float* v, const void* grad, int grad_dtype, float inv_scale, float clip, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, long long n, void* p16, int copy_dtype, const int* skip, void* stream);

#ifdef __cplusplus
}
#endif
#endif
