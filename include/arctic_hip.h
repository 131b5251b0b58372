/*
 * arctic_hip.h — C ABI of libarctic_hip.so, the MI355X (gfx950) native library behind the
 * arctic_inference plugin surface for the speculative-decode + Ulysses hot path.
 *
 * One shared library replaces both native extensions of the reference:
 *   csrc/suffix_cache  (pybind11 module arctic_inference.common.suffix_cache._C)
 *   csrc/custom_ops    (TORCH_LIBRARY op arctic_inference::reshape_and_cache_flash_bulk)
 * and adds the kernels the reference delegates to vLLM / PyTorch on its hot path (draft model,
 * rejection acceptance, verify attention, Ulysses pack/unpack).
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch types.  Device pointers are `void*` / typed
 *     pointers into HBM, host pointers are marked "host".
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - every function returns an int status: AIC_OK (0) or a negative AIC_ERR_*; the text of the
 *     last error on the calling thread is available from aic_last_error().
 *   - nothing here allocates or synchronises inside a launch function unless its comment says so
 *     (safe to capture in a hipGraph where noted).
 *   - there is NO CPU fallback: compute entry points fail with AIC_ERR_NO_DEVICE without a GPU.
 *
 * Reference interfaces replaced are cited as file:line under /root/reference.
 */
#ifndef ARCTIC_HIP_H_
#define ARCTIC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIC_OK 0
#define AIC_ERR_INVALID (-1)    /* bad argument / shape (reference: TORCH_CHECK -> RuntimeError) */
#define AIC_ERR_NO_DEVICE (-2)  /* no HIP device visible */
#define AIC_ERR_HIP (-3)        /* a HIP runtime call failed; see aic_last_error() */
#define AIC_ERR_NOT_FOUND (-4)  /* unknown request / handle */
#define AIC_ERR_EXISTS (-5)     /* duplicate request */
#define AIC_ERR_UNSUPPORTED (-6)
#define AIC_ERR_BUFFER_TOO_SMALL (-7) /* a caller-provided buffer is too small; the needed sizes are in the outputs */

/* element types of activation / cache buffers */
#define AIC_DT_F32 0
#define AIC_DT_F16 1
#define AIC_DT_BF16 2
#define AIC_DT_FP8_E4M3 3 /* OCP e4m3fn (gfx950 native), not fnuz */
#define AIC_DT_FP8_E5M2 4

const char* aic_last_error(void);
int aic_version(void);
/* number of visible HIP devices (0 on a CPU-only box); never initialises a context */
int aic_device_count(void);
/* Device timing of the dominant kernel (verify_attn_kernel) for bench.py's roofline figure: while
 * enabled (on = n >= 1) every n-th launch of that kernel is bracketed by a HIP event pair on its own
 * stream; aic_profile_read synchronises them, returns {sum of microseconds, timed launches} and resets. */
int aic_profile_enable(int on);
int aic_profile_read(double* total_us, int* launches);
/* The instrument's own reading: `pairs` (1..4096) event pairs recorded on `stream` with nothing between the two events
 * of a pair; mean (and, if asked, minimum) elapsed microseconds.  Synchronises the stream's events. */
int aic_profile_event_overhead(void* stream, int pairs, double* mean_us, double* min_us);

/* ------------------------------------------------------------------------------------------
 * A1/A2  Suffix tree — replaces pybind module `_C` (csrc/suffix_cache/pybind.cc:24-38,
 *        suffix_tree.h:63-110, suffix_tree.cc:31-274).
 *
 * Host side: the online update (append/extend) is an inherently serial pointer walk per token and
 * stays in C++ on the host, exactly where the reference runs it, but in an index arena that is
 * mirrored incrementally into HBM (nodes, a (parent,token)->child hash, the token store).
 * Device side: candidate matching (speculate) is a HIP kernel, one wavefront per
 * (query, tree, suffix start).
 * ---------------------------------------------------------------------------------------- */
typedef struct aic_suffix_tree aic_suffix_tree;

aic_suffix_tree* aic_st_create(int max_depth);                 /* SuffixTree(int) pybind.cc:33 */
void aic_st_destroy(aic_suffix_tree* t);
int aic_st_num_seqs(const aic_suffix_tree* t);                 /* num_seqs   pybind.cc:34 */
int aic_st_append(aic_suffix_tree* t, int seq_id, int token);  /* append     pybind.cc:35 */
int aic_st_extend(aic_suffix_tree* t, int seq_id, const int32_t* tokens /*host*/, int n); /* :36 */

/* speculate (pybind.cc:37, suffix_tree.cc:135-165).  Runs the HIP matcher on `stream` and synchronises it
 * before returning: path mode (use_tree_spec == 0: _speculate_path) and, since r04, tree mode (_speculate_tree, the
 * priority-queue expansion: simulator-only in the reference, never used in serving, model_runner.py:734-740) — see
 * aic_sc_speculate_batch_tree; a tree-mode query the device gives back, and tree mode without a device, are evaluated by the
 * host tree.
 * Outputs are host arrays of capacity `cap`; returns the number of tokens written (>= 0) or an
 * error (< 0). */
int aic_st_speculate(aic_suffix_tree* t, const int32_t* pattern /*host*/, int n, int max_spec_tokens,
                     float max_spec_factor, float max_spec_offset, float min_token_prob,
                     int use_tree_spec, int32_t* out_tokens, int32_t* out_parents, float* out_probs,
                     int cap, float* out_score, int32_t* out_match_len, void* stream);

/* Debug / test export of the flattened device image from the host mirror (no GPU needed).
 * Fills counts; when the arrays are non-NULL copies them (caller sizes them from a first call).
 * nodes: int32[n_nodes][8] = {count,parent,seq_id,start,length,best_child,alive,0}
 * hash : int32[n_slots][4] = {parent,token,child,state(0 empty,1 full,2 tombstone)}
 * tokens: int32[n_tokens]; seq_base: int32[n_seq_slots] offset of each sequence in `tokens`
 * seq_ids: int32[n_seq_slots] the caller-visible seq id of each slot */
int aic_st_export(aic_suffix_tree* t, int32_t* n_nodes, int32_t* n_slots, int32_t* n_tokens,
                  int32_t* n_seq_slots, int32_t* nodes, int32_t* hash, int32_t* tokens,
                  int32_t* seq_base, int32_t* seq_ids);
/* Verifies the incrementally maintained best_child of every node against a full scan in container
 * order (the reference's tie rule, suffix_tree.cc:208-214). Returns the number of mismatches. */
int aic_st_selfcheck(aic_suffix_tree* t);

/* ------------------------------------------------------------------------------------------
 * A3  SuffixCache core — native side of arctic_inference.common.suffix_cache.SuffixCache
 *     (common/suffix_cache/suffix_cache.py:57-222): one global tree of responses + one prompt
 *     tree per live request, batched device speculation for a whole engine step.
 *     Requests are named by an int64 key chosen by the Python layer.
 * ---------------------------------------------------------------------------------------- */
typedef struct aic_suffix_cache aic_suffix_cache;

aic_suffix_cache* aic_sc_create(int max_depth);
void aic_sc_destroy(aic_suffix_cache* c);
int aic_sc_has_prompt(const aic_suffix_cache* c, int64_t req);
/* cache_prompt suffix_cache.py:75-96 (AIC_ERR_EXISTS -> ValueError) */
int aic_sc_cache_prompt(aic_suffix_cache* c, int64_t req, const int32_t* tokens /*host*/, int n);
/* builds several prompt trees concurrently on host threads (independent trees) */
int aic_sc_cache_prompts(aic_suffix_cache* c, int n_req, const int64_t* reqs, const int32_t* tokens /*host, concatenated*/,
                         const int32_t* lens, int n_threads);
/* cache_prompt (suffix_cache.py:75-96) followed by update_response(req, response) (:118-149) for a request whose
 * prompt tree is not needed yet (vLLM knows a request's prompt when its prefill is scheduled; the reference builds the
 * tree on the engine thread at the first sampled token, model_runner.py:664-671, 3-4 ms per 4096-token prompt): the
 * response tokens enter the global tree now, in call order, and the prompt tree (prompt + response) is built on a host
 * thread.  Every later call that reads or extends that prompt tree joins the build first, so results are those of the
 * synchronous pair of calls.  AIC_ERR_EXISTS as aic_sc_cache_prompt. */
int aic_sc_cache_prompt_async(aic_suffix_cache* c, int64_t req, const int32_t* tokens /*host*/, int n,
                              const int32_t* response /*host, may be NULL*/, int n_response);
/* evict_prompt suffix_cache.py:98-111 (AIC_ERR_NOT_FOUND -> ValueError) */
int aic_sc_evict_prompt(aic_suffix_cache* c, int64_t req);
/* update_response suffix_cache.py:118-149; seq ids are dense in first-seen order (:113-116) */
int aic_sc_update_response(aic_suffix_cache* c, int64_t req, const int32_t* tokens /*host*/, int n);
/* the per-request loop of _update_suffix_cache (model_runner.py:657-678) in one call: request r appends
 * lens[r] tokens of the concatenated host array, in list order (identical to n_req update_response calls) */
int aic_sc_update_responses(aic_suffix_cache* c, int n_req, const int64_t* reqs, const int32_t* tokens /*host, concatenated*/,
                            const int32_t* lens);

/* read-only cache warm-up for the requests' trees (what the next update_responses touches first); call it while waiting for
 * the GPU: between two engine steps the trees fall out of the CPU cache and the update is a chain of dependent misses */
int aic_sc_warm(aic_suffix_cache* c, int n_req, const int64_t* reqs /*host*/);

/* speculate for a batch of requests (suffix_cache.py:151-222 applied per request; the call
 * pattern of model_runner.py:680-744).  All arrays are host arrays of length n_query unless
 * noted.  patterns: concatenated int32, pattern_lens[i] tokens each (only the last max_depth of
 * each are used, :197-198).  use_prompt[i] != 0 -> also search the request's prompt tree
 * (AIC_ERR_NOT_FOUND if it has none).  Outputs: out_tokens/out_probs [n_query][cap] row-major,
 * out_n / out_match_len / out_score [n_query].  Pending tree updates are mirrored to HBM, the
 * matcher runs on `stream`, results are copied back and the stream is synchronised. */
int aic_sc_speculate_batch(aic_suffix_cache* c, int n_query, const int64_t* reqs, const int32_t* patterns,
                           const int32_t* pattern_lens, const int32_t* max_spec_tokens,
                           const float* max_spec_factor, const float* max_spec_offset,
                           const float* min_token_prob, const int32_t* use_prompt, int cap,
                           int32_t* out_tokens, float* out_probs, int32_t* out_n, float* out_score,
                           int32_t* out_match_len, void* stream);
/* The same for tree-mode speculation (use_tree_spec = True: SuffixTree::_speculate_tree, suffix_tree.cc:226-274) ON THE DEVICE:
 * one wavefront per (query, tree, suffix start) grows the candidate with a priority queue that reproduces libstdc++'s
 * std::priority_queue (push_heap / pop_heap order among equal probabilities) over child lists mirrored in the host
 * container's iteration order (switched on per tree by its first tree-mode query).  out_parents [n_query][cap].
 * out_n[i] = -1: query i was given back (a node with more than 15 children, or a queue beyond 256 entries) — evaluate it on
 * the host trees (aic_st_speculate with use_tree_spec on aic_sc_prompt_tree / aic_sc_global_tree after
 * aic_debug_tree_mode_on_host(1)). */
int aic_sc_speculate_batch_tree(aic_suffix_cache* c, int n_query, const int64_t* reqs, const int32_t* patterns,
                                const int32_t* pattern_lens, const int32_t* max_spec_tokens,
                                const float* max_spec_factor, const float* max_spec_offset,
                                const float* min_token_prob, const int32_t* use_prompt, int cap,
                                int32_t* out_tokens, int32_t* out_parents, float* out_probs, int32_t* out_n,
                                float* out_score, int32_t* out_match_len, void* stream);
/* tree-mode bookkeeping: queries answered by the device / given back to the host so far; _on_host(1) sends every tree-mode
 * aic_st_speculate to the host trees (the A/B reference and the fallback's entry) */
int aic_debug_tree_mode_stats(int64_t* on_device, int64_t* given_back);
int aic_debug_tree_mode_on_host(int on);
/* device time of the last aic_sc_speculate_batch matcher launch pair in microseconds (HIP events
 * on the caller's stream), and the number of bytes mirrored host->device for it */
int aic_sc_last_stats(const aic_suffix_cache* c, float* match_us, int64_t* mirrored_bytes, int64_t* n_nodes_total);
/* wall time of the last aic_sc_speculate_batch: host-side delta / query collection, then staging copy .. stream sync */
int aic_sc_last_timing(const aic_suffix_cache* c, float* build_us, float* device_us);
aic_suffix_tree* aic_sc_global_tree(aic_suffix_cache* c);
aic_suffix_tree* aic_sc_prompt_tree(aic_suffix_cache* c, int64_t req);

/* debug aid (tools/microbench.py trace): while `buf` is set, every launch of the one-grid short + long attention kernel
 * records per workgroup {start, end (100 MHz ticks), HW_ID | XCC_ID << 32, kind (0 short, 1 long, 2 pad)} into
 * buf[workgroup][4] (device int64, at least capacity_wgs rows).  NULL switches it off. */
int aic_debug_attn_trace(int64_t* buf, int capacity_wgs);
/* debug aid (tools/microbench.py phases): while `buf` is set, short-only launches of host-partitioned calls record per
 * workgroup eight 100 MHz timestamps at the short body's phase boundaries into buf[workgroup][8] (device int64) */
int aic_debug_attn_phase_trace(int64_t* buf, int capacity_wgs);
/* debug aid: force the short attention body's kv heads per workgroup (4 / 2 / 1) and / or its cross-workgroup split count
 * for host-partitioned calls (0 = chosen by the library); every setting computes the same result. */
int aic_debug_attn_layout(int heads_per_wg, int splits);
/* debug aid: a call with short requests AND long drafts goes out as one grid, or — when the grid has room for a single split of
 * the long part only — as two launches on the same stream (long part, then short part).  1 / 0 = always / never two launches,
 * -1 = chosen by the library; every setting computes the same result. */
int aic_debug_attn_sequential(int mode);
/* debug aid: share (percent) of a full token-range split that a short workgroup takes when it will share its CU with a
 * long-draft workgroup of the one-grid launch (0 = the built-in 88, 100 = equal splits); every setting computes the same
 * result. */
int aic_debug_attn_light(int pct);
/* debug aid: upper bound on the cross-workgroup split count of the long-draft part of a mixed call (0 = chosen by the
 * library); every setting computes the same result. */
int aic_debug_attn_long_splits(int splits);
/* debug: which waves of a long-draft workgroup issue the tile DMA when the row tiles do not divide by four (0 = every wave its
 * quarter, 1 = the waves with one row tile more issue nothing, 2 = they keep their K pieces, 3 = wave 0 keeps one piece). */
int aic_debug_attn_long_dma(int pattern);
/* aic_verify_attention_layers sends a run of >= 4 layers out as HIP graph launches (an instantiated graph per sequence of
 * kernels, its nodes' parameters rewritten per call; 12 layers or more: the first 4 as one graph, the rest as a second)
 * unless the stream is being captured by the caller.  0 switches that off (kernel-by-kernel launches), 1 (default) on,
 * 2 = always a single graph; every setting computes the same result.  While aic_profile_enable() is on, every 5th call is
 * launched kernel by kernel so that its launches can carry event pairs.  _stats: graph launches so far and distinct
 * graphs instantiated. */
int aic_debug_attn_graph(int on);
int aic_debug_attn_graph_stats(uint64_t* launches, uint64_t* builds);

/* ------------------------------------------------------------------------------------------
 * (f)-1  SwiftKV token selection — the index_fn gathers of LlamaSwiftKVModel.swiftkv_select
 *      (vllm/swiftkv/llama_swiftkv.py:665-685): dst[t][i, :] = src[t][index[i], :] for up to 8 row-major 2-D tensors
 *      (hidden states, residual, positions, k_states, v_states) in ONE launch.  Pointer / stride tables are HOST arrays
 *      passed by value; `index` is a device int64 array (SpecDecode / sampling rows: logits_indices).  Destination rows
 *      may be the decode runner's persistent graph buffers.  Out-of-range indices are skipped.  Graph-capture safe.
 * ---------------------------------------------------------------------------------------- */
int aic_row_gather(int n_tensors, const void* const* src /*host array of device ptrs*/, void* const* dst /*host*/,
                   const int64_t* src_stride_bytes /*host*/, const int64_t* dst_stride_bytes /*host*/,
                   const int32_t* row_bytes /*host*/, const int64_t* index /*device*/, int n_sel, int n_src_rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * A16  Bulk paged-KV write — replaces torch.ops.arctic_inference.reshape_and_cache_flash_bulk
 *      (csrc/custom_ops/torch_bindings.cpp:5-18, kernels.cu:12-155, py_custom_ops.py:40-54).
 *      cache[layer][slot / block_size][slot % block_size][h][d] = cvt(src[token][layer*H*D + h*D + d])
 *      for K and V of all layers in ONE launch; slot < 0 tokens are skipped (kernels.cu:33-35).
 *      Pointer tables are HOST arrays of device pointers (length num_layers) and are passed to the
 *      kernel by value: no per-call H2D copy (the reference does four, kernels.cu:116-146).
 *      kv_dtype: same as src_dtype ("auto") or AIC_DT_FP8_E4M3 / AIC_DT_FP8_E5M2 with x/scale and
 *      saturate-to-finite conversion (quant_utils.cuh:455-489); scale tables may be NULL for auto.
 *      Strides are in elements.  Graph-capture safe.
 * ---------------------------------------------------------------------------------------- */
int aic_reshape_and_cache_flash_bulk(const void* keys, const void* values, void* const* key_cache_ptrs /*host*/,
                                     void* const* value_cache_ptrs /*host*/, const int64_t* slot_mapping,
                                     int num_tokens, int num_layers, int num_heads, int head_size,
                                     int block_size, int64_t block_stride, int64_t key_stride,
                                     int64_t value_stride, int src_dtype, int kv_dtype,
                                     const float* const* k_scale_ptrs /*host*/, const float* const* v_scale_ptrs /*host*/,
                                     void* stream);

/* ------------------------------------------------------------------------------------------
 * A6  Rejection acceptance — replaces the call into vllm.v1.sample.rejection_sampler
 *     (model_runner.py:405-411, parsed at :456-459) for draft_probs == None.
 *     target_logits: [num_draft_total, vocab] (AIC_DT_BF16 / F16 / F32), row stride in elements.
 *     draft_token_ids int32 [num_draft_total]; cu_num_draft int32 [B] inclusive prefix sums;
 *     bonus_token_ids int32 [B]; out int32 [B][max_spec_len+1] (filled with -1 first).
 *     Greedy rows: out[i][p] = argmax(target_logits[row]) until the first draft != argmax; bonus
 *     token at position n_i if nothing was rejected.  Row arg-max ties -> lowest index.
 *     Also emits, per request, what the next proposer step needs (arctic_proposer.py:133-147):
 *     num_accepted[i] (tokens written), last_token[i], hidden_index[i] = (gen_len_i - 1) +
 *     sum_{j<i}(n_j + 1).  Any of those three may be NULL.
 *     target_row_index (int64 [num_draft_total], may be NULL): row r of the call is
 *     target_logits[target_row_index[r]] — the caller passes the model's [T, vocab] logits and
 *     SpecDecodeMetadata.target_logits_indices (model_runner.py:404) instead of gathering the rows.
 *     bonus_row_index (int64 [B], greedy entry point only, may be NULL): the bonus token of request i is the
 *     arg-max of target_logits[bonus_row_index[i]] (the greedy sampler on bonus_logits_indices, model_runner.py:394,
 *     folded into the same launch); bonus_token_ids may then be NULL and the workspace is sized for
 *     num_draft_total + B rows.
 *     workspace: >= aic_rejection_workspace_bytes(num_draft_total, vocab) bytes of HBM.
 * ---------------------------------------------------------------------------------------- */
size_t aic_rejection_workspace_bytes(int num_draft_total, int vocab);
int aic_rejection_greedy(const void* target_logits, int logits_dtype, int64_t row_stride, int vocab,
                         const int32_t* draft_token_ids, const int32_t* cu_num_draft,
                         const int32_t* bonus_token_ids, int batch, int num_draft_total, int max_spec_len,
                         int32_t* out_token_ids, int32_t* num_accepted, int32_t* last_token,
                         int32_t* hidden_index, const int64_t* target_row_index, const int64_t* bonus_row_index,
                         void* workspace, void* stream);
/* Random rows (temperature > 0), draft_probs == None: accept draft iff softmax(logits/T)[draft] >= u,
 * else emit the recovered token argmax_v(p_v / q_v) with p[draft] := 0, q ~ Exp(1).
 * uniform: f64 [num_draft_total]; exp_noise: f32 [B][vocab]; temperature f32 [B]
 * (<= 0 means greedy row).  is_greedy rows use the greedy rule. */
int aic_rejection_random(const void* target_logits, int logits_dtype, int64_t row_stride, int vocab,
                         const int32_t* draft_token_ids, const int32_t* cu_num_draft,
                         const int32_t* bonus_token_ids, const float* temperature, const double* uniform,
                         const float* exp_noise, int batch, int num_draft_total, int max_spec_len,
                         int32_t* out_token_ids, int32_t* num_accepted, int32_t* last_token,
                         int32_t* hidden_index, const int64_t* target_row_index, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * A7-A11  Arctic LSTM speculator (sum_lstm) — replaces ArcticLSTMSpeculator.generate_proposals /
 *     generate_token_ids / generate_states and LogitsProcessorOpt + arg-max
 *     (vllm/spec_dec/arctic_speculator.py:648-866, logits_processor_opt.py:83-107,
 *     fp8.py:276-308, arctic_proposer.py:113-166).
 *     Weights stay owned by the caller (torch tensors); the handle stores pointers + scratch.
 * ---------------------------------------------------------------------------------------- */
typedef struct aic_lstm aic_lstm;

typedef struct aic_lstm_config {
  int32_t vocab_size;        /* rows of the (local) LM head shard */
  int32_t vocab_offset;      /* global index of local row 0 (tp_rank * shard, vocab_parallel_embedding.py:32-35) */
  int32_t input_hidden_dim;  /* H */
  int32_t inner_dim;         /* Ds == proj_dim == emb_dim (arctic_speculator.py:667-688) */
  int32_t n_predict;         /* used for state_weight = 0.5^(0.5/n_predict) (:575-577) */
  int32_t scale_input;       /* ln0 + 1/sqrt(2) on head 0 (:656-657) */
  int32_t max_batch;         /* scratch sizing (padded batch) */
  int32_t head_fp8_max_batch;/* use the fp8 LM head when padded batch <= this (32, :726-728); 0 = never */
} aic_lstm_config;

typedef struct aic_lstm_weights {
  const void* forget_emb;   /* bf16 [V_full, Ds]   replicated embedding (:546-547) */
  const void* proj0;        /* bf16 [4Ds, H]       projs.0.weight = f|i|o|c rows (:874-891) */
  const void* proj1;        /* bf16 [4Ds, Ds]      projs.1.weight */
  const void* cell_ln_w;    /* bf16 [Ds] */
  const void* cell_ln_b;    /* bf16 [Ds] */
  const void* state_ln_w;   /* bf16 [Ds] */
  const void* state_ln_b;   /* bf16 [Ds] */
  const void* head;         /* bf16 [V_local, Ds]  tied LM head */
  const void* head_fp8;     /* e4m3fn [V_local, Ds] per-tensor quantised copy (fp8.py:207-223) or NULL */
  float head_fp8_scale;     /* weight_scale of head_fp8 */
} aic_lstm_weights;

int aic_lstm_create(const aic_lstm_config* cfg, const aic_lstm_weights* w, aic_lstm** out);
/* MLP speculator (ArcticMLPSpeculator, arctic_speculator.py:102-401): per head i
 *   states = proj[i](h) + (emb_weight / state_weight) emb[i][tok];  h = gelu(ln[i](states));  tok = argmax(head[i] h)
 * with ln = MLPSpeculatorLayerNorm (scale and shift), head 0 preceded by ln0(h)/sqrt(2) when scale_input.  The handle
 * is used with the same propose / begin / head / destroy entry points as the LSTM one.  cfg->inner_dim is the model's
 * inner_dim, cfg->input_hidden_dim its emb_dim.  With tie_weights pass the same pointers for the tied stages
 * (proj[1..], emb, ln, head): every distinct matrix is re-laid out once. */
typedef struct aic_mlp_weights {
  int32_t num_heads;        /* max_speculative_tokens, <= 8 */
  const void* emb[8];       /* bf16 [V_full, Ds] */
  const void* proj[8];      /* bf16 [Ds, H] for head 0, [Ds, Ds] after */
  const void* ln_w[8];      /* bf16 [Ds] */
  const void* ln_b[8];      /* bf16 [Ds] */
  const void* head[8];      /* bf16 [V_local, Ds] */
} aic_mlp_weights;
int aic_mlp_create(const aic_lstm_config* cfg, const aic_mlp_weights* w, aic_lstm** out);
/* The LSTM class's "sum_rnn" form with multi-entry dimension lists ("4096.4096": arctic_speculator.py:478-542): emb.i and
 * proj.i are nn.Sequentials [base, (LayerNorm, GELU, Linear)*], ln.i is [LayerNorm, (GELU, Linear, LayerNorm)*]; the base
 * modules are the aic_mlp_weights entries, the extra stages come here (per head i <= 8, stage j <= 3; bf16; every width is
 * inner_dim; LayerNorm = MLPSpeculatorLayerNorm with scale and shift).  Tied stages pass the same pointers. */
typedef struct aic_mlp_stack {
  int32_t n_emb, n_proj, n_ln;       /* extra stages per stack, 0..3 */
  int32_t pad;
  const void* emb_ln_w[8][3];  const void* emb_ln_b[8][3];  const void* emb_lin[8][3];    /* [Ds], [Ds], [Ds, Ds] */
  const void* proj_ln_w[8][3]; const void* proj_ln_b[8][3]; const void* proj_lin[8][3];
  const void* ln_lin[8][3];    const void* ln_ln_w[8][3];   const void* ln_ln_b[8][3];
} aic_mlp_stack;
int aic_mlp_create_stacked(const aic_lstm_config* cfg, const aic_mlp_weights* w, const aic_mlp_stack* stack, aic_lstm** out);
/* C9 (vocab_parallel_embedding.py:425-444): with the token embedding sharded over the speculator's TP group the caller
 * looks the rows up on its shard (zeros for tokens it does not own), all-reduces them, and hands the [batch][inner_dim]
 * bf16 rows to the next aic_lstm_head call of an MLP speculator; NULL goes back to the handle's own tables (which
 * aic_mlp_create accepts as NULL pointers when only looked-up rows will be used). */
int aic_mlp_set_embedding_rows(aic_lstm* m, const void* rows /*device, or NULL*/);
void aic_lstm_destroy(aic_lstm* m);
/* per-tensor e4m3fn quantisation of a bf16 matrix: scale = amax/448, q = sat(x/scale)
 * (ops.scaled_fp8_quant with scale=None, fp8.py:207-210).  scale_out: device f32[1]. */
int aic_quantize_fp8_per_tensor(const void* src_bf16, void* dst_fp8, float* scale_out, int64_t n, void* stream);
/* padding_size(): arctic_speculator.py:39-44 */
int aic_lstm_padding_size(int batch);
/* One call = the whole k-head draft loop on `stream` (no host sync, graph-capture safe):
 *   hidden: bf16 [*, H]; row i of the batch uses hidden[hidden_index[i]] (hidden_index NULL -> i)
 *   last_tokens: int32 [B];  out_tokens: int64 [B][k] GLOBAL token ids (local arg-max + vocab_offset)
 *   out_vals: f32 [B][k] bf16-rounded max logit per head (for the TP arg-max exchange, :733-744), may be NULL */
int aic_lstm_propose(aic_lstm* m, const void* hidden, const int32_t* hidden_index, const int32_t* last_tokens,
                     int batch, int num_predict_tokens, int64_t* out_tokens, float* out_vals, void* stream);
/* debug aid: 1 (default) the whole-draft entry point of an LSTM speculator runs the fused schedule (LM head of head h and
 * gate projection of head h + 1 in one launch, arg-max finished inside the next cell launch: 4 k + 3 launches with an fp8
 * head, 3 k + 3 with a bf16 one), 0 the head-by-head one (5-6 launches per head), 2 the fused one with the fp8 head
 * quantising its activations on the way into LDS (no quant launch; measured slower); all compute the same tokens. */
int aic_debug_lstm_fused(int on);
/* debug aid: launches per LSTM cell in the fused schedule: 1 (default) = one launch whose parts meet at counters in device
 * memory (row sum of the second normalisation, batch |max| of the fp8 head's activation scale), 3 = gates / state /
 * quantisation as three launches (r03's form); bit-identical results. */
int aic_debug_lstm_cell_launches(int n);
/* debug aid: per-phase timestamps of the one-launch cell, buf[head][64 rows][4 parts][12] int64 in device memory (NULL = off) */
int aic_debug_lstm_cell_trace(int64_t* buf);
/* single-head entry points for the vocab-parallel (TP/SP > 1) loop, where an all-gather of
 * (value, index) sits between heads.  State lives in the handle. */
int aic_lstm_begin(aic_lstm* m, const void* hidden, const int32_t* hidden_index, int batch, void* stream);
int aic_lstm_head(aic_lstm* m, int head_index, const int32_t* last_tokens /* [B] global ids */, int batch,
                  int64_t* out_tokens /* [B] */, float* out_vals /* [B] */, void* stream);

/* ------------------------------------------------------------------------------------------
 * A5  Multi-token verify paged attention — what `self._orig_forward(q_, k_, v_)` reaches through
 *     vLLM's attention backend on the reference path (ulysses.py:510).  Causal var-len attention
 *     of q_len_i = 1 + n_draft_i query tokens per request over that request's paged KV
 *     (FlashAttention cache layout [num_blocks, block_size, Hkv, D], llama_swiftkv.py:617), new
 *     K/V already written.  Linear chains only (model_runner.py:734-740): the "tree" mask is
 *     causal within the chunk.
 *       q/out: bf16 [T][Hq][D] (token stride q_stride / out_stride elements)
 *       block_table int32 [B][max_blocks]; seq_lens int32 [B] (context incl. the new tokens);
 *       query_start_loc int32 [B+1].  head_size 128, or 64 with a bf16 cache (anything else:
 *       AIC_ERR_UNSUPPORTED); block_size a multiple of 16.  kv_dtype BF16 or FP8_E4M3 (+ k_scale / v_scale
 *       device scalars).
 *     workspace >= aic_verify_attention_workspace_bytes(...), private to the call until it completes on
 *     `stream`.  Graph-capture safe: this entry point (no host-side request lists) is the form the vLLM
 *     route records under a capturing stream — batch, num_tokens, max_q_len and max_seq_len are then upper
 *     bounds of every replay, requests behind the live ones must have an empty query (query_start_loc
 *     repeated) and are skipped.  Calls for one device come from one host thread at a time (the
 *     fallback that runs long drafts on a library-owned side stream shares its fork/join events).
 * ---------------------------------------------------------------------------------------- */
size_t aic_verify_attention_workspace_bytes(int num_tokens, int num_q_heads, int head_size, int num_splits_max);
int aic_verify_attention(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                         int64_t block_stride, int kv_dtype, const float* k_scale, const float* v_scale,
                         const int32_t* block_table, int max_blocks_per_seq, const int32_t* seq_lens,
                         const int32_t* query_start_loc, int batch, int num_tokens, int max_q_len,
                         int num_q_heads, int num_kv_heads, int head_size, int block_size, float sm_scale,
                         void* out, int64_t out_stride, void* workspace, size_t workspace_bytes,
                         int max_seq_len, void* stream);
/* Same, for callers that know the query lengths on the host (vLLM keeps num_scheduled_tokens there):
 * the batch is partitioned into `short_reqs` (device int32[n_short], requests with q_len * Hq/Hkv <= 32
 * query rows: one pass of the short body, one or two MFMA row tiles per workgroup by its request's rows, KV streamed
 * once) and `long_reqs` (device int32[n_long], e.g. 33-token suffix
 * drafts: a shared-tile body reads their KV once for up to 192 rows instead of once per 16-row group).
 * Both kinds run in ONE launch per call whenever all of its workgroups fit on the chip at once (<= 512;
 * otherwise two launches, the long one on a side stream).  max_q_len bounds the long requests.
 * Lists NULL/0 -> identical to aic_verify_attention. */
int aic_verify_attention_ex(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                            int64_t block_stride, int kv_dtype, const float* k_scale, const float* v_scale,
                            const int32_t* block_table, int max_blocks_per_seq, const int32_t* seq_lens,
                            const int32_t* query_start_loc, int batch, int num_tokens, int max_q_len,
                            int num_q_heads, int num_kv_heads, int head_size, int block_size, float sm_scale,
                            void* out, int64_t out_stride, void* workspace, size_t workspace_bytes,
                            int max_seq_len, const int32_t* short_reqs, int n_short, const int32_t* long_reqs,
                            int n_long, void* stream);
/* The same with the two per-layer features of gpt-oss-class models (BASELINE configs[4]; on the reference path they are
 * arguments of vLLM's Attention layer, which `self._orig_forward` reaches, ulysses.py:510):
 *   sliding_window W > 0: query position p attends keys p - W + 1 .. p only (0: the whole context) — a request then
 *       streams its last W + q_len tokens instead of its context;
 *   sinks: device f32 [num_q_heads] (NULL: none) — one extra logit per head that takes part in the soft-max
 *       normalisation and carries no value (not multiplied by sm_scale).
 * aic_verify_attention_ex(...) == aic_verify_attention_win(..., 0, NULL, stream). */
int aic_verify_attention_win(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                             int64_t block_stride, int kv_dtype, const float* k_scale, const float* v_scale,
                             const int32_t* block_table, int max_blocks_per_seq, const int32_t* seq_lens,
                             const int32_t* query_start_loc, int batch, int num_tokens, int max_q_len,
                             int num_q_heads, int num_kv_heads, int head_size, int block_size, float sm_scale,
                             void* out, int64_t out_stride, void* workspace, size_t workspace_bytes,
                             int max_seq_len, const int32_t* short_reqs, int n_short, const int32_t* long_reqs,
                             int n_long, int sliding_window, const float* sinks, void* stream);
/* aic_verify_attention_ex for `n_layers` layers of one engine step in one call: k_caches / v_caches are HOST arrays of
 * device pointers (one cache pair per layer, same shape / dtype / strides), q and out advance by *_layer_stride elements
 * per layer (0: shared).  For drivers that own the whole step (arcticinference_amd/engine.py); inside vLLM attention is
 * called per layer. */
int aic_verify_attention_layers(const void* q, int64_t q_stride, int64_t q_layer_stride, const void* const* k_caches /*host*/,
                                const void* const* v_caches /*host*/, int n_layers, int64_t block_stride, int kv_dtype,
                                const float* k_scale, const float* v_scale, const int32_t* block_table,
                                int max_blocks_per_seq, const int32_t* seq_lens, const int32_t* query_start_loc, int batch,
                                int num_tokens, int max_q_len, int num_q_heads, int num_kv_heads, int head_size,
                                int block_size, float sm_scale, void* out, int64_t out_stride, int64_t out_layer_stride,
                                void* workspace, size_t workspace_bytes, int max_seq_len, const int32_t* short_reqs,
                                int n_short, const int32_t* long_reqs, int n_long, void* stream);

/* ------------------------------------------------------------------------------------------
 * A12  Ulysses head/sequence repartition — the copies around the two all-to-alls of
 *      UlyssesAttentionPatch.forward (ulysses.py:493-507, :513-517).  The collectives
 *      themselves stay torch.distributed (RCCL) calls in the Python layer.
 *      pack:   q [n][SP*hq*D], k,v [n][SP*hkv*D]  ->  send [SP][n][(hq+2hkv)*D]
 *      split:  recv [SP*n][(hq+2hkv)*D] -> q_ [SP*n][hq*D], k_, v_ [SP*n][hkv*D]   (contiguous outputs)
 *      unpack: recv [SP][n][hq*D] -> out [n][SP*hq*D]
 *      bf16/f16 (2-byte elements).  Graph-capture safe.
 * ---------------------------------------------------------------------------------------- */
int aic_ulysses_pack_qkv(const void* q, const void* k, const void* v, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, void* send, int n_local, int sp, int q_width, int kv_width, void* stream);
int aic_ulysses_split_qkv(const void* recv, void* q, void* k, void* v, int64_t rows, int q_width, int kv_width,
                          void* stream);
int aic_ulysses_unpack_out(const void* recv, void* out, int n_local, int sp, int width, void* stream);

/* KV-replicated variant, fewer kv heads than SP ranks (ulysses.py:462-490): q is packed alone for the all-to-all over
 * SP, K|V are packed for the all-to-all inside the rank's SP_AA group (`parts` = kv heads), and the all-gathered
 * K|V chunks (SP_AG-major) are put back into rank order and split:
 *   pack_pair   : a [n][parts*aw], b [n][parts*bw]  ->  send [parts][n][aw + bw]      (b NULL / bw 0: a alone)
 *   reorder     : gathered [sp][n][2*kw]  ->  k [sp*n][kw], v [sp*n][kw] with destination chunk c = source chunk order[c]
 * `order` is a HOST array of sp ints (a permutation: [j * aa + i for i in range(aa) for j in range(ag)], :449-451). */
int aic_ulysses_pack_pair(const void* a, const void* b, int64_t a_stride, int64_t b_stride, void* send, int n_local,
                          int parts, int a_width, int b_width, void* stream);
int aic_ulysses_reorder_split_kv(const void* gathered, void* k, void* v, int n_chunk_rows, int sp, int kv_width,
                                 const int32_t* order /*host*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * A4 (host)  Index arithmetic of one engine step for drivers that own the step (arcticinference_amd/engine.py), native:
 *     what the reference's execute_model does in Python around the model call — query offsets, contexts, KV slots, target /
 *     bonus rows of SpecDecodeMetadata (model_runner.py:394-404), parse_output + commit of the sampled ids (:456-486).
 *     Pure host code (no device needed); layouts in csrc/engine_host.cpp.
 * ---------------------------------------------------------------------------------------- */
/* `max_num_seqs` = rows of every per-slot array (num_tokens, n_draft, draft_ids, draft_row, block_table, token_ids): each
 * live[i] is checked against it.  aic_step_build returns AIC_ERR_BUFFER_TOO_SMALL when stage_a / stage_b cannot hold the
 * step, with the needed byte counts in totals[6] / totals[7] (filled in either case).  aic_step_parse validates every row
 * (slot ids, row overflow) BEFORE it commits anything: on an error the token rows and counts are untouched. */
int aic_step_build(int n, const int64_t* live, int max_num_seqs, const int32_t* num_tokens, const int32_t* n_draft,
                   const int32_t* draft_ids, int draft_stride, const int64_t* draft_row, int lstm_k,
                   const int32_t* block_table, int blocks_per_seq, int block_size, int group_size, void* stage_a,
                   int64_t cap_a, void* stage_b, int64_t cap_b, int64_t* offs_a /*[5]*/, int64_t* offs_b /*[7]*/,
                   int64_t* totals /*[8]*/, int64_t* ctx_sum);
int aic_step_parse(int n, const int64_t* live, int max_num_seqs, const int32_t* out, int width, int vocab,
                   int32_t* token_ids, int64_t row_stride, int32_t* num_tokens, int32_t* n_emit, int32_t* flat_emit,
                   int64_t* total);

#ifdef __cplusplus
}
#endif
#endif /* ARCTIC_HIP_H_ */
