// Host-side index arithmetic of one engine step, native (no GPU code here).
//
// The stand-alone engine (arcticinference_amd/engine.py) mirrors the bookkeeping of the reference's patched
// execute_model (/root/reference/arctic_inference/vllm/model_runner.py:218-524, vLLM's InputBatch arrays behind it):
// per-slot token rows, token counts, scheduled draft ids.  Building a step's index arrays from them (query offsets,
// contexts, KV slots, short / long request order, draft ids, target / bonus rows) and parsing the acceptance output
// back into the rows was ~15 small numpy calls per step; under sequence parallelism that host time is replicated on every
// rank and serial with the GPU (the next step's query lengths depend on it).  Here each half is one call that writes
// straight into the caller's pinned staging buffers.
#include <algorithm>
#include <cstdint>
#include <cstring>

#include "aic_common.h"

using namespace aic;

namespace {
inline int64_t align16(int64_t x) { return (x + 15) & ~static_cast<int64_t>(15); }
}  // namespace

extern "C" {

// Layout written into stage_a / stage_b (byte offsets returned in offs_a[5] / offs_b[7], every section 16-byte aligned):
//   A: ctx int32[n] | query_start_loc int32[n+1] | live int64[n] | slot_mapping int64[T] | order int32[n]
//   B: draft_flat int32[D] | cu_draft int32[n] | plant int64[T] (left for the caller) | target_rows int64[D] |
//      bonus_rows int64[n] | fill_pos int64[F] | fill_src int64[F]
// totals[8] = {T, max_q, max_ctx, n_short, D (draft tokens), F (pending draft ids to fill on the device), bytes_a, bytes_b};
// ctx_sum = sum of contexts (the attention launches' algorithmic KV tokens).
// Request i of the step is slot live[i]; its query is its last sampled token + its n_draft scheduled draft tokens; token t of
// it sits at position num_tokens - 1 + t and goes to KV slot block_table[slot][pos / bs] * bs + pos % bs.
// order: request indices with q_len * group_size <= 32 first (the short list of aic_verify_attention_ex), then the rest.
// draft_row[slot] >= 0: the slot's draft ids are still on the device (row draft_row of a [*, lstm_k] tensor): its
// positions in draft_flat are listed in fill_pos with fill_src = row * lstm_k + j.
int aic_step_build(int n, const int64_t* live, int max_num_seqs, const int32_t* num_tokens, const int32_t* n_draft,
                   const int32_t* draft_ids, int draft_stride, const int64_t* draft_row, int lstm_k,
                   const int32_t* block_table, int blocks_per_seq, int block_size, int group_size, void* stage_a,
                   int64_t cap_a, void* stage_b, int64_t cap_b, int64_t* offs_a, int64_t* offs_b, int64_t* totals,
                   int64_t* ctx_sum) {
  AIC_REQUIRE(n > 0 && live && num_tokens && n_draft && draft_ids && block_table && stage_a && stage_b && offs_a && offs_b &&
                  totals && ctx_sum && block_size > 0 && group_size > 0 && blocks_per_seq > 0 && draft_stride > 0 &&
                  max_num_seqs > 0 && n <= max_num_seqs,
              "bad arguments to aic_step_build");
  int64_t T = 0, D = 0, F = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t s = live[i];
    // every per-slot array below (num_tokens, n_draft, draft_ids, draft_row, block_table) has max_num_seqs rows
    AIC_REQUIRE(s >= 0 && s < max_num_seqs, "live[%d] = %lld is not a slot of a %d-slot batch", i, static_cast<long long>(s),
                max_num_seqs);
    AIC_REQUIRE(n_draft[s] >= 0 && n_draft[s] <= draft_stride && num_tokens[s] >= 1, "slot %lld: bad draft / token count",
                static_cast<long long>(s));
    T += n_draft[s] + 1;
    D += n_draft[s];
    if (draft_row && draft_row[s] >= 0) F += n_draft[s];
  }
  int64_t o = 0;
  offs_a[0] = o; o = align16(o + 4 * static_cast<int64_t>(n));
  offs_a[1] = o; o = align16(o + 4 * static_cast<int64_t>(n + 1));
  offs_a[2] = o; o = align16(o + 8 * static_cast<int64_t>(n));
  offs_a[3] = o; o = align16(o + 8 * T);
  offs_a[4] = o; o = align16(o + 4 * static_cast<int64_t>(n));
  const int64_t bytes_a = o;
  o = 0;
  offs_b[0] = o; o = align16(o + 4 * D);
  offs_b[1] = o; o = align16(o + 4 * static_cast<int64_t>(n));
  offs_b[2] = o; o = align16(o + 8 * T);
  offs_b[3] = o; o = align16(o + 8 * D);
  offs_b[4] = o; o = align16(o + 8 * static_cast<int64_t>(n));
  offs_b[5] = o; o = align16(o + 8 * F);
  offs_b[6] = o; o = align16(o + 8 * F);
  const int64_t bytes_b = o;
  // the sizes are reported whether or not the buffers hold them: a caller grows to totals[6..7] and calls again on
  // AIC_ERR_BUFFER_TOO_SMALL (a code of its own — nobody should have to read the message to tell this case apart)
  totals[0] = T; totals[4] = D; totals[5] = F; totals[6] = bytes_a; totals[7] = bytes_b;
  if (bytes_a > cap_a || bytes_b > cap_b) {
    set_error("staging buffers too small (%lld / %lld bytes needed, %lld / %lld given)", static_cast<long long>(bytes_a),
              static_cast<long long>(bytes_b), static_cast<long long>(cap_a), static_cast<long long>(cap_b));
    return AIC_ERR_BUFFER_TOO_SMALL;
  }
  char* A = static_cast<char*>(stage_a);
  char* B = static_cast<char*>(stage_b);
  int32_t* ctx = reinterpret_cast<int32_t*>(A + offs_a[0]);
  int32_t* qsl = reinterpret_cast<int32_t*>(A + offs_a[1]);
  int64_t* live_o = reinterpret_cast<int64_t*>(A + offs_a[2]);
  int64_t* slots = reinterpret_cast<int64_t*>(A + offs_a[3]);
  int32_t* order = reinterpret_cast<int32_t*>(A + offs_a[4]);
  int32_t* draft_flat = reinterpret_cast<int32_t*>(B + offs_b[0]);
  int32_t* cu_draft = reinterpret_cast<int32_t*>(B + offs_b[1]);
  int64_t* target_rows = reinterpret_cast<int64_t*>(B + offs_b[3]);
  int64_t* bonus_rows = reinterpret_cast<int64_t*>(B + offs_b[4]);
  int64_t* fill_pos = reinterpret_cast<int64_t*>(B + offs_b[5]);
  int64_t* fill_src = reinterpret_cast<int64_t*>(B + offs_b[6]);

  int64_t t = 0, d = 0, f = 0, csum = 0;
  int max_q = 0, max_ctx = 0, n_short = 0;
  qsl[0] = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t s = live[i];
    const int nd = n_draft[s], ql = nd + 1, ntok = num_tokens[s];
    const int c = ntok + nd;   // context once this step's tokens are written
    ctx[i] = c;
    live_o[i] = s;
    csum += c;
    max_q = std::max(max_q, ql);
    max_ctx = std::max(max_ctx, c);
    if (ql * group_size <= 32) ++n_short;
    const int32_t* bt = block_table + s * blocks_per_seq;
    for (int j = 0; j < ql; ++j) {
      const int pos = ntok - 1 + j;
      AIC_REQUIRE(pos / block_size < blocks_per_seq, "slot %lld: position %d beyond its block table", static_cast<long long>(s), pos);
      slots[t + j] = static_cast<int64_t>(bt[pos / block_size]) * block_size + pos % block_size;
      if (j < nd) target_rows[d + j] = t + j;   // every row but the request's last verifies a draft token
    }
    bonus_rows[i] = t + ql - 1;
    const int32_t* di = draft_ids + s * draft_stride;
    const bool pending = draft_row && draft_row[s] >= 0;
    for (int j = 0; j < nd; ++j) {
      draft_flat[d + j] = di[j];
      if (pending) {
        fill_pos[f] = d + j;
        fill_src[f] = draft_row[s] * lstm_k + j;
        ++f;
      }
    }
    t += ql;
    d += nd;
    cu_draft[i] = static_cast<int32_t>(d);
    qsl[i + 1] = static_cast<int32_t>(t);
  }
  // short requests first, each group in batch order (ops.split_order)
  int a = 0, b = n_short;
  for (int i = 0; i < n; ++i) {
    const int ql = n_draft[live[i]] + 1;
    if (ql * group_size <= 32) order[a++] = i; else order[b++] = i;
  }
  totals[0] = T; totals[1] = max_q; totals[2] = max_ctx; totals[3] = n_short;
  totals[4] = D; totals[5] = F; totals[6] = bytes_a; totals[7] = bytes_b;
  *ctx_sum = csum;
  return AIC_OK;
}

// parse_output + commit (model_runner.py:456-486) for a whole step: row i of `out` (int32 [n][width], -1 padded) keeps
// the ids that are not -1 and < vocab; they are appended to token_ids[slot] at num_tokens[slot], which advances.
// n_emit[i] and the concatenated ids (flat_emit, capacity n * width) are returned for the suffix-cache update.
int aic_step_parse(int n, const int64_t* live, int max_num_seqs, const int32_t* out, int width, int vocab,
                   int32_t* token_ids, int64_t row_stride, int32_t* num_tokens, int32_t* n_emit, int32_t* flat_emit,
                   int64_t* total) {
  AIC_REQUIRE(n >= 0 && live && out && token_ids && num_tokens && n_emit && flat_emit && total && width > 0 && row_stride > 0 &&
                  max_num_seqs > 0 && n <= max_num_seqs,
              "bad arguments to aic_step_parse");
  // Pass 1 — nothing is written: every slot id is a slot, no slot twice (two rows committing into one token row), no row
  // overflows.  A failure leaves the engine's state exactly as it was (no half-applied step).
  for (int i = 0; i < n; ++i) {
    const int64_t s = live[i];
    AIC_REQUIRE(s >= 0 && s < max_num_seqs, "live[%d] = %lld is not a slot of a %d-slot batch", i, static_cast<long long>(s),
                max_num_seqs);
    AIC_REQUIRE(num_tokens[s] >= 0, "slot %lld: negative token count", static_cast<long long>(s));
    int cnt = 0;
    const int32_t* o = out + static_cast<int64_t>(i) * width;
    for (int j = 0; j < width; ++j) cnt += (o[j] != -1 && o[j] < vocab);
    AIC_REQUIRE(static_cast<int64_t>(num_tokens[s]) + cnt <= row_stride, "slot %lld: token row overflow (%d + %d > %lld)",
                static_cast<long long>(s), num_tokens[s], cnt, static_cast<long long>(row_stride));
  }
  // Pass 2 — commit
  int64_t k = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t s = live[i];
    int32_t* row = token_ids + s * row_stride;
    int at = num_tokens[s], cnt = 0;
    const int32_t* o = out + static_cast<int64_t>(i) * width;
    for (int j = 0; j < width; ++j) {
      const int32_t v = o[j];
      if (v != -1 && v < vocab) {
        row[at++] = v;
        flat_emit[k++] = v;
        ++cnt;
      }
    }
    num_tokens[s] = at;
    n_emit[i] = cnt;
  }
  *total = k;
  return AIC_OK;
}

}  // extern "C"
