// A6 — rejection acceptance of draft tokens against the target model's verify logits.
//
// The reference delegates this to vllm.v1.sample.rejection_sampler.RejectionSampler
// (call site /root/reference/arctic_inference/vllm/model_runner.py:405-411, output parsed at
// :456-459; draft_probs is always None there, :407).  Semantics restated in oracle/spec_oracle.py.
//
// CDNA4 mapping.  The cost is reading the [num_draft, vocab] logits once (770 KB per request at
// k=3, bf16): an HBM-bound row reduction.  num_draft is small (<= ~200 rows at B=64), far fewer
// than the 256 CUs need, so every row is split into S vocab segments -> grid (rows, S) of
// 256-thread workgroups streaming 16 bytes per lane, each writing a partial (max, argmax); the
// accept step is one wavefront per request: lane p owns draft position p, folds that row's S
// partials, and the first rejected position comes from a 64-bit ballot + ffs (the "accept scan"),
// so the whole scan is two cross-lane operations instead of a serial loop.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "aic_common.h"

namespace aic {

constexpr int kMaxSplits = 64;
constexpr int kSegQuantum = 2048;  // 256 threads x 8 elements

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kVec = 4;
  static __device__ __forceinline__ float get(const float* p, int e) { return p[e]; }
  static __device__ __forceinline__ float round(float x) { return x; }
};
struct bf16_t { uint16_t b; };
struct f16_t { uint16_t b; };
template <>
struct Elem<bf16_t> {
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float get(const bf16_t* p, int e) { return bf16_to_f32(p[e].b); }
  static __device__ __forceinline__ float round(float x) { return round_bf16(x); }
};
template <>
struct Elem<f16_t> {
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float get(const f16_t* p, int e) { return f16_to_f32(p[e].b); }
  static __device__ __forceinline__ float round(float x) { return round_f16(x); }
};

struct Best {
  float v;
  int i;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  // larger value wins; equal values -> lower index (torch.argmax on a row returns the first maximum)
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
__device__ __forceinline__ Best wave_best(Best x) {
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.v = __shfl_xor(x.v, off);
    o.i = __shfl_xor(x.i, off);
    x = better(x, o);
  }
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
  for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

__device__ __forceinline__ int request_of_row(const int32_t* cu, int batch, int row) {
  int lo = 0, hi = batch - 1;  // first request whose inclusive prefix sum exceeds row
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cu[mid] > row) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// MODE 0: arg-max of the raw logits (greedy rows).
// MODE 1: softmax statistics of x = round(l / T): partial (max, sum exp(x - max)).
// MODE 2: recovered-token arg-max of p_v / q_v with p[draft] := 0, and p[draft] itself;
//         greedy rows (T <= 0) fall back to the MODE 0 arg-max.
template <typename T, int MODE, bool VEC_OK>
__global__ void __launch_bounds__(256)
logits_row_kernel(const T* __restrict__ logits, int64_t row_stride, int vocab, int seg_len, int n_splits,
                  const int32_t* __restrict__ draft_ids, const int32_t* __restrict__ cu, int batch,
                  const float* __restrict__ temperature, const float* __restrict__ exp_noise,
                  float* __restrict__ part_val, int32_t* __restrict__ part_idx, float* __restrict__ stat_max,
                  float* __restrict__ stat_sum, float* __restrict__ p_draft, const int64_t* __restrict__ row_index,
                  const int64_t* __restrict__ bonus_row_index, int n_draft_rows) {
  constexpr int V = Elem<T>::kVec;
  const int row = blockIdx.x;
  const int seg = blockIdx.y;
  const int begin = seg * seg_len;
  const int end = min(begin + seg_len, vocab);
  // row r of the call lives at logits[row_index[r]] when the caller passes the un-gathered [T, V] logits; rows past
  // the draft rows are the requests' bonus rows (greedy sampler folded into this launch)
  const T* base = logits + (row >= n_draft_rows ? bonus_row_index[row - n_draft_rows]
                                                : (row_index ? row_index[row] : static_cast<int64_t>(row))) * row_stride;
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ float s_m, s_z;

  float inv_t_is_greedy = 0.0f;
  float temp = 1.0f;
  int req = 0, draft = -1;
  const float* q = nullptr;
  bool greedy_row = (MODE == 0);
  if (MODE != 0) {
    req = request_of_row(cu, batch, row);
    temp = temperature[req];
    greedy_row = !(temp > 0.0f);
    draft = draft_ids[row];
    q = exp_noise ? exp_noise + static_cast<int64_t>(req) * vocab : nullptr;
  }
  (void)inv_t_is_greedy;
  if (MODE == 1 && seg == 0 && threadIdx.x == 0) p_draft[row] = 0.0f;  // a draft id outside the vocab has p = 0

  float row_max = 0.0f, row_z = 1.0f;
  if (MODE == 2 && !greedy_row) {
    // fold the softmax partials of this row (every block redoes this tiny reduction)
    if (threadIdx.x < 64) {
      float m = -INFINITY;
      for (int s = threadIdx.x; s < n_splits; s += 64) m = fmaxf(m, stat_max[row * n_splits + s]);
      m = wave_max(m);
      float z = 0.0f;
      for (int s = threadIdx.x; s < n_splits; s += 64) {
        const float pm = stat_max[row * n_splits + s];
        if (pm > -INFINITY) z += stat_sum[row * n_splits + s] * __expf(pm - m);
      }
      z = wave_sum(z);
      if (threadIdx.x == 0) {
        s_m = m;
        s_z = z;
      }
    }
    __syncthreads();
    row_max = s_m;
    row_z = s_z;
  }

  Best best;
  best.v = -INFINITY;
  best.i = begin;
  float run_m = -INFINITY, run_z = 0.0f;

  auto visit = [&](int idx, float l) {
    if (MODE == 0 || (MODE == 2 && greedy_row)) {
      if (l > best.v) {
        best.v = l;
        best.i = idx;
      }
    } else {
      const float x = Elem<T>::round(__fdiv_rn(l, temp));  // logits.div_(temperature) in the logits dtype
      if (MODE == 1) {
        if (x > run_m) {
          run_z = run_z * __expf(run_m - x) + 1.0f;
          run_m = x;
        } else if (x > -INFINITY) {
          run_z += __expf(x - run_m);
        }
      } else {
        float p = __fdiv_rn(__expf(x - row_max), row_z);
        if (idx == draft) {
          p_draft[row] = p;
          p = 0.0f;
        }
        const float r = __fdiv_rn(p, q[idx]);
        if (r > best.v) {
          best.v = r;
          best.i = idx;
        }
      }
    }
  };

  if (VEC_OK) {
    for (int i = begin + threadIdx.x * V; i < end; i += 256 * V) {
      if (i + V <= end) {
        T buf[V];
        *reinterpret_cast<uint4*>(buf) = *reinterpret_cast<const uint4*>(base + i);
#pragma unroll
        for (int e = 0; e < V; ++e) visit(i + e, Elem<T>::get(buf, e));
      } else {
        for (int e = 0; i + e < end; ++e) visit(i + e, Elem<T>::get(base + i, e));
      }
    }
  } else {
    for (int i = begin + threadIdx.x; i < end; i += 256) visit(i, Elem<T>::get(base + i, 0));
  }

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (MODE == 1 && !greedy_row) {
    // combine (max, sum) pairs across the block
    float m = wave_max(run_m);
    float z = run_m > -INFINITY ? run_z * __expf(run_m - m) : 0.0f;
    z = wave_sum(z);
    if (lane == 0) {
      s_v[wave] = m;
      reinterpret_cast<float*>(s_i)[wave] = z;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float M = fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]));
      float Z = 0.0f;
      for (int w = 0; w < 4; ++w)
        if (s_v[w] > -INFINITY) Z += reinterpret_cast<float*>(s_i)[w] * __expf(s_v[w] - M);
      stat_max[row * n_splits + seg] = M;
      stat_sum[row * n_splits + seg] = Z;
    }
    return;
  }
  if (MODE == 1) return;  // greedy row: nothing to prepare

  best = wave_best(best);
  if (lane == 0) {
    s_v[wave] = best.v;
    s_i[wave] = best.i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best b;
    b.v = s_v[0];
    b.i = s_i[0];
    for (int w = 1; w < 4; ++w) {
      Best o;
      o.v = s_v[w];
      o.i = s_i[w];
      b = better(b, o);
    }
    part_val[row * n_splits + seg] = b.v;
    part_idx[row * n_splits + seg] = b.i;
  }
}

// One wavefront per request; lane p owns draft position p.
template <bool RANDOM>
__global__ void __launch_bounds__(64)
accept_kernel(const float* __restrict__ part_val, const int32_t* __restrict__ part_idx, int n_splits,
              const int32_t* __restrict__ draft_ids, const int32_t* __restrict__ cu,
              const int32_t* __restrict__ bonus, const float* __restrict__ temperature,
              const double* __restrict__ uniform, const float* __restrict__ p_draft, int max_spec_len,
              int32_t* __restrict__ out, int32_t* __restrict__ num_accepted, int32_t* __restrict__ last_token,
              int32_t* __restrict__ hidden_index, int bonus_part_row0) {
  const int req = blockIdx.x;
  const int lane = threadIdx.x;
  const int start = req == 0 ? 0 : cu[req - 1];
  const int n = cu[req] - start;
  const int width = max_spec_len + 1;
  int32_t* orow = out + static_cast<int64_t>(req) * width;

  // bonus token: the sampler's id, or (bonus_part_row0 >= 0) the arg-max partials of this request's bonus row
  int bonus_tok;
  if (bonus_part_row0 >= 0) {
    const int brow = bonus_part_row0 + req;
    Best b;
    b.v = part_val[brow * n_splits];
    b.i = part_idx[brow * n_splits];
    for (int s = 1; s < n_splits; ++s) {
      Best o;
      o.v = part_val[brow * n_splits + s];
      o.i = part_idx[brow * n_splits + s];
      b = better(b, o);
    }
    bonus_tok = b.i;
  } else {
    bonus_tok = bonus[req];
  }
  bool reject = false;
  int token = -1;
  if (lane < n) {
    const int row = start + lane;
    Best b;
    b.v = part_val[row * n_splits];
    b.i = part_idx[row * n_splits];
    for (int s = 1; s < n_splits; ++s) {
      Best o;
      o.v = part_val[row * n_splits + s];
      o.i = part_idx[row * n_splits + s];
      b = better(b, o);
    }
    const int draft = draft_ids[row];
    bool greedy = true;
    if (RANDOM) greedy = !(temperature[req] > 0.0f);
    if (greedy) {
      token = b.i;  // the target's own arg-max is emitted whether or not it equals the draft
      reject = (draft != b.i);
    } else {
      const bool accept = static_cast<double>(p_draft[row]) >= uniform[row];
      token = accept ? draft : b.i;  // b.i = recovered token
      reject = !accept;
    }
  }
  const unsigned long long rej = __ballot(reject);
  const int first_rej = rej ? (__ffsll(static_cast<long long>(rej)) - 1) : n;  // accept scan
  const int written = first_rej < n ? first_rej + 1 : n + 1;
  if (lane < width) {
    int32_t v = -1;
    if (lane < n && lane <= first_rej) v = token;
    if (lane == n && first_rej == n) v = bonus_tok;
    orow[lane] = v;
  }
  // what the proposer needs next (arctic_proposer.py:133-147)
  const int last_lane = written - 1;
  int last = __shfl(token, min(last_lane, 63));
  if (last_lane == n) last = bonus_tok;
  if (lane == 0) {
    if (num_accepted) num_accepted[req] = written;
    if (last_token) last_token[req] = last;
    if (hidden_index) hidden_index[req] = (written - 1) + start + req;
  }
}

static int pick_splits(int rows, int vocab) {
  int s = (1024 + rows - 1) / rows;
  const int max_by_len = (vocab + kSegQuantum - 1) / kSegQuantum;
  if (s > max_by_len) s = max_by_len;
  if (s > kMaxSplits) s = kMaxSplits;
  if (s < 1) s = 1;
  return s;
}

template <typename T>
static int run_rejection(const void* logits, int64_t row_stride, int vocab, const int32_t* draft,
                         const int32_t* cu, const int32_t* bonus, const float* temperature, const double* uniform,
                         const float* noise, int batch, int rows, int max_spec_len, int32_t* out, int32_t* nacc,
                         int32_t* last, int32_t* hidx, const int64_t* row_index, const int64_t* bonus_rows, void* workspace,
                         hipStream_t stream, bool random) {
  const int grid_rows = rows + (bonus_rows ? batch : 0);   // bonus rows are reduced in the same launch
  const int S = grid_rows > 0 ? pick_splits(grid_rows, vocab) : 1;  // a step without drafts still emits bonus tokens
  int seg_len = (vocab + S - 1) / S;
  seg_len = (seg_len + kSegQuantum - 1) / kSegQuantum * kSegQuantum;
  const int n_splits = (vocab + seg_len - 1) / seg_len;
  float* part_val = static_cast<float*>(workspace);
  const size_t wr = static_cast<size_t>(grid_rows);
  int32_t* part_idx = reinterpret_cast<int32_t*>(part_val + wr * kMaxSplits);
  float* stat_max = reinterpret_cast<float*>(part_idx + wr * kMaxSplits);
  float* stat_sum = stat_max + wr * kMaxSplits;
  float* p_draft = stat_sum + wr * kMaxSplits;
  const T* lg = static_cast<const T*>(logits);
  const bool vec_ok = (reinterpret_cast<uintptr_t>(logits) % 16 == 0) && ((row_stride * sizeof(T)) % 16 == 0);
  if (grid_rows > 0) {
    dim3 grid(grid_rows, n_splits);
#define AIC_ROW_LAUNCH(MODE)                                                                                      \
  if (vec_ok)                                                                                                     \
    hipLaunchKernelGGL((logits_row_kernel<T, MODE, true>), grid, dim3(256), 0, stream, lg, row_stride, vocab,     \
                       seg_len, n_splits, draft, cu, batch, temperature, noise, part_val, part_idx, stat_max,     \
                       stat_sum, p_draft, row_index, bonus_rows, rows);                                           \
  else                                                                                                            \
    hipLaunchKernelGGL((logits_row_kernel<T, MODE, false>), grid, dim3(256), 0, stream, lg, row_stride, vocab,    \
                       seg_len, n_splits, draft, cu, batch, temperature, noise, part_val, part_idx, stat_max,     \
                       stat_sum, p_draft, row_index, bonus_rows, rows);
    if (!random) {
      AIC_ROW_LAUNCH(0)
    } else {
      AIC_ROW_LAUNCH(1)
      int rc = launch_status("logits_row_kernel<stats>");
      if (rc != AIC_OK) return rc;
      AIC_ROW_LAUNCH(2)
    }
#undef AIC_ROW_LAUNCH
    int rc = launch_status("logits_row_kernel");
    if (rc != AIC_OK) return rc;
  }
  if (random)
    hipLaunchKernelGGL((accept_kernel<true>), dim3(batch), dim3(64), 0, stream, part_val, part_idx, n_splits, draft,
                       cu, bonus, temperature, uniform, p_draft, max_spec_len, out, nacc, last, hidx, bonus_rows ? rows : -1);
  else
    hipLaunchKernelGGL((accept_kernel<false>), dim3(batch), dim3(64), 0, stream, part_val, part_idx, n_splits, draft,
                       cu, bonus, temperature, uniform, p_draft, max_spec_len, out, nacc, last, hidx, bonus_rows ? rows : -1);
  return launch_status("accept_kernel");
}

static int dispatch(const void* logits, int dtype, int64_t row_stride, int vocab, const int32_t* draft,
                    const int32_t* cu, const int32_t* bonus, const float* temperature, const double* uniform,
                    const float* noise, int batch, int rows, int max_spec_len, int32_t* out, int32_t* nacc,
                    int32_t* last, int32_t* hidx, const int64_t* row_index, const int64_t* bonus_rows, void* ws, void* stream,
                    bool random) {
  if (batch == 0) return AIC_OK;
  AIC_REQUIRE(cu && (bonus || bonus_rows) && out && batch > 0 && rows >= 0 && vocab > 0, "bad arguments to rejection");
  AIC_REQUIRE(rows == 0 || (logits && draft && ws), "null logits / draft ids / workspace");
  AIC_REQUIRE(!bonus_rows || (!random && logits && ws), "bonus rows are folded into the greedy launch only");
  AIC_REQUIRE(max_spec_len >= 0 && max_spec_len + 1 <= 64, "max_spec_len must be <= 63 (one lane per position)");
  AIC_REQUIRE(!random || (temperature && uniform && noise), "random rejection needs temperature, uniform and noise");
  AIC_NEED_DEVICE();
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case AIC_DT_F32:
      return run_rejection<float>(logits, row_stride, vocab, draft, cu, bonus, temperature, uniform, noise, batch,
                                  rows, max_spec_len, out, nacc, last, hidx, row_index, bonus_rows, ws, s, random);
    case AIC_DT_BF16:
      return run_rejection<bf16_t>(logits, row_stride, vocab, draft, cu, bonus, temperature, uniform, noise, batch,
                                   rows, max_spec_len, out, nacc, last, hidx, row_index, bonus_rows, ws, s, random);
    case AIC_DT_F16:
      return run_rejection<f16_t>(logits, row_stride, vocab, draft, cu, bonus, temperature, uniform, noise, batch,
                                  rows, max_spec_len, out, nacc, last, hidx, row_index, bonus_rows, ws, s, random);
    default:
      set_error("unsupported logits dtype %d", dtype);
      return AIC_ERR_UNSUPPORTED;
  }
}

}  // namespace aic

using namespace aic;

extern "C" {

size_t aic_rejection_workspace_bytes(int num_draft_total, int vocab) {
  (void)vocab;
  const size_t rows = static_cast<size_t>(num_draft_total > 0 ? num_draft_total : 1);
  return rows * kMaxSplits * 4 * sizeof(float) + rows * sizeof(float) + 256;
}

int aic_rejection_greedy(const void* target_logits, int logits_dtype, int64_t row_stride, int vocab,
                         const int32_t* draft_token_ids, const int32_t* cu_num_draft, const int32_t* bonus_token_ids,
                         int batch, int num_draft_total, int max_spec_len, int32_t* out_token_ids,
                         int32_t* num_accepted, int32_t* last_token, int32_t* hidden_index,
                         const int64_t* target_row_index, const int64_t* bonus_row_index, void* workspace,
                         void* stream) {
  return dispatch(target_logits, logits_dtype, row_stride, vocab, draft_token_ids, cu_num_draft, bonus_token_ids,
                  nullptr, nullptr, nullptr, batch, num_draft_total, max_spec_len, out_token_ids, num_accepted,
                  last_token, hidden_index, target_row_index, bonus_row_index, workspace, stream, false);
}

int aic_rejection_random(const void* target_logits, int logits_dtype, int64_t row_stride, int vocab,
                         const int32_t* draft_token_ids, const int32_t* cu_num_draft, const int32_t* bonus_token_ids,
                         const float* temperature, const double* uniform, const float* exp_noise, int batch,
                         int num_draft_total, int max_spec_len, int32_t* out_token_ids, int32_t* num_accepted,
                         int32_t* last_token, int32_t* hidden_index, const int64_t* target_row_index, void* workspace,
                         void* stream) {
  return dispatch(target_logits, logits_dtype, row_stride, vocab, draft_token_ids, cu_num_draft, bonus_token_ids,
                  temperature, uniform, exp_noise, batch, num_draft_total, max_spec_len, out_token_ids, num_accepted,
                  last_token, hidden_index, target_row_index, nullptr, workspace, stream, true);
}

}  // extern "C"
