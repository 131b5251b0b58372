// Suffix-tree candidate matching on MI355X + the host<->HBM mirror of the trees, and the C ABI of
// aic_st_* / aic_sc_* (include/arctic_hip.h).
//
// Reference behaviour: SuffixTree::speculate / _match_pattern / _speculate_path
// (csrc/suffix_cache/suffix_tree.cc:135-224) and the SuffixCache policy
// (arctic_inference/common/suffix_cache/suffix_cache.py:151-222).
//
// Kernel mapping (CDNA4): the work of one engine step is
//     queries x {prompt tree, global tree} x suffix starts   (<= 64 x 2 x 64 = 8192 items)
// and every item is a chain of *dependent* HBM/L2 reads (hash probe -> node record -> edge tokens).
// It is latency-bound, not bandwidth-bound, so the mapping maximises independent chains in flight:
// one 64-lane wavefront per item (8192 waves = one full residency of 256 CUs x 32 waves), lanes
// cooperating only where the data is wide: comparing / emitting up to 64 edge tokens per step.
// A second one-wave-per-query kernel reduces the per-start candidates with the reference's tie
// rules (strict >, earlier start wins; prompt tree wins ties against the global tree).
// Tree mode (_speculate_tree, suffix_tree.cc:226-274; the reference's simulator default, never the serving path) uses the
// same decomposition with suffix_tree_spec_kernel: the walk is shared, the candidate is grown by a per-wave priority queue
// over child lists mirrored in the host container's order (see the comment at that kernel).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "aic_common.h"
#include "suffix_host.hpp"
#include "suffix_layout.h"

namespace aic {

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Applies mirror deltas: job j copies n records of rec_words words from the blob to their slots.
__global__ void __launch_bounds__(256) mirror_apply_kernel(const ApplyJob* __restrict__ jobs,
                                                           const int32_t* __restrict__ blob) {
  const ApplyJob job = jobs[blockIdx.y];
  const int64_t total = static_cast<int64_t>(job.n) * job.rec_words;
  const int32_t* src = blob + job.src_off;
  const int32_t* idx = job.idx_off >= 0 ? blob + job.idx_off : nullptr;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = t / job.rec_words;
    const int32_t w = static_cast<int32_t>(t - r * job.rec_words);
    const int64_t d = idx ? static_cast<int64_t>(idx[r]) : (static_cast<int64_t>(job.dst_first) + r);
    job.dst[d * job.rec_words + w] = src[t];
  }
}

__device__ __forceinline__ int32_t lookup_child(const TreeDesc& T, int32_t parent, int32_t token) {
  uint32_t h = edge_hash(parent, token) & T.hash_mask;
  // load factor <= 1/2: ~1.5 probes on average; every lane probes the same slot (one 16-B request)
  for (uint32_t guard = 0; guard <= T.hash_mask; ++guard) {
    const int4 s = *reinterpret_cast<const int4*>(&T.hash[h]);
    const int32_t state = uni(s.w);
    if (state == SLOT_EMPTY) return -1;
    if (state == SLOT_FULL && uni(s.x) == parent && uni(s.y) == token) return uni(s.z);
    h = (h + 1) & T.hash_mask;
  }
  return -1;
}

// State of a position in a tree: `idx` tokens into the edge that leads into `node`.
struct Cursor {
  int32_t node, idx, node_len, node_count, node_best;
  int64_t label;  // offset of the current edge's first token in T.tokens
};

__device__ __forceinline__ void load_node(const TreeDesc& T, int32_t node, Cursor& c) {
  const int4 a = *reinterpret_cast<const int4*>(&T.nodes[node]);                // count,parent,seq_slot,start
  const int4 b = *(reinterpret_cast<const int4*>(&T.nodes[node]) + 1);          // length,best,alive,pad
  c.node = node;
  c.node_count = uni(a.x);
  c.node_best = uni(b.y);
  const int2 sm = *reinterpret_cast<const int2*>(&T.seq_base[2 * uni(a.z)]);   // {region base, sequence length}
  c.node_len = uni(b.x) != kOpenLength ? uni(b.x) : uni(sm.y) - uni(a.w);       // open leaf: to the sequence's end
  c.label = static_cast<int64_t>(uni(sm.x)) + uni(a.w);
}

// walk the pattern suffix pat[s:n) down from the root (_match_pattern, suffix_tree.cc:167-188); wave-uniform result
__device__ __forceinline__ bool match_walk(const TreeDesc& T, const int32_t* pat, int s, int n, int lane, Cursor& c) {
  c.node = 0;
  c.idx = 0;
  c.node_len = 0;
  c.node_count = 0;
  c.node_best = -1;
  c.label = 0;
  int i = s;
  while (i < n) {
    if (c.idx >= c.node_len) {
      const int32_t child = lookup_child(T, c.node, uni(pat[i]));
      if (child < 0) return false;
      load_node(T, child, c);
      c.idx = 0;
    }
    // compare the rest of this edge with the pattern, 64 tokens per step across the lanes
    const int m = min(c.node_len - c.idx, n - i);
    bool same = true;
    for (int j = lane; j < m; j += 64) same &= (T.tokens[c.label + c.idx + j] == pat[i + j]);
    if (!__all(same)) return false;
    c.idx += m;
    i += m;
  }
  return true;
}

// budget (:149-152): float multiply-add without contraction, double +1e-6, truncate
__device__ __forceinline__ int spec_budget_dev(int match_len, const QueryRec& Q, int cap) {
  const float scaled = __fadd_rn(__fmul_rn(static_cast<float>(match_len), Q.factor), Q.offset);
  int budget = static_cast<int>(static_cast<double>(scaled) + 1e-6);
  budget = max(min(budget, Q.max_spec), 0);
  return min(budget, cap);
}

// One wavefront per (query, tree, suffix start).
__global__ void __launch_bounds__(256)
suffix_match_kernel(const QueryRec* __restrict__ queries, const TreeDesc* __restrict__ trees,
                    const int32_t* __restrict__ patterns, int n_starts, int cap, int n_items,
                    float* __restrict__ scr_score, int32_t* __restrict__ scr_n,
                    int32_t* __restrict__ scr_tok, float* __restrict__ scr_prob) {
  const int lane = threadIdx.x & 63;
  const int item = uni(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (item >= n_items) return;
  const int per_q = 2 * n_starts;
  const int qi = item / per_q;
  const int rem = item - qi * per_q;
  const int tr = rem / n_starts;
  const int s = rem - tr * n_starts;

  const QueryRec Q = queries[qi];
  const int tree_idx = uni(tr == 0 ? Q.prompt_tree : Q.global_tree);
  const int n = uni(Q.pattern_len);
  float score = 0.0f;
  int nt = 0;

  if (tree_idx >= 0 && s < n) {
    const TreeDesc T = trees[tree_idx];
    const int32_t* pat = patterns + uni(Q.pattern_off);
    Cursor c;
    if (match_walk(T, pat, s, n, lane, c)) {
      const int budget = spec_budget_dev(n - s, Q, cap);
      const float min_prob = Q.min_prob;

      // ---- follow the most frequent continuation (_speculate_path, :190-224) --------------------
      float prob = 1.0f;
      const int64_t out = static_cast<int64_t>(item) * cap;
      while (nt < budget && prob >= min_prob) {
        if (c.idx < c.node_len) {
          const int m = min(c.node_len - c.idx, budget - nt);
          for (int j = lane; j < m; j += 64) {
            scr_tok[out + nt + j] = T.tokens[c.label + c.idx + j];
            scr_prob[out + nt + j] = prob;
          }
          for (int j = 0; j < m; ++j) score = __fadd_rn(score, prob);  // same order of f32 adds
          nt += m;
          c.idx += m;
        } else {
          if (c.node_best < 0) break;
          const int32_t parent_count = c.node_count;
          load_node(T, c.node_best, c);
          c.idx = 0;
          prob = __fmul_rn(prob, __fdiv_rn(static_cast<float>(c.node_count), static_cast<float>(parent_count)));
        }
      }
    }
  }
  if (lane == 0) {
    scr_score[item] = score;
    scr_n[item] = nt;
  }
}

// ---- tree-mode speculation (SuffixTree::_speculate_tree, suffix_tree.cc:226-274) -------------------------------------
// Same decomposition — one wavefront per (query, tree, suffix start) — and the same walk; the candidate is then grown by a
// priority queue on the estimated probability.  The reference's queue is std::priority_queue<HeapItem, vector, a.prob <
// b.prob>: WHICH of several equally probable entries is popped first is decided by libstdc++'s heap algorithms
// (std::push_heap = __push_heap, std::pop_heap = swap + __adjust_heap) and by the order in which the children were pushed
// (the unordered_map's iteration order).  Both are reproduced here: the kid lists of the image are in container order
// (suffix_host.hpp: serialize_kids), and heap_push / heap_pop below are those two algorithms statement by statement.
// Every lane of the wave runs the same scalar program; the heap (kHeapCap entries of {prob bits, node, idx, parent}) lives
// in a per-wave LDS region that all lanes read and write uniformly.  A node with more than kKidMax children, or a heap that
// would outgrow its region, gives the item up (scr_n = -1): the host then evaluates that query (suffix_host.hpp: grow_tree).
constexpr int kHeapCap = 256;

__device__ __forceinline__ bool heap_less(const volatile int4* h, int a, float vprob) {   // comp(first[a], value)
  return __int_as_float(h[a].x) < vprob;
}
// std::__push_heap(first, holeIndex, topIndex = 0, value)
__device__ __forceinline__ void heap_sift_up(volatile int4* h, int hole, int4 value) {
  const float vp = __int_as_float(value.x);
  int parent = (hole - 1) / 2;
  while (hole > 0 && heap_less(h, parent, vp)) {
    const int4 e = make_int4(h[parent].x, h[parent].y, h[parent].z, h[parent].w);
    h[hole].x = e.x; h[hole].y = e.y; h[hole].z = e.z; h[hole].w = e.w;
    hole = parent;
    parent = (hole - 1) / 2;
  }
  h[hole].x = value.x; h[hole].y = value.y; h[hole].z = value.z; h[hole].w = value.w;
}
// priority_queue::push: push_back, then std::push_heap
__device__ __forceinline__ void heap_push(volatile int4* h, int& n, int4 value) {
  heap_sift_up(h, n, value);
  ++n;
}
// priority_queue::pop after top(): std::pop_heap (value = last, last = first, __adjust_heap(first, 0, len - 1, value)), pop_back
__device__ __forceinline__ int4 heap_pop(volatile int4* h, int& n) {
  const int4 top = make_int4(h[0].x, h[0].y, h[0].z, h[0].w);
  const int len = n - 1;
  if (len > 0) {
    const int4 value = make_int4(h[len].x, h[len].y, h[len].z, h[len].w);
    int hole = 0, second = 0;
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      if (__int_as_float(h[second].x) < __int_as_float(h[second - 1].x)) --second;     // comp(first[second], first[second - 1])
      const int4 e = make_int4(h[second].x, h[second].y, h[second].z, h[second].w);
      h[hole].x = e.x; h[hole].y = e.y; h[hole].z = e.z; h[hole].w = e.w;
      hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      const int4 e = make_int4(h[second - 1].x, h[second - 1].y, h[second - 1].z, h[second - 1].w);
      h[hole].x = e.x; h[hole].y = e.y; h[hole].z = e.z; h[hole].w = e.w;
      hole = second - 1;
    }
    heap_sift_up(h, hole, value);     // (__push_heap with topIndex 0: the hole started at 0)
  }
  n = len;
  return top;
}

__global__ void __launch_bounds__(256)
suffix_tree_spec_kernel(const QueryRec* __restrict__ queries, const TreeDesc* __restrict__ trees,
                        const int32_t* __restrict__ patterns, int n_starts, int cap, int n_items,
                        float* __restrict__ scr_score, int32_t* __restrict__ scr_n, int32_t* __restrict__ scr_tok,
                        float* __restrict__ scr_prob, int32_t* __restrict__ scr_par) {
  __shared__ int4 heap_all[4][kHeapCap];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int item = uni(blockIdx.x * (blockDim.x >> 6) + wave);
  if (item >= n_items) return;
  volatile int4* heap = heap_all[wave];
  const int per_q = 2 * n_starts;
  const int qi = item / per_q;
  const int rem = item - qi * per_q;
  const int tr = rem / n_starts;
  const int s = rem - tr * n_starts;

  const QueryRec Q = queries[qi];
  const int tree_idx = uni(tr == 0 ? Q.prompt_tree : Q.global_tree);
  const int n = uni(Q.pattern_len);
  float score = 0.0f;
  int nt = 0;
  bool gave_up = false;

  if (tree_idx >= 0 && s < n) {
    const TreeDesc T = trees[tree_idx];
    const int32_t* pat = patterns + uni(Q.pattern_off);
    Cursor c;
    if (match_walk(T, pat, s, n, lane, c)) {
      const int budget = spec_budget_dev(n - s, Q, cap);
      const float min_prob = Q.min_prob;
      const int64_t out = static_cast<int64_t>(item) * cap;
      int hn = 0;
      heap_push(heap, hn, make_int4(__float_as_int(1.0f), c.node, c.idx, -1));
      while (nt < budget && hn > 0) {
        const int4 it = heap_pop(heap, hn);
        const float iprob = unif(__int_as_float(it.x));
        const int32_t inode = uni(it.y), iidx = uni(it.z), ipar = uni(it.w);
        Cursor k;
        load_node(T, inode, k);
        if (iidx < k.node_len) {
          if (lane == 0) {
            scr_tok[out + nt] = T.tokens[k.label + iidx];
            scr_prob[out + nt] = iprob;
            scr_par[out + nt] = ipar;
          }
          score = __fadd_rn(score, iprob);
          if (hn >= kHeapCap) {
            gave_up = true;
            break;
          }
          heap_push(heap, hn, make_int4(__float_as_int(iprob), inode, iidx + 1, nt));
          ++nt;
        } else {
          const int32_t* kb = T.kids + static_cast<int64_t>(inode) * kKidBlock;
          const int nk = uni(kb[0]);
          if (nk == kKidOverflow) {
            gave_up = true;
            break;
          }
          // lane j fetches child j and its count; the pushes then run in list order
          int32_t my_child = -1, my_count = 0;
          if (lane < nk) {
            my_child = kb[1 + lane];
            my_count = T.nodes[my_child].count;
          }
          for (int j = 0; j < nk; ++j) {
            const int32_t child = uni(__shfl(my_child, j));
            const int32_t ccount = uni(__shfl(my_count, j));
            // item.prob * child->count / static_cast<float>(item.node->count)   (float * int -> float, then / float)
            const float p = __fdiv_rn(__fmul_rn(iprob, static_cast<float>(ccount)), static_cast<float>(k.node_count));
            if (p >= min_prob) {
              if (hn >= kHeapCap) {
                gave_up = true;
                break;
              }
              heap_push(heap, hn, make_int4(__float_as_int(p), child, 0, ipar));
            }
          }
          if (gave_up) break;
        }
      }
    }
  }
  if (lane == 0) {
    scr_score[item] = score;
    scr_n[item] = gave_up ? -1 : nt;
  }
}

// One wavefront per query: reduce the per-start candidates of both trees and emit the winner.
__global__ void __launch_bounds__(64)
suffix_select_kernel(const QueryRec* __restrict__ queries, int n_starts, int cap,
                     const float* __restrict__ scr_score, const int32_t* __restrict__ scr_n,
                     const int32_t* __restrict__ scr_tok, const float* __restrict__ scr_prob,
                     int32_t* __restrict__ out_n, float* __restrict__ out_score,
                     int32_t* __restrict__ out_match, int32_t* __restrict__ out_tok,
                     float* __restrict__ out_prob, int32_t* __restrict__ done_counter, int32_t* done_flag,
                     int32_t epoch, const int32_t* __restrict__ scr_par, int32_t* __restrict__ out_par) {
  const int qi = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = queries[qi].pattern_len;
  float win_score = 0.0f;
  int win_item = -1, win_s = 0;
  bool gave_up = false;      // tree mode: an item of this query was given back to the host (scr_n = -1): so is the query
  for (int tr = 0; tr < 2; ++tr) {
    const int base = (qi * 2 + tr) * n_starts;
    float my = 0.0f;
    int my_s = 0x7fffffff;
    for (int s = lane; s < n_starts; s += 64) {
      if (scr_par != nullptr && scr_n[base + s] < 0) gave_up = true;
      const float v = scr_score[base + s];
      if (v > my) {  // strict: the earliest start (longest match) of equal scores stays
        my = v;
        my_s = s;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_xor(my, off);
      const int os = __shfl_xor(my_s, off);
      if (o > my || (o == my && os < my_s)) {
        my = o;
        my_s = os;
      }
    }
    // prompt tree first; the global tree replaces it only with a strictly higher score
    if (my > win_score) {
      win_score = my;
      win_item = base + my_s;
      win_s = my_s;
    }
  }
  int cnt = 0;
  gave_up = __any(gave_up);
  if (win_item >= 0 && !gave_up) {
    cnt = scr_n[win_item];
    for (int j = lane; j < cnt; j += 64) {
      out_tok[static_cast<int64_t>(qi) * cap + j] = scr_tok[static_cast<int64_t>(win_item) * cap + j];
      out_prob[static_cast<int64_t>(qi) * cap + j] = scr_prob[static_cast<int64_t>(win_item) * cap + j];
      if (out_par != nullptr) out_par[static_cast<int64_t>(qi) * cap + j] = scr_par[static_cast<int64_t>(win_item) * cap + j];
    }
  }
  if (gave_up) cnt = -1;
  if (lane == 0) {
    out_n[qi] = cnt;
    out_score[qi] = win_score;
    out_match[qi] = win_item >= 0 ? n - win_s : 0;
  }
  // zero-copy results (done_flag != nullptr): the outputs above went straight to pinned host memory; the workgroup that
  // finishes last raises the flag the host is polling.  Every workgroup's stores are pushed out system-wide before it is
  // counted, so the flag is never seen ahead of any of them.
  if (done_flag != nullptr) {
    __threadfence_system();
    __syncthreads();
    if (lane == 0) {
      const int prev = atomicAdd(done_counter, 1);
      if (prev == static_cast<int>(gridDim.x) - 1) {
        atomicExch(done_counter, 0);        // ready for the next call
        __threadfence_system();
        __hip_atomic_store(done_flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// device memory: power-of-two size classes, cached (prompt trees come and go every request)
// ------------------------------------------------------------------------------------------------
class DevPool {
 public:
  ~DevPool() {
    for (auto& kv : free_)
      for (void* p : kv.second) (void)hipFree(p);
  }
  static size_t round(size_t bytes) {
    size_t c = 1 << 16;
    while (c < bytes) c <<= 1;
    return c;
  }
  int get(size_t bytes, void** out) {
    const size_t c = round(bytes);
    auto& fl = free_[c];
    if (!fl.empty()) {
      *out = fl.back();
      fl.pop_back();
      return AIC_OK;
    }
    AIC_HIP_TRY(hipMalloc(out, c));
    return AIC_OK;
  }
  void put(void* p, size_t bytes) {
    if (p) free_[round(bytes)].push_back(p);
  }

 private:
  std::unordered_map<size_t, std::vector<void*>> free_;
};

struct TreeImage {
  NodeRec* nodes = nullptr;
  size_t nodes_cap = 0;  // records
  HashSlot* hash = nullptr;
  size_t hash_cap = 0;
  int32_t* tokens = nullptr;
  size_t tokens_cap = 0;
  int32_t* seq_base = nullptr;
  size_t seq_cap = 0;
  size_t seq_synced = 0;  // seq_base entries already mirrored
  int32_t* kids = nullptr;   // [kids_cap / kKidBlock][kKidBlock]: child lists (trees that were asked for tree mode)
  size_t kids_cap = 0;       // words
};

class Mirror;

}  // namespace aic

struct aic_suffix_tree {
  explicit aic_suffix_tree(int depth) : host(depth) {}
  ~aic_suffix_tree() { join_build(); }
  aic::HostTree host;
  aic::TreeImage img;
  aic::DevPool* pool = nullptr;           // who owns img's buffers
  std::unique_ptr<aic::Mirror> own;       // standalone trees mirror themselves
  std::unique_ptr<aic::DevPool> own_pool;
  // a prompt tree may still be under construction on a host thread (aic_sc_cache_prompt_async): whoever touches
  // `host` or `img` first joins it
  std::thread builder;
  void join_build() {
    if (builder.joinable()) builder.join();
  }
  // the device image is stale beyond what the dirty lists say (a mirror batch failed after this tree's deltas were
  // collected): the next add_tree uploads the whole tree
  bool force_full = false;
};

namespace aic {

// Collects the deltas of any number of trees (+ the query data) into one pinned blob, ships it with
// one H2D copy and applies it with one kernel launch.
class Mirror {
 public:
  ~Mirror() {
    if (pinned_) (void)hipHostFree(pinned_);
    if (dblob_) (void)hipFree(dblob_);
    if (dscr_) (void)hipFree(dscr_);
    if (pin_out_) (void)hipHostFree(pin_out_);
    if (done_counter_) (void)hipFree(done_counter_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
  }

  void begin() {
    words_ = 0;
    jobs_.clear();
    descs_.clear();
    max_job_words_ = 0;
    batch_trees_.clear();
  }
  // a step of the current batch failed: the device images of its trees may lack deltas whose dirty marks are gone
  int fail_batch(int rc) {
    for (aic_suffix_tree* t : batch_trees_) t->force_full = true;
    batch_trees_.clear();
    return rc;
  }

  // reserves `n` int32 words (16-byte aligned start) in the pinned blob, returns the word offset
  int reserve(size_t n, int64_t* off) {
    words_ = (words_ + 3) & ~static_cast<size_t>(3);
    const size_t need = (words_ + n + 64) * 4;
    if (need > pinned_cap_) {
      size_t cap = pinned_cap_ ? pinned_cap_ : (1 << 20);
      while (cap < need) cap *= 2;
      void* np = nullptr;
      AIC_HIP_TRY(hipHostMalloc(&np, cap, hipHostMallocDefault));
      if (pinned_) {
        std::memcpy(np, pinned_, words_ * 4);
        (void)hipHostFree(pinned_);
      }
      pinned_ = static_cast<int32_t*>(np);
      pinned_cap_ = cap;
    }
    *off = static_cast<int64_t>(words_);
    words_ += n;
    return AIC_OK;
  }
  int32_t* at(int64_t off) { return pinned_ + off; }

  int add_job(void* dst, int64_t src_off, int64_t idx_off, int32_t dst_first, int32_t n, int32_t rec_words) {
    if (n <= 0) return AIC_OK;
    ApplyJob j;
    j.dst = static_cast<int32_t*>(dst);
    j.src_off = src_off;
    j.idx_off = idx_off;
    j.dst_first = dst_first;
    j.n = n;
    j.rec_words = rec_words;
    j.pad = 0;
    jobs_.push_back(j);
    max_job_words_ = std::max<int64_t>(max_job_words_, static_cast<int64_t>(n) * rec_words);
    return AIC_OK;
  }

  template <typename T>
  int grow(DevPool& pool, T** buf, size_t* cap, size_t need, bool* moved) {
    *moved = false;
    if (need <= *cap && *buf) return AIC_OK;
    size_t c = *cap ? *cap : 1024;
    while (c < need) c *= 2;
    void* np = nullptr;
    int rc = pool.get(c * sizeof(T), &np);
    if (rc != AIC_OK) return rc;
    // round the capacity up to what the pool really handed out
    c = DevPool::round(c * sizeof(T)) / sizeof(T);
    if (*buf) pool.put(*buf, *cap * sizeof(T));
    *buf = static_cast<T*>(np);
    *cap = c;
    *moved = true;
    return AIC_OK;
  }

  // Queues everything tree `t` changed since its last mirror; returns its TreeDesc index.
  int add_tree(aic_suffix_tree* t, DevPool& pool, int* desc_index) {
    t->join_build();
    HostTree& H = t->host;
    TreeImage& I = t->img;
    t->pool = &pool;
    bool moved = false;
    int rc;
    int64_t off, ioff;
    // from here on the dirty information of `t` is consumed before the device has applied it: a failure anywhere in
    // this batch (allocation, copy, launch) marks every tree of the batch for a full upload (fail_batch())
    batch_trees_.push_back(t);
    const bool full = t->force_full;
    t->force_full = false;

    // nodes
    const size_t n_nodes = H.recs().size();
    if ((rc = grow(pool, &I.nodes, &I.nodes_cap, n_nodes, &moved)) != AIC_OK) return rc;
    if (moved || full) {
      if ((rc = reserve(n_nodes * 8, &off)) != AIC_OK) return rc;
      std::memcpy(at(off), H.recs().data(), n_nodes * sizeof(NodeRec));
      add_job(I.nodes, off, -1, 0, static_cast<int32_t>(n_nodes), 8);
    } else if (!H.dirty_nodes().empty()) {
      const size_t d = H.dirty_nodes().size();
      if ((rc = reserve(d * 8, &off)) != AIC_OK) return rc;
      if ((rc = reserve(d, &ioff)) != AIC_OK) return rc;
      int32_t* rec = at(off);
      int32_t* idx = at(ioff);
      for (size_t k = 0; k < d; ++k) {
        const int32_t ni = H.dirty_nodes()[k];
        std::memcpy(rec + k * 8, &H.recs()[ni], sizeof(NodeRec));
        idx[k] = ni;
      }
      add_job(I.nodes, off, ioff, 0, static_cast<int32_t>(d), 8);
    }

    // hash table
    const size_t n_slots = H.slots().size();
    if ((rc = grow(pool, &I.hash, &I.hash_cap, n_slots, &moved)) != AIC_OK) return rc;
    if (moved || full || H.hash_rebuilt()) {
      if ((rc = reserve(n_slots * 4, &off)) != AIC_OK) return rc;
      std::memcpy(at(off), H.slots().data(), n_slots * sizeof(HashSlot));
      add_job(I.hash, off, -1, 0, static_cast<int32_t>(n_slots), 4);
    } else if (!H.dirty_slots().empty()) {
      const size_t d = H.dirty_slots().size();
      if ((rc = reserve(d * 4, &off)) != AIC_OK) return rc;
      if ((rc = reserve(d, &ioff)) != AIC_OK) return rc;
      int32_t* rec = at(off);
      int32_t* idx = at(ioff);
      for (size_t k = 0; k < d; ++k) {
        const int32_t si = H.dirty_slots()[k];
        std::memcpy(rec + k * 4, &H.slots()[si], sizeof(HashSlot));
        idx[k] = si;
      }
      add_job(I.hash, off, ioff, 0, static_cast<int32_t>(d), 4);
    }

    // token regions: make every sequence fit, then upload what is new
    auto& seqs = H.seqs();
    bool any_moved = false;
    for (auto& s : seqs) any_moved |= H.fit_region(s, static_cast<int32_t>(s.toks.size()));
    if ((rc = grow(pool, &I.tokens, &I.tokens_cap, static_cast<size_t>(H.pool_end()), &moved)) != AIC_OK) return rc;
    if (moved || full)
      for (auto& s : seqs) s.synced = 0;
    for (auto& s : seqs) {
      const int32_t have = static_cast<int32_t>(s.toks.size());
      if (have > s.synced) {
        const int32_t cnt = have - s.synced;
        if ((rc = reserve(cnt, &off)) != AIC_OK) return rc;
        std::memcpy(at(off), s.toks.data() + s.synced, static_cast<size_t>(cnt) * 4);
        add_job(I.tokens, off, -1, s.base + s.synced, cnt, 1);
        s.synced = have;
      }
    }
    // sequence table: {region base, current length} per sequence slot (open leaves read the length)
    if ((rc = grow(pool, &I.seq_base, &I.seq_cap, 2 * std::max<size_t>(seqs.size(), 1), &moved)) != AIC_OK) return rc;
    if (moved || full || any_moved || I.seq_synced != seqs.size()) {
      if ((rc = reserve(2 * std::max<size_t>(seqs.size(), 1), &off)) != AIC_OK) return rc;
      for (size_t k = 0; k < seqs.size(); ++k) {
        at(off)[2 * k] = seqs[k].base;
        at(off)[2 * k + 1] = static_cast<int32_t>(seqs[k].toks.size());
      }
      add_job(I.seq_base, off, -1, 0, static_cast<int32_t>(seqs.size()), 2);
      I.seq_synced = seqs.size();
    } else if (!H.dirty_seqs().empty()) {
      const size_t d = H.dirty_seqs().size();
      if ((rc = reserve(d * 2, &off)) != AIC_OK) return rc;
      if ((rc = reserve(d, &ioff)) != AIC_OK) return rc;
      for (size_t k = 0; k < d; ++k) {
        const int32_t si = H.dirty_seqs()[k];
        at(off)[2 * k] = seqs[si].base;
        at(off)[2 * k + 1] = static_cast<int32_t>(seqs[si].toks.size());
        at(ioff)[k] = si;
      }
      add_job(I.seq_base, off, ioff, 0, static_cast<int32_t>(d), 2);
    }
    // child lists in container order (tree-mode speculation): only for trees that were ever asked for it
    if (H.track_kids()) {
      const bool first_time = I.kids == nullptr;
      if ((rc = grow(pool, &I.kids, &I.kids_cap, n_nodes * kKidBlock, &moved)) != AIC_OK) return rc;
      if (moved || full || first_time) {
        if ((rc = reserve(n_nodes * kKidBlock, &off)) != AIC_OK) return rc;
        for (size_t k = 0; k < n_nodes; ++k) H.serialize_kids(static_cast<int32_t>(k), at(off) + k * kKidBlock);
        add_job(I.kids, off, -1, 0, static_cast<int32_t>(n_nodes), kKidBlock);
      } else if (!H.dirty_kids().empty()) {
        const size_t d = H.dirty_kids().size();
        if ((rc = reserve(d * kKidBlock, &off)) != AIC_OK) return rc;
        if ((rc = reserve(d, &ioff)) != AIC_OK) return rc;
        for (size_t k = 0; k < d; ++k) {
          const int32_t ni = H.dirty_kids()[k];
          H.serialize_kids(ni, at(off) + k * kKidBlock);
          at(ioff)[k] = ni;
        }
        add_job(I.kids, off, ioff, 0, static_cast<int32_t>(d), kKidBlock);
      }
    }
    H.clear_dirty();

    TreeDesc d;
    d.nodes = I.nodes;
    d.hash = I.hash;
    d.tokens = I.tokens;
    d.seq_base = I.seq_base;
    d.hash_mask = H.hash_mask();
    d.n_nodes = static_cast<int32_t>(n_nodes);
    d.kids = I.kids;
    *desc_index = static_cast<int>(descs_.size());
    descs_.push_back(d);
    return AIC_OK;
  }

  // Runs the batch: blob H2D, delta apply, match + select, results D2H, stream sync.
  // `out_parents` != nullptr: TREE mode (suffix_tree_spec_kernel; every tree of the batch must carry kid lists) — out_n[i] = -1
  // marks a query the device gave back (a node with more than kKidMax children, or a heap beyond kHeapCap)
  int run(const std::vector<QueryRec>& queries, const std::vector<int32_t>& pattern_pool, int n_starts, int cap,
          int32_t* out_tokens, float* out_probs, int32_t* out_n, float* out_score, int32_t* out_match,
          hipStream_t stream, int32_t* out_parents = nullptr) {
    const int rc = run_batch(queries, pattern_pool, n_starts, cap, out_tokens, out_probs, out_n, out_score, out_match, stream,
                             out_parents);
    if (rc != AIC_OK) return fail_batch(rc);
    batch_trees_.clear();   // the deltas are on the device (stream synchronised): the sync state stands
    return AIC_OK;
  }

  int run_batch(const std::vector<QueryRec>& queries, const std::vector<int32_t>& pattern_pool, int n_starts, int cap,
                int32_t* out_tokens, float* out_probs, int32_t* out_n, float* out_score, int32_t* out_match,
                hipStream_t stream, int32_t* out_parents) {
    const int nq = static_cast<int>(queries.size());
    const bool tree_mode = out_parents != nullptr;
    int rc;
    int64_t q_off, p_off, d_off, j_off;
    // fix pattern offsets relative to the blob once the pool position is known
    if ((rc = reserve(pattern_pool.size() + 1, &p_off)) != AIC_OK) return rc;
    std::memcpy(at(p_off), pattern_pool.data(), pattern_pool.size() * 4);
    if ((rc = reserve(static_cast<size_t>(nq) * 8, &q_off)) != AIC_OK) return rc;
    std::memcpy(at(q_off), queries.data(), static_cast<size_t>(nq) * sizeof(QueryRec));
    const size_t desc_words = descs_.size() * sizeof(TreeDesc) / 4;
    if ((rc = reserve(desc_words + 4, &d_off)) != AIC_OK) return rc;
    std::memcpy(at(d_off), descs_.data(), descs_.size() * sizeof(TreeDesc));
    const size_t job_words = (jobs_.size() + 1) * sizeof(ApplyJob) / 4;      // + the query-section job of the zero-copy path
    if ((rc = reserve(job_words + 4, &j_off)) != AIC_OK) return rc;
    std::memcpy(at(j_off), jobs_.data(), jobs_.size() * sizeof(ApplyJob));

    const size_t blob_bytes = words_ * 4;
    if (blob_bytes > dblob_cap_) {
      if (dblob_) AIC_HIP_TRY(hipFree(dblob_));
      dblob_cap_ = std::max<size_t>(blob_bytes * 2, 1 << 20);
      AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dblob_), dblob_cap_));
    }
    // scratch: per item {score, n} + cap tokens + cap probs; outputs per query
    const size_t n_items = static_cast<size_t>(nq) * 2 * n_starts;
    const size_t scr_words = n_items * 2 + n_items * cap * (tree_mode ? 3 : 2);
    const size_t out_words = static_cast<size_t>(nq) * (3 + (tree_mode ? 3 : 2) * static_cast<size_t>(cap));
    const size_t scr_bytes = (scr_words + out_words + 16) * 4;
    if (scr_bytes > dscr_cap_) {
      if (dscr_) AIC_HIP_TRY(hipFree(dscr_));
      dscr_cap_ = scr_bytes * 2;
      AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dscr_), dscr_cap_));
    }
    if ((out_words + 16) * 4 > pin_out_cap_) {
      if (pin_out_) AIC_HIP_TRY(hipHostFree(pin_out_));
      pin_out_cap_ = (out_words + 16) * 8;
      AIC_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pin_out_), pin_out_cap_, hipHostMallocDefault));
      std::memset(pin_out_, 0, pin_out_cap_);
    }
    if (!ev0_) {
      AIC_HIP_TRY(hipEventCreate(&ev0_));
      AIC_HIP_TRY(hipEventCreate(&ev1_));
    }
    if (!done_counter_) {
      AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&done_counter_), 64));
      AIC_HIP_TRY(hipMemset(done_counter_, 0, 64));
    }
    // ZERO-COPY round trip, built and measured in r03, NOT the default (AIC_SUFFIX_ZEROCOPY=1 selects it): pinned host memory is
    // mapped into the device's address space, so
    //   * the apply kernel reads its job list and the delta records straight from the pinned blob, and one more job brings
    //     the query section (patterns, queries, tree descriptors — read by thousands of waves) into device memory;
    //   * the select kernel writes the winners straight into the pinned result block and its last workgroup raises a flag
    //     there, which this thread polls.
    // It takes two copy commands, their dependency barriers and the stream synchronisation out of the round trip — and gives
    // the time back inside the kernels, which now wait on the host link: rocprofv3 on the bench (profiles/r03_kernel_stats.csv
    // against r02's): mirror_apply 17.7 -> 30.1 us, suffix_select 5.4 -> 13.7 us per launch; the round trip as the host sees it
    // 0.21 -> 0.22 ms per lane step on one GPU, 0.194 -> 0.204 ms at rehearsed SP = 8 (same-box A/B).  The staged form stays.
    static const bool staged = [] {
      const char* e = std::getenv("AIC_SUFFIX_ZEROCOPY");
      return !(e && e[0] == '1');
    }();
    const int32_t* blob_dev = dblob_;          // where the kernels read the blob from
    int32_t* pin_out_dev = nullptr;
    // a large blob (whole trees: the first step after prompts were cached, hundreds of megabytes) goes by the copy engine
    const bool blob_direct = !staged && blob_bytes <= (static_cast<size_t>(4) << 20);
    if (!staged) {
      void* p = nullptr;
      AIC_HIP_TRY(hipHostGetDevicePointer(&p, pin_out_, 0));
      pin_out_dev = static_cast<int32_t*>(p);
    }
    if (blob_direct) {
      void* p = nullptr;
      AIC_HIP_TRY(hipHostGetDevicePointer(&p, pinned_, 0));
      blob_dev = static_cast<const int32_t*>(p);
      // the query section [p_off, j_off) -> the same offsets of the device blob (one contiguous job)
      ApplyJob q;
      q.dst = dblob_ + p_off;
      q.src_off = p_off;
      q.idx_off = -1;
      q.dst_first = 0;
      q.n = static_cast<int32_t>(j_off - p_off);
      q.rec_words = 1;
      q.pad = 0;
      jobs_.push_back(q);
      max_job_words_ = std::max<int64_t>(max_job_words_, q.n);
      // the job list was already written behind the descriptors: re-write it with the extra job (room was reserved)
      std::memcpy(at(j_off), jobs_.data(), jobs_.size() * sizeof(ApplyJob));
    } else {
      AIC_HIP_TRY(hipMemcpyAsync(dblob_, pinned_, blob_bytes, hipMemcpyHostToDevice, stream));
    }
    mirrored_bytes_ = static_cast<int64_t>(blob_bytes);
    if (!jobs_.empty()) {
      const int bx = static_cast<int>(std::min<int64_t>(std::max<int64_t>(max_job_words_ / 1024, 1), 256));
      dim3 grid(bx, static_cast<unsigned>(jobs_.size()));
      hipLaunchKernelGGL(mirror_apply_kernel, grid, dim3(256), 0, stream,
                         reinterpret_cast<const ApplyJob*>(blob_dev + j_off), blob_dev);
      if ((rc = launch_status("mirror_apply_kernel")) != AIC_OK) return rc;
    }
    float* scr_score = reinterpret_cast<float*>(dscr_);
    int32_t* scr_n = dscr_ + n_items;
    int32_t* scr_tok = dscr_ + 2 * n_items;
    float* scr_prob = reinterpret_cast<float*>(dscr_ + 2 * n_items + n_items * cap);
    int32_t* scr_par = tree_mode ? dscr_ + 2 * n_items + 2 * n_items * cap : nullptr;
    int32_t* d_out = staged ? dscr_ + scr_words : pin_out_dev;
    int32_t* o_n = d_out;
    float* o_score = reinterpret_cast<float*>(d_out + nq);
    int32_t* o_match = d_out + 2 * nq;
    int32_t* o_tok = d_out + 3 * nq;
    float* o_prob = reinterpret_cast<float*>(d_out + 3 * nq + static_cast<size_t>(nq) * cap);
    int32_t* o_par = tree_mode ? d_out + 3 * nq + 2 * static_cast<size_t>(nq) * cap : nullptr;
    int32_t* flag_host = pin_out_ + pin_out_cap_ / 4 - 16;       // last 64 bytes of the result block
    int32_t* flag_dev = staged ? nullptr : pin_out_dev + pin_out_cap_ / 4 - 16;
    epoch_ = epoch_ == 0x7ffffff0 ? 1 : epoch_ + 1;

    AIC_HIP_TRY(hipEventRecord(ev0_, stream));
    const int waves_per_block = 4;
    const int blocks = static_cast<int>((n_items + waves_per_block - 1) / waves_per_block);
    if (tree_mode)
      hipLaunchKernelGGL(suffix_tree_spec_kernel, dim3(blocks), dim3(64 * waves_per_block), 0, stream,
                         reinterpret_cast<const QueryRec*>(dblob_ + q_off),
                         reinterpret_cast<const TreeDesc*>(dblob_ + d_off), dblob_ + p_off, n_starts, cap,
                         static_cast<int>(n_items), scr_score, scr_n, scr_tok, scr_prob, scr_par);
    else
      hipLaunchKernelGGL(suffix_match_kernel, dim3(blocks), dim3(64 * waves_per_block), 0, stream,
                         reinterpret_cast<const QueryRec*>(dblob_ + q_off),
                         reinterpret_cast<const TreeDesc*>(dblob_ + d_off), dblob_ + p_off, n_starts, cap,
                         static_cast<int>(n_items), scr_score, scr_n, scr_tok, scr_prob);
    if ((rc = launch_status("suffix_match_kernel")) != AIC_OK) return rc;
    hipLaunchKernelGGL(suffix_select_kernel, dim3(nq), dim3(64), 0, stream,
                       reinterpret_cast<const QueryRec*>(dblob_ + q_off), n_starts, cap, scr_score, scr_n,
                       scr_tok, scr_prob, o_n, o_score, o_match, o_tok, o_prob, done_counter_, flag_dev, epoch_,
                       static_cast<const int32_t*>(scr_par), o_par);
    if ((rc = launch_status("suffix_select_kernel")) != AIC_OK) return rc;
    AIC_HIP_TRY(hipEventRecord(ev1_, stream));
    if (staged) {
      AIC_HIP_TRY(hipMemcpyAsync(pin_out_, d_out, out_words * 4, hipMemcpyDeviceToHost, stream));
      AIC_HIP_TRY(hipStreamSynchronize(stream));
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) match_us_ = ms * 1000.0f;
    } else {
      // poll the flag; a launch this size takes tens of microseconds — after 50 ms something else is wrong and the stream
      // is asked instead (an error there is reported; a completed stream without the flag is one too)
      timing_pending_ = true;
      const auto t0 = std::chrono::steady_clock::now();
      bool seen = false;
      for (uint32_t spin = 0;; ++spin) {
        if (__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) == epoch_) {
          seen = true;
          break;
        }
        __builtin_ia32_pause();
        if ((spin & 1023) == 1023 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 50.0)
          break;
      }
      if (!seen) {
        AIC_HIP_TRY(hipStreamSynchronize(stream));
        AIC_REQUIRE(__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) == epoch_, "suffix select launch finished without raising its flag");
      }
    }

    const int32_t* h_n = pin_out_;
    const float* h_score = reinterpret_cast<const float*>(pin_out_ + nq);
    const int32_t* h_match = pin_out_ + 2 * nq;
    const int32_t* h_tok = pin_out_ + 3 * nq;
    const float* h_prob = reinterpret_cast<const float*>(pin_out_ + 3 * nq + static_cast<size_t>(nq) * cap);
    const int32_t* h_par = pin_out_ + 3 * nq + 2 * static_cast<size_t>(nq) * cap;
    for (int i = 0; i < nq; ++i) {
      out_n[i] = h_n[i];
      out_score[i] = h_score[i];
      out_match[i] = h_match[i];
      const size_t cnt = h_n[i] > 0 ? static_cast<size_t>(h_n[i]) : 0;
      std::memcpy(out_tokens + static_cast<size_t>(i) * cap, h_tok + static_cast<size_t>(i) * cap, cnt * 4);
      std::memcpy(out_probs + static_cast<size_t>(i) * cap, h_prob + static_cast<size_t>(i) * cap, cnt * 4);
      if (tree_mode) std::memcpy(out_parents + static_cast<size_t>(i) * cap, h_par + static_cast<size_t>(i) * cap, cnt * 4);
    }
    return AIC_OK;
  }

  float match_us() {
    if (timing_pending_) {         // the zero-copy path does not wait for the events inside the call
      float ms = 0.0f;
      if (hipEventSynchronize(ev1_) == hipSuccess && hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) match_us_ = ms * 1000.0f;
      timing_pending_ = false;
    }
    return match_us_;
  }
  int64_t mirrored_bytes() const { return mirrored_bytes_; }

 private:
  int32_t* pinned_ = nullptr;
  size_t pinned_cap_ = 0;
  size_t words_ = 0;
  std::vector<ApplyJob> jobs_;
  std::vector<TreeDesc> descs_;
  int64_t max_job_words_ = 0;
  std::vector<aic_suffix_tree*> batch_trees_;
  int32_t* dblob_ = nullptr;
  size_t dblob_cap_ = 0;
  int32_t* dscr_ = nullptr;
  size_t dscr_cap_ = 0;
  int32_t* pin_out_ = nullptr;
  size_t pin_out_cap_ = 0;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  int32_t* done_counter_ = nullptr;   // device: workgroups of the select launch that have finished
  int32_t epoch_ = 0;                 // value the flag (last word of pin_out_) takes when the current call's results are in
  bool timing_pending_ = false;
  float match_us_ = 0.0f;
  int64_t mirrored_bytes_ = 0;
};

static void release_image(aic_suffix_tree* t) {
  if (!t->pool) return;
  TreeImage& I = t->img;
  t->pool->put(I.nodes, I.nodes_cap * sizeof(NodeRec));
  t->pool->put(I.hash, I.hash_cap * sizeof(HashSlot));
  t->pool->put(I.tokens, I.tokens_cap * sizeof(int32_t));
  t->pool->put(I.seq_base, I.seq_cap * sizeof(int32_t));
  t->pool->put(I.kids, I.kids_cap * sizeof(int32_t));
  I = TreeImage();
}

}  // namespace aic

struct aic_suffix_cache {
  explicit aic_suffix_cache(int depth) : max_depth(depth), global(new aic_suffix_tree(depth)) {}
  float last_build_us = 0.0f, last_device_us = 0.0f;   // aic_sc_last_timing
  // evicted prompt trees are destroyed by a detached-from-the-engine host thread; at most one is in flight
  std::thread reaper;
  void retire(std::unique_ptr<aic_suffix_tree> t) {
    if (reaper.joinable()) reaper.join();
    aic_suffix_tree* raw = t.release();
    reaper = std::thread([raw]() { delete raw; });
  }
  ~aic_suffix_cache() {
    if (reaper.joinable()) reaper.join();
    // trees hand their buffers back to `pool` before the pool frees them
    for (auto& kv : prompts) {
      kv.second->join_build();
      aic::release_image(kv.second.get());
    }
    aic::release_image(global.get());
  }
  int max_depth;
  aic::DevPool pool;
  aic::Mirror mirror;
  std::unique_ptr<aic_suffix_tree> global;
  std::unordered_map<int64_t, std::unique_ptr<aic_suffix_tree>> prompts;
  std::unordered_map<int64_t, int32_t> seq_of;
};

using namespace aic;

static int64_t g_tree_mode_device = 0, g_tree_mode_fallbacks = 0;
static bool g_tree_mode_on_host = false;

extern "C" {

// ---- tree-level API (pybind.cc:24-38) -----------------------------------------------------------
aic_suffix_tree* aic_st_create(int max_depth) {
  if (max_depth <= 0) {
    set_error("max_depth must be positive");
    return nullptr;
  }
  return new aic_suffix_tree(max_depth);
}
void aic_st_destroy(aic_suffix_tree* t) {
  if (!t) return;
  if (t->own_pool) release_image(t);
  delete t;
}
int aic_st_num_seqs(const aic_suffix_tree* t) { return t ? t->host.num_seqs() : AIC_ERR_INVALID; }
int aic_st_append(aic_suffix_tree* t, int seq_id, int token) {
  AIC_REQUIRE(t, "null tree");
  t->host.append(seq_id, token);
  return AIC_OK;
}
int aic_st_extend(aic_suffix_tree* t, int seq_id, const int32_t* tokens, int n) {
  AIC_REQUIRE(t && (tokens || n == 0) && n >= 0, "bad arguments to aic_st_extend");
  for (int i = 0; i < n; ++i) t->host.append(seq_id, tokens[i]);
  return AIC_OK;
}

int aic_st_speculate(aic_suffix_tree* t, const int32_t* pattern, int n, int max_spec_tokens, float factor,
                     float offset, float min_prob, int use_tree_spec, int32_t* out_tokens, int32_t* out_parents,
                     float* out_probs, int cap, float* out_score, int32_t* out_match_len, void* stream) {
  AIC_REQUIRE(t && pattern && n > 0 && cap >= 0 && out_score && out_match_len, "bad arguments to aic_st_speculate");
  const int depth = t->host.max_depth();
  if (n > depth) {  // only the last max_depth tokens can match (suffix_tree.cc:142)
    pattern += n - depth;
    n = depth;
  }
  // tree mode without a device (CPU tests of the host trees), or when the device gives the query back: the host's own
  // priority-queue expansion (suffix_host.hpp: grow_tree)
  auto host_tree_mode = [&]() -> int {
    HostCandidate c = t->host.speculate_tree(pattern, n, max_spec_tokens, factor, offset, min_prob);
    const int m = std::min<int>(static_cast<int>(c.token_ids.size()), cap);
    for (int i = 0; i < m; ++i) {
      out_tokens[i] = c.token_ids[i];
      if (out_parents) out_parents[i] = c.parents[i];
      out_probs[i] = c.probs[i];
    }
    *out_score = c.score;
    *out_match_len = c.match_len;
    return m;
  };
  if (use_tree_spec && (aic_device_count() <= 0 || g_tree_mode_on_host)) return host_tree_mode();
  AIC_NEED_DEVICE();
  if (!t->own) {
    t->own.reset(new Mirror());
    t->own_pool.reset(new DevPool());
  }
  DevPool& pool = t->pool ? *t->pool : *t->own_pool;
  Mirror& mir = *t->own;
  if (use_tree_spec) {
    t->join_build();
    t->host.enable_kid_tracking();
  }
  mir.begin();
  int di = -1;
  int rc = mir.add_tree(t, pool, &di);
  if (rc != AIC_OK) return mir.fail_batch(rc);
  QueryRec q;
  q.pattern_off = 0;
  q.pattern_len = n;
  q.max_spec = max_spec_tokens;
  q.prompt_tree = -1;
  q.global_tree = di;
  q.factor = factor;
  q.offset = offset;
  q.min_prob = min_prob;
  std::vector<QueryRec> qs(1, q);
  std::vector<int32_t> pool_words(pattern, pattern + n);
  // a path candidate cannot be longer than max_depth; a tree candidate has branches and is bounded by max_spec_tokens only
  const int kcap = use_tree_spec ? std::max(max_spec_tokens, 1) : std::max(std::min(std::max(max_spec_tokens, 0), depth), 1);
  std::vector<int32_t> toks(kcap), pars(kcap);
  std::vector<float> probs(kcap);
  int32_t cnt = 0, mlen = 0;
  float score = 0.0f;
  rc = mir.run(qs, pool_words, n, kcap, toks.data(), probs.data(), &cnt, &score, &mlen,
               static_cast<hipStream_t>(stream), use_tree_spec ? pars.data() : nullptr);
  if (rc != AIC_OK) return rc;
  if (use_tree_spec) {
    if (cnt < 0) {                       // given back by the device
      ++g_tree_mode_fallbacks;
      return host_tree_mode();
    }
    ++g_tree_mode_device;
  }
  const int m = std::min<int>(cnt, cap);
  for (int i = 0; i < m; ++i) {
    out_tokens[i] = toks[i];
    if (out_parents) out_parents[i] = use_tree_spec ? pars[i] : i - 1;
    out_probs[i] = probs[i];
  }
  *out_score = score;
  *out_match_len = mlen;
  return m;
}

// tree-mode bookkeeping / switches (tests): queries answered by the device, queries the device gave back to the host;
// aic_debug_tree_mode_on_host(1) sends every tree-mode query to the host trees (the A/B reference)
int aic_debug_tree_mode_stats(int64_t* on_device, int64_t* given_back) {
  if (on_device) *on_device = g_tree_mode_device;
  if (given_back) *given_back = g_tree_mode_fallbacks;
  return AIC_OK;
}
int aic_debug_tree_mode_on_host(int on) {
  g_tree_mode_on_host = on != 0;
  return AIC_OK;
}

int aic_st_export(aic_suffix_tree* t, int32_t* n_nodes, int32_t* n_slots, int32_t* n_tokens, int32_t* n_seq_slots,
                  int32_t* nodes, int32_t* hash, int32_t* tokens, int32_t* seq_base, int32_t* seq_ids) {
  AIC_REQUIRE(t && n_nodes && n_slots && n_tokens && n_seq_slots, "bad arguments to aic_st_export");
  HostTree& H = t->host;
  for (auto& s : H.seqs()) H.fit_region(s, static_cast<int32_t>(s.toks.size()));
  *n_nodes = static_cast<int32_t>(H.recs().size());
  *n_slots = static_cast<int32_t>(H.slots().size());
  *n_tokens = H.pool_end();
  *n_seq_slots = static_cast<int32_t>(H.seqs().size());
  if (nodes) {
    std::memcpy(nodes, H.recs().data(), H.recs().size() * sizeof(NodeRec));
    // the exported image carries concrete edge lengths (open leaves resolved against their sequence's length)
    for (size_t i = 0; i < H.recs().size(); ++i)
      if (H.recs()[i].alive && H.recs()[i].length == kOpenLength) nodes[i * 8 + 4] = H.len_of(static_cast<int32_t>(i));
  }
  if (hash) std::memcpy(hash, H.slots().data(), H.slots().size() * sizeof(HashSlot));
  if (tokens) {
    std::memset(tokens, 0xff, static_cast<size_t>(H.pool_end()) * 4);
    for (const auto& s : H.seqs()) std::memcpy(tokens + s.base, s.toks.data(), s.toks.size() * 4);
  }
  for (size_t k = 0; k < H.seqs().size(); ++k) {
    if (seq_base) seq_base[k] = H.seqs()[k].base;
    if (seq_ids) seq_ids[k] = H.seqs()[k].id;
  }
  return AIC_OK;
}

int aic_st_selfcheck(aic_suffix_tree* t) {
  AIC_REQUIRE(t, "null tree");
  return t->host.selfcheck();
}

// ---- cache-level API (suffix_cache.py:57-222) ---------------------------------------------------
aic_suffix_cache* aic_sc_create(int max_depth) {
  if (max_depth <= 0) {
    set_error("max_depth must be positive");
    return nullptr;
  }
  return new aic_suffix_cache(max_depth);
}
void aic_sc_destroy(aic_suffix_cache* c) { delete c; }
int aic_sc_has_prompt(const aic_suffix_cache* c, int64_t req) {
  return c && c->prompts.count(req) ? 1 : 0;
}
int aic_sc_cache_prompt(aic_suffix_cache* c, int64_t req, const int32_t* tokens, int n) {
  AIC_REQUIRE(c && (tokens || n == 0) && n >= 0, "bad arguments to aic_sc_cache_prompt");
  if (c->prompts.count(req)) {
    set_error("prompt already exists for request %lld", static_cast<long long>(req));
    return AIC_ERR_EXISTS;
  }
  std::unique_ptr<aic_suffix_tree> t(new aic_suffix_tree(c->max_depth));
  for (int i = 0; i < n; ++i) t->host.append(0, tokens[i]);
  c->prompts.emplace(req, std::move(t));
  return AIC_OK;
}
int aic_sc_cache_prompt_async(aic_suffix_cache* c, int64_t req, const int32_t* tokens, int n, const int32_t* response,
                              int n_response) {
  AIC_REQUIRE(c && (tokens || n == 0) && n >= 0 && (response || n_response == 0) && n_response >= 0,
              "bad arguments to aic_sc_cache_prompt_async");
  if (c->prompts.count(req)) {
    set_error("prompt already exists for request %lld", static_cast<long long>(req));
    return AIC_ERR_EXISTS;
  }
  // the global tree is this thread's: the response tokens go in now, in call order (suffix_cache.py:141-149)
  if (n_response > 0) {
    auto it = c->seq_of.find(req);
    if (it == c->seq_of.end()) it = c->seq_of.emplace(req, static_cast<int32_t>(c->seq_of.size())).first;
    for (int i = 0; i < n_response; ++i) c->global->host.append(it->second, response[i]);
  }
  // the prompt tree is independent of everything else until somebody asks for it: build it on a host thread
  std::unique_ptr<aic_suffix_tree> t(new aic_suffix_tree(c->max_depth));
  std::vector<int32_t> all(static_cast<size_t>(n) + n_response);
  if (n) std::memcpy(all.data(), tokens, static_cast<size_t>(n) * 4);
  if (n_response) std::memcpy(all.data() + n, response, static_cast<size_t>(n_response) * 4);
  aic_suffix_tree* raw = t.get();
  t->builder = std::thread([raw, toks = std::move(all)]() {
    for (int32_t tok : toks) raw->host.append(0, tok);
  });
  c->prompts.emplace(req, std::move(t));
  return AIC_OK;
}
int aic_sc_cache_prompts(aic_suffix_cache* c, int n_req, const int64_t* reqs, const int32_t* tokens,
                         const int32_t* lens, int n_threads) {
  AIC_REQUIRE(c && reqs && lens && n_req >= 0, "bad arguments to aic_sc_cache_prompts");
  std::vector<int64_t> offs(n_req + 1, 0);
  for (int i = 0; i < n_req; ++i) {
    AIC_REQUIRE(lens[i] >= 0, "negative prompt length");
    if (c->prompts.count(reqs[i])) {
      set_error("prompt already exists for request %lld", static_cast<long long>(reqs[i]));
      return AIC_ERR_EXISTS;
    }
    for (int j = 0; j < i; ++j) AIC_REQUIRE(reqs[j] != reqs[i], "duplicate request in batch");
    offs[i + 1] = offs[i] + lens[i];
  }
  std::vector<std::unique_ptr<aic_suffix_tree>> built(n_req);
  for (int i = 0; i < n_req; ++i) built[i].reset(new aic_suffix_tree(c->max_depth));
  // prompt trees are independent: build them on host threads
  const int nt = std::max(1, std::min(n_threads, n_req));
  auto work = [&](int tid) {
    for (int i = tid; i < n_req; i += nt)
      for (int j = 0; j < lens[i]; ++j) built[i]->host.append(0, tokens[offs[i] + j]);
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int k = 0; k < nt; ++k) th.emplace_back(work, k);
    for (auto& x : th) x.join();
  }
  for (int i = 0; i < n_req; ++i) c->prompts.emplace(reqs[i], std::move(built[i]));
  return AIC_OK;
}
int aic_sc_evict_prompt(aic_suffix_cache* c, int64_t req) {
  AIC_REQUIRE(c, "null cache");
  auto it = c->prompts.find(req);
  if (it == c->prompts.end()) {
    set_error("prompt does not exist for request %lld", static_cast<long long>(req));
    return AIC_ERR_NOT_FOUND;
  }
  it->second->join_build();
  release_image(it->second.get());   // device buffers go back to the pool now (the next prompt tree reuses them)
  // tearing down the host arena (thousands of small maps) costs ~0.5 ms: done on a host thread, off the engine thread
  c->retire(std::move(it->second));
  c->prompts.erase(it);
  return AIC_OK;
}
int aic_sc_update_response(aic_suffix_cache* c, int64_t req, const int32_t* tokens, int n) {
  AIC_REQUIRE(c && (tokens || n == 0) && n >= 0, "bad arguments to aic_sc_update_response");
  auto it = c->seq_of.find(req);
  if (it == c->seq_of.end()) it = c->seq_of.emplace(req, static_cast<int32_t>(c->seq_of.size())).first;
  const int32_t sid = it->second;
  for (int i = 0; i < n; ++i) c->global->host.append(sid, tokens[i]);
  auto pt = c->prompts.find(req);
  if (pt != c->prompts.end()) {
    pt->second->join_build();
    for (int i = 0; i < n; ++i) pt->second->host.append(0, tokens[i]);
  }
  return AIC_OK;
}

// Pulls the tree state that the next aic_sc_update_responses for these requests will touch first into the CPU cache
// (read-only; HostTree::warm).  Meant to be called while the caller waits for the GPU anyway.
int aic_sc_warm(aic_suffix_cache* c, int n_req, const int64_t* reqs) {
  AIC_REQUIRE(c && n_req >= 0 && (n_req == 0 || reqs), "bad arguments to aic_sc_warm");
  volatile int64_t sink = 0;
  for (int r = 0; r < n_req; ++r) {
    auto it = c->seq_of.find(reqs[r]);
    if (it != c->seq_of.end()) sink = sink + c->global->host.warm(it->second);
    auto pt = c->prompts.find(reqs[r]);
    if (pt != c->prompts.end() && !pt->second->builder.joinable()) sink = sink + pt->second->host.warm(0);
  }
  (void)sink;
  return AIC_OK;
}

int aic_sc_update_responses(aic_suffix_cache* c, int n_req, const int64_t* reqs, const int32_t* tokens,
                            const int32_t* lens) {
  AIC_REQUIRE(c && n_req >= 0 && (n_req == 0 || (reqs && lens)), "bad arguments to aic_sc_update_responses");
  std::vector<int64_t> at(static_cast<size_t>(n_req) + 1, 0);
  for (int r = 0; r < n_req; ++r) {
    AIC_REQUIRE(lens[r] >= 0 && (tokens || lens[r] == 0), "bad token run for request %d", r);
    at[r + 1] = at[r] + lens[r];
  }
  // seq ids in first-seen order, exactly as n_req single calls would hand them out (suffix_cache.py:113-116)
  std::vector<int32_t> sid(n_req);
  for (int r = 0; r < n_req; ++r) {
    auto it = c->seq_of.find(reqs[r]);
    if (it == c->seq_of.end()) it = c->seq_of.emplace(reqs[r], static_cast<int32_t>(c->seq_of.size())).first;
    sid[r] = it->second;
  }
  // same order as one update_response per request.  (Extending the prompt trees on a helper thread while this
  // thread extends the global tree was measured twice — a thread per call and a parked worker woken per call — and was
  // slower both times, 0.25 -> 0.35 ms for 64 requests, and slowed the speculation that follows: the second core's
  // cache traffic costs more than the ~150 appends it takes off this thread.)
  for (int r = 0; r < n_req; ++r) {
    for (int64_t i = at[r]; i < at[r + 1]; ++i) c->global->host.append(sid[r], tokens[i]);
    auto pt = c->prompts.find(reqs[r]);
    if (pt != c->prompts.end() && at[r + 1] > at[r]) {
      pt->second->join_build();
      for (int64_t i = at[r]; i < at[r + 1]; ++i) pt->second->host.append(0, tokens[i]);
    }
  }
  return AIC_OK;
}

static int sc_speculate_batch(aic_suffix_cache* c, int n_query, const int64_t* reqs, const int32_t* patterns,
                              const int32_t* pattern_lens, const int32_t* max_spec_tokens, const float* factor,
                              const float* offset, const float* min_prob, const int32_t* use_prompt, int cap,
                              int32_t* out_tokens, float* out_probs, int32_t* out_n, float* out_score,
                              int32_t* out_match_len, void* stream, int32_t* out_parents) {
  AIC_REQUIRE(c && n_query >= 0 && cap > 0, "bad arguments to aic_sc_speculate_batch");
  if (n_query == 0) return AIC_OK;
  AIC_REQUIRE(reqs && patterns && pattern_lens && max_spec_tokens && factor && offset && min_prob && use_prompt &&
                  out_tokens && out_probs && out_n && out_score && out_match_len,
              "null array passed to aic_sc_speculate_batch");
  // validate before touching the device (suffix_cache.py:189-192)
  for (int i = 0; i < n_query; ++i) {
    if (pattern_lens[i] <= 0) {
      set_error("pattern must not be empty (query %d)", i);
      return AIC_ERR_INVALID;
    }
    if (use_prompt[i] && !c->prompts.count(reqs[i])) {
      set_error("prompt does not exist for request %lld", static_cast<long long>(reqs[i]));
      return AIC_ERR_NOT_FOUND;
    }
  }
  AIC_NEED_DEVICE();
  Mirror& mir = c->mirror;
  const auto t_begin = std::chrono::steady_clock::now();
  const bool tree_mode = out_parents != nullptr;
  if (tree_mode) c->global->host.enable_kid_tracking();      // (its first mirror after this uploads every child list)
  mir.begin();
  int rc, gdesc = -1;
  if ((rc = mir.add_tree(c->global.get(), c->pool, &gdesc)) != AIC_OK) return mir.fail_batch(rc);
  std::unordered_map<int64_t, int> pdesc;
  std::vector<QueryRec> qs(n_query);
  std::vector<int32_t> pool_words;
  int64_t src = 0;
  int n_starts = 1;
  for (int i = 0; i < n_query; ++i) {
    int len = pattern_lens[i];
    const int32_t* p = patterns + src;
    src += len;
    if (len > c->max_depth) {  // suffix_cache.py:197-198
      p += len - c->max_depth;
      len = c->max_depth;
    }
    QueryRec& q = qs[i];
    q.pattern_off = static_cast<int32_t>(pool_words.size());
    q.pattern_len = len;
    pool_words.insert(pool_words.end(), p, p + len);
    q.max_spec = max_spec_tokens[i];
    q.factor = factor[i];
    q.offset = offset[i];
    q.min_prob = min_prob[i];
    q.global_tree = gdesc;
    q.prompt_tree = -1;
    if (use_prompt[i]) {
      auto it = pdesc.find(reqs[i]);
      if (it == pdesc.end()) {
        int di = -1;
        aic_suffix_tree* pt = c->prompts[reqs[i]].get();
        if (tree_mode) {
          pt->join_build();
          pt->host.enable_kid_tracking();
        }
        if ((rc = mir.add_tree(pt, c->pool, &di)) != AIC_OK) return mir.fail_batch(rc);
        it = pdesc.emplace(reqs[i], di).first;
      }
      q.prompt_tree = it->second;
    }
    n_starts = std::max(n_starts, len);
  }
  const auto t_built = std::chrono::steady_clock::now();
  rc = mir.run(qs, pool_words, n_starts, cap, out_tokens, out_probs, out_n, out_score, out_match_len,
               static_cast<hipStream_t>(stream), out_parents);
  const auto t_done = std::chrono::steady_clock::now();
  c->last_build_us = std::chrono::duration<float, std::micro>(t_built - t_begin).count();
  c->last_device_us = std::chrono::duration<float, std::micro>(t_done - t_built).count();
  if (rc == AIC_OK && tree_mode)
    for (int i = 0; i < n_query; ++i) (out_n[i] < 0 ? g_tree_mode_fallbacks : g_tree_mode_device) += 1;
  return rc;
}

int aic_sc_speculate_batch(aic_suffix_cache* c, int n_query, const int64_t* reqs, const int32_t* patterns,
                           const int32_t* pattern_lens, const int32_t* max_spec_tokens, const float* factor,
                           const float* offset, const float* min_prob, const int32_t* use_prompt, int cap,
                           int32_t* out_tokens, float* out_probs, int32_t* out_n, float* out_score,
                           int32_t* out_match_len, void* stream) {
  return sc_speculate_batch(c, n_query, reqs, patterns, pattern_lens, max_spec_tokens, factor, offset, min_prob, use_prompt,
                            cap, out_tokens, out_probs, out_n, out_score, out_match_len, stream, nullptr);
}

// Tree-mode speculation for a batch (use_tree_spec = True, suffix_tree.cc:245-274) on the device: as above, plus
// out_parents [n_query][cap].  out_n[i] = -1: the device gave query i back (a node with more than 15 children on its way,
// or a priority queue beyond 256 entries) — the caller evaluates it on the host trees (aic_st_speculate on
// aic_sc_prompt_tree / aic_sc_global_tree with aic_debug_tree_mode_on_host, or SuffixCache._speculate_tree_mode).
int aic_sc_speculate_batch_tree(aic_suffix_cache* c, int n_query, const int64_t* reqs, const int32_t* patterns,
                                const int32_t* pattern_lens, const int32_t* max_spec_tokens, const float* factor,
                                const float* offset, const float* min_prob, const int32_t* use_prompt, int cap,
                                int32_t* out_tokens, int32_t* out_parents, float* out_probs, int32_t* out_n,
                                float* out_score, int32_t* out_match_len, void* stream) {
  AIC_REQUIRE(out_parents, "aic_sc_speculate_batch_tree needs out_parents");
  return sc_speculate_batch(c, n_query, reqs, patterns, pattern_lens, max_spec_tokens, factor, offset, min_prob, use_prompt,
                            cap, out_tokens, out_probs, out_n, out_score, out_match_len, stream, out_parents);
}

// where the last aic_sc_speculate_batch spent its wall time: collecting the trees' deltas and the queries on the host,
// then everything from the staging copy to the stream synchronisation (tools/host_profile.py)
int aic_sc_last_timing(const aic_suffix_cache* c, float* build_us, float* device_us) {
  AIC_REQUIRE(c, "null cache");
  if (build_us) *build_us = c->last_build_us;
  if (device_us) *device_us = c->last_device_us;
  return AIC_OK;
}

int aic_sc_last_stats(const aic_suffix_cache* c, float* match_us, int64_t* mirrored_bytes, int64_t* n_nodes_total) {
  AIC_REQUIRE(c, "null cache");
  if (match_us) *match_us = const_cast<aic_suffix_cache*>(c)->mirror.match_us();
  if (mirrored_bytes) *mirrored_bytes = c->mirror.mirrored_bytes();
  if (n_nodes_total) {
    int64_t n = static_cast<int64_t>(c->global->host.num_nodes());
    for (const auto& kv : c->prompts) {
      kv.second->join_build();
      n += static_cast<int64_t>(kv.second->host.num_nodes());
    }
    *n_nodes_total = n;
  }
  return AIC_OK;
}
aic_suffix_tree* aic_sc_global_tree(aic_suffix_cache* c) { return c ? c->global.get() : nullptr; }
aic_suffix_tree* aic_sc_prompt_tree(aic_suffix_cache* c, int64_t req) {
  if (!c) return nullptr;
  auto it = c->prompts.find(req);
  if (it == c->prompts.end()) return nullptr;
  it->second->join_build();
  return it->second.get();
}

}  // extern "C"
