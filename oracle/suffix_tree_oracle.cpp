// ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not on any product path.
//
// CPU restatement of the reference's depth-bounded suffix tree
// (/root/reference/csrc/suffix_cache/suffix_tree.cc:31-274, types in
// suffix_tree.h:24-61).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product library
// (arcticinference_amd/csrc) never links or calls it.
//
// Why C++ and not plain C: the reference resolves "most frequent child" ties in
// libstdc++ std::unordered_map<int,...> iteration order (suffix_tree.cc:208-214,
// :263-271).  The only restatement that is bit-exact by construction keeps the
// same container type and the same sequence of insert / assign / erase
// operations on it, so children are std::unordered_map<int,int> here as well.
// Everything else is restated in a different shape: nodes live in one index
// arena (no pointers, no unique_ptr), edges are (seq,start,len) triples of
// int32, and the C ABI below is what ctypes binds.
//
// Parity status: PINNED.  tests/test_oracle_suffix.py replays the golden vectors
// in tests/golden/suffix_*.json, which were produced by the real reference
// sources compiled by oracle/Makefile into oracle/_ref (gen_golden.py).
//
// Build: see oracle/Makefile (g++ -O2 -std=c++17 -ffp-contract=off).

#include <cstdint>
#include <cstring>
#include <deque>
#include <queue>
#include <unordered_map>
#include <vector>

namespace {

struct ONode {
  int32_t count = 0;
  int32_t parent = -1;
  int32_t seq_id = -1;
  int32_t start = 0;
  int32_t length = 0;
  std::unordered_map<int, int> kids;  // first token of child edge -> child index
};

struct OSeq {
  std::vector<int32_t> toks;
  std::deque<int32_t> active;  // node indices, oldest (longest suffix) first
};

struct OCand {
  std::vector<int32_t> token_ids;
  std::vector<int32_t> parents;
  std::vector<float> probs;
  float score = 0.0f;
  int32_t match_len = 0;
};

class OTree {
 public:
  explicit OTree(int max_depth) : max_depth_(max_depth) {
    nodes_.emplace_back();  // index 0 is the root
  }

  int num_seqs() const { return static_cast<int>(seqs_.size()); }

  // suffix_tree.cc:31-126
  void append(int seq_id, int token) {
    OSeq& s = seqs_[seq_id];
    s.active.push_back(0);
    nodes_[0].count += 1;
    if (s.active.size() > static_cast<size_t>(max_depth_)) s.active.pop_front();
    s.toks.push_back(token);
    const int32_t n_tok = static_cast<int32_t>(s.toks.size());

    for (size_t i = 0; i < s.active.size(); ++i) {
      const int32_t ni = s.active[i];
      auto hit = nodes_[ni].kids.find(token);
      const int32_t ci = (hit == nodes_[ni].kids.end()) ? -1 : hit->second;

      if (ci < 0) {
        if (nodes_[ni].count == 1 && ni != 0) {
          // sole suffix through a leaf: lengthen the leaf edge (:50-56)
          nodes_[ni].length += 1;
        } else {
          // branch off a fresh leaf (:57-67)
          const int32_t leaf = alloc();
          ONode& L = nodes_[leaf];
          L.parent = ni;
          L.count = 1;
          L.seq_id = seq_id;
          L.start = n_tok - 1;
          L.length = 1;
          nodes_[ni].kids.emplace(token, leaf);
          s.active[i] = leaf;
        }
      } else if (nodes_[ni].count == nodes_[ci].count + 1 && ni != 0) {
        // this suffix is the only one ending inside `ni`; it has one child
        if (nodes_[ci].length == 1) {
          // fuse ni + child into the child, child takes ni's slot (:72-87)
          const int32_t pi = nodes_[ni].parent;
          ONode& C = nodes_[ci];
          C.count += 1;
          C.seq_id = seq_id;
          C.length = nodes_[ni].length + 1;
          C.start = n_tok - C.length;
          C.parent = pi;
          const int first = seqs_[C.seq_id].toks[C.start];
          nodes_[pi].kids[first] = ci;  // key exists: value overwrite only
          release(ni);
          s.active[i] = ci;
        } else {
          // lengthen ni by the child's first token, shorten the child (:88-100)
          ONode& N = nodes_[ni];
          ONode& C = nodes_[ci];
          N.seq_id = seq_id;
          N.length += 1;
          N.start = n_tok - N.length;
          C.start += 1;
          C.length -= 1;
          const int first = seqs_[C.seq_id].toks[C.start];
          if (first != token) {
            N.kids[first] = ci;  // new key is inserted first ...
            N.kids.erase(token);  // ... then the old key is removed
          }
        }
      } else {
        if (nodes_[ci].length == 1) {
          nodes_[ci].count += 1;  // (:104-106)
          s.active[i] = ci;
        } else {
          // split the child edge after its first token (:107-123)
          const int32_t mid = alloc();
          ONode& C = nodes_[ci];
          ONode& M = nodes_[mid];
          M.parent = ni;
          M.count = C.count + 1;
          M.seq_id = seq_id;
          M.start = n_tok - 1;
          M.length = 1;
          const int second = seqs_[C.seq_id].toks[C.start + 1];
          M.kids[second] = ci;
          nodes_[ni].kids[token] = mid;  // key exists: value overwrite only
          C.parent = mid;
          C.start += 1;
          C.length -= 1;
          s.active[i] = mid;
        }
      }
    }
  }

  // suffix_tree.cc:135-165
  OCand speculate(const int32_t* pat, int n, int max_spec_tokens, float factor,
                  float offset, float min_prob, bool tree_mode) {
    OCand best;
    int first = n - max_depth_;
    if (first < 0) first = 0;
    for (int s = first; s < n; ++s) {
      int32_t node;
      int32_t idx;
      if (!walk(pat, n, s, &node, &idx)) continue;
      const int match_len = n - s;
      // float product and sum, then a double add of 1e-6, then truncation (:149-152)
      const float scaled = match_len * factor + offset;
      int budget = static_cast<int>(scaled + 1e-6);
      if (budget > max_spec_tokens) budget = max_spec_tokens;
      if (budget < 0) budget = 0;
      OCand c = tree_mode ? grow_tree(node, idx, budget, min_prob)
                          : grow_path(node, idx, budget, min_prob);
      if (c.score > best.score) {
        best = std::move(c);
        best.match_len = match_len;
      }
    }
    return best;
  }

 private:
  // suffix_tree.cc:167-188
  bool walk(const int32_t* pat, int n, int s, int32_t* out_node, int32_t* out_idx) {
    int32_t node = 0;
    int32_t idx = 0;
    for (int i = s; i < n; ++i) {
      const int c = pat[i];
      if (idx >= nodes_[node].length) {
        auto hit = nodes_[node].kids.find(c);
        if (hit == nodes_[node].kids.end()) return false;
        node = hit->second;
        idx = 0;
      }
      const ONode& N = nodes_[node];
      if (seqs_[N.seq_id].toks[N.start + idx] != c) return false;
      ++idx;
    }
    *out_node = node;
    *out_idx = idx;
    return true;
  }

  // suffix_tree.cc:190-224
  OCand grow_path(int32_t node, int32_t idx, int budget, float min_prob) {
    OCand out;
    float prob = 1.0f;
    while (static_cast<int>(out.token_ids.size()) < budget && prob >= min_prob) {
      const ONode& N = nodes_[node];
      if (idx < N.length) {
        out.parents.push_back(static_cast<int32_t>(out.token_ids.size()) - 1);
        out.token_ids.push_back(seqs_[N.seq_id].toks[N.start + idx]);
        out.probs.push_back(prob);
        out.score += prob;
        ++idx;
      } else {
        int32_t pick = -1;
        int32_t top = 0;
        for (const auto& kv : N.kids) {  // container iteration order decides ties
          const int32_t c = nodes_[kv.second].count;
          if (c > top) {
            pick = kv.second;
            top = c;
          }
        }
        if (pick < 0) break;
        prob *= static_cast<float>(top) / N.count;
        node = pick;
        idx = 0;
      }
    }
    return out;
  }

  struct Pending {
    float prob;
    int32_t node;
    int32_t idx;
    int32_t parent;
  };
  struct ByProb {
    bool operator()(const Pending& a, const Pending& b) const { return a.prob < b.prob; }
  };

  // suffix_tree.cc:245-274
  OCand grow_tree(int32_t node, int32_t idx, int budget, float min_prob) {
    OCand out;
    std::priority_queue<Pending, std::vector<Pending>, ByProb> heap;
    heap.push(Pending{1.0f, node, idx, -1});
    while (static_cast<int>(out.token_ids.size()) < budget && !heap.empty()) {
      const Pending it = heap.top();
      heap.pop();
      const ONode& N = nodes_[it.node];
      if (it.idx < N.length) {
        out.token_ids.push_back(seqs_[N.seq_id].toks[N.start + it.idx]);
        out.parents.push_back(it.parent);
        out.probs.push_back(it.prob);
        out.score += it.prob;
        heap.push(Pending{it.prob, it.node, it.idx + 1,
                          static_cast<int32_t>(out.token_ids.size()) - 1});
      } else {
        for (const auto& kv : N.kids) {
          const float p = it.prob * nodes_[kv.second].count / static_cast<float>(N.count);
          if (p >= min_prob) heap.push(Pending{p, kv.second, 0, it.parent});
        }
      }
    }
    return out;
  }

  int32_t alloc() {
    if (!free_.empty()) {
      const int32_t i = free_.back();
      free_.pop_back();
      nodes_[i] = ONode();
      return i;
    }
    nodes_.emplace_back();
    return static_cast<int32_t>(nodes_.size()) - 1;
  }
  void release(int32_t i) {
    nodes_[i].kids.clear();
    free_.push_back(i);
  }

  int max_depth_;
  std::vector<ONode> nodes_;
  std::vector<int32_t> free_;
  std::unordered_map<int, OSeq> seqs_;
};

}  // namespace

extern "C" {

void* orc_st_create(int max_depth) { return new OTree(max_depth); }
void orc_st_destroy(void* h) { delete static_cast<OTree*>(h); }
int orc_st_num_seqs(void* h) { return static_cast<OTree*>(h)->num_seqs(); }
void orc_st_append(void* h, int seq_id, int token) { static_cast<OTree*>(h)->append(seq_id, token); }
void orc_st_extend(void* h, int seq_id, const int32_t* toks, int n) {
  OTree* t = static_cast<OTree*>(h);
  for (int i = 0; i < n; ++i) t->append(seq_id, toks[i]);
}

// Returns the number of speculated tokens (<= cap).  score/match_len always written.
int orc_st_speculate(void* h, const int32_t* pattern, int n, int max_spec_tokens, float factor,
                     float offset, float min_prob, int use_tree, int32_t* out_tokens,
                     int32_t* out_parents, float* out_probs, int cap, float* out_score,
                     int32_t* out_match_len) {
  OCand c = static_cast<OTree*>(h)->speculate(pattern, n, max_spec_tokens, factor, offset,
                                              min_prob, use_tree != 0);
  int m = static_cast<int>(c.token_ids.size());
  if (m > cap) m = cap;
  for (int i = 0; i < m; ++i) {
    out_tokens[i] = c.token_ids[i];
    out_parents[i] = c.parents[i];
    out_probs[i] = c.probs[i];
  }
  *out_score = c.score;
  *out_match_len = c.match_len;
  return m;
}

}  // extern "C"
