// Host half of the suffix tree: online update in an index arena that is kept byte-identical to the
// HBM image the matcher reads (suffix_layout.h), with dirty tracking so a step mirrors only what it
// touched.
//
// Behaviour follows the reference's SuffixTree::append (csrc/suffix_cache/suffix_tree.cc:31-126):
// at most max_depth "active" nodes per sequence, four update cases.  What is different here:
//   * nodes are NodeRec records in one vector (indices, not pointers), released slots are reused;
//   * child lookup for the device is a flat (parent,token)->child open-addressing table;
//   * the reference decides "most frequent child" ties by libstdc++ unordered_map iteration order
//     (suffix_tree.cc:208-214).  We keep one std::unordered_map<int,int32_t> per *internal* node,
//     driven through the same insert / overwrite / erase sequence so its iteration order is the
//     reference's by construction, and we resolve the scan at update time into NodeRec::best, so
//     the device never iterates children: one dependent load per speculated hop;
//   * OPEN LEAVES.  In the reference every appended token lengthens the private leaf of each of the
//     <= max_depth active suffixes by one (suffix_tree.cc: `node->length += 1` for count == 1 nodes) —
//     ~60 node writes per token, and here as many dirty records to mirror.  A private leaf's edge
//     always runs to the END of its sequence while its suffix is active, so such a leaf stores
//     length = kOpenLength and its length is read as seq_len(seq_slot) - start (Ukkonen's open
//     leaves): appending a token costs it nothing.  It is given its concrete length ("closed") when
//     its suffix leaves the active window, and before any structural operation reads its length —
//     every such reader is an active suffix processed AFTER the leaf's owner in the same append (a
//     node's children are deeper than the node, so only a shorter suffix can walk into a longer
//     one's leaf), which is exactly when the reference has already lengthened it.  What remains per
//     token: the root count, one closing, and the handful of structural updates.
#pragma once

#include <cstdint>
#include <cstring>
#include <deque>
#include <queue>
#include <unordered_map>
#include <vector>

#include "suffix_layout.h"

namespace aic {

struct HostCandidate {
  std::vector<int32_t> token_ids;
  std::vector<int32_t> parents;
  std::vector<float> probs;
  float score = 0.0f;
  int32_t match_len = 0;
};

constexpr int32_t kOpenLength = -1;  // NodeRec::length of an open leaf (also read by the matcher kernels)

// Tree-mode speculation on the device (suffix_tree.cc:245-274) expands EVERY child of a node in the container's iteration
// order, so trees that are asked for it mirror one fixed block per node: [n, child_0 .. child_{n-1}] in iteration order,
// n = kKidOverflow for a node with more than kKidMax children (the device gives such a query back to the host).  Opt-in
// per tree (enable_kid_tracking(), switched on by the first tree-mode query): serving never asks for tree mode
// (model_runner.py:734-740) and pays nothing.
constexpr int kKidBlock = 16;
constexpr int kKidMax = kKidBlock - 1;
constexpr int32_t kKidOverflow = -2;

class HostTree {
 public:
  using KidMap = std::unordered_map<int, int32_t>;

  struct Seq {
    int32_t id = 0;
    std::vector<int32_t> toks;
    std::deque<int32_t> active;
    int32_t base = 0;    // region in the device token pool
    int32_t cap = 0;
    int32_t synced = 0;  // tokens already mirrored into the region
  };

  explicit HostTree(int max_depth) : max_depth_(max_depth) {
    recs_.push_back(blank());
    recs_[0].alive = 1;
    kids_.push_back(nullptr);
    node_dirty_.push_back(0);
    mark_node(0);
    rebuild_hash(64);
  }
  ~HostTree() {
    for (KidMap* k : kids_) delete k;
  }
  HostTree(const HostTree&) = delete;
  HostTree& operator=(const HostTree&) = delete;

  int max_depth() const { return max_depth_; }
  int num_seqs() const { return static_cast<int>(seqs_.size()); }
  size_t num_nodes() const { return recs_.size(); }

  void append(int seq_id, int token) {
    const int32_t slot = seq_slot(seq_id);
    {
      Seq& s0 = seqs_[slot];
      s0.active.push_back(0);
      recs_[0].count += 1;
      mark_node(0);
      if (s0.active.size() > static_cast<size_t>(max_depth_)) {
        // its suffix stops growing here: the edge ends at the sequence's current end
        const int32_t oldest = s0.active.front();
        close_leaf(oldest < 0 ? ~oldest : oldest);
        s0.active.pop_front();
      }
      s0.toks.push_back(token);
      mark_seq(slot);
    }
    const int32_t n_tok = static_cast<int32_t>(seqs_[slot].toks.size());
    const size_t n_active = seqs_[slot].active.size();

    for (size_t i = 0; i < n_active; ++i) {
      // an active entry < 0 is ~(index of an OPEN leaf this suffix owns): the token lengthens it implicitly, and the
      // node record is not even read (most of the <= max_depth entries are such leaves; on a tree that has gone cold
      // between two engine steps every record read is a cache miss)
      const int32_t entry = seqs_[slot].active[i];
      if (entry < 0) continue;
      const int32_t ni = entry;
      const int32_t ci = find_kid(ni, token);

      if (ci < 0) {
        if (recs_[ni].count == 1 && ni != 0) {
          // a leaf that only this suffix runs through: its edge grows by the new token — implicitly if open
          if (recs_[ni].length != kOpenLength) {
            recs_[ni].length += 1;
            mark_node(ni);
          }
        } else {
          const int32_t leaf = alloc_node();
          NodeRec& L = recs_[leaf];
          L.parent = ni;
          L.count = 1;
          L.seq_slot = slot;
          L.start = n_tok - 1;
          L.length = kOpenLength;  // one token now, and every token this suffix still receives
          attach_new_kid(ni, token, leaf);
          seqs_[slot].active[i] = ~leaf;
        }
      } else if (close_leaf(ci), close_leaf(ni), recs_[ni].count == recs_[ci].count + 1 && ni != 0) {
        if (recs_[ci].length == 1) {
          // the child absorbs `ni` and takes its place under ni's parent
          const int32_t pi = recs_[ni].parent;
          NodeRec& C = recs_[ci];
          C.count += 1;
          C.seq_slot = slot;
          C.length = recs_[ni].length + 1;
          C.start = n_tok - C.length;
          C.parent = pi;
          mark_node(ci);
          const int first = tok_at(C.seq_slot, C.start);
          (*kids_[pi])[first] = ci;  // existing key, value overwrite: iteration order untouched
          mark_kids(pi);
          hash_set(pi, first, ci);
          if (pi != 0 && recs_[pi].best == ni) {
            recs_[pi].best = ci;  // same slot in the order, same count
            mark_node(pi);
          }
          hash_erase(ni, token);
          release_node(ni);
          seqs_[slot].active[i] = ci;
        } else {
          // `ni` swallows the child's first token
          NodeRec& N = recs_[ni];
          NodeRec& C = recs_[ci];
          N.seq_slot = slot;
          N.length += 1;
          N.start = n_tok - N.length;
          C.start += 1;
          C.length -= 1;
          mark_node(ni);
          mark_node(ci);
          const int first = tok_at(C.seq_slot, C.start);
          if (first != token) {
            KidMap& km = *kids_[ni];
            km[first] = ci;   // insert the new key first ...
            km.erase(token);  // ... then drop the old one (same order of operations as the reference)
            mark_kids(ni);
            hash_erase(ni, token);
            hash_insert(ni, first, ci);
          }
          // single child: best stays ci
        }
      } else {
        if (recs_[ci].length == 1) {
          recs_[ci].count += 1;
          mark_node(ci);
          count_raised(ni, ci);
          seqs_[slot].active[i] = ci;
        } else {
          // split the child's edge after its first token
          const int32_t mid = alloc_node();
          NodeRec& C = recs_[ci];
          NodeRec& M = recs_[mid];
          M.parent = ni;
          M.count = C.count + 1;
          M.seq_slot = slot;
          M.start = n_tok - 1;
          M.length = 1;
          M.best = ci;
          const int second = tok_at(C.seq_slot, C.start + 1);
          kids_[mid] = new KidMap();
          (*kids_[mid])[second] = ci;
          mark_kids(mid);
          hash_insert(mid, second, ci);
          (*kids_[ni])[token] = mid;  // existing key, value overwrite
          mark_kids(ni);
          hash_set(ni, token, mid);
          C.parent = mid;
          C.start += 1;
          C.length -= 1;
          mark_node(ci);
          mark_node(mid);
          if (ni != 0) {
            if (recs_[ni].best == ci) {
              recs_[ni].best = mid;
              mark_node(ni);
            } else {
              count_raised(ni, mid);
            }
          }
          seqs_[slot].active[i] = mid;
        }
      }
    }
  }

  // Tree-mode speculation (reference: _speculate_tree, suffix_tree.cc:245-274).  Only the offline
  // simulator of the reference asks for it; serving proposes linear chains (model_runner.py:734-740)
  // and those go through the device matcher.
  HostCandidate speculate_tree(const int32_t* pat, int n, int max_spec_tokens, float factor, float offset,
                               float min_prob) const {
    HostCandidate best;
    int first = n - max_depth_;
    if (first < 0) first = 0;
    for (int s = first; s < n; ++s) {
      int32_t node, idx;
      if (!walk(pat, n, s, &node, &idx)) continue;
      const int match_len = n - s;
      const int budget = spec_budget(match_len, max_spec_tokens, factor, offset);
      HostCandidate c = grow_tree(node, idx, budget, min_prob);
      if (c.score > best.score) {
        best = std::move(c);
        best.match_len = match_len;
      }
    }
    return best;
  }

  // float product + float sum, then a double add of 1e-6 and truncation (suffix_tree.cc:149-152)
  static int spec_budget(int match_len, int max_spec_tokens, float factor, float offset) {
    const float scaled = match_len * factor + offset;
    int b = static_cast<int>(scaled + 1e-6);
    if (b > max_spec_tokens) b = max_spec_tokens;
    return b < 0 ? 0 : b;
  }

  // Full "most frequent child" scan in container order for every node; returns mismatches with the
  // incrementally maintained NodeRec::best.
  int selfcheck() const {
    int bad = 0;
    for (size_t i = 1; i < recs_.size(); ++i) {
      if (!recs_[i].alive) continue;
      const int32_t want = kids_[i] ? scan_best(static_cast<int32_t>(i)) : -1;
      if (want != recs_[i].best) ++bad;
    }
    // the flat table must agree with the per-node maps
    size_t entries = 0;
    for (size_t i = 0; i < recs_.size(); ++i) {
      if (!recs_[i].alive || !kids_[i]) continue;
      for (const auto& kv : *kids_[i]) {
        ++entries;
        if (hash_find(static_cast<int32_t>(i), kv.first) != kv.second) ++bad;
      }
    }
    if (entries != n_full_) ++bad;
    size_t n_open = 0, n_owned = 0;
    for (size_t i = 1; i < recs_.size(); ++i) {
      if (!recs_[i].alive || recs_[i].length != kOpenLength) continue;
      ++n_open;
      if (kids_[i] || recs_[i].count != 1 || len_of(static_cast<int32_t>(i)) < 1) ++bad;
    }
    // every open leaf is owned by exactly one active suffix of its own sequence, tagged ~index
    for (size_t k = 0; k < seqs_.size(); ++k)
      for (int32_t e : seqs_[k].active)
        if (e < 0) {
          ++n_owned;
          const int32_t i = ~e;
          if (i <= 0 || static_cast<size_t>(i) >= recs_.size() || !recs_[i].alive || recs_[i].length != kOpenLength ||
              recs_[i].seq_slot != static_cast<int32_t>(k))
            ++bad;
        }
    if (n_open != n_owned) ++bad;
    return bad;
  }

  // ---- mirror state (consumed by the device side) ----------------------------------------------
  const std::vector<NodeRec>& recs() const { return recs_; }
  const std::vector<HashSlot>& slots() const { return slots_; }
  uint32_t hash_mask() const { return mask_; }
  std::vector<Seq>& seqs() { return seqs_; }
  const std::vector<Seq>& seqs() const { return seqs_; }
  int32_t pool_end() const { return pool_end_; }
  std::vector<int32_t>& dirty_nodes() { return dirty_nodes_; }
  std::vector<int32_t>& dirty_slots() { return dirty_slots_; }
  bool hash_rebuilt() const { return hash_rebuilt_; }
  // ---- child lists for tree-mode speculation on the device (kKidBlock words per node) ----------------------
  bool track_kids() const { return track_kids_; }
  void enable_kid_tracking() { track_kids_ = true; }       // (the first mirror after this uploads every node's list)
  std::vector<int32_t>& dirty_kids() { return dirty_kids_; }
  void serialize_kids(int32_t p, int32_t* out) const {
    const KidMap* km = kids_[p];
    const size_t n = km ? km->size() : 0;
    for (int j = 1; j < kKidBlock; ++j) out[j] = -1;
    if (n > static_cast<size_t>(kKidMax)) {
      out[0] = kKidOverflow;
      return;
    }
    out[0] = static_cast<int32_t>(n);
    int j = 1;
    if (km)
      for (const auto& kv : *km) out[j++] = kv.second;     // the container's iteration order (suffix_tree.cc:263)
  }
  void clear_dirty() {
    for (int32_t i : dirty_nodes_) node_dirty_[i] = 0;
    dirty_nodes_.clear();
    for (int32_t i : dirty_slots_) slot_dirty_[i] = 0;
    dirty_slots_.clear();
    for (int32_t i : dirty_seqs_) seq_dirty_[i] = 0;
    dirty_seqs_.clear();
    for (int32_t i : dirty_kids_) kid_dirty_[i] = 0;
    dirty_kids_.clear();
    hash_rebuilt_ = false;
  }
  // Reserve (or enlarge) the token-pool region of a sequence so that `need` tokens fit.
  // Returns true if the region moved (its tokens must be uploaded again from 0).
  bool fit_region(Seq& s, int32_t need) {
    if (need <= s.cap) return false;
    int32_t cap = s.cap ? s.cap : 64;
    while (cap < need) cap *= 2;
    s.base = pool_end_;
    s.cap = cap;
    s.synced = 0;
    pool_end_ += cap;
    return true;
  }

 private:
  static NodeRec blank() {
    NodeRec r;
    std::memset(&r, 0, sizeof(r));
    r.parent = -1;
    r.seq_slot = -1;
    r.best = -1;
    return r;
  }

  int32_t seq_slot(int seq_id) {
    auto it = slot_of_.find(seq_id);
    if (it != slot_of_.end()) return it->second;
    const int32_t s = static_cast<int32_t>(seqs_.size());
    seqs_.emplace_back();
    seqs_.back().id = seq_id;
    slot_of_.emplace(seq_id, s);
    return s;
  }
  int tok_at(int32_t slot, int32_t pos) const { return seqs_[slot].toks[pos]; }

  // ---- cache warm-up ------------------------------------------------------------------------------
 public:
  // Reads (and so pulls into the cache) what the next append() to `seq_id` will look at first: the records of its active
  // suffixes that are not open leaves and the headers / first buckets of their child maps.  Between two engine steps the
  // host does ~7 ms of other work and the trees go cold; append() is then a chain of dependent cache misses.  The engine
  // calls this while it would otherwise only wait for the GPU.  Returns a checksum so the reads cannot be dropped.
  int64_t warm(int seq_id) const {
    auto it = slot_of_.find(seq_id);
    if (it == slot_of_.end()) return 0;
    int64_t acc = 0;
    for (int32_t e : seqs_[it->second].active) {
      if (e < 0) continue;
      acc += recs_[e].count;
      const KidMap* km = kids_[e];
      if (km && !km->empty()) acc += km->begin()->second + static_cast<int64_t>(km->bucket_count());
    }
    return acc;
  }

  // ---- open leaves -------------------------------------------------------------------------------
 public:
  // tokens on the edge into node i (an open leaf's edge runs to the current end of its sequence)
  int32_t len_of(int32_t i) const {
    const NodeRec& N = recs_[i];
    return N.length != kOpenLength ? N.length : static_cast<int32_t>(seqs_[N.seq_slot].toks.size()) - N.start;
  }
  std::vector<int32_t>& dirty_seqs() { return dirty_seqs_; }

 private:
  void close_leaf(int32_t i) {
    if (recs_[i].length == kOpenLength) {
      recs_[i].length = len_of(i);
      mark_node(i);
      // its owner (an active suffix of the leaf's own sequence) goes back to the explicit bookkeeping
      for (int32_t& e : seqs_[recs_[i].seq_slot].active)
        if (e == ~i) {
          e = i;
          break;
        }
    }
  }
  void mark_seq(int32_t slot) {
    if (static_cast<size_t>(slot) >= seq_dirty_.size()) seq_dirty_.resize(slot + 1, 0);
    if (!seq_dirty_[slot]) {
      seq_dirty_[slot] = 1;
      dirty_seqs_.push_back(slot);
    }
  }

  int32_t find_kid(int32_t ni, int token) const {
    const KidMap* km = kids_[ni];
    if (!km) return -1;
    auto it = km->find(token);
    return it == km->end() ? -1 : it->second;
  }

  int32_t alloc_node() {
    int32_t i;
    if (!free_.empty()) {
      i = free_.back();
      free_.pop_back();
      recs_[i] = blank();
    } else {
      i = static_cast<int32_t>(recs_.size());
      recs_.push_back(blank());
      kids_.push_back(nullptr);
      node_dirty_.push_back(0);
    }
    recs_[i].alive = 1;
    mark_node(i);
    mark_kids(i);      // (a reused slot starts with an empty child list)
    return i;
  }
  void release_node(int32_t i) {
    delete kids_[i];
    kids_[i] = nullptr;
    recs_[i] = blank();
    mark_node(i);
    mark_kids(i);
    free_.push_back(i);
  }
  void mark_node(int32_t i) {
    if (!node_dirty_[i]) {
      node_dirty_[i] = 1;
      dirty_nodes_.push_back(i);
    }
  }
  void mark_kids(int32_t i) {
    if (!track_kids_) return;
    if (static_cast<size_t>(i) >= kid_dirty_.size()) kid_dirty_.resize(recs_.size(), 0);
    if (!kid_dirty_[i]) {
      kid_dirty_[i] = 1;
      dirty_kids_.push_back(i);
    }
  }

  // ---- "most frequent child" bookkeeping -------------------------------------------------------
  int32_t scan_best(int32_t p) const {
    int32_t pick = -1, top = 0;
    for (const auto& kv : *kids_[p]) {
      const int32_t c = recs_[kv.second].count;
      if (c > top) {
        pick = kv.second;
        top = c;
      }
    }
    return pick;
  }
  int32_t first_with_count(int32_t p, int32_t c) const {
    for (const auto& kv : *kids_[p])
      if (recs_[kv.second].count == c) return kv.second;
    return -1;
  }
  void set_best(int32_t p, int32_t b) {
    if (recs_[p].best != b) {
      recs_[p].best = b;
      mark_node(p);
    }
  }
  // a brand-new count-1 leaf under `p`
  void attach_new_kid(int32_t p, int token, int32_t leaf) {
    if (!kids_[p]) kids_[p] = new KidMap();
    KidMap& km = *kids_[p];
    const size_t buckets = km.bucket_count();
    km.emplace(token, leaf);
    mark_kids(p);
    hash_insert(p, token, leaf);
    if (p == 0) return;  // nobody ever asks for the root's most frequent child
    const int32_t b = recs_[p].best;
    if (b < 0) {
      set_best(p, leaf);
    } else if (km.bucket_count() != buckets) {
      set_best(p, scan_best(p));  // a rehash reorders the container
    } else if (recs_[b].count == 1) {
      set_best(p, km.begin()->second);  // every child has count 1: the first in order wins
    }
  }
  // child x of p (already holding its new, larger count) may have overtaken the current pick
  void count_raised(int32_t p, int32_t x) {
    if (p == 0) return;
    const int32_t b = recs_[p].best;
    if (b == x) return;
    if (b < 0) {
      set_best(p, x);
      return;
    }
    const int32_t cx = recs_[x].count, cb = recs_[b].count;
    if (cx > cb) {
      set_best(p, x);
    } else if (cx == cb) {
      set_best(p, first_with_count(p, cx));
    }
  }

  // ---- flat (parent,token)->child table --------------------------------------------------------
  void mark_slot(uint32_t i) {
    if (!slot_dirty_[i]) {
      slot_dirty_[i] = 1;
      dirty_slots_.push_back(static_cast<int32_t>(i));
    }
  }
  void rebuild_hash(size_t capacity) {
    hash_rebuilt_ = true;  // the whole table is uploaded again; no per-slot tracking until then
    std::vector<HashSlot> old;
    old.swap(slots_);
    slots_.assign(capacity, HashSlot{0, 0, 0, SLOT_EMPTY});
    slot_dirty_.assign(capacity, 0);
    dirty_slots_.clear();
    mask_ = static_cast<uint32_t>(capacity - 1);
    n_full_ = 0;
    n_tomb_ = 0;
    for (const HashSlot& s : old)
      if (s.state == SLOT_FULL) place(s.parent, s.token, s.child);
  }
  void place(int32_t parent, int32_t token, int32_t child) {
    uint32_t h = edge_hash(parent, token) & mask_;
    while (slots_[h].state == SLOT_FULL) h = (h + 1) & mask_;
    if (slots_[h].state == SLOT_TOMB) --n_tomb_;
    slots_[h] = HashSlot{parent, token, child, SLOT_FULL};
    ++n_full_;
    if (!hash_rebuilt_) mark_slot(h);
  }
  void hash_insert(int32_t parent, int32_t token, int32_t child) {
    if ((n_full_ + n_tomb_ + 1) * 2 > slots_.size()) {
      size_t cap = 64;
      while (cap < (n_full_ + 1) * 4) cap *= 2;
      rebuild_hash(cap);
    }
    place(parent, token, child);
  }
  int64_t hash_locate(int32_t parent, int32_t token) const {
    uint32_t h = edge_hash(parent, token) & mask_;
    while (slots_[h].state != SLOT_EMPTY) {
      if (slots_[h].state == SLOT_FULL && slots_[h].parent == parent && slots_[h].token == token) return h;
      h = (h + 1) & mask_;
    }
    return -1;
  }
  int32_t hash_find(int32_t parent, int32_t token) const {
    const int64_t h = hash_locate(parent, token);
    return h < 0 ? -1 : slots_[h].child;
  }
  void hash_set(int32_t parent, int32_t token, int32_t child) {
    const int64_t h = hash_locate(parent, token);
    if (h < 0) {
      hash_insert(parent, token, child);
      return;
    }
    slots_[h].child = child;
    if (!hash_rebuilt_) mark_slot(static_cast<uint32_t>(h));
  }
  void hash_erase(int32_t parent, int32_t token) {
    const int64_t h = hash_locate(parent, token);
    if (h < 0) return;
    slots_[h].state = SLOT_TOMB;
    --n_full_;
    ++n_tomb_;
    if (!hash_rebuilt_) mark_slot(static_cast<uint32_t>(h));
  }

  // ---- host-side walk (tree mode only) ---------------------------------------------------------
  bool walk(const int32_t* pat, int n, int s, int32_t* out_node, int32_t* out_idx) const {
    int32_t node = 0, idx = 0;
    for (int i = s; i < n; ++i) {
      if (idx >= len_of(node)) {
        const int32_t c = find_kid(node, pat[i]);
        if (c < 0) return false;
        node = c;
        idx = 0;
      }
      if (tok_at(recs_[node].seq_slot, recs_[node].start + idx) != pat[i]) return false;
      ++idx;
    }
    *out_node = node;
    *out_idx = idx;
    return true;
  }
  struct Pending {
    float prob;
    int32_t node, idx, parent;
  };
  struct ByProb {
    bool operator()(const Pending& a, const Pending& b) const { return a.prob < b.prob; }
  };
  HostCandidate grow_tree(int32_t node, int32_t idx, int budget, float min_prob) const {
    HostCandidate out;
    std::priority_queue<Pending, std::vector<Pending>, ByProb> heap;
    heap.push(Pending{1.0f, node, idx, -1});
    while (static_cast<int>(out.token_ids.size()) < budget && !heap.empty()) {
      const Pending it = heap.top();
      heap.pop();
      const NodeRec& N = recs_[it.node];
      if (it.idx < len_of(it.node)) {
        out.token_ids.push_back(tok_at(N.seq_slot, N.start + it.idx));
        out.parents.push_back(it.parent);
        out.probs.push_back(it.prob);
        out.score += it.prob;
        heap.push(Pending{it.prob, it.node, it.idx + 1, static_cast<int32_t>(out.token_ids.size()) - 1});
      } else if (kids_[it.node]) {
        for (const auto& kv : *kids_[it.node]) {
          const float p = it.prob * recs_[kv.second].count / static_cast<float>(N.count);
          if (p >= min_prob) heap.push(Pending{p, kv.second, 0, it.parent});
        }
      }
    }
    return out;
  }

  int max_depth_;
  std::vector<NodeRec> recs_;
  std::vector<KidMap*> kids_;
  std::vector<int32_t> free_;
  std::vector<uint8_t> node_dirty_;
  std::vector<int32_t> dirty_nodes_;
  bool track_kids_ = false;
  std::vector<uint8_t> kid_dirty_;
  std::vector<int32_t> dirty_kids_;

  std::vector<HashSlot> slots_;
  std::vector<uint8_t> slot_dirty_;
  std::vector<int32_t> dirty_slots_;
  uint32_t mask_ = 0;
  size_t n_full_ = 0, n_tomb_ = 0;
  bool hash_rebuilt_ = true;

  std::unordered_map<int, int32_t> slot_of_;
  std::vector<Seq> seqs_;
  std::vector<uint8_t> seq_dirty_;      // sequences whose length changed since the last mirror (open leaves read it)
  std::vector<int32_t> dirty_seqs_;
  int32_t pool_end_ = 0;
};

}  // namespace aic
