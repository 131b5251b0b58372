// HBM image of one suffix tree, shared by the host mirror (suffix_host.hpp) and the matcher
// kernels (suffix_device.hip).  Everything is int32 so the same words are valid on both sides.
//
//   nodes   : NodeRec[n_nodes]      32 B each (two 16-B loads)
//   hash    : HashSlot[n_slots]     16 B each, open addressing, linear probing, n_slots = 2^k
//   tokens  : int32[...]            every sequence owns one contiguous region
//   seq_base: int32[2 * n_seq_slots]  {offset of that region, current length of the sequence} per slot
//   kids    : int32[n_nodes][16]    (only for trees that were asked for tree-mode speculation) {n, child x 15} per node,
//                                   children in the host container's iteration order, n = -2: more than 15 children
//
// An edge label is (seq_slot, start, length): token j of the edge is
// tokens[seq_base[2 * seq_slot] + start + j]  (reference: Node{seq_id,start,length}, suffix_tree.h:24-44).
// length == -1 marks an OPEN leaf (suffix_host.hpp): its edge runs to the end of its sequence,
// length = seq_base[2 * seq_slot + 1] - start.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define AIC_HD __host__ __device__
#else
#define AIC_HD
#endif

namespace aic {

struct NodeRec {
  int32_t count;     // suffixes ending at or passing through the node
  int32_t parent;    // -1 for the root
  int32_t seq_slot;  // dense sequence slot of the edge label
  int32_t start;     // first token of the label in that sequence
  int32_t length;    // tokens on the edge into this node (0 for the root; -1 = open leaf, see above)
  int32_t best;      // child the reference's "most frequent child" scan would pick, -1 if none
  int32_t alive;     // 0 for a released slot
  int32_t pad;
};
static_assert(sizeof(NodeRec) == 32, "NodeRec must be 32 bytes");

enum : int32_t { SLOT_EMPTY = 0, SLOT_FULL = 1, SLOT_TOMB = 2 };

struct HashSlot {
  int32_t parent;
  int32_t token;
  int32_t child;
  int32_t state;
};
static_assert(sizeof(HashSlot) == 16, "HashSlot must be 16 bytes");

AIC_HD inline uint32_t edge_hash(int32_t parent, int32_t token) {
  uint32_t h = static_cast<uint32_t>(parent) * 0x9E3779B1u ^ (static_cast<uint32_t>(token) * 0x85EBCA77u);
  h ^= h >> 15;
  h *= 0xC2B2AE3Du;
  h ^= h >> 13;
  return h;
}

// What the matcher needs to address one tree.
struct TreeDesc {
  const NodeRec* nodes;
  const HashSlot* hash;
  const int32_t* tokens;
  const int32_t* seq_base;
  uint32_t hash_mask;
  int32_t n_nodes;
  const int32_t* kids;   // [n_nodes][kKidBlock] child lists in container order (tree-mode speculation) or nullptr
};
static_assert(sizeof(TreeDesc) == 48, "TreeDesc is copied through the int32 blob");

// One speculation query (one request of the engine step).
struct QueryRec {
  int32_t pattern_off;   // into the int32 pattern pool
  int32_t pattern_len;   // already truncated to max_depth
  int32_t max_spec;      // max_spec_tokens
  int32_t prompt_tree;   // index into TreeDesc[] or -1
  int32_t global_tree;   // index into TreeDesc[] or -1
  float factor;
  float offset;
  float min_prob;
};
static_assert(sizeof(QueryRec) == 32, "QueryRec must be 32 bytes");

// Scatter job of the mirror-apply kernel: copy `n` records of `rec_words` int32 words from the
// staging blob to dst[index] (index list at idx_off, or contiguous from dst_first when idx_off < 0).
struct ApplyJob {
  int32_t* dst;
  int64_t src_off;   // word offset of the records in the blob
  int64_t idx_off;   // word offset of the int32 index list, or -1
  int32_t dst_first; // first destination record when contiguous
  int32_t n;
  int32_t rec_words;
  int32_t pad;
};

}  // namespace aic
