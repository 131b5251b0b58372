// A5 — multi-token verify attention over a paged KV cache (the dominant kernel of a spec-decode step).
//
// Reference path: UlyssesAttentionPatch.forward hands q_/k_/v_ to vLLM's attention backend
// (/root/reference/arctic_inference/vllm/ulysses.py:510); no attention source exists in the
// reference.  Semantics (restated in oracle/spec_oracle.py): for request i with
// q_len_i = 1 + n_draft_i new tokens and context seq_lens[i] (new tokens included, K/V already
// written), query position j attends keys [0, seq_lens[i] - q_len_i + j]; GQA; FlashAttention page
// layout [num_blocks, block_size, Hkv, D] (llama_swiftkv.py:617).
//
// Roofline: HBM.  Per request and layer the kernel must read S x 2 x Hkv x D x 2 B of KV (16.8 MB
// at S=4096, Llama-3.1-8B) for 2 x q_len x Hq x D x S flops -> ~32 flop/B at q_len = 4, two orders
// of magnitude under the MFMA ridge.  So the design goal is bytes in flight, not MFMA utilisation:
//   * flash-decoding split: grid (request x kv head, splits, row groups) of 4-wave workgroups and
//     every WAVE owns its own token range with its own online-softmax state -> no workgroup
//     barrier in the main loop, thousands of independent streams, each with the next 32-token tile
//     (8 KiB K + 8 KiB V) prefetched into registers while the current one is computed;
//   * all G x q_len query rows of a kv head ride one MFMA tile (16 rows = 4 positions x G=4), so
//     the KV bytes are read once for all draft positions;
//   * S^T = K Q^T ("swapped" product): the accumulator then holds, per lane, 8 tokens of ONE query
//     row — soft-max statistics are lane-local plus two shuffles, and the same registers are
//     directly the B operand of O^T = V^T P^T (k-slot permutation shared by both operands), so P
//     never touches LDS;
//   * V is loaded row-contiguous (full 256-byte rows), staged in a wave-private LDS tile and read
//     back transposed with ds_read_b64_tr_b16 as the MFMA A operand.
// Partials (m, l, unnormalised O) per (split, wave) go to a workspace; a second kernel merges them.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>
#include <cstdlib>

#include "aic_common.h"

namespace aic {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// The block table is read-only for the kernels: addressed through the constant address space with wave-uniform
// indices its lookups become scalar loads (s_load_dword, lgkmcnt) and stay out of the vector-memory queue, where
// a dependent lookup between data loads would force the in-order vmcnt to drain every load issued before it.
typedef const int32_t __attribute__((address_space(4)))* const_i32_ptr;

constexpr int kTile = 32;  // tokens per inner step
constexpr int kD = 128;

struct AttnParams {
  const uint16_t* q;
  const uint16_t* k_cache;
  const uint16_t* v_cache;
  const int32_t* block_table;
  const int32_t* seq_lens;
  const int32_t* query_start_loc;
  float* ws_o;   // [n_splits][T*Hq][D]   one partial per workgroup
  float* ws_ml;  // [n_splits][T*Hq][2]
  int64_t q_stride;
  int64_t block_stride;
  int max_blocks;
  int num_q_heads;
  int num_kv_heads;
  int block_size;
  int long_dma;   // long-draft body: who issues the tile DMA (aic_debug_attn_long_dma; see verify_attn_long4_body)
  int bs_shift;   // log2(block_size) when it is a power of two, else -1 (the long-draft body then divides)
  int n_splits;       // token-range splits of THIS launch
  int n_parts_total;  // partial slots per row the combine kernel reads (max over the launches of a call)
  int total_rows;  // T * Hq
  int m_groups;    // row groups of MTQ*16 query rows (folded into blockIdx.x)
  int n_items;     // requests of this launch * head groups
  const int32_t* req_list;  // request ids of this launch (device) or nullptr = identity
  const float* k_scale;     // fp8 KV cache: per-tensor dequantisation scales (device scalars)
  const float* v_scale;
  float sm_scale;
  int dbg;
  // direct output (this launch's n_splits == 1 and its rows need no cross-workgroup merge): the workgroup that holds a
  // row's whole context writes the finished bf16 row to out; mark_final = a combine launch follows for other rows of
  // the call and must skip these (slot 0 gets the "already final" mark, l = -1)
  uint16_t* out;
  int64_t out_stride;
  int direct;
  int mark_final;
  int64_t* trace;  // debug (aic_debug_attn_trace): per workgroup {start, end (100 MHz ticks), HW_ID | XCC_ID << 32, kind}
  // short workgroups whose index in the launch (by * light_stride + bx) is >= light_first take light_pct % of a full
  // split's tiles (100: all equal): the ones that will share their CU with a long-draft workgroup
  int light_first;
  int light_stride;
  int light_pct;
  // gpt-oss layers.  window > 0: query position p sees keys p - window + 1 .. p only — the token range of a request starts
  // at the tile that holds key (ctx - q_len) - window + 1 instead of at 0, and the lower bound joins the causal mask.
  // sinks (f32 [Hq], may be null): one extra logit per head in the soft-max normalisation, no value — it is the initial
  // (max, sum = 1) state of the FIRST partial of every row, so it is counted once however the range is split.
  int window;
  const float* sinks;
  // debug (aic_debug_attn_phase_trace): per short-body workgroup (by * ptrace_stride + bx) eight 100 MHz timestamps —
  // entry, request geometry known, first tile requested, first tile consumed, loop end, partials stored — or nullptr
  int64_t* ptrace;
  int ptrace_stride;
};

// first token of the range a request's rows can see, rounded down to a tile (0 without a window)
__device__ __forceinline__ int window_begin(const AttnParams& P, int ctx, int q_len) {
  if (P.window <= 0) return 0;
  const int lo = ctx - q_len - P.window + 1;
  return lo > 0 ? (lo & ~(kTile - 1)) : 0;
}

// 16-byte load of KV bytes, non-temporal: every byte of the cache is read once per call, and with the default
// policy the stream evicts itself through the L2 (measured on the B = 64 x 4096-token case: 190.7 -> 171.0 us per
// call, 5.63 -> 6.28 TB/s).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ uint4 ld_kv16(const char* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Byte offset of 16-byte chunk `ch` (0..15) of token `t` (0..31) in the wave's V tile.  256-byte rows with
// the chunk index XOR-ed by a function of the row (cdna guide T10, image (b)), and token groups 4-7 /
// 8-11 swapped between rows so that the two 4-row blocks a 32-lane half reads transposed sit 8 rows
// apart: both the ds_write_b128 fill and the ds_read_b64_tr_b16 reads are then bank-conflict free.
__device__ __forceinline__ int v_tile_off(int t, int ch) {
  const int row = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// WH (waves = heads): the four waves of a workgroup take four consecutive kv heads over the SAME token
// range, so the workgroup reads 4 x 256 B = 1 KiB contiguous per token and the heads of a token are
// fetched together (DRAM page locality: with one head per workgroup every 2 KiB token row was touched by
// eight workgroups at eight different times).  !WH (Hkv not a multiple of 4, e.g. one kv head per rank under
// SP=8, where rows are contiguous anyway): the four waves split the token range of one head and are merged
// through LDS at the end.
// Image of a 32-token tile for head size D: D = 128 is v_tile_off; D = 64 (128-byte rows, 8 chunks) XORs the chunk
// with the row (the secondary head size: correct and DMA-fillable first, bank-optimal second).
template <int D>
__device__ __forceinline__ int tile_off(int t, int ch) {
  if constexpr (D == 128) {
    return v_tile_off(t, ch);
  } else {
    return 128 * t + 16 * (ch ^ (t & 7));
  }
}

// Lanes c16 + 16 g (g = 0..3) hold the same query row.  gfx950's row / half swaps combine them on the VALU
// (ds_bpermute costs an LDS round trip per step, and the long-draft kernel is VALU/latency bound).  The
// s_nops are the VALU-write -> permlane-swap -> VALU-read hazards the assembler does not see inside asm.
__device__ __forceinline__ void swap16(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
}
// maxima without the canonicalising v_max(x, x) the compiler puts in front of every fmaxf of a value it cannot prove quiet
// (MFMA results): scores are never signalling NaNs
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max_nc(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kLazySlack = 6.0f;   // long-draft body: a row's adopted maximum may trail the true one by this much (base-2 exponent)

__device__ __forceinline__ float rowgroup_max(float x) {
  float a = x, b = x;
  swap16(a, b);
  a = a > b ? a : b;
  b = a;
  swap32(a, b);
  return a > b ? a : b;
}
__device__ __forceinline__ float rowgroup_sum(float x) {
  float a = x, b = x;
  swap16(a, b);
  a += b;
  b = a;
  swap32(a, b);
  return a + b;
}

// Online-softmax update of one 16-row query tile against one 32-token KV tile, in the S^T layout: the lane owns
// query row c16 and the 8 tokens tt + 16 th + 4 g + e.  Scores stay raw; the scale (folded with log2 e, v_exp_f32 is base 2)
// enters the exponent's fma.  Produces P as bf16 head + tail fragments (B operand of O^T = V^T P^T).
// r04: the row maximum is LAZY, as in the long-draft body — a row keeps the maximum it last adopted (m_run, scaled; -inf =
// nothing visible yet) while no score of the tile exceeds it by more than kLazySlack: the common tile has no cross-lane
// exchange, no alpha and no rescale; l_part is this LANE's share of the row's sum (rowgroup_sum once, after the loop).
// Masking is a real branch (the uniform need_mask test used to be if-converted: 16 selects, 22 compares and 24 scalar
// ands on EVERY tile — ~60 of the fp8 body's ~380 instructions per tile, a loop SQ counters show issue-bound).
template <int DT>
__device__ __forceinline__ void softmax_tile(const f32x4& s0, const f32x4& s1, bool need_mask, bool row_ok, int tt,
                                             int g, int t_end, int limit, int wnd, float scale_log2, float inv_scale,
                                             float& m_run, float& l_part, f32x4 (&o)[DT], bf16x8& pf, bf16x8& pl) {
  float sc[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = s0[e];
    sc[4 + e] = s1[e];
  }
  if (need_mask) {
    asm volatile("" ::: "memory");   // keeps this a branch: the block is rare and long
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int tok = tt + 16 * th + 4 * g + e;
        if (!(row_ok && tok < t_end && tok <= limit && tok > limit - wnd)) sc[th * 4 + e] = -INFINITY;
      }
  }
  const float lmax = max3f(max3f(max3f(sc[0], sc[1], sc[2]), sc[3], sc[4]), sc[5], max_nc(sc[6], sc[7]));
  // raw score above which the row must adopt a new maximum (-inf while nothing is visible: any finite score then does)
  const float thr = __builtin_fmaf(m_run, inv_scale, kLazySlack * inv_scale);
  if (__any(lmax > thr)) {
    const float tmax = rowgroup_max(lmax);
    const float m_new = fmaxf(m_run, tmax * scale_log2);
    const float m_to = m_new == -INFINITY ? 0.0f : m_new;   // nothing visible yet: exponents against 0 (all scores are -inf)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_to);
    m_run = m_new;
    l_part *= alpha;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
  }
  const float m_use = m_run == -INFINITY ? 0.0f : m_run;
  float pv[8], p0 = 0.0f, p1 = 0.0f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], scale_log2, -m_use));
    if (e & 1) p1 += pv[e];
    else p0 += pv[e];
  }
  l_part += p0 + p1;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    pf[e] = static_cast<__bf16>(pv[e]);
    pl[e] = static_cast<__bf16>(pv[e] - static_cast<float>(pf[e]));
  }
}

// 8 e4m3 bytes (two dwords) -> 8 bf16 (exact: every e4m3 value is a bf16 value).  gfx950's scaled converts take two
// fp8 to two bf16 in ONE instruction (v_cvt_scalef32_pk_bf16_fp8, scale 1.0): 4 VALU per 8 elements where the
// fp8 -> f32 -> bf16 route took 8.  SQ counters on the fp8 short body (profiles/r02_pmc_secondary_kernels.txt): 70.8 % of
// its wave cycles were instruction issue (bf16 body: 32 %), 1.15e8 VALU instructions for half the bytes of the bf16
// body's 8.3e7 — the conversion made it issue-bound.
__device__ __forceinline__ bf16x8 fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(static_cast<int>(lo), 1.0f, false);
  const bf16x2_t b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(static_cast<int>(lo), 1.0f, true);
  const bf16x2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(static_cast<int>(hi), 1.0f, false);
  const bf16x2_t d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(static_cast<int>(hi), 1.0f, true);
  return __builtin_shufflevector(__builtin_shufflevector(a, b, 0, 1, 2, 3), __builtin_shufflevector(c, d, 0, 1, 2, 3), 0, 1,
                                 2, 3, 4, 5, 6, 7);
}

// Rows written by a launch with fewer splits than the call's slot count end their slot list with a sentinel
// {-inf, -2} in the first unused slot: the combine kernel stops there.  (Marking every unused slot empty — 14 scattered
// 8-byte writes per row when one long draft took 16 splits beside 2-split short rows — made the combine launch of such
// calls 14 us instead of 5: tools/microbench.py ql, 63 short + 1 long 197 us per call against 190 us for 59 + 5.)
__device__ __forceinline__ void mark_unused_parts(const AttnParams& P, int64_t grow) {
  if (P.n_splits < P.n_parts_total) {
    float* mp = P.ws_ml + (static_cast<int64_t>(P.n_splits) * P.total_rows + grow) * 2;
    mp[0] = -INFINITY;
    mp[1] = -2.0f;
  }
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// KV8: the cache holds OCP e4m3 bytes (A16 writes them); tiles are dequantised to bf16 in registers (exact),
// k_scale folds into the soft-max scale and v_scale into the output, so no per-element scaling is needed.
// One 16-byte K load then covers the k-slots of TWO MFMA steps, and the Q fragments use the same slot map.
template <int MTQ, int NW, int HD = kD>
struct ShortLds {
  // per wave: one 32-token V tile (8 KiB; 4 KiB of it used with 128-byte rows); reused at the end for the cross-wave merge
  static constexpr int kMergeU4 = (8 * MTQ * 16 + 3 * MTQ * 16 * HD) / 4;
  static constexpr int kU4 = NW * kTile * 16 > kMergeU4 ? NW * kTile * 16 : kMergeU4;
};

// Body of the short / generic kernel for workgroup (bx, by) of its grid; v_lds_raw = ShortLds<MTQ, NW>::kU4 uint4 of LDS.
// HPW = kv heads per workgroup (NW, NW/2 ... 1): the NW waves are HPW groups of R = NW / HPW waves; a group owns one kv
// head and its R waves take R consecutive token ranges of the workgroup's share, merged through LDS at the end.
// HPW == NW: every wave its own head (the streaming layout: 1 KiB contiguous per token per workgroup), no merge.
// Smaller HPW trades that for more workgroups per request WITHOUT cross-workgroup partials: with 64 requests x 8 kv heads,
// HPW = 2 gives 256 workgroups whose waves stream exactly what the HPW = 4 / two-split form streamed per wave, but the two
// halves of a head meet in LDS and the finished row goes straight to `out` — no partial write, no combine launch.
// HD = head size: 128, or 64 with a bf16 cache (gpt-oss).  A 64-wide bf16 head is a 128-byte token row like a 128-wide fp8
// one: K and V are fetched with the fp8 form's lane mapping (two 16-byte K loads per token group, four V loads per tile),
// used as they are (no conversion), with two k-steps per score MFMA chain and four 16-wide output tiles.
template <int MTQ, int HPW, bool KV8, int NW, int HD = kD>
__device__ __forceinline__ void verify_attn_body(const AttnParams& P, uint4* v_lds_raw, const int bx, const int by) {
  static_assert(HD == 128 || (HD == 64 && !KV8), "head size 128, or 64 with a bf16 cache");
  constexpr bool ROW128 = KV8 || HD == 64;   // bytes of a token row of one head: 128 (else 256)
  constexpr int DT = HD / 16;                // 16-wide tiles of the output's head dimension
  constexpr bool WH = HPW == NW;
  constexpr int R = NW / HPW;   // token ranges (waves) per head inside the workgroup
  static_assert(NW % HPW == 0, "waves must divide evenly over the heads of a workgroup");
  uint4(*v_lds)[kTile * 16] = reinterpret_cast<uint4(*)[kTile * 16]>(v_lds_raw);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps loop control scalar
  int64_t* ptr_ = P.ptrace ? P.ptrace + (static_cast<int64_t>(by) * P.ptrace_stride + bx) * 8 : nullptr;
#define AIC_PSTAMP(i_) if (ptr_ && threadIdx.x == 0) ptr_[i_] = static_cast<int64_t>(__builtin_amdgcn_s_memrealtime());
  AIC_PSTAMP(0)
  const int g = lane >> 4, c16 = lane & 15;
  const int Hkv = P.num_kv_heads, Hq = P.num_q_heads, G = Hq / Hkv;
  const int hgroups = Hkv / HPW;
  // blockIdx.x = ((item / 8) * m_groups + row_group) * 8 + item % 8: the row groups of one (request, heads)
  // item are dispatched back to back AND on the same XCD (workgroups are dealt round-robin over the 8 XCDs),
  // so the extra row groups of a long (suffix) draft re-read their KV through that XCD's L2 instead of HBM;
  // the row groups a short request does not have exit below after two scalar loads.
  const int m_groups = P.m_groups;
  const int item = (bx / (8 * m_groups)) * 8 + (bx & 7);
  const int row_group = (bx >> 3) % m_groups;
  if (item >= P.n_items) return;
  const int ridx = item / hgroups;
  // The request's geometry is two dependent lookups (list entry -> offsets and length), ~0.7 us of every workgroup's start.  A
  // list of short requests is usually the identity (a lane step without a long draft): the entry's geometry is requested
  // together with the entry, as if it were, and read again only when the entry says otherwise.
  int req = ridx, q0 = P.query_start_loc[ridx], q1 = P.query_start_loc[ridx + 1], ctx = P.seq_lens[ridx];
  if (P.req_list) {
    req = P.req_list[ridx];
    if (req != ridx) {
      q0 = P.query_start_loc[req];
      q1 = P.query_start_loc[req + 1];
      ctx = P.seq_lens[req];
    }
  }
  req = __builtin_amdgcn_readfirstlane(req);
  q0 = __builtin_amdgcn_readfirstlane(q0);
  ctx = __builtin_amdgcn_readfirstlane(ctx);
  const int q_len = __builtin_amdgcn_readfirstlane(q1) - q0;
  const int head_local = wave / R, range = wave - head_local * R;
  const int h = (item - ridx * hgroups) * HPW + head_local;
  const int n_rows = q_len * G;
  const int row0 = row_group * (MTQ * 16);
  if (row0 >= n_rows) return;
  AIC_PSTAMP(1)

  const int n_parts = P.n_splits * R;
  const int part = by * R + range;
  const int t_lo = window_begin(P, ctx, q_len);           // 0 unless the layer has a sliding window
  const int wnd = P.window > 0 ? P.window : (1 << 30);
  const int tiles_total = (ctx - t_lo + kTile - 1) / kTile;
  int t_begin, t_end;
  if (P.light_pct == 100) {
    const int tiles_per_part = (tiles_total + n_parts - 1) / n_parts;
    t_begin = t_lo + part * tiles_per_part * kTile;
    t_end = min(ctx, t_begin + tiles_per_part * kTile);
  } else {
    // every split of THIS item weighs light_pct if its workgroup is one of the sharing ones, else 100: cumulative
    // weights -> tile boundaries (all workgroups of an item compute the same boundaries)
    int c0 = 0, c1 = 0, w_all = 0;
    for (int b2 = 0; b2 < P.n_splits; ++b2) {
      const int w = (b2 * P.light_stride + bx >= P.light_first) ? P.light_pct : 100;
      if (b2 < by) c0 += w * R;
      if (b2 == by) {
        c0 += w * range;
        c1 = c0 + w;
      }
      w_all += w * R;
    }
    t_begin = t_lo + (tiles_total * c0 / w_all) * kTile;
    t_end = min(ctx, t_lo + (tiles_total * c1 / w_all) * kTile);
  }

  const int64_t kv_row = static_cast<int64_t>(Hkv) * HD;  // elements between consecutive tokens of a page
  const_i32_ptr btab = (const_i32_ptr)(P.block_table + static_cast<int64_t>(req) * P.max_blocks);
  const int bs = P.block_size;  // multiple of 16: a 16-token group never straddles two pages
  const float scale_log2 = P.sm_scale * kLog2e * (KV8 ? *P.k_scale : 1.0f);
  const float inv_scale = 1.0f / scale_log2;
  const float out_scale = KV8 ? *P.v_scale : 1.0f;
  constexpr int ES = KV8 ? 1 : 2;  // bytes per cache element
  const char* kc = reinterpret_cast<const char*>(P.k_cache);
  const char* vc = reinterpret_cast<const char*>(P.v_cache);

  // ---- query fragments (B operand of S^T = K Q^T): lane (row c16, k-group g) ----------------------
  uint4 qf[MTQ][4];
  int row_pos[MTQ];  // query position of this lane's row
  bool row_ok[MTQ];
#pragma unroll
  for (int mt = 0; mt < MTQ; ++mt) {
    const int rr = row0 + mt * 16 + c16;
    row_ok[mt] = rr < n_rows;
    const int rc = min(rr, n_rows - 1);
    const int pos = rc / G, gq = rc - pos * G;
    row_pos[mt] = pos;
    const uint16_t* qrow = P.q + static_cast<int64_t>(q0 + pos) * P.q_stride + static_cast<int64_t>(h * G + gq) * HD;
#pragma unroll
    for (int s = 0; s < HD / 32; ++s)  // k-slot (s, g, j) -> d = 32 s + 8 g + j (bf16) | 64 (s/2) + 16 g + 8 (s%2) + j (fp8)
      qf[mt][s] = *reinterpret_cast<const uint4*>(qrow + (KV8 ? 64 * (s >> 1) + 16 * g + 8 * (s & 1) : 32 * s + 8 * g));
  }

  float m_run[MTQ], l_run[MTQ];
  f32x4 o_acc[MTQ][DT];
#pragma unroll
  for (int mt = 0; mt < MTQ; ++mt) {
    m_run[mt] = -INFINITY;
    l_run[mt] = 0.0f;
    if (P.sinks != nullptr && part == 0) {      // the row's first partial starts from the sink: (max, sum) = (sink, 1)
      const int rc = min(row0 + mt * 16 + c16, n_rows - 1);
      m_run[mt] = P.sinks[h * G + (rc - (rc / G) * G)] * kLog2e;      // log2 domain, like the scaled scores
      l_run[mt] = g == 0 ? 1.0f : 0.0f;         // (inside the loop l_run is this lane's share of the row's sum: one lane carries the 1)
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o_acc[mt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  char* vt = reinterpret_cast<char*>(v_lds[wave]);

  // Page lookup for the two 16-token groups of a tile.  It is wave-uniform (scalar), and it is issued
  // one tile AHEAD of the data loads that use it: a block-table load placed between the data loads
  // would serialise them (vmcnt retires in order), which costs one HBM round trip per group.
  struct TilePages {
    int64_t base0, base1;  // element offset of each group's first token row inside the cache
    int first0, first1;    // first token of each group (clamped into the context)
  };
  const int last_group = (ctx - 1) & ~15;
  // (unsigned 32-bit page arithmetic, a shift for power-of-two pages: the signed divisions and 64-bit products this replaced
  // were ~50 of the loop's ~135 scalar instructions per tile, and a scalar instruction costs a lone wave 7-8 cycles —
  // tools/exp/salu_rate.hip — in a loop whose fp8 form is issue-bound)
  const int bsh = P.bs_shift;
  const unsigned kv_row32 = static_cast<unsigned>(kv_row), head_off = static_cast<unsigned>(h * HD);
  const uint64_t block_stride_u = static_cast<uint64_t>(P.block_stride);
  auto tile_pages = [&](int tt) -> TilePages {
    TilePages tp;
    tp.first0 = min(tt, last_group);
    tp.first1 = min(tt + 16, last_group);
    const unsigned f0 = static_cast<unsigned>(tp.first0), f1 = static_cast<unsigned>(tp.first1);
    unsigned pg0, pg1, po0, po1;
    if (bsh >= 0) {
      pg0 = f0 >> bsh;
      pg1 = f1 >> bsh;
      po0 = f0 & static_cast<unsigned>(bs - 1);
      po1 = f1 & static_cast<unsigned>(bs - 1);
    } else {
      pg0 = f0 / static_cast<unsigned>(bs);
      pg1 = f1 / static_cast<unsigned>(bs);
      po0 = f0 - pg0 * static_cast<unsigned>(bs);
      po1 = f1 - pg1 * static_cast<unsigned>(bs);
    }
    const unsigned b0 = static_cast<unsigned>(btab[pg0]), b1 = static_cast<unsigned>(btab[pg1]);
    tp.base0 = static_cast<int64_t>(b0 * block_stride_u + (po0 * kv_row32 + head_off));
    tp.base1 = static_cast<int64_t>(b1 * block_stride_u + (po1 * kv_row32 + head_off));
    return tp;
  };

  // One 32-token tile: K as MFMA A fragments (token c16, d = 32 s + 8 g ..), V row-contiguous.
  // The tile registers are named scalars on purpose: carried across the loop as arrays they are kept in
  // scratch memory by the compiler whenever 3 waves per SIMD are requested.
  uint4 k00, k01, k02, k03, k10, k11, k12, k13;  // k<th><s>
  uint4 v0, v1, v2, v3, v4, v5, v6, v7;          // v<iv>
  if (ROW128) k02 = k03 = k12 = k13 = v4 = v5 = v6 = v7 = make_uint4(0, 0, 0, 0);
  // (addresses = a wave-uniform 64-bit base per 16-token group + a 32-bit per-lane offset, a 24-bit multiply: the 64-bit
  // per-lane products this replaced were ~18 quarter-rate instructions per tile in the fp8 body, which is issue-bound)
  const unsigned row_bytes = static_cast<unsigned>(kv_row) * ES;
#define AIC_LOAD_K(tt_, tp_)                                                                                   \
  {                                                                                                            \
    const unsigned off0_ = static_cast<unsigned>(min((tt_) + c16, ctx - 1) - (tp_).first0);                    \
    const unsigned off1_ = static_cast<unsigned>(min((tt_) + 16 + c16, ctx - 1) - (tp_).first1);               \
    const char* kp0_ = kc + (tp_).base0 * ES + static_cast<size_t>(__umul24(off0_, row_bytes) + 16u * g);      \
    const char* kp1_ = kc + (tp_).base1 * ES + static_cast<size_t>(__umul24(off1_, row_bytes) + 16u * g);      \
    k00 = ld_kv16(kp0_);                                                               \
    k01 = ld_kv16(kp0_ + 64);                                                          \
    k10 = ld_kv16(kp1_);                                                               \
    k11 = ld_kv16(kp1_ + 64);                                                          \
    if (!ROW128) {                                                                                             \
      k02 = ld_kv16(kp0_ + 128);                                                       \
      k03 = ld_kv16(kp0_ + 192);                                                       \
      k12 = ld_kv16(kp1_ + 128);                                                       \
      k13 = ld_kv16(kp1_ + 192);                                                       \
    }                                                                                                          \
  }
  // bf16: instruction iv moves tokens 4 iv + g, 16-byte chunk c16;  fp8: tokens 8 iv + lane/8, chunk lane%8
#define AIC_V_ADDR(tt_, tp_, iv_)                                                                              \
  (ROW128 ? vc + ((iv_) >= 2 ? (tp_).base1 : (tp_).base0) * ES +                                                \
             static_cast<size_t>(__umul24(static_cast<unsigned>(min((tt_) + 8 * (iv_) + (lane >> 3), ctx - 1) - ((iv_) >= 2 ? (tp_).first1 : (tp_).first0)), row_bytes) + 16u * (lane & 7)) \
       : vc + ((iv_) >= 4 ? (tp_).base1 : (tp_).base0) * 2 +                                                    \
               static_cast<size_t>(__umul24(static_cast<unsigned>(min((tt_) + 4 * (iv_) + g, ctx - 1) - ((iv_) >= 4 ? (tp_).first1 : (tp_).first0)), row_bytes) + 16u * c16))
#define AIC_LOAD_V(tt_, tp_)                                                                                   \
  {                                                                                                            \
    v0 = ld_kv16(AIC_V_ADDR(tt_, tp_, 0));                                             \
    v1 = ld_kv16(AIC_V_ADDR(tt_, tp_, 1));                                             \
    v2 = ld_kv16(AIC_V_ADDR(tt_, tp_, 2));                                             \
    v3 = ld_kv16(AIC_V_ADDR(tt_, tp_, 3));                                             \
    if (!ROW128) {                                                                                             \
      v4 = ld_kv16(AIC_V_ADDR(tt_, tp_, 4));                                           \
      v5 = ld_kv16(AIC_V_ADDR(tt_, tp_, 5));                                           \
      v6 = ld_kv16(AIC_V_ADDR(tt_, tp_, 6));                                           \
      v7 = ld_kv16(AIC_V_ADDR(tt_, tp_, 7));                                           \
    }                                                                                                          \
  }
#define AIC_STORE_V(iv_, reg_)                                                                                  \
  if (KV8) {                                                                                                   \
    if ((iv_) < 4) {                                                                                           \
      const int tok_ = 8 * (iv_) + (lane >> 3), ch_ = 2 * (lane & 7);                                          \
      *reinterpret_cast<bf16x8*>(vt + v_tile_off(tok_, ch_)) = fp8x8_to_bf16x8(reg_.x, reg_.y);                \
      *reinterpret_cast<bf16x8*>(vt + v_tile_off(tok_, ch_ + 1)) = fp8x8_to_bf16x8(reg_.z, reg_.w);            \
    }                                                                                                          \
  } else if (HD == 64) {                                                                                       \
    if ((iv_) < 4) *reinterpret_cast<uint4*>(vt + tile_off<64>(8 * (iv_) + (lane >> 3), lane & 7)) = reg_;     \
  } else {                                                                                                     \
    *reinterpret_cast<uint4*>(vt + v_tile_off(4 * (iv_) + g, c16)) = reg_;                                     \
  }

  // Software pipeline with ONE register set per operand: the registers of a tile are re-armed with the
  // next tile's loads as soon as their last consumer has issued (V: right after its LDS write, K: right
  // after the QK^T MFMAs), so each load has about a full iteration to land and the kernel stays under
  // 168 VGPRs (3 waves per SIMD = 12 x 16 KiB in flight per CU).  Loads past a wave's range are clamped
  // into the context by tile_pages(): always valid addresses, results never used.
  if (t_begin < t_end) {
    TilePages pages = tile_pages(t_begin);
    AIC_LOAD_V(t_begin, pages)
    AIC_LOAD_K(t_begin, pages)
    TilePages pages_next = tile_pages(t_begin + kTile);
    AIC_PSTAMP(2)

    for (int tt = t_begin; tt < t_end; tt += kTile) {
      const int tn = tt + kTile;
      // 1. V(t) registers -> wave-private LDS tile; 2. re-arm them with V(t+1)
      AIC_STORE_V(0, v0) AIC_STORE_V(1, v1) AIC_STORE_V(2, v2) AIC_STORE_V(3, v3)
      AIC_STORE_V(4, v4) AIC_STORE_V(5, v5) AIC_STORE_V(6, v6) AIC_STORE_V(7, v7)
      AIC_LOAD_V(tn, pages_next)

      // 3. S^T tiles (K registers), online softmax per 16-row query tile
      // P is split into a bf16 head and a bf16 tail (p = hi + lo up to 2^-17 relative): the kernel is
      // HBM-bound with the matrix pipe a few percent busy, so a second P.V MFMA is free and removes the
      // 2^-9 relative rounding a single bf16 P would put on every term
      f32x4 st[MTQ][2];
#pragma unroll
      for (int mt = 0; mt < MTQ; ++mt) {
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#define AIC_QK(acc_, kfrag_, s_) \
  acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag_, __builtin_bit_cast(bf16x8, qf[mt][s_]), acc_, 0, 0, 0);
#define AIC_B16(r_) __builtin_bit_cast(bf16x8, r_)
        if (KV8) {
          AIC_QK(a0, fp8x8_to_bf16x8(k00.x, k00.y), 0) AIC_QK(a0, fp8x8_to_bf16x8(k00.z, k00.w), 1)
          AIC_QK(a0, fp8x8_to_bf16x8(k01.x, k01.y), 2) AIC_QK(a0, fp8x8_to_bf16x8(k01.z, k01.w), 3)
          AIC_QK(a1, fp8x8_to_bf16x8(k10.x, k10.y), 0) AIC_QK(a1, fp8x8_to_bf16x8(k10.z, k10.w), 1)
          AIC_QK(a1, fp8x8_to_bf16x8(k11.x, k11.y), 2) AIC_QK(a1, fp8x8_to_bf16x8(k11.z, k11.w), 3)
        } else if (HD == 64) {      // two k-steps: d = 8 g + j and 32 + 8 g + j
          AIC_QK(a0, AIC_B16(k00), 0) AIC_QK(a0, AIC_B16(k01), 1)
          AIC_QK(a1, AIC_B16(k10), 0) AIC_QK(a1, AIC_B16(k11), 1)
        } else {
          AIC_QK(a0, AIC_B16(k00), 0) AIC_QK(a0, AIC_B16(k01), 1) AIC_QK(a0, AIC_B16(k02), 2) AIC_QK(a0, AIC_B16(k03), 3)
          AIC_QK(a1, AIC_B16(k10), 0) AIC_QK(a1, AIC_B16(k11), 1) AIC_QK(a1, AIC_B16(k12), 2) AIC_QK(a1, AIC_B16(k13), 3)
        }
#undef AIC_B16
#undef AIC_QK
        st[mt][0] = a0;
        st[mt][1] = a1;
      }
      // 4. re-arm the K registers with K(t+1); page lookup one tile ahead of its data loads
      AIC_LOAD_K(tn, pages_next)
      pages_next = tile_pages(tn + kTile);

      bf16x8 pfrag[MTQ], pfrag_lo[MTQ];
      // the tile reaches the causal edge / range end, or (window) lies below the last row's lower bound ctx - window
      const bool need_mask = tn > t_end || tn > ctx - q_len + 1 || tt < ctx - wnd;
#pragma unroll
      for (int mt = 0; mt < MTQ; ++mt) {
        softmax_tile<DT>(st[mt][0], st[mt][1], need_mask, row_ok[mt], tt, g, t_end, ctx - q_len + row_pos[mt], wnd, scale_log2,
                         inv_scale, m_run[mt], l_run[mt], o_acc[mt], pfrag[mt], pfrag_lo[mt]);
      }

      // 5. O^T += V^T P^T : A = V^T fragment via transposing LDS reads
      {
        const int q4 = c16 >> 2, p4 = c16 & 3;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          // lane 4q+p of a 16-lane group supplies row q of the block, columns 4p..4p+3 (8 bytes)
          const char* a_lo = vt + tile_off<HD>(4 * g + q4, 2 * dt + (p4 >> 1)) + 8 * (p4 & 1);
          const char* a_hi = vt + tile_off<HD>(16 + 4 * g + q4, 2 * dt + (p4 >> 1)) + 8 * (p4 & 1);
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(a_lo)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(a_hi)));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 vfrag = __builtin_bit_cast(bf16x8, both);
#pragma unroll
          for (int mt = 0; mt < MTQ; ++mt) {
            o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pfrag[mt], o_acc[mt][dt], 0, 0, 0);
            o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pfrag_lo[mt], o_acc[mt][dt], 0, 0, 0);
          }
        }
      }
    }
  }
  AIC_PSTAMP(4)
  // the loop kept per-lane shares of the rows' sums: from here on l_run is the row's sum on each of its four lanes
#pragma unroll
  for (int mt = 0; mt < MTQ; ++mt) l_run[mt] = rowgroup_sum(l_run[mt]);

#undef AIC_LOAD_K
#undef AIC_LOAD_V
#undef AIC_V_ADDR
#undef AIC_STORE_V

  // finished rows (direct mode): normalise, round to bf16, 8 bytes per lane and 16-wide d tile
  auto store_final = [&](int mt, const f32x4 (&o)[DT], float inv_l, int64_t tok, int hq) {
    uint16_t* op = P.out + tok * P.out_stride + static_cast<int64_t>(hq) * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const uint32_t lo = static_cast<uint32_t>(f32_to_bf16(o[dt][0] * inv_l)) | (static_cast<uint32_t>(f32_to_bf16(o[dt][1] * inv_l)) << 16);
      const uint32_t hi = static_cast<uint32_t>(f32_to_bf16(o[dt][2] * inv_l)) | (static_cast<uint32_t>(f32_to_bf16(o[dt][3] * inv_l)) << 16);
      *reinterpret_cast<uint2*>(op + dt * 16) = make_uint2(lo, hi);
    }
  };
  auto mark_row_final = [&](int64_t grow) {   // tells the combine launch of a mixed call to leave this row alone
    float* mp = P.ws_ml + grow * 2;
    mp[0] = -INFINITY;
    mp[1] = -1.0f;
  };
  if (WH) {
    // every wave owns its own head: its partial goes straight to the workspace (or, alone on the row, to `out`)
#pragma unroll
    for (int mt = 0; mt < MTQ; ++mt) {
      if (!row_ok[mt]) continue;
      const int rr = row0 + mt * 16 + c16;
      const int pos = rr / G, gq = rr - pos * G;
      const int64_t grow = static_cast<int64_t>(q0 + pos) * Hq + h * G + gq;
      if (P.direct) {
        const float l = l_run[mt];   // already summed over the row's lanes by softmax_tile
        store_final(mt, o_acc[mt], l > 0.0f ? out_scale / l : 0.0f, q0 + pos, h * G + gq);
        if (g == 0 && P.mark_final) mark_row_final(grow);
        continue;
      }
      float* op = P.ws_o + (static_cast<int64_t>(by) * P.total_rows + grow) * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *reinterpret_cast<float4*>(op + dt * 16) =
            make_float4(o_acc[mt][dt][0] * out_scale, o_acc[mt][dt][1] * out_scale, o_acc[mt][dt][2] * out_scale,
                        o_acc[mt][dt][3] * out_scale);
      if (g == 0) {
        float* mp = P.ws_ml + (static_cast<int64_t>(by) * P.total_rows + grow) * 2;
        mp[0] = m_run[mt] * kLn2;  // the combine kernel works in natural-log units
        mp[1] = l_run[mt];
        if (by == 0) mark_unused_parts(P, grow);
      }
    }
    AIC_PSTAMP(5)
    if (ptr_ && threadIdx.x == 0) {
      ptr_[6] = (static_cast<int64_t>(t_end - t_begin) << 32) | static_cast<uint32_t>(ctx);
    }
    return;
  }
  // ---- merge the R waves of each head (they hold disjoint token ranges of the same rows) through LDS, so one
  // partial per workgroup and row goes to the workspace instead of R — or, in direct mode, the finished row to `out`
  __syncthreads();  // every wave is done with its V tile: the LDS is free
  float* xm = reinterpret_cast<float*>(v_lds_raw);  // [NW waves][MTQ][16 rows] running max
  float* xl = xm + NW * MTQ * 16;                   // [NW][MTQ][16] running sum
  float* xo = xl + NW * MTQ * 16;                   // [NW - HPW non-leader waves][MTQ][16 rows][HD] rescaled O
#pragma unroll
  for (int mt = 0; mt < MTQ; ++mt)
    if (g == 0) {
      xm[(wave * MTQ + mt) * 16 + c16] = m_run[mt];
      xl[(wave * MTQ + mt) * 16 + c16] = l_run[mt];
    }
  __syncthreads();
  const int w0 = head_local * R;   // first wave of this head's group
  float scale_w[MTQ], m_all[MTQ], l_all[MTQ];
#pragma unroll
  for (int mt = 0; mt < MTQ; ++mt) {
    float M = -INFINITY;
    for (int w = w0; w < w0 + R; ++w) M = fmaxf(M, xm[(w * MTQ + mt) * 16 + c16]);
    float L = 0.0f;
    for (int w = w0; w < w0 + R; ++w) {
      const float mw = xm[(w * MTQ + mt) * 16 + c16];
      if (mw > -INFINITY) L += xl[(w * MTQ + mt) * 16 + c16] * __builtin_amdgcn_exp2f(mw - M);
    }
    m_all[mt] = M;
    l_all[mt] = L;
    scale_w[mt] = (m_run[mt] > -INFINITY ? __builtin_amdgcn_exp2f(m_run[mt] - M) : 0.0f) * out_scale;
  }
  const int nl = head_local * (R - 1) + (range - 1);   // slot of a non-leader wave in xo
  if (range > 0) {
#pragma unroll
    for (int mt = 0; mt < MTQ; ++mt)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        float* dst = xo + ((static_cast<size_t>(nl) * MTQ + mt) * 16 + c16) * HD + dt * 16 + 4 * g;
        *reinterpret_cast<float4*>(dst) = make_float4(o_acc[mt][dt][0] * scale_w[mt], o_acc[mt][dt][1] * scale_w[mt],
                                                       o_acc[mt][dt][2] * scale_w[mt], o_acc[mt][dt][3] * scale_w[mt]);
      }
  }
  __syncthreads();
  if (range != 0) return;
  const int bpart = by;  // one partial per workgroup and head
#pragma unroll
  for (int mt = 0; mt < MTQ; ++mt) {
    if (!row_ok[mt]) continue;
    const int rr = row0 + mt * 16 + c16;
    const int pos = rr / G, gq = rr - pos * G;
    const int64_t grow = static_cast<int64_t>(q0 + pos) * Hq + h * G + gq;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      acc[dt] = f32x4{o_acc[mt][dt][0] * scale_w[mt], o_acc[mt][dt][1] * scale_w[mt], o_acc[mt][dt][2] * scale_w[mt],
                      o_acc[mt][dt][3] * scale_w[mt]};
      for (int w = 0; w < R - 1; ++w) {
        const float4 o = *reinterpret_cast<const float4*>(
            xo + ((static_cast<size_t>(head_local * (R - 1) + w) * MTQ + mt) * 16 + c16) * HD + dt * 16 + 4 * g);
        acc[dt][0] += o.x;
        acc[dt][1] += o.y;
        acc[dt][2] += o.z;
        acc[dt][3] += o.w;
      }
    }
    if (P.direct) {
      store_final(mt, acc, l_all[mt] > 0.0f ? 1.0f / l_all[mt] : 0.0f, q0 + pos, h * G + gq);   // out_scale is in scale_w
      if (g == 0 && P.mark_final) mark_row_final(grow);
      continue;
    }
    float* op = P.ws_o + (static_cast<int64_t>(bpart) * P.total_rows + grow) * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      *reinterpret_cast<float4*>(op + dt * 16) = make_float4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]);
    if (g == 0) {
      float* mp = P.ws_ml + (static_cast<int64_t>(bpart) * P.total_rows + grow) * 2;
      mp[0] = m_all[mt] * kLn2;
      mp[1] = l_all[mt];
      if (by == 0) mark_unused_parts(P, grow);
    }
  }
}

#undef AIC_PSTAMP

// Host-partitioned calls (m_groups == 1): the short list holds requests of up to 32 query rows, and every workgroup
// takes the one- or the two-row-tile form of the body by the rows of ITS request.  Most requests of a step have no draft
// or a k = 3 draft (<= 16 rows with Hq/Hkv <= 4); suffix drafts of 4-7 tokens (17-32 rows; 46 % of all suffix drafts in
// the r02 bench, `bench.py --qlen-hist`) used to go through the shared-tile long-draft body, where a request costs ~6 us
// per layer whatever its length, against ~2.7 us for a workgroup of this streaming body.
template <int HPW, bool KV8, int NW, int HD = kD>
__device__ __forceinline__ void verify_attn_body_dual(const AttnParams& P, uint4* v_lds_raw, const int bx, const int by) {
  if (bx >= P.n_items) return;             // m_groups == 1: item == bx
  const int ridx = bx / (P.num_kv_heads / HPW);
  const int req = __builtin_amdgcn_readfirstlane(P.req_list ? P.req_list[ridx] : ridx);
  const int q_len = __builtin_amdgcn_readfirstlane(P.query_start_loc[req + 1] - P.query_start_loc[req]);
  if (q_len * (P.num_q_heads / P.num_kv_heads) <= 16)
    verify_attn_body<1, HPW, KV8, NW, HD>(P, v_lds_raw, bx, by);
  else
    verify_attn_body<2, HPW, KV8, NW, HD>(P, v_lds_raw, bx, by);
}

// MTQ = 0: the per-workgroup choice above
template <int MTQ, int HPW, bool KV8, int NW = 4, int HD = kD>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu((MTQ == 1 && NW == 4) ? 3 : 2, (MTQ == 1 && NW == 4) ? 3 : 2))) verify_attn_kernel(AttnParams P) {
  __shared__ uint4 v_lds_raw[ShortLds<MTQ == 0 ? 2 : MTQ, NW, HD>::kU4];
  if constexpr (MTQ == 0)
    verify_attn_body_dual<HPW, KV8, NW, HD>(P, v_lds_raw, blockIdx.x, blockIdx.y);
  else
    verify_attn_body<MTQ, HPW, KV8, NW, HD>(P, v_lds_raw, blockIdx.x, blockIdx.y);
}

// One 32-token KV tile (K and V images in LDS at kb / vb, v_tile_off layout) against the NT row tiles of a wave
// of the long-draft kernels: all NT x 8 score MFMAs first, then a branch-free soft-max over the NT tiles (the
// scale folded into the exponent's fma, row maxima by VALU lane swaps, row sums kept as per-lane partials), then
// the PV MFMAs.  Straight-line code: with one soft-max call per row tile the calls' branches kept the MFMA
// chains and the VALU work of different row tiles from overlapping, and the wave (alone on its SIMD) ran at the
// sum of every latency.
template <int NT, int RT, int D, bool TR, int NP, typename Dma>
__device__ __forceinline__ void long_tile_compute(unsigned ka0, unsigned va0, unsigned slot_off,
                                                  const uint4 (&qf)[RT][D / 32],
                                                  const bool (&row_ok)[RT], const int (&row_pos)[RT],
                                                  float (&m_run)[RT], float (&m_use)[RT], float (&thr)[RT], f32x2 (&l_run)[RT],
                                                  f32x4 (&o_acc)[RT][D / 16], int tt, int t_end, int ctx, int q_len, int wnd,
                                                  float scale_log2, float inv_scale, int g, int c16, const Dma& dma,
                                                  uint64_t* cyc = nullptr) {
    // ka0 / va0: this lane's LDS byte address of its first K fragment (k-step 0, token c16; 16 + c16 is 16 rows on) and of its
    // first transposed V read (output tile 0) in ring slot 0.  The images' chunk swizzle is an XOR of the 16-byte chunk index
    // with a function of the row, and the k-step / output tile sits in chunk bits the row function only XORs: k-step s is
    // address ^ 64 s, output tile dt address ^ 32 dt (the LDS array is 1 KiB aligned) — two registers instead of twelve.
    // dma(i), i < NP: the next tile's LDS-DMA pieces of this wave, issued one by one between the row tiles' soft-max blocks
    // (VALU-only stretches) instead of back to back at the loop head
    // TR (aic_debug_attn_phase_trace on a long-only call): shader-clock cycles of this wave per phase, summed over its tiles —
    // cyc[1] K fragments read + score MFMAs issued, cyc[2] soft-max (waits for the scores), cyc[3] V reads + PV MFMAs issued
    uint64_t c0 = 0;
    if constexpr (TR) c0 = __builtin_amdgcn_s_memtime();
    constexpr int DS = D / 32, DT = D / 16;   // k-steps of the score MFMAs, 16-wide output tiles
    constexpr unsigned kHalf = 16 * D * 2;    // 16 token rows further on in an image
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(3))) * lds_u4;
    u32x4 kf[2][DS];
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const unsigned a = (ka0 + slot_off) ^ (64u * s);
      kf[0][s] = *reinterpret_cast<lds_u4>(a);
      kf[1][s] = *reinterpret_cast<lds_u4>(a + kHalf);
    }

    f32x4 st[NT][2];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int th = 0; th < 2; ++th) st[mt][th] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DS; ++s)
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int th = 0; th < 2; ++th)
          st[mt][th] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[th][s]),
                                                               __builtin_bit_cast(bf16x8, qf[mt][s]), st[mt][th], 0, 0, 0);
    if constexpr (TR) {
      const uint64_t c1 = __builtin_amdgcn_s_memtime();
      cyc[1] += c1 - c0;
      c0 = c1;
    }
    const bool need_mask = tt + kTile > t_end || tt + kTile > ctx - q_len + 1 || tt < ctx - wnd;
    if (need_mask) {
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        const int limit = ctx - q_len + row_pos[mt];
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int tok = tt + 16 * th + 4 * g + e;
            if (!(row_ok[mt] && tok < t_end && tok <= limit && tok > limit - wnd)) st[mt][th][e] = -INFINITY;
          }
      }
    }
    // The soft-max of this body is VALU-bound (tools/microbench.py longphases: of 4485 cycles per KV tile on the wave with
    // three row tiles, 1937 were this block, 1311 the 72 MFMAs), so the row maximum is LAZY: a row keeps the maximum m it
    // last adopted while no score of the tile exceeds it by more than kLazySlack (in the exponent's base-2 units: weights
    // stay below 2^kLazySlack, far inside f32 and bf16 range; the hi/lo split of P keeps relative precision at any
    // magnitude).  The common tile then costs one max3 chain and one compare per row tile — no cross-lane exchange, no
    // alpha, no rescale; when some lane of the wave sees a larger score, every row tile of the wave takes the exact path.
    float lmax[NT];
    bool over = false;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      const f32x4 &a = st[mt][0], &b = st[mt][1];
      lmax[mt] = max3f(max3f(max3f(a[0], a[1], a[2]), a[3], b[0]), b[1], max_nc(b[2], b[3]));
      over = over || lmax[mt] > thr[mt];
    }
    if (__any(over)) {
      float alpha[NT];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        const float tmax = rowgroup_max(lmax[mt]);
        const float m_new = fmaxf(m_run[mt], tmax * scale_log2);
        // a row with nothing visible yet keeps m = -inf; exponents are then taken against 0 (all scores are -inf)
        m_use[mt] = m_new == -INFINITY ? 0.0f : m_new;
        alpha[mt] = __builtin_amdgcn_exp2f(m_run[mt] - m_use[mt]);
        m_run[mt] = m_new;
        thr[mt] = __builtin_fmaf(m_new, inv_scale, kLazySlack * inv_scale);
        l_run[mt] *= alpha[mt];
      }
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o_acc[mt][dt] *= alpha[mt];
    }
    // (P as ONE bf16 operand with the row sum taken over the rounded weights — what kernels that feed P to the matrix unit
    // in the value dtype do — was measured: long-only B=16 x 33 134 -> 103 us together with two workgroups per CU, but a
    // 40-token context then misses the 1e-3 tolerance: few terms, nothing averages the 2^-9 weight errors out.)
    bf16x8 pfrag[NT], pfrag_lo[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      // pieces i with i * (NT + 1) / NP == mt go out before this row tile's block, the rest after the last one.  Nothing
      // orders a volatile asm against arithmetic, so each piece is tied into the data flow: it consumes the results of the block
      // before it and the block after it reads its first score through it.
      if constexpr (NP > 0) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
          if (i * (NT + 1) / NP == mt) {
            if (mt == 0) asm volatile("" : "+v"(st[0][0][0]) : "v"(lmax[0]));
            else asm volatile("" : "+v"(st[mt][0][0]) : "v"(pfrag[mt > 0 ? mt - 1 : 0]), "v"(pfrag_lo[mt > 0 ? mt - 1 : 0]));
            dma(i);
            asm volatile("" : "+v"(st[mt][0][0]));
          }
      }
      // (scalar f32 arithmetic on purpose: packed f32 VALU beside MFMAs costs more issue time than the two scalar
      // instructions it replaces — MI355X_MICROARCH.md, cycle constants)
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[mt][e >> 2][e & 3], scale_log2, -m_use[mt]));
        l_run[mt][e & 1] += pv[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pfrag[mt][e] = static_cast<__bf16>(pv[e]);
        pfrag_lo[mt][e] = static_cast<__bf16>(pv[e] - static_cast<float>(pfrag[mt][e]));
      }
    }
    if constexpr (NP > 0 && NT > 0) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (i * (NT + 1) / NP >= NT) {
          asm volatile("" : "+v"(pfrag[0]) : "v"(pfrag_lo[NT - 1]));
          dma(i);
          asm volatile("" : "+v"(pfrag[0]));
        }
    }
    if constexpr (TR) {
      asm volatile("" :: "v"(pfrag[0]), "v"(pfrag_lo[NT - 1]));
      const uint64_t c2 = __builtin_amdgcn_s_memtime();
      cyc[2] += c2 - c0;
      c0 = c2;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      typedef s16x4 __attribute__((address_space(3))) * lds_tr;
      const unsigned a = (va0 + slot_off) ^ (32u * dt);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr>(a));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr>(a + kHalf));
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      const bf16x8 vfrag = __builtin_bit_cast(bf16x8, both);
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pfrag[mt], o_acc[mt][dt], 0, 0, 0);
        o_acc[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pfrag_lo[mt], o_acc[mt][dt], 0, 0, 0);
      }
    }
    if constexpr (TR) cyc[3] += __builtin_amdgcn_s_memtime() - c0;
}

constexpr int kLongTilesPerWave = 3;

// ------------------------------------------------------------------------------------------------------
// Long-draft body (suffix drafts: up to 33 query positions = 132 rows at G = 4).  One kv head and one token
// range per workgroup; wave w owns row tiles w, w+4, w+8 (up to 192 rows) against a 32-token K/V tile that is in
// LDS once per workgroup, so the KV bytes of a long request are read once instead of once per 16-row group.
// Four waves and NO loader waves.  Two earlier forms are what this one answers: (1) compute waves loading their
// own tiles through registers ran at HBM latency per tile (hipcc drains vmcnt(0) at the loop head whenever a
// register set is re-armed across the back edge); (2) four compute + four loader waves fixed that, but VGPR
// allocation is per kernel, so the loader waves held half of a CU's register file for nothing and a workgroup
// needed a whole EMPTY CU: beside the short-request kernel (one 4-wave workgroup per CU) it only started when
// that kernel drained, and the "overlapped" pair ran back to back.  Here each wave moves its quarter of every
// K/V tile with LDS-DMA (global_load_lds_dwordx4: no VGPR destination, 1 KiB per wave-instruction) into a ring
// of four tile images, three tiles in flight, counted vmcnt + one raw s_barrier per tile; at <= 256 VGPRs and
// 64 KiB of LDS a workgroup fits beside a short-request workgroup on the same CU and its MFMA / VALU work fills
// that body's memory stalls.  The bf16 image is the v_tile_off layout: the DMA writes lane-linear, so the row
// permutation and chunk swizzle are applied on the per-lane SOURCE address (cdna guide rule 21).
// The loads are inline asm: hipcc does not count them, the s_waitcnt below are the only vmcnt waits in the
// loop (the Q loads are drained before the first DMA is issued).
// ------------------------------------------------------------------------------------------------------
constexpr int kLongRing = 4;   // LDS tile images
constexpr int kLongAhead = 3;  // tiles in flight

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// the same with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit per-lane address arithmetic
__device__ __forceinline__ void glds16s(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

constexpr int kLong4LdsU4 = kLongRing * 2 * kTile * 16;

// Body of the co-resident long-draft kernel for workgroup (bx, by, bz); lds = kLong4LdsU4 uint4 of LDS.
// KV8 (fp8 e4m3 cache): the DMA ring holds the raw bytes (4 slots x {K, V} x 4 KiB); every wave converts its
// quarter of the NEXT tile into one of two bf16 images (tile_off layout, exact: every e4m3 value is a bf16 value)
// while the current one is being consumed, so the compute code is the bf16 one and there is still one barrier per
// tile.  k_scale folds into the soft-max scale, v_scale into the output.
// D = 64 (bf16 cache only): 128-byte rows, i.e. the DMA geometry of the fp8 case without the conversion (long drafts of a
// head-size-64 model; requests of up to 32 rows take the streaming body's HD = 64 form).
template <bool KV8, int D, bool TR = false>
__device__ __forceinline__ void verify_attn_long4_body(const AttnParams& P, uint4* lds, const int bx, const int by, const int bz) {
  static_assert(D == 128 || (D == 64 && !KV8), "head size 128, or 64 with a bf16 cache");
  uint64_t cyc[6] = {0, 0, 0, 0, 0, 0};   // TR: [0] issuing the next tile's DMA, [1..3] long_tile_compute, [4] vmcnt wait, [5] barrier
  uint64_t c_entry = 0, rt_entry = 0;
  if constexpr (TR) {
    c_entry = __builtin_amdgcn_s_memtime();
    rt_entry = __builtin_amdgcn_s_memrealtime();
  }
  constexpr int RT = kLongTilesPerWave;
  constexpr int DS = D / 32, DT = D / 16;
  constexpr bool ROW128 = KV8 || D == 64;          // a token row of the DMA source is 128 bytes (else 256)
  constexpr int kImg = kTile * D * 2;              // bytes of one bf16 K (or V) image
  constexpr int kImgBase = KV8 ? kLong4LdsU4 * 8 : 0;   // fp8: raw ring in the first half of the LDS, images behind it
  constexpr int kRawSlot = kTile * 256;            // fp8 raw ring slot: K 4 KiB + V 4 KiB
  char* const lds_b = reinterpret_cast<char*>(lds);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int Hkv = P.num_kv_heads, Hq = P.num_q_heads, G = Hq / Hkv;
  const int ridx = bx / Hkv;
  const int h = bx - ridx * Hkv;
  const int req = __builtin_amdgcn_readfirstlane(P.req_list ? P.req_list[ridx] : ridx);
  const int q0 = __builtin_amdgcn_readfirstlane(P.query_start_loc[req]);
  const int q_len = __builtin_amdgcn_readfirstlane(P.query_start_loc[req + 1]) - q0;
  const int ctx = __builtin_amdgcn_readfirstlane(P.seq_lens[req]);
  const int n_rows = q_len * G;
  const int row_base = bz * (4 * RT * 16);
  if (row_base >= n_rows) return;

  const int t_lo = window_begin(P, ctx, q_len);
  const int wnd = P.window > 0 ? P.window : (1 << 30);
  const int tiles_total = (ctx - t_lo + kTile - 1) / kTile;
  const int tiles_per_part = (tiles_total + P.n_splits - 1) / P.n_splits;
  const int t_begin = t_lo + by * tiles_per_part * kTile;
  const int t_end = min(ctx, t_begin + tiles_per_part * kTile);
  const int n_iter = t_begin < t_end ? (t_end - t_begin + kTile - 1) / kTile : 0;

  const int64_t kv_row = static_cast<int64_t>(Hkv) * D;
  const_i32_ptr btab = (const_i32_ptr)(P.block_table + static_cast<int64_t>(req) * P.max_blocks);
  const int bs = P.block_size;
  const int last_group = (ctx - 1) & ~15;

  // ---- query rows of this wave ----
  float scale_log2 = P.sm_scale * kLog2e, out_scale = 1.0f;
  if (KV8) {
    scale_log2 *= *P.k_scale;
    out_scale = *P.v_scale;
  }
  uint4 qf[RT][DS];
  int row_pos[RT];
  bool row_ok[RT];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt) {
    const int rr = row_base + (wave + 4 * mt) * 16 + c16;
    row_ok[mt] = rr < n_rows;
    const int rc = min(rr, n_rows - 1);
    const int pos = rc / G, gq = rc - pos * G;
    row_pos[mt] = pos;
    const uint16_t* qp = P.q + static_cast<int64_t>(q0 + pos) * P.q_stride + static_cast<int64_t>(h * G + gq) * D + 8 * g;
#pragma unroll
    for (int s = 0; s < DS; ++s) qf[mt][s] = *reinterpret_cast<const uint4*>(qp + 32 * s);
  }
  // the Q loads are the only compiler-counted vector loads: retire them before any DMA is in flight, otherwise
  // hipcc's wait for them (vmcnt(0), placed at their first use INSIDE the loop) would drain the ring every tile
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int s = 0; s < DS; ++s)
      asm volatile("" : "+v"(qf[mt][s].x), "+v"(qf[mt][s].y), "+v"(qf[mt][s].z), "+v"(qf[mt][s].w));

  // per row tile: m_run the maximum the row last adopted (scaled; -inf = nothing visible yet), m_use the value exponents are
  // taken against, thr the raw score above which the row must adopt a new maximum (long_tile_compute), l_run this lane's
  // share of the row's sum (two partial sums)
  const float inv_scale = 1.0f / scale_log2;
  float m_run[RT], m_use[RT], thr[RT];
  f32x2 l_run[RT];
  f32x4 o_acc[RT][DT];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt) {
    m_run[mt] = -INFINITY;
    m_use[mt] = 0.0f;
    thr[mt] = -INFINITY;
    l_run[mt] = f32x2{0.0f, 0.0f};
    if (P.sinks != nullptr && by == 0) {        // first partial of the row: starts from the sink
      const int rc = min(row_base + (wave + 4 * mt) * 16 + c16, n_rows - 1);
      m_run[mt] = m_use[mt] = P.sinks[h * G + (rc - (rc / G) * G)] * kLog2e;
      thr[mt] = __builtin_fmaf(m_run[mt], inv_scale, kLazySlack * inv_scale);
      l_run[mt][0] = g == 0 ? 1.0f : 0.0f;      // (a per-lane share of the row's sum: one lane carries the 1)
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o_acc[mt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int n_row_tiles = (min(n_rows - row_base, 4 * RT * 16) + 15) >> 4;
  const int my_tiles = __builtin_amdgcn_readfirstlane(n_row_tiles > wave ? (n_row_tiles - wave + 3) >> 2 : 0);

  const char* kc = reinterpret_cast<const char*>(P.k_cache);
  const char* vc = reinterpret_cast<const char*>(P.v_cache);
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) void*)(lds)));
  // ---- this wave's share of a tile ----
  // 256-byte rows (bf16, D = 128): LDS rows 4 w .. 4 w + 3 and 16 + 4 w .. (1 KiB each) of K and of V; LDS slot
  //     (row r, chunk slot c) holds chunk c ^ f(r) of token swap23(r)  (the inverse of v_tile_off)
  // 128-byte rows (fp8 raw bytes, or bf16 with D = 64): tokens 8 w .. 8 w + 7 (1 KiB of K, 1 KiB of V); lane =
  //     (token, 16-byte chunk); the D = 64 image XORs the chunk with the row, fp8 raw bytes stay linear
  const int r0 = 4 * wave + g;                                             // < 16: first page of the tile
  const int tk = ROW128 ? 8 * wave + (lane >> 3) : (r0 & ~12) | ((r0 & 4) << 1) | ((r0 & 8) >> 1);
  const int ch = KV8 ? (lane & 7) : D == 64 ? ((lane & 7) ^ (tk & 7)) : c16 ^ (((r0 & 3) << 2) | ((r0 >> 2) & 3));
  // (Everything per tile below is scalar, unsigned and 32-bit where it can be: one wave issues about one instruction every
  // four cycles whatever its kind, and the first count of this loop had 215 scalar instructions per tile beside 152 vector
  // ones — four emulated signed divisions and six 64-bit multiplies among them.)
  const int bsh = P.bs_shift;
  auto page_of = [&](unsigned f, unsigned& pg, unsigned& po) {
    if (bsh >= 0) {
      pg = f >> bsh;
      po = f & static_cast<unsigned>(bs - 1);
    } else {
      pg = f / static_cast<unsigned>(bs);
      po = f - pg * static_cast<unsigned>(bs);
    }
  };
  // The block-table entries of this workgroup's token range live in ONE register, lane i holding page pg_base + i (one
  // coalesced load covers 64 pages = 1024 tokens at 16-token pages); a tile takes its two with v_readlane.  Scalar loads
  // looked cheaper and were not: they share lgkmcnt with the LDS reads and return out of order, so the first fragment wait
  // of every tile (lgkmcnt(0)) also waited out their round trip — ~600 cycles per tile on every wave (longphases).  A range
  // longer than 64 pages refills the register (a plain load the compiler waits for: once per 32 tiles).
  unsigned pg_base, po_unused;
  page_of(static_cast<unsigned>(min(t_begin, last_group)), pg_base, po_unused);
  int vpages = btab[min(static_cast<int>(pg_base) + lane, P.max_blocks - 1)];
  // (`steady`: tile jt is not the last of its range — all 32 tokens exist, nothing to clamp)
  auto pages = [&](int jt, int& pa, int& pb, auto steady) {
    const int tt = t_begin + jt * kTile;
    unsigned pg0, pg1, po;
    if constexpr (ROW128) {
      page_of(static_cast<unsigned>(min(tt + 16 * (wave >> 1), last_group)), pg0, po);
      pg1 = pg0;
    } else if constexpr (decltype(steady)::value) {
      page_of(static_cast<unsigned>(tt), pg0, po);
      page_of(static_cast<unsigned>(tt + 16), pg1, po);
    } else {
      page_of(static_cast<unsigned>(min(tt, last_group)), pg0, po);
      page_of(static_cast<unsigned>(min(tt + 16, last_group)), pg1, po);
    }
    if (pg1 - pg_base >= 64u) {
      pg_base = pg0;
      vpages = btab[min(static_cast<int>(pg_base) + lane, P.max_blocks - 1)];
      asm volatile("" : "+v"(vpages));   // the compiler's wait for this load stays inside the branch
    }
    pa = __builtin_amdgcn_readlane(vpages, static_cast<int>(pg0 - pg_base));
    pb = __builtin_amdgcn_readlane(vpages, static_cast<int>(pg1 - pg_base));
  };
  // A tile's DMA = kPieces LDS-DMA instructions per wave (1 KiB each).  Their source addresses are a wave-uniform base per
  // 16-token group (page, row within the page, kv head: scalar arithmetic) plus a per-lane byte offset that is the same for
  // every tile whose 32 tokens all exist (`loff`); only the context's last tile clamps rows per lane.  (The cycle accounting
  // had 770-870 cycles per tile and wave in this block when it computed 64-bit per-lane addresses and issued the four
  // instructions back to back: an LDS-DMA instruction among busy phases costs 100-185 cycles to issue, 25-60 in a VALU-only
  // stretch — MI355X_MICROARCH.md, cycle constants — so the pieces are handed to long_tile_compute one by one.)
  constexpr int kPieces = ROW128 ? 2 : 4;
  constexpr int EB = KV8 ? 1 : 2;   // bytes per cache element
  const unsigned row_bytes = static_cast<unsigned>(kv_row) * EB;
  // Who moves what (256-byte rows only).  With 4 k + 1 or 4 k + 2 row tiles the waves that hold one tile more set the pace of
  // every iteration while the others wait at the barrier (cycle accounting: 450-800 cycles per tile), and a wave's DMA duty —
  // page lookup, four base addresses, four LDS-DMA instructions — is ~65 instructions per tile: those waves issue NOTHING,
  // and their quarters of the tile are moved by the lighter waves (dma_mode 0 = none, 1 = own quarter, 2 = own + donated
  // pieces).  4 k + 1: wave 0's quarter goes to waves 1 (K lo, K hi), 2 (V lo), 3 (V hi); 4 k + 2: wave 2 moves wave 0's
  // quarter, wave 3 wave 1's.
  int dma_mode = 1, donor_q = 0;
  unsigned donor_mask = 0, own_mask = 0xfu;
  if constexpr (!ROW128) {
    const int extra = n_row_tiles & 3;
    const int pat = P.long_dma;   // 0 symmetric; 1: heavy waves issue nothing; 2: heavy waves keep their K pieces; 3: 4 k + 1 only, wave 0 keeps K lo
    if (extra == 1 && pat > 0) {
      if (wave == 0) {
        own_mask = pat == 1 ? 0u : pat == 2 ? 0x3u : 0x1u;
        dma_mode = own_mask ? 1 : 0;
      } else {
        const unsigned give = 0xfu & ~(pat == 1 ? 0u : pat == 2 ? 0x3u : 0x1u);            // what wave 0 hands over
        const unsigned want = wave == 1 ? 0x3u : wave == 2 ? 0x4u : 0x8u;                    // K lo + K hi / V lo / V hi
        donor_mask = give & want;
        dma_mode = donor_mask ? 2 : 1;
      }
    } else if (extra == 2 && pat > 0) {
      if (wave < 2) {
        own_mask = pat == 1 ? 0u : 0x3u;
        dma_mode = own_mask ? 1 : 0;
      } else {
        dma_mode = 2;
        donor_q = wave - 2;
        donor_mask = pat == 1 ? 0xfu : 0xcu;
      }
    }
  }
  const int my_pieces = __builtin_amdgcn_readfirstlane(ROW128 ? kPieces : __builtin_popcount(own_mask) + __builtin_popcount(donor_mask));
  // per-lane byte offset of quarter q's piece from the first row of its 16-token group (rows 4 q .. 4 q + 3 of the image)
  auto lane_off = [&](int q, int f, int tt_half, bool clamp) -> unsigned {
    const int rq = 4 * q + g;
    const int tq = (rq & ~12) | ((rq & 4) << 1) | ((rq & 8) >> 1);
    const int cq = c16 ^ (((rq & 3) << 2) | ((rq >> 2) & 3));
    const int rel = clamp ? min(tt_half + tq, ctx - 1) - f : tq;
    return static_cast<unsigned>(rel) * row_bytes + 16u * cq;
  };
  const unsigned loff = ROW128 ? static_cast<unsigned>(tk - 16 * (wave >> 1)) * row_bytes + 16u * ch
                               : static_cast<unsigned>(tk) * row_bytes + 16u * ch;
  const unsigned loff_donor = ROW128 ? 0u : lane_off(donor_q, 0, 0, false);
  struct TileDma {
    const char *k0, *k1, *v0, *v1;
    unsigned l0, slot;      // ROW128: this wave's lane offset; slot: LDS address of the tile's image (256-byte rows: without the wave's rows)
    int tt, f0, f1;         // 256-byte rows: the tile's first token and its two groups' first tokens
    bool full, on;
  } td;
  td.on = false;
  const uint64_t stride_bytes = static_cast<uint64_t>(P.block_stride) * EB;
  const unsigned head_off = static_cast<unsigned>(h * D);
  // byte offset of (page, first token f of a 16-token group) for this kv head
  auto group_base = [&](int page, unsigned f) -> uint64_t {
    unsigned pg, po;
    page_of(f, pg, po);
    return static_cast<uint64_t>(static_cast<unsigned>(page)) * stride_bytes + (po * static_cast<unsigned>(kv_row) + head_off) * EB;
  };
  auto prep = [&](int jt, int pa, int pb, auto steady) {
    constexpr bool kSteady = decltype(steady)::value;
    const int tt = t_begin + jt * kTile;
    td.on = true;
    td.full = kSteady || tt + kTile <= ctx;
    if constexpr (ROW128) {
      const int f = min(tt + 16 * (wave >> 1), last_group);   // tokens 0-15 / 16-31 of the tile: one page per wave
      const uint64_t b = group_base(pa, static_cast<unsigned>(f));
      td.k0 = kc + b;
      td.v0 = vc + b;
      td.l0 = td.full ? loff : static_cast<unsigned>(min(tt + tk, ctx - 1) - f) * row_bytes + 16u * ch;
      td.slot = lds0 + static_cast<unsigned>(jt & (kLongRing - 1)) * kRawSlot + 1024 * wave;
    } else {
      td.tt = tt;
      td.f0 = kSteady ? tt : min(tt, last_group);
      td.f1 = kSteady ? tt + 16 : min(tt + 16, last_group);
      const uint64_t b0 = group_base(pa, static_cast<unsigned>(td.f0)), b1 = group_base(pb, static_cast<unsigned>(td.f1));
      td.k0 = kc + b0;
      td.k1 = kc + b1;
      td.v0 = vc + b0;
      td.v1 = vc + b1;
      td.slot = lds0 + static_cast<unsigned>(jt & (kLongRing - 1)) * (2 * kImg);
    }
  };
  // piece i of the tile prepared last (i is a compile-time constant at every call site): 0-3 = this wave's quarter
  // (K lo, K hi, V lo, V hi), 4-7 = the same of the donor quarter where donor_mask has the bit
  auto piece = [&](int i) {
    if (!td.on) return;
    if constexpr (ROW128) {
      if (i == 0) glds16s(td.k0, td.l0, td.slot);
      else if (i == 1) glds16s(td.v0, td.l0, td.slot + kRawSlot / 2);
    } else {
      const int kind = i & 3;
      const bool donated = i >= 4;
      if (!(((donated ? donor_mask : own_mask) >> kind) & 1u)) return;
      const int q = donated ? donor_q : wave;
      const bool hi = kind & 1;
      unsigned off = donated ? loff_donor : loff;
      if (!td.full) off = lane_off(q, hi ? td.f1 : td.f0, td.tt + (hi ? 16 : 0), true);   // the context's last tile: rows clamp
      const char* base = kind == 0 ? td.k0 : kind == 1 ? td.k1 : kind == 2 ? td.v0 : td.v1;
      glds16s(base, off, td.slot + 1024u * q + (hi ? 4096u : 0u) + (kind >= 2 ? static_cast<unsigned>(kImg) : 0u));
    }
  };
  auto issue = [&](int jt, int pa, int pb) {
    if (dma_mode == 0) return;
    prep(jt, pa, pb, std::false_type{});
#pragma unroll
    for (int i = 0; i < 2 * kPieces; ++i) piece(i);
    td.on = false;
  };
  // fp8: this wave's quarter of raw tile jt -> bf16 image jt & 1
  auto convert = [&](int jt) {
    const char* raw = lds_b + (jt & (kLongRing - 1)) * kRawSlot + 1024 * wave + 16 * lane;
    const uint4 k8 = *reinterpret_cast<const uint4*>(raw);
    const uint4 v8 = *reinterpret_cast<const uint4*>(raw + kRawSlot / 2);
    char* kb = lds_b + kImgBase + (jt & 1) * (2 * kImg);
    char* vb = kb + kImg;
    *reinterpret_cast<bf16x8*>(kb + tile_off<D>(tk, 2 * ch)) = fp8x8_to_bf16x8(k8.x, k8.y);
    *reinterpret_cast<bf16x8*>(kb + tile_off<D>(tk, 2 * ch + 1)) = fp8x8_to_bf16x8(k8.z, k8.w);
    *reinterpret_cast<bf16x8*>(vb + tile_off<D>(tk, 2 * ch)) = fp8x8_to_bf16x8(v8.x, v8.y);
    *reinterpret_cast<bf16x8*>(vb + tile_off<D>(tk, 2 * ch + 1)) = fp8x8_to_bf16x8(v8.z, v8.w);
  };
  // wait until at most `tiles` of this wave's most recent tile loads are still in flight (DMAs retire in order); a tile is
  // my_pieces instructions of this wave (0, 2, 4, 5, 6 or 8)
  auto wait_tiles = [&](int tiles) {
    const int n = tiles * my_pieces;
    if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  constexpr int kAhead = KV8 ? kLongRing : kLongAhead;   // tiles issued before the loop
  asm volatile("" : "+v"(vpages));   // (a compiler-counted load, like Q: retired before the first DMA is in flight)
#pragma unroll
  for (int d = 0; d < kAhead; ++d)
    if (d < n_iter) {
      int pa, pb;
      pages(d, pa, pb, std::false_type{});
      issue(d, pa, pb);
    }
  if (KV8 && n_iter > 0) {
    wait_tiles(min(n_iter, kAhead) - 1);
    __builtin_amdgcn_s_barrier();   // tile 0 is in the ring
    asm volatile("" ::: "memory");
    convert(0);
  }

  // LDS addresses of this lane's first K fragment and first transposed V read in image slot 0 (see long_tile_compute)
  const unsigned ka0 = lds0 + kImgBase + tile_off<D>(c16, g);
  const unsigned va0 = lds0 + kImgBase + kImg + tile_off<D>(4 * g + (c16 >> 2), (c16 & 3) >> 1) + 8 * (c16 & 1);
  uint64_t c_loop = 0;
  if constexpr (TR) c_loop = __builtin_amdgcn_s_memtime();
  // One iteration = one KV tile.  `steady` iterations (bf16 cache: those whose issued tile it + 3 is not the last of the range)
  // run with constant waits and unclamped addresses; the last four take the general form.  np_tag: the DMA pieces this wave
  // may issue per tile (0: none, kPieces: its quarter, 2 kPieces: its quarter + donated pieces).
  auto tile_iter = [&](int it, auto nt_tag, auto np_tag, auto steady) {
    constexpr int NT = decltype(nt_tag)::value;
    constexpr int NPT = decltype(np_tag)::value;
    constexpr bool kSteady = decltype(steady)::value;
    uint64_t c_top = 0;
    if constexpr (TR) c_top = __builtin_amdgcn_s_memtime();
    if constexpr (KV8) {
      // tile it + 1 must have landed (it is converted below); tiles it + 2, it + 3 may still be in flight
      wait_tiles(min(2, max(0, n_iter - it - 2)));
      __builtin_amdgcn_s_barrier();  // bf16 image of tile `it` complete; raw slot it & 3 and image (it + 1) & 1 are free
      asm volatile("" ::: "memory");
      if (it + kLongRing < n_iter) {
        int pa, pb;
        pages(it + kLongRing, pa, pb, std::false_type{});
        issue(it + kLongRing, pa, pb);
      }
      if (it + 1 < n_iter) convert(it + 1);
    } else {
      // tile `it` has landed when at most the tiles issued after it are still outstanding
      if constexpr (NPT > 0) {
        if constexpr (!kSteady) {
          wait_tiles(min(2, n_iter - 1 - it));
        } else {
          if (my_pieces == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else wait_tiles(2);
        }
      }
      if constexpr (TR) {
        const uint64_t c = __builtin_amdgcn_s_memtime();
        cyc[4] += c - c_top;
        c_top = c;
      }
      __builtin_amdgcn_s_barrier();  // every wave's quarter of tile `it` is in LDS; ring slot (it - 1) & 3 is free
      asm volatile("" ::: "memory");
      if constexpr (TR) {
        const uint64_t c = __builtin_amdgcn_s_memtime();
        cyc[5] += c - c_top;
        c_top = c;
      }
      td.on = false;
      if constexpr (NPT > 0) {
        if (kSteady || it + kLongAhead < n_iter) {
          int pa, pb;
          pages(it + kLongAhead, pa, pb, steady);
          prep(it + kLongAhead, pa, pb, steady);
        }
        if constexpr (NT == 0) {      // a wave without rows: nothing to spread the pieces over
#pragma unroll
          for (int i = 0; i < NPT; ++i) piece(i);
        }
      }
    }
    if constexpr (TR) cyc[0] += __builtin_amdgcn_s_memtime() - c_top;
    if constexpr (NT > 0) {
      const unsigned slot_off = static_cast<unsigned>(KV8 ? (it & 1) : (it & (kLongRing - 1))) * (2 * kImg);
      auto dma = [&](int i) {
        if constexpr (!KV8) piece(i);   // (fp8: the tile's DMA went out at the loop head, before the conversion)
      };
      long_tile_compute<NT, RT, D, TR, KV8 ? 0 : NPT>(ka0, va0, slot_off, qf, row_ok, row_pos, m_run, m_use, thr, l_run, o_acc,
                                                       t_begin + it * kTile, t_end, ctx, q_len, wnd, scale_log2, inv_scale,
                                                       g, c16, dma, cyc);
    }
  };
  auto tile_loop_np = [&](auto nt_tag, auto np_tag) {
    int it = 0;
    if constexpr (!KV8) {
      const int n_steady = n_iter - 1 - kLongAhead;   // issued tile it + 3 <= n_iter - 2
      for (; it < n_steady; ++it) tile_iter(it, nt_tag, np_tag, std::true_type{});
    }
    for (; it < n_iter; ++it) tile_iter(it, nt_tag, np_tag, std::false_type{});
  };
  auto tile_loop = [&](auto nt_tag) {
    constexpr int NT = decltype(nt_tag)::value;
    if constexpr (ROW128) {
      tile_loop_np(nt_tag, std::integral_constant<int, kPieces>{});
    } else {
      // (a wave that issues nothing holds at least one row tile; one that takes donated pieces at most RT - 1)
      if (NT >= 1 && dma_mode == 0) {
        if constexpr (NT >= 1) tile_loop_np(nt_tag, std::integral_constant<int, 0>{});
      } else if (NT < RT && dma_mode == 2) {
        if constexpr (NT < RT) tile_loop_np(nt_tag, std::integral_constant<int, 2 * kPieces>{});
      } else {
        tile_loop_np(nt_tag, std::integral_constant<int, kPieces>{});
      }
    }
  };
  if (my_tiles >= 3) {
    tile_loop(std::integral_constant<int, 3>{});
  } else if (my_tiles == 2) {
    tile_loop(std::integral_constant<int, 2>{});
  } else if (my_tiles == 1) {
    tile_loop(std::integral_constant<int, 1>{});
  } else {
    tile_loop(std::integral_constant<int, 0>{});  // no rows of its own: the wave still moves its quarter of the tiles
  }

  if constexpr (TR) {
    if (P.ptrace && lane == 0) {
      const uint64_t c_end = __builtin_amdgcn_s_memtime();
      const int nx_ = P.ptrace_stride & 0xffff, ny_ = P.ptrace_stride >> 16;   // the long part's grid (x, y): set by the host
      int64_t* r = P.ptrace + (((static_cast<int64_t>(bz) * ny_ + by) * nx_ + bx) * 4 + wave) * 16;
      r[0] = static_cast<int64_t>(c_loop - c_entry);   // prologue: geometry, Q rows, first tiles issued
      r[1] = static_cast<int64_t>(c_end - c_loop);     // the tile loop
      r[2] = static_cast<int64_t>(cyc[0]);
      r[3] = static_cast<int64_t>(cyc[1]);
      r[4] = static_cast<int64_t>(cyc[2]);
      r[5] = static_cast<int64_t>(cyc[3]);
      r[6] = (static_cast<int64_t>(n_iter) << 8) | my_tiles;
      r[7] = static_cast<int64_t>(__builtin_amdgcn_s_memrealtime());
      r[8] = static_cast<int64_t>(cyc[4]);
      r[9] = static_cast<int64_t>(cyc[5]);
      r[10] = static_cast<int64_t>(rt_entry);
      r[11] = static_cast<int64_t>(c_end - c_entry);
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      r[12] = static_cast<int64_t>(hw) | (static_cast<int64_t>(xcc) << 32);
    }
  }
  float l_tot[RT];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt) l_tot[mt] = rowgroup_sum(l_run[mt][0] + l_run[mt][1]);
#pragma unroll
  for (int mt = 0; mt < RT; ++mt) {
    if (!row_ok[mt]) continue;
    const int rr = row_base + (wave + 4 * mt) * 16 + c16;
    const int pos = rr / G, gq = rr - pos * G;
    const int64_t grow = static_cast<int64_t>(q0 + pos) * Hq + h * G + gq;
    float* op = P.ws_o + (static_cast<int64_t>(by) * P.total_rows + grow) * D + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      *reinterpret_cast<float4*>(op + dt * 16) =
          make_float4(o_acc[mt][dt][0] * out_scale, o_acc[mt][dt][1] * out_scale, o_acc[mt][dt][2] * out_scale,
                      o_acc[mt][dt][3] * out_scale);
    if (g == 0) {
      float* mp = P.ws_ml + (static_cast<int64_t>(by) * P.total_rows + grow) * 2;
      mp[0] = m_run[mt] * kLn2;
      mp[1] = l_tot[mt];
      if (by == 0) mark_unused_parts(P, grow);
    }
  }
}

template <bool KV8, int D = 128, bool TR = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) verify_attn_long4_kernel(AttnParams P) {
  __shared__ __attribute__((aligned(1024))) uint4 lds[kLong4LdsU4];
  verify_attn_long4_body<KV8, D, TR>(P, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Short requests and long drafts of one call in ONE launch: workgroups [0, n_long_wg) run the long-draft body,
// the rest the short-request body.  With two launches on two streams every layer paid a fork and a join event
// (two extra barrier packets, ~10 us of idle queue each on this runtime); one grid has neither, the long-draft
// workgroups still sit beside the short ones on the CUs (both bodies are 4 waves, <= 256 VGPRs, 64 KiB LDS: two
// workgroups per CU), and they come first in the grid so that they are placed before the CUs fill up.
// The long part is padded to a multiple of 8 workgroups (idle ones), which keeps the short body's XCD-aware item mapping.
template <int HPW, bool KV8, int MTQ, bool TR = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
verify_attn_pair_kernel(AttnParams PS, AttnParams PL, int n_long_wg, int n_long_pad, int long_x, int long_y, int short_x) {
  constexpr int kShortU4 = ShortLds<MTQ == 0 ? 2 : MTQ, 4>::kU4;
  __shared__ __attribute__((aligned(1024))) uint4 lds[kLong4LdsU4 > kShortU4 ? kLong4LdsU4 : kShortU4];
  const int b = blockIdx.x;
  int64_t t0 = 0;
  if (PS.trace) t0 = static_cast<int64_t>(__builtin_amdgcn_s_memrealtime());
  if (b < n_long_pad) {
    if (b < n_long_wg) {
      const int x = b % long_x, yz = b / long_x;
      const int y = yz % long_y, z = yz / long_y;
      verify_attn_long4_body<KV8, 128, TR>(PL, lds, x, y, z);   // (TR: the diagnostic cycle account, tools/microbench.py mixphases)
    }
  } else {
    const int sb = b - n_long_pad;
    __builtin_amdgcn_s_setprio(3);   // the memory-bound short body sets the launch's end: its waves issue first
    if constexpr (MTQ == 0)
      verify_attn_body_dual<HPW, KV8, 4>(PS, lds, sb % short_x, sb / short_x);
    else
      verify_attn_body<MTQ, HPW, KV8, 4>(PS, lds, sb % short_x, sb / short_x);
  }
  if (PS.trace) {
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      int64_t* r = PS.trace + static_cast<int64_t>(b) * 4;
      r[0] = t0;
      r[1] = static_cast<int64_t>(__builtin_amdgcn_s_memrealtime());
      r[2] = static_cast<int64_t>(hw) | (static_cast<int64_t>(xcc) << 32);
      r[3] = b < n_long_wg ? 1 : (b < n_long_pad ? 2 : 0);
    }
  }
}

// one wavefront per output row (token, q head): merge the n_parts partials
// (Merging inside the attention launch instead — tickets, last wave to arrive reads the slots back — was built and
// measured: no faster for short-only calls, 20-35 us slower with long drafts; profiles/r02_in_launch_merge_experiment.txt.
// r03: requesting the first four slots' values together with the {max, sum} pairs — one memory round trip per row instead
// of two — left the launch at 4.85 us (rocprofv3, bench); two to four rows per wave with every phase issued for all of them
// (a grid of about one workgroup per CU) took 5.2 / 9.8 / 11.2 us: one row per wave it stays.)
template <int D>
__global__ void __launch_bounds__(256)
verify_attn_combine_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml, int n_parts, int total_rows,
                           int num_q_heads, uint16_t* __restrict__ out, int64_t out_stride) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= total_rows) return;
  // lane p holds slot p's {max, sum} (n_parts <= 64): the weights of all slots come from one load and two wave
  // reductions, and the loop below only issues independent loads (a weight read inside the loop made every
  // slot's load wait for the previous one)
  float m_l = -INFINITY, l_l = 0.0f;
  if (lane < n_parts) {
    const float2 ml = *reinterpret_cast<const float2*>(ws_ml + (static_cast<int64_t>(lane) * total_rows + row) * 2);
    m_l = ml.x;
    l_l = ml.y;
  }
  // slot 0 with l = -1: the attention launch already wrote this row's finished value (direct mode); leave it alone
  if (__shfl(l_l, 0) == -1.0f) return;
  // a sentinel (l = -2) ends the row's slot list early (mark_unused_parts); what lies behind it is stale
  const unsigned long long stop = __ballot(l_l == -2.0f);
  if (stop) {
    const int n_row = __ffsll(stop) - 1;
    if (lane >= n_row) {
      m_l = -INFINITY;
      l_l = 0.0f;
    }
  }
  float M = m_l;
  for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
  const float w_l = m_l == -INFINITY ? 0.0f : __expf(m_l - M);
  float L = w_l * l_l;
  for (int off = 32; off > 0; off >>= 1) L += __shfl_xor(L, off);
  // empty slots (weight 0: never written, may hold anything) are redirected to a live slot so that the loads
  // stay unconditional and independent of each other; their weight of 0 drops the value
  const unsigned long long live = __ballot(w_l != 0.0f);
  const int p_ok = live ? __ffsll(live) - 1 : 0;
  float acc0 = 0.0f, acc1 = 0.0f;
  for (int p0 = 0; p0 < n_parts; p0 += 4) {
    if (((live >> p0) & 0xfull) == 0) continue;  // four empty slots (a short request's row in a call with long drafts)
    float w[4];
    float2 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pj = p0 + j;
      w[j] = pj < n_parts ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_l), pj & 63)) : 0.0f;
      const int pe = w[j] != 0.0f ? pj : p_ok;
      o[j] = make_float2(0.0f, 0.0f);
      if (D == 128 || 2 * lane < D)   // D = 64: the upper half of the wave only takes part in the reductions
        o[j] = *reinterpret_cast<const float2*>(ws_o + (static_cast<int64_t>(pe) * total_rows + row) * D + 2 * lane);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc0 += w[j] * o[j].x;
      acc1 += w[j] * o[j].y;
    }
  }
  const float inv = L > 0.0f ? 1.0f / L : 0.0f;
  if (!(L > 0.0f)) acc0 = acc1 = 0.0f;  // a row with nothing visible (cannot happen for a real token): zeros
  const int tok = row / num_q_heads, head = row - tok * num_q_heads;
  uint16_t* op = out + static_cast<int64_t>(tok) * out_stride + static_cast<int64_t>(head) * D + 2 * lane;
  const uint32_t packed = static_cast<uint32_t>(f32_to_bf16(acc0 * inv)) | (static_cast<uint32_t>(f32_to_bf16(acc1 * inv)) << 16);
  if (D == 128 || 2 * lane < D) *reinterpret_cast<uint32_t*>(op) = packed;
}

// Side stream for the long-draft kernel: a step's few long requests run beside the short-request kernel
// instead of behind it (fork / join with two events; capturable in a hipGraph).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int device = -1;
};
// one per device, created on first use and kept for the life of the process (a process drives one device in this
// design; a caller that switches devices gets that device's own stream and events, nothing is re-created or leaked)
static SideStream g_side[16];
static int side_stream(SideStream** out) {
  int dev = 0;
  AIC_HIP_TRY(hipGetDevice(&dev));
  AIC_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
  SideStream& S = g_side[dev];
  if (S.stream == nullptr) {
    AIC_HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    AIC_HIP_TRY(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
    AIC_HIP_TRY(hipEventCreateWithFlags(&S.join, hipEventDisableTiming));
    S.device = dev;
  }
  *out = &S;
  return AIC_OK;
}

// Token-range splits per (request, head group) item.  Measured on MI355X (B=64, ctx 4096, 128 items): what matters
// is not occupancy but BALANCE — 256 workgroups (one per CU) ran 171 us, 512 (two per CU) 181 us, 384 205 us, 768 197 us,
// 1024 204 us: a grid that is not a multiple of the CU count leaves a tail in which some CUs stream a second
// workgroup while the others idle.  With 59 requests (118 items) "one workgroup per CU at least" gave 3 splits = 354
// workgroups and 202 us, slower than 64 requests (179 us).  pick_splits therefore scores every split count by the tail
// it leaves — ceil(W / CUs) CU-rounds for W workgroups of equal size — plus a small charge per extra round and per
// split (partials to write and merge), and takes the cheapest.
// compute units of the current device (256 on an MI355X in SPX mode; fewer in a partitioned mode)
static int cu_count() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cached = n;
    else
      cached = 256;
  }
  return cached;
}

// `wgs_per_cu`: how many workgroups of the body should share a CU.  The short body streams: one per CU measured best.
// The shared-tile (long-draft) body is bound by its own MFMA + VALU chain per tile and wants a second workgroup on the
// CU to fill its stalls (alone, one wave per SIMD, it ran at 2 us per 32-token tile).
// `max_wgs` > 0: never more workgroups than that (the one-grid short + long launch keeps the short part to one
// workgroup per CU so that the long part has a resident slot on every CU).
static int pick_splits(int n_items, int max_seq_len, int min_tiles_per_split, int wgs_per_cu = 1, int max_wgs = 0) {
  const int kCUs = cu_count() * wgs_per_cu;
  const int max_tiles = (max_seq_len + kTile - 1) / kTile;
  int cap = std::max(1, max_tiles / std::max(1, min_tiles_per_split));
  if (cap > 64) cap = 64;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= cap; ++s) {
    if (s > 1 && max_wgs > 0 && static_cast<int64_t>(n_items) * s > max_wgs) break;
    const double wgs = static_cast<double>(n_items) * s;
    const double rounds = std::ceil(wgs / kCUs);
    // time ~ rounds * (work per workgroup) = rounds / s; normalised by the ideal n_items / kCUs
    const double tail = rounds * kCUs / wgs;
    const double cost = tail * (1.0 + 0.03 * (rounds - 1.0)) + 0.004 * s;
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}


// ---- launch recording (aic_verify_attention_layers) ---------------------------------------------------------------
// Every launch of the call path below goes through launch(): normally hipLaunchKernelGGL; while a Recorder is installed
// (thread-local) the launch is written down instead — kernel, geometry, argument bytes — so that a whole run of layers can
// be replayed as one HIP graph whose kernel nodes only have their parameters refreshed per step.
struct LaunchRec {
  void* func;
  dim3 grid, block;
  std::vector<char> blob;         // argument values, 16-byte aligned each
  std::vector<uint32_t> offs;     // their offsets in blob
};
struct Recorder {
  bool ok = true;                 // false: the call took a path that cannot be a plain kernel chain (side stream)
  size_t n = 0;                   // records in use; the vector and its blobs keep their capacity from call to call
  std::vector<LaunchRec> recs;
  void reset() {
    ok = true;
    n = 0;
  }
  LaunchRec& next() {
    if (n == recs.size()) recs.emplace_back();
    LaunchRec& r = recs[n++];
    r.blob.clear();
    r.offs.clear();
    return r;
  }
};
static thread_local Recorder* t_rec = nullptr;

template <typename... Params, typename... Args>
static void launch(void (*kernel)(Params...), dim3 grid, dim3 block, hipStream_t s, Args&&... args) {
  static_assert(sizeof...(Params) == sizeof...(Args), "argument count");
  if (t_rec == nullptr) {
    hipLaunchKernelGGL(kernel, grid, block, 0, s, static_cast<Params>(args)...);
    return;
  }
  LaunchRec& r = t_rec->next();
  r.func = reinterpret_cast<void*>(kernel);
  r.grid = grid;
  r.block = block;
  auto put = [&r](const auto& v) {
    const size_t off = (r.blob.size() + 15) & ~static_cast<size_t>(15);
    r.blob.resize(off + sizeof(v));
    std::memcpy(r.blob.data() + off, &v, sizeof(v));
    r.offs.push_back(static_cast<uint32_t>(off));
  };
  (put(static_cast<Params>(args)), ...);
}
static int launch_ok(const char* what) { return t_rec ? AIC_OK : launch_status(what); }

}  // namespace aic

using namespace aic;

static int64_t* g_attn_trace = nullptr;
static int g_attn_trace_cap = 0;
static int64_t* g_attn_ptrace = nullptr;   // aic_debug_attn_phase_trace
static int g_attn_ptrace_cap = 0;
static int g_long_dma = 1;    // aic_debug_attn_long_dma: AttnParams.long_dma
static int g_long_splits = 0; // aic_debug_attn_long_splits: split count of the long-draft part of a mixed call (0 = the default)
static int g_light_pct = 0;   // aic_debug_attn_light: weight of the light splits in percent (0 = the default, 100 = off)
static int g_force_hpw = 0, g_force_splits = 0;   // aic_debug_attn_layout (tools/microbench.py sweeps); 0 = choose
static int g_force_sequential = -1;               // aic_debug_attn_sequential: 1 / 0 = always / never the two-launch form of a mixed call

// Layout of the short body for a call whose query lengths the host knows: kv heads per workgroup (4, 2 or 1: the four
// waves are HPW head groups of R = 4 / HPW token ranges each, merged through LDS) and cross-workgroup token splits.
// One split means no partials at all: the workgroup that saw a row's whole context writes the finished row ("direct"),
// and a call without long drafts needs no combine launch.
// MEASURED (tools/microbench.py layout, profiles/r02_microbench_earlier.txt; 4224-token contexts, per call incl. combine):
//   64 requests: HPW 4 x 2 splits 172 us | HPW 2 x 1 split (direct, no combine) 179 us | HPW 1 x 1 split 202 us
//   32 requests: HPW 4 x 4 splits  91 us | HPW 2 x 2 splits 96 us | HPW 1 x 1 split (direct) 104 us | HPW 2 x 1 143 us
//   16 requests: HPW 4 x 8 splits  51 us | HPW 2 x 4 splits 53 us | HPW 1 x 2 splits 58 us
// i.e. what a workgroup reads per token (HPW x 256 contiguous bytes: DRAM page locality) outweighs the partial write and
// the combine launch that cross-workgroup splits cost.  So: as many heads per workgroup as divide the head count, splits
// by pick_splits, and the direct form whenever that leaves a single split (many requests, or few heads per rank).
// r03 also built the next step of that series — EIGHT waves per workgroup, one per kv head (512 threads, 2 KiB contiguous per
// token with a bf16 cache, 1 KiB with fp8; 215-242 VGPRs, no spills, two waves per SIMD) — and measured it on short-only calls
// (tools/microbench.py fp8layout, profiles/r03_layout8_experiment.txt): fp8 cache 64 requests 95.5 against 95.7 us, 32 requests
// 53.8 against 51.6; bf16 cache 179.5 against 171.5 and 98.0 against 91.3.  Not kept: the four-wave form already streams at
// what the memory system gives (6.2 TB/s bf16, 5.9 TB/s fp8 without the combine launch).
struct ShortLayout {
  int hpw, splits;
};
static ShortLayout pick_short_layout(int n_req, int num_kv_heads, int max_seq_len, int max_wgs) {
  int hpw = num_kv_heads % 4 == 0 ? 4 : (num_kv_heads % 2 == 0 ? 2 : 1);
  if (g_force_hpw && num_kv_heads % g_force_hpw == 0) hpw = g_force_hpw;
  const int R = 4 / hpw;
  int splits = pick_splits(n_req * (num_kv_heads / hpw), max_seq_len, 2 * R, 1, max_wgs);
  if (g_force_splits) splits = g_force_splits;
  return ShortLayout{hpw, splits};
}

extern "C" {

// debug: the next pair-kernel launches record per-workgroup start / end times and placement into `buf`
// ([capacity][4] int64, device); nullptr switches it off
int aic_debug_attn_trace(int64_t* buf, int capacity_wgs) {
  g_attn_trace = buf;
  g_attn_trace_cap = capacity_wgs;
  return AIC_OK;
}

// debug: the next short-only launches of host-partitioned calls record eight timestamps per workgroup (entry, request
// geometry known, first tile requested, first tile consumed, loop end, partials stored; [6] = tokens of the range << 32 |
// context) into buf[workgroup][8] (device int64); nullptr switches it off
int aic_debug_attn_phase_trace(int64_t* buf, int capacity_wgs) {
  g_attn_ptrace = buf;
  g_attn_ptrace_cap = capacity_wgs;
  return AIC_OK;
}

// debug: force the short body's heads per workgroup (4 / 2 / 1) and / or split count for host-partitioned calls
// (0 = let pick_short_layout choose); every setting computes the same result
// debug: weight (percent of a full split's tiles) of the short-body splits that share CUs with long-draft workgroups in the
// one-grid launch; 0 = the built-in value, 100 = equal splits
int aic_debug_attn_light(int pct) {
  g_light_pct = pct;
  return AIC_OK;
}

// debug: 1 = every mixed call as two launches (long part, then short part), 0 = the one-grid form whenever it fits,
// -1 = choose (the default); every setting computes the same result
int aic_debug_attn_sequential(int mode) {
  g_force_sequential = mode;
  return AIC_OK;
}

int aic_debug_attn_layout(int heads_per_wg, int splits) {
  g_force_hpw = heads_per_wg;
  g_force_splits = splits;
  return AIC_OK;
}

size_t aic_verify_attention_workspace_bytes(int num_tokens, int num_q_heads, int head_size, int num_splits_max) {
  const size_t rows = static_cast<size_t>(num_tokens) * num_q_heads;
  const size_t parts = static_cast<size_t>(num_splits_max > 0 ? num_splits_max : 64);
  return parts * rows * (static_cast<size_t>(head_size) + 2) * sizeof(float) + 256;
}

int aic_verify_attention_ex(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                            int64_t block_stride, int kv_dtype, const float* k_scale, const float* v_scale,
                            const int32_t* block_table, int max_blocks_per_seq, const int32_t* seq_lens,
                            const int32_t* query_start_loc, int batch, int num_tokens, int max_q_len, int num_q_heads,
                            int num_kv_heads, int head_size, int block_size, float sm_scale, void* out,
                            int64_t out_stride, void* workspace, size_t workspace_bytes, int max_seq_len,
                            const int32_t* short_reqs, int n_short, const int32_t* long_reqs, int n_long, void* stream) {
  return aic_verify_attention_win(q, q_stride, k_cache, v_cache, block_stride, kv_dtype, k_scale, v_scale, block_table,
                                  max_blocks_per_seq, seq_lens, query_start_loc, batch, num_tokens, max_q_len, num_q_heads,
                                  num_kv_heads, head_size, block_size, sm_scale, out, out_stride, workspace, workspace_bytes,
                                  max_seq_len, short_reqs, n_short, long_reqs, n_long, 0, nullptr, stream);
}

int aic_verify_attention_win(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                             int64_t block_stride, int kv_dtype, const float* k_scale, const float* v_scale,
                             const int32_t* block_table, int max_blocks_per_seq, const int32_t* seq_lens,
                             const int32_t* query_start_loc, int batch, int num_tokens, int max_q_len, int num_q_heads,
                             int num_kv_heads, int head_size, int block_size, float sm_scale, void* out,
                             int64_t out_stride, void* workspace, size_t workspace_bytes, int max_seq_len_full,
                             const int32_t* short_reqs, int n_short, const int32_t* long_reqs, int n_long,
                             int sliding_window, const float* sinks, void* stream) {
  if (batch == 0 || num_tokens == 0) return AIC_OK;
  AIC_REQUIRE(sliding_window >= 0, "sliding_window must be >= 0 (0 = none)");
  // what a request streams at most: the whole context, or — sliding window — the window plus the draft and tile slack
  const int max_seq_len = sliding_window > 0 ? std::min(max_seq_len_full, sliding_window + max_q_len + kTile) : max_seq_len_full;
  AIC_REQUIRE(q && k_cache && v_cache && block_table && seq_lens && query_start_loc && out && workspace,
              "null pointer argument to aic_verify_attention");
  AIC_REQUIRE(batch > 0 && num_tokens > 0 && max_q_len > 0 && num_q_heads > 0 && num_kv_heads > 0 && block_size > 0 &&
                  max_blocks_per_seq > 0 && max_seq_len > 0,
              "non-positive size argument");
  AIC_REQUIRE(num_q_heads % num_kv_heads == 0, "num_q_heads must be a multiple of num_kv_heads");
  const bool split_lists = short_reqs != nullptr || long_reqs != nullptr;
  AIC_REQUIRE(!split_lists || (n_short >= 0 && n_long >= 0 && n_short + n_long == batch &&
                               (n_short == 0 || short_reqs) && (n_long == 0 || long_reqs)),
              "short/long request lists must partition the batch");
  if (head_size != kD && head_size != 64) {
    set_error("head_size %d not supported (128, or 64 with a bf16 cache)", head_size);
    return AIC_ERR_UNSUPPORTED;
  }
  if (head_size == 64 && kv_dtype != AIC_DT_BF16) {
    set_error("head_size 64 needs a bf16 kv cache");
    return AIC_ERR_UNSUPPORTED;
  }
  if (kv_dtype != AIC_DT_BF16 && kv_dtype != AIC_DT_FP8_E4M3) {
    set_error("kv cache dtype %d not supported (bf16 or fp8 e4m3)", kv_dtype);
    return AIC_ERR_UNSUPPORTED;
  }
  const bool kv8 = kv_dtype == AIC_DT_FP8_E4M3;
  AIC_REQUIRE(!kv8 || (k_scale && v_scale), "an fp8 kv cache needs k_scale and v_scale (device scalars)");
  AIC_REQUIRE(q_stride % 8 == 0 && out_stride % 2 == 0 && block_stride % 16 == 0, "strides must keep 16-byte alignment");
  if (block_size % 16 != 0) {
    set_error("block_size %d not supported (must be a multiple of 16)", block_size);
    return AIC_ERR_UNSUPPORTED;
  }
  AIC_NEED_DEVICE();

  const bool d64 = head_size == 64;   // the secondary head size (gpt-oss; bf16 cache only)
  const int G = num_q_heads / num_kv_heads;
  const int max_rows = max_q_len * G;
  const bool wave_heads = num_kv_heads % 4 == 0;
  const int hgroups = wave_heads ? num_kv_heads / 4 : num_kv_heads;
  const size_t rows = static_cast<size_t>(num_tokens) * num_q_heads;
  // splits of the short / generic launch and of the long-draft launch (its items are few: more splits)
  // with long drafts in the call the short part stays within one workgroup per CU (see pick_splits): measured with
  // rocprofv3 on the bench, 50 short requests otherwise took 5 splits = 500 workgroups, left no room for the long part,
  // and the call fell back to two launches whose long kernel alone ran 212 us
  const bool mixed = split_lists && n_short > 0 && n_long > 0;
  int hpw = wave_heads ? 4 : 1;   // the generic launch (lengths unknown on the host) keeps r01's two forms
  int n_splits;
  if (split_lists && n_short > 0) {
    const ShortLayout lay = pick_short_layout(n_short, num_kv_heads, max_seq_len, mixed ? cu_count() : 0);
    hpw = lay.hpw;
    n_splits = lay.splits;
  } else {
    n_splits = pick_splits(batch * hgroups, max_seq_len, wave_heads ? 2 : 8, 1, 0);
  }
  const int hgroups_s = num_kv_heads / hpw;              // head groups of the short launch
  const bool direct = split_lists && n_short > 0 && n_splits == 1;
  int n_splits_long = (split_lists && n_long > 0) ? pick_splits(n_long * num_kv_heads, max_seq_len, mixed ? 4 : 8, 2) : 0;
  // The long part of a mixed call: its workgroups start first and should end with the short ones.  Per-workgroup traces
  // (tools/microbench.py sptrace / longsplits, 4224-token contexts, profiles/r02_long_splits.txt): a short workgroup takes
  // 4.3 + 0.30 X us with X = kv heads x requests of the call (X = 512 on one GPU, 64 on a rank of SP = 8), a long one
  // (8.1 + 0.15 X) + 0.049 us per token of its range — a latency chain (request list -> lengths -> block table -> first
  // tile, then one memory round trip per 32-token tile) that stretches with the load the short workgroups put on memory.
  // Equal ends: tokens per long workgroup = 2.6 X - 70 (fitted a little generously), never under 4 tiles; X scales with
  // the context length and halves with an fp8 cache.  r02's earlier rule (twice the short splits, at most 8) was fitted at
  // X >= 256 only; a rank of SP = 8 (X = 64) gained 50.5 -> 36 us per layer with 63 + 1 requests, SP = 4 63.7 -> 56 us.
  if (mixed) {
    int cap = g_long_splits;
    if (cap == 0) {
      const double X = static_cast<double>(num_kv_heads) * batch * (kv8 ? 0.5 : 1.0) * max_seq_len / 4224.0;
      const double tokens = std::max(128.0 + 16.0 * (n_long - 1), 2.6 * X - 70.0);
      cap = std::max(2, std::min(32, static_cast<int>(std::ceil(max_seq_len / tokens))));
    }
    n_splits_long = std::min(n_splits_long, cap);
  }
  // Many long drafts in one call (a lane step in which suffix decoding hit for half the requests): the one-grid form —
  // every workgroup resident at once, two per CU — has room for ONE split of the long part only, every long workgroup then
  // walks a whole context alone, and the launch takes as long as that walk (rocprofv3 of the r03 bench: 15 long + 17 short
  // requests, 400 workgroups, 240 us per layer where the usual mixes take 90-103).  Such a call goes out as two plain
  // launches on the same stream instead, the long part with the split count it would have alone, then the short part
  // (16 long drafts alone take 105 us, 17 short requests ~50).
  bool sequential = false;
  if (mixed && !d64) {
    const int long_z_ = (max_rows + 4 * kLongTilesPerWave * 16 - 1) / (4 * kLongTilesPerWave * 16);
    const int short_wg_ = (n_short * hgroups_s + 7) / 8 * 8 * n_splits;
    const int per_split_ = n_long * num_kv_heads * long_z_;
    const int fit = (2 * cu_count() - short_wg_) / per_split_;
    const bool want = g_force_sequential >= 0 ? g_force_sequential == 1 : (fit == 1 && n_splits_long >= 2);
    if (fit < 1 || want) {
      sequential = true;
      n_splits_long = pick_splits(n_long * num_kv_heads, max_seq_len, 8, 2);
    }
  }
  auto fits = [&](int parts) { return static_cast<size_t>(parts) * rows * (head_size + 2) * sizeof(float) <= workspace_bytes; };
  while (n_splits > 1 && !fits(n_splits)) --n_splits;
  while (n_splits_long > 1 && !fits(n_splits_long)) --n_splits_long;
  AIC_REQUIRE(fits(n_splits), "workspace too small (%zu bytes)", workspace_bytes);
  const int n_parts_total = std::max(n_splits, n_splits_long);

  AttnParams P;
  P.q = static_cast<const uint16_t*>(q);
  P.k_cache = static_cast<const uint16_t*>(k_cache);
  P.v_cache = static_cast<const uint16_t*>(v_cache);
  P.block_table = block_table;
  P.seq_lens = seq_lens;
  P.query_start_loc = query_start_loc;
  P.ws_o = static_cast<float*>(workspace);
  P.ws_ml = P.ws_o + static_cast<size_t>(n_parts_total) * rows * head_size;
  P.q_stride = q_stride;
  P.block_stride = block_stride;
  P.max_blocks = max_blocks_per_seq;
  P.num_q_heads = num_q_heads;
  P.num_kv_heads = num_kv_heads;
  P.block_size = block_size;
  P.long_dma = g_long_dma;
  P.bs_shift = (block_size & (block_size - 1)) == 0 ? __builtin_ctz(static_cast<unsigned>(block_size)) : -1;
  P.n_splits = n_splits;
  P.n_parts_total = n_parts_total;
  P.total_rows = static_cast<int>(rows);
  P.sm_scale = sm_scale;
  P.req_list = nullptr;
  P.k_scale = k_scale;
  P.v_scale = v_scale;
  P.dbg = 0;
  P.trace = nullptr;
  P.ptrace = nullptr;
  P.ptrace_stride = 0;
  P.out = static_cast<uint16_t*>(out);
  P.out_stride = out_stride;
  P.direct = 0;
  P.mark_final = 0;
  P.light_first = 0;
  P.light_stride = 0;
  P.light_pct = 100;
  P.window = sliding_window;
  P.sinks = sinks;

  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  // short / generic kernel for (row tiles, heads per workgroup, cache dtype)
  auto launch_short = [&](int mtq_, int hpw_, dim3 grid_) {
#define AIC_SHORT(MTQ_, HPW_)                                                                  \
  if (d64)                                                                                     \
    launch(verify_attn_kernel<MTQ_, HPW_, false, 4, 64>, grid_, dim3(256), s, P);                  \
  else if (kv8)                                                                                \
    launch(verify_attn_kernel<MTQ_, HPW_, true>, grid_, dim3(256), s, P);                      \
  else                                                                                         \
    launch(verify_attn_kernel<MTQ_, HPW_, false>, grid_, dim3(256), s, P);
    if (mtq_ == 0) {
      if (hpw_ == 4) { AIC_SHORT(0, 4) } else if (hpw_ == 2) { AIC_SHORT(0, 2) } else { AIC_SHORT(0, 1) }
    } else if (mtq_ == 1) {
      if (hpw_ == 4) { AIC_SHORT(1, 4) } else if (hpw_ == 2) { AIC_SHORT(1, 2) } else { AIC_SHORT(1, 1) }
    } else {
      if (hpw_ == 4) { AIC_SHORT(2, 4) } else if (hpw_ == 2) { AIC_SHORT(2, 2) } else { AIC_SHORT(2, 1) }
    }
#undef AIC_SHORT
  };
  if (!split_lists) {
    if (!t_rec) profile_begin(s);
    // query lengths unknown on the host: rows per request in the common case decide the tile shape; long
    // drafts of a mixed batch take extra row groups (re-reading their KV through L2)
    const int avg_rows = (num_tokens + batch - 1) / batch * G;
    const int mtq = (max_rows <= 16 || avg_rows <= 24) ? 1 : 2;
    P.m_groups = (max_rows + mtq * 16 - 1) / (mtq * 16);
    P.n_items = batch * hgroups;
    dim3 grid(static_cast<unsigned>((P.n_items + 7) / 8 * 8 * P.m_groups), n_splits, 1);
    launch_short(mtq, hpw, grid);
    if (!t_rec) profile_end(s);
  } else {
    // the caller partitioned the batch: `short_reqs` have q_len * G <= 32 rows (one pass of the short body, one or
    // two MFMA row tiles chosen per workgroup: verify_attn_body_dual), `long_reqs` go through the shared-tile body
    // that reads their KV once for up to 192 rows
    const int mtq_short = 0;
    SideStream* side = nullptr;
    const int per_block_rows_p = 4 * kLongTilesPerWave * 16;
    const int long_z = (max_rows + per_block_rows_p - 1) / per_block_rows_p;
    const int short_wg = (n_short * hgroups_s + 7) / 8 * 8 * n_splits;
    P.direct = direct ? 1 : 0;
    P.mark_final = (direct && n_long > 0) ? 1 : 0;
    // one launch for both kinds of request when every workgroup of it can be resident at once (two per CU).  (Letting
    // the long part take the whole chip first — 2 workgroups per CU, the short ones dispatched as those retire — was
    // measured and is worse: 59 short + 5 long 214 us against 206 us, and the bench's real mix 267 us against 182 us.)
    bool pair = n_short > 0 && n_long > 0 && !d64;   // (the one-grid form is instantiated for head size 128)
    if (sequential) pair = false;
    if (pair) {
      const int room = 2 * cu_count() - short_wg;
      const int per_split = n_long * num_kv_heads * long_z;
      if (room < per_split) {
        pair = false;
      } else if (n_splits_long * per_split > room) {
        n_splits_long = room / per_split;
      }
    }
    if (pair) {
      AttnParams PL = P;
      PL.req_list = long_reqs;
      PL.n_splits = n_splits_long;
      P.req_list = short_reqs;
      P.m_groups = 1;
      P.n_items = n_short * hgroups_s;
      const int long_x = n_long * num_kv_heads;
      const int n_long_wg = long_x * n_splits_long * long_z, n_long_pad = (n_long_wg + 7) / 8 * 8;
      const int short_x = (P.n_items + 7) / 8 * 8;
      // The dispatcher places the long workgroups (first in the grid) one per CU, then the short ones on the EMPTY CUs
      // first: only the last (long + short - CUs) short workgroups share a CU with a long one, and those run 7-10 % longer
      // (per-workgroup trace, tools/microbench.py trace with AIC_TRACE_DETAIL: 31 short + 1 long, last 64 of 256: 86.7
      // against 80.3 us; 30 + 2, last 128: 89.6 against 81.5; 62 + 2 at 64 requests, last 64: 167.5 against 156.6).  Those
      // workgroups get a shorter token range; the other splits of their items take up the difference.  Measured with equal
      // / 92 % / 88 % ranges: a 32-request lane with one 33-token draft 97.8 / 95.5 / 94.1 us, with two 101.3 / 99.5 / 98.6;
      // 63 requests + one draft 182.8 / 176.5 / 174.4 us, 62 + 2: 182.9 / 178.1 / 177.1, 59 + 5: 185.9 / 184.8 / 183.9.
      if (n_splits > 1 && short_wg <= cu_count()) {
        const int first_shared = std::max(0, cu_count() - n_long_wg);
        if (first_shared > 0 && first_shared < short_wg) {
          P.light_first = first_shared;
          P.light_stride = short_x;
          P.light_pct = g_light_pct ? g_light_pct : 88;
        }
      }
      if (!t_rec) profile_begin(s);
      const dim3 pgrid(static_cast<unsigned>(n_long_pad + short_wg));
      P.trace = (g_attn_trace && static_cast<int>(pgrid.x) <= g_attn_trace_cap) ? g_attn_trace : nullptr;
#define AIC_PAIR_LAUNCH(HPW_, KV8_)                                                                             \
  launch(verify_attn_pair_kernel<HPW_, KV8_, 0>, pgrid, dim3(256), s, P, PL, n_long_wg, n_long_pad, long_x, n_splits_long, short_x);
      if (g_attn_ptrace && hpw == 4 && !kv8 && n_long_wg * 8 <= g_attn_ptrace_cap && long_x < 65536) {
        // debug: the long workgroups of a MIXED call keep the per-wave cycle account (tools/microbench.py mixphases)
        PL.ptrace = g_attn_ptrace;
        PL.ptrace_stride = long_x | (n_splits_long << 16);
        launch(verify_attn_pair_kernel<4, false, 0, true>, pgrid, dim3(256), s, P, PL, n_long_wg, n_long_pad, long_x, n_splits_long, short_x);
      } else if (hpw == 4) {
        if (kv8) { AIC_PAIR_LAUNCH(4, true) } else { AIC_PAIR_LAUNCH(4, false) }
      } else if (hpw == 2) {
        if (kv8) { AIC_PAIR_LAUNCH(2, true) } else { AIC_PAIR_LAUNCH(2, false) }
      } else {
        if (kv8) { AIC_PAIR_LAUNCH(1, true) } else { AIC_PAIR_LAUNCH(1, false) }
      }
#undef AIC_PAIR_LAUNCH
      if (!t_rec) profile_end(s);
      n_short = n_long = 0;  // both done
    }
    const bool overlap = n_short > 0 && n_long > 0 && !sequential;   // (head size 64: long part beside the short one on a side stream)
    const bool short_first = false;
    if (overlap && t_rec) {      // a fork / join over the side stream is not a plain kernel chain
      t_rec->ok = false;
      return AIC_OK;
    }
    if (n_short > 0 && !t_rec) profile_begin(s);  // bench.py's roofline figure: the short-request kernel alone (the begin
                                        // marker sits before the fork so that it delays neither kernel's start)
    if (overlap) {
      if ((rc = side_stream(&side)) != AIC_OK) return rc;
      AIC_HIP_TRY(hipEventRecord(side->fork, s));
      AIC_HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
    }
    auto launch_long = [&]() -> int {
      AttnParams PL = P;
      PL.req_list = long_reqs;
      const int per_block_rows = 4 * kLongTilesPerWave * 16;
      PL.n_splits = n_splits_long;
      dim3 grid(static_cast<unsigned>(n_long * num_kv_heads), n_splits_long, (max_rows + per_block_rows - 1) / per_block_rows);
      if (g_attn_ptrace && !d64 && !kv8 && n_short == 0 && static_cast<int>(grid.x * grid.y * grid.z) * 8 <= g_attn_ptrace_cap) {
        PL.ptrace = g_attn_ptrace;   // debug: per-wave cycle accounting of a long-only call (tools/microbench.py longphases)
        PL.ptrace_stride = static_cast<int>(grid.x) | (static_cast<int>(grid.y) << 16);
        launch(verify_attn_long4_kernel<false, 128, true>, grid, dim3(256), s, PL);
      } else if (d64)
        launch(verify_attn_long4_kernel<false, 64>, grid, dim3(256), overlap ? side->stream : s, PL);
      else if (kv8)
        launch(verify_attn_long4_kernel<true, 128>, grid, dim3(256), overlap ? side->stream : s, PL);
      else
        launch(verify_attn_long4_kernel<false, 128>, grid, dim3(256), overlap ? side->stream : s, PL);
      if (overlap) AIC_HIP_TRY(hipEventRecord(side->join, side->stream));
      return AIC_OK;
    };
    if (n_long > 0 && !short_first && (rc = launch_long()) != AIC_OK) return rc;
    if (n_short > 0) {
      P.req_list = short_reqs;
      P.m_groups = 1;
      P.n_items = n_short * hgroups_s;
      dim3 grid(static_cast<unsigned>((P.n_items + 7) / 8 * 8), n_splits, 1);
      if (g_attn_ptrace && static_cast<int>(grid.x * grid.y) <= g_attn_ptrace_cap) {
        P.ptrace = g_attn_ptrace;
        P.ptrace_stride = static_cast<int>(grid.x);
      }
      launch_short(mtq_short, hpw, grid);
      P.ptrace = nullptr;
      if (!t_rec) profile_end(s);
    }
    if (n_long > 0 && short_first && (rc = launch_long()) != AIC_OK) return rc;
    if (overlap) AIC_HIP_TRY(hipStreamWaitEvent(s, side->join, 0));
  }
  if ((rc = launch_ok("verify_attn_kernel")) != AIC_OK) return rc;
  if (P.direct && !P.mark_final) return AIC_OK;   // every row of the call is already final: no combine launch
  if (d64)
    launch(verify_attn_combine_kernel<64>, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), s, P.ws_o, P.ws_ml, n_parts_total,
           static_cast<int>(rows), num_q_heads, static_cast<uint16_t*>(out), out_stride);
  else
    launch(verify_attn_combine_kernel<128>, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), s, P.ws_o, P.ws_ml, n_parts_total,
           static_cast<int>(rows), num_q_heads, static_cast<uint16_t*>(out), out_stride);
  return launch_ok("verify_attn_combine_kernel");
}

// ---- a run of layers as one graph launch ------------------------------------------------------------------------------
// 32 layers are 64 kernel launches per engine step, ~4.4 us of host time each (profiles/r02_microbench.txt, graph probe):
// more than the kernels themselves take once the sequence is sharded 8 ways.  The launches of one step form a plain chain
// whose SHAPE (which kernel at which position) repeats from step to step while grids and arguments change, so the chain is
// kept as an instantiated HIP graph per shape and each step only rewrites the kernel nodes' parameters
// (hipGraphExecKernelNodeSetParams, ~0.7 us per node) and launches it once.
// An instantiated graph may still be in flight when the next call of the same shape arrives (interleaved lanes enqueue
// one lane's step while the other's runs), and rewriting the node parameters of an exec whose launch has not finished is
// not something the runtime promises to keep apart: every shape therefore owns a few execs, each followed by an event,
// and a call takes one whose last launch has completed (instantiating another, up to kExecsPerShape, before it waits).
struct GraphExecSlot {
  hipGraphExec_t exec;
  hipEvent_t done;
  bool launched;
};
constexpr size_t kExecsPerShape = 4;
struct LayerGraph {
  int device;
  std::vector<void*> funcs;           // the shape: kernel of every node, in order
  hipGraph_t graph;
  std::vector<GraphExecSlot> execs;
  size_t next_exec;
  std::vector<hipGraphNode_t> nodes;
  uint64_t last_use;
};
static std::mutex g_graph_mu;
static std::vector<LayerGraph> g_graphs;
static uint64_t g_graph_tick = 0;
static std::atomic<int> g_graph_mode{1};          // aic_debug_attn_graph: 0 = always launch kernel by kernel
static std::atomic<unsigned> g_layers_calls{0};
static std::atomic<uint64_t> g_graph_launches{0}, g_graph_builds{0}, g_graph_execs{0};

static void destroy_layer_graph(LayerGraph& c) {
  for (GraphExecSlot& e : c.execs) {
    if (e.launched) (void)hipEventSynchronize(e.done);
    (void)hipGraphExecDestroy(e.exec);
    (void)hipEventDestroy(e.done);
  }
  (void)hipGraphDestroy(c.graph);
}

// A fresh exec of `c`, instantiated from the graph as it stands (callers set every node's parameters right after).
static int add_exec(LayerGraph& c) {
  GraphExecSlot e{};
  AIC_HIP_TRY(hipGraphInstantiate(&e.exec, c.graph, nullptr, nullptr, 0));
  const hipError_t err = hipEventCreateWithFlags(&e.done, hipEventDisableTiming);
  if (err != hipSuccess) {
    (void)hipGraphExecDestroy(e.exec);
    AIC_HIP_TRY(err);
  }
  e.launched = false;
  c.execs.push_back(e);
  g_graph_execs.fetch_add(1, std::memory_order_relaxed);
  return AIC_OK;
}
constexpr size_t kMaxLayerGraphs = 24;

static int replay_as_graph(const Recorder& rec, hipStream_t s) {
  int device = 0;
  AIC_HIP_TRY(hipGetDevice(&device));
  const size_t n = rec.n;
  static thread_local std::vector<void*> argv;            // argument pointers of all nodes, node after node
  static thread_local std::vector<hipKernelNodeParams> kp;
  size_t n_args = 0;
  for (size_t i = 0; i < n; ++i) n_args += rec.recs[i].offs.size();
  argv.resize(n_args);
  kp.resize(n);
  for (size_t i = 0, at = 0; i < n; ++i) {
    const LaunchRec& r = rec.recs[i];
    kp[i] = hipKernelNodeParams{};
    kp[i].func = r.func;
    kp[i].gridDim = r.grid;
    kp[i].blockDim = r.block;
    kp[i].sharedMemBytes = 0;
    kp[i].kernelParams = argv.data() + at;
    kp[i].extra = nullptr;
    for (size_t a = 0; a < r.offs.size(); ++a) argv[at++] = const_cast<char*>(r.blob.data()) + r.offs[a];
  }
  std::lock_guard<std::mutex> lock(g_graph_mu);
  LayerGraph* g = nullptr;
  for (LayerGraph& c : g_graphs) {
    if (c.device != device || c.funcs.size() != n) continue;
    bool same = true;
    for (size_t i = 0; i < n && same; ++i) same = c.funcs[i] == rec.recs[i].func;
    if (same) { g = &c; break; }
  }
  if (g == nullptr) {
    if (g_graphs.size() >= kMaxLayerGraphs) {     // drop the shape that has not been seen for the longest time
      size_t victim = 0;
      for (size_t i = 1; i < g_graphs.size(); ++i)
        if (g_graphs[i].last_use < g_graphs[victim].last_use) victim = i;
      destroy_layer_graph(g_graphs[victim]);
      g_graphs.erase(g_graphs.begin() + static_cast<long>(victim));
    }
    LayerGraph c;
    c.device = device;
    c.next_exec = 0;
    c.nodes.resize(n);
    AIC_HIP_TRY(hipGraphCreate(&c.graph, 0));
    for (size_t i = 0; i < n; ++i) {
      const hipError_t e = hipGraphAddKernelNode(&c.nodes[i], c.graph, i ? &c.nodes[i - 1] : nullptr, i ? 1 : 0, &kp[i]);
      if (e != hipSuccess) {
        (void)hipGraphDestroy(c.graph);
        AIC_HIP_TRY(e);
      }
      c.funcs.push_back(rec.recs[i].func);
    }
    const int rc = add_exec(c);
    if (rc != AIC_OK) {
      (void)hipGraphDestroy(c.graph);
      return rc;
    }
    g_graphs.push_back(std::move(c));
    g = &g_graphs.back();
    g_graph_builds.fetch_add(1, std::memory_order_relaxed);
  }
  // an exec whose previous launch has finished (round robin from the one after the last used)
  GraphExecSlot* slot = nullptr;
  for (size_t k = 0; k < g->execs.size() && !slot; ++k) {
    GraphExecSlot& e = g->execs[(g->next_exec + k) % g->execs.size()];
    if (!e.launched || hipEventQuery(e.done) == hipSuccess) {
      slot = &e;
      g->next_exec = (g->next_exec + k + 1) % g->execs.size();
    }
  }
  (void)hipGetLastError();                        // a "not ready" answer above is not an error of this call
  if (!slot) {
    if (g->execs.size() < kExecsPerShape) {
      const int rc = add_exec(*g);
      if (rc != AIC_OK) return rc;
      slot = &g->execs.back();
      g->next_exec = 0;
    } else {                                      // every exec busy: wait for the oldest
      slot = &g->execs[g->next_exec];
      g->next_exec = (g->next_exec + 1) % g->execs.size();
      AIC_HIP_TRY(hipEventSynchronize(slot->done));
    }
  }
  for (size_t i = 0; i < n; ++i) AIC_HIP_TRY(hipGraphExecKernelNodeSetParams(slot->exec, g->nodes[i], &kp[i]));
  g->last_use = ++g_graph_tick;
  AIC_HIP_TRY(hipGraphLaunch(slot->exec, s));
  AIC_HIP_TRY(hipEventRecord(slot->done, s));
  slot->launched = true;
  g_graph_launches.fetch_add(1, std::memory_order_relaxed);
  return AIC_OK;
}

// 0: kernel-by-kernel launches only; 1 (default): runs of >= 4 layers go out as one graph launch.
int aic_debug_attn_long_dma(int pattern) {
  AIC_REQUIRE(pattern >= 0 && pattern <= 3, "bad pattern");
  g_long_dma = pattern;
  return AIC_OK;
}
int aic_debug_attn_long_splits(int splits) {
  AIC_REQUIRE(splits >= 0 && splits <= 64, "bad split count");
  g_long_splits = splits;
  return AIC_OK;
}
int aic_debug_attn_graph(int on) {
  g_graph_mode.store(on, std::memory_order_relaxed);
  return AIC_OK;
}
// graph launches so far and distinct shapes instantiated
int aic_debug_attn_graph_stats(uint64_t* launches, uint64_t* builds) {
  if (launches) *launches = g_graph_launches.load(std::memory_order_relaxed);
  if (builds) *builds = g_graph_builds.load(std::memory_order_relaxed);
  return AIC_OK;
}

// The same call for a run of layers that share the step's geometry (one engine step of a stand-alone driver: q / out
// advance by a fixed stride per layer or stay, the caches come from pointer tables): one foreign call instead of one per
// layer.  q_layer_stride / out_layer_stride are in elements (0 = every layer reads the same q / writes the same out).
int aic_verify_attention_layers(const void* q, int64_t q_stride, int64_t q_layer_stride, const void* const* k_caches,
                                const void* const* v_caches, int n_layers, int64_t block_stride, int kv_dtype,
                                const float* k_scale, const float* v_scale, const int32_t* block_table,
                                int max_blocks_per_seq, const int32_t* seq_lens, const int32_t* query_start_loc, int batch,
                                int num_tokens, int max_q_len, int num_q_heads, int num_kv_heads, int head_size,
                                int block_size, float sm_scale, void* out, int64_t out_stride, int64_t out_layer_stride,
                                void* workspace, size_t workspace_bytes, int max_seq_len, const int32_t* short_reqs,
                                int n_short, const int32_t* long_reqs, int n_long, void* stream) {
  AIC_REQUIRE(n_layers >= 0 && (n_layers == 0 || (k_caches && v_caches)), "bad layer tables");
  auto layers = [&](int l0, int l1) -> int {
    for (int l = l0; l < l1; ++l) {
      const uint16_t* ql = static_cast<const uint16_t*>(q) + static_cast<int64_t>(l) * q_layer_stride;
      uint16_t* ol = static_cast<uint16_t*>(out) + static_cast<int64_t>(l) * out_layer_stride;
      const int rc = aic_verify_attention_ex(ql, q_stride, k_caches[l], v_caches[l], block_stride, kv_dtype, k_scale, v_scale,
                                             block_table, max_blocks_per_seq, seq_lens, query_start_loc, batch, num_tokens,
                                             max_q_len, num_q_heads, num_kv_heads, head_size, block_size, sm_scale, ol,
                                             out_stride, workspace, workspace_bytes, max_seq_len, short_reqs, n_short,
                                             long_reqs, n_long, stream);
      if (rc != AIC_OK || (t_rec && !t_rec->ok)) return rc;
    }
    return AIC_OK;
  };
  bool as_graph = n_layers >= 4 && g_graph_mode.load(std::memory_order_relaxed) != 0 && g_attn_trace == nullptr;
  // with aic_profile_enable() on, every 5th call goes out kernel by kernel with its event pairs (graphs carry none): the
  // live per-launch duration bench.py reports stays a sample of the same steps (an odd stride: interleaved lanes alternate)
  const unsigned call = g_layers_calls.fetch_add(1, std::memory_order_relaxed);
  if (as_graph && profile_enabled() && call % 5 == 0) as_graph = false;
  if (as_graph) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
      as_graph = false;                      // the caller is capturing its own graph: plain launches join that one
  }
  int done = 0;
  if (as_graph) {
    // two graph launches: a short head (4 layers) so that the GPU starts while the host writes the parameters of the rest
    // (an idle GPU otherwise waits ~100 us for 64 node updates: rocprofv3 trace of the rehearsed SP = 8 step)
    const int head = (n_layers >= 12 && g_graph_mode.load(std::memory_order_relaxed) != 2) ? 4 : n_layers;
    for (int l0 = 0; l0 < n_layers;) {
      const int l1 = l0 == 0 ? head : n_layers;
      static thread_local Recorder rec;
      rec.reset();
      int rc;
      {
        struct Recording {                   // whatever ends the scope, later launches of this thread are real again
          explicit Recording(Recorder* r) { t_rec = r; }
          ~Recording() { t_rec = nullptr; }
        } recording(&rec);
        rc = layers(l0, l1);
      }
      if (rc != AIC_OK) return rc;
      if (!rec.ok) break;                    // a side-stream fork in this geometry: kernel by kernel from here
      if (rec.n > 0) {
        const int rc2 = replay_as_graph(rec, static_cast<hipStream_t>(stream));
        if (rc2 != AIC_OK) return rc2;
      }
      done = l0 = l1;
    }
  }
  return layers(done, n_layers);
}

int aic_verify_attention(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                         int64_t block_stride, int kv_dtype, const float* k_scale, const float* v_scale,
                         const int32_t* block_table, int max_blocks_per_seq, const int32_t* seq_lens,
                         const int32_t* query_start_loc, int batch, int num_tokens, int max_q_len, int num_q_heads,
                         int num_kv_heads, int head_size, int block_size, float sm_scale, void* out,
                         int64_t out_stride, void* workspace, size_t workspace_bytes, int max_seq_len, void* stream) {
  return aic_verify_attention_ex(q, q_stride, k_cache, v_cache, block_stride, kv_dtype, k_scale, v_scale, block_table,
                                 max_blocks_per_seq, seq_lens, query_start_loc, batch, num_tokens, max_q_len,
                                 num_q_heads, num_kv_heads, head_size, block_size, sm_scale, out, out_stride, workspace,
                                 workspace_bytes, max_seq_len, nullptr, 0, nullptr, 0, stream);
}

}  // extern "C"
