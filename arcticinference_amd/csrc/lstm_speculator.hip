// A7-A11 — Arctic LSTM speculator (method "sum_lstm") draft loop on MI355X.
//
// Reference (read as text, vLLM is not importable):
//   ArcticLSTMSpeculator.generate_states / generate_token_ids / generate_proposals
//     /root/reference/arctic_inference/vllm/spec_dec/arctic_speculator.py:648-691, :706-751, :753-866
//   MLPSpeculatorLayerNorm.forward  :88-95     LogitsProcessorOpt._get_logits logits_processor_opt.py:83-107
//   OriginalFp8LinearMethod (per-tensor W8A8, dynamic activation scale) fp8.py:207-223, :276-308
//   hidden-state pick  arctic_proposer.py:133-147
// The reference runs ~15 eager torch ops per head and hides the launches in a CUDA graph; every op
// rounds to bf16.  Here a head is five launches, the elementwise chain is one kernel, and the
// per-op bf16 roundings of the reference are kept (r() below) so draft tokens agree.
//
// Cost model (Llama-3.1-8B speculator, Ds = H = 4096, V = 128256, k = 3, B <= 64): per head the
// gate projection streams 4Ds x K bf16 = 134 MB and the LM head V x Ds = 525 MB (fp8) / 1.05 GB
// (bf16) -> HBM-bound skinny GEMMs (M = padded batch <= 64, ridge far away), MFMA only because
// 2*M*N*K at M=64 exceeds the vector ALU rate.  Design for that bound:
//   * weights are re-laid out once at load into MFMA-fragment-major tiles: the A fragment of
//     (16 weight rows x 32 k) is one contiguous 1 KiB chunk, 16 B per lane -> every weight load is
//     a perfectly coalesced 1 KiB wave instruction streaming a 128 KiB contiguous run per row tile;
//     loads go straight to VGPRs one K-chunk ahead (8 KiB per wave in flight, 64 KiB per CU at
//     2 workgroups/CU);
//   * activations (<= 512 KiB, L2 resident) are kept fragment-major as well, copied linearly into a
//     double-buffered LDS chunk shared by the 4 waves of a workgroup and read with conflict-free
//     ds_read_b128;
//   * the LM head never materialises [B, V] logits: bf16-rounded logits are arg-max-reduced in the
//     MFMA accumulator layout (4 regs -> 2 shuffles -> LDS across waves) to one (value, index) per
//     workgroup and batch row.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "aic_common.h"

namespace aic {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kChunkSteps = 4;   // k-steps per chunk (bf16: 4 x 32 k = 128; fp8: 4 x 64 k = 256)
constexpr int kRowsPerBlock = 64;  // 4 waves x one 16-row tile

__device__ __forceinline__ float r(float x) { return round_bf16(x); }

// ---- fragment-major activation layout --------------------------------------------------------
// bf16: 16-byte unit u = ((k/32) * MT + m/16) * 64 + ((k/8)%4)*16 + m%16 holds k%8 = 0..7
// fp8 : 16-byte unit u = ((k/64) * MT + m/16) * 64 + ((k/16)%4)*16 + m%16 holds k%16 = 0..15
__device__ __forceinline__ int64_t xunit_bf16(int m, int k8, int MT) {  // k8 = k / 8
  return (static_cast<int64_t>(k8 >> 2) * MT + (m >> 4)) * 64 + (k8 & 3) * 16 + (m & 15);
}
__device__ __forceinline__ int64_t xunit_fp8(int m, int k16, int MT) {  // k16 = k / 16
  return (static_cast<int64_t>(k16 >> 2) * MT + (m >> 4)) * 64 + (k16 & 3) * 16 + (m & 15);
}

// ---- block reductions --------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {  // sh: >= 16 floats
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  __syncthreads();  // sh may still be read from a previous reduction
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < n_waves; ++w) t += sh[w];
  return t;
}

// ---- weight repacking (one-time, at load) ------------------------------------------------------
// dst unit ((rt * KS + ks) * 64 + l) <- src[rt*16 + l%16][ks*32 + 8*(l/16) .. +8]   (bf16)
__global__ void __launch_bounds__(256)
repack_bf16_kernel(const uint16_t* __restrict__ src, uint4* __restrict__ dst, int n_rows, int K, int n_rowtiles) {
  const int KS = K / 32;
  const int64_t total = static_cast<int64_t>(n_rowtiles) * KS * 64;
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; u < total;
       u += static_cast<int64_t>(gridDim.x) * 256) {
    const int l = static_cast<int>(u & 63);
    const int64_t t = u >> 6;
    const int ks = static_cast<int>(t % KS);
    const int rt = static_cast<int>(t / KS);
    const int row = rt * 16 + (l & 15);
    const int col = ks * 32 + 8 * (l >> 4);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n_rows) v = *reinterpret_cast<const uint4*>(src + static_cast<int64_t>(row) * K + col);
    dst[u] = v;
  }
}

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return static_cast<uint32_t>(w);
}
__device__ __forceinline__ float clamp448(float x) { return __builtin_amdgcn_fmed3f(x, 448.0f, -448.0f); }

// dst unit ((rt * KP + kp) * 64 + l) <- q(src[rt*16 + l%16][kp*64 + 16*(l/16) .. +16])   (e4m3fn)
__global__ void __launch_bounds__(256)
quant_repack_fp8_kernel(const uint16_t* __restrict__ src, uint4* __restrict__ dst, const float* __restrict__ scale,
                        int n_rows, int K, int n_rowtiles) {
  const int KP = K / 64;
  const float inv = 1.0f / *scale;
  const int64_t total = static_cast<int64_t>(n_rowtiles) * KP * 64;
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; u < total;
       u += static_cast<int64_t>(gridDim.x) * 256) {
    const int l = static_cast<int>(u & 63);
    const int64_t t = u >> 6;
    const int kp = static_cast<int>(t % KP);
    const int rt = static_cast<int>(t / KP);
    const int row = rt * 16 + (l & 15);
    const int col = kp * 64 + 16 * (l >> 4);
    uint4 out = make_uint4(0, 0, 0, 0);
    if (row < n_rows) {
      const uint16_t* p = src + static_cast<int64_t>(row) * K + col;
      uint16_t h[16];
      *reinterpret_cast<uint4*>(h) = *reinterpret_cast<const uint4*>(p);
      *reinterpret_cast<uint4*>(h + 8) = *reinterpret_cast<const uint4*>(p + 8);
      float f[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) f[e] = clamp448(bf16_to_f32(h[e]) * inv);
      out.x = pack4_fp8(f[0], f[1], f[2], f[3]);
      out.y = pack4_fp8(f[4], f[5], f[6], f[7]);
      out.z = pack4_fp8(f[8], f[9], f[10], f[11]);
      out.w = pack4_fp8(f[12], f[13], f[14], f[15]);
    }
    dst[u] = out;
  }
}

// |x| maximum of a bf16 array into *amax_bits (non-negative floats order like their bit patterns)
__global__ void __launch_bounds__(256)
amax_bf16_kernel(const uint16_t* __restrict__ src, int64_t n, unsigned int* __restrict__ amax_bits) {
  float m = 0.0f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256)
    m = fmaxf(m, fabsf(bf16_to_f32(src[i])));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));
}
// scale = max(amax / 448, 1 / (448 * 512))   (vLLM dynamic per-tensor scaled_fp8_quant)
__global__ void finish_scale_kernel(const unsigned int* __restrict__ amax_bits, float* __restrict__ scale) {
  *scale = fmaxf(__uint_as_float(*amax_bits) / 448.0f, 1.0f / (448.0f * 512.0f));
}
__global__ void __launch_bounds__(256)
quant_rowmajor_fp8_kernel(const uint16_t* __restrict__ src, uint8_t* __restrict__ dst, const float* __restrict__ scale,
                          int64_t n) {
  const float inv = 1.0f / *scale;
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256 * 4) {
    float f[4] = {0, 0, 0, 0};
    for (int e = 0; e < 4 && i + e < n; ++e) f[e] = clamp448(bf16_to_f32(src[i + e]) * inv);
    const uint32_t w = pack4_fp8(f[0], f[1], f[2], f[3]);
    for (int e = 0; e < 4 && i + e < n; ++e) dst[i + e] = static_cast<uint8_t>(w >> (8 * e));
  }
}

// ---- head 0 input: ln0(x) / sqrt(2), gathered by hidden_index, fragment-major bf16 ---------------
// MLPSpeculatorLayerNorm without affine (arctic_speculator.py:88-91) with the reference's per-op
// bf16 roundings; rows >= batch are zero.  Also clears the per-call device state.
__global__ void __launch_bounds__(256)
ln0_kernel(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ hidden_index, int batch, int H, int MT,
           int scale_input, uint4* __restrict__ x_out, uint16_t* __restrict__ cell, int Ds,
           unsigned int* __restrict__ amax_bits, int n_amax, const int32_t* __restrict__ last_tokens,
           int32_t* __restrict__ tokens, unsigned int* __restrict__ sync_words, int n_sync) {
  __shared__ float sh[16];
  const int m = blockIdx.x;
  if (m == 0 && threadIdx.x < n_amax) amax_bits[threadIdx.x] = 0u;
  // the slots of the draft's one-launch cells (lstm_cell_kernel mode 2) start every draft empty: the launch boundary behind
  // this kernel orders these stores before the first slot store of a cell
  if (m == 0 && sync_words)
    for (int i = threadIdx.x; i < n_sync; i += 256) sync_words[i] = 0xffffffffu;     // kSyncEmpty
  // the whole-draft entry point hands the conditioning tokens over here (a launch of their own cost 4.8 us, rocprofv3)
  if (m == 0 && last_tokens && threadIdx.x < batch) tokens[threadIdx.x] = last_tokens[threadIdx.x];
  // zero initial cell state (arctic_speculator.py:781-785)
  for (int j = threadIdx.x * 8; j < Ds; j += 256 * 8)
    *reinterpret_cast<uint4*>(cell + static_cast<int64_t>(m) * Ds + j) = make_uint4(0, 0, 0, 0);
  if (m >= batch) {
    for (int k8 = threadIdx.x; k8 < H / 8; k8 += 256) x_out[xunit_bf16(m, k8, MT)] = make_uint4(0, 0, 0, 0);
    return;
  }
  const int64_t row = hidden_index ? hidden_index[m] : m;
  const uint16_t* x = hidden + row * H;
  float ss = 0.0f;
  if (scale_input) {
    for (int k8 = threadIdx.x; k8 < H / 8; k8 += 256) {
      uint16_t h[8];
      *reinterpret_cast<uint4*>(h) = *reinterpret_cast<const uint4*>(x + k8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = bf16_to_f32(h[e]);
        ss += r(v * v);  // xf.pow(2) is a bf16 tensor
      }
    }
    ss = block_sum(ss, sh);
  }
  const float mean = r(ss / static_cast<float>(H));
  const float rs = r(rsqrtf(r(mean + 1e-6f)));
  const float inv_sqrt2 = 0.70710678118654752f;
  for (int k8 = threadIdx.x; k8 < H / 8; k8 += 256) {
    uint16_t h[8];
    *reinterpret_cast<uint4*>(h) = *reinterpret_cast<const uint4*>(x + k8 * 8);
    if (scale_input) {
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = f32_to_bf16(r(bf16_to_f32(h[e]) * rs) * inv_sqrt2);
    }
    x_out[xunit_bf16(m, k8, MT)] = *reinterpret_cast<uint4*>(h);
  }
}

// ---- skinny GEMM: out[m][n] = sum_k X[m][k] * W[n][k],  M = 16*MT <= 64 ---------------------------
// EPI 0: fp32 partials  part[split][m][n]          (gate projection, split-K over blockIdx.y)
// EPI 1: arg-max of bf16(acc * scale) over the block's 64 rows -> best_val/best_idx[m][block]  (row-major per batch row
//        since r04: the reduction over a row's blocks then reads 2 x 8 KB contiguous; as [block][m] every thread of it
//        touched a cache line of its own — 4008 lines per workgroup, +4 us in the cell kernel's in-kernel trace)
//
// Pipeline: K is walked in chunks of S = 4 k-steps (4 KiB of weights per wave).  Weight fragments go
// straight from HBM to VGPRs through a ring of FOUR register sets, three chunks (12 KiB per wave) ahead of
// the MFMAs; with ~128 VGPRs four workgroups fit a CU, i.e. 16 waves x 12 KiB = 192 KiB of weight loads in
// flight per CU — the kernel is latency-bound on HBM, not MFMA-bound, so bytes in flight are the lever
// (one set, one chunk ahead measured 2.6 TB/s).  The activation chunk (L2 resident) is copied one chunk
// ahead into a double-buffered LDS tile shared by the four waves; one barrier per chunk.  The register
// sets are named scalars: kept as arrays the compiler demotes them to scratch at this occupancy.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ uint4 ld_stream16(const uint4* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// XQ (fp8 only): the activation operand is read as bf16 (fragment-major, X) and quantised to e4m3 on its way into the LDS
// tile with the dynamic per-tensor scale max(amax / 448, 2^-9 / 448) taken from *amax_bits — the arithmetic of
// quant_act_kernel (fp8.py:303-308), unit for unit, without its launch and without the fp8 copy of the activations.
struct GemmArgs {
  const uint4* W;
  const uint4* X;
  int n_rowtiles, steps_total, steps_per_split;
  float* part;
  int n_cols_out;
  const float* x_scale;
  float w_scale;
  int n_valid_rows, row_offset;
  float* best_val;
  int32_t* best_idx;
  const unsigned int* amax_bits;
};

template <bool FP8, int MT, int EPI, bool XQ>
__device__ __forceinline__ void skinny_gemm_body(const GemmArgs& A, uint4 (*lds)[kChunkSteps * MT * 64], const int bx,
                                                 const int by) {
  static_assert(!XQ || (FP8 && MT <= 2), "on-the-fly activation quantisation: fp8 head, at most 32 rows");
  constexpr int S = kChunkSteps;
  constexpr int XV = S * MT * 64 / 256;  // uint4 per thread per activation chunk ( = MT )
  const uint4* __restrict__ W = A.W;
  const uint4* __restrict__ X = A.X;
  const int n_rowtiles = A.n_rowtiles, steps_total = A.steps_total, steps_per_split = A.steps_per_split;
  float* __restrict__ part = A.part;
  const int n_cols_out = A.n_cols_out;
  const float w_scale = A.w_scale;
  const int n_valid_rows = A.n_valid_rows, row_offset = A.row_offset;
  float* __restrict__ best_val = A.best_val;
  int32_t* __restrict__ best_idx = A.best_idx;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int rt = bx * 4 + wave;
  const int split = by;
  const int ks0 = split * steps_per_split;
  const int n_chunks = steps_per_split / S;
  const bool live = rt < n_rowtiles;
  float xq_scale = 1.0f, xq_inv = 1.0f;
  if (XQ) {
    xq_scale = fmaxf(__uint_as_float(*A.amax_bits) / 448.0f, 1.0f / (448.0f * 512.0f));
    xq_inv = 1.0f / xq_scale;
  }

  const uint4* a_ptr = W + (static_cast<int64_t>(live ? rt : 0) * steps_total + ks0) * 64 + lane;
  const uint4* x_ptr = X + static_cast<int64_t>(ks0) * MT * 64 + tid;

  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 xr0, xr1, xr2, xr3;  // XV = MT <= 4 of them are live (XQ: two bf16 units per fp8 unit, XV <= 2)
  (void)xr1; (void)xr2; (void)xr3;
  // XQ: fp8 unit U = ((kp * MT + mt) * 64 + g * 16 + mm) <- bf16 units (row mt*16 + mm, k8 = (kp*64 + 16 g) / 8 and + 1)
  auto xq_src = [&](int64_t U) -> int64_t {
    const int mm = static_cast<int>(U & 15), g = static_cast<int>((U >> 4) & 3);
    const int64_t t = U >> 6;
    const int mt_ = static_cast<int>(t % MT);
    const int kp = static_cast<int>(t / MT);
    return xunit_bf16(mt_ * 16 + mm, (kp * 64 + 16 * g) / 8, MT);
  };
  // (no arrays: a dword holds bf16 values 2 i (low half) and 2 i + 1 (high half); bf16 -> f32 is a 16-bit shift)
  auto q2 = [&](uint32_t a, uint32_t b) -> uint32_t {
    return pack4_fp8(clamp448(__uint_as_float(a << 16) * xq_inv), clamp448(__uint_as_float(a & 0xffff0000u) * xq_inv),
                     clamp448(__uint_as_float(b << 16) * xq_inv), clamp448(__uint_as_float(b & 0xffff0000u) * xq_inv));
  };
  auto xq_pack = [&](const uint4& lo, const uint4& hi) -> uint4 {
    return make_uint4(q2(lo.x, lo.y), q2(lo.z, lo.w), q2(hi.x, hi.y), q2(hi.z, hi.w));
  };
  uint4 a0_0, a0_1, a0_2, a0_3, a1_0, a1_1, a1_2, a1_3, a2_0, a2_1, a2_2, a2_3, a3_0, a3_1, a3_2, a3_3;
  // weights are read once per launch (and the next launch's are long gone from the caches): non-temporal loads
#define AIC_LOAD_A(set_, c_)                                              \
  {                                                                        \
    const uint4* p_ = a_ptr + static_cast<int64_t>(c_) * S * 64;           \
    set_##_0 = ld_stream16(p_);                                            \
    set_##_1 = ld_stream16(p_ + 64);                                       \
    set_##_2 = ld_stream16(p_ + 128);                                      \
    set_##_3 = ld_stream16(p_ + 192);                                      \
  }
#define AIC_LOAD_X(c_)                                                        \
  { if (XQ) {                                                                 \
    const int64_t u0_ = (static_cast<int64_t>(ks0) + static_cast<int64_t>(c_) * S) * MT * 64 + tid; \
    const int64_t b0_ = xq_src(u0_);                                          \
    xr0 = X[b0_];                                                             \
    xr1 = X[b0_ + 16];      /* k8 + 1 of the same row: 16 units on (k8 & 3 is 0 or 2 here) */ \
    if (XV > 1) {                                                             \
      const int64_t b1_ = xq_src(u0_ + 256);                                  \
      xr2 = X[b1_];                                                           \
      xr3 = X[b1_ + 16];                                                      \
    }                                                                         \
  } else {                                                                    \
    const uint4* xp_ = x_ptr + static_cast<int64_t>(c_) * S * MT * 64;        \
    xr0 = xp_[0];                                                             \
    if (XV > 1) xr1 = xp_[256];                                               \
    if (XV > 2) xr2 = xp_[512];                                               \
    if (XV > 3) xr3 = xp_[768];                                               \
  } }
#define AIC_STORE_X(buf_)                    \
  { if (XQ) {                                \
    lds[buf_][tid] = xq_pack(xr0, xr1);      \
    if (XV > 1) lds[buf_][256 + tid] = xq_pack(xr2, xr3); \
  } else {                                   \
    lds[buf_][tid] = xr0;                    \
    if (XV > 1) lds[buf_][256 + tid] = xr1;  \
    if (XV > 2) lds[buf_][512 + tid] = xr2;  \
    if (XV > 3) lds[buf_][768 + tid] = xr3;  \
  } }
#define AIC_MMA_STEP(areg_, s_, buf_)                                                                             \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                             \
    const uint4 b = lds[buf_][((s_) * MT + mt) * 64 + lane];                                                      \
    if (FP8) {                                                                                                    \
      const long a_lo = static_cast<long>(areg_.x) | (static_cast<long>(areg_.y) << 32);                          \
      const long a_hi = static_cast<long>(areg_.z) | (static_cast<long>(areg_.w) << 32);                          \
      const long b_lo = static_cast<long>(b.x) | (static_cast<long>(b.y) << 32);                                  \
      const long b_hi = static_cast<long>(b.z) | (static_cast<long>(b.w) << 32);                                  \
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_lo, b_lo, acc[mt], 0, 0, 0);                         \
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a_hi, b_hi, acc[mt], 0, 0, 0);                         \
    } else {                                                                                                      \
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, areg_),                        \
                                                        __builtin_bit_cast(bf16x8, b), acc[mt], 0, 0, 0);         \
    }                                                                                                             \
  }
#define AIC_COMPUTE(set_, buf_) \
  AIC_MMA_STEP(set_##_0, 0, buf_) AIC_MMA_STEP(set_##_1, 1, buf_) AIC_MMA_STEP(set_##_2, 2, buf_) AIC_MMA_STEP(set_##_3, 3, buf_)
  // one pipeline stage: chunk c is computed from register set `cur_` and LDS buffer c & 1
#define AIC_STAGE(c_, cur_, refill_)                                  \
  if ((c_) < n_chunks) {                                               \
    if ((c_) + 1 < n_chunks) AIC_LOAD_X((c_) + 1)                      \
    if ((c_) + 3 < n_chunks) AIC_LOAD_A(refill_, (c_) + 3)             \
    AIC_COMPUTE(cur_, (c_) & 1)                                        \
    if ((c_) + 1 < n_chunks) AIC_STORE_X(((c_) + 1) & 1)               \
    __syncthreads();                                                   \
  }

  AIC_LOAD_X(0)
  AIC_LOAD_A(a0, 0)
  if (1 < n_chunks) AIC_LOAD_A(a1, 1)
  if (2 < n_chunks) AIC_LOAD_A(a2, 2)
  AIC_STORE_X(0)
  __syncthreads();
  for (int c = 0; c < n_chunks; c += 4) {
    AIC_STAGE(c, a0, a3)
    AIC_STAGE(c + 1, a1, a0)
    AIC_STAGE(c + 2, a2, a1)
    AIC_STAGE(c + 3, a3, a2)
  }
#undef AIC_STAGE
#undef AIC_COMPUTE
#undef AIC_MMA_STEP
#undef AIC_STORE_X
#undef AIC_LOAD_X
#undef AIC_LOAD_A

  // accumulator layout (16x16): column (batch row) = lane & 15, weight row = (lane >> 4) * 4 + reg
  const int n0 = rt * 16 + (lane >> 4) * 4;
  if (EPI == 0) {
    if (live) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + (lane & 15);
        float4 v = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
        *reinterpret_cast<float4*>(part + (static_cast<int64_t>(split) * (MT * 16) + m) * n_cols_out + n0) = v;
      }
    }
  } else {
    // the K loop ended with a barrier: the staging buffer is free to carry the cross-wave reduction
    float(*s_val)[MT * 16] = reinterpret_cast<float(*)[MT * 16]>(&lds[0][0]);
    int(*s_idx)[MT * 16] = reinterpret_cast<int(*)[MT * 16]>(&lds[1][0]);
    const float sc = FP8 ? (XQ ? xq_scale : *A.x_scale) * w_scale : 1.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + e;
        const float v = r(acc[mt][e] * sc);  // logits are a bf16 tensor in the reference
        if (live && n < n_valid_rows && (v > bv || (v == bv && n < bi))) {
          bv = v;
          bi = n;
        }
      }
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      if (lane < 16) {
        s_val[wave][mt * 16 + lane] = bv;
        s_idx[wave][mt * 16 + lane] = bi;
      }
    }
    __syncthreads();
    if (tid < MT * 16) {
      float bv = s_val[0][tid];
      int bi = s_idx[0][tid];
      for (int w = 1; w < 4; ++w) {
        const float ov = s_val[w][tid];
        const int oi = s_idx[w][tid];
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      const int64_t n_blocks = (A.n_rowtiles + 3) / 4;
      best_val[tid * n_blocks + bx] = bv;
      best_idx[tid * n_blocks + bx] = bi == 0x7fffffff ? 0x7fffffff : bi + row_offset;
    }
  }
}

template <bool FP8, int MT, int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
skinny_gemm_kernel(const uint4* __restrict__ W, const uint4* __restrict__ X, int n_rowtiles, int steps_total,
                   int steps_per_split, float* __restrict__ part, int n_cols_out, const float* __restrict__ x_scale,
                   float w_scale, int n_valid_rows, int row_offset, float* __restrict__ best_val,
                   int32_t* __restrict__ best_idx) {
  __shared__ uint4 lds[2][kChunkSteps * MT * 64];
  const GemmArgs A{W, X, n_rowtiles, steps_total, steps_per_split, part, n_cols_out, x_scale, w_scale, n_valid_rows,
                   row_offset, best_val, best_idx, nullptr};
  skinny_gemm_body<FP8, MT, EPI, false>(A, lds, blockIdx.x, blockIdx.y);
}

// The LM head of draft head h and the gate projection of head h + 1 in ONE launch: both read the state h_h and neither
// needs the other's result (the token of head h enters head h + 1 in the cell, after the projection), so the 134 MB gate
// projection — alone a 30 us launch at 4.4 TB/s, a third of it ramp and tail — rides inside the 0.5-1 GB head launch.
// Workgroups [0, gate_bx * gate_by) run the gate body (split-K partials), the rest the head body (arg-max epilogue;
// fp8 head: activations quantised on the way into LDS).
template <bool FP8H, int MT, bool XQ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
skinny_pair_kernel(GemmArgs G, GemmArgs Hd, int gate_bx, int gate_by) {
  __shared__ uint4 lds[2][kChunkSteps * MT * 64];
  const int b = blockIdx.x, n_gate = gate_bx * gate_by;
  if (b < n_gate)
    skinny_gemm_body<false, MT, 0, false>(G, lds, b % gate_bx, b / gate_bx);
  else
    skinny_gemm_body<FP8H, MT, 1, XQ>(Hd, lds, b - n_gate, 0);
}

// the head alone with on-the-fly activation quantisation (last draft head of the fused path)
template <int MT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
skinny_head_xq_kernel(GemmArgs Hd) {
  __shared__ uint4 lds[2][kChunkSteps * MT * 64];
  skinny_gemm_body<true, MT, 1, true>(Hd, lds, blockIdx.x, 0);
}

// ---- LSTM cell: everything between the gate projection and the LM head (arctic_speculator.py:667-689)
// one workgroup per batch row; r() marks every place the reference materialises a bf16 tensor
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// The cell is kCellParts workgroups per batch row with two cross-workgroup steps: the second normalisation needs a sum over
// the whole row, and the fp8 LM head's dynamic per-tensor scale needs the |max| over the whole batch.  History: one
// 1024-thread workgroup per row (r01-r02) did everything on 64 of the 256 CUs and was bound by its own arithmetic there (two
// erf, three exp, ~25 bf16 roundings per element: 26-33 us per head); r02-r03 cut the cell into two launches at the row sum
// plus a third for the quantisation (11.6 + 11.6 + 4.7 us per head, rocprofv3).  r04 found what those microseconds are:
// not arithmetic but DEPENDENT MEMORY ROUND TRIPS, ~2 us each on this part once a load leaves the XCD (the gate partials,
// the arg-max partials and the embedding row were written by other CUs or never read): the first one-launch form (counters
// in device memory between the phases, values in registers) still took 25.8 us, because it had kept the chain — arg-max
// partials -> token -> embedding row + partials -> block sum -> second-pass loads -> share swap -> counter -> poll -> shares
// -> third-pass loads -> atomicMax -> counter -> poll -> |max| — ten round trips.  This form issues every load that does
// not depend on the token at the very top (all gate partials of the thread's columns, layer-norm weights, old cell state,
// arg-max partials), the embedding row as soon as the token is known, and crosses workgroups through SLOTS instead of
// counters: a part stores its share (agent-scope atomic store) into its own slot, which ln0_kernel set to a sentinel at the
// start of the draft, and the readers poll the slots themselves — one store and one poll per step, nothing returned.
// Modes 0 and 1 are the same code cut at the row sum into two launches (the head-by-head / vocab-parallel entry points,
// and the A/B reference of mode 2: identical thread -> column mapping and reduction order, so the modes are bit-identical).
constexpr int kCellParts = 4;
constexpr int kCellThreads = 256;
constexpr int kCellMaxIter = 2;          // own columns per thread: 4 x kCellMaxIter (Ds <= 8192)
constexpr int kCellRowIter = 4 * kCellMaxIter;   // row-wide pass: 4 x kCellRowIter columns per thread
constexpr int kSyncHeads = 16;
constexpr int kSyncRowWords = 64 * kCellParts;   // per head: one row-sum share per (row, part) ...
constexpr int kSyncStride = 2 * kSyncRowWords;   // ... and one |max| per workgroup (row, part)
constexpr unsigned int kSyncEmpty = 0xffffffffu; // not a value a share or a |max| can take (both are non-negative floats)

struct CellArgs {
  const float* part;   // [n_splits][m_pad][4 Ds] gate projection partials
  int n_splits, m_pad, batch;
  const int32_t* tokens;
  const uint16_t* emb;
  int vocab_rows;
  float alpha;
  int Ds;
};

__device__ __forceinline__ float cell_sigmoid(float x) { return r(1.0f / (1.0f + expf(-x))); }

// sum over the split-K partials of gate columns n .. n + 3 of row m (r(s) of it is the bf16 projection output)
__device__ __forceinline__ float4 cell_part4(const CellArgs& a, int m, int n) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = 0; sp < a.n_splits; ++sp) {
    const float4 p = *reinterpret_cast<const float4*>(a.part + (static_cast<int64_t>(sp) * a.m_pad + m) * (4 * a.Ds) + n);
    s.x += p.x;
    s.y += p.y;
    s.z += p.z;
    s.w += p.w;
  }
  return s;
}
__device__ __forceinline__ void unpack4(uint2 v, float (&f)[4]) {
  f[0] = bf16_to_f32(static_cast<uint16_t>(v.x & 0xffff));
  f[1] = bf16_to_f32(static_cast<uint16_t>(v.x >> 16));
  f[2] = bf16_to_f32(static_cast<uint16_t>(v.y & 0xffff));
  f[3] = bf16_to_f32(static_cast<uint16_t>(v.y >> 16));
}
// torch.add(states, z, alpha=emb_weight / state_weight) for four gate columns (s = their projection sums, zz = z's columns)
__device__ __forceinline__ void cell_added4(float alpha, float4 s, uint2 zz, float (&out)[4]) {
  float zf[4];
  unpack4(zz, zf);
  out[0] = r(fmaf(alpha, zf[0], r(s.x)));
  out[1] = r(fmaf(alpha, zf[1], r(s.y)));
  out[2] = r(fmaf(alpha, zf[2], r(s.z)));
  out[3] = r(fmaf(alpha, zf[3], r(s.w)));
}

// Fused draft path (PrevArgmax.best_val != nullptr): the arg-max over the previous head's per-workgroup partials is
// finished HERE — every part of a row reduces the same n_blocks candidates to the same token (ties to the lowest index, so
// the order of the reduction does not matter), part 0 publishes it (tokens[m], the draft's output column) — instead of in
// a launch of its own between the LM head and this cell.
struct PrevArgmax {
  const float* best_val;      // [m_pad][n_blocks] or nullptr: the token is tokens[m] already
  const int32_t* best_idx;
  int n_blocks;
  int32_t* tokens;            // [m_pad] written by part 0
  int64_t* out_tokens;        // [batch][out_stride] or nullptr
  float* out_vals;
  int out_stride, out_col;
};

__device__ __forceinline__ void block_argmax(float& bv, int& bi, float* s_v, int* s_i) {
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_v[wave] = bv;
    s_i[wave] = bi;
  }
  __syncthreads();
  bv = s_v[0];
  bi = s_i[0];
  for (int w = 1; w < n_waves; ++w)
    if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
      bv = s_v[w];
      bi = s_i[w];
    }
  __syncthreads();
}

// Cross-workgroup steps of mode 2: agent-scope atomic stores and loads only (coherent across the XCDs' L2s by themselves:
// no release / acquire fence, i.e. no L2 write-back or invalidate).  Every slot a poll waits for belongs to a live workgroup
// of the same launch that stores into it unconditionally; all 4 x m_pad workgroups of the launch are resident together
// (256 threads, 49 VGPRs).  A poll still gives up after ~2^22 rounds rather than hang the device should that be violated.
__device__ __forceinline__ void slot_store(unsigned int* slot, unsigned int bits) {
  __hip_atomic_store(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int slot_poll(const unsigned int* slot) {
  unsigned int v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int rounds = 0; v == kSyncEmpty && rounds < (1 << 22); ++rounds) {
    __builtin_amdgcn_s_sleep(1);
    v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return v;
}

struct CellSync {
  unsigned int* row_share;    // [m_pad][kCellParts] kSyncEmpty at the start of the draft (ln0_kernel), then the share's bits
  unsigned int* wg_amax;      // [m_pad][kCellParts] likewise: the workgroup's |max| bits
  uint4* h_fp8;               // fragment-major e4m3 activations (fp8 LM head) or nullptr
  float* x_scale;
  int64_t* trace;             // debug (aic_debug_lstm_cell_trace): per workgroup 8 timestamps (100 MHz) at the phase boundaries
};

// mode 0: gates + new cell state, row-sum shares to ss2_part (stops there);  mode 1: state from `cell` + ss2_part (a launch
// boundary lies between 0 and 1);  mode 2: both in one launch, + the fp8 head's activation quantisation when sy.h_fp8.
// Workgroup (row m, part q) owns columns [q Ds/4, (q+1) Ds/4); thread t of it the four columns 4 t .. 4 t + 3 of every
// 1024-column block of that range (and of the whole row, for the first normalisation's sum).
__global__ void __launch_bounds__(kCellThreads)
lstm_cell_kernel(CellArgs a, int mode, const uint16_t* __restrict__ cln_w, const uint16_t* __restrict__ cln_b,
                 const uint16_t* __restrict__ sln_w, const uint16_t* __restrict__ sln_b, uint16_t* __restrict__ cell,
                 float* __restrict__ ss2_part, PrevArgmax pa, int MT, uint4* __restrict__ h_out,
                 unsigned int* __restrict__ amax_bits, CellSync sy) {
  __shared__ float sh[16];
  __shared__ float s_v[16];
  __shared__ int s_i[16];
  __shared__ float s_share[kCellParts];
  const int m = blockIdx.x, q = blockIdx.y, Ds = a.Ds;
  const int j0 = q * (Ds / kCellParts), j1 = j0 + Ds / kCellParts;
  const int tid = threadIdx.x;
  if (m >= a.batch) {
    // rows of the padded batch: zero activations for the GEMMs that follow (no part in any cross-workgroup step)
    if (mode != 0)
      for (int j = j0 + 4 * tid; j < j1; j += 4 * kCellThreads) {
        uint2* hp = reinterpret_cast<uint2*>(h_out + xunit_bf16(m, j >> 3, MT)) + ((j >> 2) & 1);
        *hp = make_uint2(0u, 0u);
        if (mode == 2 && sy.h_fp8) reinterpret_cast<uint32_t*>(sy.h_fp8 + xunit_fp8(m, j >> 4, MT))[(j >> 2) & 3] = 0u;
      }
    return;
  }
  const bool do_a = mode != 1, do_b = mode != 0;
  int64_t* tr = sy.trace ? sy.trace + (static_cast<int64_t>(m) * kCellParts + q) * 12 : nullptr;
#define AIC_STAMP(i_) if (tr && tid == 0) tr[i_] = static_cast<int64_t>(__builtin_amdgcn_s_memrealtime());
  AIC_STAMP(0)

  // ---- 1. every load that does not depend on the token, issued together (one memory round trip) ----
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  const bool finish_argmax = do_a && pa.best_val != nullptr;
  if (finish_argmax) {
    // eight (value, index) pairs per thread in flight at once (a load-compare-load loop ran at one memory round trip per
    // iteration: +2.7 us on the heads that finish an arg-max, in-kernel trace)
    const float* pv = pa.best_val + static_cast<int64_t>(m) * pa.n_blocks;
    const int32_t* pi = pa.best_idx + static_cast<int64_t>(m) * pa.n_blocks;
    for (int b0 = 0; b0 < pa.n_blocks; b0 += 8 * kCellThreads) {
      float v[8];
      int ix[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u * kCellThreads + tid;
        const bool ok = b < pa.n_blocks;
        v[u] = ok ? pv[b] : -INFINITY;
        ix[u] = ok ? pi[b] : 0x7fffffff;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u] > bv || (v[u] == bv && ix[u] < bi)) {
          bv = v[u];
          bi = ix[u];
        }
    }
  }
  float4 row_c[kCellRowIter];                        // candidate-gate projection sums of the whole row (first normalisation)
  float4 own_c[kCellMaxIter], own_i[kCellMaxIter], own_f[kCellMaxIter], own_o[kCellMaxIter];
  uint2 w_cln[kCellMaxIter], b_cln[kCellMaxIter], w_sln[kCellMaxIter], b_sln[kCellMaxIter], c_old[kCellMaxIter];
  if (do_a) {
#pragma unroll
    for (int i = 0; i < kCellRowIter; ++i) {
      const int j = 4 * tid + i * 4 * kCellThreads;
      if (j < Ds) row_c[i] = cell_part4(a, m, 3 * Ds + j);
    }
  }
#pragma unroll
  for (int it = 0; it < kCellMaxIter; ++it) {
    const int j = j0 + 4 * tid + it * 4 * kCellThreads;
    if (j >= j1) continue;
    if (do_a) {
      own_c[it] = cell_part4(a, m, 3 * Ds + j);
      own_i[it] = cell_part4(a, m, Ds + j);
      own_f[it] = cell_part4(a, m, j);
      w_cln[it] = *reinterpret_cast<const uint2*>(cln_w + j);
      b_cln[it] = *reinterpret_cast<const uint2*>(cln_b + j);
    }
    if (do_b) {
      own_o[it] = cell_part4(a, m, 2 * Ds + j);
      w_sln[it] = *reinterpret_cast<const uint2*>(sln_w + j);
      b_sln[it] = *reinterpret_cast<const uint2*>(sln_b + j);
    }
    c_old[it] = *reinterpret_cast<const uint2*>(cell + static_cast<int64_t>(m) * Ds + j);
  }

  // ---- 2. the token, then its embedding row (second round trip) ----
  AIC_STAMP(1)
  int tok;
  if (finish_argmax) {
    block_argmax(bv, bi, s_v, s_i);
    tok = bi;
    if (q == 0 && tid == 0) {
      pa.tokens[m] = bi;
      if (pa.out_tokens) pa.out_tokens[static_cast<int64_t>(m) * pa.out_stride + pa.out_col] = bi;
      if (pa.out_vals) pa.out_vals[static_cast<int64_t>(m) * pa.out_stride + pa.out_col] = bv;
    }
  } else {
    tok = a.tokens[m];
  }
  if (tok < 0 || tok >= a.vocab_rows) tok = 0;  // never read outside the table
  const uint16_t* z = a.emb + static_cast<int64_t>(tok) * Ds;
  uint2 z_row[kCellRowIter], z_own[kCellMaxIter];
  if (do_a) {
#pragma unroll
    for (int i = 0; i < kCellRowIter; ++i) {
      const int j = 4 * tid + i * 4 * kCellThreads;
      if (j < Ds) z_row[i] = *reinterpret_cast<const uint2*>(z + j);
    }
  }
#pragma unroll
  for (int it = 0; it < kCellMaxIter; ++it) {
    const int j = j0 + 4 * tid + it * 4 * kCellThreads;
    if (j < j1) z_own[it] = *reinterpret_cast<const uint2*>(z + j);
  }

  float cn[kCellMaxIter][4];     // new cell state of this thread's columns (bf16 values)
  AIC_STAMP(2)
  if (do_a) {
    // first normalisation: its row sum is computed by every part for itself
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < kCellRowIter; ++i) {
      const int j = 4 * tid + i * 4 * kCellThreads;
      if (j >= Ds) continue;
      float c[4];
      cell_added4(a.alpha, row_c[i], z_row[i], c);
#pragma unroll
      for (int e = 0; e < 4; ++e) ss += r(c[e] * c[e]);
    }
    ss = block_sum(ss, sh);
    AIC_STAMP(3)
    const float rs = r(rsqrtf(r(r(ss / static_cast<float>(Ds)) + 1e-6f)));
    float ss2 = 0.0f;
#pragma unroll
    for (int it = 0; it < kCellMaxIter; ++it) {
      const int j = j0 + 4 * tid + it * 4 * kCellThreads;
      if (j >= j1) continue;
      float c[4], gi[4], gf[4], wv[4], bw[4], ov[4];
      cell_added4(a.alpha, own_c[it], z_own[it], c);
      cell_added4(a.alpha, own_i[it], z_own[it], gi);
      cell_added4(a.alpha, own_f[it], z_own[it], gf);
      unpack4(w_cln[it], wv);
      unpack4(b_cln[it], bw);
      unpack4(c_old[it], ov);
      uint16_t nb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = r(c[e] * rs);
        y = r(wv[e] * y);
        y = r(y + bw[e]);
        const float cand = r(r(gelu_erf(y)) * cell_sigmoid(gi[e]));           // * input gate
        const float kept = r(ov[e] * cell_sigmoid(gf[e]));                    // * forget gate
        const float cnew = r(kept + cand);
        nb[e] = f32_to_bf16(cnew);
        ss2 += r(cnew * cnew);
        cn[it][e] = cnew;
      }
      c_old[it] = make_uint2(static_cast<uint32_t>(nb[0]) | (static_cast<uint32_t>(nb[1]) << 16),
                             static_cast<uint32_t>(nb[2]) | (static_cast<uint32_t>(nb[3]) << 16));
    }
    ss2 = block_sum(ss2, sh);
    // (global stores go out BEHIND the block reductions: a barrier waits for every store issued before it)
#pragma unroll
    for (int it = 0; it < kCellMaxIter; ++it) {
      const int j = j0 + 4 * tid + it * 4 * kCellThreads;
      if (j < j1) *reinterpret_cast<uint2*>(cell + static_cast<int64_t>(m) * Ds + j) = c_old[it];
    }
    if (mode == 0) {
      if (tid == 0) ss2_part[m * kCellParts + q] = ss2;
      return;
    }
    // the row's parts meet: own share into its slot, then poll the four slots (lanes 0-3 of the first wave, one each)
    AIC_STAMP(4)
    if (tid == 0) slot_store(sy.row_share + m * kCellParts + q, __float_as_uint(ss2));
    if (tid < kCellParts) s_share[tid] = __uint_as_float(slot_poll(sy.row_share + m * kCellParts + tid));
    __syncthreads();
  } else {
    if (tid < kCellParts) s_share[tid] = ss2_part[m * kCellParts + tid];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kCellMaxIter; ++it) unpack4(c_old[it], cn[it]);
  }
  // second normalisation: the row's mean square is the sum of the kCellParts shares in part order
  AIC_STAMP(5)
  float ss2 = 0.0f;
  for (int p = 0; p < kCellParts; ++p) ss2 += s_share[p];
  const float rs2 = r(rsqrtf(r(r(ss2 / static_cast<float>(Ds)) + 1e-6f)));
  float amax = 0.0f;
  uint32_t hb[kCellMaxIter][2];   // this thread's state values as packed bf16 pairs (for the quantisation)
#pragma unroll
  for (int it = 0; it < kCellMaxIter; ++it) {
    const int j = j0 + 4 * tid + it * 4 * kCellThreads;
    if (j >= j1) continue;
    float go[4], wv[4], bw[4];
    cell_added4(a.alpha, own_o[it], z_own[it], go);
    unpack4(w_sln[it], wv);
    unpack4(b_sln[it], bw);
    uint16_t h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y = r(cn[it][e] * rs2);
      y = r(wv[e] * y);
      y = r(y + bw[e]);
      const float st = r(r(gelu_erf(y)) * cell_sigmoid(go[e]));  // * output gate
      h[e] = f32_to_bf16(st);
      amax = fmaxf(amax, fabsf(st));
    }
    hb[it][0] = static_cast<uint32_t>(h[0]) | (static_cast<uint32_t>(h[1]) << 16);
    hb[it][1] = static_cast<uint32_t>(h[2]) | (static_cast<uint32_t>(h[3]) << 16);
  }
  AIC_STAMP(8)
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((tid & 63) == 0) s_v[tid >> 6] = amax;
  AIC_STAMP(9)
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kCellMaxIter; ++it) {
    const int j = j0 + 4 * tid + it * 4 * kCellThreads;
    if (j < j1) reinterpret_cast<uint2*>(h_out + xunit_bf16(m, j >> 3, MT))[(j >> 2) & 1] = make_uint2(hb[it][0], hb[it][1]);
  }
  AIC_STAMP(6)
  const bool quantise = mode == 2 && sy.h_fp8 != nullptr;
  if (tid == 0) {
    float wg = s_v[0];
    for (int w = 1; w < kCellThreads / 64; ++w) wg = fmaxf(wg, s_v[w]);
    // the batch |max| for a reader behind the launch (quant_act_kernel, the XQ head): ONE atomic per workgroup — one per
    // wave (512 on one address at 32 rows) serialised into ~6 us that the next barrier's vmcnt(0) then waited for (in-kernel
    // trace, r04).  With the quantisation in this launch the slots carry it instead.
    if (quantise)
      slot_store(sy.wg_amax + m * kCellParts + q, __float_as_uint(wg));
    else
      atomicMax(amax_bits, __float_as_uint(wg));
  }
  if (!quantise) return;
  // ---- dynamic per-tensor quantisation for the fp8 LM head (fp8.py:303-308; quant_act_kernel's arithmetic on the values
  // this thread still holds): needs the |max| over the WHOLE batch -> every live workgroup has stored its own into its slot,
  // and every workgroup polls all of them (thread t slot t: batch x kCellParts <= 256 slots) and takes the maximum
  float all = 0.0f;
  if (tid < a.batch * kCellParts) all = __uint_as_float(slot_poll(sy.wg_amax + tid));
  for (int off = 32; off > 0; off >>= 1) all = fmaxf(all, __shfl_xor(all, off));
  __syncthreads();                               // (s_v is reused)
  if ((tid & 63) == 0) s_v[tid >> 6] = all;
  __syncthreads();
  all = s_v[0];
  for (int w = 1; w < kCellThreads / 64; ++w) all = fmaxf(all, s_v[w]);
  const float scale = fmaxf(all / 448.0f, 1.0f / (448.0f * 512.0f));
  const float inv = 1.0f / scale;
  if (m == 0 && q == 0 && tid == 0) *sy.x_scale = scale;
  AIC_STAMP(7)
#undef AIC_STAMP
#pragma unroll
  for (int it = 0; it < kCellMaxIter; ++it) {
    const int j = j0 + 4 * tid + it * 4 * kCellThreads;
    if (j >= j1) continue;
    const uint32_t p0 = hb[it][0], p1 = hb[it][1];
    const float f0 = clamp448(bf16_to_f32(static_cast<uint16_t>(p0 & 0xffff)) * inv);
    const float f1 = clamp448(bf16_to_f32(static_cast<uint16_t>(p0 >> 16)) * inv);
    const float f2 = clamp448(bf16_to_f32(static_cast<uint16_t>(p1 & 0xffff)) * inv);
    const float f3 = clamp448(bf16_to_f32(static_cast<uint16_t>(p1 >> 16)) * inv);
    reinterpret_cast<uint32_t*>(sy.h_fp8 + xunit_fp8(m, j >> 4, MT))[(j >> 2) & 3] = pack4_fp8(f0, f1, f2, f3);
  }
}

// ---- MLP speculator head (ArcticMLPSpeculator.generate_states, arctic_speculator.py:264-283): everything between
// the projection and the LM head: states = proj(h) + (emb_weight / state_weight) z; states = gelu(ln(states)).
// ln is MLPSpeculatorLayerNorm with scale and shift, evaluated in the tensor's dtype (bf16): r() at every op.
__global__ void __launch_bounds__(1024)
mlp_cell_kernel(const float* __restrict__ part, int n_splits, int m_pad, int batch, const int32_t* __restrict__ tokens,
                const uint16_t* __restrict__ emb, int vocab_rows, float alpha, const uint16_t* __restrict__ ln_w,
                const uint16_t* __restrict__ ln_b, int Ds, int MT, uint4* __restrict__ h_out,
                unsigned int* __restrict__ amax_bits, int ext_rows) {
  extern __shared__ float smem[];  // [Ds]
  __shared__ float sh[16];
  const int m = blockIdx.x;
  if (m >= batch) {
    for (int k8 = threadIdx.x; k8 < Ds / 8; k8 += blockDim.x) h_out[xunit_bf16(m, k8, MT)] = make_uint4(0, 0, 0, 0);
    return;
  }
  int tok = tokens[m];
  if (tok < 0 || tok >= vocab_rows) tok = 0;
  // ext_rows: `emb` is [batch][Ds], the embedding rows already looked up (vocab-sharded table + all-reduce, C9)
  const uint16_t* z = emb + static_cast<int64_t>(ext_rows ? m : tok) * Ds;
  float ss = 0.0f;
  for (int j = threadIdx.x; j < Ds; j += blockDim.x) {
    float s = 0.0f;
    for (int sp = 0; sp < n_splits; ++sp) s += part[(static_cast<int64_t>(sp) * m_pad + m) * Ds + j];
    const float a = r(fmaf(alpha, bf16_to_f32(z[j]), r(s)));   // states.add_(z, alpha=...)
    smem[j] = a;
    ss += r(a * a);
  }
  ss = block_sum(ss, sh);
  const float rs = r(rsqrtf(r(r(ss / static_cast<float>(Ds)) + 1e-6f)));
  float amax = 0.0f;
  for (int k8 = threadIdx.x; k8 < Ds / 8; k8 += blockDim.x) {
    uint16_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = k8 * 8 + e;
      float y = r(smem[j] * rs);
      y = r(bf16_to_f32(ln_w[j]) * y);
      y = r(y + bf16_to_f32(ln_b[j]));
      const float st = r(gelu_erf(y));
      h[e] = f32_to_bf16(st);
      amax = fmaxf(amax, fabsf(st));
    }
    h_out[xunit_bf16(m, k8, MT)] = *reinterpret_cast<uint4*>(h);
  }
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(amax));
}

// ---- stacked "sum_rnn" stages (ArcticLSTMSpeculator with multi-entry dimension lists, arctic_speculator.py:478-542):
// emb / proj are nn.Sequentials [base, (LayerNorm, GELU, Linear)*], ln is [LayerNorm, (GELU, Linear, LayerNorm)*].  One
// generic row kernel covers every elementwise piece between two Linears:
//     x = A  [+ alpha * B]   ->   [LayerNorm(w, b)]   ->   [GELU]   ->   fragment-major bf16 for the next GEMM
// where A / B are either split-K partials of a Linear (summed in fp32 and rounded: the Linear's bf16 output) or bf16 rows
// (an embedding table gathered by token, or rows looked up by the caller).  Every op rounds to bf16 like the reference.
struct StageSrc {
  const float* part;        // [n_splits][m_pad][Ds] or nullptr
  int n_splits;
  const uint16_t* rows;     // bf16 [*, Ds]: row `token` (by_token) or row m
  int by_token;
};
__device__ __forceinline__ float stage_value(const StageSrc& a, int m, int tok, int m_pad, int Ds, int j) {
  if (a.part != nullptr) {
    float s = 0.0f;
    for (int sp = 0; sp < a.n_splits; ++sp) s += a.part[(static_cast<int64_t>(sp) * m_pad + m) * Ds + j];
    return r(s);
  }
  return bf16_to_f32(a.rows[static_cast<int64_t>(a.by_token ? tok : m) * Ds + j]);
}
__global__ void __launch_bounds__(1024)
stage_kernel(StageSrc a, StageSrc b, int has_b, float alpha, const int32_t* __restrict__ tokens, int m_pad, int batch,
             const uint16_t* __restrict__ ln_w, const uint16_t* __restrict__ ln_b, int do_ln, int do_gelu, int Ds, int MT,
             uint4* __restrict__ out_frag, unsigned int* __restrict__ amax_bits) {
  extern __shared__ float smem[];  // [Ds]
  __shared__ float sh[16];
  const int m = blockIdx.x;
  if (m >= batch) {
    for (int k8 = threadIdx.x; k8 < Ds / 8; k8 += blockDim.x) out_frag[xunit_bf16(m, k8, MT)] = make_uint4(0, 0, 0, 0);
    return;
  }
  int tok = tokens ? tokens[m] : 0;
  if (tok < 0) tok = 0;
  float ss = 0.0f;
  for (int j = threadIdx.x; j < Ds; j += blockDim.x) {
    float x = stage_value(a, m, tok, m_pad, Ds, j);
    if (has_b) x = r(fmaf(alpha, stage_value(b, m, tok, m_pad, Ds, j), x));   // states.add_(z, alpha=...)
    smem[j] = x;
    ss += r(x * x);
  }
  float rs = 1.0f;
  if (do_ln) {
    ss = block_sum(ss, sh);
    rs = r(rsqrtf(r(r(ss / static_cast<float>(Ds)) + 1e-6f)));
  } else {
    __syncthreads();
  }
  float amax = 0.0f;
  for (int k8 = threadIdx.x; k8 < Ds / 8; k8 += blockDim.x) {
    uint16_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = k8 * 8 + e;
      float y = smem[j];
      if (do_ln) {
        y = r(y * rs);
        y = r(bf16_to_f32(ln_w[j]) * y);
        y = r(y + bf16_to_f32(ln_b[j]));
      }
      if (do_gelu) y = r(gelu_erf(y));
      h[e] = f32_to_bf16(y);
      amax = fmaxf(amax, fabsf(y));
    }
    out_frag[xunit_bf16(m, k8, MT)] = *reinterpret_cast<uint4*>(h);
  }
  if (amax_bits) {
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(amax));
  }
}

// ---- dynamic per-tensor activation quantisation for the fp8 LM head (fp8.py:303-308) ------------
__global__ void __launch_bounds__(256)
quant_act_kernel(const uint4* __restrict__ h_bf16, uint4* __restrict__ h_fp8, const unsigned int* __restrict__ amax_bits,
                 float* __restrict__ x_scale, int Ds, int MT) {
  const float scale = fmaxf(__uint_as_float(*amax_bits) / 448.0f, 1.0f / (448.0f * 512.0f));
  const float inv = 1.0f / scale;
  if (blockIdx.x == 0 && threadIdx.x == 0) *x_scale = scale;
  const int total = (Ds / 16) * MT * 16;  // fp8 16-byte units
  for (int u = blockIdx.x * 256 + threadIdx.x; u < total; u += gridDim.x * 256) {
    // unit u = ((kp * MT + mt) * 64 + g * 16 + mm) covers k = kp*64 + 16 g .. +16 of row mt*16 + mm
    const int mm = u & 15, g = (u >> 4) & 3;
    const int t = u >> 6;
    const int mt = t % MT, kp = t / MT;
    const int m = mt * 16 + mm;
    const int k8 = (kp * 64 + 16 * g) / 8;
    uint16_t h[16];
    *reinterpret_cast<uint4*>(h) = h_bf16[xunit_bf16(m, k8, MT)];
    *reinterpret_cast<uint4*>(h + 8) = h_bf16[xunit_bf16(m, k8 + 1, MT)];
    float f[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) f[e] = clamp448(bf16_to_f32(h[e]) * inv);
    uint4 o;
    o.x = pack4_fp8(f[0], f[1], f[2], f[3]);
    o.y = pack4_fp8(f[4], f[5], f[6], f[7]);
    o.z = pack4_fp8(f[8], f[9], f[10], f[11]);
    o.w = pack4_fp8(f[12], f[13], f[14], f[15]);
    h_fp8[u] = o;
  }
}

// ---- final arg-max over the per-workgroup partials; feeds the next head ---------------------------
__global__ void __launch_bounds__(256)
argmax_finish_kernel(const float* __restrict__ best_val, const int32_t* __restrict__ best_idx, int n_blocks, int m_pad,
                     int batch, int32_t* __restrict__ tokens, int64_t* __restrict__ out_tokens, int out_stride,
                     int out_col, float* __restrict__ out_vals) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int m = blockIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  const float* pv = best_val + static_cast<int64_t>(m) * n_blocks;
  const int32_t* pi = best_idx + static_cast<int64_t>(m) * n_blocks;
  for (int b0 = 0; b0 < n_blocks; b0 += 8 * 256) {     // eight pairs per thread in flight (see lstm_cell_kernel)
    float v[8];
    int ix[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + u * 256 + static_cast<int>(threadIdx.x);
      const bool ok = b < n_blocks;
      v[u] = ok ? pv[b] : -INFINITY;
      ix[u] = ok ? pi[b] : 0x7fffffff;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v[u] > bv || (v[u] == bv && ix[u] < bi)) {
        bv = v[u];
        bi = ix[u];
      }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    s_v[threadIdx.x >> 6] = bv;
    s_i[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) {
        bv = s_v[w];
        bi = s_i[w];
      }
    if (m < batch) {
      tokens[m] = bi;
      if (out_tokens) out_tokens[static_cast<int64_t>(m) * out_stride + out_col] = bi;
      if (out_vals) out_vals[static_cast<int64_t>(m) * out_stride + out_col] = bv;
    }
  }
}

__global__ void copy_tokens_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

}  // namespace aic

using namespace aic;

struct aic_lstm {
  aic_lstm_config cfg;
  aic_lstm_weights w;
  float alpha;
  // repacked weights (owned)
  uint4 *proj0_t = nullptr, *proj1_t = nullptr, *head_t = nullptr, *head8_t = nullptr;
  float head8_scale = 0.0f;
  int gate_rowtiles = 0, head_rowtiles = 0, head_blocks = 0;
  // per-call state (owned)
  int max_mt = 0;
  uint4 *x0 = nullptr, *h_bf16 = nullptr, *h_fp8 = nullptr;
  uint16_t* cell = nullptr;
  float* part = nullptr;
  float* ss2_part = nullptr;   // [64][kCellParts] row-sum shares of the cell's parts
  unsigned int* sync = nullptr;   // [kSyncHeads][kSyncStride] slots of the one-launch cell (emptied by ln0_kernel)
  float* best_val = nullptr;
  int32_t* best_idx = nullptr;
  int32_t* tokens = nullptr;
  unsigned int* amax = nullptr;
  float* x_scale = nullptr;
  int gate_splits = 2;
  int cur_mt = 0, cur_batch = 0;
  // MLP speculator mode (aic_mlp_create): per-head tables; tied heads point at the same repacked copy
  bool mlp = false;
  int mlp_heads = 0;
  uint4* mlp_proj_t[8] = {};
  uint4* mlp_head_t[8] = {};
  uint4* mlp_head8_t[8] = {};
  float mlp_head8_scale[8] = {};
  const uint16_t* mlp_emb[8] = {};
  const uint16_t* ext_rows = nullptr;   // [batch][Ds] embedding rows of the current head, looked up by the caller (C9)
  const uint16_t* mlp_ln_w[8] = {};
  const uint16_t* mlp_ln_b[8] = {};
  std::vector<void*> mlp_owned;
  // stacked sum_rnn stages (aic_mlp_create_stacked): per head and stage the LayerNorm parameters (caller-owned) and the
  // fragment-major copy of the Linear (owned); a second partial buffer and a fragment-major scratch activation
  int st_emb = 0, st_proj = 0, st_ln = 0;
  const uint16_t* st_emb_ln_w[8][3] = {};
  const uint16_t* st_emb_ln_b[8][3] = {};
  uint4* st_emb_lin[8][3] = {};
  const uint16_t* st_proj_ln_w[8][3] = {};
  const uint16_t* st_proj_ln_b[8][3] = {};
  uint4* st_proj_lin[8][3] = {};
  uint4* st_ln_lin[8][3] = {};
  const uint16_t* st_ln_ln_w[8][3] = {};
  const uint16_t* st_ln_ln_b[8][3] = {};
  float* part2 = nullptr;
  uint4* tmp_frag = nullptr;
};

static int pad_mt(int batch) { return batch <= 16 ? 1 : (batch <= 32 ? 2 : 4); }

template <bool FP8, int EPI>
static int launch_gemm(int mt, dim3 grid, hipStream_t s, const uint4* W, const uint4* X, int rowtiles, int steps_total,
                       int steps_per_split, float* part, int n_cols, const float* x_scale, float w_scale, int n_valid,
                       int row_offset, float* bv, int32_t* bi) {
#define AIC_GEMM(MT)                                                                                              \
  hipLaunchKernelGGL((skinny_gemm_kernel<FP8, MT, EPI>), grid, dim3(256), 0, s, W, X, rowtiles, steps_total,      \
                     steps_per_split, part, n_cols, x_scale, w_scale, n_valid, row_offset, bv, bi)
  if (mt == 1) AIC_GEMM(1);
  else if (mt == 2) AIC_GEMM(2);
  else AIC_GEMM(4);
#undef AIC_GEMM
  return launch_status("skinny_gemm_kernel");
}

// LM head + arg-max of an MLP-class speculator from the state in h_bf16 (amax in m->amax[head])
static int run_mlp_lm_head(aic_lstm* m, int head_index, hipStream_t s, int64_t* out_tokens, int out_stride, int out_col,
                           float* out_vals) {
  const aic_lstm_config& c = m->cfg;
  const int mt = m->cur_mt, mpad = mt * 16, B = m->cur_batch, Ds = c.inner_dim;
  int rc;
  const bool fp8m = m->mlp_head8_t[head_index] && mpad <= c.head_fp8_max_batch;
  dim3 hgrid(m->head_blocks, 1);
  if (fp8m) {
    hipLaunchKernelGGL(quant_act_kernel, dim3(64), dim3(256), 0, s, m->h_bf16, m->h_fp8, m->amax + head_index, m->x_scale, Ds,
                       mt);
    if ((rc = launch_status("quant_act_kernel")) != AIC_OK) return rc;
    rc = launch_gemm<true, 1>(mt, hgrid, s, m->mlp_head8_t[head_index], m->h_fp8, m->head_rowtiles, Ds / 64, Ds / 64, nullptr,
                              0, m->x_scale, m->mlp_head8_scale[head_index], c.vocab_size, c.vocab_offset, m->best_val,
                              m->best_idx);
  } else {
    rc = launch_gemm<false, 1>(mt, hgrid, s, m->mlp_head_t[head_index], m->h_bf16, m->head_rowtiles, Ds / 32, Ds / 32, nullptr,
                               0, nullptr, 1.0f, c.vocab_size, c.vocab_offset, m->best_val, m->best_idx);
  }
  if (rc != AIC_OK) return rc;
  hipLaunchKernelGGL(argmax_finish_kernel, dim3(mpad), dim3(256), 0, s, m->best_val, m->best_idx, m->head_blocks, mpad, B,
                     m->tokens, out_tokens, out_stride, out_col, out_vals);
  return launch_status("argmax_finish_kernel");
}

// generate_states of a stacked sum_rnn head (arctic_speculator.py:693-703 over the Sequentials of :478-542): leaves the new
// state fragment-major in h_bf16 (+ its amax), like mlp_cell_kernel does for the plain form
static int run_stacked_states(aic_lstm* m, int head, hipStream_t s) {
  const aic_lstm_config& c = m->cfg;
  const int mt = m->cur_mt, mpad = mt * 16, B = m->cur_batch, Ds = c.inner_dim;
  const bool first = head == 0;
  const int K0 = first ? c.input_hidden_dim : Ds;
  int rc;
  auto splits_for = [&](int K) {
    int sp = m->gate_splits;
    while (sp > 1 && ((K / 32) % (sp * kChunkSteps) != 0)) --sp;
    return sp;
  };
  auto gemm = [&](const uint4* W, const uint4* X, int K, float* part, int* splits_out) -> int {
    const int sp = splits_for(K);
    *splits_out = sp;
    const int rowtiles = Ds / 16;
    return launch_gemm<false, 0>(mt, dim3((rowtiles + 3) / 4, sp), s, W, X, rowtiles, K / 32, (K / 32) / sp, part, Ds, nullptr,
                                 1.0f, 0, 0, nullptr, nullptr);
  };
  auto stage = [&](StageSrc a, StageSrc b, int has_b, const uint16_t* lw, const uint16_t* lb, int do_ln, int do_gelu, uint4* out,
                   unsigned int* amax) -> int {
    hipLaunchKernelGGL(stage_kernel, dim3(mpad), dim3(1024), Ds * sizeof(float), s, a, b, has_b, m->alpha, m->tokens, mpad, B, lw,
                       lb, do_ln, do_gelu, Ds, mt, out, amax);
    return launch_status("stage_kernel");
  };
  const StageSrc none{nullptr, 0, nullptr, 0};
  // z = emb(last_tokens): the table row (or the caller's looked-up rows), then (LayerNorm, GELU, Linear)*
  const uint16_t* emb_src = m->ext_rows ? m->ext_rows : m->mlp_emb[head];
  AIC_REQUIRE(emb_src, "head %d has no embedding table and no looked-up rows (aic_mlp_set_embedding_rows)", head);
  StageSrc z{nullptr, 0, emb_src, m->ext_rows ? 0 : 1};
  int z_splits = 1;
  for (int j = 0; j < m->st_emb; ++j) {
    if ((rc = stage(z, none, 0, m->st_emb_ln_w[head][j], m->st_emb_ln_b[head][j], 1, 1, m->tmp_frag, nullptr)) != AIC_OK) return rc;
    if ((rc = gemm(m->st_emb_lin[head][j], m->tmp_frag, Ds, m->part2, &z_splits)) != AIC_OK) return rc;
    z = StageSrc{m->part2, z_splits, nullptr, 0};
  }
  // states = proj(prev), then (LayerNorm, GELU, Linear)*
  int s_splits = 1;
  if ((rc = gemm(m->mlp_proj_t[head], first ? m->x0 : m->h_bf16, K0, m->part, &s_splits)) != AIC_OK) return rc;
  for (int j = 0; j < m->st_proj; ++j) {
    if ((rc = stage(StageSrc{m->part, s_splits, nullptr, 0}, none, 0, m->st_proj_ln_w[head][j], m->st_proj_ln_b[head][j], 1, 1,
                    m->tmp_frag, nullptr)) != AIC_OK)
      return rc;
    if ((rc = gemm(m->st_proj_lin[head][j], m->tmp_frag, Ds, m->part, &s_splits)) != AIC_OK) return rc;
  }
  // states.add_(z, alpha); y = ln[0](states); then (GELU, Linear, LayerNorm)*; state = GELU(y)
  const bool plain_ln = m->st_ln == 0;
  if ((rc = stage(StageSrc{m->part, s_splits, nullptr, 0}, z, 1, m->mlp_ln_w[head], m->mlp_ln_b[head], 1, 1,
                  plain_ln ? m->h_bf16 : m->tmp_frag, plain_ln ? m->amax + head : nullptr)) != AIC_OK)
    return rc;
  for (int j = 0; j < m->st_ln; ++j) {
    if ((rc = gemm(m->st_ln_lin[head][j], m->tmp_frag, Ds, m->part, &s_splits)) != AIC_OK) return rc;
    const bool last = j + 1 == m->st_ln;
    if ((rc = stage(StageSrc{m->part, s_splits, nullptr, 0}, none, 0, m->st_ln_ln_w[head][j], m->st_ln_ln_b[head][j], 1, 1,
                    last ? m->h_bf16 : m->tmp_frag, last ? m->amax + head : nullptr)) != AIC_OK)
      return rc;
  }
  return AIC_OK;
}

static int run_head(aic_lstm* m, int head_index, hipStream_t s, int64_t* out_tokens, int out_stride, int out_col,
                    float* out_vals) {
  const aic_lstm_config& c = m->cfg;
  const int mt = m->cur_mt, mpad = mt * 16, B = m->cur_batch;
  const int Ds = c.inner_dim;
  int rc;
  if (m->mlp) {
    AIC_REQUIRE(head_index < m->mlp_heads, "head %d of an MLP speculator with %d heads", head_index, m->mlp_heads);
    if (m->st_emb + m->st_proj + m->st_ln > 0) {
      if ((rc = run_stacked_states(m, head_index, s)) != AIC_OK) return rc;
      return run_mlp_lm_head(m, head_index, s, out_tokens, out_stride, out_col, out_vals);
    }
    // 1. projection (Ds x K), split-K partials in fp32;  2. + embedding, layer norm, gelu
    const bool first = head_index == 0;
    const int K = first ? c.input_hidden_dim : Ds;
    const int steps_total = K / 32;
    int splits = m->gate_splits;
    while (splits > 1 && (steps_total % (splits * kChunkSteps) != 0)) --splits;
    const int rowtiles = Ds / 16;
    dim3 grid((rowtiles + 3) / 4, splits);
    rc = launch_gemm<false, 0>(mt, grid, s, m->mlp_proj_t[head_index], first ? m->x0 : m->h_bf16, rowtiles, steps_total,
                               steps_total / splits, m->part, Ds, nullptr, 1.0f, 0, 0, nullptr, nullptr);
    if (rc != AIC_OK) return rc;
    const uint16_t* emb_src = m->ext_rows ? m->ext_rows : m->mlp_emb[head_index];
    AIC_REQUIRE(emb_src, "head %d has no embedding table and no looked-up rows (aic_mlp_set_embedding_rows)", head_index);
    hipLaunchKernelGGL(mlp_cell_kernel, dim3(mpad), dim3(1024), Ds * sizeof(float), s, m->part, splits, mpad, B, m->tokens,
                       emb_src, 0x7fffffff, m->alpha, m->mlp_ln_w[head_index], m->mlp_ln_b[head_index], Ds,
                       mt, m->h_bf16, m->amax + head_index, m->ext_rows ? 1 : 0);
    if ((rc = launch_status("mlp_cell_kernel")) != AIC_OK) return rc;
    // 3. LM head + fused arg-max
    const bool fp8m = m->mlp_head8_t[head_index] && mpad <= c.head_fp8_max_batch;
    dim3 hgrid(m->head_blocks, 1);
    if (fp8m) {
      hipLaunchKernelGGL(quant_act_kernel, dim3(64), dim3(256), 0, s, m->h_bf16, m->h_fp8, m->amax + head_index,
                         m->x_scale, Ds, mt);
      if ((rc = launch_status("quant_act_kernel")) != AIC_OK) return rc;
      rc = launch_gemm<true, 1>(mt, hgrid, s, m->mlp_head8_t[head_index], m->h_fp8, m->head_rowtiles, Ds / 64, Ds / 64,
                                nullptr, 0, m->x_scale, m->mlp_head8_scale[head_index], c.vocab_size, c.vocab_offset,
                                m->best_val, m->best_idx);
    } else {
      rc = launch_gemm<false, 1>(mt, hgrid, s, m->mlp_head_t[head_index], m->h_bf16, m->head_rowtiles, Ds / 32, Ds / 32,
                                 nullptr, 0, nullptr, 1.0f, c.vocab_size, c.vocab_offset, m->best_val, m->best_idx);
    }
    if (rc != AIC_OK) return rc;
    hipLaunchKernelGGL(argmax_finish_kernel, dim3(mpad), dim3(256), 0, s, m->best_val, m->best_idx, m->head_blocks, mpad,
                       B, m->tokens, out_tokens, out_stride, out_col, out_vals);
    return launch_status("argmax_finish_kernel");
  }
  // 1. gate projection (4Ds x K), split-K partials in fp32
  {
    const bool first = head_index == 0;
    const int K = first ? c.input_hidden_dim : Ds;
    const int steps_total = K / 32;
    int splits = m->gate_splits;
    while (splits > 1 && (steps_total % (splits * kChunkSteps) != 0)) --splits;
    dim3 grid((m->gate_rowtiles + 3) / 4, splits);
    rc = launch_gemm<false, 0>(mt, grid, s, first ? m->proj0_t : m->proj1_t, first ? m->x0 : m->h_bf16,
                               m->gate_rowtiles, steps_total, steps_total / splits, m->part, 4 * Ds, nullptr, 1.0f, 0,
                               0, nullptr, nullptr);
    if (rc != AIC_OK) return rc;
    // 2. cell update: the two-launch form of lstm_cell_kernel (mode 0, launch boundary, mode 1)
    CellArgs ca{m->part, splits, mpad, B, m->tokens, static_cast<const uint16_t*>(m->w.forget_emb), 0x7fffffff, m->alpha, Ds};
    for (int mode = 0; mode < 2; ++mode)
      hipLaunchKernelGGL(lstm_cell_kernel, dim3(mpad, kCellParts), dim3(kCellThreads), 0, s, ca, mode,
                         static_cast<const uint16_t*>(m->w.cell_ln_w), static_cast<const uint16_t*>(m->w.cell_ln_b),
                         static_cast<const uint16_t*>(m->w.state_ln_w), static_cast<const uint16_t*>(m->w.state_ln_b),
                         m->cell, m->ss2_part, PrevArgmax{}, mt, m->h_bf16, m->amax + head_index, CellSync{});
    if ((rc = launch_status("lstm_cell_kernel")) != AIC_OK) return rc;
  }
  // 3. LM head + fused arg-max
  const bool fp8 = m->head8_t && mpad <= c.head_fp8_max_batch;
  dim3 hgrid(m->head_blocks, 1);
  if (fp8) {
    hipLaunchKernelGGL(quant_act_kernel, dim3(64), dim3(256), 0, s, m->h_bf16, m->h_fp8, m->amax + head_index,
                       m->x_scale, Ds, mt);
    if ((rc = launch_status("quant_act_kernel")) != AIC_OK) return rc;
    rc = launch_gemm<true, 1>(mt, hgrid, s, m->head8_t, m->h_fp8, m->head_rowtiles, Ds / 64, Ds / 64, nullptr, 0,
                              m->x_scale, m->head8_scale, c.vocab_size, c.vocab_offset, m->best_val, m->best_idx);
  } else {
    rc = launch_gemm<false, 1>(mt, hgrid, s, m->head_t, m->h_bf16, m->head_rowtiles, Ds / 32, Ds / 32, nullptr, 0,
                               nullptr, 1.0f, c.vocab_size, c.vocab_offset, m->best_val, m->best_idx);
  }
  if (rc != AIC_OK) return rc;
  hipLaunchKernelGGL(argmax_finish_kernel, dim3(mpad), dim3(256), 0, s, m->best_val, m->best_idx, m->head_blocks, mpad,
                     B, m->tokens, out_tokens, out_stride, out_col, out_vals);
  return launch_status("argmax_finish_kernel");
}

extern "C" {

int aic_lstm_padding_size(int size) {  // arctic_speculator.py:39-44
  if (size <= 0) return size;
  int bits = 0;
  for (int v = size - 1; v > 0; v >>= 1) ++bits;
  const int mult = (1 << bits) / 4;
  if (mult < 1) return size;
  return (size + mult - 1) / mult * mult;
}

int aic_quantize_fp8_per_tensor(const void* src_bf16, void* dst_fp8, float* scale_out, int64_t n, void* stream) {
  AIC_REQUIRE(src_bf16 && dst_fp8 && scale_out && n > 0, "bad arguments to aic_quantize_fp8_per_tensor");
  AIC_NEED_DEVICE();
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned int* amax = nullptr;
  AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&amax), 4));
  AIC_HIP_TRY(hipMemsetAsync(amax, 0, 4, s));
  const unsigned g = static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 2048));
  hipLaunchKernelGGL(amax_bf16_kernel, dim3(g), dim3(256), 0, s, static_cast<const uint16_t*>(src_bf16), n, amax);
  hipLaunchKernelGGL(finish_scale_kernel, dim3(1), dim3(1), 0, s, amax, scale_out);
  hipLaunchKernelGGL(quant_rowmajor_fp8_kernel, dim3(g), dim3(256), 0, s, static_cast<const uint16_t*>(src_bf16),
                     static_cast<uint8_t*>(dst_fp8), scale_out, n);
  int rc = launch_status("quantize_fp8_per_tensor");
  AIC_HIP_TRY(hipStreamSynchronize(s));
  AIC_HIP_TRY(hipFree(amax));
  return rc;
}

int aic_lstm_create(const aic_lstm_config* cfg, const aic_lstm_weights* w, aic_lstm** out) {
  AIC_REQUIRE(cfg && w && out, "null argument to aic_lstm_create");
  AIC_REQUIRE(cfg->inner_dim > 0 && cfg->inner_dim % 512 == 0, "inner_dim must be a positive multiple of 512");
  AIC_REQUIRE(cfg->inner_dim <= 4 * kCellMaxIter * 4 * kCellThreads, "inner_dim above %d is not supported by the cell kernel",
              4 * kCellMaxIter * 4 * kCellThreads);
  AIC_REQUIRE(cfg->input_hidden_dim > 0 && cfg->input_hidden_dim % 256 == 0,
              "input_hidden_dim must be a positive multiple of 256");
  AIC_REQUIRE(cfg->vocab_size > 0 && cfg->n_predict > 0 && cfg->max_batch > 0 && cfg->max_batch <= 64,
              "vocab_size / n_predict must be positive and max_batch in 1..64");
  AIC_REQUIRE(w->forget_emb && w->proj0 && w->proj1 && w->cell_ln_w && w->cell_ln_b && w->state_ln_w &&
                  w->state_ln_b && w->head,
              "missing weight pointer");
  AIC_NEED_DEVICE();
  aic_lstm* m = new aic_lstm();
  m->cfg = *cfg;
  m->w = *w;
  if (const char* e = std::getenv("AIC_LSTM_GATE_SPLITS")) {              // A/B switch: split-K of the gate projection (default 2)
    const int v = std::atoi(e);
    if (v >= 1 && v <= 8) m->gate_splits = v;
  }
  const int Ds = cfg->inner_dim, H = cfg->input_hidden_dim, V = cfg->vocab_size;
  const double sw = std::pow(0.5, 0.5 / cfg->n_predict);                 // state_weight (:575)
  const double ew = std::sqrt((1.0 - sw * sw) * (static_cast<double>(Ds) / 2.0));  // emb_weight (:576-577)
  m->alpha = static_cast<float>(ew / sw);
  m->gate_rowtiles = 4 * Ds / 16;
  m->head_blocks = (V + kRowsPerBlock - 1) / kRowsPerBlock;
  m->head_rowtiles = m->head_blocks * 4;
  m->max_mt = pad_mt(cfg->max_batch);
  const int mpad = m->max_mt * 16;
  const int Kmax = std::max(Ds, H);
  hipStream_t s = nullptr;
#define AIC_ALLOC(ptr, bytes) AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&(ptr)), (bytes)))
  AIC_ALLOC(m->proj0_t, static_cast<size_t>(4) * Ds * H * 2);
  AIC_ALLOC(m->proj1_t, static_cast<size_t>(4) * Ds * Ds * 2);
  AIC_ALLOC(m->head_t, static_cast<size_t>(m->head_rowtiles) * 16 * Ds * 2);
  AIC_ALLOC(m->x0, static_cast<size_t>(mpad) * Kmax * 2);
  AIC_ALLOC(m->h_bf16, static_cast<size_t>(mpad) * Kmax * 2);
  AIC_ALLOC(m->h_fp8, static_cast<size_t>(mpad) * Ds);
  AIC_ALLOC(m->cell, static_cast<size_t>(mpad) * Ds * 2);
  AIC_ALLOC(m->part, static_cast<size_t>(m->gate_splits) * mpad * 4 * Ds * 4);
  AIC_ALLOC(m->ss2_part, static_cast<size_t>(64) * kCellParts * sizeof(float));
  AIC_ALLOC(m->sync, static_cast<size_t>(kSyncHeads) * kSyncStride * sizeof(unsigned int));
  AIC_HIP_TRY(hipMemset(m->sync, 0xff, static_cast<size_t>(kSyncHeads) * kSyncStride * sizeof(unsigned int)));
  AIC_ALLOC(m->best_val, static_cast<size_t>(m->head_blocks) * mpad * 4);
  AIC_ALLOC(m->best_idx, static_cast<size_t>(m->head_blocks) * mpad * 4);
  AIC_ALLOC(m->tokens, static_cast<size_t>(mpad) * 4);
  AIC_ALLOC(m->amax, 64 * 4);
  AIC_ALLOC(m->x_scale, 4);
  hipLaunchKernelGGL(repack_bf16_kernel, dim3(2048), dim3(256), 0, s, static_cast<const uint16_t*>(w->proj0),
                     m->proj0_t, 4 * Ds, H, m->gate_rowtiles);
  hipLaunchKernelGGL(repack_bf16_kernel, dim3(2048), dim3(256), 0, s, static_cast<const uint16_t*>(w->proj1),
                     m->proj1_t, 4 * Ds, Ds, m->gate_rowtiles);
  hipLaunchKernelGGL(repack_bf16_kernel, dim3(4096), dim3(256), 0, s, static_cast<const uint16_t*>(w->head), m->head_t,
                     V, Ds, m->head_rowtiles);
  int rc = launch_status("repack_bf16_kernel");
  if (rc != AIC_OK) return rc;
  if (cfg->head_fp8_max_batch > 0) {
    AIC_ALLOC(m->head8_t, static_cast<size_t>(m->head_rowtiles) * 16 * Ds);
    if (w->head_fp8) {
      // caller-provided e4m3 copy is only used for its scale: quantising from the bf16 head with that
      // scale is the same arithmetic as the reference's post-load hook (fp8.py:207-223)
      m->head8_scale = w->head_fp8_scale;
      AIC_HIP_TRY(hipMemcpy(m->x_scale, &m->head8_scale, 4, hipMemcpyHostToDevice));
    } else {
      AIC_HIP_TRY(hipMemset(m->amax, 0, 4));
      hipLaunchKernelGGL(amax_bf16_kernel, dim3(2048), dim3(256), 0, s, static_cast<const uint16_t*>(w->head),
                         static_cast<int64_t>(V) * Ds, m->amax);
      hipLaunchKernelGGL(finish_scale_kernel, dim3(1), dim3(1), 0, s, m->amax, m->x_scale);
      AIC_HIP_TRY(hipMemcpy(&m->head8_scale, m->x_scale, 4, hipMemcpyDeviceToHost));
    }
    hipLaunchKernelGGL(quant_repack_fp8_kernel, dim3(4096), dim3(256), 0, s, static_cast<const uint16_t*>(w->head),
                       m->head8_t, m->x_scale, V, Ds, m->head_rowtiles);
    if ((rc = launch_status("quant_repack_fp8_kernel")) != AIC_OK) return rc;
  }
#undef AIC_ALLOC
  AIC_HIP_TRY(hipDeviceSynchronize());
  *out = m;
  return AIC_OK;
}

// MLP speculator (arctic_speculator.py:102-401) on the same handle type and the same propose / begin / head entry
// points.  Tied stages pass the same pointer for several heads; each distinct matrix is repacked once.
int aic_mlp_create(const aic_lstm_config* cfg, const aic_mlp_weights* w, aic_lstm** out) {
  AIC_REQUIRE(cfg && w && out, "null argument to aic_mlp_create");
  AIC_REQUIRE(cfg->inner_dim > 0 && cfg->inner_dim % 512 == 0, "inner_dim must be a positive multiple of 512");
  AIC_REQUIRE(cfg->input_hidden_dim > 0 && cfg->input_hidden_dim % 256 == 0,
              "input_hidden_dim must be a positive multiple of 256");
  AIC_REQUIRE(cfg->vocab_size > 0 && cfg->n_predict > 0 && cfg->max_batch > 0 && cfg->max_batch <= 64,
              "vocab_size / n_predict must be positive and max_batch in 1..64");
  AIC_REQUIRE(w->num_heads > 0 && w->num_heads <= 8, "an MLP speculator has 1..8 heads");
  for (int i = 0; i < w->num_heads; ++i)
    // emb[i] may be NULL: a vocab-sharded embedding is looked up by the caller (aic_mlp_set_embedding_rows)
    AIC_REQUIRE(w->proj[i] && w->ln_w[i] && w->ln_b[i] && w->head[i], "missing weight pointer of head %d", i);
  AIC_NEED_DEVICE();
  aic_lstm* m = new aic_lstm();
  m->cfg = *cfg;
  m->mlp = true;
  m->mlp_heads = w->num_heads;
  const int Ds = cfg->inner_dim, H = cfg->input_hidden_dim, V = cfg->vocab_size;
  const double sw = std::pow(0.5, 0.5 / cfg->n_predict);                           // state_weight (:221)
  const double ew = std::sqrt((1.0 - sw * sw) * (static_cast<double>(Ds) / 2.0));  // emb_weight (:222-223)
  m->alpha = static_cast<float>(ew / sw);
  m->head_blocks = (V + kRowsPerBlock - 1) / kRowsPerBlock;
  m->head_rowtiles = m->head_blocks * 4;
  m->max_mt = pad_mt(cfg->max_batch);
  const int mpad = m->max_mt * 16;
  const int Kmax = std::max(Ds, H);
  hipStream_t s = nullptr;
#define AIC_ALLOC(ptr, bytes) AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&(ptr)), (bytes)))
  AIC_ALLOC(m->x0, static_cast<size_t>(mpad) * Kmax * 2);
  AIC_ALLOC(m->h_bf16, static_cast<size_t>(mpad) * Kmax * 2);
  AIC_ALLOC(m->h_fp8, static_cast<size_t>(mpad) * Ds);
  AIC_ALLOC(m->cell, static_cast<size_t>(mpad) * Ds * 2);   // written by ln0_kernel, otherwise unused here
  AIC_ALLOC(m->part, static_cast<size_t>(m->gate_splits) * mpad * Ds * 4);
  AIC_ALLOC(m->best_val, static_cast<size_t>(m->head_blocks) * mpad * 4);
  AIC_ALLOC(m->best_idx, static_cast<size_t>(m->head_blocks) * mpad * 4);
  AIC_ALLOC(m->tokens, static_cast<size_t>(mpad) * 4);
  AIC_ALLOC(m->amax, 64 * 4);
  AIC_ALLOC(m->x_scale, 4);
  int rc;
  for (int i = 0; i < w->num_heads; ++i) {
    m->mlp_emb[i] = static_cast<const uint16_t*>(w->emb[i]);
    m->mlp_ln_w[i] = static_cast<const uint16_t*>(w->ln_w[i]);
    m->mlp_ln_b[i] = static_cast<const uint16_t*>(w->ln_b[i]);
    int same = -1;
    for (int j = 1; j < i; ++j)   // head 0's projection has its own input width
      if (w->proj[j] == w->proj[i]) same = j;
    if (same >= 0) {
      m->mlp_proj_t[i] = m->mlp_proj_t[same];
    } else {
      const int K = i == 0 ? H : Ds;
      AIC_ALLOC(m->mlp_proj_t[i], static_cast<size_t>(Ds) * K * 2);
      m->mlp_owned.push_back(m->mlp_proj_t[i]);
      hipLaunchKernelGGL(repack_bf16_kernel, dim3(2048), dim3(256), 0, s, static_cast<const uint16_t*>(w->proj[i]),
                         m->mlp_proj_t[i], Ds, K, Ds / 16);
    }
    same = -1;
    for (int j = 0; j < i; ++j)
      if (w->head[j] == w->head[i]) same = j;
    if (same >= 0) {
      m->mlp_head_t[i] = m->mlp_head_t[same];
      m->mlp_head8_t[i] = m->mlp_head8_t[same];
      m->mlp_head8_scale[i] = m->mlp_head8_scale[same];
      continue;
    }
    AIC_ALLOC(m->mlp_head_t[i], static_cast<size_t>(m->head_rowtiles) * 16 * Ds * 2);
    m->mlp_owned.push_back(m->mlp_head_t[i]);
    hipLaunchKernelGGL(repack_bf16_kernel, dim3(4096), dim3(256), 0, s, static_cast<const uint16_t*>(w->head[i]),
                       m->mlp_head_t[i], V, Ds, m->head_rowtiles);
    if ((rc = launch_status("repack_bf16_kernel")) != AIC_OK) return rc;
    if (cfg->head_fp8_max_batch > 0) {   // per-tensor e4m3 copy of the head, as the reference's qhead (:147-158)
      AIC_ALLOC(m->mlp_head8_t[i], static_cast<size_t>(m->head_rowtiles) * 16 * Ds);
      m->mlp_owned.push_back(m->mlp_head8_t[i]);
      AIC_HIP_TRY(hipMemset(m->amax, 0, 4));
      hipLaunchKernelGGL(amax_bf16_kernel, dim3(2048), dim3(256), 0, s, static_cast<const uint16_t*>(w->head[i]),
                         static_cast<int64_t>(V) * Ds, m->amax);
      hipLaunchKernelGGL(finish_scale_kernel, dim3(1), dim3(1), 0, s, m->amax, m->x_scale);
      AIC_HIP_TRY(hipMemcpy(&m->mlp_head8_scale[i], m->x_scale, 4, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(quant_repack_fp8_kernel, dim3(4096), dim3(256), 0, s, static_cast<const uint16_t*>(w->head[i]),
                         m->mlp_head8_t[i], m->x_scale, V, Ds, m->head_rowtiles);
      if ((rc = launch_status("quant_repack_fp8_kernel")) != AIC_OK) return rc;
      AIC_HIP_TRY(hipDeviceSynchronize());   // x_scale is reused for the next head's scale
    }
  }
#undef AIC_ALLOC
  if ((rc = launch_status("repack_bf16_kernel")) != AIC_OK) return rc;
  AIC_HIP_TRY(hipDeviceSynchronize());
  *out = m;
  return AIC_OK;
}

// aic_mlp_create + the extra stages of a stacked sum_rnn speculator.  Every width is inner_dim (the reference's LayerNorms
// are built with the NEXT entry of the dimension list and applied to the previous stage's output, :488-492, so the lists can
// only hold equal entries).
int aic_mlp_create_stacked(const aic_lstm_config* cfg, const aic_mlp_weights* w, const aic_mlp_stack* st, aic_lstm** out) {
  AIC_REQUIRE(st && st->n_emb >= 0 && st->n_emb <= 3 && st->n_proj >= 0 && st->n_proj <= 3 && st->n_ln >= 0 && st->n_ln <= 3,
              "0..3 extra stages per stack");
  int rc = aic_mlp_create(cfg, w, out);
  if (rc != AIC_OK) return rc;
  aic_lstm* m = *out;
  const int Ds = cfg->inner_dim, mpad = m->max_mt * 16;
  m->st_emb = st->n_emb;
  m->st_proj = st->n_proj;
  m->st_ln = st->n_ln;
  std::vector<std::pair<const void*, uint4*>> seen;     // tied stages pass the same matrix: one fragment-major copy
  auto repack = [&](const void* src, uint4** dst) -> int {
    AIC_REQUIRE(src, "missing Linear weight of a stacked stage");
    for (auto& kv : seen)
      if (kv.first == src) {
        *dst = kv.second;
        return AIC_OK;
      }
    uint4* p = nullptr;
    AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), static_cast<size_t>(Ds) * Ds * 2));
    m->mlp_owned.push_back(p);
    hipLaunchKernelGGL(repack_bf16_kernel, dim3(2048), dim3(256), 0, nullptr, static_cast<const uint16_t*>(src), p, Ds, Ds,
                       Ds / 16);
    seen.emplace_back(src, p);
    *dst = p;
    return launch_status("repack_bf16_kernel");
  };
  for (int i = 0; i < w->num_heads; ++i) {
    for (int j = 0; j < st->n_emb; ++j) {
      AIC_REQUIRE(st->emb_ln_w[i][j] && st->emb_ln_b[i][j], "missing LayerNorm of emb stage %d, head %d", j, i);
      m->st_emb_ln_w[i][j] = static_cast<const uint16_t*>(st->emb_ln_w[i][j]);
      m->st_emb_ln_b[i][j] = static_cast<const uint16_t*>(st->emb_ln_b[i][j]);
      if ((rc = repack(st->emb_lin[i][j], &m->st_emb_lin[i][j])) != AIC_OK) return rc;
    }
    for (int j = 0; j < st->n_proj; ++j) {
      AIC_REQUIRE(st->proj_ln_w[i][j] && st->proj_ln_b[i][j], "missing LayerNorm of proj stage %d, head %d", j, i);
      m->st_proj_ln_w[i][j] = static_cast<const uint16_t*>(st->proj_ln_w[i][j]);
      m->st_proj_ln_b[i][j] = static_cast<const uint16_t*>(st->proj_ln_b[i][j]);
      if ((rc = repack(st->proj_lin[i][j], &m->st_proj_lin[i][j])) != AIC_OK) return rc;
    }
    for (int j = 0; j < st->n_ln; ++j) {
      AIC_REQUIRE(st->ln_ln_w[i][j] && st->ln_ln_b[i][j], "missing LayerNorm of ln stage %d, head %d", j, i);
      m->st_ln_ln_w[i][j] = static_cast<const uint16_t*>(st->ln_ln_w[i][j]);
      m->st_ln_ln_b[i][j] = static_cast<const uint16_t*>(st->ln_ln_b[i][j]);
      if ((rc = repack(st->ln_lin[i][j], &m->st_ln_lin[i][j])) != AIC_OK) return rc;
    }
  }
  AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->part2), static_cast<size_t>(m->gate_splits) * mpad * Ds * 4));
  AIC_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->tmp_frag), static_cast<size_t>(mpad) * Ds * 2));
  AIC_HIP_TRY(hipDeviceSynchronize());
  return AIC_OK;
}

void aic_lstm_destroy(aic_lstm* m) {
  if (!m) return;
  void* bufs[] = {m->proj0_t, m->proj1_t, m->head_t, m->head8_t, m->x0,   m->h_bf16, m->h_fp8,
                  m->cell,    m->part,    m->ss2_part, m->best_val, m->best_idx, m->tokens, m->amax, m->x_scale, m->sync};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (void* b : m->mlp_owned)
    if (b) (void)hipFree(b);
  if (m->part2) (void)hipFree(m->part2);
  if (m->tmp_frag) (void)hipFree(m->tmp_frag);
  delete m;
}

int aic_mlp_set_embedding_rows(aic_lstm* m, const void* rows) {
  AIC_REQUIRE(m && m->mlp, "aic_mlp_set_embedding_rows needs an MLP speculator handle");
  m->ext_rows = static_cast<const uint16_t*>(rows);
  return AIC_OK;
}

static int lstm_begin(aic_lstm* m, const void* hidden, const int32_t* hidden_index, const int32_t* last_tokens, int batch,
                      void* stream) {
  AIC_REQUIRE(m && hidden && batch > 0 && batch <= m->cfg.max_batch, "bad arguments to aic_lstm_begin (batch %d)", batch);
  hipStream_t s = static_cast<hipStream_t>(stream);
  m->cur_batch = batch;
  m->cur_mt = pad_mt(batch);
  hipLaunchKernelGGL(ln0_kernel, dim3(m->cur_mt * 16), dim3(256), 0, s, static_cast<const uint16_t*>(hidden),
                     hidden_index, batch, m->cfg.input_hidden_dim, m->cur_mt, m->cfg.scale_input, m->x0, m->cell,
                     m->cfg.inner_dim, m->amax, 64, last_tokens, m->tokens, m->sync, m->sync ? kSyncHeads * kSyncStride : 0);
  return launch_status("ln0_kernel");
}

int aic_lstm_begin(aic_lstm* m, const void* hidden, const int32_t* hidden_index, int batch, void* stream) {
  return lstm_begin(m, hidden, hidden_index, nullptr, batch, stream);
}

int aic_lstm_head(aic_lstm* m, int head_index, const int32_t* last_tokens, int batch, int64_t* out_tokens,
                  float* out_vals, void* stream) {
  AIC_REQUIRE(m && last_tokens && batch == m->cur_batch && head_index >= 0 && head_index < 64,
              "bad arguments to aic_lstm_head");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(copy_tokens_kernel, dim3(1), dim3(64), 0, s, last_tokens, m->tokens, batch);
  int rc = launch_status("copy_tokens_kernel");
  if (rc != AIC_OK) return rc;
  return run_head(m, head_index, s, out_tokens, 1, 0, out_vals);
}

static std::atomic<int> g_lstm_xq{0};
static int64_t* g_lstm_cell_trace = nullptr;   // aic_debug_lstm_cell_trace
static std::atomic<int> g_lstm_cell_launches{1};   // aic_debug_lstm_cell_launches: 1 = one cell launch per head (default), 3 = three

// The whole k-head draft of the LSTM speculator in 3 k + 3 launches (12 at k = 3; the head-by-head form takes 5-6 per
// head + 1): the gate projection of head h + 1 shares a launch with the LM head of head h (skinny_pair_kernel), the arg-max
// over the LM head's partials is finished inside the next cell launch (or once at the end), and the fp8 head quantises its
// activations on the way into LDS.  Same arithmetic as run_head() operation by operation: bit-identical tokens.
static int propose_fused(aic_lstm* m, int k, int64_t* out_tokens, float* out_vals, hipStream_t s) {
  const aic_lstm_config& c = m->cfg;
  const int mt = m->cur_mt, mpad = mt * 16, B = m->cur_batch, Ds = c.inner_dim;
  const bool fp8 = m->head8_t && mpad <= c.head_fp8_max_batch;
  auto gate_args = [&](int head, int* splits_out) {
    const bool first = head == 0;
    const int K = first ? c.input_hidden_dim : Ds;
    const int steps_total = K / 32;
    int splits = m->gate_splits;
    while (splits > 1 && (steps_total % (splits * kChunkSteps) != 0)) --splits;
    *splits_out = splits;
    return GemmArgs{first ? m->proj0_t : m->proj1_t, first ? m->x0 : m->h_bf16, m->gate_rowtiles, steps_total,
                    steps_total / splits, m->part, 4 * Ds, nullptr, 1.0f, 0, 0, nullptr, nullptr, nullptr};
  };
  // (XQ — the fp8 head quantising its activations on the way into LDS — is built and bit-identical, but measured slower
  // than the 5 us quant_act launch it removes: head alone 98.1 against 89.5 us at 32 rows, rocprofv3
  // profiles/r03_lstm_kernel_stats.csv; the 2004 workgroups each redo the conversion of the whole activation matrix.)
  const bool xq = fp8 && g_lstm_xq.load() != 0;
  auto head_args = [&](int head) {
    if (fp8 && xq)
      return GemmArgs{m->head8_t, m->h_bf16, m->head_rowtiles, Ds / 64, Ds / 64, nullptr, 0, nullptr, m->head8_scale,
                      c.vocab_size, c.vocab_offset, m->best_val, m->best_idx, m->amax + head};
    if (fp8)
      return GemmArgs{m->head8_t, m->h_fp8, m->head_rowtiles, Ds / 64, Ds / 64, nullptr, 0, m->x_scale, m->head8_scale,
                      c.vocab_size, c.vocab_offset, m->best_val, m->best_idx, nullptr};
    return GemmArgs{m->head_t, m->h_bf16, m->head_rowtiles, Ds / 32, Ds / 32, nullptr, 0, nullptr, 1.0f, c.vocab_size,
                    c.vocab_offset, m->best_val, m->best_idx, nullptr};
  };
  int splits = 1, rc;
  {
    const GemmArgs G = gate_args(0, &splits);
    dim3 grid((m->gate_rowtiles + 3) / 4, splits);
    rc = launch_gemm<false, 0>(mt, grid, s, G.W, G.X, G.n_rowtiles, G.steps_total, G.steps_per_split, G.part, G.n_cols_out,
                               nullptr, 1.0f, 0, 0, nullptr, nullptr);
    if (rc != AIC_OK) return rc;
  }
  for (int h = 0; h < k; ++h) {
    CellArgs ca{m->part, splits, mpad, B, m->tokens, static_cast<const uint16_t*>(m->w.forget_emb), 0x7fffffff, m->alpha, Ds};
    PrevArgmax pa{};
    if (h > 0) pa = PrevArgmax{m->best_val, m->best_idx, m->head_blocks, m->tokens, out_tokens, out_vals, k, h - 1};
    // ONE cell launch per head (mode 2: row-sum and |max| steps inside the launch, the fp8 head's activations quantised from
    // registers): 4 x mpad workgroups of 256 threads, all resident at once.  Drafts of more than kSyncHeads heads, and
    // aic_debug_lstm_cell_launches(3), keep the three-launch form (mode 0, mode 1, quant_act_kernel).
    const bool one_launch = g_lstm_cell_launches.load() == 1 && h < kSyncHeads && m->sync != nullptr;
    const uint16_t *clw = static_cast<const uint16_t*>(m->w.cell_ln_w), *clb = static_cast<const uint16_t*>(m->w.cell_ln_b);
    const uint16_t *slw = static_cast<const uint16_t*>(m->w.state_ln_w), *slb = static_cast<const uint16_t*>(m->w.state_ln_b);
    if (one_launch) {
      unsigned int* sw = m->sync + static_cast<size_t>(h) * kSyncStride;
      CellSync sy{sw, sw + kSyncRowWords, (fp8 && !xq) ? m->h_fp8 : nullptr, m->x_scale,
                  g_lstm_cell_trace ? g_lstm_cell_trace + static_cast<int64_t>(h) * 64 * kCellParts * 12 : nullptr};
      hipLaunchKernelGGL(lstm_cell_kernel, dim3(mpad, kCellParts), dim3(kCellThreads), 0, s, ca, 2, clw, clb, slw, slb, m->cell,
                         m->ss2_part, pa, mt, m->h_bf16, m->amax + h, sy);
      if ((rc = launch_status("lstm_cell_kernel")) != AIC_OK) return rc;
    } else {
      hipLaunchKernelGGL(lstm_cell_kernel, dim3(mpad, kCellParts), dim3(kCellThreads), 0, s, ca, 0, clw, clb, slw, slb, m->cell,
                         m->ss2_part, pa, mt, m->h_bf16, m->amax + h, CellSync{});
      hipLaunchKernelGGL(lstm_cell_kernel, dim3(mpad, kCellParts), dim3(kCellThreads), 0, s, ca, 1, clw, clb, slw, slb, m->cell,
                         m->ss2_part, PrevArgmax{}, mt, m->h_bf16, m->amax + h, CellSync{});
      if ((rc = launch_status("lstm_cell_kernel")) != AIC_OK) return rc;
      if (fp8 && !xq) {
        hipLaunchKernelGGL(quant_act_kernel, dim3(64), dim3(256), 0, s, m->h_bf16, m->h_fp8, m->amax + h, m->x_scale, Ds, mt);
        if ((rc = launch_status("quant_act_kernel")) != AIC_OK) return rc;
      }
    }
    const GemmArgs Hd = head_args(h);
    if (h + 1 < k) {
      const GemmArgs G = gate_args(h + 1, &splits);
      const int gate_bx = (m->gate_rowtiles + 3) / 4;
      const dim3 grid(static_cast<unsigned>(gate_bx * splits + m->head_blocks));
#define AIC_PAIR(FP8H_, MT_, XQ_) \
  hipLaunchKernelGGL((skinny_pair_kernel<FP8H_, MT_, XQ_>), grid, dim3(256), 0, s, G, Hd, gate_bx, splits)
      if (fp8 && xq) {
        if (mt == 1) AIC_PAIR(true, 1, true); else AIC_PAIR(true, 2, true);
      } else if (fp8) {
        if (mt == 1) AIC_PAIR(true, 1, false); else AIC_PAIR(true, 2, false);
      } else {
        if (mt == 1) AIC_PAIR(false, 1, false); else if (mt == 2) AIC_PAIR(false, 2, false); else AIC_PAIR(false, 4, false);
      }
#undef AIC_PAIR
      if ((rc = launch_status("skinny_pair_kernel")) != AIC_OK) return rc;
    } else if (fp8 && xq) {
      if (mt == 1)
        hipLaunchKernelGGL((skinny_head_xq_kernel<1>), dim3(m->head_blocks), dim3(256), 0, s, Hd);
      else
        hipLaunchKernelGGL((skinny_head_xq_kernel<2>), dim3(m->head_blocks), dim3(256), 0, s, Hd);
      if ((rc = launch_status("skinny_head_xq_kernel")) != AIC_OK) return rc;
    } else if (fp8) {
      rc = launch_gemm<true, 1>(mt, dim3(m->head_blocks, 1), s, Hd.W, Hd.X, Hd.n_rowtiles, Hd.steps_total,
                                Hd.steps_per_split, nullptr, 0, m->x_scale, m->head8_scale, Hd.n_valid_rows, Hd.row_offset,
                                Hd.best_val, Hd.best_idx);
      if (rc != AIC_OK) return rc;
    } else {
      rc = launch_gemm<false, 1>(mt, dim3(m->head_blocks, 1), s, Hd.W, Hd.X, Hd.n_rowtiles, Hd.steps_total,
                                 Hd.steps_per_split, nullptr, 0, nullptr, 1.0f, Hd.n_valid_rows, Hd.row_offset,
                                 Hd.best_val, Hd.best_idx);
      if (rc != AIC_OK) return rc;
    }
  }
  hipLaunchKernelGGL(argmax_finish_kernel, dim3(mpad), dim3(256), 0, s, m->best_val, m->best_idx, m->head_blocks, mpad, B,
                     m->tokens, out_tokens, k, k - 1, out_vals);
  return launch_status("argmax_finish_kernel");
}

static std::atomic<int> g_lstm_fused{1};
// debug / A-B aid: launches per cell in the fused draft schedule — 1 (default: lstm_cell_kernel mode 2) or 3 (mode 0,
// mode 1, quant_act_kernel); bit-identical results
// debug: the next fused drafts record, per head and workgroup (row, part), 8 timestamps (100 MHz ticks) at the phase
// boundaries of lstm_cell_kernel into buf[head][64][4][12] (device memory, int64); nullptr switches it off
int aic_debug_lstm_cell_trace(int64_t* buf) {
  g_lstm_cell_trace = buf;
  return AIC_OK;
}
int aic_debug_lstm_cell_launches(int n) {
  g_lstm_cell_launches.store(n == 3 ? 3 : 1);
  return AIC_OK;
}
int aic_debug_lstm_fused(int on) {     // 0 head by head, 1 fused (default), 2 fused + on-the-fly fp8 activation quantisation
  g_lstm_fused.store(on);
  g_lstm_xq.store(on == 2 ? 1 : 0);
  return AIC_OK;
}

int aic_lstm_propose(aic_lstm* m, const void* hidden, const int32_t* hidden_index, const int32_t* last_tokens,
                     int batch, int num_predict_tokens, int64_t* out_tokens, float* out_vals, void* stream) {
  AIC_REQUIRE(m && hidden && last_tokens && out_tokens, "null argument to aic_lstm_propose");
  AIC_REQUIRE(num_predict_tokens > 0 && num_predict_tokens <= 64, "num_predict_tokens out of range");
  int rc = lstm_begin(m, hidden, hidden_index, last_tokens, batch, stream);
  if (rc != AIC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!m->mlp && g_lstm_fused.load() != 0) return propose_fused(m, num_predict_tokens, out_tokens, out_vals, s);
  for (int h = 0; h < num_predict_tokens; ++h) {
    rc = run_head(m, h, s, out_tokens, num_predict_tokens, h, out_vals);
    if (rc != AIC_OK) return rc;
  }
  return AIC_OK;
}

}  // extern "C"
