// A16 — multi-layer paged-KV write in one launch.
//
// Replaces torch.ops.arctic_inference.reshape_and_cache_flash_bulk
// (/root/reference/csrc/custom_ops/kernels.cu:12-155, torch_bindings.cpp:5-18):
//   cache[layer][slot / bs][slot % bs][h][d] = cvt(src[token][layer*H*D + h*D + d])   for K and V,
// tokens with slot < 0 skipped, fp8 caches get x / scale with saturate-to-finite conversion
// (quant_utils.cuh:455-489).
//
// CDNA4 mapping: the op is a pure HBM row scatter — each (token, layer, K|V) row of H*D elements is
// contiguous in the source and in its cache page — so rows are moved with 16-byte-per-lane vector
// accesses, several rows per 256-thread workgroup, grid sized to the row count (>> 256 CUs for any
// real batch).  The per-layer pointer tables ride in the kernel arguments (<= 32 layers per
// launch), so a call makes no host->device copy at all; the reference makes four per call.
#include <hip/hip_runtime.h>

#include "aic_common.h"

namespace aic {

constexpr int kMaxLayersPerLaunch = 32;

struct KvTables {
  void* kc[kMaxLayersPerLaunch];
  void* vc[kMaxLayersPerLaunch];
  const float* ks[kMaxLayersPerLaunch];
  const float* vs[kMaxLayersPerLaunch];
};

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) { return bf16_to_f32(v); }  // bf16 bits
struct half_bits { uint16_t b; };
template <>
__device__ __forceinline__ float to_f32<half_bits>(half_bits v) {
  return f16_to_f32(v.b);
}

// OCP fp8 with saturation to the largest finite value; NaN stays NaN (CUDA __NV_SATFINITE).
template <bool E5M2>
__device__ __forceinline__ uint8_t f32_to_fp8_sat(float x) {
  if (x != x) return E5M2 ? 0x7f : 0x7f;
  const float lim = E5M2 ? 57344.0f : 448.0f;
  x = __builtin_amdgcn_fmed3f(x, lim, -lim);
  int packed;
  if (E5M2) {
    packed = __builtin_amdgcn_cvt_pk_bf8_f32(x, x, 0, false);
  } else {
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false);
  }
  return static_cast<uint8_t>(packed & 0xff);
}

// KIND: 0 = same-type copy, 1 = e4m3, 2 = e5m2.  VEC = source elements per lane per step.
template <typename SrcT, int KIND, int VEC>
__global__ void __launch_bounds__(256)
kv_bulk_write_kernel(const SrcT* __restrict__ keys, const SrcT* __restrict__ values, KvTables tab,
                     const int64_t* __restrict__ slot_mapping, int num_tokens, int num_layers, int layer0, int n,
                     int block_size, int64_t block_stride, int64_t key_stride, int64_t value_stride,
                     int lanes_per_row, int rows_per_block) {
  const int row_in_block = threadIdx.x / lanes_per_row;
  const int lane = threadIdx.x - row_in_block * lanes_per_row;
  if (row_in_block >= rows_per_block) return;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * rows_per_block + row_in_block;
  const int64_t total_rows = static_cast<int64_t>(num_tokens) * num_layers * 2;
  if (row >= total_rows) return;
  // row -> (token, layer, K|V); consecutive rows walk the layers of one token (contiguous source)
  const int64_t token = row / (2 * num_layers);
  const int rem = static_cast<int>(row - token * 2 * num_layers);
  const int is_v = rem / num_layers;
  const int layer = rem - is_v * num_layers;

  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padded token
  const int64_t block_idx = slot / block_size;
  const int64_t block_off = slot - block_idx * block_size;

  const SrcT* src = (is_v ? values + token * value_stride : keys + token * key_stride) +
                    static_cast<int64_t>(layer0 + layer) * n;
  const int64_t dst_off = block_idx * block_stride + block_off * n;

  if (KIND == 0) {
    SrcT* dst = static_cast<SrcT*>(is_v ? tab.vc[layer] : tab.kc[layer]) + dst_off;
    for (int i = lane * VEC; i < n; i += lanes_per_row * VEC) {
      if (VEC * sizeof(SrcT) == 16) {
        *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[i + e] = src[i + e];
      }
    }
  } else {
    uint8_t* dst = static_cast<uint8_t*>(is_v ? tab.vc[layer] : tab.kc[layer]) + dst_off;
    const float scale = *(is_v ? tab.vs[layer] : tab.ks[layer]);
    for (int i = lane * VEC; i < n; i += lanes_per_row * VEC) {
      SrcT in[VEC];
      if (VEC * sizeof(SrcT) == 16) {
        *reinterpret_cast<uint4*>(in) = *reinterpret_cast<const uint4*>(src + i);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) in[e] = src[i + e];
      }
      uint8_t q[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) q[e] = f32_to_fp8_sat<KIND == 2>(__fdiv_rn(to_f32<SrcT>(in[e]), scale));
      if (VEC == 8) {
        *reinterpret_cast<uint2*>(dst + i) = *reinterpret_cast<uint2*>(q);
      } else if (VEC == 4) {
        *reinterpret_cast<uint32_t*>(dst + i) = *reinterpret_cast<uint32_t*>(q);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[i + e] = q[e];
      }
    }
  }
}

template <typename SrcT, int KIND>
static int launch_kv(const void* keys, const void* values, const KvTables& tab, const int64_t* slots, int T, int L,
                     int layer0, int n, int block_size, int64_t block_stride, int64_t ks, int64_t vs, bool vec_ok,
                     hipStream_t stream) {
  constexpr int kVec = 16 / sizeof(SrcT);
  const int64_t rows = static_cast<int64_t>(T) * L * 2;
  const int vec = vec_ok ? kVec : 1;
  int lanes = (n + vec - 1) / vec;
  if (lanes > 256) lanes = 256;
  // round lanes per row up to a power of two so rows do not straddle wavefronts unevenly
  int p = 1;
  while (p < lanes) p <<= 1;
  lanes = p;
  const int rpb = 256 / lanes;
  const unsigned grid = static_cast<unsigned>((rows + rpb - 1) / rpb);
  if (vec_ok) {
    hipLaunchKernelGGL((kv_bulk_write_kernel<SrcT, KIND, kVec>), dim3(grid), dim3(256), 0, stream,
                       static_cast<const SrcT*>(keys), static_cast<const SrcT*>(values), tab, slots, T, L, layer0, n,
                       block_size, block_stride, ks, vs, lanes, rpb);
  } else {
    hipLaunchKernelGGL((kv_bulk_write_kernel<SrcT, KIND, 1>), dim3(grid), dim3(256), 0, stream,
                       static_cast<const SrcT*>(keys), static_cast<const SrcT*>(values), tab, slots, T, L, layer0, n,
                       block_size, block_stride, ks, vs, lanes, rpb);
  }
  return launch_status("kv_bulk_write_kernel");
}

}  // namespace aic

using namespace aic;

extern "C" int aic_reshape_and_cache_flash_bulk(const void* keys, const void* values, void* const* key_cache_ptrs,
                                                void* const* value_cache_ptrs, const int64_t* slot_mapping,
                                                int num_tokens, int num_layers, int num_heads, int head_size,
                                                int block_size, int64_t block_stride, int64_t key_stride,
                                                int64_t value_stride, int src_dtype, int kv_dtype,
                                                const float* const* k_scale_ptrs, const float* const* v_scale_ptrs,
                                                void* stream) {
  if (num_layers == 0 || num_tokens == 0) return AIC_OK;  // kernels.cu:99-101
  AIC_REQUIRE(keys && values && key_cache_ptrs && value_cache_ptrs && slot_mapping, "null pointer argument");
  AIC_REQUIRE(num_layers > 0 && num_tokens > 0 && num_heads > 0 && head_size > 0 && block_size > 0,
              "non-positive size argument");
  AIC_REQUIRE(src_dtype == AIC_DT_F32 || src_dtype == AIC_DT_F16 || src_dtype == AIC_DT_BF16,
              "unsupported source dtype %d", src_dtype);
  const bool fp8 = kv_dtype == AIC_DT_FP8_E4M3 || kv_dtype == AIC_DT_FP8_E5M2;
  AIC_REQUIRE(fp8 || kv_dtype == src_dtype, "kv cache dtype %d does not match source dtype %d (\"auto\")", kv_dtype,
              src_dtype);
  AIC_REQUIRE(!fp8 || (k_scale_ptrs && v_scale_ptrs), "fp8 kv cache needs k/v scale tables");
  AIC_NEED_DEVICE();

  const int n = num_heads * head_size;
  const size_t es = src_dtype == AIC_DT_F32 ? 4 : 2;
  const size_t ds = fp8 ? 1 : es;
  const int vec = static_cast<int>(16 / es);
  bool vec_ok = (n % vec == 0) && ((key_stride * es) % 16 == 0) && ((value_stride * es) % 16 == 0) &&
                ((block_stride * ds) % 16 == 0) && ((static_cast<size_t>(n) * ds) % (fp8 ? static_cast<size_t>(vec) : 16) == 0) &&
                (reinterpret_cast<uintptr_t>(keys) % 16 == 0) && (reinterpret_cast<uintptr_t>(values) % 16 == 0);
  for (int l = 0; l < num_layers && vec_ok; ++l)
    vec_ok = (reinterpret_cast<uintptr_t>(key_cache_ptrs[l]) % 16 == 0) &&
             (reinterpret_cast<uintptr_t>(value_cache_ptrs[l]) % 16 == 0);

  for (int l0 = 0; l0 < num_layers; l0 += kMaxLayersPerLaunch) {
    const int L = std::min(kMaxLayersPerLaunch, num_layers - l0);
    KvTables tab;
    for (int l = 0; l < L; ++l) {
      AIC_REQUIRE(key_cache_ptrs[l0 + l] && value_cache_ptrs[l0 + l], "null cache pointer for layer %d", l0 + l);
      tab.kc[l] = key_cache_ptrs[l0 + l];
      tab.vc[l] = value_cache_ptrs[l0 + l];
      tab.ks[l] = fp8 ? k_scale_ptrs[l0 + l] : nullptr;
      tab.vs[l] = fp8 ? v_scale_ptrs[l0 + l] : nullptr;
      AIC_REQUIRE(!fp8 || (tab.ks[l] && tab.vs[l]), "null scale pointer for layer %d", l0 + l);
    }
    for (int l = L; l < kMaxLayersPerLaunch; ++l) {
      tab.kc[l] = tab.vc[l] = nullptr;
      tab.ks[l] = tab.vs[l] = nullptr;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = AIC_ERR_UNSUPPORTED;
#define AIC_KV_DISPATCH(SrcT)                                                                                       \
  if (!fp8)                                                                                                         \
    rc = launch_kv<SrcT, 0>(keys, values, tab, slot_mapping, num_tokens, L, l0, n, block_size, block_stride,        \
                            key_stride, value_stride, vec_ok, s);                                                   \
  else if (kv_dtype == AIC_DT_FP8_E4M3)                                                                             \
    rc = launch_kv<SrcT, 1>(keys, values, tab, slot_mapping, num_tokens, L, l0, n, block_size, block_stride,        \
                            key_stride, value_stride, vec_ok, s);                                                   \
  else                                                                                                              \
    rc = launch_kv<SrcT, 2>(keys, values, tab, slot_mapping, num_tokens, L, l0, n, block_size, block_stride,        \
                            key_stride, value_stride, vec_ok, s);
    if (src_dtype == AIC_DT_F32) {
      AIC_KV_DISPATCH(float)
    } else if (src_dtype == AIC_DT_BF16) {
      AIC_KV_DISPATCH(uint16_t)
    } else {
      AIC_KV_DISPATCH(half_bits)
    }
#undef AIC_KV_DISPATCH
    if (rc != AIC_OK) return rc;
  }
  return AIC_OK;
}
