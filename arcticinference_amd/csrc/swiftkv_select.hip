// SwiftKV token selection — the gather half of LlamaSwiftKVModel.swiftkv_select
// (/root/reference/arctic_inference/vllm/swiftkv/llama_swiftkv.py:573-685, index_fn :665-675).
//
// After the prefill half of a SwiftKV model has produced hidden states, residuals and the K/V projections of
// every remaining layer for ALL tokens of the step, only the tokens that are sampled (logits_indices) continue
// through the decode half.  The reference selects the five tensors with five index_select launches; here it is
// ONE launch: every (selected row, tensor) pair is a contiguous row copy with 16-byte lanes.  The destination rows
// are the decode runner's persistent graph buffers, so the selection is graph-capture safe.
#include <hip/hip_runtime.h>

#include "aic_common.h"

namespace aic {

constexpr int kMaxGatherTensors = 8;

struct GatherTable {
  const char* src[kMaxGatherTensors];
  char* dst[kMaxGatherTensors];
  int64_t src_stride[kMaxGatherTensors];  // bytes between rows
  int64_t dst_stride[kMaxGatherTensors];
  int32_t row_bytes[kMaxGatherTensors];
};

// grid (ceil(n_sel / rows_per_block), n_tensors); a wave copies one row at a time, 16 B per lane per step
__global__ void __launch_bounds__(256) row_gather_kernel(GatherTable tab, const int64_t* __restrict__ index, int n_sel,
                                                         int n_src_rows) {
  const int t = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n_sel) return;
  const int64_t src_row = index[row];
  if (src_row < 0 || src_row >= n_src_rows) return;  // checked on the host for host-known indices; never fault here
  const char* s = tab.src[t] + src_row * tab.src_stride[t];
  char* d = tab.dst[t] + static_cast<int64_t>(row) * tab.dst_stride[t];
  const int nbytes = tab.row_bytes[t];
  const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | static_cast<uintptr_t>(nbytes)) & 15) == 0;
  if (vec) {
    for (int off = lane * 16; off < nbytes; off += 64 * 16)
      *reinterpret_cast<uint4*>(d + off) = *reinterpret_cast<const uint4*>(s + off);
  } else {
    for (int off = lane; off < nbytes; off += 64) d[off] = s[off];
  }
}

}  // namespace aic

using namespace aic;

extern "C" int aic_row_gather(int n_tensors, const void* const* src, void* const* dst, const int64_t* src_stride_bytes,
                              const int64_t* dst_stride_bytes, const int32_t* row_bytes, const int64_t* index, int n_sel,
                              int n_src_rows, void* stream) {
  if (n_sel == 0 || n_tensors == 0) return AIC_OK;
  AIC_REQUIRE(n_tensors > 0 && n_tensors <= kMaxGatherTensors, "aic_row_gather takes 1..%d tensors", kMaxGatherTensors);
  AIC_REQUIRE(src && dst && src_stride_bytes && dst_stride_bytes && row_bytes && index && n_sel > 0 && n_src_rows > 0,
              "bad arguments to aic_row_gather");
  GatherTable tab;
  for (int t = 0; t < n_tensors; ++t) {
    AIC_REQUIRE(src[t] && dst[t] && row_bytes[t] > 0 && src_stride_bytes[t] >= row_bytes[t] && dst_stride_bytes[t] >= row_bytes[t],
                "tensor %d: null pointer or a row stride shorter than the row", t);
    tab.src[t] = static_cast<const char*>(src[t]);
    tab.dst[t] = static_cast<char*>(dst[t]);
    tab.src_stride[t] = src_stride_bytes[t];
    tab.dst_stride[t] = dst_stride_bytes[t];
    tab.row_bytes[t] = row_bytes[t];
  }
  AIC_NEED_DEVICE();
  dim3 grid(static_cast<unsigned>((n_sel + 3) / 4), static_cast<unsigned>(n_tensors));
  hipLaunchKernelGGL(row_gather_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), tab, index, n_sel, n_src_rows);
  return launch_status("row_gather_kernel");
}
