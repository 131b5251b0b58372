// A12 — Ulysses sequence<->head repartition copies around the two all-to-alls.
//
// Reference: UlyssesAttentionPatch.forward (/root/reference/arctic_inference/vllm/ulysses.py:493-507
// pack + split, :513-517 unpack) builds these layouts with torch.cat / transpose / reshape, i.e. three
// to four elementwise copy kernels per attention layer.  Here each direction is one fused copy that
// writes straight into the all-to-all send layout / reads straight out of the receive layout.
//
//   pack  : q [n][SP*qw], k [n][SP*kw], v [n][SP*kw]  ->  send [SP][n][qw + 2kw]
//           (chunk j of the head axis goes to rank j; GQA groups stay aligned)
//   split : recv [rows][qw + 2kw] -> q_ [rows][qw], k_ [rows][kw], v_ [rows][kw]
//   unpack: recv [SP][n][w] -> out [n][SP*w]
//
// Pure HBM copies of 2-byte elements: 16 bytes per lane, destination-major indexing so stores are
// fully coalesced and loads are >= 256-byte contiguous segments.
#include <hip/hip_runtime.h>

#include "aic_common.h"

namespace aic {

// widths in 16-byte chunks (8 elements)
__global__ void __launch_bounds__(256)
ulysses_pack_kernel(const uint4* __restrict__ q, const uint4* __restrict__ k, const uint4* __restrict__ v,
                    int64_t q_stride, int64_t k_stride, int64_t v_stride, uint4* __restrict__ send, int n, int sp,
                    int qw, int kw) {
  const int W = qw + 2 * kw;
  const int64_t total = static_cast<int64_t>(sp) * n * W;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t row = t / W;  // = j * n + tok
    const int c = static_cast<int>(t - row * W);
    const int j = static_cast<int>(row / n);
    const int64_t tok = row - static_cast<int64_t>(j) * n;
    uint4 val;
    if (c < qw) {
      val = q[tok * q_stride + static_cast<int64_t>(j) * qw + c];
    } else if (c < qw + kw) {
      val = k[tok * k_stride + static_cast<int64_t>(j) * kw + (c - qw)];
    } else {
      val = v[tok * v_stride + static_cast<int64_t>(j) * kw + (c - qw - kw)];
    }
    send[t] = val;
  }
}

__global__ void __launch_bounds__(256)
ulysses_split_kernel(const uint4* __restrict__ recv, uint4* __restrict__ q, uint4* __restrict__ k,
                     uint4* __restrict__ v, int64_t rows, int qw, int kw) {
  const int W = qw + 2 * kw;
  const int64_t total = rows * W;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t row = t / W;
    const int c = static_cast<int>(t - row * W);
    const uint4 val = recv[t];
    if (c < qw) {
      q[row * qw + c] = val;
    } else if (c < qw + kw) {
      k[row * kw + (c - qw)] = val;
    } else {
      v[row * kw + (c - qw - kw)] = val;
    }
  }
}

__global__ void __launch_bounds__(256)
ulysses_unpack_kernel(const uint4* __restrict__ recv, uint4* __restrict__ out, int n, int sp, int w) {
  const int64_t total = static_cast<int64_t>(n) * sp * w;
  const int W = sp * w;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t tok = t / W;
    const int c = static_cast<int>(t - tok * W);
    const int j = c / w;
    const int e = c - j * w;
    out[t] = recv[(static_cast<int64_t>(j) * n + tok) * w + e];
  }
}

static unsigned grid_for(int64_t chunks) {
  int64_t g = (chunks + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return static_cast<unsigned>(g);
}

}  // namespace aic

using namespace aic;

extern "C" {

int aic_ulysses_pack_qkv(const void* q, const void* k, const void* v, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, void* send, int n_local, int sp, int q_width, int kv_width, void* stream) {
  if (n_local == 0) return AIC_OK;
  AIC_REQUIRE(q && k && v && send && n_local > 0 && sp > 0 && q_width > 0 && kv_width > 0, "bad arguments");
  AIC_REQUIRE(q_width % 8 == 0 && kv_width % 8 == 0 && q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0,
              "widths and strides must be multiples of 8 elements (16 bytes)");
  AIC_NEED_DEVICE();
  const int qw = q_width / 8, kw = kv_width / 8;
  const int64_t chunks = static_cast<int64_t>(sp) * n_local * (qw + 2 * kw);
  hipLaunchKernelGGL(ulysses_pack_kernel, dim3(grid_for(chunks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(q), static_cast<const uint4*>(k), static_cast<const uint4*>(v),
                     q_stride / 8, k_stride / 8, v_stride / 8, static_cast<uint4*>(send), n_local, sp, qw, kw);
  return launch_status("ulysses_pack_kernel");
}

int aic_ulysses_split_qkv(const void* recv, void* q, void* k, void* v, int64_t rows, int q_width, int kv_width,
                          void* stream) {
  if (rows == 0) return AIC_OK;
  AIC_REQUIRE(recv && q && k && v && rows > 0 && q_width > 0 && kv_width > 0, "bad arguments");
  AIC_REQUIRE(q_width % 8 == 0 && kv_width % 8 == 0, "widths must be multiples of 8 elements (16 bytes)");
  AIC_NEED_DEVICE();
  const int qw = q_width / 8, kw = kv_width / 8;
  hipLaunchKernelGGL(ulysses_split_kernel, dim3(grid_for(rows * (qw + 2 * kw))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint4*>(recv), static_cast<uint4*>(q),
                     static_cast<uint4*>(k), static_cast<uint4*>(v), rows, qw, kw);
  return launch_status("ulysses_split_kernel");
}

int aic_ulysses_unpack_out(const void* recv, void* out, int n_local, int sp, int width, void* stream) {
  if (n_local == 0) return AIC_OK;
  AIC_REQUIRE(recv && out && n_local > 0 && sp > 0 && width > 0, "bad arguments");
  AIC_REQUIRE(width % 8 == 0, "width must be a multiple of 8 elements (16 bytes)");
  AIC_NEED_DEVICE();
  const int w = width / 8;
  hipLaunchKernelGGL(ulysses_unpack_kernel, dim3(grid_for(static_cast<int64_t>(n_local) * sp * w)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint4*>(recv), static_cast<uint4*>(out),
                     n_local, sp, w);
  return launch_status("ulysses_unpack_kernel");
}

}  // extern "C"
