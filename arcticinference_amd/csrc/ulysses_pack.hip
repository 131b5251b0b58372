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
                    int qw, int kw, int vw) {
  const int W = qw + kw + vw;   // a width of 0 drops that tensor (the KV-replicated variant packs q alone and k|v alone)
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
      val = v[tok * v_stride + static_cast<int64_t>(j) * vw + (c - qw - kw)];
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

// KV-replicated variant (ulysses.py:486-490): the all-gathered K|V rows arrive in SP_AG-major chunk order; chunk
// order[c] holds the tokens of SP rank c.  One pass puts the chunks back in rank order and splits K from V.
struct ChunkOrder {
  int32_t src[64];
};
__global__ void __launch_bounds__(256)
ulysses_reorder_split_kernel(const uint4* __restrict__ gathered, uint4* __restrict__ k, uint4* __restrict__ v, int n,
                             int sp, int kw, ChunkOrder order) {
  const int W = 2 * kw;
  const int64_t total = static_cast<int64_t>(sp) * n * W;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t row = t / W;  // destination row = c * n + r
    const int e = static_cast<int>(t - row * W);
    const int c = static_cast<int>(row / n);
    const int64_t r = row - static_cast<int64_t>(c) * n;
    const uint4 val = gathered[(static_cast<int64_t>(order.src[c]) * n + r) * W + e];
    if (e < kw) {
      k[row * kw + e] = val;
    } else {
      v[row * kw + (e - kw)] = val;
    }
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
                     q_stride / 8, k_stride / 8, v_stride / 8, static_cast<uint4*>(send), n_local, sp, qw, kw, kw);
  return launch_status("ulysses_pack_kernel");
}

int aic_ulysses_pack_pair(const void* a, const void* b, int64_t a_stride, int64_t b_stride, void* send, int n_local,
                          int parts, int a_width, int b_width, void* stream) {
  if (n_local == 0) return AIC_OK;
  AIC_REQUIRE(a && send && n_local > 0 && parts > 0 && a_width > 0 && b_width >= 0 && (b || b_width == 0), "bad arguments");
  AIC_REQUIRE(a_width % 8 == 0 && b_width % 8 == 0 && a_stride % 8 == 0 && b_stride % 8 == 0,
              "widths and strides must be multiples of 8 elements (16 bytes)");
  AIC_NEED_DEVICE();
  const int aw = a_width / 8, bw = b_width / 8;
  const int64_t chunks = static_cast<int64_t>(parts) * n_local * (aw + bw);
  hipLaunchKernelGGL(ulysses_pack_kernel, dim3(grid_for(chunks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(a), static_cast<const uint4*>(b), static_cast<const uint4*>(nullptr),
                     a_stride / 8, b_stride / 8, static_cast<int64_t>(0), static_cast<uint4*>(send), n_local, parts, aw, bw, 0);
  return launch_status("ulysses_pack_kernel");
}

int aic_ulysses_reorder_split_kv(const void* gathered, void* k, void* v, int n_chunk_rows, int sp, int kv_width,
                                 const int32_t* order /*host*/, void* stream) {
  if (n_chunk_rows == 0) return AIC_OK;
  AIC_REQUIRE(gathered && k && v && order && n_chunk_rows > 0 && sp > 0 && sp <= 64 && kv_width > 0, "bad arguments");
  AIC_REQUIRE(kv_width % 8 == 0, "kv_width must be a multiple of 8 elements (16 bytes)");
  ChunkOrder o;
  unsigned long long seen = 0;
  for (int c = 0; c < sp; ++c) {
    AIC_REQUIRE(order[c] >= 0 && order[c] < sp, "chunk order entry %d out of range", c);
    seen |= 1ull << order[c];
    o.src[c] = order[c];
  }
  AIC_REQUIRE(seen == (sp == 64 ? ~0ull : ((1ull << sp) - 1)), "chunk order must be a permutation");
  AIC_NEED_DEVICE();
  const int kw = kv_width / 8;
  hipLaunchKernelGGL(ulysses_reorder_split_kernel, dim3(grid_for(static_cast<int64_t>(sp) * n_chunk_rows * 2 * kw)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const uint4*>(gathered), static_cast<uint4*>(k),
                     static_cast<uint4*>(v), n_chunk_rows, sp, kw, o);
  return launch_status("ulysses_reorder_split_kernel");
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
