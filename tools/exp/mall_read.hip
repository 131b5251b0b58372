// Experiment (r04): does a read stream that fits the 256 MB memory-side cache (Infinity Cache) run faster than one from HBM?
// (Question behind it: would pulling part of the next LM head's weights in during the latency-bound LSTM cell kernel pay?)
// hipcc --offload-arch=gfx950 -O3 tools/exp/mall_read.hip -o tools/exp/mall_read.bin && tools/exp/mall_read.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <bool NT>
__global__ void __launch_bounds__(256) rd(const u32x4* __restrict__ p, size_t n_vec, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    u32x4 a, b, c, d;
    if (NT) {
      a = __builtin_nontemporal_load(p + i);
      b = __builtin_nontemporal_load(p + i + stride);
      c = __builtin_nontemporal_load(p + i + 2 * stride);
      d = __builtin_nontemporal_load(p + i + 3 * stride);
    } else {
      a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride];
    }
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n_vec; i += stride) acc ^= p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

int main() {
  const size_t max_bytes = 4ull << 30;
  u32x4* buf;
  unsigned* out;
  if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, max_bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t sizes_mb[] = {32, 64, 128, 192, 256, 384, 512, 1024, 2048, 4096};
  for (int nt = 0; nt < 2; ++nt)
    for (size_t mb : sizes_mb) {
      const size_t n_vec = (mb << 20) / 16;
      const int reps = static_cast<int>(16384 / mb) + 4;
      for (int grid : {1024, 2048}) {
        for (int w = 0; w < 3; ++w) {
          if (nt) hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, buf, n_vec, out);
          else hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(256), 0, 0, buf, n_vec, out);
        }
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) {
          if (nt) hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, buf, n_vec, out);
          else hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(256), 0, 0, buf, n_vec, out);
        }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("%s loads, %5zu MB re-read back to back, grid %4d: %8.1f us per pass = %6.2f TB/s\n", nt ? "nt     " : "default", mb, grid, us,
               static_cast<double>(mb << 20) / us / 1e6);
      }
    }
  return 0;
}
