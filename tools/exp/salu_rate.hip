// Experiment (r04): what one wave pays per scalar instruction on gfx950, alone on its SIMD and with a partner wave that
// runs VALU + MFMA work — the long-draft attention body had ~110 scalar instructions per 32-token tile beside its vector work.
// hipcc --offload-arch=gfx950 -O3 tools/exp/salu_rate.hip -o tools/exp/salu_rate.bin && tools/exp/salu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define STAMP(x) x = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// kind: which sequence the MEASURED waves (even workgroups when partner = 1) run; odd workgroups run a VALU/MFMA filler
__global__ void __launch_bounds__(256) k(int64_t* out, int kind, int partner, int reps) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const bool filler = partner && (blockIdx.x & 1);
  uint64_t t0, t1;
  if (filler) {
    f32x4 acc = {0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = b[i] = static_cast<__bf16>(1.0f + threadIdx.x);
    float x = threadIdx.x;
    for (int r = 0; r < reps * 6; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        x = x * 1.0001f + 0.5f;
        x = x * 0.9999f + 0.25f;
      }
    }
    if (acc[0] + x == 12345.678f) out[0] = 1;
    return;
  }
  unsigned s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 7;
  STAMP(t0);
  for (int r = 0; r < reps; ++r) {
    if (kind == 0) {          // 64 dependent s_add_u32
      asm volatile(".rept 64\n\ts_add_u32 %0, %0, %1\n\t.endr" : "+s"(s0) : "s"(s1) : "scc");
    } else if (kind == 1) {   // 64 s_add_u32 on four independent registers
      asm volatile(".rept 16\n\ts_add_u32 %0, %0, %4\n\ts_add_u32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_add_u32 %3, %3, %4\n\t.endr"
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(blockIdx.x) : "scc");
    } else if (kind == 2) {   // 64 dependent s_mul_hi_u32
      asm volatile(".rept 64\n\ts_mul_hi_u32 %0, %0, %1\n\t.endr" : "+s"(s0) : "s"(s1));
    } else if (kind == 3) {   // 64 dependent s_mul_i32
      asm volatile(".rept 64\n\ts_mul_i32 %0, %0, %1\n\t.endr" : "+s"(s0) : "s"(s1));
    } else if (kind == 4) {   // 64 dependent v_add_u32 (VALU calibration)
      unsigned v = threadIdx.x;
      asm volatile(".rept 64\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(v) : "s"(s1));
      s0 += __builtin_amdgcn_readfirstlane(v);
    } else if (kind == 5) {   // 64 s_nop 0
      asm volatile(".rept 64\n\ts_nop 0\n\t.endr");
    } else if (kind == 6) {   // 16 stamps
      uint64_t t;
      for (int i = 0; i < 16; ++i) { STAMP(t); s0 += static_cast<unsigned>(t); }
    } else if (kind == 7) {   // 32 x (s_cmp + s_cselect)
      asm volatile(".rept 32\n\ts_cmp_lt_u32 %0, %1\n\ts_cselect_b32 %0, %0, %1\n\t.endr" : "+s"(s0) : "s"(s1) : "scc");
    } else if (kind == 8) {   // 32 v_readlane_b32 (independent)
      unsigned v = threadIdx.x * 3;
      asm volatile(".rept 32\n\tv_readlane_b32 %0, %1, 5\n\t.endr" : "+s"(s0) : "v"(v));
    }
  }
  STAMP(t1);
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = static_cast<int64_t>(t1 - t0) + (s0 == 0x7fffffffu);
}

int main() {
  int64_t* d;
  const int n = 1024 * 4 * 2;
  hipMalloc(&d, n * 8);
  const char* names[] = {"64 dependent s_add_u32", "64 s_add_u32, 4 independent chains", "64 dependent s_mul_hi_u32", "64 dependent s_mul_i32",
                         "64 dependent v_add_u32", "64 s_nop 0", "16 x (s_memtime + lgkmcnt(0))", "32 x (s_cmp + s_cselect)", "32 v_readlane_b32"};
  const int per[] = {64, 64, 64, 64, 64, 64, 16, 64, 32};
  const int reps = 200;
  for (int partner = 0; partner < 2; ++partner)
    for (int kind = 0; kind < 9; ++kind) {
      hipMemset(d, 0, n * 8);
      // partner = 0: 256 workgroups of 4 waves = one wave per SIMD; partner = 1: 512 workgroups, odd ones are fillers
      hipLaunchKernelGGL(k, dim3(partner ? 512 : 256), dim3(256), 0, 0, d, kind, partner, reps);
      hipDeviceSynchronize();
      std::vector<int64_t> h(n);
      hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
      std::vector<double> v;
      for (int i = 0; i < n; i += 2)
        if (h[i] > 0) v.push_back(static_cast<double>(h[i]) / (reps * per[kind]));
      std::sort(v.begin(), v.end());
      printf("%-12s %-40s %6.2f cycles per instruction (median of %zu waves; min %.2f max %.2f)\n",
             partner ? "MFMA partner" : "alone", names[kind], v[v.size() / 2], v.size(), v.front(), v.back());
    }
  return 0;
}
