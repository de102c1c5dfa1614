// Microbenchmark (dev probe; not part of the product): v_mfma_f64_4x4x4_f64 -
//  (1) lane layout of A / B / D (unit-vector probing: which (lane of A, lane of B) products land in which lane of D),
//  (2) issue rate alone, and (3) overlap with independent v_fma_f64 on the vector pipe (the question behind
//      "MFMA beside VALU" for k_ket: do partner sums on the matrix pipe free VALU issue slots?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(double* out) {
  const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  out[blockIdx.x * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

// NM matrix instructions (KIND 0: 4x4x4, 1: 16x16x4) + NV independent vector FMAs per iteration
template <int KIND, int NM, int NV>
__global__ __launch_bounds__(512) void k_mix(double* out, int iters, double a0, double b0) {
  double acc1[4] = {0, 0, 0, 0};
  d4 acc4[4];
  for (int i = 0; i < 4; ++i) acc4[i] = (d4){0, 0, 0, 0};
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 * (i + 1) + threadIdx.x;
  const double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < (NM > NV ? NM : NV); ++i) {
      if (i < NM) {
        if (KIND == 0) acc1[i & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i & 3], 0, 0, 0);
        else acc4[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i & 3], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < (NM ? (NV + NM - 1) / NM : 0); ++r)
        if (i < NM && i * ((NV + NM - 1) / NM) + r < NV) v[(i + r) & 7] = __builtin_fma(v[(i + r) & 7], b, a);
      if (NM == 0 && i < NV) v[i & 7] = __builtin_fma(v[i & 7], b, a);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc1[i] + acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NM, int NV>
void run(int blocks, int threads, int iters) {
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mix<KIND, NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 0.999);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_mix<KIND, NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 0.999);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%s NM=%2d NV=%2d blocks=%4d threads=%3d: %8.3f ms  = %7.1f ns per iteration per wave\n",
         KIND ? "16x16x4" : "4x4x4  ", NM, NV, blocks, threads, ms, ms * 1e6 / iters);
  hipFree(out);
}

int main() {
  // ---- layout ----
  double* out;
  hipMalloc(&out, sizeof(double) * 4096 * 64);
  hipLaunchKernelGGL(k_layout, dim3(4096), dim3(64), 0, 0, out);
  std::vector<double> h(4096 * 64);
  hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  // hypothesis: A(i, k) in lane i + 4 k + 16 blk?  print what is found instead of assuming
  int shown = 0;
  for (int L = 0; L < 64; ++L) {
    printf("D lane %2d <-", L);
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (h[((size_t)la * 64 + lb) * 64 + L] != 0.0) printf(" (A%d,B%d)", la, lb);
    printf("\n");
    if (++shown >= 24) break;
  }
  // check the formula D[blk][i][j] = sum_k A[blk][i][k] B[blk][k][j] under candidate lane maps
  const char* names[4] = {"A:i+4k B:j+4k D:j+4i", "A:i+4k B:j+4k D:i+4j", "A:k+4i B:k+4j D:j+4i", "A:k+4i B:j+4k D:j+4i"};
  for (int c = 0; c < 4; ++c) {
    bool ok = true;
    for (int blk = 0; blk < 4 && ok; ++blk)
      for (int i = 0; i < 4 && ok; ++i)
        for (int j = 0; j < 4 && ok; ++j)
          for (int la = 0; la < 64 && ok; ++la)
            for (int lb = 0; lb < 64 && ok; ++lb) {
              const int L = 16 * blk + (c == 1 ? i + 4 * j : j + 4 * i);
              double expect = 0.0;
              for (int k = 0; k < 4; ++k) {
                const int al = 16 * blk + ((c == 2 || c == 3) ? k + 4 * i : i + 4 * k);
                const int bl = 16 * blk + (c == 2 ? k + 4 * j : j + 4 * k);
                if (al == la && bl == lb) expect += 1.0;
              }
              if (h[((size_t)la * 64 + lb) * 64 + L] != expect) ok = false;
            }
    printf("layout candidate [%s]: %s\n", names[c], ok ? "MATCH" : "no");
  }
  hipFree(out);
  // ---- rates: 256 blocks x 512 threads = two waves per SIMD on every CU (the k_ket configuration) ----
  const int it = 20000;
  run<0, 0, 32>(256, 512, it);
  run<0, 8, 0>(256, 512, it);
  run<0, 8, 32>(256, 512, it);
  run<0, 16, 32>(256, 512, it);
  run<0, 4, 32>(256, 512, it);
  run<1, 8, 0>(256, 512, it);
  run<1, 8, 32>(256, 512, it);
  run<1, 4, 32>(256, 512, it);
  run<1, 2, 32>(256, 512, it);
  run<0, 8, 0>(256, 256, it);   // one wave per SIMD
  run<0, 0, 32>(256, 256, it);
  run<0, 8, 32>(256, 256, it);
  return 0;
}
