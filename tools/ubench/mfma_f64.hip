// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (dev probe; not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int threads, int iters) {
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 2.0);
  hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)blocks * threads / 64, mf = waves * iters * NACC;
  printf("NACC=%d blocks=%d threads=%d: %.3f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", NACC, blocks, threads, ms,
         ms * 1e6 / (iters * NACC), mf * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  run<8>(1, 64, 100000);     // one wave
  run<8>(1, 256, 100000);    // one wave per SIMD of one CU
  run<8>(256, 256, 100000);  // one wave per SIMD, all CUs
  run<8>(512, 256, 100000);  // two waves per SIMD
  run<8>(768, 256, 100000);  // three
  run<8>(1024, 256, 100000); // four
  run<4>(2048, 256, 50000);  // eight waves per SIMD (4 accumulators)
  run<4>(256, 256, 100000);
  run<2>(256, 256, 100000);
  run<1>(256, 256, 100000);  // dependent chain
  return 0;
}
