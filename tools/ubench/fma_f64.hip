// Microbenchmark: v_fma_f64 issue rate and shader clock under fp64 load (dev probe).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters, double a, double b) {
  double x[NCH];
  for (int i = 0; i < NCH; ++i) x[i] = a + threadIdx.x + i;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) x[i] = fma(x[i], b, a);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NCH>
void run(int blocks, int threads, int iters) {
  double* out; long long* clk; hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, clk, 10, 1.0, 0.999);
  hipEventRecord(e0); hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 1.0, 0.999); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double fl = 2.0 * blocks * threads * (double)iters * NCH;
  printf("NCH=%d blocks=%d threads=%d: %.3f ms, %.1f TFLOP/s fp64 FMA; clock64 %lld ticks, wall_clock64 %lld ticks (100 MHz -> %.3f ms) => clock64 rate %.0f MHz\n",
         NCH, blocks, threads, ms, fl / (ms * 1e-3) / 1e12, h[0], h[1], h[1] / 1e5, h[0] / (h[1] / 100.0));
}
int main() {
  run<8>(256, 256, 200000);
  run<8>(1024, 256, 100000);
  run<8>(2048, 256, 50000);
  run<16>(1024, 256, 50000);
  return 0;
}
