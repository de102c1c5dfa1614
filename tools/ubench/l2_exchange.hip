// Micro-benchmark: what does ONE exchange of a 64-KB tile between two workgroups on CUs of the same
// XCD cost, through L2, with flag words - the hand-off a 4-CU version of k_split_reg (a 14-atom ket spread over 4 CUs,
// two cross-CU bits per stage) would pay once or twice per stage, against the 3.7 us a 2^12 tile's stage takes on one CU.
//
//   hipcc --offload-arch=gfx950 -O3 -o l2_exchange l2_exchange.hip && ./l2_exchange
//
// Protocol per iteration (both partners, symmetric): store my tile[it & 1] (16-byte stores) -> s_waitcnt + barrier ->
// lane 0: agent-scope RELEASE store of my flag = it + 1 -> lane 0 polls the partner's flag (relaxed loads + s_sleep) until
// >= it + 1, agent-scope ACQUIRE fence -> barrier -> load the partner's tile[it & 1] (plain 16-byte loads) and fold it
// into the registers (so that nothing is optimised away).  Two slots: a slot is rewritten at it + 2, after the partner's
// flag it + 2 has been seen, i.e. after it finished reading the slot.  Partners are workgroups b and b + 8 (workgroups
// are dealt round-robin over the 8 XCDs: same XCD, different CUs when the grid has <= 256 workgroups).
// Variants: store policy plain / sc1 write-through (inline asm), pairs active 1 / 8 / 64 / 128, tile 16 / 64 KB.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  double2* tiles;      // [n_wg][2][tile_elems]
  unsigned* flags;     // [n_wg] (64-byte apart)
  double2* sink;       // [n_wg][threads]
  int tile_elems, iters, n_pairs, sc1;
};

__device__ __forceinline__ void store16(double2* p, double2 v, int sc1) {
  if (sc1) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d w;
    w.x = v.x;
    w.y = v.y;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
  } else {
    *p = v;
  }
}

__global__ __launch_bounds__(1024) void k_exchange(Args A) {
  const int wg = blockIdx.x;
  const int pair = wg % 8 + 8 * (wg / 16);  // pairs (b, b + 8) inside groups of 16 workgroups
  const bool active = pair < A.n_pairs;
  if (!active) return;
  const int partner = (wg / 8) % 2 == 0 ? wg + 8 : wg - 8;
  const int per = A.tile_elems / blockDim.x;
  double2 r[8];
  for (int j = 0; j < per; ++j) r[j] = make_double2(wg + 0.001 * threadIdx.x, j);
  double2* mine = A.tiles + (size_t)wg * 2 * A.tile_elems;
  const double2* theirs = A.tiles + (size_t)partner * 2 * A.tile_elems;
  unsigned* my_flag = A.flags + wg * 16;
  unsigned* their_flag = A.flags + partner * 16;
  for (int it = 0; it < A.iters; ++it) {
    double2* slot = mine + (size_t)(it & 1) * A.tile_elems;
    for (int j = 0; j < per; ++j) store16(slot + j * blockDim.x + threadIdx.x, r[j], A.sc1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(my_flag, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 24)) break;  // (bounded: a lost partner must not hang the box)
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope by default for device code
    }
    __syncthreads();
    const double2* src = theirs + (size_t)(it & 1) * A.tile_elems;
    for (int j = 0; j < per; ++j) {
      const double2 v = src[j * blockDim.x + threadIdx.x];
      r[j].x = 0.5 * (r[j].x + v.x);
      r[j].y = 0.5 * (r[j].y + v.y);
    }
  }
  double2 acc = make_double2(0, 0);
  for (int j = 0; j < per; ++j) { acc.x += r[j].x; acc.y += r[j].y; }
  A.sink[(size_t)wg * blockDim.x + threadIdx.x] = acc;
}

// the same loop without a partner: stores + barrier + loads of the workgroup's OWN tile (what the data movement costs
// without any hand-off)
__global__ __launch_bounds__(1024) void k_selfcopy(Args A) {
  const int wg = blockIdx.x;
  if (wg % 8 + 8 * (wg / 16) >= A.n_pairs) return;
  const int per = A.tile_elems / blockDim.x;
  double2 r[8];
  for (int j = 0; j < per; ++j) r[j] = make_double2(wg + 0.001 * threadIdx.x, j);
  double2* mine = A.tiles + (size_t)wg * 2 * A.tile_elems;
  for (int it = 0; it < A.iters; ++it) {
    double2* slot = mine + (size_t)(it & 1) * A.tile_elems;
    for (int j = 0; j < per; ++j) store16(slot + j * blockDim.x + threadIdx.x, r[j], A.sc1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < per; ++j) {
      const double2 v = slot[j * blockDim.x + ((threadIdx.x + 64) & (blockDim.x - 1))];
      r[j].x = 0.5 * (r[j].x + v.x);
      r[j].y = 0.5 * (r[j].y + v.y);
    }
    __syncthreads();
  }
  double2 acc = make_double2(0, 0);
  for (int j = 0; j < per; ++j) { acc.x += r[j].x; acc.y += r[j].y; }
  A.sink[(size_t)wg * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const int n_wg = 256, threads = 1024, iters = 2000;
  Args A;
  CHECK(hipMalloc(&A.tiles, (size_t)n_wg * 2 * 4096 * sizeof(double2)));
  CHECK(hipMalloc(&A.flags, n_wg * 16 * sizeof(unsigned)));
  CHECK(hipMalloc(&A.sink, (size_t)n_wg * threads * sizeof(double2)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("| kernel | tile | stores | pairs | us per iteration |\n|---|---|---|---|---|\n");
  for (int self = 0; self < 2; ++self)
    for (int tile_kb : {16, 64})
      for (int sc1 = 0; sc1 < 2; ++sc1)
        for (int pairs : {1, 8, 64, 128}) {
          A.tile_elems = tile_kb * 1024 / 16;
          A.iters = iters;
          A.n_pairs = pairs;
          A.sc1 = sc1;
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(A.flags, 0, n_wg * 16 * sizeof(unsigned)));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            if (self) hipLaunchKernelGGL(k_selfcopy, dim3(n_wg), dim3(threads), 0, 0, A);
            else hipLaunchKernelGGL(k_exchange, dim3(n_wg), dim3(threads), 0, 0, A);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          if (!self) {  // every active workgroup must have reached `iters` (no partner was lost, no spin bound was hit)
            std::vector<unsigned> f(n_wg * 16);
            CHECK(hipMemcpy(f.data(), A.flags, f.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (int w = 0; w < n_wg; ++w)
              if (w % 8 + 8 * (w / 16) < pairs && f[w * 16] != (unsigned)iters) { printf("workgroup %d stopped at %u\n", w, f[w * 16]); exit(2); }
          }
          printf("| %s | %d KB | %s | %d | %.2f |\n", self ? "own tile (no hand-off)" : "exchange with partner", tile_kb,
                 sc1 ? "sc1 write-through" : "plain", pairs, best * 1e3 / iters);
        }
  return 0;
}
