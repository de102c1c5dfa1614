// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                        \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess)                                                   \
      return fail(RYD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                  hipGetErrorString(e_), __FILE__, __LINE__);               \
  } while (0)

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
struct Segs {
  int lo[3];
  int len[3];
};

__host__ __device__ __forceinline__ unsigned long long deposit(
    unsigned long long v, const Segs& s) {
  unsigned long long r = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r |= (v & ((1ull << s.len[i]) - 1ull)) << s.lo[i];
    v >>= s.len[i];
  }
  return r;
}

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx cfma(cplx a, cplx b, cplx c) {  // a*b + c
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)),
                      fma(a.x, b.y, fma(a.y, b.x, c.y)));
}

// wave-uniform double -> scalar registers
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
