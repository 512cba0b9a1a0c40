// exp() for the covariance and gradient kernels: the argument is log_sf2 - 1/2 |p - z|^2 / ell^2, evaluated ~2e9 times
// per pass at the headline shape.  The compiler's code for the library exp() spends as many instructions on
// materialising its 64-bit polynomial coefficients (one v_mov_b64 in front of every v_fmac_f64) as on arithmetic;
// here the coefficients are pinned in scalar registers, so a Horner step is a single v_fma_f64:
//   n = rint(x log2 e),  r = x - n ln2 (Cody-Waite, two FMAs),  exp(r) = 1 + r + r^2 q(r),  q of degree 11,
//   result = ldexp(., n)
// |r| <= ln2/2: truncation error r^14/14! < 4.2e-18 (0.04 ulp); measured against the host's libm over [-745, 10]:
// <= 1 ulp (tests/cpp/exp_check.cpp, run by tests/test_host.py).  Arguments below -800 (results far below the
// smallest subnormal) are clamped so that the integer conversion stays in range; NaN propagates.
#pragma once

namespace gprhip {

struct ExpK {
  double l2e, ln2hi, ln2lo, c[12];
};

#if defined(__HIP_DEVICE_COMPILE__)
#define GPRHIP_PIN_SGPR(v) asm volatile("" : "+s"(v))
#else
#define GPRHIP_PIN_SGPR(v) (void)0
#endif

// Call once per kernel, outside the loops (30 scalar registers).
__host__ __device__ __forceinline__ ExpK exp_consts() {
  ExpK k;
  k.l2e = 1.4426950408889634074;          // log2(e)
  k.ln2hi = 6.93147180369123816490e-01;   // 0x3fe62e42fee00000: the low 21 bits are zero, n * ln2hi is exact
  k.ln2lo = 1.90821492927058770002e-10;   // ln2 - ln2hi
  double f = 2.0;
  for (int i = 0; i < 12; ++i) {          // c[i] = 1 / (i + 2)!
    k.c[i] = 1.0 / f;
    f *= (double)(i + 3);
  }
  GPRHIP_PIN_SGPR(k.l2e);
  GPRHIP_PIN_SGPR(k.ln2hi);
  GPRHIP_PIN_SGPR(k.ln2lo);
#pragma unroll
  for (int i = 0; i < 12; ++i) GPRHIP_PIN_SGPR(k.c[i]);
  return k;
}

__host__ __device__ __forceinline__ double exp_fast(double x, const ExpK& k) {
  x = (x < -800.0) ? -800.0 : x;
  const double n = __builtin_rint(x * k.l2e);
  double r = __builtin_fma(-n, k.ln2hi, x);
  r = __builtin_fma(-n, k.ln2lo, r);
  double q = k.c[11];
#pragma unroll
  for (int i = 10; i >= 0; --i) q = __builtin_fma(q, r, k.c[i]);
  const double p = __builtin_fma(r * r, q, r);
  return __builtin_ldexp(1.0 + p, (int)n);
}

}  // namespace gprhip
