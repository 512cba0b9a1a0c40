// Host-side accuracy check of gpr_amd/csrc/exp_fast.h against libm (the arithmetic is the same IEEE fma sequence the
// device executes).  Prints the largest error in ulps over the argument range the kernels produce.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define __host__
#define __device__
#define __forceinline__ inline
#include "../../gpr_amd/csrc/exp_fast.h"

static double ulp_err(double got, double ref) {
  if (ref == 0.0 || !std::isfinite(ref)) return got == ref ? 0.0 : 1e9;
  int e;
  std::frexp(ref, &e);
  const double ulp = std::ldexp(1.0, e - 53 < -1074 ? -1074 : e - 53);
  return std::fabs(got - ref) / ulp;
}

int main() {
  const gprhip::ExpK k = gprhip::exp_consts();
  double worst = 0.0, worst_x = 0.0;
  uint64_t s = 88172645463325252ULL;
  for (int i = 0; i < 20000000; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    const double x = (i & 1) ? -745.0 * u * u : 10.0 - 60.0 * u;   // dense near 0, reaching the subnormal range
    const double e = ulp_err(gprhip::exp_fast(x, k), std::exp(x));
    if (e > worst) { worst = e; worst_x = x; }
  }
  const double sp[] = {0.0, -0.0, 1e-300, -1e-300, -708.3, -745.2, -800.0, -1e5, -1e300, 709.0};
  int bad = 0;
  for (double x : sp) {
    const double g = gprhip::exp_fast(x, k), r = std::exp(x);
    if (ulp_err(g, r) > 1.0) { printf("special x=%g got %.17g ref %.17g\n", x, g, r); ++bad; }
  }
  const double nn = gprhip::exp_fast(std::nan(""), k);
  if (nn == nn) { printf("NaN did not propagate\n"); ++bad; }
  printf("exp_fast: max error %.3f ulp at x=%.17g; special cases bad=%d\n", worst, worst_x, bad);
  return (worst <= 1.0 && bad == 0) ? 0 : 1;
}
