/* Wall time of gprhip_eval through the C ABI alone (no Python around it): what a compiled host -- the OCaml stubs of
 * bindings/ -- pays per optimiser callback at small problem sizes.  Plain C99: also shows the header compiles as C.
 *   usage (GPU box): gpr_amd/_build/eval_latency [n m d [reps]]        (default 2000 50 3 200) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "gprhip.h"

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
static unsigned long long state = 88172645463325252ULL;
static double uniform(void) {  /* xorshift64 */
  state ^= state << 13; state ^= state >> 7; state ^= state << 17;
  return (double)(state >> 11) / 9007199254740992.0;
}
static double normal(void) { return sqrt(-2.0 * log(uniform() + 1e-300)) * cos(6.283185307179586 * uniform()); }
static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 2000;
  const int m = argc > 2 ? atoi(argv[2]) : 50, d = argc > 3 ? atoi(argv[3]) : 3, reps = argc > 4 ? atoi(argv[4]) : 200;
  double* X = malloc(sizeof(double) * n * d);
  double* y = malloc(sizeof(double) * n);
  double* Z = malloc(sizeof(double) * m * d);
  for (long i = 0; i < n; ++i) {
    double s = 0.0;
    for (int k = 0; k < d; ++k) s += (X[i * d + k] = normal());
    y[i] = sin(s) + 0.1 * normal();
  }
  for (int c = 0; c < m; ++c)
    for (int k = 0; k < d; ++k) Z[c * d + k] = X[(long)(c * (n / m)) * d + k] + 0.01 * normal();
  gprhip_problem* p = NULL;
  if (gprhip_problem_create(0, GPRHIP_COV_SE_ISO, n, d, d, m, 0, &p) != GPRHIP_OK || gprhip_set_inputs(p, X, d) != GPRHIP_OK ||
      gprhip_set_targets(p, y) != GPRHIP_OK) {
    fprintf(stderr, "%s\n", gprhip_last_error());
    return 1;
  }
  gprhip_hypers h = {0};
  h.log_ell = 0.5 * log((double)d); h.log_sf2 = 0.0; h.sigma2 = 0.1; h.inducing = Z; h.jitter = 1e-6;
  gprhip_result res;
  double* grad = malloc(sizeof(double) * (2 + (size_t)m * d));
  double* t = malloc(sizeof(double) * reps);
  for (int want_grad = 1; want_grad >= 0; --want_grad) {
    for (int it = 0; it < reps + 5; ++it) {
      Z[0] += 1e-9;  /* (inducing points change every call, as under an optimiser) */
      const double t0 = now();
      if (gprhip_eval(p, &h, want_grad, &res, grad, NULL) != GPRHIP_OK) {
        fprintf(stderr, "%s\n", gprhip_last_error());
        return 1;
      }
      if (it >= 5) t[it - 5] = now() - t0;
    }
    qsort(t, reps, sizeof(double), cmp);
    printf("n=%ld m=%d d=%d %s: median %.1f us, min %.1f us over %d calls (l = %.6f)\n", n, m, d,
           want_grad ? "gprhip_eval with gradient" : "gprhip_eval, log evidence only", 1e6 * t[reps / 2], 1e6 * t[0], reps, res.l);
  }
  gprhip_problem_destroy(p);
  return 0;
}
