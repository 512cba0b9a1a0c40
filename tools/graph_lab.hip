// Lab: does a hipGraph shorten a chain of ~40 small dependent kernels (the launch sequence of one gradient evaluation
// at n = 2000, m = 50) on this platform?  Times (a) plain stream launches + one synchronisation, (b) hipGraphLaunch of
// the captured chain + one synchronisation, for kernels that spin for ~t microseconds each.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin_kernel(double* x, int iters) {
  double v = x[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
  x[threadIdx.x] = v;
}
int main() {
  double* d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int iters : {0, 2000, 8000}) {
    for (int nk : {10, 40}) {
      auto chain = [&]() {
        for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, iters);
      };
      chain();
      hipStreamSynchronize(s);
      double best_a = 1e9, best_b = 1e9;
      for (int rep = 0; rep < 20; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        chain();
        hipStreamSynchronize(s);
        best_a = std::min(best_a, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      hipGraph_t g;
      hipGraphExec_t ge;
      hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
      chain();
      hipStreamEndCapture(s, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
      for (int rep = 0; rep < 20; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        best_b = std::min(best_b, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      printf("kernel spin iters %5d, chain of %2d: stream launches %.1f us, graph launch %.1f us\n", iters, nk, best_a, best_b);
      hipGraphExecDestroy(ge);
      hipGraphDestroy(g);
    }
  }
  return 0;
}
