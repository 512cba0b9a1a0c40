// Internal hooks of a device problem that the multi-device context (ctx.hip) drives; not part of the C ABI.
#pragma once
#include "../../include/gprhip.h"
#include "common.h"

namespace gprhip {

// the problem's own exchange buffers (gprhip_ar1_len / gprhip_ar2_len doubles, device memory) and stream
double* problem_ar1(gprhip_problem* p);
double* problem_ar2(gprhip_problem* p);
hipStream_t problem_hip_stream(gprhip_problem* p);
int problem_device(const gprhip_problem* p);
int64_t problem_rows(const gprhip_problem* p);

// The finish stage in two halves (gprhip_eval_finish = both): enqueue puts the m x m work of the gradient and the
// result copies on the problem's stream without blocking; collect waits for the stream and assembles the results.
// light != 0: only the factorisation flags are fetched and the model state validated (shards whose results nobody reads).
// Both return a GPRHIP_* status with the message left in the calling thread's gprhip_last_error().
int problem_finish_enqueue(gprhip_problem* p, const double* d_ar2, int light);
int problem_finish_collect(gprhip_problem* p, gprhip_result* res, double* grad, double* coeffs, int light);

}  // namespace gprhip
