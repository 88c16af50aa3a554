// ccmp_kernels_debug.hip — device side of include/ccmp_debug.h; linked into lib/libccmp_debug.so only.
// Built in the canonical rounding model with the throughput kernel's elementary functions (-DCCMP_LEAN_SQRT included): what the
// probe reports is what project_fd_kernel computes with.
#include <hip/hip_runtime.h>

#include "ccmp_detmath.h"

namespace {

__global__ void detmath_probe_kernel(const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  ccmp_sincos(x[i], &s, &c);
  out[5 * i + 0] = s;
  out[5 * i + 1] = c;
  out[5 * i + 2] = ccmp_atan2_nn(ccmp_abs(x[i]), ccmp_abs(y[i]));
  out[5 * i + 3] = ccmp_sqrt(ccmp_abs(x[i]));
  out[5 * i + 4] = x[i] / y[i];
}

} // namespace

extern "C" hipError_t ccmp_launch_detmath_probe(const double *x, const double *y, double *out, size_t n, hipStream_t st)
{
  hipLaunchKernelGGL(detmath_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, out, n);
  return hipGetLastError();
}
