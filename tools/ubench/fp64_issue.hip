// fp64_issue.hip — what one wavefront can issue on a gfx950 SIMD: cycles per v_fma_f64 for NACC independent
// accumulators (1 = a dependent chain), alone on its SIMD and with 1 or 2 more wavefronts of the same block beside it
// (a block of 4*k waves puts k on each SIMD).  Build: hipcc --offload-arch=gfx950 -O3 -o fp64_issue fp64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NACC, bool F32>
__global__ void issue_kernel(unsigned long long *cycles, double *sink, double m, double c0)
{
  double a[NACC];
  float af[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) { a[k] = threadIdx.x + k; af[k] = threadIdx.x + k; }
  const float mf = (float)m, cf = (float)c0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < 64; r++) {
#pragma unroll
    for (int u = 0; u < 64 / NACC; u++)
#pragma unroll
      for (int k = 0; k < NACC; k++) {
        if (F32) af[k] = __builtin_fmaf(af[k], mf, cf);
        else a[k] = __builtin_fma(a[k], m, c0);
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int k = 0; k < NACC; k++) s += a[k] + af[k];
  if (s == 12345.678) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, bool F32>
static void run(int waves)
{
  unsigned long long *d, h[16];
  double *sink;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&sink, 8);
  for (int rep = 0; rep < 2; rep++) issue_kernel<NACC, F32><<<1, 64 * waves>>>(d, sink, 1.0000001, 1e-9);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
  printf("%s acc=%2d waves/block=%2d (per SIMD %d): %.2f cycles per instruction and wave (slowest wave), %.2f per SIMD-instruction\n",
         F32 ? "f32" : "f64", NACC, waves, (waves + 3) / 4, mx / 4096.0, mx / 4096.0 / ((waves + 3) / 4));
  hipFree(d);
  hipFree(sink);
}

int main()
{
  run<1, false>(1); run<2, false>(1); run<4, false>(1); run<8, false>(1); run<16, false>(1);
  run<8, false>(4); run<8, false>(8); run<8, false>(12); run<8, false>(16);
  run<1, false>(8); run<1, false>(16); run<2, false>(8);
  run<1, true>(1); run<8, true>(1); run<8, true>(8); run<8, true>(16);
  return 0;
}
