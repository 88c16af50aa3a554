// lone_wave_slots.hip — what the instructions that are not arithmetic cost ONE wavefront alone on its SIMD (the latency
// kernel's situation): a stream of dependent v_fma_f64 with one extra instruction after every FMA; cycles per (FMA +
// extra) pair minus the plain FMA stream = the price of the extra.  Build: hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void slots_kernel(unsigned long long *cycles, double *sink, double m, double c0, const double *tab)
{
  __shared__ double lds[64];
  lds[threadIdx.x & 63] = tab[threadIdx.x & 63];
  __syncthreads();
  double a = threadIdx.x, b = 0.0;
  int v = threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < 64; r++) {
    if (KIND == 0) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]" : [a] "+v"(a) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 1) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n s_nop 0" : [a] "+v"(a) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 2) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n s_mov_b32 s20, 0x12345678" : [a] "+v"(a) : [m] "v"(m), [c] "v"(c0) : "s20");) }
    if (KIND == 3) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n s_waitcnt lgkmcnt(0)" : [a] "+v"(a) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 4) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n v_mov_b32 %[v], 0x12345678" : [a] "+v"(a), [v] "=v"(v) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 5) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n v_cndmask_b32 %[v], %[v], %[v], vcc" : [a] "+v"(a), [v] "+v"(v) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 6) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x3ff12345" : [a] "+v"(a) : [m] "v"(m), [c] "v"(c0) : "s20", "s21");) }
    if (KIND == 7) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n ds_read_b64 %[b], %[p]" : [a] "+v"(a), [b] "=v"(b) : [m] "v"(m), [c] "v"(c0), [p] "v"((threadIdx.x & 63) * 8));) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if (KIND == 8) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n v_fma_f64 %[b], %[b], %[m], %[c]" : [a] "+v"(a), [b] "+v"(b) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 9) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n v_mov_b32_dpp %[v], %[v] row_shl:1 row_mask:0xf bank_mask:0xf" : [a] "+v"(a), [v] "+v"(v) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 10) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n v_rcp_f64 %[b], %[b]" : [a] "+v"(a), [b] "+v"(b) : [m] "v"(m), [c] "v"(c0));) }
    if (KIND == 11) { REP64(asm volatile("v_fma_f64 %[a], %[a], %[m], %[c]\n v_fma_f64 %[b], %[b], %[m], s[20:21]" : [a] "+v"(a), [b] "+v"(b) : [m] "v"(m), [c] "v"(c0));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (a + b + v == 12345.678) sink[0] = a;
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
static double run(const char *what, double base)
{
  unsigned long long *d, h = 0;
  double *sink, *tab;
  hipMalloc(&d, 8);
  hipMalloc(&sink, 8);
  hipMalloc(&tab, 512);
  hipMemset(tab, 0, 512);
  for (int rep = 0; rep < 2; rep++) slots_kernel<KIND><<<1, 64>>>(d, sink, 1.0000001, 1e-9, tab);
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double per = h / 4096.0;
  printf("%-44s %6.2f cycles per pair  -> the extra costs %5.2f\n", what, per, per - base);
  hipFree(d); hipFree(sink); hipFree(tab);
  return per;
}

int main()
{
  const double base = run<0>("v_fma_f64 (dependent) alone", 0.0);
  run<8>("+ a second, independent v_fma_f64", base);
  run<1>("+ s_nop 0", base);
  run<2>("+ s_mov_b32 (literal)", base);
  run<6>("+ 2 x s_mov_b32 (one FP64 literal)", base);
  run<3>("+ s_waitcnt lgkmcnt(0), nothing outstanding", base);
  run<4>("+ v_mov_b32 (literal)", base);
  run<5>("+ v_cndmask_b32", base);
  run<9>("+ v_mov_b32_dpp row_shl:1", base);
  run<7>("+ ds_read_b64 (waited for once per 64)", base);
  run<10>("+ v_rcp_f64", base);
  run<11>("+ independent v_fma_f64 with an SGPR-pair operand", base);
  return 0;
}
