// fma_operands.hip — cycles per dependent FP64 instruction of ONE wavefront by where its operands live (VGPR pairs,
// which ones, an SGPR pair), and the same with a second wavefront on the SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define BODY(txt) asm volatile(REP64(txt "\n") : : : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v31", "v32", "v33", "v34", "v35", "s20", "s21", "s22", "s23")

template <int KIND>
__global__ void k(unsigned long long *cycles)
{
  asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n v_mov_b32 v12, 1\n v_mov_b32 v13, 0x3ff00000\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3e100000\n"
               "v_mov_b32 v16, 3\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v20, 1\n v_mov_b32 v21, 0x3ff00000\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0x3e100000\n"
               "v_mov_b32 v31, 1\n v_mov_b32 v32, 0x3ff00000\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0x3e100000\n v_mov_b32 v35, 0x3e100000\n"
               "s_mov_b32 s20, 1\n s_mov_b32 s21, 0x3ff00000\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0x3e100000\n"
               : : : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v31", "v32", "v33", "v34", "v35", "s20", "s21", "s22", "s23");
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < 64; r++) {
    if (KIND == 0) BODY("v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]");
    if (KIND == 1) BODY("v_fma_f64 v[10:11], v[10:11], s[20:21], v[14:15]");
    if (KIND == 2) BODY("v_fma_f64 v[10:11], v[10:11], v[12:13], s[22:23]");
    if (KIND == 3) BODY("v_mul_f64 v[10:11], v[10:11], v[12:13]");
    if (KIND == 4) BODY("v_add_f64 v[10:11], v[10:11], v[14:15]");
    if (KIND == 5) BODY("v_fmac_f64 v[10:11], v[12:13], v[14:15]");
    if (KIND == 6) BODY("v_fma_f64 v[10:11], v[10:11], v[12:13], v[12:13]");
    if (KIND == 7) BODY("v_fma_f64 v[10:11], v[10:11], v[20:21], v[22:23]");
    if (KIND == 8) BODY("v_fma_f64 v[10:11], v[10:11], v[12:13], v[22:23]");
    if (KIND == 9) BODY("v_fma_f64 v[10:11], v[10:11], v[32:33], v[34:35]");
    if (KIND == 10) BODY("v_fma_f64 v[10:11], v[12:13], v[14:15], v[10:11]");
    if (KIND == 11) BODY("v_fma_f64 v[10:11], v[12:13], v[10:11], v[14:15]");
    if (KIND == 12) BODY("v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n v_fma_f64 v[16:17], v[16:17], v[12:13], v[14:15]");
    if (KIND == 13) BODY("v_fma_f64 v[10:11], v[10:11], v[12:13], v[14:15]\n v_mul_f64 v[16:17], v[16:17], v[12:13]");
    if (KIND == 14) BODY("v_fma_f64 v[10:11], v[10:11], 1.0, v[14:15]");
    if (KIND == 15) BODY("v_fma_f64 v[16:17], v[12:13], v[14:15], v[20:21]");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cycles[threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char *what, int per_loop)
{
  unsigned long long *d, h[8];
  hipMalloc(&d, sizeof(h));
  for (int waves = 1; waves <= 8; waves *= 8) {
    for (int rep = 0; rep < 2; rep++) k<KIND><<<1, 64 * waves>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
    printf("%-64s %s: %6.2f cycles per instruction%s\n", what, waves == 1 ? "lone wave      " : "2 waves per SIMD", mx / 4096.0 / per_loop / (waves == 1 ? 1 : 2),
           waves == 1 ? "" : " and SIMD");
  }
  hipFree(d);
}

int main()
{
  run<0>("v_fma_f64 a, a, v12, v14   (three VGPR pairs)", 1);
  run<1>("v_fma_f64 a, a, s20, v14", 1);
  run<2>("v_fma_f64 a, a, v12, s22", 1);
  run<14>("v_fma_f64 a, a, 1.0, v14", 1);
  run<3>("v_mul_f64 a, a, v12", 1);
  run<4>("v_add_f64 a, a, v14", 1);
  run<5>("v_fmac_f64 a, v12, v14", 1);
  run<6>("v_fma_f64 a, a, v12, v12", 1);
  run<7>("v_fma_f64 a, a, v20, v22", 1);
  run<8>("v_fma_f64 a, a, v12, v22", 1);
  run<9>("v_fma_f64 a, a, v[32:33], v[34:35] (pairs 32, 34)", 1);
  run<10>("v_fma_f64 a, v12, v14, a  (accumulator in src2)", 1);
  run<11>("v_fma_f64 a, v12, a, v14  (accumulator in src1)", 1);
  run<15>("v_fma_f64 v16, v12, v14, v20 (independent, three VGPR pairs)", 1);
  run<12>("two chains: fma a,a,v12,v14 ; fma b,b,v12,v14", 2);
  run<13>("fma a,a,v12,v14 ; mul b,b,v12", 2);
  return 0;
}
