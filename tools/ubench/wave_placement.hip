// wave_placement.hip — on which SIMD of its CU does each wavefront of a 128-thread block land?  The latency kernels give wave 0
// of a block the heavier role (arm 0 + the serial solve, ~1 620 instructions per round against ~1 090): if the dispatcher puts
// every block's wave 0 on the same two SIMDs, those two carry 60 % of the CU's instruction stream.  Launches the latency kernel's
// shape (128 threads, 12 KB LDS, 128 registers, 2 048 co-resident blocks) and prints, per SIMD, how many wave-0s and wave-1s it
// holds.  Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/wave_placement tools/ubench/wave_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(128, 4) void placement_kernel(unsigned int *out, unsigned int *arrived, unsigned int nblocks)
{
  __shared__ double pad[1536]; // 12 KB like the latency kernel's record
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    const unsigned int hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID
    const unsigned int xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);  // HW_REG_XCC_ID
    out[2 * (2 * blockIdx.x + (threadIdx.x >> 6))] = hw;
    out[2 * (2 * blockIdx.x + (threadIdx.x >> 6)) + 1] = xcc;
  }
  // stay resident until every block has arrived (all 2 048 fit: 8 per CU), bounded so that a smaller chip still ends
  if (threadIdx.x == 0) {
    atomicAdd(arrived, 1u);
    for (int spin = 0; spin < 2000000 && __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks; spin++) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  if (pad[threadIdx.x] < 0) out[0] = 0;
}

int main()
{
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const unsigned int nblocks = 8 * prop.multiProcessorCount;
  unsigned int *out, *arrived;
  hipMalloc(&out, 4 * nblocks * sizeof(unsigned int));
  hipMalloc(&arrived, sizeof(unsigned int));
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(arrived, 0, sizeof(unsigned int));
    hipLaunchKernelGGL(placement_kernel, dim3(nblocks), dim3(128), 0, 0, out, arrived, nblocks);
    hipDeviceSynchronize();
    std::vector<unsigned int> h(4 * nblocks);
    hipMemcpy(h.data(), out, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost);
    // per CU (xcc, se, sh, cu) and SIMD: wave-0 count, wave-1 count
    std::map<unsigned int, std::vector<int>> cu;
    int same_simd = 0, pair_hist[4][4] = {};
    for (unsigned int b = 0; b < nblocks; b++) {
      unsigned int simd[2];
      for (int w = 0; w < 2; w++) {
        const unsigned int hw = h[2 * (2 * b + w)], xcc = h[2 * (2 * b + w) + 1];
        simd[w] = (hw >> 4) & 3;
        const unsigned int key = (xcc << 16) | (hw & 0xff00); // cu_id 11:8, sh_id 12, se_id 15:13
        auto &v = cu[key];
        if (v.empty()) v.assign(8, 0);
        v[2 * simd[w] + w]++;
      }
      same_simd += simd[0] == simd[1];
      pair_hist[simd[0]][simd[1]]++;
    }
    // distribution of (wave-0s on the SIMD) over all SIMDs of all CUs
    int hist0[17] = {}, worst = 0;
    long heavy_max_sum = 0;
    for (auto &kv : cu) {
      int mx = 0;
      for (int s = 0; s < 4; s++) {
        hist0[kv.second[2 * s] > 16 ? 16 : kv.second[2 * s]]++;
        const int load = 1620 * kv.second[2 * s] + 1090 * kv.second[2 * s + 1];
        if (load > mx) mx = load;
      }
      heavy_max_sum += mx;
      if (mx > worst) worst = mx;
    }
    printf("launch %d: %zu CUs seen, %u blocks; blocks with both waves on one SIMD: %d\n", rep, cu.size(), nblocks, same_simd);
    printf("  (SIMD of wave 0, SIMD of wave 1) counts:");
    for (int a = 0; a < 4; a++) for (int c = 0; c < 4; c++) if (pair_hist[a][c]) printf(" (%d,%d) %d", a, c, pair_hist[a][c]);
    printf("\n  SIMDs holding k wave-0s:");
    for (int k = 0; k <= 16; k++) if (hist0[k]) printf(" k=%d: %d", k, hist0[k]);
    printf("\n  instructions per round on the busiest SIMD of a CU (wave 0 = 1620, wave 1 = 1090): mean %.0f, worst %d; balanced would be %d\n",
           (double)heavy_max_sum / cu.size(), worst, (1620 + 1090) * 8 / 4);
  }
  return 0;
}
