// atomic_rate.hip — what a ticket word costs on gfx950 (round 5): G one-wavefront blocks, lane 0 of each issues N dependent
// agent-scope operations on (a) ONE shared word: fetch-add, (b) one shared word: relaxed atomic load, (c) a word of its own:
// fetch-add, (d) one shared word: fetch-add whose result is not used (fire and forget, then one wait).  Prints ns per operation
// as the kernel's duration / N (= the latency a block sees per operation under that contention) and the chip-wide rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/atomic_rate tools/ubench/atomic_rate.hip && tools/ubench/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_shared_add(unsigned long long *w, int n, unsigned long long *sink)
{
  if (threadIdx.x) return;
  unsigned long long acc = 0;
  for (int i = 0; i < n; i++) acc += atomicAdd(w, 1ull);
  sink[blockIdx.x] = acc;
}
__global__ void k_shared_load(unsigned long long *w, int n, unsigned long long *sink)
{
  if (threadIdx.x) return;
  unsigned long long acc = 0;
  for (int i = 0; i < n; i++) acc += __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + i;
  sink[blockIdx.x] = acc;
}
__global__ void k_own_add(unsigned long long *w, int n, unsigned long long *sink)
{
  if (threadIdx.x) return;
  unsigned long long acc = 0;
  for (int i = 0; i < n; i++) acc += atomicAdd(w + 32 * (blockIdx.x + 1), 1ull);
  sink[blockIdx.x] = acc;
}
__global__ void k_shared_add_10lanes(unsigned long long *w, int n, unsigned long long *sink) // ten lanes of the wavefront, same word: the group kernels' refill
{
  if (threadIdx.x % 6 != 0 || threadIdx.x >= 60) return;
  unsigned long long acc = 0;
  for (int i = 0; i < n; i++) acc += atomicAdd(w, 1ull);
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}
__global__ void k_shared_add_noret(unsigned long long *w, int n, unsigned long long *sink)
{
  if (threadIdx.x) return;
  for (int i = 0; i < n; i++) __hip_atomic_fetch_add(w, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  sink[blockIdx.x] = 0;
}

int main()
{
  unsigned long long *w, *sink;
  hipMalloc(&w, 8 * 32 * 4100);
  hipMalloc(&sink, 8 * 4100);
  hipMemset(w, 0, 8 * 32 * 4100);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int n = 200;
  struct { const char *name; void (*k)(unsigned long long *, int, unsigned long long *); } ks[] = {
      {"shared word, fetch-add (result used)", k_shared_add}, {"shared word, atomic load", k_shared_load},
      {"own word, fetch-add", k_own_add}, {"shared word, fetch-add (no return)", k_shared_add_noret},
      {"shared word, fetch-add from 10 lanes", k_shared_add_10lanes}};
  for (int g : {1, 256, 2048, 3072}) {
    for (auto &k : ks) {
      hipLaunchKernelGGL(k.k, dim3(g), dim3(64), 0, 0, w, n, sink);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k.k, dim3(g), dim3(64), 0, 0, w, n, sink);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%5d blocks  %-40s %8.1f ns per operation and block, %7.2f ns per operation chip-wide (%.1f M/s)\n", g, k.name, ms * 1e6 / n,
             ms * 1e6 / n / g, (double)n * g / ms / 1e3);
    }
  }
  return 0;
}
