// cu_mask.hip — where a CU-masked stream's blocks run on gfx950 (round 5).  hipExtStreamCreateWithCUMask takes one bit per CU;
// this prints, for a few masks, which (XCC, SE, SH, CU) the blocks of a 4 096-block launch landed on — the bit -> CU layout a
// caller needs before reserving CUs for one launch beside another — and what a masked launch costs beside an unmasked stream's.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/cu_mask tools/ubench/cu_mask.hip && tools/ubench/cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(unsigned int *out, int spin)
{
  if (threadIdx.x == 0) {
    const unsigned int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    const unsigned int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay a while so that the launch spreads over every CU the queue may use
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
}

__global__ void busy_kernel(double *out, int n)
{
  double a = threadIdx.x * 1e-3, b = 1.000001;
  for (int i = 0; i < n; i++) a = a * b + 1e-9;
  if (a == 42.0) out[0] = a;
}

static int report(const char *name, hipStream_t st, unsigned int *d, std::vector<unsigned int> &h)
{
  const int G = 4096;
  hipLaunchKernelGGL(where_kernel, dim3(G), dim3(64), 0, st, d, 2000 /* 20 us */);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(h.data(), d, 2 * G * sizeof(unsigned int), hipMemcpyDeviceToHost));
  std::set<unsigned int> cus;
  int per_xcc[16] = {0};
  std::set<unsigned int> per_xcc_cus[16];
  for (int b = 0; b < G; b++) {
    const unsigned int hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    const unsigned int key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cus.insert(key);
    per_xcc[xcc]++;
    per_xcc_cus[xcc].insert(key & 0xfff);
  }
  printf("%-34s distinct CUs %3zu; per XCC:", name, cus.size());
  for (int x = 0; x < 8; x++) printf(" %zu", per_xcc_cus[x].size());
  printf("\n");
  if (cus.size() <= 40) {
    printf("    (xcc.se.sh.cu):");
    for (unsigned int k : cus) printf(" %u.%u.%u.%u", k >> 12, (k >> 8) & 0xf, (k >> 4) & 0xf, k & 0xf);
    printf("\n");
  }
  return 0;
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
  unsigned int *d;
  CK(hipMalloc(&d, 2 * 4096 * sizeof(unsigned int)));
  std::vector<unsigned int> h(2 * 4096);
  hipStream_t plain;
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  if (report("unmasked stream", plain, d, h)) return 1;

  struct { const char *name; unsigned int w[8]; } masks[] = {
      {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 0,8,16,24", {0x01010101u, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu}},
      {"all but bits 0..31", {0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
      {"every fourth bit (64 CUs)", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}},
  };
  std::vector<hipStream_t> sts;
  for (auto &m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
    if (e != hipSuccess) { printf("%-34s hipExtStreamCreateWithCUMask -> %s\n", m.name, hipGetErrorString(e)); continue; }
    sts.push_back(st);
    if (report(m.name, st, d, h)) return 1;
  }
  // two launches side by side: 32 reserved CUs + the other 224, against both on unmasked streams
  if (sts.size() >= 5) {
    double *sink;
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    hipStream_t plain2;
    CK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
    for (int masked = 0; masked < 2; masked++) {
      hipStream_t a = masked ? sts[0] : plain, b = masked ? sts[4] : plain2;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, a));
        hipLaunchKernelGGL(busy_kernel, dim3(64), dim3(128), 0, a, sink, 200000); // 64 two-wave blocks: two per reserved CU
        CK(hipEventRecord(e1, a));
        hipLaunchKernelGGL(busy_kernel, dim3(8192), dim3(256), 0, b, sink, 20000);
        CK(hipEventRecord(e2, b));
        CK(hipDeviceSynchronize());
        float ta, tb;
        CK(hipEventElapsedTime(&ta, e0, e1));
        CK(hipEventElapsedTime(&tb, e0, e2));
        if (rep == 2) printf("%s: 64 serial-chain blocks %.3f ms beside a chip-filling launch (ends %.3f ms after the chain's start)\n",
                             masked ? "32 reserved CUs + 224" : "two unmasked streams   ", ta, tb);
      }
    }
  }
  return 0;
}
