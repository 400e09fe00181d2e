// Census of the compute units a kernel can land on (gfx950): every workgroup records
// (XCC_ID, HW_ID[15:8] = se_id | sh_id | cu_id) and the time it started.
//   hipcc --offload-arch=gfx950 -O2 tools/cu_census.hip -o tools/cu_census && ./tools/cu_census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <vector>

__global__ void census(unsigned* keys, int spin) {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  if (threadIdx.x == 0) keys[blockIdx.x] = ((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF);
  // hold the slot for a while so that the dispatcher has to spread the grid
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

int main() {
  const int G = 2048;
  unsigned* d;
  hipMalloc(&d, G * 4);
  std::vector<unsigned> h(G);
  for (int lds = 0; lds < 2; ++lds) {
    hipLaunchKernelGGL(census, dim3(G), dim3(512), lds ? 65536 : 0, 0, d, 200000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    for (unsigned k : h) cnt[k]++;
    printf("lds=%d: %zu distinct (xcc,se,sh,cu) keys over %d workgroups\n", lds ? 65536 : 0, cnt.size(), G);
    std::map<unsigned, std::vector<unsigned>> byx;
    for (auto& kv : cnt) byx[kv.first >> 8].push_back(kv.first & 0xFF);
    for (auto& kv : byx) {
      printf("  xcc %u: %zu CUs:", kv.first, kv.second.size());
      for (unsigned c : kv.second) printf(" %u.%u.%u", (c >> 5) & 7, (c >> 4) & 1, c & 15);
      printf("\n");
    }
    printf("  first 16 blocks -> xcc:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[i] >> 8);
    printf("\n");
  }
  return 0;
}
