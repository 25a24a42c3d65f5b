#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out){
  unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0]; out[64+lane] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128+lane] = q[0]; out[192+lane] = q[1];
  int m = __builtin_amdgcn_update_dpp(0, (int)lane, 0x141, 0xf, 0xf, false);
  out[256+lane] = m;
  int ro = __builtin_amdgcn_update_dpp(0, (int)lane, 0x128, 0xf, 0xf, false);
  out[320+lane] = ro;
}
int main(){
  unsigned* d; hipMalloc(&d, 384*4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"p32.r0","p32.r1","p16.r0","p16.r1","half_mirror","ror8"};
  for (int s=0;s<6;++s){ printf("%s:", names[s]); for(int i=0;i<64;++i) printf(" %u", h[s*64+i]); printf("\n"); }
  return 0;
}
