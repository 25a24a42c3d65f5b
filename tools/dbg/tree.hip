#include "../../3dgs_monocular_depth_init_amd/csrc/common.h"
#include <cstdio>
namespace gsr { void set_error(const char*, ...) {} }
__global__ void k(double* out){
  int lane = threadIdx.x;
  float v[8];
  for (int i=0;i<8;++i) v[i] = (i+1) * 1.0f + 0.0f*lane;   // each lane holds (k+1): total = 64*(k+1)
  float s01 = gsr::swap32_add(v[0], v[1]);
  float s23 = gsr::swap32_add(v[2], v[3]);
  float r1 = gsr::swap16_add(s01, s23);
  float t1 = gsr::dpp_add<0x128>(r1);
  out[lane] = s01; out[64+lane] = s23; out[128+lane] = r1; out[192+lane] = t1;
  out[256+lane] = gsr::tree_reduce8(v, lane);
  out[320+lane] = gsr::tree8_index(lane);
}
int main(){
  double* d; (void)hipMalloc(&d, 384*8); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[384]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"s01","s23","r1","t1","tree","idx"};
  for (int s=0;s<6;++s){ printf("%s:", names[s]); for(int i=0;i<64;i+=4) printf(" %g", h[s*64+i]); printf("\n"); }
  return 0;
}
