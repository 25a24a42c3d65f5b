// Gap between two dependent kernels of one stream as a function of the SECOND kernel's grid size (and of its LDS /
// workgroup shape): is the ~6 us hole rocprofv3 shows in front of every 4 000-8 000-workgroup launch of the training step
// (and in front of none of the <= 1 020-workgroup ones) the dispatcher's start-up for a large grid?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/launch_gap.hip -o /tmp/launch_gap
//   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/lg -o lg -- /tmp/launch_gap && python3 tools/ubench/launch_gap_digest.py /tmp/lg/lg_kernel_trace.csv
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void producer(float *p, int n) {      // ~5 us of dependent work in one workgroup
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
template <int TAG>
__global__ void consumer(const float *p, float *out, int work) {
  float v = p[threadIdx.x & 63];
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) out[blockIdx.x] = v;
}
#define RUN(TAG, GRID, BLOCK)                                                         \
  for (int r = 0; r < 30; ++r) {                                                      \
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, 0, p, 3000);                   \
    hipLaunchKernelGGL(consumer<TAG>, dim3(GRID), dim3(BLOCK), 0, 0, p, out, 2000);   \
  }                                                                                   \
  hipDeviceSynchronize();
int main() {
  float *p, *out;
  hipMalloc(&p, 4096);
  hipMalloc(&out, 1 << 22);
  hipMemset(p, 0, 4096);
  RUN(256, 256, 64)
  RUN(1020, 1020, 64)
  RUN(2048, 2048, 64)
  RUN(4096, 4096, 64)
  RUN(8160, 8160, 64)
  RUN(32640, 32640, 64)
  RUN(3907, 3907, 256)
  RUN(1021, 1020, 1024)
  printf("done\n");
  return 0;
}
