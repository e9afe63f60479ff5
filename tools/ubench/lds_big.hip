// Does a >64 KiB static LDS array behave on gfx950?  Each thread writes a tagged value to every
// slot of its column, a second pass checks them (per-thread columns, as the step kernel's cold slots).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(256) void k(int *bad, int rounds) {
  __shared__ double s[N][256];
  const int t = threadIdx.x;
  int nb = 0;
  for (int r = 0; r < rounds; r++) {
    for (int j = 0; j < N; j++) s[j][t] = (double)(j * 1000 + t + r + blockIdx.x);
    for (int j = 0; j < N; j++) nb += s[j][t] != (double)(j * 1000 + t + r + blockIdx.x);
    __syncthreads();
  }
  if (nb) atomicAdd(bad, nb);
}
int main() {
  int *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  k<36><<<2048, 256>>>(bad, 50);
  hipError_t e = hipDeviceSynchronize();
  int h = -1; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("N=36 (73728 B): err=%s bad=%d\n", hipGetErrorString(e), h);
  hipMemset(bad, 0, 4);
  k<70><<<2048, 256>>>(bad, 50);
  e = hipDeviceSynchronize();
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("N=70 (143360 B): err=%s bad=%d\n", hipGetErrorString(e), h);
  return 0;
}
