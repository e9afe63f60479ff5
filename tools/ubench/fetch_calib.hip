// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS path's access widths (the guide's x2 FETCH correction was measured
// on 16 B/lane streams; step_kernel loads 8 B/lane doubles and 1 B/lane bytes): known byte counts in, counters out.
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib   and   rocprofv3 --pmc WRITE_SIZE -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void copy_f64(const double *__restrict__ a, double *__restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i] * 1.0000001;
}
__global__ void copy_u8(const uint8_t *__restrict__ a, uint8_t *__restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = (uint8_t)(a[i] + 1);
}
__global__ void copy_f32x30(const float *__restrict__ a, float *__restrict__ b, size_t n) {  // the obs row pattern: float2 runs
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ((float2 *)b)[i] = ((const float2 *)a)[i];
}
int main() {
  const size_t n = (size_t)64 << 20;  // 64 Mi elements: 512 MiB of doubles, far beyond L2 + Infinity Cache
  double *a, *b;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
  (void)hipMemset(a, 0, n * 8);
  for (int r = 0; r < 3; r++) {
    hipLaunchKernelGGL(copy_f64, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(copy_u8, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const uint8_t *)a, (uint8_t *)b, n);
    hipLaunchKernelGGL(copy_f32x30, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const float *)a, (float *)b, n);
  }
  (void)hipDeviceSynchronize();
  printf("copy_f64: %zu B read, %zu B written per launch; copy_u8: %zu / %zu; copy_f32x30 (float2): %zu / %zu\n", n * 8, n * 8, n, n, n * 8, n * 8);
  return 0;
}
