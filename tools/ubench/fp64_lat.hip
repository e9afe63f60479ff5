// micro-benchmark: fp64 VALU issue cost vs dependent-chain latency on gfx950 (cycles per wave-instruction), at 1, 2 and
// 4 waves per SIMD.  Answers: how much of a Horner chain's latency do N co-resident waves hide?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CHAINS, int OP> __global__ void k(double *out, long long *cyc, int iters, double c0, double c1) {
  double a[CHAINS];
  for (int j = 0; j < CHAINS; j++) a[j] = threadIdx.x * 1e-3 + j;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int j = 0; j < CHAINS; j++) {
        if (OP == 0) a[j] = __builtin_fma(a[j], c0, c1);          // v_fma_f64
        else if (OP == 1) a[j] = a[j] * c0;                       // v_mul_f64
        else if (OP == 2) a[j] = a[j] + c1;                       // v_add_f64
        else a[j] = c1 / a[j];                                    // full IEEE division sequence
      }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int j = 0; j < CHAINS; j++) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS, int OP> void run(const char *name, double *out, long long *cyc, int wps) {
  const int iters = 20000;
  long long h;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, OP>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, 0.999999, 1e-9);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<CHAINS, OP>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters, 0.999999, 1e-9);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-8s chains=%d waves/SIMD=%d : %6.2f ticks per wave-instruction (per wave), %5.2f per SIMD-issued instruction; kernel %.3f ms for %lld ticks -> %.2f GHz tick rate\n", name, CHAINS, wps,
         (double)h / (iters * 8.0 * CHAINS), (double)h / (iters * 8.0 * CHAINS * wps), ms, h, h / (ms * 1e6));
}
int main() {
  double *out; long long *cyc;
  (void)hipMalloc(&out, 256 * 1024 * 8); (void)hipMalloc(&cyc, 8);
  for (int wps = 1; wps <= 4; wps *= 2) {
    run<1, 0>("fma_f64", out, cyc, wps); run<2, 0>("fma_f64", out, cyc, wps); run<4, 0>("fma_f64", out, cyc, wps); run<8, 0>("fma_f64", out, cyc, wps);
    run<1, 1>("mul_f64", out, cyc, wps); run<4, 1>("mul_f64", out, cyc, wps);
    run<1, 2>("add_f64", out, cyc, wps); run<4, 2>("add_f64", out, cyc, wps);
    run<1, 3>("div_f64", out, cyc, wps); run<4, 3>("div_f64", out, cyc, wps);
  }
  return 0;
}
