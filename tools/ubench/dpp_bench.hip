// micro-benchmark: cost of the partner-exchange primitives on gfx950 (cycles per wave-instruction)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int M> __device__ __forceinline__ int dppx(int v) {
  if constexpr (M == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
  else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
  else if constexpr (M == 7) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
  else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);
  else return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
}
template <int KIND> __global__ void k(int *out, long long *cyc, int iters) {
  int v = threadIdx.x, acc = 0;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { v = dppx<1>(v) + 1; v = dppx<2>(v) + 1; v = dppx<1>(v) + 1; v = dppx<2>(v) + 1; }        // quad_perm chain
    else if (KIND == 1) { v = dppx<7>(v) + 1; v = dppx<8>(v) + 1; v = dppx<15>(v) + 1; v = dppx<7>(v) + 1; }  // row ops chain
    else if (KIND == 2) { v = __shfl_xor(v, 1, 64) + 1; v = __shfl_xor(v, 2, 64) + 1; v = __shfl_xor(v, 4, 64) + 1; v = __shfl_xor(v, 5, 64) + 1; }  // bpermute chain
    else if (KIND == 3) { v = v * 3 + 1; v = v * 5 + 1; v = v * 7 + 1; v = v * 9 + 1; }                       // plain VALU chain
    else if (KIND == 4) {  // 8 independent DPP (throughput)
      int a0 = dppx<1>(v), a1 = dppx<2>(v + 1), a2 = dppx<7>(v + 2), a3 = dppx<8>(v + 3);
      int a4 = dppx<1>(v + 4), a5 = dppx<2>(v + 5), a6 = dppx<7>(v + 6), a7 = dppx<15>(v + 7);
      v = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {  // 8 independent bpermutes (throughput)
      int a0 = __shfl_xor(v, 1, 64), a1 = __shfl_xor(v + 1, 2, 64), a2 = __shfl_xor(v + 2, 3, 64), a3 = __shfl_xor(v + 3, 4, 64);
      int a4 = __shfl_xor(v + 4, 5, 64), a5 = __shfl_xor(v + 5, 6, 64), a6 = __shfl_xor(v + 6, 7, 64), a7 = __shfl_xor(v + 7, 1, 64);
      v = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    acc += v;
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  int *out; long long *cyc, h;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  const char *names[6] = {"dpp quad_perm chain (4 dpp + 4 add)", "dpp row-op chain (4 dpp + 4 add)", "ds_bpermute chain (4 + 4 add)", "valu mad chain (4)", "8 indep dpp + 15 add", "8 indep bpermute + 15 add"};
  for (int waves = 1; waves <= 2; waves++) {
    for (int kind = 0; kind < 6; kind++) {
      dim3 g(256), b(64 * waves * 4);  // waves per SIMD = waves
      switch (kind) {
        case 0: k<0><<<g, b>>>(out, cyc, iters); break; case 1: k<1><<<g, b>>>(out, cyc, iters); break;
        case 2: k<2><<<g, b>>>(out, cyc, iters); break; case 3: k<3><<<g, b>>>(out, cyc, iters); break;
        case 4: k<4><<<g, b>>>(out, cyc, iters); break; default: k<5><<<g, b>>>(out, cyc, iters); break;
      }
      hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("waves/SIMD %d  %-40s %.1f clock64-ticks per loop iteration\n", waves, names[kind], (double)h / iters);
    }
  }
  return 0;
}
