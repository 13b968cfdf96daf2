// GPU box: issue rate of the fp64 instructions the viscosity rows are made of, per SIMD, at 1 / 2 / 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/fp64_rate tools/ubench/fp64_rate.hip ; run: /tmp/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND, int CHAINS>
__global__ void __launch_bounds__(256) k(double* out, float* fin, int iters, double a, double b) {
  double v[CHAINS];
  float f[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { v[c] = threadIdx.x * 1e-3 + c; f[c] = fin[(threadIdx.x + c) & 255]; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (KIND == 0) v[c] = __builtin_fma(v[c], a, b);                    // dependent v_fma_f64 chains
        if (KIND == 1) { v[c] = v[c] * a; }                                  // v_mul_f64
        if (KIND == 2) { v[c] = v[c] + b; }                                  // v_add_f64
        if (KIND == 3) { v[c] += (double)f[c]; f[c] = f[c] * 1.0001f; }      // cvt + add + an fp32 mul
        if (KIND == 4) { f[c] = __builtin_fmaf(f[c], 1.0001f, 0.5f); }       // v_fma_f32
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += v[c] + f[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

template <int KIND, int CHAINS>
void run(const char* name, int blocks_per_cu, int cus, double* dout, float* dfin) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, dout, dfin, iters, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, dout, dfin, iters, 1.0000001, 1e-9);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc; hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost);
  const double ops = (double)iters * 16 * CHAINS;                 // wave-instructions of the measured kind per wave
  printf("%-28s chains %d  waves/SIMD %d : %.2f shader cycles per wave-instruction (one wave's view), %.2f cycles per "
         "instruction per SIMD, wall %.3f ms\n", name, CHAINS, blocks_per_cu, cyc / ops, cyc / ops / blocks_per_cu, ms);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* dout; float* dfin;
  hipMalloc(&dout, 8 * 256 * cus * 8); hipMalloc(&dfin, 4 * 256);
  std::vector<float> h(256, 1.5f); hipMemcpy(dfin, h.data(), 1024, hipMemcpyHostToDevice);
  for (int w : {1, 2, 4}) {
    run<0, 1>("v_fma_f64 dependent", w, cus, dout, dfin);
    run<0, 4>("v_fma_f64", w, cus, dout, dfin);
    run<0, 8>("v_fma_f64", w, cus, dout, dfin);
    run<1, 4>("v_mul_f64", w, cus, dout, dfin);
    run<2, 4>("v_add_f64", w, cus, dout, dfin);
    run<3, 4>("cvt_f64_f32+add_f64+mul_f32", w, cus, dout, dfin);
    run<4, 4>("v_fma_f32", w, cus, dout, dfin);
    run<4, 8>("v_fma_f32", w, cus, dout, dfin);
  }
  return 0;
}
