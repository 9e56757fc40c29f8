// micro-benchmark: issue rate of the two f64 MFMA shapes on gfx950 (cycles per instruction per SIMD, one wave per SIMD and 4 waves per SIMD)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_rate.hip -o tools/micro/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ void k(double* out, long long* cyc, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = 0, t1 = 0;
  if constexpr (SHAPE == 16) {
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int SHAPE>
static void run(const char* name, int threads) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * 1024 * 256); hipMalloc(&cyc, 8);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  int waves_per_simd = threads / 64 / 4; if (waves_per_simd < 1) waves_per_simd = 1;
  double per_simd_instr = 4.0 * iters * waves_per_simd;
  printf("%-28s threads/WG %4d: %.3f ms -> %.1f ns per MFMA per SIMD (clock counter: %.1f ticks per MFMA of wave 0)\n", name, threads, ms, 1e6 * ms / per_simd_instr,
         (double)c / (4.0 * iters));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<16>("v_mfma_f64_16x16x4_f64", 64);
  run<16>("v_mfma_f64_16x16x4_f64", 256);
  run<16>("v_mfma_f64_16x16x4_f64", 1024);
  run<4>("v_mfma_f64_4x4x4_4b_f64", 64);
  run<4>("v_mfma_f64_4x4x4_4b_f64", 256);
  run<4>("v_mfma_f64_4x4x4_4b_f64", 1024);
  return 0;
}
