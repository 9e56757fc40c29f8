// micro-benchmark: start-up latency of a kernel whose every wave walks a chain of DEPENDENT uniform loads before it ends, launched
// behind a streaming kernel in the same stream (what the prologue of the sweep kernels does: launch arguments -> job table ->
// descriptor -> state -> data).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) kA(float* p, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
template <int DEPTH, bool VECTOR>
__global__ void __launch_bounds__(256) kB(const int* __restrict__ chain, float* out) {
  int idx = 0;
  if (VECTOR) {
    for (int d = 0; d < DEPTH; d++) idx = chain[idx + (threadIdx.x & 1)];        // per-lane (vector) loads, two addresses
  } else {
    for (int d = 0; d < DEPTH; d++) idx = chain[idx];                            // uniform -> scalar loads
  }
  if (idx == 12345) out[threadIdx.x] = 1.f;
}
template <int DEPTH, bool VECTOR>
static void run(float* p, int n, int* chain, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  float ms_a = 0, ms_ab = 0;
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kA, dim3((n + 255) / 256), dim3(256), 0, 0, p, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) {
      hipLaunchKernelGGL(kA, dim3((n + 255) / 256), dim3(256), 0, 0, p, n);
      hipLaunchKernelGGL((kB<DEPTH, VECTOR>), dim3(1568), dim3(256), 0, 0, chain, out);
    }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_ab, e0, e1);
  }
  printf("%s chain depth %d: A alone %.1f us, A + B %.1f us -> B adds %.1f us\n", VECTOR ? "vector" : "scalar", DEPTH, 1e3 * ms_a / reps, 1e3 * ms_ab / reps, 1e3 * (ms_ab - ms_a) / reps);
}
int main() {
  int n = 32 * 100032;
  float* p; int* chain; float* out;
  hipMalloc(&p, sizeof(float) * n); hipMemset(p, 0, sizeof(float) * n);
  int h[4096]; for (int i = 0; i < 4096; i++) h[i] = (i * 64 + 64) % 4096;   // each step lands on another cache line
  hipMalloc(&chain, sizeof(h)); hipMemcpy(chain, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&out, 4096);
  run<0, false>(p, n, chain, out); run<1, false>(p, n, chain, out); run<2, false>(p, n, chain, out); run<4, false>(p, n, chain, out); run<8, false>(p, n, chain, out);
  run<1, true>(p, n, chain, out); run<4, true>(p, n, chain, out); run<8, true>(p, n, chain, out);
  return 0;
}
