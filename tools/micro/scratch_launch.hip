// micro-benchmark: what does a dependent, EMPTY kernel cost in a stream when it is compiled with / without scratch (private memory),
// in 1-wave or 4-wave workgroups?  (A -> B pairs, B exits after one scalar load.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) kA(float* p, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
template <int SCRATCH, int THREADS>
__global__ void __launch_bounds__(THREADS) kB(const int* flag, float* out, int idx) {
  __shared__ float lds[1536];
  if (flag[0] == 0) return;
  float a[SCRATCH > 0 ? SCRATCH : 1];
  for (int k = 0; k < (SCRATCH > 0 ? SCRATCH : 1); k++) a[k] = out[k] + threadIdx.x;
  lds[threadIdx.x] = a[(idx + threadIdx.x) % (SCRATCH > 0 ? SCRATCH : 1)];
  __syncthreads();
  out[blockIdx.x * THREADS + threadIdx.x] = lds[(threadIdx.x + 1) % THREADS];
}
template <int SCRATCH, int THREADS>
static void run(const char* name, int grid, float* p, int n, int* flag, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  float ms_a, ms_ab;
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kA, dim3((n + 255) / 256), dim3(256), 0, 0, p, n);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) {
      hipLaunchKernelGGL(kA, dim3((n + 255) / 256), dim3(256), 0, 0, p, n);
      hipLaunchKernelGGL((kB<SCRATCH, THREADS>), dim3(grid), dim3(THREADS), 0, 0, flag, out, r);
    }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_ab, e0, e1);
  }
  printf("%-34s grid %5d x %3d: A alone %.1f us, A + empty B %.1f us -> B adds %.1f us\n", name, grid, THREADS, 1e3 * ms_a / reps, 1e3 * ms_ab / reps, 1e3 * (ms_ab - ms_a) / reps);
}
int main() {
  int n = 32 * 100032;
  float* p; int* flag; float* out;
  hipMalloc(&p, sizeof(float) * n); hipMemset(p, 0, sizeof(float) * n);
  hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
  hipMalloc(&out, sizeof(float) * 8192 * 256);
  run<0, 64>("no scratch, 1-wave WGs", 6272, p, n, flag, out);
  run<0, 256>("no scratch, 4-wave WGs", 1568, p, n, flag, out);
  run<200, 64>("800 B scratch, 1-wave WGs", 6272, p, n, flag, out);
  run<200, 256>("800 B scratch, 4-wave WGs", 1568, p, n, flag, out);
  run<200, 256>("800 B scratch, 4-wave WGs, big grid", 12504, p, n, flag, out);
  run<0, 256>("no scratch, 4-wave WGs, big grid", 12504, p, n, flag, out);
  return 0;
}
