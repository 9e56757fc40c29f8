// probe: operand / result lane map of v_mfma_f64_4x4x4_4b_f64 on gfx950 (one-hot A lane x one-hot B lane -> which D lane is non-zero)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m ? (int)__ffsll((long long)m) - 1 : -1;
    }
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // hypothesis: A lane = 16 k + 4 b + i ; B lane = 16 k + 4 b + j ; D lane = 16 i + 4 b + j
  int bad = 0, nz = 0;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      int ka = la >> 4, ba = (la >> 2) & 3, i = la & 3, kb = lb >> 4, bb = (lb >> 2) & 3, j = lb & 3;
      int expect = (ka == kb && ba == bb) ? 16 * i + 4 * ba + j : -1;
      if (h[la * 64 + lb] >= 0) nz++;
      if (h[la * 64 + lb] != expect) bad++;
    }
  printf("non-zero combos %d, mismatches against hypothesis (A: 16k+4b+i, B: 16k+4b+j, D: 16i+4b+j): %d\n", nz, bad);
  if (bad) for (int la = 0; la < 64; la += 1) { printf("la %2d:", la); for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf(" (%d->%d)", lb, h[la * 64 + lb]); printf("\n"); }
  return 0;
}
