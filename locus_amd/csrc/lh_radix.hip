// lh_radix.hip -- every sort and scan of the path, hand-written for gfx950 (wave64 ballot ranking, LDS digit tables): the SEGMENTED
// least-significant-digit radix sort of the batched index build (K2), the one-segment sort of the voxel grid and the local map (K1,
// SURVEY 8f-1), and the inclusive scan behind every stream compaction.  No library sort or scan is linked.
//
// The build sorts (cloud id << 32 | 30-bit Morton key) -> point index for all targets admitted together.  The cloud id only
// says which segment of the concatenated array an element belongs to, and the segments are already in id order, so the
// library sort's 37-bit passes over one long array are replaced by three 10-bit passes over 30-bit keys INSIDE each segment
// (grid.y = cloud): 32-bit keys in flight instead of 64-bit ones, three passes instead of five, nothing shared between clouds.
// A pass is three launches:
//   k_rs_hist     a workgroup counts the digits of its tile (4096 consecutive elements) in LDS -> hist[cloud][tile][digit]
//   k_rs_scan     one 256-thread workgroup per cloud, four digits per thread: exclusive offsets in (digit, tile) order
//   k_rs_scatter  the tile again: a wave ranks 64 consecutive elements at a time with wave64 ballots (ten ballots give every
//                 lane the mask of the lanes that hold its digit; rank = popcount of the lower lanes), running per-wave digit
//                 counters in LDS, then the waves' counters are prefixed in wave order -> stable positions
// Stable, deterministic, and identical in output to a stable sort of the 64-bit keys (equal keys keep ascending point index):
// tests/test_gpu_kernels.py::test_index_sort_matches_library_sort checks it against the one-segment sort below through LH_SORT=check.
#include "lh_kernels.hpp"

namespace lh {

constexpr int RS_BITS = 10, RS_BINS = 1 << RS_BITS, RS_TILE = 4096, RS_PER_THREAD = RS_TILE / 256;

// Elements in flight between the passes are (key, point index) PAIRS in one 8-byte word: an LSD pass scatters every element
// to its own place (the low key digits of neighbouring points are unrelated), so what counts is the number of isolated
// stores, and a pair costs one instead of two.  k_key_b writes the pairs, the last pass writes (cloud << 32 | key) and the index.
__device__ __forceinline__ uint32_t rs_key(const void* keys, size_t g) { return reinterpret_cast<const uint2*>(keys)[g].x; }

__global__ void __launch_bounds__(256) k_rs_hist(const IndexDesc* __restrict__ descs, const void* __restrict__ keys, int shift,
                                                 uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[RS_BINS];
  const int cloud = blockIdx.y, tile = blockIdx.x;
  const int n = descs[cloud].n, off = descs[cloud].offset;
  if (tile * RS_TILE >= n) return;
  for (int b = threadIdx.x; b < RS_BINS; b += 256) h[b] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_PER_THREAD; r++) {
    int i = tile * RS_TILE + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(rs_key(keys, (size_t)off + i) >> shift) & (RS_BINS - 1)], 1u);
  }
  __syncthreads();
  uint32_t* out = hist + ((size_t)descs[cloud].tile0 + tile) * RS_BINS;
  for (int b = threadIdx.x; b < RS_BINS; b += 256) out[b] = h[b];
}

// thread = digit: offsets[tile][digit] = (elements of smaller digits) + (elements of this digit in earlier tiles)
// 256 threads, four digits each (one 16-byte load per tile row): a 1024-thread workgroup has to wait until a whole CU's worth of wave
// slots is free, and beside sixteen streams of sweeps that took 100-600 us for 6 us of work -- on the critical path of every group's
// index build (rocprofv3, timed region of the bench).
__global__ void __launch_bounds__(256) k_rs_scan(const IndexDesc* __restrict__ descs, uint32_t* __restrict__ hist) {
  __shared__ uint32_t wave_tot[4];
  const int cloud = blockIdx.x, b = threadIdx.x, lane = b & 63, wave = b >> 6;
  const int tiles = (descs[cloud].n + RS_TILE - 1) / RS_TILE;
  uint4* h = reinterpret_cast<uint4*>(hist + (size_t)descs[cloud].tile0 * RS_BINS) + b;   // row stride RS_BINS / 4 uint4
  constexpr int RS = RS_BINS / 4;
  uint4 tot = make_uint4(0u, 0u, 0u, 0u);
  int t0 = 0;
  for (; t0 + 4 <= tiles; t0 += 4) {   // four independent loads in flight
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = h[(size_t)(t0 + k) * RS];
#pragma unroll
    for (int k = 0; k < 4; k++) { tot.x += v[k].x; tot.y += v[k].y; tot.z += v[k].z; tot.w += v[k].w; }
  }
  for (; t0 < tiles; t0++) { uint4 v = h[(size_t)t0 * RS]; tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w; }
  const uint32_t mine = tot.x + tot.y + tot.z + tot.w;
  uint32_t inc = mine;   // inclusive scan over the 256 four-digit totals: inside the wave, then across the 4 waves
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; w++) base += wave_tot[w];
  uint4 run;
  run.x = base + inc - mine;
  run.y = run.x + tot.x;
  run.z = run.y + tot.y;
  run.w = run.z + tot.z;
  t0 = 0;
  for (; t0 + 4 <= tiles; t0 += 4) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = h[(size_t)(t0 + k) * RS];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      h[(size_t)(t0 + k) * RS] = run;
      run.x += v[k].x; run.y += v[k].y; run.z += v[k].z; run.w += v[k].w;
    }
  }
  for (; t0 < tiles; t0++) {
    uint4 c = h[(size_t)t0 * RS];
    h[(size_t)t0 * RS] = run;
    run.x += c.x; run.y += c.y; run.z += c.z; run.w += c.w;
  }
}

template <bool kLast>
__global__ void __launch_bounds__(256) k_rs_scatter(const IndexDesc* __restrict__ descs, const void* __restrict__ keys_in,
                                                    int shift, const uint32_t* __restrict__ offs, void* __restrict__ keys_out,
                                                    uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t cnt[4][RS_BINS];   // per-wave running digit counters, then the waves' base positions
  const int cloud = blockIdx.y, tile = blockIdx.x;
  const int n = descs[cloud].n, off = descs[cloud].offset;
  if (tile * RS_TILE >= n) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int b = tid; b < 4 * RS_BINS; b += 256) (&cnt[0][0])[b] = 0;
  __syncthreads();
  // wave w owns elements [w * 512, (w + 1) * 512) of the tile, 64 consecutive ones per batch: (wave, batch, lane) is tile order
  uint32_t key[RS_PER_THREAD], val[RS_PER_THREAD], rank[RS_PER_THREAD];
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RS_PER_THREAD; r++) {
    const int i = tile * RS_TILE + wave * (RS_TILE / 4) + r * 64 + lane;
    const bool live = i < n;
    key[r] = 0xffffffffu; val[r] = 0;
    if (live) {
      const uint2 kv = reinterpret_cast<const uint2*>(keys_in)[(size_t)off + i];
      key[r] = kv.x; val[r] = kv.y;
    }
    const uint32_t dig = live ? ((key[r] >> shift) & (RS_BINS - 1)) : RS_BINS;   // bit 10 set: idle lanes match only each other
    unsigned long long m = ~0ull;
#pragma unroll
    for (int bit = 0; bit <= RS_BITS; bit++) {
      const unsigned long long bal = __ballot((dig >> bit) & 1u);
      m &= ((dig >> bit) & 1u) ? bal : ~bal;
    }
    // the lowest lane of every digit group advances the wave's counter of that digit and hands the old value to its group
    const int leader = __ffsll((long long)m) - 1;
    uint32_t old = 0;
    if (live && lane == leader) { old = cnt[wave][dig]; cnt[wave][dig] = old + (uint32_t)__popcll(m); }
    old = __shfl(old, leader, 64);
    rank[r] = old + (uint32_t)__popcll(m & below);
  }
  __syncthreads();
  // cnt[w][d] = elements of digit d in wave w -> position of wave w's first such element
  const uint32_t* o = offs + ((size_t)descs[cloud].tile0 + tile) * RS_BINS;
  for (int b = tid; b < RS_BINS; b += 256) {
    uint32_t run = o[b];
#pragma unroll
    for (int w = 0; w < 4; w++) { uint32_t c = cnt[w][b]; cnt[w][b] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_PER_THREAD; r++) {
    const int i = tile * RS_TILE + wave * (RS_TILE / 4) + r * 64 + lane;
    if (i < n) {
      const uint32_t dig = (key[r] >> shift) & (RS_BINS - 1);
      const size_t pos = (size_t)off + cnt[wave][dig] + rank[r];
      if constexpr (kLast) {
        reinterpret_cast<uint64_t*>(keys_out)[pos] = ((uint64_t)(uint32_t)cloud << 32) | key[r];
        vals_out[pos] = val[r];
      } else
        reinterpret_cast<uint2*>(keys_out)[pos] = make_uint2(key[r], val[r]);
    }
  }
}

// ===== one-segment radix sort of u32 / u64 keys (+ u32 values): the voxel grid's and the local map's sort (K1) ========
// The same three-launch pass -- per-tile digit counts in LDS, offsets, stable ballot-ranked scatter -- over ONE array of any length:
//   k_gs_hist     tile t (4096 elements) -> hist[t][digit]
//   k_gs_colscan  one WAVE per digit: exclusive prefix of that digit's column over the tiles (64 tiles per step, wave scan) in place,
//                 and the digit's total -> total[digit]     (a tile loop in one workgroup would take ~1 us per tile: 1 000 tiles at 4 M points)
//   k_gs_scatter  the tile again; the base of digit d = sum of total[d' < d], formed by every workgroup itself (1 024 values)
// ceil(end_bit / 10) passes from bit 0, ping-ponging between the caller's in/out arrays and the temporary ones; stable.
template <class K>
__global__ void __launch_bounds__(256) k_gs_hist(const K* __restrict__ keys, int n, int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[RS_BINS];
  const int tile = blockIdx.x;
  for (int b = threadIdx.x; b < RS_BINS; b += 256) h[b] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_PER_THREAD; r++) {
    const int i = tile * RS_TILE + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (RS_BINS - 1)], 1u);
  }
  __syncthreads();
  uint32_t* out = hist + (size_t)tile * RS_BINS;
  for (int b = threadIdx.x; b < RS_BINS; b += 256) out[b] = h[b];
}
__global__ void __launch_bounds__(256) k_gs_colscan(uint32_t* __restrict__ hist, int tiles, uint32_t* __restrict__ total) {
  const int digit = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint32_t run = 0;
  for (int t0 = 0; t0 < tiles; t0 += 64) {
    const int t = t0 + lane;
    const uint32_t v = t < tiles ? hist[(size_t)t * RS_BINS + digit] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (t < tiles) hist[(size_t)t * RS_BINS + digit] = run + inc - v;
    run += __shfl(inc, 63, 64);
  }
  if (lane == 0) total[digit] = run;
}
template <class K, bool kVals>
__global__ void __launch_bounds__(256) k_gs_scatter(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int n, int shift,
                                                    const uint32_t* __restrict__ offs, const uint32_t* __restrict__ total, K* __restrict__ keys_out,
                                                    uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t cnt[4][RS_BINS];   // per-wave running digit counters, then the waves' base positions
  __shared__ uint32_t base[RS_BINS];     // elements of smaller digits in the whole array
  __shared__ uint32_t wsum[4];
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int b = tid; b < 4 * RS_BINS; b += 256) (&cnt[0][0])[b] = 0;
  {  // exclusive scan of the 1 024 digit totals: four consecutive digits per thread, wave scan, four wave sums
    const uint4 t4 = reinterpret_cast<const uint4*>(total)[tid];
    const uint32_t mine = t4.x + t4.y + t4.z + t4.w;
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t b0 = inc - mine;
    for (int w = 0; w < wave; w++) b0 += wsum[w];
    base[4 * tid] = b0;
    base[4 * tid + 1] = b0 + t4.x;
    base[4 * tid + 2] = b0 + t4.x + t4.y;
    base[4 * tid + 3] = b0 + t4.x + t4.y + t4.z;
  }
  __syncthreads();
  K key[RS_PER_THREAD];
  uint32_t val[RS_PER_THREAD], rank[RS_PER_THREAD];
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RS_PER_THREAD; r++) {
    const int i = tile * RS_TILE + wave * (RS_TILE / 4) + r * 64 + lane;
    const bool live = i < n;
    key[r] = 0; val[r] = 0;
    if (live) {
      key[r] = keys_in[i];
      if constexpr (kVals) val[r] = vals_in[i];
    }
    const uint32_t dig = live ? ((uint32_t)(key[r] >> shift) & (RS_BINS - 1)) : RS_BINS;   // bit 10 set: idle lanes match only each other
    unsigned long long m = ~0ull;
#pragma unroll
    for (int bit = 0; bit <= RS_BITS; bit++) {
      const unsigned long long bal = __ballot((dig >> bit) & 1u);
      m &= ((dig >> bit) & 1u) ? bal : ~bal;
    }
    const int leader = __ffsll((long long)m) - 1;
    uint32_t old = 0;
    if (live && lane == leader) { old = cnt[wave][dig]; cnt[wave][dig] = old + (uint32_t)__popcll(m); }
    old = __shfl(old, leader, 64);
    rank[r] = old + (uint32_t)__popcll(m & below);
  }
  __syncthreads();
  const uint32_t* o = offs + (size_t)tile * RS_BINS;
  for (int b = tid; b < RS_BINS; b += 256) {
    uint32_t run = base[b] + o[b];
#pragma unroll
    for (int w = 0; w < 4; w++) { uint32_t c = cnt[w][b]; cnt[w][b] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_PER_THREAD; r++) {
    const int i = tile * RS_TILE + wave * (RS_TILE / 4) + r * 64 + lane;
    if (i < n) {
      const uint32_t dig = (uint32_t)(key[r] >> shift) & (RS_BINS - 1);
      const size_t pos = (size_t)cnt[wave][dig] + rank[r];
      keys_out[pos] = key[r];
      if constexpr (kVals) vals_out[pos] = val[r];
    }
  }
}
// temporary storage: ping-pong keys (+ values), the tile histograms, the digit totals
template <class K>
static size_t gs_temp_bytes(int n, bool vals) {
  const size_t tiles = (size_t)(n + RS_TILE - 1) / RS_TILE + 1;
  size_t b = ((sizeof(K) * (size_t)n + 255) & ~(size_t)255) + (vals ? ((sizeof(uint32_t) * (size_t)n + 255) & ~(size_t)255) : 0);
  return b + sizeof(uint32_t) * RS_BINS * (tiles + 1) + 256;
}
template <class K, bool kVals>
static void gs_sort(void* temp, const K* keys_in, K* keys_out, const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit, hipStream_t s) {
  if (n <= 0) return;
  const int tiles = (n + RS_TILE - 1) / RS_TILE;
  char* p = static_cast<char*>(temp);
  K* ktmp = reinterpret_cast<K*>(p); p += (sizeof(K) * (size_t)n + 255) & ~(size_t)255;
  uint32_t* vtmp = nullptr;
  if (kVals) { vtmp = reinterpret_cast<uint32_t*>(p); p += (sizeof(uint32_t) * (size_t)n + 255) & ~(size_t)255; }
  uint32_t* hist = reinterpret_cast<uint32_t*>(p); p += sizeof(uint32_t) * RS_BINS * (size_t)tiles;
  uint32_t* total = reinterpret_cast<uint32_t*>(p);
  int passes = (end_bit + RS_BITS - 1) / RS_BITS;
  if (passes < 1) passes = 1;
  // the last pass must land in keys_out: with an odd number of passes in -> out -> tmp -> out, with an even one in -> tmp -> out
  const K* kin = keys_in;
  const uint32_t* vin = vals_in;
  for (int ps = 0; ps < passes; ps++) {
    const bool to_out = ((passes - 1 - ps) % 2) == 0;
    K* kout = to_out ? keys_out : ktmp;
    uint32_t* vout = to_out ? vals_out : vtmp;
    hipLaunchKernelGGL(k_gs_hist<K>, dim3(tiles), dim3(256), 0, s, kin, n, ps * RS_BITS, hist);
    hipLaunchKernelGGL(k_gs_colscan, dim3(RS_BINS / 4), dim3(256), 0, s, hist, tiles, total);
    hipLaunchKernelGGL((k_gs_scatter<K, kVals>), dim3(tiles), dim3(256), 0, s, kin, vin, n, ps * RS_BITS, (const uint32_t*)hist, (const uint32_t*)total, kout, vout);
    kin = kout;
    vin = vout;
  }
}
size_t sort_temp_bytes(int n) { return gs_temp_bytes<uint32_t>(n, true); }
void sort_pairs_u32(void* temp, size_t, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit,
                    hipStream_t s) {
  gs_sort<uint32_t, true>(temp, keys_in, keys_out, vals_in, vals_out, n, end_bit, s);
}
size_t sort64_temp_bytes(int n) { return gs_temp_bytes<uint64_t>(n, true); }
void sort_pairs_u64(void* temp, size_t, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit,
                    hipStream_t s) {
  gs_sort<uint64_t, true>(temp, keys_in, keys_out, vals_in, vals_out, n, end_bit, s);
}
size_t sort_keys64_temp_bytes(int n) { return gs_temp_bytes<uint64_t>(n, false); }
void sort_keys_u64(void* temp, size_t, const uint64_t* keys_in, uint64_t* keys_out, int n, hipStream_t s) {
  gs_sort<uint64_t, false>(temp, keys_in, keys_out, nullptr, nullptr, n, 64, s);
}

// ===== inclusive scan of u32 (stream compaction: leaf starts, voxel heads, crop / NaN / map flags) ================================
// tile sums -> one workgroup scans them -> every tile scans itself on top of its offset.  4 096 elements per tile.
constexpr int SC_TILE = 4096, SC_PER_THREAD = SC_TILE / 256;
__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t mine, uint32_t* wsum /*[4] shared*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t b0 = inc - mine;
  for (int w = 0; w < wave; w++) b0 += wsum[w];
  if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return b0;
}
__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t* __restrict__ in, int n, uint32_t* __restrict__ tsum) {
  __shared__ uint32_t wsum[4];
  const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_PER_THREAD;   // a thread owns 16 CONSECUTIVE elements (four 16-byte loads)
  uint32_t s = 0;
  if (base + SC_PER_THREAD <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int k = 0; k < SC_PER_THREAD / 4; k++) { uint4 v = p[k]; s += v.x + v.y + v.z + v.w; }
  } else {
    for (int k = 0; k < SC_PER_THREAD; k++)
      if (base + k < n) s += in[base + k];
  }
  uint32_t tot;
  (void)block_exclusive_256(s, wsum, &tot);
  if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_scan_offsets(uint32_t* __restrict__ tsum, int tiles) {   // exclusive, in place, ONE workgroup
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < tiles; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const uint32_t v = t < tiles ? tsum[t] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_256(v, wsum, &tot);
    const uint32_t carry = carry_s;
    if (t < tiles) tsum[t] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_scan_down(const uint32_t* __restrict__ in, int n, const uint32_t* __restrict__ toff, uint32_t* __restrict__ out) {
  __shared__ uint32_t wsum[4];
  const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_PER_THREAD;
  uint32_t v[SC_PER_THREAD];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SC_PER_THREAD; k++) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
  uint32_t run = toff[blockIdx.x] + block_exclusive_256(s, wsum, nullptr);
#pragma unroll
  for (int k = 0; k < SC_PER_THREAD; k++) {
    run += v[k];
    if (base + k < n) out[base + k] = run;
  }
}
size_t scan_temp_bytes(int n) { return sizeof(uint32_t) * ((size_t)(n + SC_TILE - 1) / SC_TILE + 1); }
void inclusive_scan_u32(void* temp, size_t, const uint32_t* in, uint32_t* out, int n, hipStream_t s) {
  if (n <= 0) return;
  const int tiles = (n + SC_TILE - 1) / SC_TILE;
  uint32_t* tsum = static_cast<uint32_t*>(temp);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(256), 0, s, in, n, tsum);
  hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(256), 0, s, tsum, tiles);
  hipLaunchKernelGGL(k_scan_down, dim3(tiles), dim3(256), 0, s, in, n, (const uint32_t*)tsum, out);
}

int segsort_tiles(int n) { return (n + RS_TILE - 1) / RS_TILE; }   // IndexDesc::tile0 = sum of the previous clouds' tiles
size_t segsort_hist_elems(long total_points, int n_clouds) { return (size_t)(total_points / RS_TILE + n_clouds) * RS_BINS; }

// kv_a: (30-bit key, point index) pairs of all clouds, concatenated in cloud order (descs[c].offset / .n), written by k_key_b;
// kv_b: scratch of the same size.  On return keys_out (cloud << 32 | key) / vals_out hold every cloud's segment sorted by key, ties
// in ascending input position.  Every pass moves 8-byte pairs.
void segsort_pairs(const IndexDesc* descs, int n_clouds, int max_n, uint64_t* kv_a, uint64_t* kv_b, uint64_t* keys_out, uint32_t* vals_out, uint32_t* hist,
                   hipStream_t s) {
  const int tiles = (max_n + RS_TILE - 1) / RS_TILE;
  const dim3 grid(tiles, n_clouds), blk(256);
  hipLaunchKernelGGL(k_rs_hist, grid, blk, 0, s, descs, (const void*)kv_a, 0, hist);
  hipLaunchKernelGGL(k_rs_scan, dim3(n_clouds), dim3(256), 0, s, descs, hist);
  hipLaunchKernelGGL((k_rs_scatter<false>), grid, blk, 0, s, descs, (const void*)kv_a, 0, (const uint32_t*)hist, (void*)kv_b,
                     (uint32_t*)nullptr);
  hipLaunchKernelGGL(k_rs_hist, grid, blk, 0, s, descs, (const void*)kv_b, RS_BITS, hist);
  hipLaunchKernelGGL(k_rs_scan, dim3(n_clouds), dim3(256), 0, s, descs, hist);
  hipLaunchKernelGGL((k_rs_scatter<false>), grid, blk, 0, s, descs, (const void*)kv_b, RS_BITS, (const uint32_t*)hist, (void*)kv_a,
                     (uint32_t*)nullptr);
  // last pass: pairs in, (cloud << 32 | key) and index out
  hipLaunchKernelGGL(k_rs_hist, grid, blk, 0, s, descs, (const void*)kv_a, 2 * RS_BITS, hist);
  hipLaunchKernelGGL(k_rs_scan, dim3(n_clouds), dim3(256), 0, s, descs, hist);
  hipLaunchKernelGGL((k_rs_scatter<true>), grid, blk, 0, s, descs, (const void*)kv_a, 2 * RS_BITS, (const uint32_t*)hist, (void*)keys_out,
                     vals_out);
}

}  // namespace lh
