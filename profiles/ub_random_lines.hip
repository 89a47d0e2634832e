// microbenchmark: random whole-line (128 B) reads from a large table, a quad of lanes per line
// (32 B per lane, two non-temporal dwordx4), R lines in flight per quad.  What rate does the GPU
// reach for k_match's access pattern with nothing else in the kernel?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int R>
__global__ __launch_bounds__(256) void k_lines(const uint4* __restrict__ T, uint64_t nb_mask, uint64_t lines_per_quad, uint32_t* out) {
  extern __shared__ uint32_t s_occ[];  // (dynamic LDS only bounds the workgroups per CU: `./ub bits lds_kb`)
  if (lines_per_quad == 0xFFFFFFFFFFFFFFFFull) s_occ[threadIdx.x] = 1;
  const uint64_t quad = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
  const uint32_t part = threadIdx.x & 3;
  uint32_t acc = 0;
  for (uint64_t i = 0; i < lines_per_quad; i += R) {
    u32x4_v a[R], b[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint64_t bkt = mix64(quad * 0x9E3779B97F4A7C15ull + i + r) & nb_mask;
      const u32x4_v* p = reinterpret_cast<const u32x4_v*>(T + bkt * 8) + 2 * part;
      a[r] = __builtin_nontemporal_load(p);
      b[r] = __builtin_nontemporal_load(p + 1);
    }
#pragma unroll
    for (int r = 0; r < R; r++) acc ^= a[r].x ^ a[r].w ^ b[r].y ^ b[r].z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv) {
  const int bits = argc > 1 ? atoi(argv[1]) : 30;  // buckets = 2^bits lines of 128 B
  const uint64_t nb = 1ull << bits;
  uint4* T; uint32_t* out;
  if (hipMalloc((void**)&T, nb * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc((void**)&out, 64);
  hipMemset(T, 1, nb * 128);
  const uint64_t total_lines = 30000000ull;
  // ./ub bits lds_kb: workgroups of four waves with lds_kb KB of LDS each -- 72 KB = two per CU = two waves per
  // SIMD, k_match_t's occupancy -- and R = 4 / 8 / 16 lines per quad = 64 / 128 / 256 lines (1 / 2 / 4 of
  // k_match_t's windows) in flight per wave: what the part delivers at that concurrency with no compute at all
  const size_t lds = argc > 2 ? (size_t)atoi(argv[2]) * 1024 : 0;
  if (lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lines<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lines<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lines<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int wgs : {512, 2048}) {
      for (int R : {4, 8, 16}) {
        const uint64_t quads = (uint64_t)wgs * 64;
        const uint64_t lpq = (total_lines / quads / R) * R;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(e0);
          if (R == 4) hipLaunchKernelGGL(k_lines<4>, dim3(wgs), dim3(256), lds, 0, T, nb - 1, lpq, out);
          if (R == 8) hipLaunchKernelGGL(k_lines<8>, dim3(wgs), dim3(256), lds, 0, T, nb - 1, lpq, out);
          if (R == 16) hipLaunchKernelGGL(k_lines<16>, dim3(wgs), dim3(256), lds, 0, T, nb - 1, lpq, out);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep == 2) printf("table 2^%d lines  %zu KB of LDS per workgroup (%d workgroups of 4 waves per CU)  wgs %d  %d lines in flight per wave: %.3f ms for %.1f M lines = %.1f G lines/s = %.2f TB/s\n", bits, lds / 1024, (int)(160 * 1024 / lds), wgs, 16 * R, ms, lpq * quads / 1e6, lpq * quads / ms / 1e6, lpq * quads * 128.0 / ms / 1e9);
        }
      }
    }
    return 0;
  }
  for (int wgs : {1024, 2048, 4096}) {
    for (int R : {2, 4, 8}) {
      const uint64_t quads = (uint64_t)wgs * 64;
      const uint64_t lpq = (total_lines / quads / R) * R;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (R == 2) hipLaunchKernelGGL(k_lines<2>, dim3(wgs), dim3(256), 0, 0, T, nb - 1, lpq, out);
        if (R == 4) hipLaunchKernelGGL(k_lines<4>, dim3(wgs), dim3(256), 0, 0, T, nb - 1, lpq, out);
        if (R == 8) hipLaunchKernelGGL(k_lines<8>, dim3(wgs), dim3(256), 0, 0, T, nb - 1, lpq, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("table 2^%d lines (%.0f GiB)  wgs %d  R %d: %.3f ms for %.1f M lines = %.1f G lines/s = %.2f TB/s\n", bits, nb * 128.0 / (1ull << 30), wgs, R, ms, lpq * quads / 1e6, lpq * quads / ms / 1e6, lpq * quads * 128.0 / ms / 1e9);
      }
    }
  }
  return 0;
}
