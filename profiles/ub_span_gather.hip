// microbenchmark: k_confirm's access pattern by itself -- every lane fetches its own 32-byte span (two dwordx4)
// at a random dword-aligned offset of a table the size of cfg5's packed database (2.5 GB) or larger; R spans in
// flight per lane, eight waves per SIMD, nothing else in the kernel.  What rate of spans (and of 128-byte lines:
// a span straddles a line boundary 28 times in 128) does the part deliver for that pattern?
//   mode 0: dword-aligned random start (k_confirm)        mode 1: line-aligned start (no straddle)
//   mode 2: the same spans, eight lanes x 16 B fetch the whole line the span starts in (k_match_t's shape;
//           a straddling span's second line is not fetched: rate of the shape, not a usable kernel)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_span profiles/ub_span_gather.hip && /tmp/ub_span
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int R, int MODE>
__global__ __launch_bounds__(256, 8) void k_spans(const uint32_t* __restrict__ T, uint64_t nwords, uint64_t per_lane, uint32_t* out) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  if (MODE == 2) {
    const uint64_t oct = gid >> 3;
    const uint32_t part = threadIdx.x & 7;
    for (uint64_t i = 0; i < per_lane * 8; i += R) {
      u32x4_v a[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        const uint64_t w = (mix64(oct * 0x9E3779B97F4A7C15ull + i + r) % (nwords - 64)) & ~31ull;
        a[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(T + w) + part);
      }
#pragma unroll
      for (int r = 0; r < R; r++) acc ^= a[r].x ^ a[r].w;
    }
  } else {
    for (uint64_t i = 0; i < per_lane; i += R) {
      u32x4_u a[R], b[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        uint64_t w = mix64(gid * 0x9E3779B97F4A7C15ull + i + r) % (nwords - 64);
        if (MODE == 1) w &= ~31ull;
        a[r] = *reinterpret_cast<const u32x4_u*>(T + w);
        b[r] = *reinterpret_cast<const u32x4_u*>(T + w + 4);
      }
#pragma unroll
      for (int r = 0; r < R; r++) acc ^= a[r].x ^ a[r].w ^ b[r].y ^ b[r].z;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int R, int MODE>
static void run(const uint32_t* T, uint64_t nwords, uint32_t* out, int wgs, const char* what) {
  const uint64_t total = 30000000ull;  // one cfg5-shard k_confirm launch: 30.1 M descriptors
  const uint64_t lanes = (uint64_t)wgs * 256;
  const uint64_t per = ((total / lanes) / R) * R;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_spans<R, MODE>), dim3(wgs), dim3(256), 0, 0, T, nwords, per, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double spans = (double)per * lanes;
  const double lines = MODE == 0 ? spans * (1.0 + 28.0 / 128.0) : spans;
  printf("table %.1f GB  %-34s wgs %5d  R %d: %.3f ms for %.1f M spans = %.1f G spans/s = %.1f G lines/s\n", nwords * 4 / 1e9, what, wgs, R,
         best, spans / 1e6, spans / best / 1e6, lines / best / 1e6);
}

int main(int argc, char** argv) {
  uint32_t* out; hipMalloc((void**)&out, 64);
  for (double gb : {2.5, 64.0}) {
    const uint64_t nwords = (uint64_t)(gb * 1e9 / 4);
    uint32_t* T;
    if (hipMalloc((void**)&T, nwords * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(T, 1, nwords * 4);
    for (int wgs : {2048, 8192}) {
      run<1, 0>(T, nwords, out, wgs, "own span, dword-aligned start");
      run<2, 0>(T, nwords, out, wgs, "own span, dword-aligned start");
      run<4, 0>(T, nwords, out, wgs, "own span, dword-aligned start");
      run<2, 1>(T, nwords, out, wgs, "own span, line-aligned start");
      run<4, 1>(T, nwords, out, wgs, "own span, line-aligned start");
      run<4, 2>(T, nwords, out, wgs, "eight lanes x 16 B per line");
      run<8, 2>(T, nwords, out, wgs, "eight lanes x 16 B per line");
    }
    hipFree(T);
  }
  return 0;
}
