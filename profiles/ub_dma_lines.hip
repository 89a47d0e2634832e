// microbenchmark for the round-4 kernel design: random whole-line (128 B) reads that go STRAIGHT TO LDS
// (global_load_lds_dwordx4: no register ring, no ds_write pass), consumed the way k_match_t consumes them
// (lane p reads line p with ds_read_b128), with a dummy compute load per window, at 2 / 3 / 4 waves per SIMD.
// Questions it answers before the kernel is written:
//   1. does LDS-DMA keep the random-line rate of the register path (46.8 G lines/s)?
//   2. 8 lanes x 16 B per line (one instruction = 8 whole lines) against a quad per half line
//   3. what rate survives a compute load of `work` dependent VALU rounds per window when the wave WAITS for
//      its lines (single line buffer: flights exposed) and when the next window is in flight meanwhile
//      (two line buffers), as a function of the waves per SIMD
//   4. are 64-byte probes (half lines) cheaper than 128-byte ones (request- or byte-bound?)
//   5. semantics: LDS destination = M0 base + lane * 16; sources that are only 8-byte aligned (40-byte
//      overflow entries); the 12-byte form
// build: hipcc --offload-arch=gfx950 -O3 -o ub_dma profiles/ub_dma_lines.hip ; run: ./ub_dma [bits]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// one LDS-DMA instruction: every lane's 16 bytes at gsrc land at lds_dst + 16 * lane (lds_dst wave-uniform)
template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds12(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// MODE 0: registers (quad per line, 32 B per lane, nt) + swizzled ds_write (k_match_t's arrival)
// MODE 1: LDS-DMA, 8 lanes per line (an instruction = 8 whole lines)
// MODE 2: LDS-DMA, a quad per half line (an instruction = 16 half lines)
// MODE 3: LDS-DMA, 64-byte probes only (a quad per probe, 4 instructions per 64 probes)
// DB: two line buffers -- window i + 1 is in flight while window i is consumed
template <int MODE, bool DB, bool NT>
__global__ __launch_bounds__(256) void k_dma(const uint4* __restrict__ T, uint64_t nb_mask, uint32_t iters, uint32_t work, uint32_t* out) {
  extern __shared__ uint4 s_dyn[];  // per wave: (DB ? 2 : 1) x 512 uint4; the rest bounds the workgroups per CU
  const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr uint32_t WBUF = (MODE == 3 ? 256u : 512u) * ((DB && MODE != 0) ? 2u : 1u);  // uint4 per wave
  uint4* const buf = s_dyn + wid * WBUF;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)buf);
  const uint64_t gw = (uint64_t)blockIdx.x * 4 + wid;
  uint32_t acc = 0;
  uint4 ra[4], rb[4];
  auto bucket = [&](uint32_t it, uint32_t p) -> uint64_t { return mix64((gw * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)it * 64 + p)) & nb_mask; };
  auto issue = [&](uint32_t it, uint32_t half) __attribute__((always_inline)) {
    const uint32_t lds = lds0 + half * (MODE == 3 ? 4096 : 8192);
    if (MODE == 0) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const u32x4_v* p = reinterpret_cast<const u32x4_v*>(T + bucket(it, rr * 16 + (lane >> 2)) * 8) + 2 * (lane & 3);
        const u32x4_v x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1);
        ra[rr] = make_uint4(x.x, x.y, x.z, x.w);
        rb[rr] = make_uint4(y.x, y.y, y.z, y.w);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t p = 8 * i + (lane >> 3), c = (lane & 7) ^ ((p >> 1) & 7);
        glds16<NT>(T + bucket(it, p) * 8 + c, lds + i * 1024);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        // instruction i: half (i & 1) of lines 16 * (i >> 1) .. + 15; chunk slot within the line swizzled as MODE 1
        const uint32_t p = 16 * (i >> 1) + (lane >> 2), cs = 4 * (i & 1) + (lane & 3), c = cs ^ ((p >> 1) & 7);
        // lands at lds + i * 1024 + 16 * lane: the reader's address for (p, chunk c) is computed the same way
        glds16<NT>(T + bucket(it, p) * 8 + c, lds + i * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t p = 16 * i + (lane >> 2), c = (lane & 3);
        glds16<NT>(T + bucket(it, p) * 8 + c, lds + i * 1024);
      }
    }
  };
  auto arrive = [&]() __attribute__((always_inline)) {
    uint4* const line = buf;
    if (MODE == 0) {
      const uint32_t q = lane >> 2, part = lane & 3;
      const uint32_t sw = ((q >> 1) & 7u) ^ (q & 1u);
      const uint32_t wb0 = q * 8u + ((2u * part) ^ sw);
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        line[rr * 128 + wb0] = ra[rr];
        line[rr * 128 + (wb0 ^ 1u)] = rb[rr];
      }
    }
  };
  auto consume = [&](uint32_t half) __attribute__((always_inline)) {
    uint4* const line = buf + half * (MODE == 3 ? 256 : 512);
    if (MODE == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t rbase = lane * 8u + (((lane >> 1) & 7u) ^ (lane & 1u));
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const uint4 v = line[rbase ^ (uint32_t)c];
        acc ^= v.x + v.y + v.z + v.w;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const uint4 v = line[lane * 8u + ((uint32_t)c ^ ((lane >> 1) & 7u))];
        acc ^= v.x + v.y + v.z + v.w;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const uint32_t cs = (uint32_t)c ^ ((lane >> 1) & 7u);  // slot of chunk c within line `lane`
        const uint32_t i = 2 * (lane >> 4) + (cs >> 2);
        const uint4 v = line[i * 64u + 4u * (lane & 15u) + (cs & 3u)];
        acc ^= v.x + v.y + v.z + v.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint4 v = line[(lane >> 4) * 64u + 4u * (lane & 15u) + (uint32_t)c];
        acc ^= v.x + v.y + v.z + v.w;
      }
    }
    // the compute load: `work` rounds of 8 dependent vector instructions
    for (uint32_t w = 0; w < work; w++) {
      acc = acc * 1664525u + 1013904223u;
      acc ^= acc >> 13;
      acc = acc * 22695477u + 1u;
      acc ^= acc << 7;
      acc += __popc(acc);
      acc ^= acc >> 17;
    }
  };
  if (DB) {
    issue(0, 0);
    for (uint32_t it = 0; it < iters; it++) {
      if (MODE != 0) {
        // the next window goes to the other buffer BEFORE this one is consumed; the wait below must then leave
        // the 8 (4) younger instructions in flight
        if (it + 1 < iters) {
          issue(it + 1, (it + 1) & 1);
          if (MODE == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else wait_vm0();
        consume(it & 1);
      } else {
        arrive();  // (registers: the arrival writes LDS, then the ring is refilled, then the compute -- k_match_t's order)
        if (it + 1 < iters) issue(it + 1, 0);
        consume(0);
      }
    }
  } else {
    for (uint32_t it = 0; it < iters; it++) {
      issue(it, 0);
      if (MODE == 0) arrive();
      else wait_vm0();
      consume(0);
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// semantics check: line i of the table holds words 32 * i + j; MODE 1 / 2 layouts read back by lane p
template <int MODE>
__global__ void k_check(const uint4* __restrict__ T, uint32_t nlines, uint32_t* bad) {
  __shared__ uint4 line[512];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)line);
  auto bkt = [&](uint32_t p) { return (p * 2654435761u) % nlines; };
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (MODE == 1) {
      const uint32_t p = 8 * i + (lane >> 3), c = (lane & 7) ^ ((p >> 1) & 7);
      glds16<true>(T + (uint64_t)bkt(p) * 8 + c, lds + i * 1024);
    } else {
      const uint32_t p = 16 * (i >> 1) + (lane >> 2), cs = 4 * (i & 1) + (lane & 3), c = cs ^ ((p >> 1) & 7);
      glds16<true>(T + (uint64_t)bkt(p) * 8 + c, lds + i * 1024);
    }
  }
  wait_vm0();
  uint32_t nb = 0;
  for (int c = 0; c < 8; c++) {
    uint4 v;
    if (MODE == 1) v = line[lane * 8u + ((uint32_t)c ^ ((lane >> 1) & 7u))];
    else {
      const uint32_t cs = (uint32_t)c ^ ((lane >> 1) & 7u);
      v = line[(2 * (lane >> 4) + (cs >> 2)) * 64u + 4u * (lane & 15u) + (cs & 3u)];
    }
    const uint32_t w0 = bkt(lane) * 32u + 4u * c;
    nb += (v.x != w0) + (v.y != w0 + 1) + (v.z != w0 + 2) + (v.w != w0 + 3);
  }
  if (nb) atomicAdd(bad, nb);
}
// 40-byte entries at word offsets 0 / 10 / 20 of a line (8-byte aligned sources), fetched as 16 + 12 + 12 bytes
// (r04 result: WRONG -- the 12-byte form does not land at lane * 12) and as three 16-byte pieces (48 bytes: the tail
// belongs to the entry's line), lanes 0..47 only with the planes 768 bytes apart (inactive lanes write nothing)
__global__ void k_check_entries(const uint32_t* __restrict__ E, uint32_t nent, uint32_t* bad) {
  __shared__ uint4 a16[64];
  __shared__ uint32_t b12[64 * 3], c12[64 * 3];
  __shared__ uint4 z[3 * 48];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t e = (lane * 2654435761u) % nent;
  const uint32_t* pe = E + (e / 3) * 32 + (e % 3) * 10;
  glds16<false>(pe, __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)a16));
  glds12(pe + 4, __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)b12));
  glds12(pe + 7, __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)c12));
  const uint32_t zb = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)z);
  if (lane < 48) {
    glds16<false>(pe, zb);
    glds16<false>(pe + 4, zb + 768);
    glds16<false>(pe + 8, zb + 1536);
  }
  wait_vm0();
  const uint32_t w0 = (e / 3) * 32 + (e % 3) * 10;
  const uint4 v = a16[lane];
  uint32_t nb = (v.x != w0) + (v.y != w0 + 1) + (v.z != w0 + 2) + (v.w != w0 + 3);
  for (int j = 0; j < 3; j++) nb += (b12[lane * 3 + j] != w0 + 4 + j) + (c12[lane * 3 + j] != w0 + 7 + j);
  if (nb) atomicAdd(bad, nb);
  if (lane < 48) {
    const uint4 q0 = z[lane], q1 = z[48 + lane], q2 = z[96 + lane];
    const uint32_t nz = (q0.x != w0) + (q0.y != w0 + 1) + (q0.z != w0 + 2) + (q0.w != w0 + 3) + (q1.x != w0 + 4) + (q1.y != w0 + 5) +
                        (q1.z != w0 + 6) + (q1.w != w0 + 7) + (q2.x != w0 + 8) + (q2.y != w0 + 9);
    if (nz) atomicAdd(bad + 1, nz);
  }
}

template <int MODE, bool DB, bool NT>
static void run(const char* name, const uint4* T, uint64_t nb, uint32_t* out, int wg_per_cu, uint32_t work) {
  const size_t per_wg_min = (size_t)4 * (MODE == 3 ? 4096 : 8192) * ((DB && MODE != 0) ? 2 : 1);
  size_t lds = (size_t)(160 * 1024 / wg_per_cu) & ~(size_t)2047;  // bounds the resident workgroups: floor(160 KB / lds)
  if ((size_t)(160 * 1024) / lds != (size_t)wg_per_cu) lds -= 2048;
  if (lds < per_wg_min) { printf("%-34s %d workgroups per CU do not fit\n", name, wg_per_cu); return; }
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<MODE, DB, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int wgs = 256 * wg_per_cu;
  const uint64_t total = 30000000ull;
  const uint32_t iters = (uint32_t)(total / ((uint64_t)wgs * 4 * 64));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_dma<MODE, DB, NT>), dim3(wgs), dim3(256), lds, 0, T, nb - 1, iters, work, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double probes = (double)iters * wgs * 4 * 64;
  printf("%-34s %d waves/SIMD  work %3u: %.3f ms for %.1f M probes = %.1f G probes/s (%.2f TB/s at %d B)\n", name, wg_per_cu, work, best,
         probes / 1e6, probes / best / 1e6, probes * (MODE == 3 ? 64.0 : 128.0) / best / 1e9, MODE == 3 ? 64 : 128);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int bits = argc > 1 ? atoi(argv[1]) : 30;
  const uint64_t nb = 1ull << bits;
  uint4* T; uint32_t* out;
  if (hipMalloc((void**)&T, nb * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc((void**)&out, 64);
  hipMemset(out, 0, 64);
  {  // semantics
    const uint32_t nl = 4096;
    std::vector<uint32_t> h(nl * 32);
    for (uint32_t i = 0; i < nl * 32; i++) h[i] = i;
    hipMemcpy(T, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    uint32_t bad[4] = {0, 0, 0, 0};
    hipLaunchKernelGGL(k_check<1>, dim3(1), dim3(64), 0, 0, T, nl, out);
    hipLaunchKernelGGL(k_check<2>, dim3(1), dim3(64), 0, 0, T, nl, out + 1);
    hipLaunchKernelGGL(k_check_entries, dim3(1), dim3(64), 0, 0, reinterpret_cast<const uint32_t*>(T), nl * 3, out + 2);
    hipError_t err = hipDeviceSynchronize();
    hipMemcpy(bad, out, 16, hipMemcpyDeviceToHost);
    printf("semantics: %s; wrong words: 8 lanes per line %u, quad per half line %u, 40-byte entries as 16 + 12 + 12 from 8-byte aligned sources %u, "
           "as 3 x 16 bytes (lanes 0..47, planes 768 bytes apart) %u\n", hipGetErrorString(err), bad[0], bad[1], bad[2], bad[3]);
    if (err != hipSuccess) return 1;
    if (argc > 2) return 0;  // ./ub_dma bits check: the semantics only
  }
  hipMemset(T, 1, nb * 128);
  hipDeviceSynchronize();
  for (int w : {2, 3, 4}) {
    for (uint32_t work : {0u, 100u, 200u}) {
      run<0, true, true>("registers + ds_write (k_match_t)", T, nb, out, w, work);
      run<1, false, true>("LDS-DMA 8 lanes/line, 1 buffer", T, nb, out, w, work);
      run<2, false, true>("LDS-DMA quad/half line, 1 buffer", T, nb, out, w, work);
      if (w <= 2) run<1, true, true>("LDS-DMA 8 lanes/line, 2 buffers", T, nb, out, w, work);
    }
  }
  for (int w : {3, 4}) {
    run<1, false, false>("LDS-DMA 8 lanes/line, 1 buf, no nt", T, nb, out, w, 100);
    run<3, false, true>("LDS-DMA 64-byte probes, 1 buffer", T, nb, out, w, 0);
    run<3, false, true>("LDS-DMA 64-byte probes, 1 buffer", T, nb, out, w, 100);
  }
  for (int w : {5, 6, 8}) {
    run<0, true, true>("registers + ds_write (k_match_t)", T, nb, out, w, 0);
    run<3, false, true>("LDS-DMA 64-byte probes, 1 buffer", T, nb, out, w, 0);
  }
  return 0;
}
