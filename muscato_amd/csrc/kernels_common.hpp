// kernels_common.hpp -- device helpers, packing kernels, scans
// Part of libmuscato_hip.so: included by muscato_hip.hip (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------

#define DEV __device__ __forceinline__
// Kernels that are not templates: defined once, in muscato_hip.hip.  The translation units that hold
// only k_match_t (match_lane_rw*.hip) include the same headers with MUSC_KERNEL = a static kernel
// nobody launches, which the compiler then drops.
#ifndef MUSC_KERNEL
#define MUSC_KERNEL __global__
#endif
#define READ_HAS_X 0x10000u  // length word of a read record, bit 16: the read holds an X

DEV uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// Hash of a (window, bucket) pair for the MaxMatches accounting (sketch cells and the exact block
// table take its top bits): 32-bit multiply / xor-shift rounds -- full-rate instructions, where a
// 64-bit mix costs a dozen quarter-rate multiplies per pair.  Every kernel that counts, and
// k_hot_probes which names the suspects afterwards, must use the same function.
DEV uint32_t block_hash32(uint32_t k, uint32_t b) {
  uint32_t h = b ^ (k * 0x9E3779B9u + 0x7F4A7C15u);
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// 64 bits of a little-endian bit stream held in u32 words, starting at bit `bo`.
DEV uint64_t ext64(const uint32_t* __restrict__ w, uint64_t bo) {
  const uint64_t i = bo >> 5;
  const uint32_t sh = (uint32_t)bo & 31u;
  const uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
  if (sh == 0) return lo;
  return (lo >> sh) | ((uint64_t)w[i + 2] << (64 - sh));
}

DEV uint64_t lowmask64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

// Index bucket of the ww-base window starting at bit `bo` of plane w (mask plane m or null).
// Identity when the key fits the table and holds no X (exact, no false candidates);
// otherwise a 64-bit mix.  Any deterministic function is correct: k_confirm re-verifies
// the window bases, so collisions only cost extra pairs.
DEV uint32_t bucket_of(const uint32_t* __restrict__ w, const uint32_t* __restrict__ m, uint64_t bo,
                       int ww, int bits, int direct) {
  const int nb = 2 * ww;
  uint64_t h = 0, anymask = 0, key0 = 0;
  for (int c = 0; c < nb; c += 64) {
    const int take = nb - c < 64 ? nb - c : 64;
    const uint64_t key = ext64(w, bo + c) & lowmask64(take);
    const uint64_t mk = m ? (ext64(m, bo + c) & lowmask64(take)) : 0ull;
    if (c == 0) key0 = key;
    anymask |= mk;
    h = mix64(h ^ key ^ mix64(mk + 0x9E3779B97F4A7C15ull * (uint64_t)(c + 1)));
  }
  // direct mode: first base in the most significant bits, so that bytewise-sorted reads
  // (the order of reads_sorted.txt.sz) walk the table and the entry lists front to back
  if (direct && anymask == 0) return (uint32_t)(__brevll(key0) >> (64 - nb));
  return (uint32_t)(h >> (64 - bits));
}

// Sum `v` over the block and add it to *dst with ONE atomic (single-address atomics
// serialise at ~90 M/s on MI355X, so per-wave atomics from a big grid cost milliseconds).
DEV void block_add_u64(unsigned long long v, unsigned long long* dst) {
  __shared__ unsigned long long s_acc[16];
  for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
  const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) s_acc[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < nw; w++) t += s_acc[w];
    if (t) atomicAdd(dst, t);
  }
  __syncthreads();
}

#ifndef MAX_GRID
#define MAX_GRID 4096u  // grid-stride kernels: enough blocks to fill 256 CUs several times
#endif

// ------------------------------------------------------------------------------------
// packing kernels (ASCII / 2-bit stream -> device layout)
// ------------------------------------------------------------------------------------

DEV uint32_t ascii_code(unsigned char c, uint32_t* isx) {
  const uint32_t a = c == 'A', cc = c == 'C', g = c == 'G', t = c == 'T';
  *isx = !(a | cc | g | t);
  return cc | (g << 1) | (t * 3u);
}

// one thread per u32 word (16 bases) of the database stream
MUSC_KERNEL void k_pack_db_ascii(const unsigned char* __restrict__ s, uint64_t nbases,
                                uint32_t* __restrict__ db2, uint32_t* __restrict__ dbm2,
                                uint64_t nwords, uint32_t* __restrict__ has_x) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint32_t v = 0, mv = 0;
  const uint64_t b0 = w * 16;
  for (int j = 0; j < 16; j++) {
    const uint64_t b = b0 + j;
    if (b < nbases) {
      uint32_t isx;
      const uint32_t c = ascii_code(s[b], &isx);
      v |= c << (2 * j);
      mv |= isx << (2 * j);
    }
  }
  db2[w] = v;
  dbm2[w] = mv;
  if (mv) atomicOr(has_x, 1u);
}

// 2-bit stream + optional 1-bit mask (ABI packed form) -> internal planes
MUSC_KERNEL void k_pack_db_packed(const uint32_t* __restrict__ in2, const uint16_t* __restrict__ inm,
                                 uint32_t* __restrict__ db2, uint32_t* __restrict__ dbm2,
                                 uint64_t nwords, uint32_t* __restrict__ has_x) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint32_t mv = 0;
  if (inm) {
    const uint32_t m16 = inm[w];
    for (int j = 0; j < 16; j++) mv |= ((m16 >> j) & 1u) << (2 * j);
  }
  db2[w] = in2[w] & ~(mv | (mv << 1));
  dbm2[w] = mv;
  if (mv) atomicOr(has_x, 1u);
}

// dbx: one bit per block of 64 database bases (4 words of the mask plane), set when the block
// holds an X.  2 MB per Gbp: stays in L2, so k_confirm gathers the mask plane only for the few
// spans that need it.
MUSC_KERNEL void k_db_xblocks(const uint32_t* __restrict__ dbm2, uint64_t nwords, uint32_t* __restrict__ dbx) {
  const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blk * 4 >= nwords) return;
  uint32_t any = 0;
  for (int q = 0; q < 4; q++)
    if (blk * 4 + q < nwords) any |= dbm2[blk * 4 + q];
  if (any) atomicOr(&dbx[blk >> 5], 1u << (blk & 31));
}

DEV bool db_span_has_x(const uint32_t* __restrict__ dbx, uint64_t gpos, uint32_t nbases) {
  const uint64_t b0 = gpos >> 6, b1 = (gpos + nbases - 1) >> 6;  // at most 64 blocks apart for the spans used
  const uint64_t w = (uint64_t)dbx[b0 >> 5] | ((uint64_t)dbx[(b0 >> 5) + 1] << 32);
  const uint32_t nb = (uint32_t)(b1 - b0) + 1;
  return ((w >> (b0 & 31)) & (nb >= 32 ? 0xFFFFFFFFull : ((1ull << nb) - 1ull))) != 0;
}

MUSC_KERNEL void k_max_len(const uint64_t* __restrict__ off, uint64_t n, unsigned long long* out) {
  __shared__ unsigned long long s_m[16];
  unsigned long long l = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long v = off[i + 1] - off[i];
    l = v > l ? v : l;
  }
  for (int d = 32; d; d >>= 1) {
    const unsigned long long o = __shfl_xor(l, d);
    l = o > l ? o : l;
  }
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned w = 1; w < (blockDim.x + 63) / 64; w++) l = s_m[w] > l ? s_m[w] : l;
    if (l) atomicMax(out, l);
  }
}

// one thread per (read, record word).  Record = rw u32 words: bases in words 0..rw-2
// (2 bits each, zero filled past the read), word rw-1 = len | valid_windows << 16.
template <bool PACKED>
__global__ void k_pack_reads(const unsigned char* __restrict__ s, const uint32_t* __restrict__ in2,
                             const uint32_t* __restrict__ inm, const uint64_t* __restrict__ off,
                             uint64_t nreads, int rw, uint32_t* __restrict__ rd,
                             uint32_t* __restrict__ rdm, uint32_t* __restrict__ has_x) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r = t / rw;
  const int j = (int)(t % rw);
  if (r >= nreads) return;
  const uint64_t o = off[r];
  const uint32_t len = (uint32_t)(off[r + 1] - o);
  if (j == rw - 1) {
    // length word: bits 0-15 the length, bit 16 (READ_HAS_X) set when the read holds an X --
    // the kernels fetch a read's mask words only then
    uint32_t anyx = 0;
    for (uint32_t q = 0; q < len; q++) {
      if (PACKED) {
        const uint64_t g = o + q;
        anyx |= inm ? ((inm[g >> 5] >> (g & 31)) & 1u) : 0u;
      } else {
        uint32_t isx;
        (void)ascii_code(s[o + q], &isx);
        anyx |= isx;
      }
    }
    rd[t] = (len & 0xFFFFu) | (anyx ? READ_HAS_X : 0u);
    if (rdm) rdm[t] = 0;
    return;
  }
  uint32_t v = 0, mv = 0;
  for (int b = 0; b < 16; b++) {
    const uint32_t q = (uint32_t)j * 16 + b;
    if (q < len) {
      uint32_t c, isx;
      if (PACKED) {
        const uint64_t g = o + q;
        c = (in2[g >> 4] >> ((g & 15) * 2)) & 3u;
        isx = inm ? ((inm[g >> 5] >> (g & 31)) & 1u) : 0u;
        if (isx) c = 0;
      } else {
        c = ascii_code(s[o + q], &isx);
      }
      v |= c << (2 * b);
      mv |= isx << (2 * b);
    }
  }
  rd[t] = v;
  if (rdm) rdm[t] = mv;  // (no plane: the caller knows there is no X -- packed input without a mask)
  if (mv) atomicOr(has_x, 1u);
}

// reads of one length L, back to back in a 2-bit stream (no X): one thread per (read, record word);
// reads [first, first + n) of the stream -> their records
MUSC_KERNEL void k_pack_reads_fixed(const uint32_t* __restrict__ in2, uint64_t first, uint64_t n, uint32_t L, int rw,
                                    uint32_t* __restrict__ rd) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (uint64_t)rw) return;
  const uint64_t r = first + t / (uint64_t)rw;
  const uint32_t j = (uint32_t)(t % (uint64_t)rw);
  uint32_t v = L;  // the length word
  if (j + 1 < (uint32_t)rw) {
    const uint32_t b0 = 16u * j;  // first base of this word within the read
    v = 0;
    if (b0 < L) {
      const uint32_t nb = L - b0 < 16u ? L - b0 : 16u;
      v = (uint32_t)ext64(in2, 2ull * (r * (uint64_t)L + b0)) & (nb == 16u ? 0xFFFFFFFFu : ((1u << (2u * nb)) - 1u));
    }
  }
  rd[r * (uint64_t)rw + j] = v;
}

// u32 lengths -> u64 (the input of the offset scan)
MUSC_KERNEL void k_widen_u32(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
  else if (i == n) out[i] = 0;
}

// offsets of reads of one length
MUSC_KERNEL void k_iota_mul(uint64_t* __restrict__ out, uint64_t n, uint64_t step) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) out[i] = i * step;
}

// ------------------------------------------------------------------------------------
// exclusive / inclusive scan (u32), 2048 items per 256-thread block
// ------------------------------------------------------------------------------------

#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

// 8 consecutive elements per thread, moved as two 16-byte accesses when the whole group is
// in range (all pointers handed to the scan are 16-byte aligned)
DEV void scan_load8(const uint32_t* __restrict__ in, uint64_t base, uint64_t n, uint32_t (&v)[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n) {
    const uint4 a = *reinterpret_cast<const uint4*>(in + base);
    const uint4 b = *reinterpret_cast<const uint4*>(in + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (base + i < n) ? in[base + i] : 0u;
  }
}

DEV void scan_store8(uint32_t* __restrict__ out, uint64_t base, uint64_t n, const uint32_t (&v)[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n) {
    *reinterpret_cast<uint4*>(out + base) = make_uint4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<uint4*>(out + base + 4) = make_uint4(v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
      if (base + i < n) out[base + i] = v[i];
  }
}

template <bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_block(const uint32_t* __restrict__ in,
                                                          uint32_t* __restrict__ out,
                                                          uint32_t* __restrict__ block_sums,
                                                          uint64_t n) {
  __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  scan_load8(in, base, n, v);
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) sum += v[i];
  // wave inclusive scan of the per-thread sums
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wid] = inc;
  __syncthreads();
  uint32_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; w++) {
    if (w < wid) wave_off += s_wave[w];
    total += s_wave[w];
  }
  uint32_t run = wave_off + inc - sum;  // exclusive prefix of this thread
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const uint32_t x = v[i];
    v[i] = INCLUSIVE ? run + x : run;
    run += x;
  }
  scan_store8(out, base, n, v);
  if (block_sums && threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

MUSC_KERNEL __launch_bounds__(SCAN_BLOCK) void k_scan_add(uint32_t* __restrict__ out,
                                                        const uint32_t* __restrict__ block_off,
                                                        uint64_t n) {
  const uint32_t add = block_off[blockIdx.x];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  scan_load8(out, base, n, v);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) v[i] += add;
  scan_store8(out, base, n, v);
}

// u64 variant of the scan for the overflow-list offsets of big databases (index build only;
// plain element accesses, no tuning needed)
MUSC_KERNEL __launch_bounds__(SCAN_BLOCK) void k_scan64_block(const uint64_t* __restrict__ in,
                                                            uint64_t* __restrict__ out,
                                                            uint64_t* __restrict__ block_sums, uint64_t n) {
  __shared__ uint64_t s_wave[SCAN_BLOCK / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = (base + i < n) ? in[base + i] : 0ull;
    sum += v[i];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint64_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wid] = inc;
  __syncthreads();
  uint64_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; w++) {
    if (w < wid) wave_off += s_wave[w];
    total += s_wave[w];
  }
  uint64_t run = wave_off + inc - sum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) out[base + i] = run;  // exclusive
    run += v[i];
  }
  if (block_sums && threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

MUSC_KERNEL __launch_bounds__(SCAN_BLOCK) void k_scan64_add(uint64_t* __restrict__ out,
                                                          const uint64_t* __restrict__ block_off, uint64_t n) {
  const uint64_t add = block_off[blockIdx.x];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n) out[base + i] += add;
}
