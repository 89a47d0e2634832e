// libmuscato_hip.so -- muscato's seed-and-extend hot path on MI355X (gfx950, wave64).
//
// Path (reference file:line, kshedden/muscato):
//   screen : cmd/muscato_screen/main.go:116-207 (read k-mer sketch), :256-366 (target scan)
//   join   : cmd/muscato/main.go:318-385 (sort) + cmd/muscato_confirm/main.go:375-416 (merge)
//   confirm: cmd/muscato_confirm/main.go:151-159 (cdiff), :171-250 (searchpairs)
//   select : cmd/muscato_combine_windows/main.go:36-60 (per-read best + MMTol)
//
// MI355X design: the target database stays resident in HBM as one 2-bit
// stream plus a k-mer -> (gene, offset) table of 64-byte buckets built once per (database,
// WindowWidth); reads are fixed-stride 2-bit records.  One pass = k_screen (window keys probe
// the table, candidates filtered from the index entry alone) -> k_confirm (XOR/popcount Hamming
// distance, then per-read best + MMTol and MaxMatches accounting in the same workgroup; bound by
// the cache lines it gathers) -> tile scan -> k_compact (tuples in read order).  Read prep
// (bytewise sort + collapse, muscato_prep.hpp) is a separate entry point.  The Bloom sketch of
// the reference only prunes work and cannot change results
// (SURVEY.md 8a note H): every candidate is verified exactly in k_confirm, including its
// window key.
//
// There is no CPU fallback in this library.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/muscato_hip.h"

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------

#define DEV __device__ __forceinline__

DEV uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
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

#define MAX_GRID 4096u  // grid-stride kernels: enough blocks to fill 256 CUs several times

// ------------------------------------------------------------------------------------
// packing kernels (ASCII / 2-bit stream -> device layout)
// ------------------------------------------------------------------------------------

DEV uint32_t ascii_code(unsigned char c, uint32_t* isx) {
  const uint32_t a = c == 'A', cc = c == 'C', g = c == 'G', t = c == 'T';
  *isx = !(a | cc | g | t);
  return cc | (g << 1) | (t * 3u);
}

// one thread per u32 word (16 bases) of the database stream
__global__ void k_pack_db_ascii(const unsigned char* __restrict__ s, uint64_t nbases,
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
__global__ void k_pack_db_packed(const uint32_t* __restrict__ in2, const uint16_t* __restrict__ inm,
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

__global__ void k_max_len(const uint64_t* __restrict__ off, uint64_t n, unsigned long long* out) {
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
    rd[t] = len & 0xFFFFu;
    rdm[t] = 0;
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
  rdm[t] = mv;
  if (mv) atomicOr(has_x, 1u);
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

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_add(uint32_t* __restrict__ out,
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
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan64_block(const uint64_t* __restrict__ in,
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

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan64_add(uint64_t* __restrict__ out,
                                                          const uint64_t* __restrict__ block_off, uint64_t n) {
  const uint64_t add = block_off[blockIdx.x];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n) out[base + i] += add;
}

// ------------------------------------------------------------------------------------
// database index: a table of 64-byte buckets, one per k-mer key (direct) or key hash.
//   Bucket = { count, cursor (build only), overflow start, 3 inline entries }.
// A probe is ONE 64-byte access (half a 128-B line, the unit every L2 miss fetches on
// gfx950): count and the first three entries arrive together; only buckets with more than
// three entries (1.5 % of the chance buckets at 1 Gbp / 15-mers) touch the overflow list E.
//
// One index entry (16 B, one dwordx4): everything k_screen needs about a window start
// without touching the per-gene offset table.
//   x gene   : target number; databases of 2^32 bases or more ("wide"): gene in bits 0-23,
//              bits 32-39 of the window start in bits 24-31
//   y gposw  : global base offset of the window start (low 32 bits)
//   z lr     : min(jx, 65535) | min(T - jx, 65535) << 16  (distances to the gene's two ends,
//              saturated: window starts and read lengths are < 65535, so every comparison
//              k_screen makes against them is exact)
//   w flank  : the 8 bases left of the window (bits 0-15, base jx-1 in bits 14-15) and the
//              8 bases right of it (bits 16-31, base jx+ww in bits 16-17), 2 bits each
// ------------------------------------------------------------------------------------
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
#define BUCKET_INLINE 3
struct __attribute__((aligned(64))) Bucket {
  uint32_t count;
  uint32_t cursor;
  uint64_t ovf;
  uint4 e[BUCKET_INLINE];
};
static_assert(sizeof(Bucket) == 64, "bucket must be half a cache line");

DEV uint32_t flank_left(const uint32_t* __restrict__ w, uint64_t base) {
  // 8 bases ending just before base index `base` of plane w (zeros before the stream start)
  if (base >= 8) return (uint32_t)ext64(w, 2 * (base - 8)) & 0xFFFFu;
  return (uint32_t)(ext64(w, 0) << (2 * (8 - base))) & 0xFFFFu;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void k_index(const uint32_t* __restrict__ db2,
                                               const uint32_t* __restrict__ dbm2,
                                               const uint64_t* __restrict__ seq_off, uint32_t nseq,
                                               uint64_t nbases, int ww, int bits, int direct, int wide,
                                               Bucket* __restrict__ T, uint4* __restrict__ E) {
  // one chunk of 256 consecutive bases per iteration (a dispatch holds fewer than 2^32
  // work-items, so a thread per base cannot cover a database of 2^32 bases or more)
  __shared__ uint32_t s_g0;
  const uint64_t nchunks = (nbases + blockDim.x - 1) / blockDim.x;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t gfirst = chunk * blockDim.x;
    __syncthreads();  // the previous chunk's readers of s_g0 are done
    if (threadIdx.x == 0) {
      uint32_t lo = 0, hi = nseq;  // largest i < nseq with seq_off[i] <= gfirst
      while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (seq_off[mid] <= gfirst) lo = mid; else hi = mid;
      }
      s_g0 = lo;
    }
    __syncthreads();
    const uint64_t g = gfirst + threadIdx.x;
    if (g >= nbases) continue;
    uint32_t gene = s_g0;
    while (seq_off[gene + 1] <= g) gene++;
    const uint64_t s = seq_off[gene], e = seq_off[gene + 1];
    const uint64_t jx = g - s;
    if (jx + (uint64_t)ww > e - s) continue;  // window would cross the target end
    const uint32_t b = bucket_of(db2, dbm2, 2 * g, ww, bits, direct);
    if (!SCATTER) {
      atomicAdd(&T[b].count, 1u);
    } else {
      const uint32_t slot = atomicAdd(&T[b].cursor, 1u);
      const uint64_t rem = e - g;  // T - jx
      const uint32_t lr = (uint32_t)(jx > 65535 ? 65535 : jx) | ((uint32_t)(rem > 65535 ? 65535 : rem) << 16);
      const uint32_t fl = flank_left(db2, g) | (((uint32_t)ext64(db2, 2 * (g + (uint64_t)ww)) & 0xFFFFu) << 16);
      const uint4 ent = make_uint4(wide ? (gene | ((uint32_t)(g >> 32) << 24)) : gene, (uint32_t)g, lr, fl);
      if (slot < BUCKET_INLINE) T[b].e[slot] = ent;
      else E[T[b].ovf + (slot - BUCKET_INLINE)] = ent;
    }
  }
}

// overflow list sizes: tmp[b] = max(count - 3, 0), scanned on the side, written back as ovf
__global__ void k_index_ovf_count(const Bucket* __restrict__ T, uint64_t nb, uint64_t* __restrict__ tmp) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) {
    const uint32_t c = T[b].count;
    tmp[b] = c > BUCKET_INLINE ? c - BUCKET_INLINE : 0u;
  } else if (b == nb) {
    tmp[b] = 0;
  }
}

__global__ void k_index_ovf_set(Bucket* __restrict__ T, uint64_t nb, const uint64_t* __restrict__ tmp) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) T[b].ovf = tmp[b];
}

// ------------------------------------------------------------------------------------
// hot path kernels
// ------------------------------------------------------------------------------------

struct PathParams {
  int32_t W;
  int32_t win[MUSC_MAX_WINDOWS];
  int32_t ww;
  int32_t min_dinuc;
  int32_t bits;
  int32_t direct;
  int32_t mmtol;
  int32_t apply_mmtol;
  uint32_t q1zero_mask;  // windows whose start is 0 (the pos-0 path of processSeq)
  int32_t wide;          // database >= 2^32 bases: 40-bit positions (gene < 2^24)
  int32_t max_len;       // longest read loaded
  int32_t dbg;           // experiments only (MUSC_DEBUG_SCREEN): 1 skip entry tests, 2 skip bucket loads, 4 skip desc writes, 8 no two-window descriptors
};

// A read record: RW u32 words, bases (2 bits each) in words 0..RW-2, length in word RW-1.
// RW > 0: the whole record sits in registers after two (or more) 16-byte loads -- the
// thread-per-read kernels are bound by the number of memory instructions they issue (each
// one touches 16+ cache lines per wave), not by arithmetic.  RW == 0: runtime stride, the
// words are read from memory on demand.
template <int RW>
struct Rec {
  uint32_t w[RW];
  DEV void load(const uint32_t* __restrict__ p, int) {
#pragma unroll
    for (int q = 0; q < RW / 4; q++) {
      const uint4 a = *reinterpret_cast<const uint4*>(p + 4 * q);
      w[4 * q] = a.x; w[4 * q + 1] = a.y; w[4 * q + 2] = a.z; w[4 * q + 3] = a.w;
    }
  }
  DEV uint32_t len() const { return w[RW - 1] & 0xFFFFu; }
  // 64 bits from bit offset bo.  bo is wave-uniform in every caller, so the word index is
  // resolved by a scalar branch and each case names its registers statically (a chain of
  // per-word selects costs 3*RW VALU per call instead).
  DEV uint64_t ext(uint32_t bo) const {
    const uint32_t sh = bo & 31u;
    uint32_t a = 0, b = 0, c = 0;
    switch (__builtin_amdgcn_readfirstlane((int)(bo >> 5))) {
#define MUSC_EXT_CASE(I)                                             \
  case I:                                                            \
    a = w[I < RW ? I : 0];                                           \
    b = (I + 1 < RW) ? w[I + 1 < RW ? I + 1 : 0] : 0u;               \
    c = (I + 2 < RW) ? w[I + 2 < RW ? I + 2 : 0] : 0u;               \
    break;
      MUSC_EXT_CASE(0) MUSC_EXT_CASE(1) MUSC_EXT_CASE(2) MUSC_EXT_CASE(3)
      MUSC_EXT_CASE(4) MUSC_EXT_CASE(5) MUSC_EXT_CASE(6) MUSC_EXT_CASE(7)
      MUSC_EXT_CASE(8) MUSC_EXT_CASE(9) MUSC_EXT_CASE(10) MUSC_EXT_CASE(11)
      MUSC_EXT_CASE(12) MUSC_EXT_CASE(13) MUSC_EXT_CASE(14) MUSC_EXT_CASE(15)
#undef MUSC_EXT_CASE
      default: break;
    }
    const uint64_t lo = (uint64_t)a | ((uint64_t)b << 32);
    return sh ? (lo >> sh) | ((uint64_t)c << (64 - sh)) : lo;
  }
};

template <>
struct Rec<-1> {};  // no record

template <>
struct Rec<0> {
  const uint32_t* __restrict__ p;
  int rw;
  DEV void load(const uint32_t* __restrict__ q, int rw_rt) { p = q; rw = rw_rt; }
  DEV uint32_t len() const { return p[rw - 1] & 0xFFFFu; }
  DEV uint64_t ext(uint32_t bo) const { return ext64(p, bo); }
};

// 8 bases ending just before base `base` of a record, like flank_left on a stream
template <class R>
DEV uint32_t rec_flank_left(const R& r, uint32_t base) {
  if (base >= 8) return (uint32_t)r.ext(2 * (base - 8)) & 0xFFFFu;
  return (uint32_t)(r.ext(0) << (2 * (8 - base))) & 0xFFFFu;
}

// bucket of the read window starting at base q1 (same function as bucket_of on the database)
template <class R>
DEV uint32_t rec_bucket(const R& r, const R& m, bool has_m, uint32_t q1, int ww, int bits, int direct) {
  const int nb = 2 * ww;
  if (direct && !has_m) return (uint32_t)(__brevll(r.ext(2 * q1) & lowmask64(nb)) >> (64 - nb));
  uint64_t h = 0, anymask = 0, key0 = 0;
  for (int c = 0; c < nb; c += 64) {
    const int take = nb - c < 64 ? nb - c : 64;
    const uint64_t key = r.ext(2 * q1 + c) & lowmask64(take);
    const uint64_t mk = has_m ? (m.ext(2 * q1 + c) & lowmask64(take)) : 0ull;
    if (c == 0) key0 = key;
    anymask |= mk;
    h = mix64(h ^ key ^ mix64(mk + 0x9E3779B97F4A7C15ull * (uint64_t)(c + 1)));
  }
  if (direct && anymask == 0) return (uint32_t)(__brevll(key0) >> (64 - nb));
  return (uint32_t)(h >> (64 - bits));
}

// utils/entropy.go:5-40 on packed bases: number of distinct adjacent letter pairs over the
// 5-letter alphabet {A,C,G,T,other}; the count does not depend on how letters are numbered.
template <class R>
DEV int rec_count_dinuc(const R& r, const R& m, bool has_m, uint32_t q1, int ww) {
  uint32_t seen25 = 0, seen16 = 0;  // pairs with / without an X involved
  for (int c = 0; c + 1 < ww; c += 31) {  // 32-base chunks overlapping by one base
    const int nbase = ww - c < 32 ? ww - c : 32;
    const uint64_t key = r.ext(2 * (q1 + c));
    const uint64_t mk = has_m ? (m.ext(2 * (q1 + c)) & lowmask64(2 * nbase)) : 0ull;
    if (mk == 0) {
      // no X in the chunk: a pair of bases is a 4-bit code, 16 possible pairs
      for (int i = 0; i + 1 < nbase; i++) seen16 |= 1u << ((uint32_t)(key >> (2 * i)) & 15u);
    } else {
      for (int i = 0; i + 1 < nbase; i++) {
        const uint32_t a = ((mk >> (2 * i)) & 1u) ? 4u : ((uint32_t)(key >> (2 * i)) & 3u);
        const uint32_t b = ((mk >> (2 * i + 2)) & 1u) ? 4u : ((uint32_t)(key >> (2 * i + 2)) & 3u);
        seen25 |= 1u << (a * 5 + b);
      }
    }
  }
  if (seen25 == 0) return __popc(seen16);  // the usual case: distinct pairs over {A,C,G,T}
  for (int q = 0; q < 16; q++)  // merge: pair code a + 4b -> a*5 + b
    if ((seen16 >> q) & 1u) seen25 |= 1u << ((q & 3) * 5 + (q >> 2));
  return __popc(seen25);
}

// per-pair result word: bits 0-15 mismatch count, bit 16 NX_DUP, bits 17-18 which of the
// descriptor's windows accept, bits 20-23 window, bits 24-31 the read's slot within its tile
#define NX_REJECT 0xFFFFFFFFu
#define NX_DUP 0x10000u  // accepted through this window, but an earlier window reports the tuple
#define NX_ACC1 0x20000u  // the descriptor's second window (k + 1) accepts the pair
#define NX_ACC0 0x40000u  // the descriptor's window k accepts the pair
#define DESC_TWO (1u << 22)  // descriptor z: windows k and k + 1 both found this placement
#define BLOCK_TABLE_BITS 22
#define WB_NONE 0xFFFFFFFFu

#define TILE 256  // reads per tile = threads per workgroup of k_screen
#define CONF_NM 256  // read lengths whose mismatch budget the kernels keep in LDS

// k_screen -- muscato_screen + the join, fused: one workgroup iteration per tile of 256 reads.
// For each window of a read that takes part (cmd/muscato_window_reads/main.go:106-118 ==
// cmd/muscato_screen/main.go:174-185: long enough, CountDinuc >= MinDinuc) the window's index
// bucket is fetched and its entries are tested (three arrive with the bucket, the rest sit in
// the overflow array).  From the entry alone: p = jx - q1 >= 0, the fit rules of
// cmd/muscato_screen/main.go:294-316 (target position 0: the literal 100) and :335-363 +
// cmd/muscato_confirm/main.go:201-203 (the read must end inside the target), and a flank
// pre-filter: a candidate whose 8+8 flanking bases already disagree with the read in more
// places than the read's whole mismatch budget can never be accepted by cdiff
// (cmd/muscato_confirm/main.go:205-211) and is dropped before it costs a target gather (chance
// k-mer hits are about half of all candidates).  The flank test never over-counts: an X is
// stored as code 0 on both sides.  The phases are described inside the kernel.
//
// Survivors are appended (one LDS atomic per wave instruction) to the tile's range inside the
// workgroup's own region of `desc` (region = desc_cap / gridDim descriptors, so no global
// cursor is needed).  A tile's descriptors are contiguous and in (read, window) order up to
// interleaving of the four waves, which keeps k_confirm's record loads local; nothing
// downstream needs more than that (k_confirm keeps per-read state in LDS and orders the tuples).
//
// Descriptor (16 B): x = read index within the batch (24 bits) | bits 32-39 of the placement's
// global offset << 24, y = its low 32 bits, z = window k | z-flag << 4 | pos_ok << 5 | position
// in the target << 6 (when it fits 16 bits exactly) | DESC_TWO (windows k and k+1 both found
// the placement), w = gene.
// counters (batch-local block = pass-level block + 8): [0] valid windows, [3] candidates (index
//           entries walked), [4] descriptors, [5] descriptors that stand for two windows, [7] the
//           largest number of descriptors any workgroup needed (region size to retry with);
//           pass-level [3] is raised when a region ran out.
#define SCR_OWN 2048  // overflow items per chunk whose owner is looked up directly
#define SCR_PROBES (2 * TILE)  // probes per chunk: two windows of every read of the tile

// fit rules + flank filter for one index entry against one probe; true = worth a target gather
DEV bool screen_entry_ok(const uint4 ent, int q1, int ww, uint32_t rfl, uint32_t lenbud, uint32_t* zflag) {
  const int q2 = q1 + ww;
  const int rlen = (int)(lenbud & 0xFFFFu);
  const int nl = q1 < 8 ? q1 : 8;                                       // bases left of the window
  const int nr = rlen - q2 < 8 ? (rlen - q2 < 0 ? 0 : rlen - q2) : 8;   // bases right of it
  const uint32_t fmask = (nl ? ((0xFFFFu << (16 - 2 * nl)) & 0xFFFFu) : 0u) | ((nr ? ((1u << (2 * nr)) - 1u) : 0u) << 16);
  const int left = (int)(ent.z & 0xFFFFu), right = (int)(ent.z >> 16);
  int lim0 = 100 - ww;            // cmd/muscato_screen/main.go:305 (q1 == 0 there)
  const int tcap = left + right;  // target length, saturated (exact below 65535)
  if (lim0 > tcap) lim0 = tcap;
  const bool fit0 = rlen <= lim0;
  bool ok = q1 <= left;                  // p = jx - q1 >= 0
  if (left == 0) ok = ok && fit0;        // window at target position 0: pos-0 path
  else ok = ok && (rlen - q1 <= right);  // p + len <= T
  const uint32_t x = rfl ^ ent.w;
  const uint32_t d = (x | (x >> 1)) & 0x55555555u & fmask;
  ok = ok && ((uint32_t)__popc(d) <= (lenbud >> 16));
  *zflag = (left == q1 && !fit0) ? 1u : 0u;  // p == 0 but the pos-0 path rejects
  return ok;
}

// Barrier for phases that communicate through LDS only: unlike __syncthreads() it does not wait
// for outstanding global loads and stores (descriptor and tuple stores drain in the background).
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

DEV uint32_t opaque(uint32_t x) {  // stops the compiler from keeping values derived from x across loop iterations
  asm volatile("" : "+v"(x));
  return x;
}

#ifndef SCR_ROUNDS
#define SCR_ROUNDS 4  // quad rounds whose bucket loads are in flight together (8 per chunk)
#endif
#ifndef SCR_WAVES
#define SCR_WAVES 4  // waves per SIMD the register allocator has to leave room for
#endif
template <int RW, bool MASK>
__global__ __launch_bounds__(TILE, SCR_WAVES) void k_screen(const uint32_t* __restrict__ rd,
                                                 const uint32_t* __restrict__ rdm, uint64_t r0,
                                                 uint32_t n, int rw_rt, PathParams pp,
                                                 const uint16_t* __restrict__ nmiss_tab,
                                                 const Bucket* __restrict__ T,
                                                 const uint4* __restrict__ E,
                                                 uint4* __restrict__ desc, uint64_t desc_cap,
                                                 uint32_t* __restrict__ rvalid,
                                                 uint32_t* __restrict__ wb,
                                                 uint32_t* __restrict__ tbase,
                                                 uint32_t* __restrict__ tcount,
                                                 unsigned long long* __restrict__ counters) {
  // Phase A: a thread per read gates the read's windows and names their buckets.
  // Phase B: the 64-byte buckets are fetched by quads of lanes (16 bytes each: one wave
  // instruction brings 16 whole buckets, every line is requested once) and stay in registers:
  // the lane that holds an inline entry tests it on the spot, so a bucket's count and its
  // first three entries cost one memory round trip and no LDS staging.  Quads are numbered in
  // (read, window) order and survivors are appended per wave in lane order, so a tile's
  // descriptors come out read-major (k_confirm's record loads stay local).
  // Phase C: entries beyond the third of a bucket live in E; all such entries of the chunk
  // are tested as ONE flat list spread evenly over the 256 threads (a probe with c overflow
  // entries owns c consecutive items): every load is independent and they are issued together.
  // The workgroup keeps little LDS (18 KB) so that eight of them share a CU and one
  // workgroup's memory round trips hide behind the others' arithmetic.
  __shared__ uint32_t s_wsum[TILE / 64];
  __shared__ uint32_t s_bb[SCR_PROBES];        // per probe: bucket, WB_NONE when the window takes no part
  __shared__ uint32_t s_rfl[SCR_PROBES];       // per probe: the read's own 8+8 flanking bases
  __shared__ uint32_t s_lenbud[SCR_PROBES];    // read length | mismatch budget << 16
  __shared__ uint32_t s_oc[SCR_PROBES];        // entries of the probe's bucket that live in E
  __shared__ uint64_t s_ovf[SCR_PROBES];       // where in E
  __shared__ uint32_t s_pref[SCR_PROBES + 1];  // exclusive prefix of s_oc
  __shared__ uint16_t s_own[SCR_OWN];          // flat item -> probe
  __shared__ uint32_t s_tilecnt;               // survivors of the tile so far
  __shared__ uint16_t s_nm[CONF_NM];           // mismatch budget of the short read lengths
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)pp.max_len ? nmiss_tab[t] : (uint16_t)0;
  lds_barrier();
  const int rw = RW ? RW : rw_rt;
  constexpr bool has_m = MASK;  // the mask planes exist (some read or target holds an X)
  const uint32_t ntiles = (n + TILE - 1) / TILE;
  unsigned long long nvalid = 0, ncand = 0, ntwo = 0;
  const uint64_t region = desc_cap / gridDim.x;
  const uint64_t region0 = region * blockIdx.x;
  uint64_t used = 0;  // descriptors this workgroup has needed so far (uniform across the workgroup)

  // workgroup exclusive scan helper: returns this thread's exclusive prefix, *total = sum
  auto wg_scan = [&](uint32_t v, uint32_t* total) -> uint32_t {
    const uint32_t tid = opaque(threadIdx.x);
    const int lane = tid & 63, wid = tid >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    lds_barrier();  // earlier readers of s_wsum are done
    if (lane == 63) s_wsum[wid] = inc;
    lds_barrier();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TILE / 64; w++) {
      if (w < wid) woff += s_wsum[w];
      tot += s_wsum[w];
    }
    *total = tot;
    return woff + inc - v;
  };

  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t tida = opaque(threadIdx.x);
    const uint32_t i = tile * TILE + tida;
    const bool active = i < n;
    const uint64_t r = r0 + (active ? i : 0);
    Rec<RW> rec;
    rec.load(rd + r * (uint64_t)rw, rw);
    Rec<has_m ? RW : -1> recm_store;
    if constexpr (has_m) recm_store.load(rdm + r * (uint64_t)rw, rw);
    const auto& recm = [&]() -> const Rec<RW>& {
      if constexpr (has_m) return recm_store; else return rec;  // never read without a mask plane
    }();
    const int len = (int)rec.len();
    const uint32_t budget = len < CONF_NM ? s_nm[len] : nmiss_tab[len];
    uint32_t valid = 0;
    if (tida == 0) s_tilecnt = 0;
    const uint64_t base = region0 + used;
    const uint64_t room = region > used ? region - used : 0;  // descriptors this tile may still write

    // one survivor per set lane of a wave-uniform vote: a wave claims its slots with one LDS
    // atomic and writes them in lane order
    auto append = [&](bool ok, const uint4 ent, uint32_t probe, int k, int q1, uint32_t z, bool two) {
      const unsigned long long vote = __ballot(ok);
      if (vote == 0) return;
      uint32_t first = 0;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      if (below == 0 && ok) first = atomicAdd(&s_tilecnt, (uint32_t)__popcll(vote));  // the first voter
      first = __builtin_amdgcn_readlane(first, __builtin_ctzll(vote));
      const uint32_t slot = first + below;
      if (ok && slot < room && !(pp.dbg & 4)) {
        const uint32_t left = ent.z & 0xFFFFu;
        const uint32_t pos_ok = left < 65535u ? 1u : 0u;
        // global offset of the placement (40 bits in wide mode: the high byte rides in x)
        const uint64_t gp = (((uint64_t)(pp.wide ? ent.x >> 24 : 0u) << 32) | ent.y) - (uint64_t)q1;
        desc[base + slot] = make_uint4((tile * TILE + (probe >> 1)) | ((uint32_t)(gp >> 32) << 24), (uint32_t)gp,
                                       (uint32_t)k | (z << 4) | (pos_ok << 5) | ((left - (uint32_t)q1) << 6) | (two ? DESC_TWO : 0u),
                                       pp.wide ? (ent.x & 0xFFFFFFu) : ent.x);
      }
    };

    for (int k0 = 0; k0 < pp.W; k0 += 2) {
      // ---- phase A: which of this read's next two windows take part, and their buckets
      const int q1a = pp.win[k0], q1b = pp.win[k0 + 1 < pp.W ? k0 + 1 : k0];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int k = k0 + j;
        uint32_t b = WB_NONE;
        if (active && k < pp.W) {
          // cmd/muscato_window_reads/main.go:106-118 == cmd/muscato_screen/main.go:174-185
          const uint32_t q1 = (uint32_t)pp.win[k], q2 = q1 + (uint32_t)pp.ww;
          bool pt = (uint32_t)len >= q2;
          if (pt && pp.min_dinuc > 0) pt = rec_count_dinuc(rec, recm, has_m, q1, pp.ww) >= pp.min_dinuc;
          if (pt) {
            b = rec_bucket(rec, recm, has_m, q1, pp.ww, pp.bits, pp.direct);
            s_rfl[2 * tida + j] = rec_flank_left(rec, q1) | (((uint32_t)rec.ext(2u * q2) & 0xFFFFu) << 16);
            s_lenbud[2 * tida + j] = (uint32_t)len | (budget << 16);
            valid |= 1u << k;
          }
          wb[(uint64_t)i * pp.W + k] = b;
        }
        s_bb[2 * tida + j] = b;
      }
      lds_barrier();
      // ---- phase B: buckets by quads; a wave fetches the 128 probes of its own 64 reads
#pragma unroll 1
      for (int h = 0; h < ((pp.dbg & 256) ? 0 : 8 / SCR_ROUNDS); h++) {
        const uint32_t tidb = opaque(threadIdx.x);
        const uint32_t lane = tidb & 63, wid = tidb >> 6;
        uint4 v[SCR_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < SCR_ROUNDS; rr++) {
          const uint32_t b = s_bb[wid * 128 + (SCR_ROUNDS * h + rr) * 16 + (lane >> 2)];
          v[rr] = make_uint4(0, 0, 0, 0);
          if (b != WB_NONE && !(pp.dbg & 2)) {
            // non-temporal: a bucket is used once (measured: random 64-B fetches run 12 % faster)
            const u32x4_v t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(T + b) + (lane & 3));
            v[rr] = make_uint4(t.x, t.y, t.z, t.w);
          }
        }
#pragma unroll
        for (int rr = 0; rr < SCR_ROUNDS; rr++) {
          const uint32_t probe = wid * 128 + (SCR_ROUNDS * h + rr) * 16 + (lane >> 2);
          const uint32_t part = lane & 3;
          // the header sits in the quad's first lane (quad_perm [0,0,0,0])
          const uint32_t cnt = (uint32_t)__builtin_amdgcn_mov_dpp((int)v[rr].x, 0x00, 0xF, 0xF, true);
          if (part == 0) {
            ncand += cnt;
            s_oc[probe] = cnt > BUCKET_INLINE ? cnt - BUCKET_INLINE : 0u;
            s_ovf[probe] = (uint64_t)v[rr].z | ((uint64_t)v[rr].w << 32);
          }
          const int k = k0 + (int)(probe & 1u), q1 = (probe & 1u) ? q1b : q1a;
          uint32_t z = 0;
          bool ok = part >= 1 && part - 1 < cnt && !(pp.dbg & 1);
          if (ok) ok = screen_entry_ok(v[rr], q1, pp.ww, s_rfl[probe], s_lenbud[probe], &z);
          // The read's two windows sit in neighbouring quads.  When both hold a surviving entry
          // for the same placement, one descriptor stands for both (k_confirm compares the
          // pair once and credits both windows); the second window's lane drops its own.
          // Every lane executes every cross-lane move: no short-circuit evaluation here.
          const uint32_t gp = v[rr].y - (uint32_t)q1;
          const uint32_t gx = ok ? v[rr].x : 0xFFFFFFFFu;  // no target has this number
          const bool odd = (probe & 1u) != 0;
          const uint32_t nx0 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)gx, 0x114, 0xF, 0xF, false);  // row_shr:4
          const uint32_t nx1 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)gx, 0x104, 0xF, 0xF, false);  // row_shl:4
          const uint32_t ng0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gp, 0x114, 0xF, 0xF, false);
          const uint32_t ng1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gp, 0x104, 0xF, 0xF, false);
          const uint32_t px = odd ? nx0 : nx1, pg = odd ? ng0 : ng1;  // the partner quad, same part
          uint32_t same = (uint32_t)(px == gx) & (uint32_t)(pg == gp);
#define MUSC_QROT(CTRL)                                                                            \
          {                                                                                        \
            const uint32_t qx = (uint32_t)__builtin_amdgcn_mov_dpp((int)px, CTRL, 0xF, 0xF, true); \
            const uint32_t qg = (uint32_t)__builtin_amdgcn_mov_dpp((int)pg, CTRL, 0xF, 0xF, true); \
            same |= (uint32_t)(qx == gx) & (uint32_t)(qg == gp);                                   \
          }
          MUSC_QROT(0x39) MUSC_QROT(0x4E) MUSC_QROT(0x93)  // the partner quad's other three parts
#undef MUSC_QROT
          if (!ok || (pp.dbg & 8)) same = 0;
          const bool two = same && !odd;
          if (same && odd) ok = false;
          ntwo += two;
          append(ok, v[rr], probe, k, q1, z, two);
        }
      }
      lds_barrier();
      // ---- phase C: the chunk's overflow entries as one flat list, in (read, window, entry) order
      const uint32_t tidc = opaque(threadIdx.x);
      if (pp.dbg & 256) continue;
      const uint32_t oc0 = s_oc[2 * tidc], oc1 = s_oc[2 * tidc + 1];
      uint32_t total = 0;
      const uint32_t pre = wg_scan(oc0 + oc1, &total);
      if (total != 0 && !(pp.dbg & 1)) {  // uniform
        s_pref[2 * tidc] = pre;
        s_pref[2 * tidc + 1] = pre + oc0;
        if (tidc == TILE - 1) s_pref[SCR_PROBES] = total;
        for (uint32_t e = 0; e < oc0 && pre + e < SCR_OWN; e++) s_own[pre + e] = (uint16_t)(2 * tidc);
        for (uint32_t e = 0; e < oc1 && pre + oc0 + e < SCR_OWN; e++) s_own[pre + oc0 + e] = (uint16_t)(2 * tidc + 1);
        lds_barrier();
        for (uint32_t t0 = 0; t0 < total; t0 += TILE) {
          const uint32_t t = t0 + tidc;
          bool ok = t < total;
          uint32_t seg = 0, z = 0;
          uint4 ent = make_uint4(0, 0, 0, 0);
          int k = k0, q1 = 0;
          if (ok) {
            if (t < SCR_OWN) {
              seg = s_own[t];
            } else {  // rare: largest seg with s_pref[seg] <= t
              uint32_t lo = 0, hi = SCR_PROBES;
              while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) / 2;
                if (s_pref[mid] <= t) lo = mid; else hi = mid;
              }
              seg = lo;
            }
            const u32x4_v te = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(E) + s_ovf[seg] + (t - s_pref[seg]));
            ent = make_uint4(te.x, te.y, te.z, te.w);
            k = k0 + (int)(seg & 1u);
            q1 = (seg & 1u) ? q1b : q1a;
            ok = screen_entry_ok(ent, q1, pp.ww, s_rfl[seg], s_lenbud[seg], &z);
          }
          append(ok, ent, seg, k, q1, z, false);
        }
      }
      lds_barrier();  // the LDS tables are reused by the next chunk
    }
    nvalid += __popc(valid);
    if (active) rvalid[i] = valid;
    const uint32_t total = s_tilecnt;
    const bool fits = total <= room;  // else: the host grows desc and repeats the batch
    used += total;
    if (tida == 0) {
      tbase[tile] = (uint32_t)base;
      tcount[tile] = fits ? total : 0u;
    }
    lds_barrier();  // before the next tile resets s_tilecnt
  }
  block_add_u64(nvalid, &counters[0]);
  block_add_u64(ncand, &counters[3]);
  block_add_u64(ntwo, &counters[5]);
  if (threadIdx.x == 0) {
    atomicAdd(&counters[4], (unsigned long long)used);
    atomicMax(&counters[7], (unsigned long long)used);
    if (used > region) atomicOr(&counters[3 - 8], 1ull);  // pass-level flag: descriptor space ran out
  }
}

// u32 mask of the bits of window [q1, q1+ww) (2 bits per base) that fall in record word j
DEV uint32_t window_word_mask(int q1, int ww, int j) {
  const int lo = 2 * q1 - 32 * j, hi = lo + 2 * ww;
  if (hi <= 0 || lo >= 32) return 0u;
  const uint32_t mh = hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u);
  const uint32_t ml = lo <= 0 ? 0xFFFFFFFFu : ~((1u << lo) - 1u);
  return mh & ml;
}


// 16 bytes at a dword-aligned address (global memory allows it on gfx950)
struct __attribute__((packed, aligned(4))) u32x4_u {
  uint32_t x, y, z, w;
};

// What one pair brings in from memory (static stride): descriptor, record, target span, masks.
template <int RW, bool MASK>
struct PairRegs {
  uint4 ds;
  uint32_t exact;  // rvalid of the read
  uint32_t r[RW ? RW : 1], t[RW ? RW : 1], rm[(RW && MASK) ? RW : 1], tm[(RW && MASK) ? RW : 1];
};

// pair_issue: every load of the pair, nothing that needs their results -- so that a caller can
// put other work between issue and finish
template <int RW, bool MASK>
DEV void pair_issue(PairRegs<RW, MASK>& P, const uint4 ds, const uint32_t* __restrict__ rd,
                    const uint32_t* __restrict__ rdm, const uint32_t* __restrict__ db2,
                    const uint32_t* __restrict__ dbm2, uint64_t r0, const uint32_t* __restrict__ rvalid) {
  static_assert(RW != 0, "static stride only");
  P.ds = ds;
  const uint32_t ri = ds.x & 0xFFFFFFu;
  const uint64_t gpos = (uint64_t)ds.y | ((uint64_t)(ds.x >> 24) << 32);
  const uint32_t* __restrict__ rec = rd + (r0 + ri) * (uint64_t)RW;
  const uint64_t widx = gpos >> 4;
  P.exact = rvalid[ri];
  // read records stream through once: non-temporal, so that the database -- the only operand
  // with reuse -- keeps the Infinity Cache
#pragma unroll
  for (int q = 0; q < RW / 4; q++) {
    const u32x4_v a = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(rec) + q);
    P.r[4 * q] = a.x; P.r[4 * q + 1] = a.y; P.r[4 * q + 2] = a.z; P.r[4 * q + 3] = a.w;
    const u32x4_u b = *reinterpret_cast<const u32x4_u*>(db2 + widx + 4 * q);
    P.t[4 * q] = b.x; P.t[4 * q + 1] = b.y; P.t[4 * q + 2] = b.z; P.t[4 * q + 3] = b.w;
    if constexpr (MASK) {
      const uint4 c = *reinterpret_cast<const uint4*>(rdm + (r0 + ri) * (uint64_t)RW + 4 * q);
      P.rm[4 * q] = c.x; P.rm[4 * q + 1] = c.y; P.rm[4 * q + 2] = c.z; P.rm[4 * q + 3] = c.w;
      const u32x4_u d = *reinterpret_cast<const u32x4_u*>(dbm2 + widx + 4 * q);
      P.tm[4 * q] = d.x; P.tm[4 * q + 1] = d.y; P.tm[4 * q + 2] = d.z; P.tm[4 * q + 3] = d.w;
    }
  }
}

// The reference's confirm for window k (and for k + 1 when the descriptor stands for both)
// accepts this pair -- it counts towards that window-key block's MaxMatches; the tuple is
// reported here only if the first window that accepts it is one of this descriptor's.
DEV uint32_t pair_code(const uint4 ds, uint32_t nx, uint32_t budget, uint32_t exact) {
  const uint32_t ri = ds.x & 0xFFFFFFu, k = ds.z & 15u;
  const bool within = nx <= budget;
  const bool a0 = within && ((exact >> k) & 1u);
  const bool a1 = within && (ds.z & DESC_TWO) && ((exact >> (k + 1)) & 1u);
  if (!(a0 || a1)) return NX_REJECT;
  const uint32_t kmin = (uint32_t)(__ffs(exact) - 1);
  const bool first = (a0 && kmin == k) || (a1 && kmin == k + 1);
  return (first ? nx : (nx | NX_DUP)) | (a0 ? NX_ACC0 : 0u) | (a1 ? NX_ACC1 : 0u) | (k << 20) | ((ri & (TILE - 1)) << 24);
}

// pair_finish: XOR + popcount = cdiff over the whole read (cmd/muscato_confirm/main.go:151-159,
// 205-211; X==X through the mask plane) and, from the same mismatch mask, which windows of the
// read match the target exactly here (the first-window rule that makes the union over windows a
// set without a sort).  Returns the pair's result word.  budget_of(len) = the read's mismatch
// budget.
template <int RW, bool MASK, class BudgetOf>
DEV uint32_t pair_finish(const PairRegs<RW, MASK>& P, const PathParams& pp, BudgetOf budget_of) {
  const uint64_t gpos = (uint64_t)P.ds.y | ((uint64_t)(P.ds.x >> 24) << 32);
  const uint32_t sh = ((uint32_t)gpos & 15u) * 2u;
  uint32_t exact = P.exact;
  if ((P.ds.z >> 4) & 1u) exact &= ~pp.q1zero_mask;
  const uint32_t len = P.r[RW - 1] & 0xFFFFu;
  const int len2 = 2 * (int)len;
  uint32_t nx = 0;
#pragma unroll
  for (int j = 0; j < RW - 1; j++) {
    const uint32_t tj = __funnelshift_r(P.t[j], P.t[j + 1], sh);
    const uint32_t x = P.r[j] ^ tj;
    uint32_t d = (x | (x >> 1)) & 0x55555555u;
    if constexpr (MASK) d |= (P.rm[j] ^ __funnelshift_r(P.tm[j], P.tm[j + 1], sh)) & 0x55555555u;
    const int rem = len2 - 32 * j;
    d &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
    nx += __popc(d);
    for (int kk = 0; kk < pp.W; kk++)
      if (d & window_word_mask(pp.win[kk], pp.ww, j)) exact &= ~(1u << kk);
  }
  return pair_code(P.ds, nx, budget_of(len), exact);
}

// confirm_pair -- cdiff for one candidate pair in one go.  Loads the 2-bit read record (aligned,
// neighbouring lanes mostly share it) and the target span at an arbitrary base offset
// (dword-aligned 16-byte gathers + funnel shift).  RW = record words (compile time) or 0 =
// runtime stride.  Returns the pair's result word (NX_REJECT, or nmiss | flags | window << 20 |
// slot << 24).
template <int RW, bool MASK, class BudgetOf>
DEV uint32_t confirm_pair(const uint4 ds, const uint32_t* __restrict__ rd, const uint32_t* __restrict__ rdm,
                          const uint32_t* __restrict__ db2, const uint32_t* __restrict__ dbm2, uint64_t r0,
                          int rw_rt, const PathParams& pp, BudgetOf budget_of, const uint32_t* __restrict__ rvalid) {
  if constexpr (RW != 0) {
    PairRegs<RW, MASK> P;
    pair_issue<RW, MASK>(P, ds, rd, rdm, db2, dbm2, r0, rvalid);
    return pair_finish<RW, MASK>(P, pp, budget_of);
  } else {
    // ---- runtime stride (reads longer than the compiled strides): streaming words
    const uint32_t ri = ds.x & 0xFFFFFFu;
    const int rw = rw_rt;
    const uint64_t gpos = (uint64_t)ds.y | ((uint64_t)(ds.x >> 24) << 32);
    const uint32_t* __restrict__ rec = rd + (r0 + ri) * (uint64_t)rw;
    const uint64_t widx = gpos >> 4;
    const uint32_t sh = ((uint32_t)gpos & 15u) * 2u;
    uint32_t exact = rvalid[ri];
    if ((ds.z >> 4) & 1u) exact &= ~pp.q1zero_mask;
    uint32_t nx = 0;
    const uint32_t len = rec[rw - 1] & 0xFFFFu;
    const int len2 = 2 * (int)len;
    const uint32_t* __restrict__ recm = MASK ? rdm + (r0 + ri) * (uint64_t)rw : nullptr;
    uint32_t tlo = db2[widx], tmlo = MASK ? dbm2[widx] : 0u;
    for (int j = 0; j < rw - 1; j++) {
      const uint32_t thi = db2[widx + j + 1];
      const uint32_t x = rec[j] ^ __funnelshift_r(tlo, thi, sh);
      tlo = thi;
      uint32_t d = (x | (x >> 1)) & 0x55555555u;
      if (MASK) {
        const uint32_t tmhi = dbm2[widx + j + 1];
        d |= (recm[j] ^ __funnelshift_r(tmlo, tmhi, sh)) & 0x55555555u;
        tmlo = tmhi;
      }
      const int rem = len2 - 32 * j;
      d &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
      nx += __popc(d);
      for (int kk = 0; kk < pp.W; kk++)
        if (d & window_word_mask(pp.win[kk], pp.ww, j)) exact &= ~(1u << kk);
    }
    return pair_code(ds, nx, budget_of(len), exact);
  }
}

#define BLOCK_LDS_BITS 11  // sketch size
#define CONF_TILES 32      // tiles per k_confirm workgroup (65536 tiles per batch / MAX_GRID = 16)


#define CODE_CAP 1024      // result words beyond a lane's first kept in LDS; a larger tile spills the rest to p_nx

// k_confirm -- muscato_confirm for one tile of k_screen per workgroup iteration, followed in
// the same workgroup by the per-read best + MMTol filter
// (cmd/muscato_combine_windows/main.go:36-60) and the MaxMatches block accounting: a tile's
// pairs are desc[tbase[tile] .. +tcount[tile]) in any order, every per-read quantity lives in
// LDS and the result words never travel through HBM (tiles of more than CODE_CAP pairs spill).
//   pass 1: one lane per pair: cdiff; best[read] = min nmiss over its reported pairs;
//           wcnt[read][window] = pairs that window's confirm accepts (its share of the
//           (window,key) block, which cmd/muscato_confirm/main.go:233-242, 424-448 truncate
//           at MaxMatches)
//   pass 2: cnt[read] = pairs with nmiss <= best + MMTol (all accepted pairs when
//           apply_mmtol == 0); scan over the tile's reads
//   pass 3: the surviving tuples go to stage[tbase[tile] + ...], reads in order, a read's
//           tuples contiguous (arrival order within one read); tcount2[tile] = how many.
//           k_compact then closes the gaps between tiles.
// block_mode 0: no MaxMatches accounting.
// block_mode 1: screening -- each workgroup keeps a count-min sketch of (window, key) -> accepted
//   pairs in LDS across all its tiles; if no sketch cell of any workgroup of any launch reaches
//   `block_thr` = floor(MaxMatches / number of workgroup-launches), then by pigeonhole no block can
//   hold more than MaxMatches pairs (cells only over-estimate).  Otherwise counters[6] is raised
//   and the host repeats the pass in mode 2.
// block_mode 2: exact -- one global atomic per (read, window) into a 2^22-cell table.
// (Eight waves per SIMD where the record fits 64 registers without spilling: measured 1.82 ms
// per cfg3 pass against 1.95 ms at the compiler's own choice.  Prefetching the next tile's gathers across the select passes was tried and lost --
// 2.1 ms: the registers it holds cost more waves than the overlap wins.)
template <int RW, bool MASK>
__global__ __launch_bounds__(TILE, (RW <= 8 && !(RW == 8 && MASK)) ? 8 : 4) void k_confirm(
    const uint32_t* __restrict__ rd, const uint32_t* __restrict__ rdm,
    const uint32_t* __restrict__ db2, const uint32_t* __restrict__ dbm2, uint64_t r0, uint32_t n, int rw_rt,
    PathParams pp, const uint16_t* __restrict__ nmiss_tab, const uint4* __restrict__ cdesc,
    const uint32_t* __restrict__ rvalid, uint32_t* __restrict__ p_nx,
    const uint32_t* __restrict__ tbase, const uint32_t* __restrict__ tcount,
    const uint32_t* __restrict__ wb, int block_mode, uint32_t block_thr, uint32_t* __restrict__ block_table,
    const uint64_t* __restrict__ seq_off, uint4* __restrict__ stage, uint32_t* __restrict__ tcount2,
    unsigned long long* __restrict__ counters) {
  extern __shared__ uint32_t s_wcnt[];  // TILE * W counters when block_mode != 0
  __shared__ uint32_t s_best[TILE], s_cnt[TILE], s_base[TILE];
  __shared__ uint32_t s_code[CODE_CAP];
  __shared__ uint32_t s_sketch[1 << BLOCK_LDS_BITS];
  __shared__ uint32_t s_wsum[TILE / 64];
  __shared__ uint16_t s_nm[CONF_NM];                 // mismatch budget of the short read lengths
  __shared__ uint32_t s_tb[CONF_TILES], s_tn[CONF_TILES];  // this workgroup's tiles: descriptor range
  const uint32_t ntiles = (n + TILE - 1) / TILE;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (block_mode == 1)
    for (uint32_t t = threadIdx.x; t < (1u << BLOCK_LDS_BITS); t += TILE) s_sketch[t] = 0;
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)pp.max_len ? nmiss_tab[t] : (uint16_t)0;
  // tile j of this workgroup = blockIdx.x + j * gridDim.x (the host keeps it to CONF_TILES)
  const uint32_t my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  for (uint32_t j = threadIdx.x; j < my_tiles && j < CONF_TILES; j += TILE) {
    s_tb[j] = tbase[blockIdx.x + j * gridDim.x];
    s_tn[j] = tcount[blockIdx.x + j * gridDim.x];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) tcount2[ntiles] = 0;
  unsigned long long acc = 0;
  auto budget_of = [&](uint32_t len) -> uint32_t { return len < CONF_NM ? s_nm[len] : nmiss_tab[len]; };
  __syncthreads();

  for (uint32_t j = 0; j < my_tiles; j++) {
    const uint32_t tile = blockIdx.x + j * gridDim.x;
    lds_barrier();  // the previous tile is done with the LDS state
    s_best[threadIdx.x] = 0xFFFFFFFFu;
    s_cnt[threadIdx.x] = 0;
    if (block_mode)
      for (uint32_t t = threadIdx.x; t < TILE * (uint32_t)pp.W; t += TILE) s_wcnt[t] = 0;
    lds_barrier();
    const uint32_t tn = s_tn[j];
    const uint64_t tb = s_tb[j];
    // ---- pass 1.  The lane's first pair (five tiles in six have no second) stays in
    // registers through all three passes; later ones park their result word in LDS.
    auto tally = [&](uint32_t w) {
      if (w == NX_REJECT) return;
      const uint32_t rl = w >> 24;
      if (block_mode) {
        const uint32_t k = (w >> 20) & 15u;
        if (w & NX_ACC0) atomicAdd(&s_wcnt[rl * pp.W + k], 1u);
        if (w & NX_ACC1) atomicAdd(&s_wcnt[rl * pp.W + k + 1], 1u);
      }
      if (w & NX_DUP) return;
      atomicMin(&s_best[rl], w & 0xFFFFu);
      acc++;
    };
    auto confirm_at = [&](uint32_t tj, uint32_t* gene, uint32_t* zword) -> uint32_t {
      // descriptors stream through once: non-temporal
      const u32x4_v dsv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(cdesc) + tb + tj);
      *gene = dsv.w;
      *zword = dsv.z;
      return confirm_pair<RW, MASK>(make_uint4(dsv.x, dsv.y, dsv.z, dsv.w), rd, rdm, db2, dbm2, r0, rw_rt, pp,
                                    budget_of, rvalid);
    };
    uint32_t w0 = NX_REJECT, gene0 = 0, z0 = 0;
    if (threadIdx.x < tn) w0 = confirm_at(threadIdx.x, &gene0, &z0);
    tally(w0);
    for (uint32_t tj = threadIdx.x + TILE; tj < tn; tj += TILE) {
      uint32_t g, z;
      const uint32_t w = confirm_at(tj, &g, &z);
      if (tj - TILE < CODE_CAP) s_code[tj - TILE] = w; else p_nx[tb + tj] = w;
      tally(w);
    }
    lds_barrier();
    // ---- pass 2
    auto count = [&](uint32_t w) {
      if (w == NX_REJECT || (w & NX_DUP)) return;
      const uint32_t rl = w >> 24;
      const uint32_t thr = pp.apply_mmtol ? s_best[rl] + (uint32_t)pp.mmtol : 0xFFFFu;
      if ((w & 0xFFFFu) <= thr) atomicAdd(&s_cnt[rl], 1u);
    };
    count(w0);
    for (uint32_t tj = threadIdx.x + TILE; tj < tn; tj += TILE)
      count(tj - TILE < CODE_CAP ? s_code[tj - TILE] : p_nx[tb + tj]);
    if (block_mode) {
      for (uint32_t t = threadIdx.x; t < TILE * (uint32_t)pp.W; t += TILE) {
        const uint32_t cw = s_wcnt[t];
        if (!cw) continue;
        const uint32_t rl = t / pp.W, k = t % pp.W;
        const uint64_t h = mix64(((uint64_t)k << 32) | wb[((uint64_t)tile * TILE + rl) * pp.W + k]);
        if (block_mode == 1) atomicAdd(&s_sketch[h >> (64 - BLOCK_LDS_BITS)], cw);
        else atomicAdd(&block_table[h >> (64 - BLOCK_TABLE_BITS)], cw);
      }
    }
    lds_barrier();
    // ---- scan of the per-read counts
    const uint32_t c = s_cnt[threadIdx.x];
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) s_wsum[wid] = inc;
    lds_barrier();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int q = 0; q < TILE / 64; q++) {
      if (q < wid) woff += s_wsum[q];
      total += s_wsum[q];
    }
    s_base[threadIdx.x] = woff + inc - c;
    s_cnt[threadIdx.x] = 0;  // now the arrival counter of the read
    if (threadIdx.x == 0) tcount2[tile] = total;
    lds_barrier();
    // ---- pass 3
    if (total == 0) continue;
    auto emit = [&](uint32_t w, uint32_t tj, uint32_t gene, uint32_t zword) {
      if (w == NX_REJECT || (w & NX_DUP)) return;
      const uint32_t rl = w >> 24, v = w & 0xFFFFu;
      const uint32_t thr = pp.apply_mmtol ? s_best[rl] + (uint32_t)pp.mmtol : 0xFFFFu;
      if (v > thr) return;
      const uint32_t ord = atomicAdd(&s_cnt[rl], 1u);
      // position in the target: carried in the descriptor unless the target is so long that
      // the entry's 16-bit distance saturated (then the placement's offset minus the gene's)
      uint32_t pos = (zword >> 6) & 0xFFFFu;
      if (!((zword >> 5) & 1u)) {
        const uint4 ds = cdesc[tb + tj];
        pos = (uint32_t)(((uint64_t)ds.y | ((uint64_t)(ds.x >> 24) << 32)) - seq_off[gene]);
      }
      stage[tb + s_base[rl] + ord] = make_uint4((uint32_t)(r0 + tile * TILE + rl), gene, pos, v);
    };
    emit(w0, threadIdx.x, gene0, z0);
    for (uint32_t tj = threadIdx.x + TILE; tj < tn; tj += TILE) {
      const uint32_t w = tj - TILE < CODE_CAP ? s_code[tj - TILE] : p_nx[tb + tj];
      if (w == NX_REJECT || (w & NX_DUP)) continue;
      const uint4 ds = cdesc[tb + tj];
      emit(w, tj, ds.w, ds.z);
    }
  }
  block_add_u64(acc, &counters[1]);
  if (block_mode == 1) {
    lds_barrier();
    uint32_t hot = 0;
    for (uint32_t t = threadIdx.x; t < (1u << BLOCK_LDS_BITS); t += TILE) hot |= s_sketch[t] >= block_thr;
    if (__any(hot) && (threadIdx.x & 63) == 0) atomicOr(&counters[6], 1ull);
  }
}

// k_compact -- hits[counters[2] + tpre[tile] ...] = the tile's staged tuples (tpre = scan of
// tcount2): plain 16-byte copies, a tile's run is contiguous on both sides.
__global__ __launch_bounds__(256) void k_compact(uint32_t ntiles, const uint32_t* __restrict__ tbase,
                                                 const uint32_t* __restrict__ tcount2,
                                                 const uint32_t* __restrict__ tpre,
                                                 const uint4* __restrict__ stage, uint4* __restrict__ hits,
                                                 uint64_t hits_cap, unsigned long long* __restrict__ counters) {
  const unsigned long long base = counters[2];
  if (base + tpre[ntiles] > hits_cap) {  // cannot happen on a sized pass; a sync-free pass re-runs sized
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&counters[3], 2ull);
    return;
  }
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t m = tcount2[tile];
    const uint4* __restrict__ src = stage + tbase[tile];
    uint4* __restrict__ dst = hits + base + tpre[tile];
    for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) dst[j] = src[j];
  }
}

// After an exact (mode 2) pass: the (read, window) probes whose (window, key) block counter is
// above MaxMatches -- the blocks the reference would have truncated.  One thread per read.
template <int RW>
__global__ __launch_bounds__(256) void k_hot_probes(const uint32_t* __restrict__ rd,
                                                    const uint32_t* __restrict__ rdm, uint64_t nreads,
                                                    int rw_rt, PathParams pp,
                                                    const uint32_t* __restrict__ block_table,
                                                    uint32_t max_matches, uint2* __restrict__ out,
                                                    uint64_t cap, unsigned long long* __restrict__ cursor) {
  const int rw = RW ? RW : rw_rt;
  const bool has_m = rdm != nullptr;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nreads; i += (uint64_t)gridDim.x * blockDim.x) {
    Rec<RW> rec;
    rec.load(rd + i * (uint64_t)rw, rw);
    Rec<RW> recm = rec;
    if (has_m) recm.load(rdm + i * (uint64_t)rw, rw);
    const uint32_t len = rec.len();
    for (int k = 0; k < pp.W; k++) {
      const uint32_t q1 = (uint32_t)pp.win[k], q2 = q1 + (uint32_t)pp.ww;
      if (len < q2) continue;
      if (pp.min_dinuc > 0 && rec_count_dinuc(rec, recm, has_m, q1, pp.ww) < pp.min_dinuc) continue;
      const uint32_t b = rec_bucket(rec, recm, has_m, q1, pp.ww, pp.bits, pp.direct);
      const uint64_t h = mix64(((uint64_t)k << 32) | b);
      if (block_table[h >> (64 - BLOCK_TABLE_BITS)] > max_matches) {
        const unsigned long long slot = atomicAdd(cursor, 1ull);
        if (slot < cap) out[slot] = make_uint2((uint32_t)i, (uint32_t)k);
      }
    }
  }
}

// counters[2] (hits so far) += tpre[ntiles] (hits of this batch)
__global__ void k_advance(const uint32_t* __restrict__ tpre, uint32_t ntiles, unsigned long long* counters) {
  if (threadIdx.x == 0 && blockIdx.x == 0) counters[2] += tpre[ntiles];
}

// number of block counters above MaxMatches (hash collisions only inflate counters, so 0 is
// a proof that no window-key block overflowed)
__global__ void k_block_overflow(const uint32_t* __restrict__ block_table, uint32_t max_matches,
                                 unsigned long long* __restrict__ counters) {
  unsigned long long c = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << BLOCK_TABLE_BITS); i += gridDim.x * blockDim.x)
    c += block_table[i] > max_matches;
  block_add_u64(c, &counters[5]);
}

// ------------------------------------------------------------------------------------
// host side
// Tuples as one u64 each for the wire (RCCL gather to rank 0): read index (+ the shard's base)
// in the top bits, then gene, position, mismatch count with caller-chosen widths; numeric order
// of the words = lexicographic order of the tuples.  *bad is raised if a field does not fit.
struct PackBits {
  int32_t read, gene, pos, nmiss;
};

__global__ __launch_bounds__(256) void k_pack_hits(const uint4* __restrict__ hits, uint64_t n, uint64_t read_base,
                                                   PackBits b, uint64_t* __restrict__ out, uint32_t* __restrict__ bad) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 h = hits[i];
    const uint64_t r = (uint64_t)h.x + read_base;
    if ((b.read < 64 && (r >> b.read)) || ((uint64_t)h.y >> b.gene) || ((uint64_t)h.z >> b.pos) || ((uint64_t)h.w >> b.nmiss))
      atomicOr(bad, 1u);
    out[i] = (((((r << b.gene) | h.y) << b.pos) | h.z) << b.nmiss) | h.w;
  }
}

__global__ __launch_bounds__(256) void k_unpack_hits(const uint64_t* __restrict__ in, uint64_t n, PackBits b,
                                                     uint4* __restrict__ hits) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v = in[i];
    const uint32_t nm = (uint32_t)(v & ((1ull << b.nmiss) - 1ull));
    v >>= b.nmiss;
    const uint32_t pos = (uint32_t)(v & ((1ull << b.pos) - 1ull));
    v >>= b.pos;
    const uint32_t gene = (uint32_t)(v & ((1ull << b.gene) - 1ull));
    v >>= b.gene;
    hits[i] = make_uint4((uint32_t)v, gene, pos, nm);
  }
}

// ------------------------------------------------------------------------------------

namespace {

std::string g_init_error;

template <class T>
struct DevBuf {
  T* p = nullptr;
  uint64_t cap = 0;  // elements
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct musc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // database
  uint32_t* db2 = nullptr;
  uint32_t* dbm2 = nullptr;  // null when the database holds no X
  uint64_t* seq_off = nullptr;
  uint32_t nseq = 0;
  uint64_t nbases = 0;
  uint64_t db_words = 0;

  // index
  int idx_ww = 0, idx_bits = 0, idx_direct = 0;
  int wide = 0;  // database >= 2^32 bases: 40-bit positions, gene numbers < 2^24
  Bucket* idx_T = nullptr;  // 2^idx_bits buckets
  uint4* idx_E = nullptr;   // overflow entries
  uint64_t idx_T_cap = 0, idx_E_cap = 0;  // allocated buckets / entries (kept across rebuilds:
                                          // hipMalloc / hipFree of tens of GiB take seconds)
  uint64_t idx_n = 0;       // indexed window starts
  uint64_t idx_novf = 0;

  // reads
  uint32_t* rd = nullptr;
  uint32_t* rdm = nullptr;  // null when no read holds an X
  uint64_t nreads = 0;
  int rw = 0;
  uint32_t max_len = 0;

  // per-batch work buffers
  // what k_screen hands to k_confirm, twice: in a pipelined pass k_screen fills one set on the
  // screen stream while k_confirm and k_compact drain the other on the confirm stream
  struct BatchSet {
    DevBuf<uint32_t> wb, rvalid, tbase, tcount;
    DevBuf<uint4> cdesc;
  } bs[2];
  int cur = 0;                     // the set the next launches use
  hipStream_t stream2 = nullptr;   // confirm stream of a pipelined pass
  hipStream_t s_confirm = nullptr; // where k_confirm .. k_advance go in the pass in flight
  hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_join = nullptr;
  DevBuf<uint32_t> scan_tmp, tcount2, tpre;
  DevBuf<uint4> stage;
  DevBuf<uint32_t> p_nx;
  DevBuf<uint16_t> nmiss_tab;
  DevBuf<uint32_t> block_table;
  bool force_exact_blocks = false;
  int cur_block_mode = 0;          // of the pass in flight
  uint32_t cur_block_thr = 0;
  PathParams last_pp;          // of the last musc_match_device
  uint32_t last_max_matches = 0;
  bool last_exact_blocks = false;  // block_table holds exact counters of that pass
  unsigned long long* counters = nullptr;  // [0] valid windows [1] accepted [2] hit cursor
  uint64_t* h_pinned = nullptr;            // 16 x u64 pinned staging

  DevBuf<musc_hit> hits;
  uint64_t nhits = 0;
  DevBuf<uint64_t> packed;      // staging of musc_hits_copy_packed / musc_hits_unpack for host pointers
  uint32_t* d_flag = nullptr;   // one device word for kernels that report "does not fit"

  uint32_t batch_reads = 16u << 20;
  // A pass over the same reads, database and parameters as the last completed one needs no
  // sizing: its buffers are known to suffice, so it runs without host round trips.
  uint64_t data_epoch = 1;       // bumped whenever reads or database change
  uint64_t sized_epoch = 0;      // data_epoch of the last completed pass
  musc_params sized_params;      // its parameters
  bool sized_exact_blocks = false;
  uint32_t sized_bsz = 0;        // reads per batch it ended up with
  musc_stats stats;
};

namespace {

int fail(musc_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_init_error = buf;
  return code;
}

#define HIPCHK(c, expr)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail((c), 10, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <class T>
int ensure(musc_ctx* c, DevBuf<T>& b, uint64_t n, bool keep = false) {
  if (n <= b.cap && b.p) return 0;
  uint64_t ncap = std::max<uint64_t>(n, b.cap + b.cap / 2);
  ncap = std::max<uint64_t>(ncap, 1024);
  T* np = nullptr;
  HIPCHK(c, hipMalloc((void**)&np, ncap * sizeof(T) + 64));
  if (keep && b.p && b.cap) {
    hipError_t e = hipMemcpyAsync(np, b.p, b.cap * sizeof(T), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      (void)hipFree(np);
      return fail(c, 10, "hit buffer grow failed: %s", hipGetErrorString(e));
    }
  }
  b.release();
  b.p = np;
  b.cap = ncap;
  return 0;
}

// blocks of b threads covering n elements; a dispatch must stay below 2^32 work-items, callers
// with potentially larger n use grid-stride kernels instead
inline unsigned nblk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

// u32 scan of n elements (in may equal out).  tmp must hold scan_tmp_elems(n).
uint64_t scan_tmp_elems(uint64_t n) {
  uint64_t t = 0;
  while (n > SCAN_TILE) {
    n = (n + SCAN_TILE - 1) / SCAN_TILE;
    t += (n + 8 + 3) & ~3ull;
  }
  return t + 8;
}

int scan_u32(musc_ctx* c, const uint32_t* in, uint32_t* out, uint64_t n, bool inclusive, uint32_t* tmp,
             hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb > 0x7FFFFFFFull) return fail(c, 11, "scan too large");
  if (nb <= 1) {
    if (inclusive) hipLaunchKernelGGL(k_scan_block<true>, dim3(1), dim3(SCAN_BLOCK), 0, st, in, out, (uint32_t*)nullptr, n);
    else hipLaunchKernelGGL(k_scan_block<false>, dim3(1), dim3(SCAN_BLOCK), 0, st, in, out, (uint32_t*)nullptr, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  uint32_t* sums = tmp;
  if (inclusive) hipLaunchKernelGGL(k_scan_block<true>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, out, sums, n);
  else hipLaunchKernelGGL(k_scan_block<false>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, out, sums, n);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u32(c, sums, sums, nb, false, tmp + ((nb + 8 + 3) & ~3ull), st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, out, sums, n);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// exclusive u64 scan (in place allowed); tmp must hold scan_tmp_elems(n) u64
int scan_u64(musc_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* tmp) {
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb > 0x7FFFFFFFull) return fail(c, 11, "scan too large");
  if (nb <= 1) {
    hipLaunchKernelGGL(k_scan64_block, dim3(1), dim3(SCAN_BLOCK), 0, c->stream, in, out, (uint64_t*)nullptr, n);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_scan64_block, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, c->stream, in, out, tmp, n);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u64(c, tmp, tmp, nb, tmp + ((nb + 8 + 3) & ~3ull));
  if (rc) return rc;
  hipLaunchKernelGGL(k_scan64_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, c->stream, out, tmp, n);
  HIPCHK(c, hipGetLastError());
  return 0;
}

void free_index(musc_ctx* c) {
  // (the allocations stay for the next build; musc_destroy releases them)
  c->idx_ww = 0;
  c->idx_n = 0;
  c->data_epoch++;
}

void free_db(musc_ctx* c) {
  free_index(c);
  if (c->db2) (void)hipFree(c->db2);
  if (c->dbm2) (void)hipFree(c->dbm2);
  if (c->seq_off) (void)hipFree(c->seq_off);
  c->db2 = c->dbm2 = nullptr;
  c->seq_off = nullptr;
  c->nseq = 0;
  c->nbases = 0;
  c->data_epoch++;
}

void free_reads(musc_ctx* c) {
  if (c->rd) (void)hipFree(c->rd);
  if (c->rdm) (void)hipFree(c->rdm);
  c->rd = c->rdm = nullptr;
  c->nreads = 0;
  c->rw = 0;
  c->data_epoch++;
}

struct EvPair {
  hipEvent_t a, b;
};

struct Timer {
  musc_ctx* c;
  std::vector<EvPair> ev[5];
  explicit Timer(musc_ctx* ctx) : c(ctx) {}
  int begin(int fam, hipStream_t s = nullptr) {
    EvPair p;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return 1;
    (void)hipEventRecord(p.a, s ? s : c->stream);
    ev[fam].push_back(p);
    return 0;
  }
  void end(int fam, hipStream_t s = nullptr) { (void)hipEventRecord(ev[fam].back().b, s ? s : c->stream); }
  float total(int fam) {
    float t = 0;
    for (auto& p : ev[fam]) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) t += ms;
    }
    return t;
  }
  ~Timer() {
    for (auto& v : ev)
      for (auto& p : v) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
      }
  }
};

int check_params(musc_ctx* c, const musc_params* P) {
  if (!P) return fail(c, 2, "params is NULL");
  if (P->n_windows < 1 || P->n_windows > MUSC_MAX_WINDOWS)
    return fail(c, 2, "n_windows=%d outside 1..%d", P->n_windows, MUSC_MAX_WINDOWS);
  if (P->window_width < 1 || P->window_width > 4096) return fail(c, 2, "bad window_width=%d", P->window_width);
  for (int k = 0; k < P->n_windows; k++)
    if (P->windows[k] < 0 || P->windows[k] > 60000) return fail(c, 2, "bad window start %d", P->windows[k]);
  if (!(P->pmatch >= 0.0 && P->pmatch <= 1.0)) return fail(c, 2, "PMatch=%g outside [0,1]", P->pmatch);
  if (P->match_mode != 0 && P->match_mode != 1) return fail(c, 2, "match_mode must be 0 (best) or 1 (first)");
  if (P->mmtol < 0) return fail(c, 2, "MMTol < 0");
  if (P->max_mismatch_p1 < 0) return fail(c, 2, "max_mismatch_p1 < 0");
  return 0;
}

template <int RW>
void launch_path(musc_ctx* c, int stage, bool mask, uint64_t r0, uint32_t n, const PathParams& pp) {
  const dim3 block(TILE);
  if (stage == 0) {
    if (c->rdm)
      hipLaunchKernelGGL((k_screen<RW, true>), dim3(std::min(nblk(n, TILE), MAX_GRID)), block, 0, c->stream, c->rd,
                         c->rdm, r0, n, c->rw, pp, c->nmiss_tab.p, c->idx_T, c->idx_E, c->bs[c->cur].cdesc.p, c->bs[c->cur].cdesc.cap,
                         c->bs[c->cur].rvalid.p, c->bs[c->cur].wb.p, c->bs[c->cur].tbase.p, c->bs[c->cur].tcount.p, c->counters + 8);
    else
      hipLaunchKernelGGL((k_screen<RW, false>), dim3(std::min(nblk(n, TILE), MAX_GRID)), block, 0, c->stream, c->rd,
                         c->rdm, r0, n, c->rw, pp, c->nmiss_tab.p, c->idx_T, c->idx_E, c->bs[c->cur].cdesc.p, c->bs[c->cur].cdesc.cap,
                         c->bs[c->cur].rvalid.p, c->bs[c->cur].wb.p, c->bs[c->cur].tbase.p, c->bs[c->cur].tcount.p, c->counters + 8);
  } else {
    // persistent over tiles; the MaxMatches screening threshold assumes at most MAX_GRID workgroups
    const dim3 grid(std::min(nblk(n, TILE), MAX_GRID));
    static_assert((1u << 24) / TILE / MAX_GRID <= CONF_TILES, "a k_confirm workgroup keeps its tile list in LDS");
    const size_t lds = c->cur_block_mode ? (size_t)TILE * pp.W * 4 : 0;
    if (mask)
      hipLaunchKernelGGL((k_confirm<RW, true>), grid, block, lds, c->s_confirm, c->rd, c->rdm, c->db2, c->dbm2, r0, n,
                         c->rw, pp, c->nmiss_tab.p, c->bs[c->cur].cdesc.p, c->bs[c->cur].rvalid.p, c->p_nx.p, c->bs[c->cur].tbase.p, c->bs[c->cur].tcount.p,
                         c->bs[c->cur].wb.p, c->cur_block_mode, c->cur_block_thr, c->block_table.p, c->seq_off, c->stage.p,
                         c->tcount2.p, c->counters);
    else
      hipLaunchKernelGGL((k_confirm<RW, false>), grid, block, lds, c->s_confirm, c->rd, c->rdm, c->db2, c->dbm2, r0, n,
                         c->rw, pp, c->nmiss_tab.p, c->bs[c->cur].cdesc.p, c->bs[c->cur].rvalid.p, c->p_nx.p, c->bs[c->cur].tbase.p, c->bs[c->cur].tcount.p,
                         c->bs[c->cur].wb.p, c->cur_block_mode, c->cur_block_thr, c->block_table.p, c->seq_off, c->stage.p,
                         c->tcount2.p, c->counters);
  }
}

void launch_stage(musc_ctx* c, int stage, bool mask, uint64_t r0, uint32_t n, const PathParams& pp) {
  switch (c->rw) {
    case 4: launch_path<4>(c, stage, mask, r0, n, pp); break;
    case 8: launch_path<8>(c, stage, mask, r0, n, pp); break;
    case 12: launch_path<12>(c, stage, mask, r0, n, pp); break;
    case 16: launch_path<16>(c, stage, mask, r0, n, pp); break;
    default: launch_path<0>(c, stage, mask, r0, n, pp); break;
  }
}

}  // namespace

extern "C" {

int musc_abi_version(void) { return MUSC_ABI_VERSION; }

const char* musc_last_error(musc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int musc_init(int device_ordinal, musc_ctx** out) {
  if (!out) return fail(nullptr, 2, "musc_init: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, 3, "musc_init: no HIP device (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= ndev)
    return fail(nullptr, 2, "musc_init: device %d outside 0..%d", device_ordinal, ndev - 1);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess)
    return fail(nullptr, 3, "musc_init: hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, 3, "musc_init: device %d is %s; this library is built for gfx950 only",
                device_ordinal, prop.gcnArchName);
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess)
    return fail(nullptr, 3, "musc_init: hipSetDevice: %s", hipGetErrorString(e));
  musc_ctx* c = new musc_ctx();
  c->device = device_ordinal;
  memset(&c->stats, 0, sizeof c->stats);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_ready[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_ready[1], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_free[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_free[1], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess ||
      (e = hipMalloc((void**)&c->counters, 16 * sizeof(unsigned long long))) != hipSuccess ||
      (e = hipHostMalloc((void**)&c->h_pinned, 16 * sizeof(uint64_t))) != hipSuccess) {
    fail(nullptr, 3, "musc_init: %s", hipGetErrorString(e));
    musc_destroy(c);
    return 3;
  }
  if (const char* br = getenv("MUSC_BATCH_READS")) {  // tests: many small batches
    const long v = atol(br);
    if (v >= 1 && v <= (16l << 20)) c->batch_reads = (uint32_t)v;
  }
  *out = c;
  return 0;
}

void musc_destroy(musc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  free_db(c);
  free_reads(c);
  if (c->idx_T) (void)hipFree(c->idx_T);
  if (c->idx_E) (void)hipFree(c->idx_E);
  for (int i = 0; i < 2; i++) {
    c->bs[i].wb.release(); c->bs[i].tbase.release(); c->bs[i].rvalid.release(); c->bs[i].tcount.release();
    c->bs[i].cdesc.release();
  }
  c->scan_tmp.release(); c->tcount2.release(); c->tpre.release(); c->stage.release();
  c->p_nx.release();
  c->nmiss_tab.release();
  c->block_table.release();
  c->hits.release();
  c->packed.release();
  if (c->d_flag) (void)hipFree(c->d_flag);
  if (c->counters) (void)hipFree(c->counters);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  for (hipEvent_t ev : {c->ev_ready[0], c->ev_ready[1], c->ev_free[0], c->ev_free[1], c->ev_join})
    if (ev) (void)hipEventDestroy(ev);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// ---------------------------------------------------------------- database

static int db_finish(musc_ctx* c, uint32_t* d_hasx) {
  uint32_t hasx = 0;
  HIPCHK(c, hipMemcpyAsync(&hasx, d_hasx, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_hasx);
  if (!hasx) {
    (void)hipFree(c->dbm2);
    c->dbm2 = nullptr;
  }
  return 0;
}

static int db_alloc(musc_ctx* c, const uint64_t* offsets, uint32_t nseq, int on_device, uint32_t** d_hasx) {
  HIPCHK(c, hipSetDevice(c->device));
  free_db(c);
  if (nseq == 0) return fail(c, 2, "database has no sequences");
  uint64_t first = 0, last = 0;
  if (on_device) {
    HIPCHK(c, hipMemcpy(&first, offsets, 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&last, offsets + nseq, 8, hipMemcpyDeviceToHost));
  } else {
    first = offsets[0];
    last = offsets[nseq];
  }
  if (first != 0) return fail(c, 2, "offsets[0] must be 0");
  if (last >= (1ull << 36)) return fail(c, 2, "database larger than 2^36 bases");
  c->nseq = nseq;
  c->nbases = last;
  c->db_words = (last + 15) / 16;
  const uint64_t alloc_words = c->db_words + 64;  // slack: k_confirm reads a whole span past the last base
  HIPCHK(c, hipMalloc((void**)&c->db2, alloc_words * 4));
  HIPCHK(c, hipMalloc((void**)&c->dbm2, alloc_words * 4));
  HIPCHK(c, hipMemsetAsync(c->db2, 0, alloc_words * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->dbm2, 0, alloc_words * 4, c->stream));
  HIPCHK(c, hipMalloc((void**)&c->seq_off, ((uint64_t)nseq + 1) * 8));
  HIPCHK(c, hipMemcpyAsync(c->seq_off, offsets, ((uint64_t)nseq + 1) * 8,
                           on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMalloc((void**)d_hasx, 4));
  HIPCHK(c, hipMemsetAsync(*d_hasx, 0, 4, c->stream));
  return 0;
}

int musc_db_load_ascii(musc_ctx* c, const char* seqs, const uint64_t* offsets, uint32_t nseq, int on_device) {
  if (!c) return 1;
  if (!seqs || !offsets) return fail(c, 2, "musc_db_load_ascii: NULL input");
  uint32_t* d_hasx = nullptr;
  int rc = db_alloc(c, offsets, nseq, on_device, &d_hasx);
  if (rc) return rc;
  const unsigned char* src = (const unsigned char*)seqs;
  unsigned char* tmp = nullptr;
  if (!on_device && c->nbases) {
    HIPCHK(c, hipMalloc((void**)&tmp, c->nbases));
    HIPCHK(c, hipMemcpyAsync(tmp, seqs, c->nbases, hipMemcpyHostToDevice, c->stream));
    src = tmp;
  }
  if (c->db_words) {
    hipLaunchKernelGGL(k_pack_db_ascii, dim3(nblk(c->db_words, 256)), dim3(256), 0, c->stream, src, c->nbases,
                       c->db2, c->dbm2, c->db_words, d_hasx);
    HIPCHK(c, hipGetLastError());
  }
  rc = db_finish(c, d_hasx);
  if (tmp) (void)hipFree(tmp);
  return rc;
}

int musc_db_load_packed(musc_ctx* c, const uint8_t* bases2bit, const uint8_t* nmask, const uint64_t* seq_offsets,
                        uint32_t nseq) {
  if (!c) return 1;
  if (!bases2bit || !seq_offsets) return fail(c, 2, "musc_db_load_packed: NULL input");
  uint32_t* d_hasx = nullptr;
  int rc = db_alloc(c, seq_offsets, nseq, 0, &d_hasx);
  if (rc) return rc;
  uint32_t* t2 = nullptr;
  uint16_t* tm = nullptr;
  const uint64_t w = c->db_words;
  if (w) {
    HIPCHK(c, hipMalloc((void**)&t2, w * 4));
    HIPCHK(c, hipMemsetAsync(t2, 0, w * 4, c->stream));
    HIPCHK(c, hipMemcpyAsync(t2, bases2bit, (c->nbases + 3) / 4, hipMemcpyHostToDevice, c->stream));
    if (nmask) {
      HIPCHK(c, hipMalloc((void**)&tm, w * 2));
      HIPCHK(c, hipMemsetAsync(tm, 0, w * 2, c->stream));
      HIPCHK(c, hipMemcpyAsync(tm, nmask, (c->nbases + 7) / 8, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_pack_db_packed, dim3(nblk(w, 256)), dim3(256), 0, c->stream, t2, tm, c->db2, c->dbm2, w,
                       d_hasx);
    HIPCHK(c, hipGetLastError());
  }
  rc = db_finish(c, d_hasx);
  if (t2) (void)hipFree(t2);
  if (tm) (void)hipFree(tm);
  return rc;
}

int musc_db_build_index(musc_ctx* c, int32_t ww) {
  if (!c) return 1;
  if (!c->db2) return fail(c, 4, "no database loaded");
  if (ww < 1 || ww > 4096) return fail(c, 2, "bad window width %d", ww);
  HIPCHK(c, hipSetDevice(c->device));
  if (c->idx_ww == ww && c->idx_T) return 0;
  free_index(c);
  c->wide = c->nbases >= 0xFFFFFFF0ull || getenv("MUSC_DEBUG_FORCE_WIDE") != nullptr;
  if (c->wide && c->nseq >= (1u << 24))
    return fail(c, 5, "a database of 2^32 bases or more may hold at most 2^24 targets (has %u)", c->nseq);
  // Direct addressing (bucket = the 2*ww-bit key itself: exact, and bytewise-sorted reads walk
  // the table front to back) when that table is at most 32x the database and at most 2^30
  // buckets (64 GiB); otherwise a hashed table with about one bucket per base, at most 2^31
  // buckets (128 GiB; longer lists go to the overflow array).
  int bits, direct = 0;
  const uint64_t floor_bases = std::max<uint64_t>(c->nbases, 1ull << 19);
  if (2 * ww <= 30 && (1ull << (2 * ww)) <= 32 * floor_bases) {
    bits = 2 * ww;
    direct = 1;
  } else {
    bits = 10;
    while (bits < 31 && (1ull << bits) < c->nbases) bits++;
  }
  if (const char* ov = getenv("MUSC_DEBUG_INDEX_BITS")) {  // experiments only: force a hashed table size
    const int v = atoi(ov);
    if (v >= 8 && v <= 31) { bits = v; direct = 0; }
  }
  const uint64_t nb = 1ull << bits;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  if (c->idx_T_cap < nb + 1) {
    if (c->idx_T) (void)hipFree(c->idx_T);
    c->idx_T = nullptr;
    c->idx_T_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->idx_T, (nb + 1) * sizeof(Bucket)));
    c->idx_T_cap = nb + 1;
  }
  uint64_t *tmp = nullptr, *stmp = nullptr;
  HIPCHK(c, hipMalloc((void**)&tmp, (nb + 1 + 16) * 8));
  HIPCHK(c, hipMalloc((void**)&stmp, scan_tmp_elems(nb + 1) * 8));
  // timed: the device work (allocation above and below is host time, seconds for a 64 GiB table
  // the first time, and not repeated)
  HIPCHK(c, hipEventRecord(e0, c->stream));
  HIPCHK(c, hipMemsetAsync(c->idx_T, 0, (nb + 1) * sizeof(Bucket), c->stream));
  const unsigned blocks = (unsigned)std::min<uint64_t>((c->nbases + 255) / 256, 1u << 22);
  if (c->nbases) {
    hipLaunchKernelGGL(k_index<false>, dim3(blocks), dim3(256), 0, c->stream, c->db2, c->dbm2, c->seq_off, c->nseq,
                       c->nbases, ww, bits, direct, c->wide, c->idx_T, (uint4*)nullptr);
    HIPCHK(c, hipGetLastError());
  }
  // overflow lists: sizes -> offsets (u64: a 10 Gbp database has billions of overflow entries)
  hipLaunchKernelGGL(k_index_ovf_count, dim3(nblk(nb + 1, 256)), dim3(256), 0, c->stream, c->idx_T, nb, tmp);
  HIPCHK(c, hipGetLastError());
  int rc = scan_u64(c, tmp, tmp, nb + 1, stmp);
  if (rc) return rc;
  uint64_t novf = 0;
  HIPCHK(c, hipMemcpyAsync(&novf, tmp + nb, 8, hipMemcpyDeviceToHost, c->stream));
  hipLaunchKernelGGL(k_index_ovf_set, dim3(nblk(nb, 256)), dim3(256), 0, c->stream, c->idx_T, nb, tmp);
  HIPCHK(c, hipGetLastError());
  hipEvent_t e2, e3;
  HIPCHK(c, hipEventCreate(&e2));
  HIPCHK(c, hipEventCreate(&e3));
  HIPCHK(c, hipEventRecord(e2, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(tmp);
  (void)hipFree(stmp);
  c->idx_novf = novf;
  if (c->idx_E_cap < novf + 16) {
    if (c->idx_E) (void)hipFree(c->idx_E);
    c->idx_E = nullptr;
    c->idx_E_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->idx_E, ((uint64_t)novf + 16) * sizeof(uint4)));
    c->idx_E_cap = novf + 16;
  }
  HIPCHK(c, hipEventRecord(e3, c->stream));
  if (c->nbases) {
    hipLaunchKernelGGL(k_index<true>, dim3(blocks), dim3(256), 0, c->stream, c->db2, c->dbm2, c->seq_off, c->nseq,
                       c->nbases, ww, bits, direct, c->wide, c->idx_T, c->idx_E);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0, ms2 = 0;
  (void)hipEventElapsedTime(&ms, e0, e2);
  (void)hipEventElapsedTime(&ms2, e3, e1);
  for (hipEvent_t ev : {e0, e1, e2, e3}) (void)hipEventDestroy(ev);
  c->stats.ms_index_build = ms + ms2;
  c->idx_ww = ww;
  c->idx_bits = bits;
  c->idx_direct = direct;
  return 0;
}

// ---------------------------------------------------------------- reads

static int reads_load(musc_ctx* c, const unsigned char* ascii, const uint8_t* bases2bit, const uint8_t* nmask,
                      const uint64_t* offsets, uint64_t nreads, int on_device, bool packed) {
  HIPCHK(c, hipSetDevice(c->device));
  free_reads(c);
  if (nreads >= 0xFFFFFFF0ull) return fail(c, 2, "too many reads for 32-bit read_idx");
  c->nreads = nreads;
  if (nreads == 0) {
    c->rw = 4;
    return 0;
  }
  uint64_t* d_off = nullptr;
  const uint64_t* offp = offsets;
  if (!on_device) {
    HIPCHK(c, hipMalloc((void**)&d_off, (nreads + 1) * 8));
    HIPCHK(c, hipMemcpyAsync(d_off, offsets, (nreads + 1) * 8, hipMemcpyHostToDevice, c->stream));
    offp = d_off;
  }
  HIPCHK(c, hipMemsetAsync(c->counters + 4, 0, 8, c->stream));
  hipLaunchKernelGGL(k_max_len, dim3(std::min(nblk(nreads, 256), MAX_GRID)), dim3(256), 0, c->stream, offp, nreads, c->counters + 4);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->counters + 4, 8, hipMemcpyDeviceToHost, c->stream));
  uint64_t first = 0, total = 0;
  HIPCHK(c, hipMemcpyAsync(&first, offp, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&total, offp + nreads, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t maxlen = c->h_pinned[0];
  if (first != 0) return fail(c, 2, "read offsets[0] must be 0");
  if (maxlen > 65535) return fail(c, 2, "read of %llu bases exceeds the 65535-base record limit", (unsigned long long)maxlen);
  c->max_len = (uint32_t)maxlen;
  int rw = (int)((2 * maxlen + 31) / 32) + 1;
  rw = (rw + 3) & ~3;
  if (rw < 4) rw = 4;
  c->rw = rw;
  const uint64_t words = nreads * (uint64_t)rw;
  HIPCHK(c, hipMalloc((void**)&c->rd, words * 4 + 256));
  HIPCHK(c, hipMalloc((void**)&c->rdm, words * 4 + 256));
  HIPCHK(c, hipMemsetAsync(c->rd + words, 0, 256, c->stream));
  HIPCHK(c, hipMemsetAsync(c->rdm + words, 0, 256, c->stream));
  uint32_t* d_hasx = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_hasx, 4));
  HIPCHK(c, hipMemsetAsync(d_hasx, 0, 4, c->stream));
  void *t1 = nullptr, *t2 = nullptr;
  if (words >= (1ull << 32)) return fail(c, 2, "too many read words for one dispatch (reads x record words >= 2^32)");
  if (!packed) {
    const unsigned char* src = ascii;
    if (!on_device && total) {
      HIPCHK(c, hipMalloc(&t1, total));
      HIPCHK(c, hipMemcpyAsync(t1, ascii, total, hipMemcpyHostToDevice, c->stream));
      src = (const unsigned char*)t1;
    }
    hipLaunchKernelGGL(k_pack_reads<false>, dim3(nblk(words, 256)), dim3(256), 0, c->stream, src,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, offp, nreads, rw, c->rd, c->rdm, d_hasx);
  } else {
    const uint64_t b2 = ((total + 3) / 4 + 7) & ~3ull, bm = ((total + 7) / 8 + 7) & ~3ull;
    HIPCHK(c, hipMalloc(&t1, b2 + 16));
    HIPCHK(c, hipMemsetAsync(t1, 0, b2 + 16, c->stream));
    HIPCHK(c, hipMemcpyAsync(t1, bases2bit, (total + 3) / 4, hipMemcpyHostToDevice, c->stream));
    if (nmask) {
      HIPCHK(c, hipMalloc(&t2, bm + 16));
      HIPCHK(c, hipMemsetAsync(t2, 0, bm + 16, c->stream));
      HIPCHK(c, hipMemcpyAsync(t2, nmask, (total + 7) / 8, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_pack_reads<true>, dim3(nblk(words, 256)), dim3(256), 0, c->stream,
                       (const unsigned char*)nullptr, (const uint32_t*)t1, (const uint32_t*)t2, offp, nreads, rw,
                       c->rd, c->rdm, d_hasx);
  }
  HIPCHK(c, hipGetLastError());
  uint32_t hasx = 0;
  HIPCHK(c, hipMemcpyAsync(&hasx, d_hasx, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_hasx);
  if (t1) (void)hipFree(t1);
  if (t2) (void)hipFree(t2);
  if (d_off) (void)hipFree(d_off);
  if (!hasx) {
    (void)hipFree(c->rdm);
    c->rdm = nullptr;
  }
  return 0;
}

int musc_reads_load_ascii(musc_ctx* c, const char* seqs, const uint64_t* offsets, uint64_t nreads, int on_device) {
  if (!c) return 1;
  if ((!seqs || !offsets) && nreads) return fail(c, 2, "musc_reads_load_ascii: NULL input");
  return reads_load(c, (const unsigned char*)seqs, nullptr, nullptr, offsets, nreads, on_device, false);
}

int musc_reads_load_packed(musc_ctx* c, const uint8_t* bases2bit, const uint8_t* nmask, const uint64_t* read_offsets,
                           uint64_t nreads) {
  if (!c) return 1;
  if ((!bases2bit || !read_offsets) && nreads) return fail(c, 2, "musc_reads_load_packed: NULL input");
  return reads_load(c, nullptr, bases2bit, nmask, read_offsets, nreads, 0, true);
}

// ---------------------------------------------------------------- hot path

int musc_match_device(musc_ctx* c, const musc_params* P, uint64_t* nhits) {
  if (!c) return 1;
  int rc = check_params(c, P);
  if (rc) return rc;
  if (!c->db2) return fail(c, 4, "no database loaded");
  if (!c->rd && c->nreads) return fail(c, 4, "no reads loaded");
  HIPCHK(c, hipSetDevice(c->device));
  rc = musc_db_build_index(c, P->window_width);
  if (rc) return rc;

  const float keep_index_ms = c->stats.ms_index_build;
  memset(&c->stats, 0, sizeof c->stats);
  c->stats.ms_index_build = keep_index_ms;
  c->stats.n_reads = c->nreads;
  c->nhits = 0;
  if (nhits) *nhits = 0;

  PathParams pp;
  memset(&pp, 0, sizeof pp);
  pp.W = P->n_windows;
  pp.ww = P->window_width;
  pp.min_dinuc = P->min_dinuc;
  pp.bits = c->idx_bits;
  pp.direct = c->idx_direct;
  pp.mmtol = P->mmtol > 0xFFFF ? 0xFFFF : P->mmtol;
  pp.apply_mmtol = P->apply_mmtol;
  pp.wide = c->wide;
  pp.max_len = (int32_t)c->max_len;
  if (const char* dv = getenv("MUSC_DEBUG_SCREEN")) pp.dbg = atoi(dv);
  for (int k = 0; k < pp.W; k++) {
    pp.win[k] = P->windows[k];
    if (P->windows[k] == 0) pp.q1zero_mask |= 1u << k;
  }

  // nmiss budget per read length: int((1-PMatch)*float64(len)), IEEE double, truncation
  // (cmd/muscato_confirm/main.go:198) -- evaluated on the host exactly as Go does.
  {
    std::vector<uint16_t> tab((size_t)c->max_len + 2);
    for (uint32_t L = 0; L < tab.size(); L++) {
      volatile double a = 1.0 - P->pmatch;
      volatile double b = a * (double)L;
      long long v = (long long)b;
      if (P->max_mismatch_p1 > 0) v = P->max_mismatch_p1 - 1;  // --MaxMismatch addition
      if (v < 0) v = 0;
      if (v > 0xFFFE) v = 0xFFFE;
      tab[L] = (uint16_t)v;
    }
    rc = ensure(c, c->nmiss_tab, tab.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->nmiss_tab.p, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // tab goes out of scope
  }

  HIPCHK(c, hipMemsetAsync(c->counters, 0, 8 * sizeof(unsigned long long), c->stream));
  // MaxMatches accounting (see k_confirm): screening first, exact only if inconclusive
  const uint64_t planned_batches = (c->nreads + c->batch_reads - 1) / c->batch_reads + 1;
  uint64_t max_matches = P->max_matches > 0 ? (uint64_t)P->max_matches : 0x7FFFFFFFull;
  if (P->n_shards > 1) max_matches /= (uint64_t)P->n_shards;  // this context sees one shard of each block
  const uint32_t block_thr = (uint32_t)std::min<uint64_t>(max_matches / (planned_batches * MAX_GRID), 0x7FFFFFFFull);
  int block_mode = P->skip_block_check ? 0 : (c->force_exact_blocks || block_thr < 2 ? 2 : 1);
  const bool check_blocks = block_mode != 0;
  c->cur_block_mode = block_mode;
  c->cur_block_thr = block_thr;
  if (block_mode == 2) {
    if ((rc = ensure(c, c->block_table, 1ull << BLOCK_TABLE_BITS))) return rc;
    HIPCHK(c, hipMemsetAsync(c->block_table.p, 0, (1ull << BLOCK_TABLE_BITS) * 4, c->stream));
  }
  Timer tm(c);
  hipEvent_t ev0, ev1;
  HIPCHK(c, hipEventCreate(&ev0));
  HIPCHK(c, hipEventCreate(&ev1));
  HIPCHK(c, hipEventRecord(ev0, c->stream));

  const bool mask = c->rdm || c->dbm2;
  // a mask plane on only one side: allocate the missing all-zero plane once
  if (mask && !c->rdm && c->nreads) {
    const uint64_t words = c->nreads * (uint64_t)c->rw;
    HIPCHK(c, hipMalloc((void**)&c->rdm, words * 4 + 256));
    HIPCHK(c, hipMemsetAsync(c->rdm, 0, words * 4 + 256, c->stream));
  }
  if (mask && !c->dbm2) {
    HIPCHK(c, hipMalloc((void**)&c->dbm2, (c->db_words + 64) * 4));
    // (the index stays valid: bucket_of treats a null and an all-zero mask plane alike)
    HIPCHK(c, hipMemsetAsync(c->dbm2, 0, (c->db_words + 64) * 4, c->stream));
  }

  // Per batch: k_screen claims descriptor space as it goes; if a batch needs more than the
  // buffer holds it reports how much and is repeated after growing the buffer (the first pass
  // of a workload sizes it).  A pass over the same reads, database and parameters as the last
  // completed one ("sized") is known to fit and runs without any host round trip; every kernel
  // still guards its writes, and the flags are checked once at the end.
  const bool sized = c->sized_epoch == c->data_epoch && c->sized_exact_blocks == (block_mode == 2) &&
                     memcmp(&c->sized_params, P, sizeof *P) == 0 && !getenv("MUSC_DEBUG_SYNC");
  uint64_t n_cand = 0, n_pairs = 0, n_windows = 0, n_two = 0;
  const uint64_t PAIR_CAP = 1ull << 31;  // u32 descriptor offsets
  uint64_t r0 = 0;
  uint32_t bsz = sized ? c->sized_bsz : c->batch_reads;
  if (sized) HIPCHK(c, hipMemsetAsync(c->counters + 8, 0, 8 * sizeof(unsigned long long), c->stream));
  // MUSC_PIPELINE=1: a sized pass of several batches is pipelined over two streams, k_screen of
  // batch b+1 beside k_confirm + k_compact of batch b, alternating between the two batch sets.
  // Off by default: measured on cfg3 / cfg4 / cfg5 shards the pass moves ~5.6 TB/s of cache lines
  // through HBM either way (both kernels are bound by the lines they fetch), so overlapping them
  // gains nothing (5.66 vs 5.44 ms on cfg3) and the second set costs memory.
  const char* pipe_env = getenv("MUSC_PIPELINE");
  const bool want_pipe = pipe_env && atoi(pipe_env) > 0;
  const bool piped = sized && want_pipe && c->nreads > bsz && c->bs[1].cdesc.cap >= c->bs[0].cdesc.cap;
  hipStream_t sA = c->stream, sB = piped ? c->stream2 : c->stream;
  c->s_confirm = sB;
  c->cur = 0;
  if (piped) {  // the confirm stream starts after the memsets above
    HIPCHK(c, hipEventRecord(c->ev_join, sA));
    HIPCHK(c, hipStreamWaitEvent(sB, c->ev_join, 0));
  }
  uint32_t batch_no = 0;
  while (r0 < c->nreads) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(bsz, c->nreads - r0);
    if (piped) {
      c->cur = (int)(batch_no & 1u);
      // the set is free once the batch before last has been compacted
      if (batch_no >= 2) HIPCHK(c, hipStreamWaitEvent(sA, c->ev_free[c->cur], 0));
    }
    const int W = pp.W;
    const uint32_t ntiles = nblk(n, TILE);
    uint64_t total = 1;  // pairs of this batch (unknown on a sized pass)
    if (!sized) {
      if ((rc = ensure(c, c->bs[c->cur].wb, (uint64_t)n * W))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].rvalid, (uint64_t)n + 1))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].tbase, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].tcount, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->scan_tmp, scan_tmp_elems((uint64_t)ntiles + 1)))) return rc;
      if ((rc = ensure(c, c->tcount2, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->tpre, (uint64_t)ntiles + 1))) return rc;
      if ((rc = ensure(c, c->bs[c->cur].cdesc, std::max<uint64_t>(4ull * n, 1024)))) return rc;
      // batch-local counters: [0] valid windows [3] candidates [4] pairs [7] descriptor cursor
      HIPCHK(c, hipMemsetAsync(c->counters + 8, 0, 8 * sizeof(unsigned long long), c->stream));
    }

    tm.begin(0);
    launch_stage(c, 0, mask, r0, n, pp);
    HIPCHK(c, hipGetLastError());
    tm.end(0);
    if (piped) {
      HIPCHK(c, hipEventRecord(c->ev_ready[c->cur], sA));
      HIPCHK(c, hipStreamWaitEvent(sB, c->ev_ready[c->cur], 0));
    }
    if (!sized) {
      HIPCHK(c, hipMemcpyAsync(&c->h_pinned[0], c->counters + 8, 8 * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(&c->h_pinned[8], c->counters + 2, 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      total = c->h_pinned[4];
      const uint64_t hits_so_far = c->h_pinned[8];
      const uint64_t sgrid = std::min(nblk(n, TILE), MAX_GRID);
      const uint64_t need = c->h_pinned[7] * sgrid;  // every workgroup region as large as the fullest
      if (need > PAIR_CAP) {
        // too many pairs for one launch: retry this range with half the reads
        if (n == 1) return fail(c, 6, "one read has %llu candidate pairs (> 2^31)", (unsigned long long)total);
        bsz = n / 2;
        HIPCHK(c, hipMemsetAsync(c->counters + 3, 0, 8, c->stream));
        continue;
      }
      if (c->h_pinned[7] > c->bs[c->cur].cdesc.cap / sgrid) {
        if ((rc = ensure(c, c->bs[c->cur].cdesc, need + need / 8 + sgrid))) return rc;
        HIPCHK(c, hipMemsetAsync(c->counters + 3, 0, 8, c->stream));
        continue;  // repeat the batch with room for every workgroup's pairs
      }
      n_windows += c->h_pinned[0];
      n_cand += c->h_pinned[3];
      n_pairs += c->h_pinned[4];
      n_two += c->h_pinned[5];
      if ((rc = ensure(c, c->p_nx, c->bs[c->cur].cdesc.cap))) return rc;
      if ((rc = ensure(c, c->stage, c->bs[c->cur].cdesc.cap))) return rc;
      if ((rc = ensure(c, c->hits, hits_so_far + total, true))) return rc;
    }
    c->stats.n_batches++;

    if (total) {
      const dim3 sg(std::min(nblk(n, TILE), MAX_GRID));

      tm.begin(3, sB);
      launch_stage(c, 2, mask, r0, n, pp);
      HIPCHK(c, hipGetLastError());
      tm.end(3, sB);
      c->stats.confirm_launches++;

      tm.begin(4, sB);
      tm.begin(1, sB);
      rc = scan_u32(c, c->tcount2.p, c->tpre.p, (uint64_t)ntiles + 1, false, c->scan_tmp.p, sB);
      if (rc) return rc;
      tm.end(1, sB);
      hipLaunchKernelGGL(k_compact, sg, dim3(256), 0, sB, ntiles, c->bs[c->cur].tbase.p, c->tcount2.p, c->tpre.p,
                         c->stage.p, reinterpret_cast<uint4*>(c->hits.p), c->hits.cap, c->counters);
      HIPCHK(c, hipGetLastError());
      hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, sB, c->tpre.p, ntiles, c->counters);
      HIPCHK(c, hipGetLastError());
      tm.end(4, sB);
    }
    if (piped) HIPCHK(c, hipEventRecord(c->ev_free[c->cur], sB));
    r0 += n;
    batch_no++;
  }
  if (piped) {  // join: everything below is ordered after both streams
    HIPCHK(c, hipEventRecord(c->ev_join, sB));
    HIPCHK(c, hipStreamWaitEvent(sA, c->ev_join, 0));
  }
  c->cur = 0;
  c->s_confirm = c->stream;
  c->last_pp = pp;
  c->last_max_matches = (uint32_t)max_matches;
  c->last_exact_blocks = block_mode == 2;
  if (block_mode == 2) {
    hipLaunchKernelGGL(k_block_overflow, dim3(1024), dim3(256), 0, c->stream, c->block_table.p,
                       (uint32_t)max_matches, c->counters);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipEventRecord(ev1, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->counters, 16 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (sized) {
    if (c->h_pinned[3]) {
      // a guard fired: the pass did not fit after all -- forget the sizing and run it the careful way
      c->sized_epoch = 0;
      (void)hipEventDestroy(ev0);
      (void)hipEventDestroy(ev1);
      return musc_match_device(c, P, nhits);
    }
    n_windows = c->h_pinned[8];  // the batch-local block accumulated over the whole pass
    n_cand = c->h_pinned[8 + 3];
    n_pairs = c->h_pinned[8 + 4];
    n_two = c->h_pinned[8 + 5];
  } else if (c->h_pinned[3]) {
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    return fail(c, 12, "internal: a capacity guard fired although every batch was sized (flags %llu)",
                (unsigned long long)c->h_pinned[3]);
  }
  c->stats.n_read_windows = n_windows;
  c->stats.n_accepted = c->h_pinned[1];
  c->stats.n_hits = c->nhits = c->h_pinned[2];
  c->stats.n_descriptors = n_pairs;
  c->stats.n_pairs = n_pairs + n_two;  // a two-window descriptor is two of the reference's candidate pairs
  c->stats.n_candidates = n_cand;
  // 0 = proven: no (window,key) block exceeded MaxMatches, the tuples equal the reference's;
  // otherwise an upper bound on the number of such blocks (or ~0ull when the check was skipped)
  c->stats.n_overflow_blocks = check_blocks ? c->h_pinned[5] : ~0ull;
  if (block_mode == 1 && (c->h_pinned[6] || c->stats.n_batches > planned_batches)) {
    // screening inconclusive (a hot sketch cell, or more launches than the threshold assumed):
    // repeat the pass with exact per-block counters
    c->force_exact_blocks = true;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    rc = musc_match_device(c, P, nhits);
    c->force_exact_blocks = false;
    return rc;
  }
  c->stats.ms_screen = tm.total(0);
  c->stats.ms_scan = tm.total(1);
  c->stats.ms_unused0 = tm.total(2);
  c->stats.ms_confirm = tm.total(3);
  c->stats.ms_select = tm.total(4);
  (void)hipEventElapsedTime(&c->stats.ms_total, ev0, ev1);
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  // 12 B descriptor + ceil(2L/8) B read + ceil(2L/8)+1 B target span per
  // candidate pair, + 16 B per tuple written
  const uint64_t L = c->max_len;
  c->stats.confirm_bytes = c->stats.n_pairs * (12 + (2 * L + 7) / 8 + (2 * L + 7) / 8 + 1) + 16 * c->stats.n_hits;
  if (nhits) *nhits = c->nhits;
  if (!sized && want_pipe) {  // the second batch set gets the capacities the first one ended up with
    if ((rc = ensure(c, c->bs[1].wb, c->bs[0].wb.cap)) || (rc = ensure(c, c->bs[1].rvalid, c->bs[0].rvalid.cap)) ||
        (rc = ensure(c, c->bs[1].tbase, c->bs[0].tbase.cap)) || (rc = ensure(c, c->bs[1].tcount, c->bs[0].tcount.cap)) ||
        (rc = ensure(c, c->bs[1].cdesc, c->bs[0].cdesc.cap)))
      return rc;
  }
  c->sized_epoch = c->data_epoch;
  c->sized_params = *P;
  c->sized_exact_blocks = block_mode == 2;
  c->sized_bsz = bsz;
  return 0;
}

int musc_hits_copy(musc_ctx* c, musc_hit* dst, uint64_t capacity, int dst_on_device) {
  if (!c) return 1;
  if (capacity < c->nhits) return fail(c, 2, "musc_hits_copy: capacity %llu < %llu hits",
                                       (unsigned long long)capacity, (unsigned long long)c->nhits);
  if (c->nhits == 0) return 0;
  if (!dst) return fail(c, 2, "musc_hits_copy: dst is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(dst, c->hits.p, c->nhits * sizeof(musc_hit),
                           dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static int check_pack_bits(musc_ctx* c, const int32_t* bits, PackBits* b) {
  if (!bits) return fail(c, 2, "bits is NULL");
  int sum = 0;
  for (int i = 0; i < 4; i++) {
    if (bits[i] < 1 || bits[i] > 32) return fail(c, 2, "field width %d outside 1..32", bits[i]);
    sum += bits[i];
  }
  if (sum > 64) return fail(c, 2, "field widths add up to %d > 64 bits", sum);
  *b = PackBits{bits[0], bits[1], bits[2], bits[3]};
  return 0;
}

int musc_hits_copy_packed(musc_ctx* c, uint64_t* dst, uint64_t capacity, int dst_on_device, uint64_t read_base,
                          const int32_t* bits) {
  if (!c) return 1;
  PackBits b;
  int rc = check_pack_bits(c, bits, &b);
  if (rc) return rc;
  if (capacity < c->nhits) return fail(c, 2, "musc_hits_copy_packed: capacity %llu < %llu hits",
                                       (unsigned long long)capacity, (unsigned long long)c->nhits);
  if (c->nhits == 0) return 0;
  if (!dst) return fail(c, 2, "musc_hits_copy_packed: dst is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_flag) HIPCHK(c, hipMalloc((void**)&c->d_flag, 4));
  uint64_t* out = dst;
  if (!dst_on_device) {
    if ((rc = ensure(c, c->packed, c->nhits))) return rc;
    out = c->packed.p;
  }
  HIPCHK(c, hipMemsetAsync(c->d_flag, 0, 4, c->stream));
  hipLaunchKernelGGL(k_pack_hits, dim3(std::min(nblk(c->nhits, 256), MAX_GRID)), dim3(256), 0, c->stream,
                     reinterpret_cast<const uint4*>(c->hits.p), c->nhits, read_base, b, out, c->d_flag);
  HIPCHK(c, hipGetLastError());
  uint32_t bad = 0;
  HIPCHK(c, hipMemcpyAsync(&bad, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
  if (!dst_on_device)
    HIPCHK(c, hipMemcpyAsync(dst, out, c->nhits * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (bad) return fail(c, 8, "musc_hits_copy_packed: a tuple field does not fit its width (%d/%d/%d/%d bits)",
                       b.read, b.gene, b.pos, b.nmiss);
  return 0;
}

int musc_hits_unpack(musc_ctx* c, const uint64_t* src, uint64_t n, int on_device, const int32_t* bits, musc_hit* dst) {
  if (!c) return 1;
  PackBits b;
  int rc = check_pack_bits(c, bits, &b);
  if (rc) return rc;
  if (n == 0) return 0;
  if (!src || !dst) return fail(c, 2, "musc_hits_unpack: NULL pointer");
  if (!on_device) {  // host to host: plain loop, same layout
    for (uint64_t i = 0; i < n; i++) {
      uint64_t v = src[i];
      dst[i].nmiss = (uint32_t)(v & ((1ull << b.nmiss) - 1ull));
      v >>= b.nmiss;
      dst[i].pos = (uint32_t)(v & ((1ull << b.pos) - 1ull));
      v >>= b.pos;
      dst[i].gene_idx = (uint32_t)(v & ((1ull << b.gene) - 1ull));
      v >>= b.gene;
      dst[i].read_idx = (uint32_t)v;
    }
    return 0;
  }
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_unpack_hits, dim3(std::min(nblk(n, 256), MAX_GRID)), dim3(256), 0, c->stream, src, n, b,
                     reinterpret_cast<uint4*>(dst));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int musc_match(musc_ctx* c, const musc_params* P, musc_hit** hits, uint64_t* nhits) {
  if (!c) return 1;
  if (!hits || !nhits) return fail(c, 2, "musc_match: NULL output pointer");
  *hits = nullptr;
  *nhits = 0;
  uint64_t n = 0;
  int rc = musc_match_device(c, P, &n);
  if (rc) return rc;
  musc_hit* h = (musc_hit*)malloc(sizeof(musc_hit) * (n ? n : 1));
  if (!h) return fail(c, 7, "out of host memory for %llu hits", (unsigned long long)n);
  rc = musc_hits_copy(c, h, n, 0);
  if (rc) {
    free(h);
    return rc;
  }
  *hits = h;
  *nhits = n;
  return 0;
}

void musc_free_hits(musc_hit* hits) { free(hits); }

int musc_overflow_probes(musc_ctx* c, uint32_t** read_idx, uint32_t** window, uint64_t* n) {
  if (!c) return 1;
  if (!read_idx || !window || !n) return fail(c, 2, "musc_overflow_probes: NULL output pointer");
  *read_idx = *window = nullptr;
  *n = 0;
  if (c->stats.n_overflow_blocks == 0 || c->stats.n_overflow_blocks == ~0ull) return 0;
  if (!c->last_exact_blocks || !c->block_table.p) return fail(c, 4, "no exact block counters from the last pass");
  HIPCHK(c, hipSetDevice(c->device));
  uint64_t cap = 1u << 20;
  for (;;) {
    uint2* d_out = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_out, cap * sizeof(uint2)));
    HIPCHK(c, hipMemsetAsync(c->counters + 8, 0, 8, c->stream));
    const dim3 grid(std::min(nblk(c->nreads, 256), MAX_GRID)), block(256);
    switch (c->rw) {
      case 4: hipLaunchKernelGGL((k_hot_probes<4>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->last_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      case 8: hipLaunchKernelGGL((k_hot_probes<8>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->last_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      case 12: hipLaunchKernelGGL((k_hot_probes<12>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->last_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      case 16: hipLaunchKernelGGL((k_hot_probes<16>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->last_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
      default: hipLaunchKernelGGL((k_hot_probes<0>), grid, block, 0, c->stream, c->rd, c->rdm, c->nreads, c->rw, c->last_pp, c->block_table.p, c->last_max_matches, d_out, cap, c->counters + 8); break;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_pinned, c->counters + 8, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      (void)hipFree(d_out);
      return fail(c, 10, "musc_overflow_probes: %s", hipGetErrorString(e));
    }
    const uint64_t found = c->h_pinned[0];
    if (found > cap) {  // retry with room for all of them
      (void)hipFree(d_out);
      cap = found + 16;
      continue;
    }
    std::vector<uint2> h(found ? found : 1);
    if (found) e = hipMemcpy(h.data(), d_out, found * sizeof(uint2), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, 10, "musc_overflow_probes: %s", hipGetErrorString(e));
    uint32_t* r = (uint32_t*)malloc(sizeof(uint32_t) * (found ? found : 1));
    uint32_t* w = (uint32_t*)malloc(sizeof(uint32_t) * (found ? found : 1));
    if (!r || !w) {
      free(r);
      free(w);
      return fail(c, 7, "musc_overflow_probes: out of host memory");
    }
    for (uint64_t j = 0; j < found; j++) {
      r[j] = h[j].x;
      w[j] = h[j].y;
    }
    *read_idx = r;
    *window = w;
    *n = found;
    return 0;
  }
}

void musc_free_u32(uint32_t* p) { free(p); }

int musc_get_stats(musc_ctx* c, musc_stats* out) {
  if (!c || !out) return 1;
  *out = c->stats;
  return 0;
}

int musc_gather(musc_ctx* const* ctxs, int n, const uint64_t* read_base, musc_hit** hits, uint64_t* nhits) {
  if (!ctxs || n < 1 || !hits || !nhits) return 1;
  uint64_t total = 0;
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return 1;
    total += ctxs[i]->nhits;
  }
  musc_hit* h = (musc_hit*)malloc(sizeof(musc_hit) * (total ? total : 1));
  if (!h) return fail(ctxs[0], 7, "musc_gather: out of host memory");
  uint64_t o = 0;
  for (int i = 0; i < n; i++) {
    musc_ctx* c = ctxs[i];
    int rc = musc_hits_copy(c, h + o, c->nhits, 0);
    if (rc) {
      free(h);
      return rc;
    }
    const uint64_t base = read_base ? read_base[i] : 0;
    for (uint64_t j = 0; j < c->nhits; j++) h[o + j].read_idx += (uint32_t)base;
    o += c->nhits;
  }
  *hits = h;
  *nhits = total;
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- read prep (sort + collapse)
#include "muscato_prep.hpp"
