// kernels_index.hpp -- the k-mer index of the database: bucket layout and build kernels
// Part of libmuscato_hip.so: included by muscato_hip.hip (one translation unit).
#pragma once

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
//
// LINE BUCKETS (LineBucket) -- the layout for dense databases (four or more window starts per key:
// BASELINE config 5, 10 Gbp over 4^15 keys = 9.3 entries per key).  With 64-byte buckets such a probe
// costs the bucket's line plus an unaligned run of overflow entries, about three random 128-byte
// lines; here a bucket IS a line -- a 16-byte header (count, first overflow line) and seven
// entries -- and the entries beyond the seventh sit in 128-byte-aligned runs of eight: a ten-entry
// probe costs 1.5 lines, a 9.3-entries-per-key database 1.75 on average.
// ------------------------------------------------------------------------------------
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
#define BUCKET_INLINE 3
struct __attribute__((aligned(64))) Bucket {
  static constexpr uint32_t NIN = BUCKET_INLINE;
  uint32_t count;
  uint32_t cursor;
  uint64_t ovf;
  uint4 e[BUCKET_INLINE];
  __device__ uint64_t ovf_entry(uint32_t slot) const { return ovf + (slot - NIN); }          // where entry `slot` sits in E
  __device__ static uint64_t ovf_units(uint32_t count) { return count > NIN ? count - NIN : 0u; }  // what the bucket needs of E, in its unit (entries)
  __device__ void set_ovf(uint64_t v) { ovf = v; }
};
static_assert(sizeof(Bucket) == 64, "bucket must be half a cache line");

#define LINE_INLINE 7
struct __attribute__((aligned(128))) LineBucket {
  static constexpr uint32_t NIN = LINE_INLINE;
  uint32_t count;
  uint32_t ovf;     // first overflow line of this bucket (E in units of 8 entries)
  uint32_t cursor;  // build only
  uint32_t pad;
  uint4 e[LINE_INLINE];
  __device__ uint64_t ovf_entry(uint32_t slot) const { return (uint64_t)ovf * 8u + (slot - NIN); }
  __device__ static uint64_t ovf_units(uint32_t count) { return count > NIN ? (count - NIN + 7u) / 8u : 0u; }  // lines
  __device__ void set_ovf(uint64_t v) { ovf = (uint32_t)v; }
};
static_assert(sizeof(LineBucket) == 128, "a line bucket is one cache line");

DEV uint32_t flank_left(const uint32_t* __restrict__ w, uint64_t base) {
  // 8 bases ending just before base index `base` of plane w (zeros before the stream start)
  if (base >= 8) return (uint32_t)ext64(w, 2 * (base - 8)) & 0xFFFFu;
  return (uint32_t)(ext64(w, 0) << (2 * (8 - base))) & 0xFFFFu;
}

template <bool SCATTER, class BT>
__global__ __launch_bounds__(256) void k_index(const uint32_t* __restrict__ db2,
                                               const uint32_t* __restrict__ dbm2,
                                               const uint64_t* __restrict__ seq_off, uint32_t nseq,
                                               uint64_t nbases, int ww, int bits, int direct, int wide,
                                               BT* __restrict__ T, uint4* __restrict__ E) {
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
      if (slot < BT::NIN) T[b].e[slot] = ent;
      else E[T[b].ovf_entry(slot)] = ent;
    }
  }
}

// overflow list sizes: tmp[b] = what the bucket needs of E (entries beyond the inline ones, or lines of
// eight), scanned on the side, written back as ovf
template <class BT>
__global__ void k_index_ovf_count(const BT* __restrict__ T, uint64_t nb, uint64_t* __restrict__ tmp) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) tmp[b] = BT::ovf_units(T[b].count);
  else if (b == nb) tmp[b] = 0;
}

template <class BT>
__global__ void k_index_ovf_set(BT* __restrict__ T, uint64_t nb, const uint64_t* __restrict__ tmp) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) T[b].set_ovf(tmp[b]);
}
