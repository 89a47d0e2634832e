// kernels_match.hpp -- context buckets (index kind 1): layout, index build, parameter block, fit rules and helpers of
// the fused kernels (screen + confirm + per-read selection in ONE kernel, no target gather, no descriptor round trip:
// k_match_t in kernels_match_lane.hpp, k_match_g in kernels_match_dma.hpp).
// Part of libmuscato_hip.so: included by muscato_hip.hip (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------
// Context buckets.  Every L2 miss on gfx950 moves a 128-byte line, so the classic 64-byte bucket
// (kernels_index.hpp) already costs a whole line per probe -- and each surviving entry then costs
// k_confirm a second random line for the target span plus a 16-byte descriptor written and read
// back.  Here the line a probe pays for carries what cdiff needs: a bucket is ONE line,
//   block 0  (32 B): count, ovf (first overflow entry of this bucket in E), gene[3], jx[3]
//   block s+1 (32 B), s = 0..2: the CONTEXT of inline entry s -- the CTX_BASES = 120 target bases
//            [jx - CL, jx - CL + 120) as a 2-bit stream (240 bits: words 0..6 and the low half of
//            word 7) and, in the high half of word 7, min(T - jx, 65535) (distance to the target
//            end; jx itself, 32 bits, sits in block 0).
// CL = the largest window start of the run, so a read placed through window k (start q1) covers
// context bases [CL - q1, CL - q1 + len): the whole read -- the window included, which also makes
// a hashed table exact (a colliding key fails the window comparison) -- is compared from the line
// the probe fetched (cmd/muscato_confirm/main.go:151-159, 205-211).  Usable when
// CL - min(q1) + max read length <= 120 (Windows 0,20 + 100-bp reads: exactly 120), at most
// CTX_MAX_W windows, database below 2^32 bases; everything else takes the classic path.  Entries
// beyond a bucket's third live in E as 40-byte CtxEntry.
// A DATABASE WITH X (N in the FASTA): the context is a 2-bit stream and cannot hold an X, so
//   * a window that holds an X is not indexed (a read window without X never equals it; runs whose
//     read windows hold an X take the classic path, k_xpos_check_db), and
//   * an entry whose context holds an X of its target carries CTX_XFLAG in its jx word (targets are
//     shorter than 2^31 bases) and lists up to two of them by their context position: the first in the
//     top byte of its gene word (at most 2^24 targets), the second (CTX_XNONE if there is none) in the
//     top byte of its last context word, whose distance to the target end then saturates at 255 (reads
//     on this path are at most 200 bases: every comparison against it stays exact) -- the usual case:
//     k_match_t<.., XM = 2> sets the bits in registers.  Three and more: CTX_XMANY in the gene word,
//     and the kernel reads the mask plane of the target span (seq_off[gene] + jx - CL,
//     MatchParams.seq_off / dbm2).  Compared as cdiff does: X == X is a match, X against a base a
//     mismatch (cmd/muscato_confirm/main.go:151-159).
// The table has 4^ww buckets (key = bucket, exact) when that is at most twice the database's
// window count, else about one bucket per base under a 64-bit mix: 1 Gbp -> 2^30 x 128 B =
// 128 GiB, 100 Mbp -> 2^27 x 128 B = 16 GiB.
// ------------------------------------------------------------------------------------
#define CTX_BASES 120
#define CTX_INLINE 3
#define CTX_MAX_W 4
#define CTX_XFLAG 0x80000000u  // in an entry's jx word: the context holds an X of its target ...
#define CTX_XMANY 0xFFu        // ... top byte of the gene word: where (context base 0..199), or CTX_XMANY = three and more
#define CTX_XNONE 0xFFu        // ... top byte of the last context word: where the second one is, or CTX_XNONE

struct __attribute__((aligned(128))) CtxBucket {
  uint32_t count;
  uint32_t ovf;
  uint32_t gene[CTX_INLINE];
  uint32_t jx[CTX_INLINE];
  uint32_t ctx[CTX_INLINE][8];
};
static_assert(sizeof(CtxBucket) == 128, "a context bucket is one cache line");

struct CtxEntry {  // overflow entry, 40 bytes; three per 128-byte line of E (ctx_entry_word)
  uint32_t gene, jx;
  uint32_t ctx[8];
};
static_assert(sizeof(CtxEntry) == 40, "overflow entries are packed");

// WIDE context buckets: the same line with TWO inline entries of 60 bytes -- target number, window
// position and CTXW_BASES = 200 target bases (400 bits: words 0..11 and the low half of word 12;
// the high half of word 12: distance to the target end) -- for runs that do not fit 120 bases of
// context: Windows 0,20,40 with 100-bp reads need 140, two windows with 150-bp reads 170.  Entries
// beyond a bucket's second live in E as 60-byte CtxEntryW.  Matched by k_match_t only.
#define CTXW_BASES 200
#define CTXW_INLINE 2
#define CTXW_WORDS 13
struct CtxEntryW {
  uint32_t gene, jx;
  uint32_t ctx[CTXW_WORDS];
};
static_assert(sizeof(CtxEntryW) == 60, "wide entries are packed");
struct __attribute__((aligned(128))) CtxBucketW {
  uint32_t count;
  uint32_t ovf;
  CtxEntryW e[CTXW_INLINE];
};
static_assert(sizeof(CtxBucketW) == 128, "a wide context bucket is one cache line");

// Where overflow entry i sits in E, in 32-bit words: an entry never straddles a 128-byte line -- three
// 40-byte entries (wide: two of 60 bytes) per line, the last 8 bytes of a line unused.  A 40-byte
// entry at 40 * i would cross a line boundary three times in ten and cost two requests.
template <bool WIDE>
__host__ __device__ inline uint64_t ctx_entry_word(uint32_t i) {
  if (WIDE) return (uint64_t)(i >> 1) * 32u + (i & 1u) * 15u;
  const uint32_t q = i / 3u;
  return (uint64_t)q * 32u + (i - 3u * q) * 10u;
}
// bytes of E for n overflow entries
inline uint64_t ctx_entries_bytes(uint64_t n, bool wide) { return (wide ? (n + 1) / 2 : (n + 2) / 3) * 128ull; }

// bits [lo, hi) of a 32-bit word as a mask, for any int lo / hi (host and device)
__host__ __device__ inline uint32_t bit_range_mask(int lo, int hi) {
  if (hi <= 0 || lo >= 32 || hi <= lo) return 0u;
  const uint32_t mh = hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u);
  const uint32_t ml = lo <= 0 ? 0xFFFFFFFFu : ~((1u << lo) - 1u);
  return mh & ml;
}

// 64 bits of the stream from a possibly negative bit offset (zeros before the stream start)
DEV uint64_t ext64s(const uint32_t* __restrict__ w, long long bo) {
  if (bo >= 0) return ext64(w, (uint64_t)bo);
  if (bo <= -64) return 0ull;
  return ext64(w, 0) << (uint32_t)(-bo);
}

// the NW context words of the window starting at global base g of a target that ends at base `e`: the
// bases [g - CL, ...) as a 2-bit stream, its last half word replaced by the distance to the target end
template <int NW>
DEV void ctx_words(const uint32_t* __restrict__ db2, uint64_t g, uint64_t e, int CL, uint32_t (&c)[NW]) {
  const long long bo = 2 * ((long long)g - (long long)CL);
#pragma unroll
  for (int i = 0; i < (NW + 1) / 2; i++) {
    const uint64_t v = ext64s(db2, bo + 64 * i);
    c[2 * i] = (uint32_t)v;
    if (2 * i + 1 < NW) c[2 * i + 1] = (uint32_t)(v >> 32);
  }
  const uint64_t rem = e - g;
  c[NW - 1] = (c[NW - 1] & 0xFFFFu) | ((uint32_t)(rem > 65535 ? 65535 : rem) << 16);
}

// bits [2 g, 2 (g + n)) of the mask plane hold an X
DEV bool plane_any_x(const uint32_t* __restrict__ dbm2, uint64_t g, uint32_t n) {
  uint64_t bo = 2 * g;
  int left = 2 * (int)n;
  while (left > 0) {
    const int take = left < 64 ? left : 64;
    if (ext64(dbm2, bo) & lowmask64(take)) return true;
    bo += 64;
    left -= 64;
  }
  return false;
}

// dbm2 / dbx: the database's mask plane and its X-block bitmap, or null for a database without X
// the last context word of a flagged entry: context bits | min(distance to the target end, 255) << 16 | the second X's place << 24
DEV uint32_t ctx_xtail(uint32_t w, uint32_t xtail) {
  const uint32_t rem = w >> 16;
  return (w & 0xFFFFu) | ((rem > 255u ? 255u : rem) << 16) | ((xtail & 0xFFu) << 24);
}
// what a kernel reads back of it
DEV uint32_t ctx_rem(uint32_t w, bool flagged) { return flagged ? (w >> 16) & 0xFFu : w >> 16; }

template <bool SCATTER, bool WIDE>
__global__ __launch_bounds__(256) void k_index_ctx(const uint32_t* __restrict__ db2, const uint32_t* __restrict__ dbm2,
                                                   const uint32_t* __restrict__ dbx,
                                                   const uint64_t* __restrict__ seq_off, uint32_t nseq,
                                                   uint64_t nbases, int ww, int bits, int direct, int CL,
                                                   CtxBucket* __restrict__ T, void* __restrict__ Ev,
                                                   uint32_t* __restrict__ cursor) {
  __shared__ uint32_t s_g0;
  const uint64_t nchunks = (nbases + blockDim.x - 1) / blockDim.x;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t gfirst = chunk * blockDim.x;
    __syncthreads();
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
    uint32_t xflag = 0, xgene = 0, xtail = 0;  // xtail: 0x100 | the second X's place, for a flagged entry
    if (dbx) {
      // a window with an X has no key here; an entry whose context holds an X of its own target is
      // flagged, with the position of the X (or CTX_XMANY) in the top byte of the gene word
      if (db_span_has_x(dbx, g, (uint32_t)ww) && plane_any_x(dbm2, g, (uint32_t)ww)) continue;
      constexpr int NB = WIDE ? CTXW_BASES : CTX_BASES;
      const long long lo = (long long)g - (long long)CL, hi = lo + NB;
      const uint64_t c0 = lo > 0 ? (uint64_t)lo : 0ull, c1 = (uint64_t)hi < nbases ? (uint64_t)hi : nbases;
      if (SCATTER && db_span_has_x(dbx, c0, (uint32_t)(c1 - c0))) {
        // context base c is global base lo + c: inside the target for c in [tlo, thi)
        const int tlo = (long long)s > lo ? (int)((long long)s - lo) : 0;
        const int thi = (long long)e < hi ? (int)((long long)e - lo) : NB;
        uint32_t nx = 0, where = 0, where2 = CTX_XNONE;
#pragma unroll
        for (int j = 0; j < (NB + 15) / 16; j++) {
          uint32_t m = (uint32_t)ext64s(dbm2, 2 * lo + 32 * j) & 0x55555555u & bit_range_mask(2 * tlo - 32 * j, 2 * thi - 32 * j);
          nx += (uint32_t)__popc(m);
          if (m && where2 == CTX_XNONE) {
            const uint32_t p = 16u * (uint32_t)j + (((uint32_t)__ffs(m) - 1u) >> 1);
            if (nx == (uint32_t)__popc(m)) {  // the first X of the context is in this word
              where = p;
              m &= m - 1u;
              if (m) where2 = 16u * (uint32_t)j + (((uint32_t)__ffs(m) - 1u) >> 1);
            } else {
              where2 = p;
            }
          }
        }
        if (nx) {
          xflag = CTX_XFLAG;
          xgene = (nx <= 2 ? where : CTX_XMANY) << 24;
          xtail = 0x100u | where2;
        }
      }
    }
    const uint32_t b = bucket_of(db2, nullptr, 2 * g, ww, bits, direct);
    if (!SCATTER) {
      atomicAdd(&T[b].count, 1u);
    } else {
      const uint32_t slot = atomicAdd(&cursor[b], 1u);
      if constexpr (WIDE) {
        CtxBucketW* const TW = reinterpret_cast<CtxBucketW*>(T);
        CtxEntryW ent;
        ent.gene = gene | xgene;
        ent.jx = (uint32_t)jx | xflag;
        ctx_words<CTXW_WORDS>(db2, g, e, CL, ent.ctx);
        if (xtail) ent.ctx[CTXW_WORDS - 1] = ctx_xtail(ent.ctx[CTXW_WORDS - 1], xtail);
        uint32_t* pw = slot < CTXW_INLINE ? reinterpret_cast<uint32_t*>(&TW[b].e[slot])
                                          : reinterpret_cast<uint32_t*>(Ev) + ctx_entry_word<true>(TW[b].ovf + (slot - CTXW_INLINE));
        pw[0] = ent.gene;
        pw[1] = ent.jx;
#pragma unroll
        for (int i = 0; i < CTXW_WORDS; i++) pw[2 + i] = ent.ctx[i];
      } else {
        uint32_t c[8];
        ctx_words<8>(db2, g, e, CL, c);
        if (xtail) c[7] = ctx_xtail(c[7], xtail);
        if (slot < CTX_INLINE) {
          T[b].gene[slot] = gene | xgene;
          T[b].jx[slot] = (uint32_t)jx | xflag;
          uint4* dst = reinterpret_cast<uint4*>(T[b].ctx[slot]);
          dst[0] = make_uint4(c[0], c[1], c[2], c[3]);
          dst[1] = make_uint4(c[4], c[5], c[6], c[7]);
        } else {
          uint32_t* p = reinterpret_cast<uint32_t*>(Ev) + ctx_entry_word<false>(T[b].ovf + (slot - CTX_INLINE));
          p[0] = gene | xgene;
          p[1] = (uint32_t)jx | xflag;
#pragma unroll
          for (int i = 0; i < 8; i++) p[2 + i] = c[i];
        }
      }
    }
  }
}

MUSC_KERNEL void k_ctx_ovf_count(const CtxBucket* __restrict__ T, uint64_t nb, uint32_t n_inline, uint64_t* __restrict__ tmp) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) {
    const uint32_t c = T[b].count;  // (count and ovf sit in the first eight bytes of either bucket layout)
    tmp[b] = c > n_inline ? c - n_inline : 0u;
  } else if (b == nb) {
    tmp[b] = 0;
  }
}

MUSC_KERNEL void k_ctx_ovf_set(CtxBucket* __restrict__ T, uint64_t nb, const uint64_t* __restrict__ tmp) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) T[b].ovf = (uint32_t)tmp[b];
}

// ------------------------------------------------------------------------------------
// k_match
// ------------------------------------------------------------------------------------

// what the kernel reads of the run's parameters, in device memory: uniform loads on demand
// instead of a by-value block that lives in (and spills from) scalar registers
struct MatchParams {
  int32_t W, ww, min_dinuc, bits, direct, mmtol, apply_mmtol, max_len, CL;
  uint32_t q1zero_mask;
  int32_t reserved0;  // (was a run-time experiment knob; experiments are compile-time now: -DMUSC_MATCH_DBG=n)
  int32_t win[CTX_MAX_W];
  // a database with X (k_match_t<.., XM = 2>): where a target starts and the mask plane, read for flagged entries only
  const uint64_t* seq_off;
  const uint32_t* dbm2;
  // Scalar mask tables, filled by the host (match_tables): a comparison through window k works in
  // the coordinates of the context stream, where the read sits at bits [sh_k, sh_k + 2 len), sh_k =
  // 2 * (CL - win[k]).  Word j of
  //   lm[len][k]  = 0x55555555 & (bits of the read): one bit per base that takes part in cdiff
  //   wm[k][kk]   = the bits of window kk of the read
  // need[k] = the windows whose exactness a comparison through window k has to establish: the
  // ones before k (first-window rule) and k itself when the table is hashed (a direct table's
  // bucket is the key: the probed window matches by construction).
  // (rows of CTXW_WORDS words: wide context buckets use them all, the 120-base ones the first eight)
  uint32_t need[CTX_MAX_W];
  uint32_t wm[CTX_MAX_W][CTX_MAX_W][CTXW_WORDS];
  uint32_t lm[CTXW_BASES + 1][CTX_MAX_W][CTXW_WORDS];
};


// host: the mask tables of a parameter block whose scalar fields are set
inline void match_tables(MatchParams& mp) {
  for (int k = 0; k < CTX_MAX_W; k++) {
    mp.need[k] = 0;
    if (k >= mp.W) continue;
    const int sh = 2 * (mp.CL - mp.win[k]);
    mp.need[k] = ((1u << k) - 1u) | (mp.direct ? 0u : (1u << k));
    for (int kk = 0; kk < mp.W; kk++)
      for (int j = 0; j < CTXW_WORDS; j++)
        mp.wm[k][kk][j] = bit_range_mask(sh + 2 * mp.win[kk] - 32 * j, sh + 2 * (mp.win[kk] + mp.ww) - 32 * j);
    for (int len = 0; len <= CTXW_BASES; len++)
      for (int j = 0; j < CTXW_WORDS; j++) mp.lm[len][k][j] = 0x55555555u & bit_range_mask(sh - 32 * j, sh + 2 * len - 32 * j);
  }
}


#ifndef MATCH_SKETCH_BITS
#define MATCH_SKETCH_BITS 10
#endif

// fit rules of one index entry for a read of rlen bases placed through the window at q1:
// p = jx - q1 >= 0; target position 0 takes the pos-0 path of processSeq with its literal 100
// (cmd/muscato_screen/main.go:294-316), everything else must end inside the target
// (cmd/muscato_screen/main.go:347-353 + cmd/muscato_confirm/main.go:201-203).  *zflag: p == 0 but the
// pos-0 path would reject -- windows starting at 0 cannot have emitted this placement.
DEV bool ctx_fit(uint32_t jx, uint32_t rem16, int q1, int ww, int rlen, uint32_t* zflag) {
  const int left = (int)(jx > 65535u ? 65535u : jx), right = (int)rem16;
  const int tcap = left + right;  // target length, saturated (exact below 65535)
  const int lim0 = 100 - ww < tcap ? 100 - ww : tcap;
  const bool fit0 = rlen <= lim0;
  const bool inside = left == 0 ? fit0 : (rlen - q1 <= right);  // no short-circuit: plain selects, no branches
  *zflag = (uint32_t)(left == q1) & (uint32_t)!fit0;
  return (q1 <= left) & inside;
}

// The meta word of a read in LDS (phase A of k_match): length | budget << 17 | windows that take
// part << 24.
#define REC_LEN(w) ((w) & 0xFFFFu)
#define REC_BUDGET(w) (((w) >> 17) & 0x7Fu)
#define REC_VALID(w) ((w) >> 24)

// utils/entropy.go:5-40 for a window of at most 16 bases without X: the window is one 32-bit
// word, a dinucleotide one 4-bit field of it (rec_count_dinuc is the general form)
DEV int key_count_dinuc16(uint32_t key, int ww) {
  uint32_t seen = 0;
#pragma unroll
  for (int i = 0; i < 15; i++)
    if (i + 1 < ww) seen |= 1u << ((key >> (2 * i)) & 15u);
  return __popc(seen);
}
template <class R>
DEV int rec_count_dinuc16(const R& r, uint32_t q1, int ww) {
  return key_count_dinuc16((uint32_t)r.ext(2 * q1), ww);
}

// single-instruction forms the compiler does not pick by itself (it re-associates the chains for
// instruction-level parallelism the kernel has no use for: it is bound by the number of vector
// instructions it issues)
DEV uint32_t bcnt_add(uint32_t x, uint32_t acc) {  // popcount(x) + acc
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
DEV uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) {  // (a & b) | c
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
DEV uint32_t and_or_s(uint32_t a, uint32_t b, uint32_t c) {  // the same with b in a scalar register
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
// one bit per base at which two 2-bit streams differ (x = their XOR), under the mask m:
// ((x | x >> 1) & m), the OR-AND as one v_bitop3
DEV uint32_t base_diff(uint32_t x, uint32_t m) { return __builtin_amdgcn_bitop3_b32(x, x >> 1, m, 0xA8); }

// utils/entropy.go:5-40 for a window of at most 16 bases without X: a dinucleotide is a 4-bit field
// of the key; two instructions each (v_bfe_u32, v_lshl_or_b32)
DEV int key_dinucs16(uint32_t key, int ww) {
  uint32_t seen = 0;
#pragma unroll
  for (int i = 0; i < 15; i++)
    if (i + 1 < ww) seen = (1u << __builtin_amdgcn_ubfe(key, 2 * i, 4)) | seen;
  return __popc(seen);
}

// Everything a wave shares through LDS is its own: a wave's LDS operations execute in order, so
// all that is needed between a write and another lane's read is that the compiler keeps the order.
DEV void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// A wave-uniform value the compiler must treat as new: what is derived from it (the scalar masks
// of a comparison) is recomputed by the scalar unit where it is used -- that unit idles otherwise
// -- instead of being kept in (and spilled from) scalar registers across the whole kernel.
DEV uint32_t opaque_s(uint32_t x) {
  asm volatile("" : "+s"(x));
  return x;
}

// inclusive scan over the 64 lanes of a wave with DPP moves only (no LDS round trips): rows of 16
// by row_shr 1/2/4/8, then row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3
DEV uint32_t wave_scan_incl(uint32_t v) {
#define MUSC_DPP_ADD(CTRL, ROWS) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false)
  MUSC_DPP_ADD(0x111, 0xF);  // row_shr:1
  MUSC_DPP_ADD(0x112, 0xF);  // row_shr:2
  MUSC_DPP_ADD(0x114, 0xF);  // row_shr:4
  MUSC_DPP_ADD(0x118, 0xF);  // row_shr:8
  MUSC_DPP_ADD(0x142, 0xA);  // row_bcast:15
  MUSC_DPP_ADD(0x143, 0xC);  // row_bcast:31
#undef MUSC_DPP_ADD
  return v;
}

#define WT 64            // reads per wave-tile (a lane per read: kernels_match_lane.hpp, kernels_match_dma.hpp)
// (r01-r03 had a first fused kernel here, k_match: a quad of lanes per probe, comparison where the line arrives; retired
// in r04 -- k_match_t and k_match_g are the two fused implementations, and every parity test runs both)

// ------------------------------------------------------------------------------------
// Reads with X on context buckets (k_match_t, RX): with an X-free database an X in a read is a mismatch wherever
// the read is placed, and a window holding one never finds its key in the index.  What the kernel
// needs of a read's X is where they are: xpos = a few positions from bit 0 up and their number (bits
// 28-31, saturating at 15) -- four positions of 7 bits for the 120-base buckets, three of 8 bits for
// the wide ones (reads of up to 200 bases).  A read with more X than its word lists takes part
// only if that many mismatches exceed its budget anyway -- then it has no tuples at all (k_xpos_check
// makes the run take the two-kernel path otherwise).
#define XPOS_CNT(w) ((w) >> 28)
template <bool WIDE>
struct XPos {
  static constexpr uint32_t MAX = WIDE ? 3u : 4u, BITS = WIDE ? 8u : 7u;
  __host__ __device__ static uint32_t at(uint32_t w, uint32_t q) { return (w >> (BITS * q)) & ((1u << BITS) - 1u); }
};

template <bool WIDE>
__global__ void k_read_xpos(const uint32_t* __restrict__ rd, const uint32_t* __restrict__ rdm, uint64_t nreads, int rw,
                            uint32_t* __restrict__ xpos) {
  typedef XPos<WIDE> XP;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  uint32_t w = 0;
  if (rd[(i + 1) * rw - 1] & READ_HAS_X) {
    uint32_t cnt = 0;
    bool far = false;  // an X beyond what a position field holds: the read counts as "too many to list"
    for (int j = 0; j < rw - 1; j++) {
      uint32_t m = rdm[i * rw + j] & 0x55555555u;
      while (m) {
        const uint32_t b = (uint32_t)__ffs(m) - 1u;
        m &= m - 1u;
        const uint32_t p = 16u * (uint32_t)j + (b >> 1);
        if (p >= (1u << XP::BITS)) far = true;
        else if (cnt < XP::MAX) w |= p << (XP::BITS * cnt);
        cnt++;
      }
    }
    w |= ((cnt > 15u || far) ? 15u : cnt) << 28;
  }
  xpos[i] = w;
}

// *bad = 1 if some read holds more X than its word lists and that many mismatches are within its budget
MUSC_KERNEL void k_xpos_check(const uint32_t* __restrict__ rd, const uint32_t* __restrict__ xpos, uint64_t nreads, int rw,
                             const uint16_t* __restrict__ nmiss_tab, uint32_t max_len, uint32_t xmax, uint32_t* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  const uint32_t cnt = XPOS_CNT(xpos[i]);
  if (cnt <= xmax) return;
  const uint32_t len = rd[(i + 1) * rw - 1] & 0xFFFFu;
  const uint32_t budget = len <= max_len ? nmiss_tab[len] : 0xFFFFu;
  if (cnt == 15u || cnt <= budget) atomicOr(bad, 1u);
}

// The same question when the DATABASE holds X as well (k_match_t<.., XM = 2>): there an X of a read can
// match an X of a target, so a read with more X than its word lists cannot be written off, and a read
// window that holds an X could equal a database window that does (not indexed): *bad = 1 if some read
// has an X beyond its word's list or inside one of the run's windows.
struct XWins { int32_t n, ww, q1[CTX_MAX_W]; };
template <bool WIDE>
__global__ void k_xpos_check_db(const uint32_t* __restrict__ xpos, uint64_t nreads, XWins wn, uint32_t* __restrict__ bad) {
  typedef XPos<WIDE> XP;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  const uint32_t w = xpos[i], cnt = XPOS_CNT(w);
  if (!cnt) return;
  bool b = cnt > XP::MAX;
  for (uint32_t q = 0; q < XP::MAX && q < cnt; q++)
    for (int k = 0; k < wn.n; k++) b |= XP::at(w, q) - (uint32_t)wn.q1[k] < (uint32_t)wn.ww;
  if (b) atomicOr(bad, 1u);
}

// k_compact_w -- hits[counters[2] + tpre[wt] ...] = the wave-tile's staged tuples (tpre = scan of
// tcount2), a wave per wave-tile: plain 16-byte copies, contiguous on both sides.
MUSC_KERNEL __launch_bounds__(256) void k_compact_w(uint32_t nwt, const uint32_t* __restrict__ tbase,
                                                   const uint32_t* __restrict__ tcount2,
                                                   const uint32_t* __restrict__ tpre, const uint4* __restrict__ stage,
                                                   uint4* __restrict__ hits, uint64_t hits_cap,
                                                   unsigned long long* __restrict__ counters) {
  const unsigned long long base = counters[2];
  if (base + tpre[nwt] > hits_cap) {  // cannot happen on a sized pass; a sync-free pass re-runs sized
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&counters[3], 2ull);
    return;
  }
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t wt = blockIdx.x * 4 + (threadIdx.x >> 6); wt < nwt; wt += gridDim.x * 4) {
    const uint32_t m = tcount2[wt];
    const uint4* __restrict__ src = stage + tbase[wt];
    uint4* __restrict__ dst = hits + base + tpre[wt];
    for (uint32_t j = lane; j < m; j += 64) dst[j] = src[j];
  }
}
