// kernels_match.hpp -- context buckets (index kind 1) and k_match: screen + confirm + per-read
// selection in ONE kernel, no target gather, no descriptor round trip.
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

// The read's IMAGE for window k: its bases moved to where the context stream holds the target
// bases it is compared with -- the read shifted left by sh = 2 * (CL - q1) bits, eight words
// (240 bits).  Phase A of k_match builds it once per (read, window); every comparison of the read
// through that window is then a plain word-by-word XOR against a bucket's context, with no
// per-entry alignment.  sh is wave-uniform.
template <int RW>
DEV void read_image(const Rec<RW>& rec, uint32_t sh, uint32_t (&img)[8]) {
  const uint32_t bs = sh & 31u;
  // word j of (bases << sh) = bases[j - wo] << bs | bases[j - wo - 1] >> (32 - bs), wo = sh / 32: the
  // word offset is resolved by a scalar branch, each case names its registers statically (the
  // record's last word is the length, not bases)
#define MUSC_IMG_W(Q) (((Q) >= 0 && (Q) < RW - 1) ? rec.w[((Q) >= 0 && (Q) < RW - 1) ? (Q) : 0] : 0u)
#define MUSC_IMG_CASE(WO)                                                  \
  case WO:                                                                 \
    _Pragma("unroll") for (int j = 0; j < 8; j++) {                        \
      const uint32_t hi = MUSC_IMG_W(j - WO), lo = MUSC_IMG_W(j - WO - 1); \
      img[j] = bs ? ((hi << bs) | (lo >> (32u - bs))) : hi;                \
    }                                                                      \
    break;
  switch (__builtin_amdgcn_readfirstlane((int)(sh >> 5))) {
    MUSC_IMG_CASE(0) MUSC_IMG_CASE(1) MUSC_IMG_CASE(2) MUSC_IMG_CASE(3)
    MUSC_IMG_CASE(4) MUSC_IMG_CASE(5) MUSC_IMG_CASE(6) MUSC_IMG_CASE(7)
    default:
#pragma unroll
      for (int j = 0; j < 8; j++) img[j] = 0u;
      break;
  }
#undef MUSC_IMG_CASE
#undef MUSC_IMG_W
}

// cdiff of a whole read against one context (cmd/muscato_confirm/main.go:151-159, 205-211): the
// read's image words x the context words c (c[7] cut to its low half), and from the same mismatch
// mask which windows of the read match the target exactly here.  Returns the pair's result word
// (NX_REJECT, or nmiss | NX_DUP | NX_ACC0 | window << 20 | read slot << 24).
//   k      = the probed window (wave-uniform), sh = 2 * (CL - win[k])
//   exact0 = windows that take part for this read (minus the ones the pos-0 rule excludes)
//   ULEN   : every read of the wave-tile has the same length `len` (wave-uniform) -- the masks are
//            rows of the host's tables then, read by the scalar unit; otherwise the length mask is
//            per lane arithmetic
//   lm     : ULEN: the row mp->lm[len][k], loaded by the caller once per window (eight scalars)
//   XM     : xm[] marks (in the image's coordinates, one bit per base like the length mask) the
//            bases of the read that are X: with an X-free database they mismatch wherever they land
template <bool ULEN, bool XM = false>
DEV uint32_t ctx_score(const uint32_t (&img)[8], const uint32_t (&c)[8], uint32_t sh, uint32_t k,
                       const MatchParams* __restrict__ mp, int W, uint32_t exact0, uint32_t budget, uint32_t slot,
                       uint32_t len, const uint32_t (&lm)[8], const uint32_t (&xm)[8]) {
  uint32_t d[8];
  uint32_t nx = 0;
  // (readfirstlane: tells the compiler the index is wave-uniform, so the rows are read by scalar loads)
  const uint32_t ku = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t x = img[j] ^ c[j];
    uint32_t m;
    if constexpr (ULEN) m = lm[j];
    else m = 0x55555555u & bit_range_mask((int)sh - 32 * j, (int)sh + 2 * (int)len - 32 * j);
    d[j] = XM ? ((x | (x >> 1)) | xm[j]) & m : (x | (x >> 1)) & m;
    nx += __popc(d[j]);
  }
  uint32_t exact = exact0;
  const uint32_t need = mp->need[ku];
  for (int kk = 0; kk < W; kk++) {
    if (!((need >> kk) & 1u)) continue;  // wave-uniform
    uint32_t acc = 0;
    const uint32_t* __restrict__ wmrow = mp->wm[ku][kk];
#pragma unroll
    for (int j = 0; j < 8; j++) acc |= d[j] & wmrow[j];
    if (acc) exact &= ~(1u << kk);
  }
  // the reference's confirm for window k accepts the pair (it counts towards that window-key
  // block's MaxMatches); the tuple is reported here only if k is the first window that accepts it
  if (!(nx <= budget && ((exact >> k) & 1u))) return NX_REJECT;
  const bool first = (uint32_t)(__ffs(exact) - 1) == k;
  return (first ? nx : (nx | NX_DUP)) | NX_ACC0 | (k << 20) | (slot << 24);
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

#define WT 64            // reads per wave-tile
#define MATCH_WLIST 128  // reported candidates of a wave-tile kept in LDS (cfg3: ~65); more spill to HBM
#define MATCH_WOWN 64    // overflow items handled per chunk of phase C (a lane per item)

// k_match -- muscato_screen + the join + muscato_confirm + the per-read best/MMTol filter.
// A WAVE works alone on a wave-tile of 64 consecutive reads (persistent waves, no workgroup
// barrier inside the loop: the waves of a CU drift apart, so one wave's memory round trips hide
// behind the others' arithmetic).
//   phase A  a lane per read: record -> registers; per window the length gate +
//            CountDinuc >= MinDinuc (cmd/muscato_window_reads/main.go:106-118 ==
//            cmd/muscato_screen/main.go:174-185), the bucket of the window key, and the read's
//            image for that window (read_image) -> LDS
//   phase B  per window four rounds of 16 probes: a QUAD of lanes fetches one 128-byte bucket (32
//            bytes per lane, two dwordx4, non-temporal: every line is requested once); lane 0 holds
//            count/ovf/gene/jx, lanes 1..3 hold one entry's context each and run cdiff on it where
//            it arrived: context words XOR the read's image from LDS (ctx_score).  All probes of a
//            round belong to one window, so every mask is a scalar.  The W x 4 rounds run through
//            a ring of four register buffers: a round's registers are refilled with the same round
//            of the next window as soon as it has been used.
//   phase C  the entries beyond a bucket's third (CtxEntry in E) of all the wave-tile's probes as
//            one flat list, a lane per entry
//   phase D  best[read] = min nmiss over its reported pairs (LDS atomicMin, filled during B/C),
//            tuples with nmiss <= best + MMTol per read (cmd/muscato_combine_windows/main.go:36-60;
//            all of them when apply_mmtol == 0), a wave scan over the 64 reads, and the tuples
//            themselves, read-major, into the wave's region of `stage`; k_compact_w closes the
//            gaps between wave-tiles afterwards.
// Reported candidates wait in an LDS list (code, gene, pos); a wave-tile with more than
// MATCH_WLIST of them spills the rest to the wave's slice of `spill`.
// MaxMatches accounting as in k_confirm: block_mode 1 = count-min sketch per workgroup in LDS,
// 2 = exact global counters.
// counters (batch-local block = counters + 8): [0] valid windows, [1] entries compared (passed the
// fit rules), [3] index entries walked, [4] overflow entries walked, [5] largest spill any wave
// needed, [6] tuples staged, [7] largest number of tuples any wave staged; pass-level [1]
// reported pairs, [3] flags (1: a stage region ran out, 4: a spill region ran out), [6] a sketch
// cell reached block_thr.
#ifndef MATCH_RING
#define MATCH_RING 4  // rounds (of 16 bucket lines) a wave keeps in flight: 1, 2 or 4
#endif
#ifndef MATCH_WAVES
#define MATCH_WAVES 4  // waves per SIMD the register allocator leaves room for (128 VGPRs)
#endif
template <int RW, bool W2>
__global__ __launch_bounds__(TILE, W2 ? MATCH_WAVES : 2) void k_match(const uint32_t* __restrict__ rd, uint64_t r0, uint32_t n,
                                                   const MatchParams* __restrict__ mp,
                                                   const uint16_t* __restrict__ nmiss_tab,
                                                   const CtxBucket* __restrict__ T, const CtxEntry* __restrict__ E,
                                                   uint4* __restrict__ stage, uint64_t stage_cap,
                                                   uint4* __restrict__ spill, uint64_t spill_cap,
                                                   uint32_t* __restrict__ tbase, uint32_t* __restrict__ tcount2,
                                                   int block_mode, uint32_t block_thr,
                                                   uint32_t* __restrict__ block_table,
                                                   unsigned long long* __restrict__ counters) {
  constexpr int WMAX = W2 ? 2 : CTX_MAX_W;
  constexpr int NWAVE = TILE / 64;
  extern __shared__ uint32_t s_dyn[];  // block_mode != 0: NWAVE x W x 64 per-(window, read) counters, then (mode 1) the sketch
  __shared__ __attribute__((aligned(16))) uint32_t s_img[NWAVE][WMAX * WT * 8];  // the read's image per (window, read)
  __shared__ uint32_t s_meta[NWAVE][WT];            // length | budget << 17 | valid windows << 24
  __shared__ uint32_t s_bb[NWAVE][WMAX * WT];       // bucket of (window, read), WB_NONE when the window takes no part
  __shared__ uint32_t s_oc[NWAVE][WMAX * WT];       // phase B/C: overflow entries of the probe; phase D: cnt[64], base[64]
  __shared__ uint32_t s_ovf[NWAVE][WMAX * WT];      // where in E
  __shared__ uint32_t s_best[2][NWAVE][WT];  // two wave-tiles are alive at once (phase A of the next one runs before phase D)
  __shared__ uint3 s_list[NWAVE][MATCH_WLIST];  // reported candidates: result word, gene, position
  __shared__ uint32_t s_oix[NWAVE][MATCH_WOWN];  // phase C: flat item -> index within its bucket's overflow list
  __shared__ uint8_t s_own[NWAVE][MATCH_WOWN];   //          flat item -> probe (window * 64 + read)
  __shared__ uint16_t s_nm[CONF_NM];

  const int W = mp->W, ww = mp->ww, CL = mp->CL;
  const int win0 = mp->win[0], win1 = mp->win[1];
  const uint32_t q1zero = mp->q1zero_mask;
  // timing experiments only, compile-time (-DMUSC_MATCH_DBG=n: 1 skip the comparisons, 2 skip the bucket loads, 4 skip phase C;
  // wrong tuples): the shipped library cannot be talked into them by an environment variable
#ifdef MUSC_MATCH_DBG
  constexpr int dbg = MUSC_MATCH_DBG;
#else
  constexpr int dbg = 0;
#endif
  uint32_t* const s_sketch = s_dyn + NWAVE * WT * W;
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)mp->max_len ? nmiss_tab[t] : (uint16_t)0;
  if (block_mode == 1)
    for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) s_sketch[t] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) tcount2[(n + WT - 1) / WT] = 0;
  __syncthreads();

  const uint32_t nwt = (n + WT - 1) / WT;
  const uint32_t gw = blockIdx.x * NWAVE + (threadIdx.x >> 6), nw = gridDim.x * NWAVE;
  const uint64_t region = stage_cap / nw, region0 = region * gw;
  const uint64_t sregion = spill_cap / nw, sregion0 = sregion * gw;
  uint64_t used = 0;      // tuples this wave has staged so far (wave-uniform)
  uint32_t maxspill = 0;  // largest spill a wave-tile of this wave needed
  uint32_t nvalid = 0, ncand = 0, ncmp = 0, novf = 0, nrep = 0;  // per lane: far below 2^32
  const uint32_t mmtol = (uint32_t)mp->mmtol;
  const bool apply = mp->apply_mmtol != 0;

  // phase A of wave-tile wt (best-buffer `par`); returns the tile's common read length or ~0
  // the records of wave-tile wt, a lane per read
  auto fetch = [&](uint32_t wt, Rec<RW>& rec) {
    const uint32_t lane = opaque(threadIdx.x) & 63;
    const uint32_t i = wt * WT + lane;
    rec.load(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW, RW);
  };
  auto phase_a = [&](uint32_t wt, uint32_t par, const Rec<RW>& rec) -> uint32_t {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint32_t* const img_l = s_img[wid];
    uint32_t* const bb_l = s_bb[wid];
    uint32_t* const wcnt_l = s_dyn + wid * WT * W;
    const uint32_t i = wt * WT + lane;
    const bool active = i < n;
    const int len = (int)rec.len();
    uint32_t valid = 0;
    for (int k = 0; k < W; k++) {
      uint32_t b = WB_NONE;
      const uint32_t q1 = (uint32_t)mp->win[k], q2 = q1 + (uint32_t)ww;
      if (active) {
        bool pt = (uint32_t)len >= q2;
        if (pt && mp->min_dinuc > 0)
          pt = (ww <= 16 ? rec_count_dinuc16(rec, q1, ww) : rec_count_dinuc(rec, rec, false, q1, ww)) >= mp->min_dinuc;
        if (pt) {
          b = rec_bucket(rec, rec, false, q1, ww, mp->bits, mp->direct);
          valid |= 1u << k;
        }
      }
      bb_l[k * WT + lane] = b;
      uint32_t img[8];
      read_image<RW>(rec, 2u * (uint32_t)(CL - (int)q1), img);
      uint4* dst = reinterpret_cast<uint4*>(&img_l[(k * WT + lane) * 8]);
      dst[0] = make_uint4(img[0], img[1], img[2], img[3]);
      dst[1] = make_uint4(img[4], img[5], img[6], img[7]);
    }
    const uint32_t budget = len < CONF_NM ? s_nm[len] : 0u;  // (reads on this path are at most 120 bases)
    s_meta[wid][lane] = (uint32_t)len | ((budget > 127u ? 127u : budget) << 17) | (valid << 24);
    s_best[par][wid][lane] = 0xFFFFFFFFu;
    nvalid += __popc(valid);
    if (block_mode)
      for (uint32_t t = lane; t < WT * (uint32_t)W; t += 64) wcnt_l[t] = 0;
    // every read of the wave-tile of one length: the comparisons use scalar length masks
    const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(len);
    return __ballot(active && (uint32_t)len != len0) == 0 ? len0 : 0xFFFFFFFFu;
  };
  // the bucket loads of one round of probes (16 probes, a quad each)
  auto issue = [&](int k, int rr, uint4& a, uint4& b2) {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6, part = lane & 3;
    const uint32_t b = s_bb[wid][k * WT + rr * 16 + (lane >> 2)];
    // a probe that takes no part reads as an empty bucket: count 0, and nothing else of the
    // registers (they keep the previous round's words) is looked at
    a.x = 0;
    if (b != WB_NONE && !(dbg & 2)) {
      const u32x4_v* p = reinterpret_cast<const u32x4_v*>(T + b) + 2 * part;
      const u32x4_v x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1);
      a = make_uint4(x.x, x.y, x.z, x.w);
      b2 = make_uint4(y.x, y.y, y.z, y.w);
    }
  };

  // Software pipeline over the wave's wave-tiles: phase A of the NEXT tile and the loads of its
  // first rounds are issued before phase D of the current one, so that a wave always has bucket
  // lines in flight (phases A..C share the per-tile LDS state; D needs only the candidate list and
  // the tile's own best[] buffer).
  uint4 va[MATCH_RING], vb[MATCH_RING];
#pragma unroll
  for (int rr = 0; rr < MATCH_RING; rr++) va[rr] = vb[rr] = make_uint4(0, 0, 0, 0);
  uint32_t ulen = 0xFFFFFFFFu, par = 0;
  if (gw < nwt) {
    Rec<RW> rec;
    fetch(gw, rec);
    ulen = phase_a(gw, 0, rec);
    wave_lds_sync();
#pragma unroll
    for (int rr = 0; rr < MATCH_RING; rr++) issue(0, rr, va[rr], vb[rr]);
  }
  for (uint32_t wt = gw; wt < nwt; wt += nw) {
    // (the lane's LDS addresses are derived afresh in every iteration: kept across the loop they
    // would occupy dozens of registers and spill)
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6, part = lane & 3;
    uint32_t* const img_l = s_img[wid];
    uint32_t* const bb_l = s_bb[wid];
    uint32_t* const oc_l = s_oc[wid];
    uint32_t* const ovf_l = s_ovf[wid];
    uint32_t* const cnt_l = s_oc[wid];
    uint32_t* const base_l = s_oc[wid] + WT;
    uint32_t* const wcnt_l = s_dyn + wid * WT * W;
    uint32_t* const best_l = s_best[par][wid];
    uint32_t nlist = 0;  // reported candidates of this wave-tile so far (wave-uniform)

    // one reported candidate per set lane of the vote, appended in lane order
    auto report = [&](uint32_t w, uint32_t gene, uint32_t pos) {
      const bool acc = w != NX_REJECT;
      if (acc && block_mode) atomicAdd(&wcnt_l[((w >> 20) & 15u) * WT + (w >> 24)], 1u);
      const bool rep = acc && !(w & NX_DUP);
      const unsigned long long vote = __ballot(rep);
      if (vote == 0) return;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      const uint32_t slot = nlist + below;
      nlist += (uint32_t)__popcll(vote);
      if (!rep) return;
      atomicMin(&best_l[w >> 24], w & 0xFFFFu);
      if (slot < MATCH_WLIST) {
        s_list[wid][slot] = make_uint3(w, gene, pos);
      } else if (slot - MATCH_WLIST < sregion) {
        spill[sregion0 + (slot - MATCH_WLIST)] = make_uint4(w, gene, pos, 0u);
      }
    };

    // one entry (context words c, c[7] still carrying rem16 in its high half) of probe (k, ri)
    auto compare = [&](auto ulen_tag, uint32_t k, int q1, uint32_t sh, const uint32_t (&lm)[8], uint32_t ri, uint32_t jx,
                       bool live, uint32_t (&c)[8]) -> uint32_t {
      constexpr bool ULEN = decltype(ulen_tag)::value;
      const uint32_t meta = s_meta[wid][ri];
      const int rlen = (int)REC_LEN(meta);
      // placements past the target's first bases (p = jx - q1 > 0) only have to end inside the
      // target; the pos-0 rules are evaluated only when some lane of the wave is at p <= 0
      uint32_t z = 0;
      bool ok;
      if (__any(live && jx <= (uint32_t)q1)) ok = live & ctx_fit(jx, c[7] >> 16, q1, ww, rlen, &z);
      else ok = live & (rlen - q1 <= (int)(c[7] >> 16));
      uint32_t w = NX_REJECT;
      {
        const unsigned long long okv = __ballot(ok);  // counted by the scalar unit, credited to one lane
        ncmp += lane == 0 ? (uint32_t)__popcll(okv) : 0u;
      }
      if (ok) {
        uint32_t img[8];
        const uint4* src = reinterpret_cast<const uint4*>(&img_l[(k * WT + ri) * 8]);
        const uint4 i0 = src[0], i1 = src[1];
        img[0] = i0.x; img[1] = i0.y; img[2] = i0.z; img[3] = i0.w; img[4] = i1.x; img[5] = i1.y; img[6] = i1.z; img[7] = i1.w;
        c[7] &= 0xFFFFu;
        const uint32_t exact0 = REC_VALID(meta) & (z ? ~q1zero : 0xFFFFFFFFu);
        w = ctx_score<ULEN>(img, c, sh, k, mp, W, exact0, REC_BUDGET(meta), ri, ULEN ? ulen : (uint32_t)rlen, lm, lm);
      }
      return w;
    };

    // ---- phase B
    auto process = [&](auto ulen_tag, int k, int q1, uint32_t sh, const uint32_t (&lm)[8], int rr, const uint4& a,
                       const uint4& b2) {
      // (opaque: the probes of round rr belong to the same reads for every window; the compiler
      // would otherwise keep what it derives from them in registers across the window loop)
      const uint32_t ri = opaque((uint32_t)rr * 16 + (lane >> 2));
      // block 0 sits in the quad's first lane (quad_perm [0,0,0,0]); every lane executes the moves
#define MUSC_Q0(X) (uint32_t)__builtin_amdgcn_mov_dpp((int)(X), 0x00, 0xF, 0xF, true)
      const uint32_t cnt = MUSC_Q0(a.x);
      const uint32_t g0 = MUSC_Q0(a.z), g1 = MUSC_Q0(a.w), g2 = MUSC_Q0(b2.x);
      const uint32_t j0 = MUSC_Q0(b2.y), j1 = MUSC_Q0(b2.z), j2 = MUSC_Q0(b2.w);
#undef MUSC_Q0
      if (part == 0) {
        ncand += cnt;
        oc_l[k * WT + ri] = cnt > CTX_INLINE ? cnt - CTX_INLINE : 0u;
        ovf_l[k * WT + ri] = a.y;
      }
      const uint32_t gene = part == 1 ? g0 : (part == 2 ? g1 : g2);
      const uint32_t jx = part == 1 ? j0 : (part == 2 ? j1 : j2);
      uint32_t c[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w};
      const bool live = part >= 1 && part - 1 < cnt && !(dbg & 1);
      const uint32_t w = compare(ulen_tag, (uint32_t)k, q1, sh, lm, ri, jx, live, c);
      report(w, gene, jx - (uint32_t)q1);
    };
    {
      // a ring of four register buffers over the W x 4 rounds: a round's registers are refilled
      // with the same round of the next window as soon as it has been used
      auto rounds = [&](auto ulen_tag) {
#pragma unroll 1
        for (int k = 0; k < W; k++) {
          // the window's geometry, new to the compiler in every iteration (opaque_s): the scalar
          // masks derived from it live for these four rounds only
          const int q1 = W2 ? (k == 0 ? win0 : win1) : mp->win[k];
          const uint32_t sh = opaque_s(2u * (uint32_t)(CL - q1));
          // the window's length-mask row: eight scalars for these four rounds
          uint32_t lm[8];
          {
            constexpr bool ULEN = decltype(ulen_tag)::value;
            const uint32_t* __restrict__ row = mp->lm[ULEN ? __builtin_amdgcn_readfirstlane((int)ulen) : 0][k];
#pragma unroll
            for (int j = 0; j < 8; j++) lm[j] = ULEN ? row[j] : 0u;
          }
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            process(ulen_tag, k, q1, sh, lm, rr, va[rr % MATCH_RING], vb[rr % MATCH_RING]);
            // refill the slot with the round MATCH_RING ahead (this window's, or the next one's)
            if (rr + MATCH_RING < 4) issue(k, rr + MATCH_RING, va[rr % MATCH_RING], vb[rr % MATCH_RING]);
            else if (k + 1 < W) issue(k + 1, rr + MATCH_RING - 4, va[rr % MATCH_RING], vb[rr % MATCH_RING]);
          }
        }
      };
      if (ulen != 0xFFFFFFFFu) rounds(std::true_type{}); else rounds(std::false_type{});
    }
    wave_lds_sync();

    const bool have_next = wt + nw < nwt;

    // ---- phase C: the overflow entries of the wave-tile's W x 64 probes, MATCH_WOWN per chunk
    if (!(dbg & 4)) {
      uint32_t oc[WMAX], pre[WMAX];
      uint32_t total = 0;
#pragma unroll
      for (int k = 0; k < WMAX; k++) {
        oc[k] = pre[k] = 0;
        if (k >= W) continue;
        oc[k] = oc_l[k * WT + lane];
        const uint32_t inc = wave_scan_incl(oc[k]);
        pre[k] = total + inc - oc[k];
        total += __builtin_amdgcn_readlane(inc, 63);
        novf += oc[k];
      }
      for (uint32_t c0 = 0; c0 < total; c0 += MATCH_WOWN) {
#pragma unroll
        for (int k = 0; k < WMAX; k++) {
          if (k >= W) continue;
          // this lane's items of window k that fall into [c0, c0 + MATCH_WOWN)
          const uint32_t e_lo = c0 > pre[k] ? c0 - pre[k] : 0u;
          const uint32_t e_hi = pre[k] + oc[k] > c0 + MATCH_WOWN ? (c0 + MATCH_WOWN > pre[k] ? c0 + MATCH_WOWN - pre[k] : 0u) : oc[k];
          for (uint32_t e = e_lo; e < e_hi; e++) {
            s_own[wid][pre[k] + e - c0] = (uint8_t)(k * WT + lane);
            s_oix[wid][pre[k] + e - c0] = e;
          }
        }
        wave_lds_sync();
        {
          const uint32_t t = c0 + lane;
          const bool have = t < total && !(dbg & 1);
          uint32_t k = 0, seg = 0, gene = 0, jx = 0, w = NX_REJECT;
          uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          if (have) {
            const uint32_t probe = s_own[wid][t - c0];
            const uint32_t e = s_oix[wid][t - c0];
            k = probe >> 6;
            seg = probe & 63u;
            const uint32_t* __restrict__ pe =
                reinterpret_cast<const uint32_t*>(E) + ctx_entry_word<false>(ovf_l[probe] + e);
            const uint2 hd = *reinterpret_cast<const uint2*>(pe);
            const u32x4_u x = *reinterpret_cast<const u32x4_u*>(pe + 2);
            const u32x4_u y = *reinterpret_cast<const u32x4_u*>(pe + 6);
            gene = hd.x;
            jx = hd.y;
            c[0] = x.x; c[1] = x.y; c[2] = x.z; c[3] = x.w; c[4] = y.x; c[5] = y.y; c[6] = y.z; c[7] = y.w;
          }
          // the window is per lane here: one masked comparison per window present
          int q1 = 0;
          for (int kk = 0; kk < W; kk++) {
            const bool mine = have && k == (uint32_t)kk;
            if (!__any(mine)) continue;
            const int q1k = W2 ? (kk == 0 ? win0 : win1) : mp->win[kk];
            const uint32_t shk = opaque_s(2u * (uint32_t)(CL - q1k));
            uint32_t lmk[8];
            {
              const uint32_t* __restrict__ row = mp->lm[ulen != 0xFFFFFFFFu ? __builtin_amdgcn_readfirstlane((int)ulen) : 0][kk];
#pragma unroll
              for (int j = 0; j < 8; j++) lmk[j] = row[j];
            }
            uint32_t w2;
            if (ulen != 0xFFFFFFFFu) w2 = compare(std::true_type{}, (uint32_t)kk, q1k, shk, lmk, seg, jx, mine, c);
            else w2 = compare(std::false_type{}, (uint32_t)kk, q1k, shk, lmk, seg, jx, mine, c);
            if (mine) {
              w = w2;
              q1 = q1k;
            }
          }
          report(w, gene, jx - (uint32_t)q1);
        }
        wave_lds_sync();  // the owner tables are rewritten by the next chunk
      }
    }

    // ---- phase D: per-read selection and the tuples
    {
      const uint32_t nl = nlist;
      const uint32_t nspill = nl > MATCH_WLIST ? nl - MATCH_WLIST : 0u;
      const bool spill_ok = nspill <= sregion;
      if (nspill > maxspill) maxspill = nspill;
      if (nspill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's spilled candidates have landed
      if (block_mode) {
        for (uint32_t t = lane; t < WT * (uint32_t)W; t += 64) {
          const uint32_t cw = wcnt_l[t];
          if (!cw) continue;
          const uint32_t kk = t >> 6;  // counters and buckets share the layout [window][read]
          const uint32_t h = block_hash32((uint32_t)kk, bb_l[t]);
          if (block_mode == 1) atomicAdd(&s_sketch[h >> (32 - MATCH_SKETCH_BITS)], cw);
          else atomicAdd(&block_table[h >> (32 - BLOCK_TABLE_BITS)], cw);
        }
      }
      wave_lds_sync();  // phases A..C of this tile are done with the per-tile LDS state
      // the next wave-tile: phase A and the loads of its first rounds, in flight during phase D
      uint32_t ulen_next = 0xFFFFFFFFu;
      if (have_next) {
        Rec<RW> nrec;
        fetch(wt + nw, nrec);
        ulen_next = phase_a(wt + nw, par ^ 1u, nrec);
        wave_lds_sync();
#pragma unroll
        for (int rr = 0; rr < MATCH_RING; rr++) issue(0, rr, va[rr], vb[rr]);
      }
      cnt_l[lane] = 0;  // s_oc becomes cnt / base
      wave_lds_sync();
      auto item = [&](uint32_t j, uint32_t* gene, uint32_t* pos) -> uint32_t {
        if (j < MATCH_WLIST) {
          const uint3 it = s_list[wid][j];
          *gene = it.y;
          *pos = it.z;
          return it.x;
        }
        // written by other lanes of this wave a moment ago: read past the L1
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(spill + sregion0 + (j - MATCH_WLIST));
        *gene = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *pos = __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      const uint32_t nuse = spill_ok ? nl : (nl < MATCH_WLIST ? nl : MATCH_WLIST);
      for (uint32_t j = lane; j < nuse; j += 64) {
        uint32_t g, p;
        const uint32_t w = item(j, &g, &p);
        const uint32_t rl = w >> 24;
        const uint32_t thr = apply ? best_l[rl] + mmtol : 0xFFFFu;
        if ((w & 0xFFFFu) <= thr) atomicAdd(&cnt_l[rl], 1u);
      }
      wave_lds_sync();
      const uint32_t cnum = cnt_l[lane];
      const uint32_t inc = wave_scan_incl(cnum);
      const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
      base_l[lane] = inc - cnum;
      cnt_l[lane] = 0;  // now the arrival counter of the read
      const uint64_t base = region0 + used;
      const bool fits = spill_ok && used + total <= region;
      if (lane == 0) {
        tbase[wt] = (uint32_t)base;
        tcount2[wt] = fits ? total : 0u;
      }
      wave_lds_sync();
      if (fits && total) {
        for (uint32_t j = lane; j < nuse; j += 64) {
          uint32_t g, p;
          const uint32_t w = item(j, &g, &p);
          const uint32_t rl = w >> 24, v = w & 0xFFFFu;
          const uint32_t thr = apply ? best_l[rl] + mmtol : 0xFFFFu;
          if (v > thr) continue;
          const uint32_t ord = atomicAdd(&cnt_l[rl], 1u);
          stage[base + base_l[rl] + ord] = make_uint4((uint32_t)(r0 + wt * WT + rl), g, p, v);
        }
      }
      used += total;
      nrep += lane == 0 ? nl : 0u;  // (one lane carries the wave-uniform count into the final reduction)
      ulen = ulen_next;
      par ^= 1u;
      wave_lds_sync();  // the next wave-tile's phase B rewrites s_oc / the candidate list
    }
  }
  // One reduction per workgroup and a handful of atomics from its first thread: atomics on one
  // address serialise at about 90 M/s on this GPU, so per-wave atomics from a large grid would
  // cost more than the kernel (the grid is also kept to the waves that are resident at once).
  {
    __shared__ unsigned long long s_red[NWAVE][8];
    unsigned long long v[5] = {nvalid, ncmp, ncand, novf, nrep};
#pragma unroll
    for (int q = 0; q < 5; q++)
      for (int d = 32; d; d >>= 1) v[q] += __shfl_xor(v[q], d);
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int q = 0; q < 5; q++) s_red[wv][q] = v[q];
      s_red[wv][5] = used <= region ? used : 0;
      s_red[wv][6] = used;
      s_red[wv][7] = ((unsigned long long)(used > region) << 32) | maxspill;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t[6] = {0, 0, 0, 0, 0, 0}, mx_used = 0, mx_spill = 0, over = 0;
      for (int w = 0; w < NWAVE; w++) {
        for (int q = 0; q < 6; q++) t[q] += s_red[w][q];
        mx_used = s_red[w][6] > mx_used ? s_red[w][6] : mx_used;
        const unsigned long long sp = s_red[w][7] & 0xFFFFFFFFull;
        mx_spill = sp > mx_spill ? sp : mx_spill;
        over |= s_red[w][7] >> 32;
      }
      if (t[0]) atomicAdd(&counters[8 + 0], t[0]);
      if (t[1]) atomicAdd(&counters[8 + 1], t[1]);
      if (t[2]) atomicAdd(&counters[8 + 3], t[2]);
      if (t[3]) atomicAdd(&counters[8 + 4], t[3]);
      if (t[4]) atomicAdd(&counters[1], t[4]);
      if (t[5]) atomicAdd(&counters[8 + 6], t[5]);
      atomicMax(&counters[8 + 7], mx_used);
      if (mx_spill) atomicMax(&counters[8 + 5], mx_spill);
      if (over) atomicOr(&counters[3], 1ull);
      if (mx_spill > sregion) atomicOr(&counters[3], 4ull);
    }
  }
  if (block_mode == 1) {
    __syncthreads();
    uint32_t hot = 0;
    for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) hot |= s_sketch[t] >= block_thr;
    if (__any(hot) && (threadIdx.x & 63) == 0) atomicOr(&counters[6], 1ull);
  }
}

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
