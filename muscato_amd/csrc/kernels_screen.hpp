// kernels_screen.hpp -- k_screen: read windows probe the index, candidates filtered from the index entry alone
// Part of libmuscato_hip.so: included by muscato_hip.hip (one translation unit).
#pragma once

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
  int32_t reserved0;     // (was a run-time experiment knob; experiments are compile-time now: -DMUSC_SCREEN_DBG=n)
};

// timing experiments only, compile-time (-DMUSC_SCREEN_DBG=n: 1 skip entry tests, 2 skip bucket loads, 4 skip descriptor
// writes, 8 no two-window descriptors, 256 skip the bucket phase; wrong tuples)
#ifdef MUSC_SCREEN_DBG
#define SCREEN_DBG (MUSC_SCREEN_DBG)
#else
#define SCREEN_DBG 0
#endif

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
  // the same as whole 16-byte pieces the compiler may not split (a plain load it narrows to overlapping dwordx2
  // pieces, one per (w[i], w[i+1]) pair that ext() shifts: 2 RW - 2 registers for RW words); records stream
  // through once: non-temporal
  DEV void load_nt(const uint32_t* __restrict__ p) {
#pragma unroll
    for (int q = 0; q < RW / 4; q++) {
      const u32x4_v a = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(p) + q);
      w[4 * q] = a.x; w[4 * q + 1] = a.y; w[4 * q + 2] = a.z; w[4 * q + 3] = a.w;
    }
  }
  DEV uint32_t len() const { return w[RW - 1] & 0xFFFFu; }
  DEV bool has_x() const { return (w[RW - 1] & READ_HAS_X) != 0; }
  DEV void zero() {
#pragma unroll
    for (int q = 0; q < RW; q++) w[q] = 0;
  }
  // 64 bits from bit offset bo.  bo is wave-uniform in every caller, so the word index is
  // resolved by a scalar branch and each case names its registers statically (a chain of
  // per-word selects costs 3*RW VALU per call instead).
  DEV uint64_t ext(uint32_t bo) const {
    const uint32_t sh = bo & 31u;
    uint32_t lo = 0, hi = 0;
    switch (__builtin_amdgcn_readfirstlane((int)(bo >> 5))) {
      // (the shifts inside each case, on single registers: with the three words selected first the compiler keeps
      // every (w[i], w[i+1]) as a 64-bit register pair -- it then loads the record as overlapping dwordx2 pieces,
      // 2 RW - 2 registers for RW words, and where registers are short spills the pair of the last word)
#define MUSC_EXT_W(J) ((J) < RW ? w[(J) < RW ? (J) : 0] : 0u)
#define MUSC_EXT_CASE(I)                                            \
  case I:                                                           \
    if (I < RW) {                                                   \
      lo = __funnelshift_r(MUSC_EXT_W(I), MUSC_EXT_W(I + 1), sh);   \
      hi = __funnelshift_r(MUSC_EXT_W(I + 1), MUSC_EXT_W(I + 2), sh); \
    }                                                               \
    break;
      MUSC_EXT_CASE(0) MUSC_EXT_CASE(1) MUSC_EXT_CASE(2) MUSC_EXT_CASE(3)
      MUSC_EXT_CASE(4) MUSC_EXT_CASE(5) MUSC_EXT_CASE(6) MUSC_EXT_CASE(7)
      MUSC_EXT_CASE(8) MUSC_EXT_CASE(9) MUSC_EXT_CASE(10) MUSC_EXT_CASE(11)
      MUSC_EXT_CASE(12) MUSC_EXT_CASE(13) MUSC_EXT_CASE(14) MUSC_EXT_CASE(15)
#undef MUSC_EXT_CASE
      default: break;
    }
    return (uint64_t)lo | ((uint64_t)hi << 32);
  }
  // 32 bits from bit offset bo (wave-uniform): two words and one v_alignbit -- for the callers that want a window
  // key of up to 16 bases or a flank (no 64-bit pairs to keep in registers)
  DEV uint32_t ext32(uint32_t bo) const {
    const uint32_t sh = bo & 31u;
    uint32_t r = 0;
    switch (__builtin_amdgcn_readfirstlane((int)(bo >> 5))) {
#define MUSC_EXT_CASE(I)                                                                      \
  case I:                                                                                     \
    if (I < RW) r = __funnelshift_r(MUSC_EXT_W(I), MUSC_EXT_W(I + 1), sh);                    \
    break;
      MUSC_EXT_CASE(0) MUSC_EXT_CASE(1) MUSC_EXT_CASE(2) MUSC_EXT_CASE(3)
      MUSC_EXT_CASE(4) MUSC_EXT_CASE(5) MUSC_EXT_CASE(6) MUSC_EXT_CASE(7)
      MUSC_EXT_CASE(8) MUSC_EXT_CASE(9) MUSC_EXT_CASE(10) MUSC_EXT_CASE(11)
      MUSC_EXT_CASE(12) MUSC_EXT_CASE(13) MUSC_EXT_CASE(14) MUSC_EXT_CASE(15)
#undef MUSC_EXT_CASE
#undef MUSC_EXT_W
      default: break;
    }
    return r;
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
  DEV bool has_x() const { return (p[rw - 1] & READ_HAS_X) != 0; }
  DEV uint64_t ext(uint32_t bo) const { return ext64(p, bo); }
  DEV uint32_t ext32(uint32_t bo) const { return (uint32_t)ext64(p, bo); }
};

// 8 bases ending just before base `base` of a record, like flank_left on a stream
template <class R>
DEV uint32_t rec_flank_left(const R& r, uint32_t base) {
  if (base >= 8) return r.ext32(2 * (base - 8)) & 0xFFFFu;
  return (r.ext32(0) << (2 * (8 - base))) & 0xFFFFu;
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
#define DESC_RX (1u << 23)   // descriptor z: the read holds an X (its mask words are needed)
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

// fit rules + flank filter for one index entry against one probe; true = worth a target gather.  Written without
// short-circuit evaluation on purpose: every term is a vector compare into a scalar mask and the masks are combined
// by the scalar unit -- with && / if-else the compiler wraps each entry's test in nested exec-mask regions.
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
  // p = jx - q1 >= 0; a window at target position 0 takes the pos-0 path, any other needs p + len <= T
  const bool fits = (left == 0) ? fit0 : (rlen - q1 <= right);
  const uint32_t x = rfl ^ ent.w;
  const uint32_t d = (x | (x >> 1)) & 0x55555555u & fmask;
  const bool ok = (q1 <= left) & fits & ((uint32_t)__popc(d) <= (lenbud >> 16));
  *zflag = ((left == q1) & !fit0) ? 1u : 0u;  // p == 0 but the pos-0 path rejects
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
#ifndef SCR_CU
#define SCR_CU 4  // phase C: overflow entries a thread has in flight
#endif
#ifndef SCR_LROUNDS
#define SCR_LROUNDS 8  // line buckets: rounds of eight lines whose loads are in flight together (16 per chunk)
#endif
#ifndef SCR_WAVES
#define SCR_WAVES 4  // waves per SIMD the register allocator has to leave room for
#endif
// ONE: at most two windows, i.e. a single chunk -- the compiler then keeps nothing alive across
// chunks (59 instead of 80 VGPRs without mask planes) and eight waves fit a SIMD.
// LINES: the index is a table of LineBucket (kernels_index.hpp): a probe's line -- header + seven
// entries -- is fetched by EIGHT lanes (16 bytes each, a wave instruction brings eight whole lines),
// and the overflow entries of a bucket are a 128-byte-aligned run.
template <int RW, bool MASK, bool ONE, bool LINES>
__global__ __launch_bounds__(TILE, (ONE && !MASK) ? 8 : SCR_WAVES) void k_screen(const uint32_t* __restrict__ rd,
                                                 const uint32_t* __restrict__ rdm, uint64_t r0,
                                                 uint32_t n, int rw_rt, const PathParams* __restrict__ ppp,
                                                 const uint16_t* __restrict__ nmiss_tab,
                                                 const Bucket* __restrict__ T,
                                                 const uint4* __restrict__ E,
                                                 uint4* __restrict__ desc, uint64_t desc_cap,
                                                 uint32_t* __restrict__ rvalid,
                                                 uint32_t* __restrict__ wb,
                                                 uint32_t* __restrict__ tbase,
                                                 uint32_t* __restrict__ tcount,
                                                 unsigned long long* __restrict__ counters,
                                                 unsigned long long* __restrict__ pass_flags) {
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
  // the run's parameters stay in device memory and are read (scalar loads) where they are used: a
  // by-value block of 28 words lived in scalar registers for the whole kernel and spilled 64 of them
  const PathParams& pp = *ppp;
  __shared__ uint32_t s_wsum[TILE / 64];
  __shared__ uint32_t s_bb[SCR_PROBES];        // per probe: bucket, WB_NONE when the window takes no part
  __shared__ uint32_t s_rfl[SCR_PROBES];       // per probe: the read's own 8+8 flanking bases
  __shared__ uint32_t s_lenbud[SCR_PROBES];    // read length | mismatch budget << 16
  __shared__ uint32_t s_oc[SCR_PROBES];        // entries of the probe's bucket that live in E
  __shared__ uint64_t s_ovf[SCR_PROBES];       // where in E
  __shared__ uint32_t s_pref[SCR_PROBES + 1];  // exclusive prefix of s_oc
  __shared__ uint16_t s_own[SCR_OWN];          // flat item -> probe
  __shared__ uint32_t s_tilecnt;               // survivors of the tile so far
  __shared__ uint8_t s_rx[TILE];               // the read holds an X (mask planes only)
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
    if constexpr (has_m) {
      // the mask words only of reads that hold an X (flag in the length word)
      if (RW == 0 || rec.has_x()) recm_store.load(rdm + r * (uint64_t)rw, rw);
      else if constexpr (RW != 0) recm_store.zero();
    }
    if constexpr (has_m) s_rx[tida] = rec.has_x() ? 1 : 0;
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
      if (ok && slot < room && !(SCREEN_DBG & 4)) {
        const uint32_t left = ent.z & 0xFFFFu;
        const uint32_t pos_ok = left < 65535u ? 1u : 0u;
        // global offset of the placement (40 bits in wide mode: the high byte rides in x)
        const uint64_t gp = (((uint64_t)(pp.wide ? ent.x >> 24 : 0u) << 32) | ent.y) - (uint64_t)q1;
        desc[base + slot] = make_uint4((tile * TILE + (probe >> 1)) | ((uint32_t)(gp >> 32) << 24), (uint32_t)gp,
                                       (uint32_t)k | (z << 4) | (pos_ok << 5) | ((left - (uint32_t)q1) << 6) | (two ? DESC_TWO : 0u) |
                                           ((has_m && s_rx[probe >> 1]) ? DESC_RX : 0u),
                                       pp.wide ? (ent.x & 0xFFFFFFu) : ent.x);
      }
    };

    for (int k0 = 0; k0 < (ONE ? 1 : pp.W); k0 += 2) {
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
      if constexpr (LINES) {
        // ---- phase B (line buckets): eight lanes per probe; the lane that holds one of the line's seven
        // entries tests it on the spot.  (No two-window descriptors here: each window's survivors are
        // compared on their own; k_confirm's first-window rule keeps the tuple set the same.)
        const uint4* __restrict__ TL = reinterpret_cast<const uint4*>(T);
#pragma unroll 1
        for (int h = 0; h < ((SCREEN_DBG & 256) ? 0 : 16 / SCR_LROUNDS); h++) {
          const uint32_t tidb = opaque(threadIdx.x);
          const uint32_t lane = tidb & 63, wid = tidb >> 6, part = lane & 7;
          uint4 v[SCR_LROUNDS];
#pragma unroll
          for (int rr = 0; rr < SCR_LROUNDS; rr++) {
            const uint32_t b = s_bb[wid * 128 + (SCR_LROUNDS * h + rr) * 8 + (lane >> 3)];
            v[rr] = make_uint4(0, 0, 0, 0);
            if (b != WB_NONE && !(SCREEN_DBG & 2)) {
              const u32x4_v t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(TL + (uint64_t)b * 8u) + part);
              v[rr] = make_uint4(t.x, t.y, t.z, t.w);
            }
          }
#pragma unroll
          for (int rr = 0; rr < SCR_LROUNDS; rr++) {
            const uint32_t probe = wid * 128 + (SCR_LROUNDS * h + rr) * 8 + (lane >> 3);
            // the header sits in the first of the eight lanes: quad_perm [0,0,0,0], then the upper quad
            // takes the lower quad's copy (row_shr:4)
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v[rr].x, 0x00, 0xF, 0xF, true);
            const uint32_t c1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c0, 0x114, 0xF, 0xF, false);
            const uint32_t cnt = part < 4 ? c0 : c1;
            if (part == 0) {
              ncand += cnt;
              s_oc[probe] = cnt > LINE_INLINE ? cnt - LINE_INLINE : 0u;
              s_ovf[probe] = (uint64_t)v[rr].y * 8u;  // the bucket's overflow run, in entries
            }
            const int k = k0 + (int)(probe & 1u), q1 = (probe & 1u) ? q1b : q1a;
            uint32_t z = 0;
            bool ok = part >= 1 && part - 1 < cnt && !(SCREEN_DBG & 1);
            if (ok) ok = screen_entry_ok(v[rr], q1, pp.ww, s_rfl[probe], s_lenbud[probe], &z);
            append(ok, v[rr], probe, k, q1, z, false);
          }
        }
      } else {
      // ---- phase B: buckets by quads; a wave fetches the 128 probes of its own 64 reads
#pragma unroll 1
      for (int h = 0; h < ((SCREEN_DBG & 256) ? 0 : 8 / SCR_ROUNDS); h++) {
        const uint32_t tidb = opaque(threadIdx.x);
        const uint32_t lane = tidb & 63, wid = tidb >> 6;
        uint4 v[SCR_ROUNDS];
#pragma unroll
        for (int rr = 0; rr < SCR_ROUNDS; rr++) {
          const uint32_t b = s_bb[wid * 128 + (SCR_ROUNDS * h + rr) * 16 + (lane >> 2)];
          v[rr] = make_uint4(0, 0, 0, 0);
          if (b != WB_NONE && !(SCREEN_DBG & 2)) {
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
          bool ok = part >= 1 && part - 1 < cnt && !(SCREEN_DBG & 1);
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
          if (!ok || (SCREEN_DBG & 8)) same = 0;
          const bool two = same && !odd;
          if (same && odd) ok = false;
          ntwo += two;
          append(ok, v[rr], probe, k, q1, z, two);
        }
      }
      }
      lds_barrier();
      // ---- phase C: the chunk's overflow entries as one flat list, in (read, window, entry) order
      const uint32_t tidc = opaque(threadIdx.x);
      if (SCREEN_DBG & 256) continue;
      const uint32_t oc0 = s_oc[2 * tidc], oc1 = s_oc[2 * tidc + 1];
      uint32_t total = 0;
      const uint32_t pre = wg_scan(oc0 + oc1, &total);
      if (total != 0 && !(SCREEN_DBG & 1)) {  // uniform
        s_pref[2 * tidc] = pre;
        s_pref[2 * tidc + 1] = pre + oc0;
        if (tidc == TILE - 1) s_pref[SCR_PROBES] = total;
        for (uint32_t e = 0; e < oc0 && pre + e < SCR_OWN; e++) s_own[pre + e] = (uint16_t)(2 * tidc);
        for (uint32_t e = 0; e < oc1 && pre + oc0 + e < SCR_OWN; e++) s_own[pre + oc0 + e] = (uint16_t)(2 * tidc + 1);
        lds_barrier();
        // SCR_CU items per thread and round: their loads are all issued before the first is tested --
        // with one load in flight per lane a wave keeps 1 KB on its way and the whole GPU 4 MB, which
        // at HBM latency is 2 TB/s however fast the tests are (measured on the cfg5 shard: 10 entries
        // per probe, 2.2 TB/s of entries)
        for (uint32_t t0 = 0; t0 < total; t0 += SCR_CU * TILE) {
          uint32_t segs[SCR_CU];
          uint4 ents[SCR_CU];
          bool oks[SCR_CU];
#pragma unroll
          for (int u = 0; u < SCR_CU; u++) {
            const uint32_t t = t0 + (uint32_t)u * TILE + tidc;
            oks[u] = t < total;
            segs[u] = 0;
            ents[u] = make_uint4(0, 0, 0, 0);
            if (oks[u]) {
              uint32_t seg;
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
              segs[u] = seg;
              const u32x4_v te = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(E) + s_ovf[seg] + (t - s_pref[seg]));
              ents[u] = make_uint4(te.x, te.y, te.z, te.w);
            }
          }
#pragma unroll
          for (int u = 0; u < SCR_CU; u++) {
            if (u > 0 && t0 + (uint32_t)u * TILE >= total) break;  // uniform
            const uint32_t seg = segs[u];
            uint32_t z = 0;
            const int k = k0 + (int)(seg & 1u), q1 = (seg & 1u) ? q1b : q1a;
            bool ok = oks[u];
            if (ok) ok = screen_entry_ok(ents[u], q1, pp.ww, s_rfl[seg], s_lenbud[seg], &z);
            append(ok, ents[u], seg, k, q1, z, false);
          }
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
    if (used > region) atomicOr(pass_flags, 1ull);  // pass-level flag: descriptor space ran out
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
