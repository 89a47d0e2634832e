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
// CTX_MAX_W windows, no X in reads or database, database below 2^32 bases; everything else takes
// the classic path.  Entries beyond a bucket's third live in E as 40-byte CtxEntry.
// The table has 4^ww buckets (key = bucket, exact) when that is at most twice the database's
// window count, else about one bucket per base under a 64-bit mix: 1 Gbp -> 2^30 x 128 B =
// 128 GiB, 100 Mbp -> 2^27 x 128 B = 16 GiB.
// ------------------------------------------------------------------------------------
#define CTX_BASES 120
#define CTX_INLINE 3
#define CTX_MAX_W 4

struct __attribute__((aligned(128))) CtxBucket {
  uint32_t count;
  uint32_t ovf;
  uint32_t gene[CTX_INLINE];
  uint32_t jx[CTX_INLINE];
  uint32_t ctx[CTX_INLINE][8];
};
static_assert(sizeof(CtxBucket) == 128, "a context bucket is one cache line");

struct CtxEntry {  // overflow entry, 40 bytes (8-byte aligned)
  uint32_t gene, jx;
  uint32_t ctx[8];
};
static_assert(sizeof(CtxEntry) == 40, "overflow entries are packed");

// 64 bits of the stream from a possibly negative bit offset (zeros before the stream start)
DEV uint64_t ext64s(const uint32_t* __restrict__ w, long long bo) {
  if (bo >= 0) return ext64(w, (uint64_t)bo);
  if (bo <= -64) return 0ull;
  return ext64(w, 0) << (uint32_t)(-bo);
}

// the eight context words of the window starting at global base g of a target that ends at base `e`
DEV void ctx_words(const uint32_t* __restrict__ db2, uint64_t g, uint64_t e, int CL, uint32_t (&c)[8]) {
  const long long bo = 2 * ((long long)g - (long long)CL);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint64_t v = ext64s(db2, bo + 64 * i);
    c[2 * i] = (uint32_t)v;
    c[2 * i + 1] = (uint32_t)(v >> 32);
  }
  const uint64_t rem = e - g;
  c[7] = (c[7] & 0xFFFFu) | ((uint32_t)(rem > 65535 ? 65535 : rem) << 16);
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void k_index_ctx(const uint32_t* __restrict__ db2,
                                                   const uint64_t* __restrict__ seq_off, uint32_t nseq,
                                                   uint64_t nbases, int ww, int bits, int direct, int CL,
                                                   CtxBucket* __restrict__ T, CtxEntry* __restrict__ E,
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
    const uint32_t b = bucket_of(db2, nullptr, 2 * g, ww, bits, direct);
    if (!SCATTER) {
      atomicAdd(&T[b].count, 1u);
    } else {
      const uint32_t slot = atomicAdd(&cursor[b], 1u);
      uint32_t c[8];
      ctx_words(db2, g, e, CL, c);
      if (slot < CTX_INLINE) {
        T[b].gene[slot] = gene;
        T[b].jx[slot] = (uint32_t)jx;
        uint4* dst = reinterpret_cast<uint4*>(T[b].ctx[slot]);
        dst[0] = make_uint4(c[0], c[1], c[2], c[3]);
        dst[1] = make_uint4(c[4], c[5], c[6], c[7]);
      } else {
        CtxEntry* p = E + ((uint64_t)T[b].ovf + (slot - CTX_INLINE));
        p->gene = gene;
        p->jx = (uint32_t)jx;
#pragma unroll
        for (int i = 0; i < 8; i++) p->ctx[i] = c[i];
      }
    }
  }
}

__global__ void k_ctx_ovf_count(const CtxBucket* __restrict__ T, uint64_t nb, uint64_t* __restrict__ tmp) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) {
    const uint32_t c = T[b].count;
    tmp[b] = c > CTX_INLINE ? c - CTX_INLINE : 0u;
  } else if (b == nb) {
    tmp[b] = 0;
  }
}

__global__ void k_ctx_ovf_set(CtxBucket* __restrict__ T, uint64_t nb, const uint64_t* __restrict__ tmp) {
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
  int32_t win[CTX_MAX_W];
};

#define MATCH_LIST 512   // reported candidates of a tile kept in LDS (a cfg3 tile has ~260); more spill to HBM
#define MATCH_OWN 1024   // overflow items per window whose owner is looked up directly
#define MATCH_SKETCH_BITS 10

// fit rules of one index entry for a read of rlen bases placed through the window at q1:
// p = jx - q1 >= 0; target position 0 takes the pos-0 path of processSeq with its literal 100
// (cmd/muscato_screen/main.go:294-316), everything else must end inside the target
// (cmd/muscato_screen/main.go:347-353 + cmd/muscato_confirm/main.go:201-203).  *zflag: p == 0 but the
// pos-0 path would reject -- windows starting at 0 cannot have emitted this placement.
DEV bool ctx_fit(uint32_t jx, uint32_t rem16, int q1, int ww, int rlen, uint32_t* zflag) {
  const int left = jx > 65535u ? 65535 : (int)jx, right = (int)rem16;
  int lim0 = 100 - ww;
  const int tcap = left + right;  // target length, saturated (exact below 65535)
  if (lim0 > tcap) lim0 = tcap;
  const bool fit0 = rlen <= lim0;
  bool ok = q1 <= left;
  if (left == 0) ok = ok && fit0;
  else ok = ok && (rlen - q1 <= right);
  *zflag = (left == q1 && !fit0) ? 1u : 0u;
  return ok;
}

// cdiff of a whole read against the context stream c (240 bits in c[0..7], the high half of c[7]
// is not part of it) read from bit `sh` on (wave-uniform: 2 * (CL - q1)); returns the pair's result
// word (NX_REJECT, or nmiss | NX_DUP | NX_ACC0 | window << 20 | read slot << 24).
//   exact0: windows that take part for this read (minus the ones the pos-0 rule excludes)
template <int RW, bool W2>
DEV uint32_t ctx_compare(const uint32_t* __restrict__ r, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                         uint32_t c4, uint32_t c5, uint32_t c6, uint32_t c7, uint32_t sh, uint32_t k,
                         const MatchParams* __restrict__ mp, int W, int ww, int win0, int win1, uint32_t exact0,
                         uint32_t budget, uint32_t slot) {
  const uint32_t c[10] = {c0, c1, c2, c3, c4, c5, c6, c7 & 0xFFFFu, 0u, 0u};
  uint32_t t[RW - 1];
  const uint32_t bs = sh & 31u;
  switch (__builtin_amdgcn_readfirstlane((int)(sh >> 5))) {
#define MUSC_CTX_CASE(WO)                                                                      \
  case WO:                                                                                     \
    _Pragma("unroll") for (int j = 0; j < RW - 1; j++) {                                       \
      const uint32_t lo = (j + WO < 8) ? c[j + WO < 8 ? j + WO : 8] : 0u;                       \
      const uint32_t hi = (j + WO + 1 < 8) ? c[j + WO + 1 < 8 ? j + WO + 1 : 8] : 0u;           \
      t[j] = __funnelshift_r(lo, hi, bs);                                                      \
    }                                                                                          \
    break;
    MUSC_CTX_CASE(0) MUSC_CTX_CASE(1) MUSC_CTX_CASE(2) MUSC_CTX_CASE(3)
    MUSC_CTX_CASE(4) MUSC_CTX_CASE(5) MUSC_CTX_CASE(6) MUSC_CTX_CASE(7)
#undef MUSC_CTX_CASE
    default:
#pragma unroll
      for (int j = 0; j < RW - 1; j++) t[j] = 0u;
      break;
  }
  const uint32_t len = r[RW - 1] & 0xFFFFu;
  const int len2 = 2 * (int)len;
  uint32_t nx = 0, exact = exact0;
#pragma unroll
  for (int j = 0; j < RW - 1; j++) {
    const uint32_t x = r[j] ^ t[j];
    uint32_t d = (x | (x >> 1)) & 0x55555555u;
    const int rem = len2 - 32 * j;
    d &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
    nx += __popc(d);
    if constexpr (W2) {
      if (d & window_word_mask(win0, ww, j)) exact &= ~1u;
      if (W > 1 && (d & window_word_mask(win1, ww, j))) exact &= ~2u;
    } else {
      for (int kk = 0; kk < W; kk++)
        if (d & window_word_mask(mp->win[kk], ww, j)) exact &= ~(1u << kk);
    }
  }
  // the reference's confirm for window k accepts the pair (it counts towards that window-key
  // block's MaxMatches); the tuple is reported here only if k is the first window that accepts it
  if (!(nx <= budget && ((exact >> k) & 1u))) return NX_REJECT;
  const bool first = (uint32_t)(__ffs(exact) - 1) == k;
  return (first ? nx : (nx | NX_DUP)) | NX_ACC0 | (k << 20) | (slot << 24);
}

// k_match -- muscato_screen + the join + muscato_confirm + the per-read best/MMTol filter for one
// tile of 256 reads per workgroup iteration (persistent workgroups).
//   phase A  a thread per read: record -> registers and LDS; per window the length gate +
//            CountDinuc >= MinDinuc (cmd/muscato_window_reads/main.go:106-118 ==
//            cmd/muscato_screen/main.go:174-185) and the bucket of the window key
//   phase B  per window, four rounds per wave: a QUAD of lanes fetches one 128-byte bucket (32
//            bytes per lane, two dwordx4, non-temporal: every line is requested once); lane 0 holds
//            count/ovf/gene/jx, lanes 1..3 hold one entry's context each and run cdiff on it where
//            it arrived (ctx_compare) against the read's record from LDS.  All rounds of a window
//            share the shift 2*(CL - q1), so it is wave-uniform.
//   phase C  per window: the entries beyond a bucket's third (CtxEntry in E) as one flat list over
//            the workgroup, like k_screen's phase C
//   phase D  best[read] = min nmiss over its reported pairs (LDS atomicMin, filled during B/C),
//            tuples with nmiss <= best + MMTol per read (cmd/muscato_combine_windows/main.go:36-60;
//            all of them when apply_mmtol == 0), a scan over the tile's reads, and the tuples
//            themselves, read-major, into the workgroup's region of `stage`; k_compact closes the
//            gaps between tiles afterwards.
// Reported candidates wait in an LDS list (code, gene, pos); a tile with more than MATCH_LIST of
// them spills the rest to the workgroup's slice of `spill`.
// MaxMatches accounting as in k_confirm: block_mode 1 = count-min sketch per workgroup in LDS,
// 2 = exact global counters.
// counters (batch-local block = counters + 8): [0] valid windows, [1] entries compared (passed the
// fit rules), [3] index entries walked, [4] overflow entries walked, [5] largest spill any
// workgroup needed, [6] tuples staged, [7] largest number of tuples any workgroup staged;
// pass-level [1] reported pairs, [3] flags (1: stage region ran out, 4: spill region ran out),
// [6] a sketch cell reached block_thr.
template <int RW, bool W2>
__global__ __launch_bounds__(TILE, 5) void k_match(const uint32_t* __restrict__ rd, uint64_t r0, uint32_t n,
                                                   const MatchParams* __restrict__ mp,
                                                   const uint16_t* __restrict__ nmiss_tab,
                                                   const CtxBucket* __restrict__ T, const CtxEntry* __restrict__ E,
                                                   uint4* __restrict__ stage, uint64_t stage_cap,
                                                   uint4* __restrict__ spill, uint64_t spill_cap,
                                                   uint32_t* __restrict__ tbase, uint32_t* __restrict__ tcount2,
                                                   int block_mode, uint32_t block_thr,
                                                   uint32_t* __restrict__ block_table,
                                                   unsigned long long* __restrict__ counters) {
  extern __shared__ uint32_t s_dyn[];  // block_mode != 0: TILE * W per-(read, window) counters, then (mode 1) the sketch
  __shared__ uint32_t s_rec[TILE * RW];
  __shared__ uint32_t s_bb[CTX_MAX_W * TILE];  // bucket of (window, read), WB_NONE when the window takes no part
  __shared__ uint32_t s_valid[TILE], s_best[TILE];
  __shared__ uint32_t s_lcode[MATCH_LIST], s_lgene[MATCH_LIST], s_lpos[MATCH_LIST];
  __shared__ uint32_t s_oc[TILE], s_ovf[TILE], s_pref[TILE + 1];  // phase C; s_oc / s_ovf double as cnt / base in phase D
  __shared__ uint16_t s_own[MATCH_OWN];
  __shared__ uint32_t s_wsum[TILE / 64];
  __shared__ uint32_t s_nlist;
  __shared__ uint16_t s_nm[CONF_NM];
  uint32_t* const s_cnt = s_oc;
  uint32_t* const s_base = s_ovf;

  const int W = mp->W, ww = mp->ww, CL = mp->CL;
  const int win0 = mp->win[0], win1 = mp->win[1];
  const uint32_t q1zero = mp->q1zero_mask;
  uint32_t* const s_wcnt = s_dyn;
  uint32_t* const s_sketch = s_dyn + TILE * W;
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)mp->max_len ? nmiss_tab[t] : (uint16_t)0;
  if (block_mode == 1)
    for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) s_sketch[t] = 0;
  lds_barrier();

  const uint32_t ntiles = (n + TILE - 1) / TILE;
  const uint64_t region = stage_cap / gridDim.x, region0 = region * blockIdx.x;
  const uint64_t sregion = spill_cap / gridDim.x, sregion0 = sregion * blockIdx.x;
  uint64_t used = 0;      // tuples this workgroup has staged so far (uniform across the workgroup)
  uint32_t maxspill = 0;  // largest spill a tile of this workgroup needed
  unsigned long long nvalid = 0, ncand = 0, ncmp = 0, novf = 0, nrep = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) tcount2[ntiles] = 0;

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
    // ---- phase A
    {
      const uint32_t tida = opaque(threadIdx.x);
      const uint32_t i = tile * TILE + tida;
      const bool active = i < n;
      Rec<RW> rec;
      rec.load(rd + (r0 + (active ? i : 0)) * (uint64_t)RW, RW);
#pragma unroll
      for (int q = 0; q < RW / 4; q++)
        *reinterpret_cast<uint4*>(&s_rec[tida * RW + 4 * q]) =
            make_uint4(rec.w[4 * q], rec.w[4 * q + 1], rec.w[4 * q + 2], rec.w[4 * q + 3]);
      const int len = (int)rec.len();
      uint32_t valid = 0;
      for (int k = 0; k < W; k++) {
        uint32_t b = WB_NONE;
        if (active) {
          const uint32_t q1 = (uint32_t)mp->win[k], q2 = q1 + (uint32_t)ww;
          bool pt = (uint32_t)len >= q2;
          if (pt && mp->min_dinuc > 0) pt = rec_count_dinuc(rec, rec, false, q1, ww) >= mp->min_dinuc;
          if (pt) {
            b = rec_bucket(rec, rec, false, q1, ww, mp->bits, mp->direct);
            valid |= 1u << k;
          }
        }
        s_bb[k * TILE + tida] = b;
      }
      s_valid[tida] = valid;
      s_best[tida] = 0xFFFFFFFFu;
      nvalid += __popc(valid);
      if (tida == 0) s_nlist = 0;
      if (block_mode)
        for (uint32_t t = tida; t < TILE * (uint32_t)W; t += TILE) s_wcnt[t] = 0;
    }
    lds_barrier();

    // one reported candidate per set lane of the vote: a wave claims its list slots with one LDS
    // atomic and fills them in lane order
    auto report = [&](uint32_t w, uint32_t gene, uint32_t pos) {
      const bool acc = w != NX_REJECT;
      if (acc && block_mode) atomicAdd(&s_wcnt[(w >> 24) * W + ((w >> 20) & 15u)], 1u);
      const bool rep = acc && !(w & NX_DUP);
      const unsigned long long vote = __ballot(rep);
      if (vote == 0) return;
      if (rep) atomicMin(&s_best[w >> 24], w & 0xFFFFu);
      uint32_t first = 0;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      if (below == 0 && rep) first = atomicAdd(&s_nlist, (uint32_t)__popcll(vote));
      first = __builtin_amdgcn_readlane(first, __builtin_ctzll(vote));
      if (!rep) return;
      nrep++;
      const uint32_t slot = first + below;
      if (slot < MATCH_LIST) {
        s_lcode[slot] = w;
        s_lgene[slot] = gene;
        s_lpos[slot] = pos;
      } else if (slot - MATCH_LIST < sregion) {
        spill[sregion0 + (slot - MATCH_LIST)] = make_uint4(w, gene, pos, 0u);
      }
    };

    for (int k = 0; k < W; k++) {
      const int q1 = mp->win[k];
      const uint32_t sh = 2u * (uint32_t)(CL - q1);
      // ---- phase B: four rounds, all loads issued before the first is used
      {
        const uint32_t tidb = opaque(threadIdx.x);
        const uint32_t lane = tidb & 63, wid = tidb >> 6, part = lane & 3;
        uint4 va[4], vb[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const uint32_t ri = wid * 64 + rr * 16 + (lane >> 2);
          const uint32_t b = s_bb[k * TILE + ri];
          va[rr] = make_uint4(0, 0, 0, 0);
          vb[rr] = make_uint4(0, 0, 0, 0);
          if (b != WB_NONE) {
            const u32x4_v* p = reinterpret_cast<const u32x4_v*>(T + b) + 2 * part;
            const u32x4_v x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1);
            va[rr] = make_uint4(x.x, x.y, x.z, x.w);
            vb[rr] = make_uint4(y.x, y.y, y.z, y.w);
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const uint32_t ri = wid * 64 + rr * 16 + (lane >> 2);
          // block 0 sits in the quad's first lane (quad_perm [0,0,0,0]); every lane executes the moves
#define MUSC_Q0(X) (uint32_t)__builtin_amdgcn_mov_dpp((int)(X), 0x00, 0xF, 0xF, true)
          const uint32_t cnt = MUSC_Q0(va[rr].x);
          const uint32_t g0 = MUSC_Q0(va[rr].z), g1 = MUSC_Q0(va[rr].w), g2 = MUSC_Q0(vb[rr].x);
          const uint32_t j0 = MUSC_Q0(vb[rr].y), j1 = MUSC_Q0(vb[rr].z), j2 = MUSC_Q0(vb[rr].w);
#undef MUSC_Q0
          if (part == 0) {
            ncand += cnt;
            s_oc[ri] = cnt > CTX_INLINE ? cnt - CTX_INLINE : 0u;
            s_ovf[ri] = va[rr].y;
          }
          const uint32_t gene = part == 1 ? g0 : (part == 2 ? g1 : g2);
          const uint32_t jx = part == 1 ? j0 : (part == 2 ? j1 : j2);
          const uint32_t* __restrict__ r = &s_rec[ri * RW];
          uint32_t rr_w[RW];
#pragma unroll
          for (int q = 0; q < RW / 4; q++) {
            const uint4 a = *reinterpret_cast<const uint4*>(r + 4 * q);
            rr_w[4 * q] = a.x; rr_w[4 * q + 1] = a.y; rr_w[4 * q + 2] = a.z; rr_w[4 * q + 3] = a.w;
          }
          const int rlen = (int)(rr_w[RW - 1] & 0xFFFFu);
          uint32_t z = 0;
          bool ok = part >= 1 && part - 1 < cnt;
          if (ok) ok = ctx_fit(jx, vb[rr].w >> 16, q1, ww, rlen, &z);
          uint32_t w = NX_REJECT;
          if (ok) {
            ncmp++;
            uint32_t exact0 = s_valid[ri];
            if (z) exact0 &= ~q1zero;
            const uint32_t budget = rlen < CONF_NM ? s_nm[rlen] : nmiss_tab[rlen];
            w = ctx_compare<RW, W2>(rr_w, va[rr].x, va[rr].y, va[rr].z, va[rr].w, vb[rr].x, vb[rr].y, vb[rr].z,
                                    vb[rr].w, sh, (uint32_t)k, mp, W, ww, win0, win1, exact0, budget, ri);
          }
          report(w, gene, jx - (uint32_t)q1);
        }
      }
      lds_barrier();
      // ---- phase C: this window's overflow entries as one flat list
      {
        const uint32_t tidc = opaque(threadIdx.x);
        const uint32_t oc = s_oc[tidc];
        uint32_t total = 0;
        const uint32_t pre = wg_scan(oc, &total);
        if (total != 0) {  // uniform
          s_pref[tidc] = pre;
          if (tidc == TILE - 1) s_pref[TILE] = total;
          for (uint32_t e = 0; e < oc && pre + e < MATCH_OWN; e++) s_own[pre + e] = (uint16_t)tidc;
          lds_barrier();
          novf += oc;
          for (uint32_t t0 = 0; t0 < total; t0 += TILE) {
            const uint32_t t = t0 + tidc;
            bool ok = t < total;
            uint32_t seg = 0, z = 0, gene = 0, jx = 0, w = NX_REJECT;
            if (ok) {
              if (t < MATCH_OWN) {
                seg = s_own[t];
              } else {  // rare: largest seg with s_pref[seg] <= t
                uint32_t lo = 0, hi = TILE;
                while (hi - lo > 1) {
                  const uint32_t mid = (lo + hi) / 2;
                  if (s_pref[mid] <= t) lo = mid; else hi = mid;
                }
                seg = lo;
              }
              const uint32_t* __restrict__ pe =
                  reinterpret_cast<const uint32_t*>(E + ((uint64_t)s_ovf[seg] + (t - s_pref[seg])));
              const uint2 hd = *reinterpret_cast<const uint2*>(pe);
              const u32x4_u x = *reinterpret_cast<const u32x4_u*>(pe + 2);
              const u32x4_u y = *reinterpret_cast<const u32x4_u*>(pe + 6);
              gene = hd.x;
              jx = hd.y;
              uint32_t rr_w[RW];
#pragma unroll
              for (int q = 0; q < RW / 4; q++) {
                const uint4 a = *reinterpret_cast<const uint4*>(&s_rec[seg * RW + 4 * q]);
                rr_w[4 * q] = a.x; rr_w[4 * q + 1] = a.y; rr_w[4 * q + 2] = a.z; rr_w[4 * q + 3] = a.w;
              }
              const int rlen = (int)(rr_w[RW - 1] & 0xFFFFu);
              ok = ctx_fit(jx, y.w >> 16, q1, ww, rlen, &z);
              if (ok) {
                ncmp++;
                uint32_t exact0 = s_valid[seg];
                if (z) exact0 &= ~q1zero;
                const uint32_t budget = rlen < CONF_NM ? s_nm[rlen] : nmiss_tab[rlen];
                w = ctx_compare<RW, W2>(rr_w, x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w, sh, (uint32_t)k, mp, W, ww, win0,
                                        win1, exact0, budget, seg);
              }
            }
            report(w, gene, jx - (uint32_t)q1);
          }
        }
        lds_barrier();  // s_oc / s_ovf / s_pref / s_own are reused by the next window
      }
    }

    // ---- phase D: per-read selection and the tuples
    {
      const uint32_t tidd = opaque(threadIdx.x);
      const int lane = tidd & 63, wid = tidd >> 6;
      const uint32_t nl = s_nlist;
      const uint32_t nspill = nl > MATCH_LIST ? nl - MATCH_LIST : 0u;
      const bool spill_ok = nspill <= sregion;
      if (nspill > maxspill) maxspill = nspill;
      s_cnt[tidd] = 0;
      if (block_mode) {
        for (uint32_t t = tidd; t < TILE * (uint32_t)W; t += TILE) {
          const uint32_t cw = s_wcnt[t];
          if (!cw) continue;
          const uint32_t rl = t / W, kk = t % W;
          const uint64_t h = mix64(((uint64_t)kk << 32) | s_bb[kk * TILE + rl]);
          if (block_mode == 1) atomicAdd(&s_sketch[h >> (64 - MATCH_SKETCH_BITS)], cw);
          else atomicAdd(&block_table[h >> (64 - BLOCK_TABLE_BITS)], cw);
        }
      }
      lds_barrier();
      const uint32_t mmtol = (uint32_t)mp->mmtol;
      const bool apply = mp->apply_mmtol != 0;
      auto item = [&](uint32_t j, uint32_t* gene, uint32_t* pos) -> uint32_t {
        if (j < MATCH_LIST) {
          *gene = s_lgene[j];
          *pos = s_lpos[j];
          return s_lcode[j];
        }
        const uint4 v = spill[sregion0 + (j - MATCH_LIST)];
        *gene = v.y;
        *pos = v.z;
        return v.x;
      };
      const uint32_t nuse = spill_ok ? nl : (nl < MATCH_LIST ? nl : MATCH_LIST);
      if (nspill) __threadfence_block();
      for (uint32_t j = tidd; j < nuse; j += TILE) {
        uint32_t g, p;
        const uint32_t w = item(j, &g, &p);
        const uint32_t rl = w >> 24;
        const uint32_t thr = apply ? s_best[rl] + mmtol : 0xFFFFu;
        if ((w & 0xFFFFu) <= thr) atomicAdd(&s_cnt[rl], 1u);
      }
      lds_barrier();
      const uint32_t cnum = s_cnt[tidd];
      uint32_t inc = cnum;
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
      s_base[tidd] = woff + inc - cnum;
      s_cnt[tidd] = 0;  // now the arrival counter of the read
      const uint64_t base = region0 + used;
      const bool fits = spill_ok && used + total <= region;
      if (tidd == 0) {
        tbase[tile] = (uint32_t)base;
        tcount2[tile] = fits ? total : 0u;
      }
      lds_barrier();
      if (fits && total) {
        for (uint32_t j = tidd; j < nuse; j += TILE) {
          uint32_t g, p;
          const uint32_t w = item(j, &g, &p);
          const uint32_t rl = w >> 24, v = w & 0xFFFFu;
          const uint32_t thr = apply ? s_best[rl] + mmtol : 0xFFFFu;
          if (v > thr) continue;
          const uint32_t ord = atomicAdd(&s_cnt[rl], 1u);
          stage[base + s_base[rl] + ord] = make_uint4((uint32_t)(r0 + tile * TILE + rl), g, p, v);
        }
      }
      used += total;
      lds_barrier();  // the next tile resets the LDS state
    }
  }
  block_add_u64(nvalid, &counters[8 + 0]);
  block_add_u64(ncmp, &counters[8 + 1]);
  block_add_u64(ncand, &counters[8 + 3]);
  block_add_u64(novf, &counters[8 + 4]);
  block_add_u64(nrep, &counters[1]);
  if (threadIdx.x == 0) {
    atomicAdd(&counters[8 + 6], (unsigned long long)(used <= region ? used : 0));
    atomicMax(&counters[8 + 7], (unsigned long long)used);
    atomicMax(&counters[8 + 5], (unsigned long long)maxspill);
    if (used > region) atomicOr(&counters[3], 1ull);
    if (maxspill > sregion) atomicOr(&counters[3], 4ull);
  }
  if (block_mode == 1) {
    lds_barrier();
    uint32_t hot = 0;
    for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) hot |= s_sketch[t] >= block_thr;
    if (__any(hot) && (threadIdx.x & 63) == 0) atomicOr(&counters[6], 1ull);
  }
}
