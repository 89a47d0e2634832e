// k_match_t -- the fused screen + confirm + select kernel on context buckets with the comparisons done
// IN THE LANE THAT OWNS THE READ (device code; compiled in match_lane_rw*.hip, declared in
// kernels_match_lane_inst.hpp; shares the bucket layout, parameter block and fit rules of
// kernels_match.hpp).
//
// k_match compares where a bucket line arrives (a quad of lanes per probe); round 2's k_match_d
// moved the entries that exist onto a stack in LDS and popped 64 of them per comparison pass.  Both
// spend most of their instructions on getting an entry and "its" read into the same lane: DPP moves,
// votes, stack writes, per-entry record reads and funnel shifts, owner tables -- 2 640 instructions
// per wave-tile for 237 comparisons that need about 50 instructions each.  Here the bucket line is
// TRANSPOSED instead: the quad that fetched a line (coalesced: 32 bytes per lane) writes it into a
// per-wave line buffer in LDS as it arrived, and once the 64 lines of a window are there, lane p
// reads line p -- all 128 bytes, eight ds_read_b128 -- into its own registers, where the read's
// record already is.  Everything cdiff needs is then in one lane:
//   * the read's image for the window is built once per (read, window) from registers
//     (read_image), not once per entry from LDS;
//   * the three inline entries of the line are compared by three straight-line blocks with no
//     memory access in them: no stack, no entry copies, no per-entry meta / record / mask lookups;
//   * best-per-read and the MaxMatches counters of (window, read) are registers of that lane.
// 56 % of the slots of those blocks hold an entry (1.67 inline entries per probe on cfg3) -- less
// than a dense pass, but a slot costs ~55 vector instructions and nothing else.
// The line buffer is XOR-swizzled at 16-byte granularity (chunk c of line p sits at slot
// c ^ ((p >> 1) & 7) ^ (p & 1)), which makes both the quads' writes and the lanes' whole-line reads
// bank-conflict free.
//
// WHAT THE WAITS LOOK LIKE decides the schedule.  On gfx9 hipcc waits for a vector load with
// s_waitcnt vmcnt(0) whenever a store may be pending or the loads after it sit under a branch (this
// kernel: always) -- every wait drains everything the wave has in flight.  A ring of loads deeper
// than "what the next wait needs" is therefore useless, and a wait right after an issue exposes a
// whole memory latency.  So:
//   * the ring is ONE window deep: after the 64 lines of window g have been written to LDS its four
//     register slots are refilled with window g + 1 (the same wave-tile's next window, or the next
//     wave-tile's first), which is what the next wait is for anyway;
//   * nothing else is waited for fresh: the records of wave-tile t + 1 are fetched behind the last
//     refill of tile t - 1 and used after the first wait of tile t; the entries beyond a bucket's
//     third (CtxEntry in E, 10 % of the entries walked on cfg3) are listed when tile t's last window
//     has arrived (a lane per entry, which takes the read's record along), LOADED right after the
//     first wait of tile t + 1 -- nothing is in flight then, so the compiler has no reason to wait
//     for anything -- and compared after tile t + 1's second wait; tile t's per-read selection
//     (phase D) follows that pass, so the candidate lists are double-buffered;
//   * loads go straight into the registers their consumer reads (a load under a branch, or one
//     whose result is re-packed, becomes "load into a temporary, wait, copy");
//   * and no register spills: a scratch reload is a vector load whose wait drains the ring.
// A wave thus alternates "64 lines in flight" with "compare a window": with two waves per SIMD one
// computes while the other waits.
//
// LDS per workgroup of four waves: 4 x 16.2 KB + the MaxMatches sketch = 73 KB: two workgroups per CU
// (registers: ~230 of the 256 two waves per SIMD may use).
//
// WIDE (template flag): the same kernel on CtxBucketW lines -- two inline entries of 200 context bases
// (13 words) instead of three of 120 (8 words), 60-byte overflow entries -- for runs whose reads do not
// fit 120 bases around every window (Windows 0,20,40 at 100 bp; 150-bp reads).  Everything that is
// "eight words" above is NW words here; a read's X are listed as three 8-bit positions (XPos<true>).
// Records of 12 words and more keep 64 reported candidates in LDS instead of 96 so that two
// workgroups still fit a CU; records of 16 words (reads of 177-200 bases) run one workgroup per CU.
#pragma once
#include "kernels_match_lane_inst.hpp"
#if defined(MUSC_LANE_DBG) && !defined(MUSC_LANE_DBG_SLOTS)
#define MUSC_LANE_DBG_SLOTS 3
#endif

// The read's image for a window: the record shifted left by sh bits (wave-uniform), NW words -- read_image
// of kernels_match.hpp for either context width
template <int RW, int NW>
DEV void read_image_n(const Rec<RW>& rec, uint32_t sh, uint32_t (&img)[NW]) {
  const uint32_t bs = sh & 31u;
#define MUSC_IMG_W(Q) (((Q) >= 0 && (Q) < RW - 1) ? rec.w[((Q) >= 0 && (Q) < RW - 1) ? (Q) : 0] : 0u)
#define MUSC_IMG_CASE(WO)                                                  \
  case WO:                                                                 \
    _Pragma("unroll") for (int j = 0; j < NW; j++) {                       \
      const uint32_t hi = MUSC_IMG_W(j - WO), lo = MUSC_IMG_W(j - WO - 1); \
      img[j] = bs ? ((hi << bs) | (lo >> (32u - bs))) : hi;                \
    }                                                                      \
    break;
  switch (__builtin_amdgcn_readfirstlane((int)(sh >> 5))) {
    MUSC_IMG_CASE(0) MUSC_IMG_CASE(1) MUSC_IMG_CASE(2) MUSC_IMG_CASE(3)
    MUSC_IMG_CASE(4) MUSC_IMG_CASE(5) MUSC_IMG_CASE(6) MUSC_IMG_CASE(7)
    MUSC_IMG_CASE(8) MUSC_IMG_CASE(9) MUSC_IMG_CASE(10) MUSC_IMG_CASE(11) MUSC_IMG_CASE(12)
    default:
#pragma unroll
      for (int j = 0; j < NW; j++) img[j] = 0u;
      break;
  }
#undef MUSC_IMG_CASE
#undef MUSC_IMG_W
}


// a tuple on its way out (tried in r04: the non-temporal form of this store -- no difference, profiles/r04_ab_shape_spec_dma.txt)
DEV void put_tuple(uint4* __restrict__ dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { *dst = make_uint4(a, b, c, d); }

template <int RW, int W, int XM, bool WIDE, int SG>
__global__ __launch_bounds__(TILE, MATCHT_WAVES) void k_match_t(const uint32_t* __restrict__ rd, uint64_t r0, uint32_t n,
                                                                const MatchParams* __restrict__ mp,
                                                                const uint16_t* __restrict__ nmiss_tab,
                                                                const CtxBucket* __restrict__ T, const CtxEntry* __restrict__ E,
                                                                uint4* __restrict__ stage, uint64_t stage_cap,
                                                                uint4* __restrict__ spill, uint64_t spill_cap,
                                                                uint32_t* __restrict__ tbase, uint32_t* __restrict__ tcount2,
                                                                int block_mode, uint32_t block_thr,
                                                                uint32_t* __restrict__ block_table,
                                                                unsigned long long* __restrict__ counters,
                                                                const uint4* __restrict__ pstage, const uint32_t* __restrict__ ptcount2,
                                                                const uint32_t* __restrict__ ptpre, uint32_t pnwt,
                                                                uint4* __restrict__ hits, uint64_t hits_cap,
                                                                const uint32_t* __restrict__ rdx) {
  static_assert(W >= 1 && W <= CTX_MAX_W, "context buckets serve at most CTX_MAX_W windows");
  // XM: 0 = neither side holds an X; 1 = reads may (their xpos words, rdx); 2 = the database does (flagged
  // entries consult its mask plane) and reads may
  constexpr bool RX = XM != 0, DBX = XM == 2;
  constexpr int NWAVE = TILE / 64;
  // the bucket layout: three inline entries with 8 context words (120 bases), or -- WIDE -- two with 13 (200 bases)
  constexpr int NIN = WIDE ? CTXW_INLINE : CTX_INLINE;
  // (records of 12 words and more: a shorter candidate list keeps two workgroups within a CU's 160 KB)
  constexpr uint32_t WLIST = RW >= 12 && MATCHT_WLIST > 64 ? 64u : (uint32_t)MATCHT_WLIST;
  typedef XPos<WIDE> XP;  // how a read's xpos word lists its X
  constexpr int NW = WIDE ? CTXW_WORDS : 8;
  constexpr int NQ = WIDE ? 3 : 2;       // an overflow entry (gene, jx, ctx[NW]) = NQ 16-byte pieces + NT words
  constexpr int NT = WIDE ? 3 : 2;
  // block_mode != 0: 2 (tile parity) x NWAVE x W x 64 counters of the overflow passes, then (mode 1) the sketch
  extern __shared__ uint32_t s_dyn[];
  __shared__ uint4 s_line[NWAVE][64 * 8];                 // the window's 64 bucket lines, swizzled
  __shared__ __attribute__((aligned(16))) uint32_t s_rec[2][NWAVE][WT * RW];  // the records of this wave-tile and of the next
  __shared__ uint32_t s_meta[2][NWAVE][WT];               // length | budget << 17 | valid windows << 24; this wave-tile's and the one before
  __shared__ uint32_t s_xp[RX ? 2 : 1][RX ? NWAVE : 1][RX ? WT : 1];  // reads with X: their xpos words, likewise
  __shared__ uint32_t s_best[2][NWAVE][WT];               // smallest mismatch count the overflow pass reported per read
  __shared__ uint3 s_list[2][NWAVE][WLIST];        // reported candidates: result word, gene, position
  __shared__ uint32_t s_cb[NWAVE][2 * WT];                // phase D: cnt[64], base[64]
  __shared__ uint32_t s_oix[NWAVE][WT];                   // overflow entries: item -> its entry in E
  __shared__ uint8_t s_own[NWAVE][WT];                    //                   item -> window * 64 + read slot
  __shared__ uint16_t s_nm[CONF_NM];
  // Three windows, no X, records of up to eight words: the previous wave-tile's bucket numbers -- kept for ONE use, the
  // MaxMatches block accounting in phase D one wave-tile later -- live in LDS, not in registers.  These instances sit at
  // 256 VGPRs and the allocator spilled exactly those values; a scratch reload is a vector memory operation, so its wait
  // (`vmcnt(0)`, per window, inside every wave-tile) drained the next wave-tile's bucket lines already in flight
  // (r04; the same mechanism as in k_screen_t).  Only where 3 KB more LDS keep two workgroups on a CU.
  constexpr bool BB_LDS = (W == 3 && XM == 0 && RW <= 8);
  __shared__ uint32_t s_bbp[BB_LDS ? NWAVE : 1][BB_LDS ? W : 1][BB_LDS ? WT : 1];

  // the run's parameters: scalars for the whole kernel -- or, in an instance specialised for one geometry
  // (SpecGeom<SG>, kernels_match_lane_inst.hpp), compile-time constants: the mask tables become immediates, the
  // image shift a constant, and a third of the scalar instructions and of the spilled scalars go away
  typedef SpecGeom<SG> SGm;
  constexpr bool SPEC = SGm::on;
  static_assert(!SPEC || (SGm::nwin == W && !WIDE && XM == 0 && SGm::ww <= 15), "a specialised instance is built for its geometry's window count, 120-base buckets, no X, a direct table of one-word keys");
  constexpr int S_WW = SGm::ww, S_CL = SGm::CL, S_L = SGm::L, S_MIND = SGm::min_dinuc;
  constexpr int S_WIN[CTX_MAX_W] = {SGm::win[0], SGm::win[1], SGm::win[2], SGm::win[3]};
  if constexpr (SPEC) {
    // the host compared the geometry before it chose this instance (spec_geom_matches); a mismatch here is a bug
    // there: refuse -- no table access, no tuple -- and say so (flag 8 of the pass-level flag word).  r03's fault
    // (gpurun_out/var.err) was a development build of this kind let loose on cfg2's hashed table.
    bool same = mp->ww == S_WW && mp->CL == S_CL && mp->min_dinuc == S_MIND && mp->direct == 1 && mp->bits == 2 * S_WW && mp->W == W;
#pragma unroll
    for (int k = 0; k < W; k++) same = same && mp->win[k] == S_WIN[k] && mp->need[k] == (1u << k) - 1u;
    if (!same) {
      if (threadIdx.x == 0) atomicOr(&counters[3], 8ull);
      return;
    }
  }
  const int ww = SPEC ? S_WW : mp->ww, CL = SPEC ? S_CL : mp->CL, min_dinuc = SPEC ? S_MIND : mp->min_dinuc,
            direct = SPEC ? 1 : mp->direct, bits = SPEC ? 2 * S_WW : mp->bits;
  const uint32_t q1zero = mp->q1zero_mask;
  int win[W];
#pragma unroll
  for (int k = 0; k < W; k++) win[k] = SPEC ? S_WIN[k] : mp->win[k];
  uint32_t* const s_sketch = s_dyn + 2 * NWAVE * WT * W;
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)mp->max_len ? nmiss_tab[t] : (uint16_t)0;
  if (block_mode) {
    for (uint32_t t = threadIdx.x; t < 2u * NWAVE * WT * W; t += TILE) s_dyn[t] = 0;
    if (block_mode == 1)
      for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) s_sketch[t] = 0;
  }
  s_best[0][threadIdx.x >> 6][threadIdx.x & 63] = 0xFFFFFFFFu;
  s_best[1][threadIdx.x >> 6][threadIdx.x & 63] = 0xFFFFFFFFu;
  if (blockIdx.x == 0 && threadIdx.x == 0) tcount2[(n + WT - 1) / WT] = 0;
  __syncthreads();

  const uint32_t nwt = (n + WT - 1) / WT;
  const uint32_t gw = blockIdx.x * NWAVE + (threadIdx.x >> 6), nw = gridDim.x * NWAVE;
  const uint64_t region = stage_cap / nw, region0 = region * gw;
  // (a wave's spill region has a half per wave-tile parity: the lists of the wave-tile in hand and of the
  // one before are alive together)
  const uint64_t sregion = spill_cap / nw / 2, sregion_w = 2 * sregion * gw;
  uint64_t used = 0;      // tuples this wave has staged so far (wave-uniform)
  uint32_t maxspill = 0;  // largest spill a wave-tile of this wave needed
  uint32_t nvalid = 0, ncand = 0, ncmp = 0, novf = 0, nrep = 0;  // per lane: far below 2^32
  const uint32_t mmtol = (uint32_t)mp->mmtol;
  const bool apply = mp->apply_mmtol != 0;

  // a lane fetches its own read's record (tried in r04: the tile's records as RW / 4 loads of 1 KB contiguous,
  // transposed through s_rec -- 1.105 ms per cfg3 launch against 1.08-1.09 for this form: the lines are shared
  // by neighbouring lanes of one load anyway, and the transposition costs a second LDS round trip per tile)
  auto fetch = [&](uint32_t wt, Rec<RW>& rec) __attribute__((always_inline)) {
    const uint32_t i = wt * WT + (opaque(threadIdx.x) & 63);
    rec.load(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW, RW);
  };
  // phase A of wave-tile wt, a lane per read: per window the length gate + CountDinuc >= MinDinuc
  // (cmd/muscato_window_reads/main.go:106-118 == cmd/muscato_screen/main.go:174-185) and the bucket
  // of the window key -> bb[] (WB_NONE when the window takes no part).  Returns the windows that take part.
  auto phase_a = [&](uint32_t wt, const Rec<RW>& rec, uint32_t& xw, uint32_t (&bb)[W]) __attribute__((always_inline)) -> uint32_t {
    const uint32_t lane = opaque(threadIdx.x) & 63;
    const bool active = wt * WT + lane < n;
    const int len = (int)rec.len();
    uint32_t valid = 0;
    // reads with X: windows that hold one never probe; a read with more X than fit the word has no tuples
    xw = 0;
    uint32_t xwin = 0;  // windows barred by an X
    if constexpr (RX) {
      if (active && rec.has_x()) xw = rdx[r0 + wt * WT + lane];
      const uint32_t xc = XPOS_CNT(xw);
      if (xc > XP::MAX) xwin = 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < W; k++) {
        const uint32_t q1 = (uint32_t)win[k];
#pragma unroll
        for (int q = 0; q < (int)XP::MAX; q++)
          if ((uint32_t)q < xc && XP::at(xw, q) - q1 < (uint32_t)ww) xwin |= 1u << k;
      }
    }
    if (ww <= 16 && direct) {
      // the usual case: the window key is one 32-bit word
      const uint32_t kmask = ww == 16 ? 0xFFFFFFFFu : ((1u << (2 * ww)) - 1u);
#pragma unroll
      for (int k = 0; k < W; k++) {
        const uint32_t q1 = (uint32_t)win[k], q2 = q1 + (uint32_t)ww;
        const uint32_t key = (uint32_t)rec.ext(2 * q1) & kmask;
        bool pt = active && (uint32_t)len >= q2 && !((xwin >> k) & 1u);
#ifdef MUSC_LANE_DBG
        if (min_dinuc > 0 && !(MUSC_LANE_DBG & 16)) pt = pt && key_dinucs16(key, ww) >= min_dinuc;
#else
        if (min_dinuc > 0) pt = pt && key_dinucs16(key, ww) >= min_dinuc;
#endif
        bb[k] = pt ? __brev(key) >> (32 - 2 * ww) : WB_NONE;
        valid |= pt ? 1u << k : 0u;
      }
    } else {
#pragma unroll
      for (int k = 0; k < W; k++) {
        uint32_t b = WB_NONE;
        const uint32_t q1 = (uint32_t)win[k], q2 = q1 + (uint32_t)ww;
        if (active) {
          bool pt = (uint32_t)len >= q2 && !((xwin >> k) & 1u);
          if (pt && min_dinuc > 0)
            pt = (ww <= 16 ? rec_count_dinuc16(rec, q1, ww) : rec_count_dinuc(rec, rec, false, q1, ww)) >= min_dinuc;
          if (pt) {
            b = rec_bucket(rec, rec, false, q1, ww, bits, direct);
            valid |= 1u << k;
          }
        }
        bb[k] = b;
      }
    }
    nvalid += __popc(valid);
    return valid;
  };
  // the bucket loads of one window: eight loads of EIGHT WHOLE LINES each -- eight lanes per line, 16 contiguous
  // bytes per lane, so a load instruction asks for eight full 128-byte lines (r04: the quad-per-line shape of r03,
  // 32 bytes per lane in two loads, made every line two strided half-requests and cost 9 % of the launch;
  // profiles/r04_ub_dma_lines.txt has the bare access patterns side by side).  bbk = this lane's read's bucket
  // for that window; the eight lanes of a line get it from the lane that owns the read (ds_bpermute: the LDS
  // crossbar, no memory), all eight before the first address is formed -- one round trip, not eight.  Lane l of
  // load i takes the chunk that belongs at slot l & 7 of line 8 i + (l >> 3) in the swizzled line buffer, so the
  // arrival writes 1 KB contiguously per load.  A probe that takes no part fetches bucket 0 (always there); its
  // owner ignores the line (the header's count is masked).
  auto issue_window = [&](uint32_t bbk, uint4 (&a)[4], uint4 (&b2)[4]) __attribute__((always_inline)) {
    const uint32_t lane = opaque(threadIdx.x) & 63;
    uint32_t bq[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bq[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((i * 8 + (lane >> 3)) * 4), (int)bbk);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      // the chunk: (lane & 7) ^ sw(p) for line p = 8 i + (lane >> 3), sw(p) = ((p >> 1) & 7) ^ (p & 1); (p >> 1) & 7 =
      // (lane >> 4) | ((i & 1) << 2) and p & 1 = (lane >> 3) & 1: one lane constant, bit 2 flipped on the odd loads
      const uint32_t c = ((lane & 7u) ^ (lane >> 4) ^ ((lane >> 3) & 1u)) ^ (((uint32_t)i & 1u) << 2);
#ifdef MUSC_LANE_DBG
      if (MUSC_LANE_DBG & 2) continue;
#endif
      const u32x4_v* p = reinterpret_cast<const u32x4_v*>(T + (bq[i] != WB_NONE ? bq[i] : 0u)) + c;
      const u32x4_v x = __builtin_nontemporal_load(p);
      if (i < 4) a[i] = make_uint4(x.x, x.y, x.z, x.w);
      else b2[i - 4] = make_uint4(x.x, x.y, x.z, x.w);
    }
  };

  // The tuples the PREVIOUS batch's launch staged (pstage != nullptr: same grid, same regions, the
  // other stage buffer) move to their final place in `hits` from inside this launch (the protocol of match_ctx_pass)
  bool pcopy = pstage != nullptr;
  unsigned long long pbase = 0;
  if (pcopy) {
    pbase = counters[2];
    if (pbase + ptpre[pnwt] > hits_cap) {  // cannot happen on a sized pass
      if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&counters[3], 2ull);
      pcopy = false;
    }
  }
  uint64_t pused = 0;  // tuples of the previous batch this wave has moved (its region is consumed in order)
  auto pcopy_begin = [&](uint32_t wt, uint4& v, uint32_t& m, uint32_t& d) __attribute__((always_inline)) {
    m = 0;
    d = 0;
    if (pcopy && wt < pnwt) {
      const uint32_t wtu = (uint32_t)__builtin_amdgcn_readfirstlane((int)wt);
      m = ptcount2[wtu];
      d = ptpre[wtu];
      const uint32_t lane = opaque(threadIdx.x) & 63;
      if (lane < m) v = pstage[region0 + pused + lane];
    }
  };
  auto pcopy_end = [&](const uint4& v, uint32_t m, uint32_t d) __attribute__((always_inline)) {
    if (!m) return;
    const uint32_t lane = opaque(threadIdx.x) & 63;
    uint4* __restrict__ dst = hits + pbase + d;
    if (lane < m) put_tuple(&dst[lane], v.x, v.y, v.z, v.w);
    for (uint32_t i = 64 + lane; i < m; i += 64) dst[i] = pstage[region0 + pused + i];  // a tile with more than 64 tuples
    pused += m;
  };

  // What a comparison through window k reads of the host's tables (MatchParams), fetched in ONE batch
  // of scalar loads per (wave-tile, window) -- a load inside the comparison would stall every entry:
  //   lm    the length mask of a read of `len` bases placed through window k (one bit per base that
  //         takes part in cdiff, in the coordinates of the context stream): a row of the host's table
  //         when every read of the wave-tile has the same length ul, per-lane arithmetic otherwise
  //   need  the windows whose exactness the comparison has to establish (first-window rule), and
  //   wm    their masks
  struct WinTab {
    uint32_t lm[NW];
    uint32_t need;
    uint32_t wm[W][NW];
  };
  auto win_tab = [&](uint32_t ul, int k, uint32_t sh, uint32_t len, WinTab& tb) __attribute__((always_inline)) {
    if constexpr (SPEC) {
      // everything but the length is known: the masks are constants (the length mask for tiles of S_L-base reads)
      tb.need = (1u << k) - 1u;
      const int shc = 2 * (S_CL - S_WIN[k < W ? k : 0]);
#pragma unroll
      for (int kk = 0; kk < W; kk++)
#pragma unroll
        for (int j = 0; j < NW; j++)
          tb.wm[kk][j] = kk <= k ? bit_range_mask(shc + 2 * S_WIN[kk] - 32 * j, shc + 2 * (S_WIN[kk] + S_WW) - 32 * j) : 0u;
      if (ul == (uint32_t)S_L) {
#pragma unroll
        for (int j = 0; j < NW; j++) tb.lm[j] = 0x55555555u & bit_range_mask(shc - 32 * j, shc + 2 * S_L - 32 * j);
      } else {
#pragma unroll
        for (int j = 0; j < NW; j++) tb.lm[j] = 0x55555555u & bit_range_mask((int)sh - 32 * j, (int)sh + 2 * (int)len - 32 * j);
      }
      return;
    }
    tb.need = (uint32_t)__builtin_amdgcn_readfirstlane((int)mp->need[k]);
#pragma unroll
    for (int kk = 0; kk < W; kk++) {
      const uint32_t* __restrict__ row = mp->wm[k][kk];
#pragma unroll
      for (int j = 0; j < NW; j++) tb.wm[kk][j] = kk <= k ? (uint32_t)__builtin_amdgcn_readfirstlane((int)row[j]) : 0u;  // (need[k] has no window beyond k)
    }
    if (ul != 0xFFFFFFFFu) {
      const uint32_t* __restrict__ row = mp->lm[__builtin_amdgcn_readfirstlane((int)ul)][k];
#pragma unroll
      for (int j = 0; j < NW; j++) tb.lm[j] = row[j];
    } else {
#pragma unroll
      for (int j = 0; j < NW; j++) tb.lm[j] = 0x55555555u & bit_range_mask((int)sh - 32 * j, (int)sh + 2 * (int)len - 32 * j);
    }
  };
  // the X of a read in the image's coordinates
  auto x_mask = [&](uint32_t xw, uint32_t sh, uint32_t (&xm)[NW]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NW; j++) xm[j] = 0;
    if constexpr (RX) {
      if (__any(xw != 0)) {
        const uint32_t xc = XPOS_CNT(xw);
#pragma unroll
        for (int q = 0; q < (int)XP::MAX; q++) {
          const uint32_t b = sh + 2u * XP::at(xw, q);
          const uint32_t bit = (uint32_t)q < xc ? 1u << (b & 31u) : 0u;
#pragma unroll
          for (int j = 0; j < NW; j++) xm[j] |= (b >> 5) == (uint32_t)j ? bit : 0u;
        }
      }
    }
  };
  // one entry (context words c, c[7] still carrying the distance to the target end in its high half)
  // against a read whose image for window k is img: the fit rules (ctx_fit), cdiff
  // (cmd/muscato_confirm/main.go:151-159, 205-211: XOR, one bit per mismatching base, popcount) and,
  // from the same mismatch mask, which windows of the read match the target exactly here (the pair is
  // reported through the first of them only).  slot = the read's slot in its wave-tile.  Returns the
  // pair's result word (NX_REJECT, or nmiss | NX_DUP | NX_ACC0 | window << 20 | slot << 24).
  // a database with X: the X of the target span an entry's context covers, in the context's coordinates
  // (one bit per base, as xm) -- read from the mask plane for the lanes whose entry is flagged, zero for
  // the others.  A rare path: its loads wait on the spot.
  // xi = 0 for an entry without the flag, else 0x100 | the top byte of its gene word: the context position
  // of the entry's one X (set here, in registers), or CTX_XMANY
  // and clast = the entry's last context word, whose top byte lists a second X (CTX_XNONE: there is none)
  auto target_xmask = [&](bool ok, uint32_t xi, uint32_t clast, uint32_t gene, uint32_t jx, uint32_t (&tm)[NW]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NW; j++) tm[j] = 0;
    if constexpr (DBX) {
      if (!__any(ok && xi)) return;
      const uint32_t xp = xi & 0xFFu, xp2 = clast >> 24;
      const bool one = ok && xi && xp != CTX_XMANY, two = one && xp2 != CTX_XNONE;
      const uint32_t bit = one ? 1u << (2u * (xp & 15u)) : 0u, bit2 = two ? 1u << (2u * (xp2 & 15u)) : 0u;
#pragma unroll
      for (int j = 0; j < NW; j++) tm[j] = ((xp >> 4) == (uint32_t)j ? bit : 0u) | ((xp2 >> 4) == (uint32_t)j ? bit2 : 0u);
      const bool want = ok && xi && xp == CTX_XMANY;
      if (__any(want)) {
        if (want) {
          const uint64_t* __restrict__ so = mp->seq_off;
          const uint32_t* __restrict__ dm = mp->dbm2;
          const long long bo = 2 * ((long long)(so[gene] + jx) - (long long)CL);
#pragma unroll
          for (int i = 0; i < (NW + 1) / 2; i++) {
            const uint64_t v = ext64s(dm, bo + 64 * i);
            tm[2 * i] = (uint32_t)v & 0x55555555u;
            if (2 * i + 1 < NW) tm[2 * i + 1] = (uint32_t)(v >> 32) & 0x55555555u;
          }
        }
      }
    }
  };
  // the mismatch bits of one context word: x = image ^ context, xm / tm = the X of the read / of the target
  // (X == X is a match, X against a base a mismatch), lm = the bases that take part
  auto diff_word = [&](uint32_t x, uint32_t xmj, uint32_t tmj, uint32_t lmj) __attribute__((always_inline)) -> uint32_t {
    if constexpr (DBX) return ((((x | (x >> 1)) & ~(xmj & tmj)) | (xmj ^ tmj))) & lmj;
    else if constexpr (RX) return ((x | (x >> 1)) | xmj) & lmj;
    else if constexpr (SPEC) return (x | (x >> 1)) & lmj;
    else return base_diff(x, lmj);
  };
  auto score = [&](bool live, int k, int q1, const uint32_t (&img)[NW], const WinTab& tb, const uint32_t (&xm)[NW], uint32_t gene, uint32_t jx,
                   uint32_t xi, const uint32_t (&c)[NW], int len, uint32_t budget, uint32_t valid, uint32_t slot) __attribute__((always_inline)) -> uint32_t {
    // a placement must start inside the target (p = jx - q1 >= 0) and end inside it; the pos-0 rules
    // (ctx_fit) are evaluated only when some lane of the wave is at p == 0 or at target position 0
    uint32_t z = 0;
    bool ok;
    const uint32_t rem = DBX ? ctx_rem(c[NW - 1], xi != 0) : c[NW - 1] >> 16;
    if (__any(live && (jx == (uint32_t)q1 || jx == 0u))) ok = live & ctx_fit(jx, rem, q1, ww, len, &z);
    else ok = live & (jx >= (uint32_t)q1) & (len - q1 <= (int)rem);
    ncmp += ok ? 1u : 0u;
    uint32_t w = NX_REJECT;
    uint32_t tm[NW];
    target_xmask(ok, xi, c[NW - 1], gene, jx, tm);
    if (ok) {
      uint32_t d[NW], nx = 0;
#pragma unroll
      for (int j = 0; j < NW; j++) {
        // (lm has no bit in the high half of the last word, where the context keeps the distance to the target end)
        // SPEC: a word no read of this record stride can reach through window k -- it lies below the read's first bit
        // 2 (CL - q1), or at and beyond 2 (CL - q1) + 2 x 16 (RW - 1) -- is left out AT COMPILE TIME.  (r03 / early r04
        // tested the run-time mask word instead: `tb.lm[j] != 0` is a per-lane value once a ragged tile's masks share
        // the variable, and every word became an exec-masked region of its own -- seven s_and_saveexec / branch /
        // s_or pairs per entry.)
        // (k and j are constants once the window lambda is instantiated and the loop unrolled)
        const int lo_k = SPEC ? 2 * (S_CL - S_WIN[k >= 0 && k < CTX_MAX_W ? k : 0]) : 0;
        const bool reached = !SPEC || (32 * (j + 1) > lo_k && 32 * j < lo_k + 32 * (RW - 1));
        d[j] = reached ? diff_word(img[j] ^ c[j], xm[j], tm[j], tb.lm[j]) : 0u;
        if (reached) nx = bcnt_add(d[j], nx);
      }
      uint32_t exact = valid & (z ? ~q1zero : 0xFFFFFFFFu);
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        if (kk > k || !((tb.need >> kk) & 1u)) continue;  // wave-uniform
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < NW; j++) {
          if constexpr (SPEC) acc |= d[j] & tb.wm[kk][j];  // (constants: the words a window does not touch fold away)
          else acc = and_or_s(d[j], tb.wm[kk][j], acc);
        }
        if (acc) exact &= ~(1u << kk);
      }
      // the reference's confirm for window k accepts the pair (it counts towards that window-key
      // block's MaxMatches); the tuple is reported here only if k is the first window that accepts it
      if (nx <= budget && ((exact >> k) & 1u)) {
        const bool first = (uint32_t)(__ffs(exact) - 1) == (uint32_t)k;
        w = (first ? nx : (nx | NX_DUP)) | NX_ACC0 | ((uint32_t)k << 20) | (slot << 24);
      }
    }
    return w;
  };

  // ---- state of the wave-tile in hand ("cur": its records in s_rec[par], its meta / xpos words in
  // s_meta[par] / s_xp[par], its candidates in s_list[par]), of the next one ("nx": records fetched
  // into registers, then -- phase A -- into s_rec[par ^ 1]) and of the one before ("prev": its
  // overflow entries ride in registers, a lane per entry with the read's record; their comparison
  // pass and prev's phase D run inside cur's second window)
  uint4 va[4], vb[4];  // the ring: the four steps of ONE window
#pragma unroll
  for (int s = 0; s < 4; s++) va[s] = vb[s] = make_uint4(0, 0, 0, 0);
  uint32_t par = 0;
  Rec<RW> rec_nx;  // the records of the NEXT wave-tile, from their fetch to its phase A
  rec_nx.zero();
  uint32_t meta_cur = 0, xw_cur = 0, bb_cur[W];
#pragma unroll
  for (int k = 0; k < W; k++) bb_cur[k] = WB_NONE;
  // prev
  uint32_t wt_prev = 0, nlist_prev = 0, best_prev = 0xFFFFFFFFu, ulen_prev = 0xFFFFFFFFu, total_prev = 0;
  // A lane's own first OWNK reported candidates stay in its registers (result word, gene, position) from the comparison
  // to phase D one wave-tile later: nine tenths of a wave-tile's candidates never see the LDS list (r04: report_own
  // without ballot / mbcnt / ds_write, phase D without list reads and counting atomics for them).  Instances that have no registers to spare (three and four windows, X on either side, wide buckets) keep
  // every candidate on the list.
  constexpr int OWNK = (W <= 2 && XM == 0 && !WIDE) ? 2 : 0;
  uint32_t kc_prev = 0, k0w_prev = NX_REJECT, k0g_prev = 0, k0p_prev = 0, k1w_prev = NX_REJECT, k1g_prev = 0, k1p_prev = 0;
  uint32_t wc_prev[W], bb_prev[W], oc_prev[W], ovf_prev[W];
#pragma unroll
  for (int k = 0; k < W; k++) wc_prev[k] = oc_prev[k] = ovf_prev[k] = 0, bb_prev[k] = WB_NONE;
  // the first 64 overflow entries of prev, a lane per entry: which probe (window, read slot), the read's
  // meta / xpos words and record (taken along when the entry was listed), and the entry itself as
  // its loads deliver it (gene, jx, ctx[0..1] | ctx[2..5] | ... | the last two or three words): the loads' destination
  // registers are carried to the pass as they are -- any re-packing would be a copy that waits for
  // the data
  uint32_t o_n = 0, o_k = 0, o_seg = 0, o_meta = 0, o_xw = 0;
  Rec<RW> o_rec;
  o_rec.zero();
  u32x4_u o_q[NQ];
  uint32_t o_t[NT];
#pragma unroll
  for (int q = 0; q < NQ; q++) o_q[q] = u32x4_u{0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < NT; q++) o_t[q] = 0;

  // phase A of a wave-tile plus what its lanes keep of it: the meta word (length | budget << 17 | valid
  // windows << 24); the record itself goes to s_rec[p]
  // the record of read slot `seg` of the wave-tile whose records are in s_rec[p]
  auto rec_of = [&](uint32_t p, uint32_t seg, Rec<RW>& rec) __attribute__((always_inline)) {
    const uint32_t wid = opaque(threadIdx.x) >> 6;
    const uint4* src = reinterpret_cast<const uint4*>(&s_rec[p][wid][seg * RW]);
#pragma unroll
    for (int q = 0; q < RW / 4; q++) {
      const uint4 v = src[q];
      rec.w[4 * q] = v.x; rec.w[4 * q + 1] = v.y; rec.w[4 * q + 2] = v.z; rec.w[4 * q + 3] = v.w;
    }
  };

  auto phase_a_all = [&](uint32_t wt, uint32_t p, const Rec<RW>& rec, uint32_t& meta, uint32_t& xw, uint32_t (&bb)[W]) __attribute__((always_inline)) {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    const uint32_t valid = phase_a(wt, rec, xw, bb);
    const uint32_t len = rec.len();
    const uint32_t budget0 = len < CONF_NM ? s_nm[len] : 0u;  // (reads on this path are at most 120 bases)
    meta = len | ((budget0 > 127u ? 127u : budget0) << 17) | (valid << 24);
    uint4* dst = reinterpret_cast<uint4*>(&s_rec[p][wid][lane * RW]);
#pragma unroll
    for (int q = 0; q < RW / 4; q++) dst[q] = make_uint4(rec.w[4 * q], rec.w[4 * q + 1], rec.w[4 * q + 2], rec.w[4 * q + 3]);
  };

  if (gw < nwt) {
    Rec<RW> rec0;
    fetch(gw, rec0);
    phase_a_all(gw, 0, rec0, meta_cur, xw_cur, bb_cur);
    if (gw + nw < nwt) fetch(gw + nw, rec_nx);
    wave_lds_sync();
    issue_window(bb_cur[0], va, vb);
  }
  uint32_t wt = gw;
  bool have_prev = false;
#ifdef MUSC_LANE_PROF
  unsigned long long pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime(), pstart = pt0;
  uint32_t pf_tiles = 0;
#define PF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pf[i] += t_ - pt0; pt0 = t_; }
#elif defined(MUSC_LANE_MARK)
#define PF(i) asm volatile("; MARK " #i);
#else
#define PF(i)
#endif
  while (wt < nwt || have_prev) {
    const bool have_cur = wt < nwt;
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint32_t* const cnt_l = s_cb[wid];
    uint32_t* const base_l = s_cb[wid] + WT;
    uint4* const line_l = s_line[wid];
    const bool have_next = wt + nw < nwt;

    // a reported candidate of any read of a wave-tile (overflow entries): list / counters of parity p
    auto report_any = [&](uint32_t p, uint32_t& nl, uint32_t w, uint32_t gene, uint32_t pos) __attribute__((always_inline)) {
      const bool acc = w != NX_REJECT;
      if (acc && block_mode) atomicAdd(&s_dyn[(p * NWAVE + wid) * WT * W + ((w >> 20) & 15u) * WT + (w >> 24)], 1u);
      const bool rep = acc && !(w & NX_DUP);
      const unsigned long long vote = __ballot(rep);
      if (vote == 0) return;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      const uint32_t slot = nl + below;
      nl += (uint32_t)__popcll(vote);
      if (!rep) return;
      atomicMin(&s_best[p][wid][w >> 24], w & 0xFFFFu);
      if (slot < WLIST) {
        s_list[p][wid][slot] = make_uint3(w, gene, pos);
      } else if (slot - WLIST < sregion) {
        spill[sregion_w + p * sregion + (slot - WLIST)] = make_uint4(w, gene, pos, 0u);
      }
    };
    // The comparison pass over overflow entries, a lane per entry: entry (ea | eb | ed), the probe it
    // belongs to (window ek, read slot seg) and that read's meta / xpos words and record; the first
    // n_items lanes hold one.  Candidates go to the lists of parity p.
    auto entry_compare = [&](uint32_t p, uint32_t& nl, uint32_t ul, uint32_t n_items, uint32_t ek, uint32_t seg, uint32_t meta,
                             uint32_t xw, const Rec<RW>& rec, const u32x4_u (&eq)[NQ], const uint32_t (&et)[NT]) __attribute__((always_inline)) {
#ifdef MUSC_LANE_DBG
      if (MUSC_LANE_DBG & 32) return;
#endif
      const bool have = lane < n_items;
      if (!__any(have)) return;
      const int len = (int)REC_LEN(meta);
      // the entry's words: gene, jx, ctx[NW]
      const uint32_t xi = DBX && (eq[0].y & CTX_XFLAG) ? 0x100u | (eq[0].x >> 24) : 0u;
      const uint32_t gene = xi ? eq[0].x & 0xFFFFFFu : eq[0].x, jx = DBX ? eq[0].y & ~CTX_XFLAG : eq[0].y;
      uint32_t c[NW];
#pragma unroll
      for (int j = 0; j < NW; j++) {
        const int wi = j + 2;  // word of the entry
        c[j] = wi < 4 * NQ ? (wi % 4 == 0 ? eq[wi / 4].x : wi % 4 == 1 ? eq[wi / 4].y : wi % 4 == 2 ? eq[wi / 4].z : eq[wi / 4].w) : et[wi - 4 * NQ];
      }
      if (W == 2 && !WIDE && direct) {
        // Two windows, a table whose bucket is the key (the usual case): ONE pass with the window per
        // lane -- the read's image for either window, the lane's window's length mask, and the only
        // exactness to establish is window 0's for the entries that came through window 1.
        const bool k1 = have && ek != 0;
        const int q1a = win[0], q1b = win[W - 1];
        const uint32_t sha = 2u * (uint32_t)(CL - q1a), shb = 2u * (uint32_t)(CL - q1b);
        const uint32_t sh = k1 ? shb : sha;
        const int q1 = k1 ? q1b : q1a;
        uint32_t lm[NW];
        if (SPEC && ul == (uint32_t)S_L) {
          // (the geometry's masks are constants: no table rows from scalar memory in front of the pass)
#pragma unroll
          for (int j = 0; j < NW; j++)
            lm[j] = k1 ? 0x55555555u & bit_range_mask(2 * (S_CL - S_WIN[W - 1]) - 32 * j, 2 * (S_CL - S_WIN[W - 1]) + 2 * S_L - 32 * j)
                       : 0x55555555u & bit_range_mask(2 * (S_CL - S_WIN[0]) - 32 * j, 2 * (S_CL - S_WIN[0]) + 2 * S_L - 32 * j);
        } else if (ul != 0xFFFFFFFFu) {
          const uint32_t (*rows)[CTXW_WORDS] = mp->lm[__builtin_amdgcn_readfirstlane((int)ul)];
#pragma unroll
          for (int j = 0; j < NW; j++) lm[j] = k1 ? rows[W - 1][j] : rows[0][j];
        } else {
#pragma unroll
          for (int j = 0; j < NW; j++) lm[j] = 0x55555555u & bit_range_mask((int)sh - 32 * j, (int)sh + 2 * len - 32 * j);
        }
        uint32_t wm0[NW];  // window 0 of the read, in the coordinates of a comparison through window 1
        if constexpr (SPEC) {
          constexpr int sh1c = 2 * (S_CL - S_WIN[W - 1]);
#pragma unroll
          for (int j = 0; j < NW; j++) wm0[j] = bit_range_mask(sh1c + 2 * S_WIN[0] - 32 * j, sh1c + 2 * (S_WIN[0] + S_WW) - 32 * j);
        } else {
          const uint32_t* __restrict__ row = mp->wm[W - 1][0];
#pragma unroll
          for (int j = 0; j < NW; j++) wm0[j] = row[j];
        }
        uint32_t z = 0;
        bool ok;
        const uint32_t rem = DBX ? ctx_rem(c[NW - 1], xi != 0) : c[NW - 1] >> 16;
        if (__any(have && (jx == (uint32_t)q1 || jx == 0u))) ok = have & ctx_fit(jx, rem, q1, ww, len, &z);
        else ok = have & (jx >= (uint32_t)q1) & (len - q1 <= (int)rem);
        ncmp += ok ? 1u : 0u;
        uint32_t w = NX_REJECT;
        uint32_t tm[NW];
        target_xmask(ok, xi, c[NW - 1], gene, jx, tm);
        if (ok) {
          uint32_t ia[NW], ib[NW], xm[NW];
          read_image_n<RW, NW>(rec, sha, ia);
          read_image_n<RW, NW>(rec, shb, ib);
          x_mask(xw, sh, xm);
          uint32_t d[NW], nx = 0, acc0 = 0;
#pragma unroll
          for (int j = 0; j < NW; j++) {
            d[j] = diff_word((k1 ? ib[j] : ia[j]) ^ c[j], xm[j], tm[j], lm[j]);
            nx = bcnt_add(d[j], nx);
            if constexpr (SPEC) acc0 |= d[j] & wm0[j];  // (constants: the words window 0 does not touch fold away)
            else acc0 = and_or_s(d[j], wm0[j], acc0);
          }
          uint32_t exact = REC_VALID(meta) & (z ? ~q1zero : 0xFFFFFFFFu);
          if (k1 && acc0) exact &= ~1u;
          const uint32_t kbit = k1 ? 2u : 1u;
          if (nx <= REC_BUDGET(meta) && (exact & kbit)) {
            const bool first = (exact & (kbit - 1u)) == 0;
            w = (first ? nx : (nx | NX_DUP)) | NX_ACC0 | (k1 ? 1u << 20 : 0u) | (seg << 24);
          }
        }
        report_any(p, nl, w, gene, jx - (uint32_t)q1);
        return;
      }
      uint32_t w = NX_REJECT;
      int q1 = 0;
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        const bool mine = have && ek == (uint32_t)kk;
        if (!__any(mine)) continue;
        const int q1k = win[kk];
        const uint32_t shk = 2u * (uint32_t)(CL - q1k);
        WinTab tb;
        win_tab(ul, kk, shk, (uint32_t)len, tb);
        uint32_t img[NW], xm[NW];
        read_image_n<RW, NW>(rec, shk, img);
        x_mask(xw, shk, xm);
        const uint32_t w2 = score(mine, kk, q1k, img, tb, xm, gene, jx, xi, c, len, REC_BUDGET(meta), REC_VALID(meta), seg);
        if (mine) {
          w = w2;
          q1 = q1k;
        }
      }
      report_any(p, nl, w, gene, jx - (uint32_t)q1);
    };
    // The loads of prev's first 64 overflow entries (their places in E were listed, s_oix, when prev's
    // last window arrived).  Issued right after a wait that left nothing in flight and straight into
    // the registers the pass reads after the NEXT wait: no copy, no wait of their own.  Every lane
    // loads -- the ones past the last item entry 0 of E, which always exists.
    auto entry_fetch = [&]() __attribute__((always_inline)) {
      const uint32_t eix = lane < o_n ? s_oix[wid][lane] : 0u;
      const uint32_t* __restrict__ pe = reinterpret_cast<const uint32_t*>(E) + ctx_entry_word<WIDE>(eix);
#pragma unroll
      for (int q = 0; q < NQ; q++) o_q[q] = *reinterpret_cast<const u32x4_u*>(pe + 4 * q);
#pragma unroll
      for (int q = 0; q < NT; q++) o_t[q] = pe[4 * NQ + q];
    };
    // owner tables of the overflow items [c0, c0 + 64) of a wave-tile: item -> (window, read slot) and its
    // place in E; oc / ovf / pre: per lane (= probe) the number of entries beyond the third, where
    // they start in E, and the items before them
    auto owner_tables = [&](uint32_t c0, const uint32_t (&oc)[W], const uint32_t (&ovf)[W], const uint32_t (&pre)[W]) __attribute__((always_inline)) {
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        const uint32_t e_lo = c0 > pre[kk] ? c0 - pre[kk] : 0u;
        const uint32_t e_hi = pre[kk] + oc[kk] > c0 + WT ? (c0 + WT > pre[kk] ? c0 + WT - pre[kk] : 0u) : oc[kk];
#pragma unroll 1
        for (uint32_t e = e_lo; e < e_hi; e++) {
          s_own[wid][pre[kk] + e - c0] = (uint8_t)(kk * WT + lane);
          s_oix[wid][pre[kk] + e - c0] = ovf[kk] + e;
        }
      }
      wave_lds_sync();
    };
    // prev's overflow entries beyond the first 64 (families of near-identical targets, low-complexity
    // keys): listed, loaded and compared on the spot, 64 at a time; their reads' records come from
    // global memory (prev's place in LDS belongs to the next wave-tile by now)
    auto overflow_rest_prev = [&]() __attribute__((always_inline)) {
      if (total_prev <= (uint32_t)WT) return;
      const uint32_t pp = par ^ 1u;
      uint32_t pre[W], total = 0;
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        const uint32_t inc = wave_scan_incl(oc_prev[kk]);
        pre[kk] = total + inc - oc_prev[kk];
        total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      }
      for (uint32_t c0 = WT; c0 < total; c0 += WT) {
        owner_tables(c0, oc_prev, ovf_prev, pre);
        const uint32_t cnt = total - c0 < (uint32_t)WT ? total - c0 : (uint32_t)WT;
        const bool mine = lane < cnt;
        const uint32_t probe = mine ? (uint32_t)s_own[wid][lane] : 0u;
        const uint32_t eix = mine ? s_oix[wid][lane] : 0u;
        const uint32_t ek = probe >> 6, seg = probe & 63u;
        const uint32_t* __restrict__ pe = reinterpret_cast<const uint32_t*>(E) + ctx_entry_word<WIDE>(eix);
        u32x4_u tq[NQ];
        uint32_t tt[NT];
#pragma unroll
        for (int q = 0; q < NQ; q++) tq[q] = *reinterpret_cast<const u32x4_u*>(pe + 4 * q);
#pragma unroll
        for (int q = 0; q < NT; q++) tt[q] = pe[4 * NQ + q];
        Rec<RW> trec;
        trec.load(rd + (r0 + (uint64_t)wt_prev * WT + seg) * (uint64_t)RW, RW);
        const uint32_t tmeta = s_meta[pp][wid][seg];
        uint32_t txw = 0;
        if constexpr (RX) txw = s_xp[pp][wid][seg];
        wave_lds_sync();  // (the owner tables are rewritten by the next chunk)
        entry_compare(pp, nlist_prev, ulen_prev, cnt, ek, seg, tmeta, txw, trec, tq, tt);
      }
    };
    // phase D for prev: per-read selection and the tuples (the protocol of match_ctx_pass)
    auto phase_d_prev = [&]() __attribute__((always_inline)) {
#ifdef MUSC_LANE_DBG
      if (MUSC_LANE_DBG & 8) return;
#endif
      const uint32_t pp = par ^ 1u;
      uint32_t* const best_l = s_best[pp][wid];
      uint32_t* const wcnt_l = s_dyn + (pp * NWAVE + wid) * WT * W;
      const uint32_t nl = nlist_prev;
      const uint32_t nspill = nl > WLIST ? nl - WLIST : 0u;
      const bool spill_ok = nspill <= sregion;
      if (nspill > maxspill) maxspill = nspill;
      if (nspill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's spilled candidates have landed
      if (block_mode) {
#pragma unroll
        for (int k = 0; k < W; k++) {
          const uint32_t cw = (BB_LDS ? 0u : wc_prev[k]) + wcnt_l[k * WT + lane];
          wcnt_l[k * WT + lane] = 0;
          if (!cw) continue;
          const uint32_t h = block_hash32((uint32_t)k, BB_LDS ? s_bbp[BB_LDS ? wid : 0][BB_LDS ? k : 0][BB_LDS ? lane : 0] : bb_prev[k]);
          if (block_mode == 1) atomicAdd(&s_sketch[h >> (32 - MATCH_SKETCH_BITS)], cw);
          else atomicAdd(&block_table[h >> (32 - BLOCK_TABLE_BITS)], cw);
        }
      }
      // the read's best: the lane's own (every own candidate, listed or not, went into best_prev) joins what the
      // overflow pass reported for it
      const uint32_t b0 = best_l[lane];
      const uint32_t bestr = best_prev < b0 ? best_prev : b0;
      const uint32_t thr_own = apply ? bestr + mmtol : 0xFFFFu;
      const bool s0 = OWNK >= 1 && kc_prev > 0 && (k0w_prev & 0xFFFFu) <= thr_own, s1 = OWNK >= 2 && kc_prev > 1 && (k1w_prev & 0xFFFFu) <= thr_own;
      const uint32_t nown = (s0 ? 1u : 0u) + (s1 ? 1u : 0u);
      uint32_t cnum = nown;
      auto item = [&](uint32_t j, uint32_t* gene, uint32_t* pos) __attribute__((always_inline)) -> uint32_t {
        if (j < WLIST) {
          const uint3 it = s_list[pp][wid][j];
          *gene = it.y;
          *pos = it.z;
          return it.x;
        }
        // written by other lanes of this wave a moment ago: read past the L1
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(spill + sregion_w + pp * sregion + (j - WLIST));
        *gene = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *pos = __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      const uint32_t nuse = spill_ok ? nl : (nl < WLIST ? nl : WLIST);
      // The first round of LISTED candidates (a lane each: the overflow entries' and an own third and later one) stays in
      // registers from the count to the store: the counting atomic also hands out the tuple's place among its read's
      // listed ones; further rounds (heavy tiles, the instances that list everything) are walked twice.
      uint32_t kw = NX_REJECT, kg = 0, kp = 0, ko = 0;
      if (nuse) {  // (wave-uniform)
        best_l[lane] = bestr;
        cnt_l[lane] = 0;
        wave_lds_sync();
        if (lane < nuse) {
          const uint32_t w = item(lane, &kg, &kp);
          const uint32_t rl = w >> 24;
          const uint32_t thr = apply ? best_l[rl] + mmtol : 0xFFFFu;
          if ((w & 0xFFFFu) <= thr) {
            kw = w;
            ko = atomicAdd(&cnt_l[rl], 1u);
          }
        }
        for (uint32_t j = 64 + lane; j < nuse; j += 64) {
          uint32_t g, p;
          const uint32_t w = item(j, &g, &p);
          const uint32_t rl = w >> 24;
          const uint32_t thr = apply ? best_l[rl] + mmtol : 0xFFFFu;
          if ((w & 0xFFFFu) <= thr) atomicAdd(&cnt_l[rl], 1u);
        }
        wave_lds_sync();
        cnum += cnt_l[lane];
      }
      const uint32_t inc = wave_scan_incl(cnum);
      const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
      const uint32_t mybase = inc - cnum;
      const uint64_t base = region0 + used;
      const bool fits = spill_ok && used + total <= region;
      if (lane == 0) {
        tbase[wt_prev] = (uint32_t)base;
        tcount2[wt_prev] = fits ? total : 0u;
      }
      if (fits && total) {
        const uint32_t rid = (uint32_t)(r0 + wt_prev * WT + lane);
        if (s0) put_tuple(&stage[base + mybase], rid, k0g_prev, k0p_prev, k0w_prev & 0xFFFFu);
        if (s1) put_tuple(&stage[base + mybase + (s0 ? 1u : 0u)], rid, k1g_prev, k1p_prev, k1w_prev & 0xFFFFu);
        if (nuse) {
          base_l[lane] = mybase + nown;  // where the read's listed tuples go: behind its own
          wave_lds_sync();
          if (kw != NX_REJECT) {
            const uint32_t rl = kw >> 24;
            put_tuple(&stage[base + base_l[rl] + ko], (uint32_t)(r0 + wt_prev * WT + rl), kg, kp, kw & 0xFFFFu);
          }
          if (nuse > 64) {
            for (uint32_t j = 64 + lane; j < nuse; j += 64) {
              uint32_t g, p;
              const uint32_t w = item(j, &g, &p);
              const uint32_t rl = w >> 24, v = w & 0xFFFFu;
              const uint32_t thr = apply ? best_l[rl] + mmtol : 0xFFFFu;
              if (v > thr) continue;
              // places of the later rounds: behind the first round's listed tuples of the read, counted down from its total
              const uint32_t ord = atomicAdd(&cnt_l[rl], 0xFFFFFFFFu) - 1u;
              put_tuple(&stage[base + base_l[rl] + ord], (uint32_t)(r0 + wt_prev * WT + rl), g, p, v);
            }
          }
        }
      }
      wave_lds_sync();
      best_l[lane] = 0xFFFFFFFFu;  // for the wave-tile after next
      used += total;
      nrep += kc_prev < (uint32_t)OWNK ? kc_prev : (uint32_t)OWNK;  // (the listed ones from one lane)
      nrep += lane == 0 ? nl : 0u;
    };
    // everything that is left of prev: the pass over its first 64 overflow entries (fetched a window
    // ago, or -- fresh -- right before this call), the rest of them, phase D
    auto finish_prev = [&]() __attribute__((always_inline)) {
      entry_compare(par ^ 1u, nlist_prev, ulen_prev, o_n, o_k, o_seg, o_meta, o_xw, o_rec, o_q, o_t);
      PF(6)
      overflow_rest_prev();
      wave_lds_sync();
      phase_d_prev();
      PF(7)
    };

    uint4 cpv = make_uint4(0, 0, 0, 0);
    uint32_t cpm = 0, cpd = 0;
    uint32_t nlist = 0;           // reported candidates of cur so far (wave-uniform)
    uint32_t kc = 0, k0w = NX_REJECT, k0g = 0, k0p = 0, k1w = NX_REJECT, k1g = 0, k1p = 0;  // this lane's own first candidates of cur
    uint32_t best = 0xFFFFFFFFu;  // smallest mismatch count reported for this lane's read by the in-lane comparisons
    uint32_t ulen = 0xFFFFFFFFu, total_cur = 0;
    uint32_t wc[W], oc[W], ovf[W];  // accepted pairs of (window, this lane's read); entries beyond the third of this lane's probe of window k, and where in E
    uint32_t meta_nx = 0, xw_nx = 0, bb_nx[W];
#pragma unroll
    for (int k = 0; k < W; k++) wc[k] = oc[k] = ovf[k] = 0, bb_nx[k] = WB_NONE;

    if (have_cur) {
      pcopy_begin(wt, cpv, cpm, cpd);
      const bool active = wt * WT + lane < n;
      // this wave-tile's per-read state: registers of the read's lane; the meta word and the xpos word
      // also go to LDS, where the overflow entries pick up their reads'
      const int rlen = (int)REC_LEN(meta_cur);
      const uint32_t budget = REC_BUDGET(meta_cur), valid_cur = REC_VALID(meta_cur);
      s_meta[par][wid][lane] = meta_cur;
      if constexpr (RX) s_xp[par][wid][lane] = xw_cur;
      // every read of the wave-tile of one length: the comparisons use scalar length masks
      const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(rlen);
      ulen = __ballot(active && (uint32_t)rlen != len0) == 0 ? len0 : 0xFFFFFFFFu;

      uint3* const list_cur = s_list[par][wid];  // (formed once per wave-tile: its address arithmetic is two multiplies)
      // a reported candidate of the lane's own read (in-lane comparisons): the first OWNK stay in the lane's registers
      // (k0 / k1), later ones are appended to the list in lane order
      auto report_own = [&](uint32_t w, uint32_t gene, uint32_t pos, uint32_t& wck) __attribute__((always_inline)) {
        const bool acc = w != NX_REJECT;
        wck += acc ? 1u : 0u;
        const bool rep = acc && !(w & NX_DUP);
        const bool first = OWNK >= 1 && rep && kc == 0, second = OWNK >= 2 && rep && kc == 1, later = rep && kc >= (uint32_t)OWNK;
        k0w = first ? w : k0w; k0g = first ? gene : k0g; k0p = first ? pos : k0p;
        k1w = second ? w : k1w; k1g = second ? gene : k1g; k1p = second ? pos : k1p;
        kc += rep ? 1u : 0u;
        const uint32_t v = w & 0xFFFFu;
        best = rep && v < best ? v : best;
        const unsigned long long vote = __ballot(later);
        if (vote == 0) return;
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
        const uint32_t slot = nlist + below;
        nlist += (uint32_t)__popcll(vote);
        if (!later) return;
        if (slot < WLIST) {
          list_cur[slot] = make_uint3(w, gene, pos);
        } else if (slot - WLIST < sregion) {
          spill[sregion_w + par * sregion + (slot - WLIST)] = make_uint4(w, gene, pos, 0u);
        }
      };

      // (a generic lambda over the window number: the window loop must be unrolled for the per-window
      // registers to stay registers, and past a certain size the unroller gives up on a plain loop)
      auto window = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const int q1 = win[k];
        const uint32_t sh = 2u * (uint32_t)(CL - q1);
        WinTab tb;  // (its scalar loads are in flight while the wave waits for the window's lines)
        win_tab(ulen, k, sh, (uint32_t)rlen, tb);
        // ---- arrival: every load's 1 KB goes into the line buffer as it came (the swizzle is in the load's choice of chunk) ...
        {
#pragma unroll
          for (int i = 0; i < 8; i++) line_l[i * 64 + lane] = i < 4 ? va[i] : vb[i - 4];
        }
        PF(0)
        // ---- ... and with nothing left in flight: the loads of prev's overflow entries (used after the
        // next wait; with one window per wave-tile, at once), then the ring is refilled with the next
        // window (this wave-tile's, or the next one's first; the records of the wave-tile after that
        // follow the last refill).  Whatever is consumed from here to the next arrival was loaded
        // before the wait that has just ended.
        if (k == 0 && have_prev) entry_fetch();
        if (W == 1) {
          if (have_prev) finish_prev();
          if (have_next) phase_a_all(wt + nw, par ^ 1u, rec_nx, meta_nx, xw_nx, bb_nx);
          wave_lds_sync();
        }
        if (k + 1 < W) {
          issue_window(bb_cur[k + 1 < W ? k + 1 : 0], va, vb);
        } else {
          if (have_next) issue_window(bb_nx[0], va, vb);
          if (wt + 2 * nw < nwt) fetch(wt + 2 * nw, rec_nx);
        }
        PF(1)
        wave_lds_sync();
        // ---- lane p takes line p: its header now, an entry's context when its turn comes (the line
        // buffer stays as it is until the next arrival)
        const uint32_t rb = lane * 8u + (((lane >> 1) & 7u) ^ (lane & 1u));
        const uint4 h0 = line_l[rb], h1 = line_l[rb ^ 1u];
        const uint32_t cnt = bb_cur[k] != WB_NONE ? h0.x : 0u;  // (a probe that takes no part fetched bucket 0)
        ncand += cnt;
        oc[k] = cnt > (uint32_t)NIN ? cnt - (uint32_t)NIN : 0u;
        ovf[k] = h0.y;
        novf += oc[k];
        PF(2)
        // the next wave-tile's records arrived with this wave-tile's first lines: its phase A
        if (W > 1 && k == 0 && have_next) phase_a_all(wt + nw, par ^ 1u, rec_nx, meta_nx, xw_nx, bb_nx);
        PF(8)
        // prev's overflow entries arrived with this window's lines: their pass, the rest of prev
        if (W > 1 && k == 1 && have_prev) finish_prev();
        if (k + 1 == W) {
          // ---- this wave-tile's overflow entries: the first 64 are listed (s_oix: where in E) and their
          // lanes take the reads' meta / xpos words and records along; the loads follow when the
          // next wave-tile's first window has arrived
          uint32_t pre[W], total = 0;
#pragma unroll
          for (int kk = 0; kk < W; kk++) {
            const uint32_t inc = wave_scan_incl(oc[kk]);
            pre[kk] = total + inc - oc[kk];
            total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
          }
#ifdef MUSC_LANE_DBG
          if (MUSC_LANE_DBG & 4) total = 0;
#endif
          total_cur = total;
          owner_tables(0, oc, ovf, pre);
          o_n = total < (uint32_t)WT ? total : (uint32_t)WT;
          const uint32_t probe = lane < o_n ? (uint32_t)s_own[wid][lane] : 0u;
          o_k = probe >> 6;
          o_seg = probe & 63u;
          o_meta = s_meta[par][wid][o_seg];
          if constexpr (RX) o_xw = s_xp[par][wid][o_seg];
          rec_of(par, o_seg, o_rec);
        }
        wave_lds_sync();
        PF(3)
        // ---- the read's image for this window, then the line's three entries, in this lane
        uint32_t img[NW], xm[NW];
        {
          Rec<RW> rec;
          rec_of(par, lane, rec);
          read_image_n<RW, NW>(rec, sh, img);
        }
        x_mask(xw_cur, sh, xm);
        if constexpr (!WIDE) {
          uint4 ca = line_l[rb ^ 2u], cb = line_l[rb ^ 3u];
          PF(9)
#pragma unroll
          for (int s = 0; s < CTX_INLINE; s++) {
#ifdef MUSC_LANE_DBG
            const bool live = (uint32_t)s < cnt && !(MUSC_LANE_DBG & 1) && s < MUSC_LANE_DBG_SLOTS;  // (timing experiments: wrong tuples)
#else
            const bool live = (uint32_t)s < cnt;
#endif
            if (!__any(live)) break;
            uint4 na = ca, nb = cb;
            if (s + 1 < CTX_INLINE) {  // the next entry's context is on its way while this one is compared
              na = line_l[rb ^ (uint32_t)(2 * s + 4)];
              nb = line_l[rb ^ (uint32_t)(2 * s + 5)];
            }
            const uint32_t gener = s == 0 ? h0.z : (s == 1 ? h0.w : h1.x);
            const uint32_t jxr = s == 0 ? h1.y : (s == 1 ? h1.z : h1.w);
            const uint32_t xi = DBX && (jxr & CTX_XFLAG) ? 0x100u | (gener >> 24) : 0u;
            const uint32_t gene = xi ? gener & 0xFFFFFFu : gener, jx = DBX ? jxr & ~CTX_XFLAG : jxr;
            const uint32_t c[NW] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
            const uint32_t w = score(live, k, q1, img, tb, xm, gene, jx, xi, c, rlen, budget, valid_cur, lane);
            report_own(w, gene, jx - (uint32_t)q1, wc[k]);
            ca = na;
            cb = nb;
          }
        } else {
          // two entries of fifteen words behind the two header words: entry 0 = words 2..16 (h0.zw,
          // chunks 1-3, chunk 4's first word), entry 1 = words 17..31 (the rest of chunk 4, chunks 5-7)
          const uint4 m4 = line_l[rb ^ 4u];
          PF(9)
          {
            const bool live = cnt > 0u;
            if (__any(live)) {
              const uint4 c2 = line_l[rb ^ 2u], c3 = line_l[rb ^ 3u];
              const uint32_t c[NW] = {h1.x, h1.y, h1.z, h1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w, m4.x};
              const uint32_t xi = DBX && (h0.w & CTX_XFLAG) ? 0x100u | (h0.z >> 24) : 0u;
              const uint32_t gene = xi ? h0.z & 0xFFFFFFu : h0.z, jx = DBX ? h0.w & ~CTX_XFLAG : h0.w;
              const uint32_t w = score(live, k, q1, img, tb, xm, gene, jx, xi, c, rlen, budget, valid_cur, lane);
              report_own(w, gene, jx - (uint32_t)q1, wc[k]);
            }
          }
          {
            const bool live = cnt > 1u;
            if (__any(live)) {
              const uint4 c5 = line_l[rb ^ 5u], c6 = line_l[rb ^ 6u], c7 = line_l[rb ^ 7u];
              const uint32_t c[NW] = {m4.w, c5.x, c5.y, c5.z, c5.w, c6.x, c6.y, c6.z, c6.w, c7.x, c7.y, c7.z, c7.w};
              const uint32_t xi = DBX && (m4.z & CTX_XFLAG) ? 0x100u | (m4.y >> 24) : 0u;
              const uint32_t gene = xi ? m4.y & 0xFFFFFFu : m4.y, jx = DBX ? m4.z & ~CTX_XFLAG : m4.z;
              const uint32_t w = score(live, k, q1, img, tb, xm, gene, jx, xi, c, rlen, budget, valid_cur, lane);
              report_own(w, gene, jx - (uint32_t)q1, wc[k]);
            }
          }
        }
        PF(4)
      };
      window(std::integral_constant<int, 0>{});
      if constexpr (W > 1) window(std::integral_constant<int, 1>{});
      if constexpr (W > 2) window(std::integral_constant<int, 2>{});
      if constexpr (W > 3) window(std::integral_constant<int, 3>{});
#ifdef MUSC_LANE_PROF
      pf_tiles++;
#endif
    } else {
      // past this wave's last wave-tile: prev's overflow entries and its phase D are all that is left
      entry_fetch();
      finish_prev();
    }
    wave_lds_sync();
    if (have_cur) pcopy_end(cpv, cpm, cpd);
    // ---- cur becomes prev
    have_prev = have_cur;
    if (have_cur) {
      wt_prev = wt;
      nlist_prev = nlist;
      best_prev = best;
      kc_prev = kc; k0w_prev = k0w; k0g_prev = k0g; k0p_prev = k0p; k1w_prev = k1w; k1g_prev = k1g; k1p_prev = k1p;
      ulen_prev = ulen;
      total_prev = total_cur;
#pragma unroll
      for (int k = 0; k < W; k++) {
        if constexpr (BB_LDS) {
          // (and the lane's own accepted-pair counts go straight into the wave-tile's counter slots, which the overflow
          // pass -- one iteration later -- adds to: nothing else touches them until then)
          const uint32_t ln = opaque(threadIdx.x) & 63;
          s_bbp[wid][k][ln] = bb_cur[k];
          if (block_mode) s_dyn[(par * NWAVE + wid) * WT * W + k * WT + ln] += wc[k];
        } else {
          bb_prev[k] = bb_cur[k];
          wc_prev[k] = wc[k];
        }
        bb_cur[k] = bb_nx[k], oc_prev[k] = oc[k], ovf_prev[k] = ovf[k];
      }
      meta_cur = meta_nx;
      xw_cur = xw_nx;
      par ^= 1u;
      wt += nw;
    }
    wave_lds_sync();  // the next wave-tile rewrites cnt / base
    PF(5)
  }
#ifdef MUSC_LANE_PROF
  // (the spread of the waves' finishing times: what a static assignment of wave-tiles leaves on the table)
  if ((gw % 37) == 0 && (threadIdx.x & 63) == 0 && pf_tiles > 4) printf("fin %u %u %llu %llu\n", gw, pf_tiles, pstart, (unsigned long long)__builtin_amdgcn_s_memtime());
  if ((gw == 0 || gw == 1001) && (threadIdx.x & 63) == 0 && pf_tiles > 4)
    printf("wave %u: %u tiles, cycles/tile: total %llu | arrive(wait+lds write) %llu issue %llu hdr %llu phA %llu finish_prev(epass %llu rest+phD %llu) expand %llu img %llu slots %llu end %llu\n", gw, pf_tiles,
           (__builtin_amdgcn_s_memtime() - pstart) / pf_tiles, pf[0] / pf_tiles, pf[1] / pf_tiles, pf[2] / pf_tiles, pf[8] / pf_tiles,
           pf[6] / pf_tiles, pf[7] / pf_tiles, pf[3] / pf_tiles, pf[9] / pf_tiles, pf[4] / pf_tiles, pf[5] / pf_tiles);
#endif
  if (pcopy) {
    for (; wt < pnwt; wt += nw) {
      uint4 cpv = make_uint4(0, 0, 0, 0);
      uint32_t cpm, cpd;
      pcopy_begin(wt, cpv, cpm, cpd);
      pcopy_end(cpv, cpm, cpd);
    }
  }
  // one reduction per workgroup and a handful of atomics from its first thread (as in k_match)
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
