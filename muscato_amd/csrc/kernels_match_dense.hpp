// k_match_d -- the fused screen + confirm + select kernel on context buckets with DENSE comparison
// passes (device code, included by muscato_hip.hip after kernels_match.hpp whose bucket layout,
// parameter block, fit rules and cdiff it shares).  It runs where its LDS budget allows three
// workgroups per CU (at most two windows, records of at most eight words: cfg2, cfg3, cfg4);
// k_match covers the rest.
//
// k_match compares where a bucket line arrives: a quad of lanes per probe, the line's three entries
// in lanes 1..3.  On cfg3 a probe holds 1.7 entries, so 42 % of the lanes of a comparison step
// carry an entry, and each of the eight steps per wave-tile pays a step's fixed cost (fit rules,
// exactness masks, the report vote: ~100 instructions); the overflow entries get passes of their
// own at 36 % -- 3230 instructions per wave-tile (rocprofv3 SQ_INSTS_*), and the SIMDs issue about
// one instruction per four cycles whatever the mix.  Here the arrival step only MOVES the entries
// that exist -- context words, target number, window position, read slot: 44 bytes -- onto a
// per-wave stack in LDS (a vote + three LDS writes), and a comparison pass pops 64 entries, one per
// lane, whenever 64 are waiting: four or five passes per wave-tile instead of eight steps + the
// overflow passes.  The entries beyond a bucket's third (CtxEntry in E) go onto the same stack.
// 2640 instructions per wave-tile.
//
// The read side of a comparison is not a pre-shifted image per (read, window) in LDS (16 KB per
// workgroup) but the read's record itself (8 KB): a pass works on one window, so the shift is
// wave-uniform and the image words are nine LDS words funnel-shifted in registers.  What the image
// holds outside the read's own bits does not matter -- the length mask of cdiff removes it -- so
// the neighbouring records' words that the shifted window runs into are harmless.
//
// The ring of bucket loads never runs dry: the buckets of the NEXT wave-tile are computed before the
// current tile's last window (phase a1), whose arrival steps refill the ring with the next tile's
// first steps; the rest of phase A (record -> LDS, meta word: a2) follows the last pass.
//
// LDS per workgroup: 52 160 B with the MaxMatches sketch (the stack is 5.25 KB per wave) = three
// workgroups per CU; 2 KB more and only two fit (LDS is handed out in pieces of more than 1 KB:
// match_resident in muscato_hip.hip), which costs 45 %.  Fewer instructions but three waves per
// SIMD instead of k_match's four: 1.16 ms per launch on cfg3 against 1.21, cfg2 0.124 against 0.142.
#pragma once

#ifndef MATCHD_QTHR
#define MATCHD_QTHR 64   // a pass runs as soon as this many entries wait
#endif
#define MATCH_QCAP (MATCHD_QTHR - 1 + 48 + 1)   // entries the stack holds: QTHR - 1 waiting + the 48 one step can add
#ifndef MATCHD_WLIST
#define MATCHD_WLIST MATCH_WLIST
#endif
#ifndef MATCHD_WAVES
#define MATCHD_WAVES 3
#endif
#define MATCH_QW 12      // words per stack entry (48 bytes: three aligned b128 accesses)
#define MATCH_DOWN 48    // overflow items moved per step

// Reads with X on context buckets (RX): with an X-free database an X in a read is a mismatch wherever
// the read is placed, and a window holding one never finds its key in the index.  What the kernel
// needs of a read's X is where they are: xpos = up to four positions (7 bits each, bits 0-27) and
// their number (bits 28-31, saturating at 15).  A read with more than four X takes part only if
// that many mismatches exceed its budget anyway -- then it has no tuples at all (k_xpos_check
// makes the run take the two-kernel path otherwise).
#define XPOS_MAX 4
#define XPOS_CNT(w) ((w) >> 28)
#define XPOS_AT(w, q) (((w) >> (7 * (q))) & 127u)

MUSC_KERNEL void k_read_xpos(const uint32_t* __restrict__ rd, const uint32_t* __restrict__ rdm, uint64_t nreads, int rw,
                            uint32_t* __restrict__ xpos) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  uint32_t w = 0;
  if (rd[(i + 1) * rw - 1] & READ_HAS_X) {
    uint32_t cnt = 0;
    for (int j = 0; j < rw - 1; j++) {
      uint32_t m = rdm[i * rw + j] & 0x55555555u;
      while (m) {
        const uint32_t b = (uint32_t)__ffs(m) - 1u;
        m &= m - 1u;
        const uint32_t p = 16u * (uint32_t)j + (b >> 1);
        if (cnt < XPOS_MAX && p < 128u) w |= p << (7u * cnt);
        cnt++;
      }
    }
    w |= (cnt > 15u ? 15u : cnt) << 28;
  }
  xpos[i] = w;
}

// *bad = 1 if some read holds more than XPOS_MAX X and that many mismatches are within its budget
MUSC_KERNEL void k_xpos_check(const uint32_t* __restrict__ rd, const uint32_t* __restrict__ xpos, uint64_t nreads, int rw,
                             const uint16_t* __restrict__ nmiss_tab, uint32_t max_len, uint32_t* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nreads) return;
  const uint32_t cnt = XPOS_CNT(xpos[i]);
  if (cnt <= XPOS_MAX) return;
  const uint32_t len = rd[(i + 1) * rw - 1] & 0xFFFFu;
  const uint32_t budget = len <= max_len ? nmiss_tab[len] : 0xFFFFu;
  if (cnt == 15u || cnt <= budget) atomicOr(bad, 1u);
}

template <int RW, bool W2, bool RX>
__global__ __launch_bounds__(TILE, MATCHD_WAVES) void k_match_d(const uint32_t* __restrict__ rd, uint64_t r0, uint32_t n,
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
  constexpr int WMAX = W2 ? 2 : CTX_MAX_W;
  constexpr int NWAVE = TILE / 64;
  constexpr int RPAD = 8;  // words in front of and behind a wave's records that a shifted window may touch
  extern __shared__ uint32_t s_dyn[];  // block_mode != 0: NWAVE x W x 64 per-(window, read) counters, then (mode 1) the sketch
  __shared__ uint32_t s_rec[NWAVE][WT * RW + 2 * RPAD];  // the wave-tile's records
  __shared__ uint32_t s_meta[NWAVE][WT];                  // length | budget << 17 | valid windows << 24
  __shared__ uint32_t s_bb[2][NWAVE][WMAX * WT];          // bucket of (window, read), WB_NONE when the window takes no part; this tile's and the next one's
  __shared__ uint32_t s_oc[NWAVE][WT];                    // current window: overflow entries of the probe
  __shared__ uint32_t s_ovf[NWAVE][WT];                   // current window: where in E
  __shared__ uint32_t s_best[NWAVE][WT];                  // smallest mismatch count reported per read
  __shared__ uint3 s_list[NWAVE][MATCHD_WLIST];            // reported candidates: result word, gene, position
  __shared__ __attribute__((aligned(16))) uint32_t s_q[NWAVE][MATCH_QCAP * MATCH_QW];  // the stack of entries waiting for a pass; phase D: cnt[64], base[64]
  __shared__ uint32_t s_oix[NWAVE][MATCH_DOWN];           // overflow step: flat item -> index within its bucket's overflow list
  __shared__ uint8_t s_own[NWAVE][MATCH_DOWN];            //                flat item -> read slot
  __shared__ uint16_t s_nm[CONF_NM];
  __shared__ uint32_t s_xp[RX ? NWAVE : 1][RX ? WT : 1];  // RX: the reads' xpos words (1 KB: what the 2 KB LDS granule leaves free)

  const int W = mp->W, ww = mp->ww, CL = mp->CL;
  const int win0 = mp->win[0], win1 = mp->win[1];
  const uint32_t q1zero = mp->q1zero_mask;
  const int dbg = mp->dbg;
  uint32_t* const s_sketch = s_dyn + NWAVE * WT * W;
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)mp->max_len ? nmiss_tab[t] : (uint16_t)0;
  if (block_mode == 1)
    for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) s_sketch[t] = 0;
  for (uint32_t t = threadIdx.x; t < NWAVE * 2 * RPAD; t += TILE) {  // the pads are read, never written again
    const uint32_t wv = t / (2 * RPAD), o = t % (2 * RPAD);
    s_rec[wv][o < RPAD ? o : WT * RW + o] = 0;
  }
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

  // Phase A of a wave-tile comes in two parts, so that the bucket lines of the NEXT wave-tile can be
  // requested while the current one is still being compared (the ring of loads never runs dry):
  //   a1  a lane per read: the record -> registers; per window the length gate + CountDinuc >=
  //       MinDinuc (cmd/muscato_window_reads/main.go:106-118 == cmd/muscato_screen/main.go:174-185)
  //       and the bucket of the window key -> s_bb[par].  Runs before the current tile's last
  //       window, whose arrival steps refill the ring with the next tile's first steps.
  //   a2  after the current tile's last comparison pass: the record -> LDS and the meta word.
  auto fetch = [&](uint32_t wt, Rec<RW>& rec) {
    const uint32_t i = wt * WT + (opaque(threadIdx.x) & 63);
    rec.load(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW, RW);
  };
  auto phase_a1 = [&](uint32_t wt, uint32_t par, const Rec<RW>& rec, uint32_t& xw) -> uint32_t {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint32_t* const bb_l = s_bb[par][wid];
    const bool active = wt * WT + lane < n;
    const int len = (int)rec.len();
    uint32_t valid = 0;
    // RX: windows that hold an X never probe; a read with more X than fit the word has no tuples
    xw = 0;
    uint32_t xwin = 0;  // windows barred by an X
    if constexpr (RX) {
      if (active && rec.has_x()) xw = rdx[r0 + wt * WT + lane];
      const uint32_t xc = XPOS_CNT(xw);
      if (xc > XPOS_MAX) xwin = 0xFFFFFFFFu;
      for (int k = 0; k < W; k++) {
        const uint32_t q1 = (uint32_t)mp->win[k];
#pragma unroll
        for (int q = 0; q < XPOS_MAX; q++)
          if ((uint32_t)q < xc && XPOS_AT(xw, q) - q1 < (uint32_t)ww) xwin |= 1u << k;
      }
    }
    if (ww <= 16 && mp->direct) {
      // the usual case: the window key is one 32-bit word
      const uint32_t kmask = ww == 16 ? 0xFFFFFFFFu : ((1u << (2 * ww)) - 1u);
      for (int k = 0; k < W; k++) {
        const uint32_t q1 = (uint32_t)mp->win[k], q2 = q1 + (uint32_t)ww;
        const uint32_t key = (uint32_t)rec.ext(2 * q1) & kmask;
        bool pt = active && (uint32_t)len >= q2 && !((xwin >> k) & 1u);
        if (mp->min_dinuc > 0) pt = pt && key_count_dinuc16(key, ww) >= mp->min_dinuc;
        bb_l[k * WT + lane] = pt ? __brev(key) >> (32 - 2 * ww) : WB_NONE;
        valid |= pt ? 1u << k : 0u;
      }
    } else {
      for (int k = 0; k < W; k++) {
        uint32_t b = WB_NONE;
        const uint32_t q1 = (uint32_t)mp->win[k], q2 = q1 + (uint32_t)ww;
        if (active) {
          bool pt = (uint32_t)len >= q2 && !((xwin >> k) & 1u);
          if (pt && mp->min_dinuc > 0)
            pt = (ww <= 16 ? rec_count_dinuc16(rec, q1, ww) : rec_count_dinuc(rec, rec, false, q1, ww)) >= mp->min_dinuc;
          if (pt) {
            b = rec_bucket(rec, rec, false, q1, ww, mp->bits, mp->direct);
            valid |= 1u << k;
          }
        }
        bb_l[k * WT + lane] = b;
      }
    }
    nvalid += __popc(valid);
    return valid;
  };
  // returns the tile's common read length or ~0
  auto phase_a2 = [&](uint32_t wt, const Rec<RW>& rec, uint32_t valid, uint32_t xw) -> uint32_t {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    if constexpr (RX) s_xp[wid][lane] = xw;
    uint32_t* const wcnt_l = s_dyn + wid * WT * W;
    const bool active = wt * WT + lane < n;
    const int len = (int)rec.len();
    {
      uint4* dst = reinterpret_cast<uint4*>(&s_rec[wid][RPAD + lane * RW]);
#pragma unroll
      for (int q = 0; q < RW / 4; q++) dst[q] = make_uint4(rec.w[4 * q], rec.w[4 * q + 1], rec.w[4 * q + 2], rec.w[4 * q + 3]);
    }
    const uint32_t budget = len < CONF_NM ? s_nm[len] : 0u;  // (reads on this path are at most 120 bases)
    s_meta[wid][lane] = (uint32_t)len | ((budget > 127u ? 127u : budget) << 17) | (valid << 24);
    if (block_mode)
      for (uint32_t t = lane; t < WT * (uint32_t)W; t += 64) wcnt_l[t] = 0;
    const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(len);
    return __ballot(active && (uint32_t)len != len0) == 0 ? len0 : 0xFFFFFFFFu;
  };
  // the bucket loads of one step (16 probes, a quad each)
  auto issue = [&](uint32_t bpar, int k, int rr, uint4& a, uint4& b2) {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6, part = lane & 3;
    const uint32_t b = s_bb[bpar][wid][k * WT + rr * 16 + (lane >> 2)];
    a.x = 0;  // a probe that takes no part reads as an empty bucket
    if (b != WB_NONE && !(dbg & 2)) {
      const u32x4_v* p = reinterpret_cast<const u32x4_v*>(T + b) + 2 * part;
      const u32x4_v x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1);
      a = make_uint4(x.x, x.y, x.z, x.w);
      b2 = make_uint4(y.x, y.y, y.z, y.w);
    }
  };

  // The tuples the PREVIOUS batch's launch staged (pstage != nullptr: same grid, same regions, the
  // other stage buffer) move to their final place in `hits` from inside this launch: the wave that
  // staged wave-tile wt of that batch copies it while it matches its own wave-tile wt -- the copy's
  // memory time hides under the comparisons instead of a k_compact_w launch between two launches.
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
  auto pcopy_begin = [&](uint32_t wt, uint4& v, uint32_t& m, uint32_t& d) {
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
  auto pcopy_end = [&](const uint4& v, uint32_t m, uint32_t d) {
    if (!m) return;
    const uint32_t lane = opaque(threadIdx.x) & 63;
    uint4* __restrict__ dst = hits + pbase + d;
    if (lane < m) dst[lane] = v;
    for (uint32_t i = 64 + lane; i < m; i += 64) dst[i] = pstage[region0 + pused + i];  // a tile with more than 64 tuples
    pused += m;
  };

  uint4 va[MATCH_RING], vb[MATCH_RING];
#pragma unroll
  for (int rr = 0; rr < MATCH_RING; rr++) va[rr] = vb[rr] = make_uint4(0, 0, 0, 0);
  uint32_t ulen = 0xFFFFFFFFu, par = 0;
  s_best[threadIdx.x >> 6][threadIdx.x & 63] = 0xFFFFFFFFu;
  if (gw < nwt) {
    Rec<RW> rec;
    fetch(gw, rec);
    uint32_t xw;
    const uint32_t valid = phase_a1(gw, 0, rec, xw);
    ulen = phase_a2(gw, rec, valid, xw);
    wave_lds_sync();
#pragma unroll
    for (int rr = 0; rr < MATCH_RING; rr++) issue(0, 0, rr, va[rr], vb[rr]);
  }
  uint32_t wt = gw;
  for (; wt < nwt; wt += nw) {
    uint4 cpv = make_uint4(0, 0, 0, 0);
    uint32_t cpm, cpd;
    pcopy_begin(wt, cpv, cpm, cpd);
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6, part = lane & 3;
    uint32_t* const bb_l = s_bb[par][wid];
    uint32_t* const oc_l = s_oc[wid];
    uint32_t* const ovf_l = s_ovf[wid];
    uint32_t* const wcnt_l = s_dyn + wid * WT * W;
    uint32_t* const best_l = s_best[wid];
    uint32_t* const q_l = s_q[wid];
    uint32_t* const cnt_l = q_l;  // (the stack is empty when phase D starts)
    uint32_t* const base_l = q_l + WT;
    const uint32_t* const rec_l = s_rec[wid] + RPAD;
    const bool have_next = wt + nw < nwt;
    Rec<RW> nrec;              // the next wave-tile's records, from a1 to a2
    uint32_t nvalid_next = 0;  // and its valid-window masks
    uint32_t nxw = 0;          // and (RX) its xpos words
    uint32_t nlist = 0;  // reported candidates of this wave-tile so far (wave-uniform)
    uint32_t qn = 0;     // entries on the stack (wave-uniform)

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
      if (slot < MATCHD_WLIST) {
        s_list[wid][slot] = make_uint3(w, gene, pos);
      } else if (slot - MATCHD_WLIST < sregion) {
        spill[sregion0 + (slot - MATCHD_WLIST)] = make_uint4(w, gene, pos, 0u);
      }
    };

    // the arrival step of 16 probes: what the lines hold goes onto the stack
    auto arrive = [&](int rr, const uint4& a, const uint4& b2) {
      const uint32_t ri = opaque((uint32_t)rr * 16 + (lane >> 2));
#define MUSC_Q0(X) (uint32_t)__builtin_amdgcn_mov_dpp((int)(X), 0x00, 0xF, 0xF, true)
      const uint32_t cnt = MUSC_Q0(a.x);
      const uint32_t g0 = MUSC_Q0(a.z), g1 = MUSC_Q0(a.w), g2 = MUSC_Q0(b2.x);
      const uint32_t j0 = MUSC_Q0(b2.y), j1 = MUSC_Q0(b2.z), j2 = MUSC_Q0(b2.w);
#undef MUSC_Q0
      if (part == 0) {
        ncand += cnt;
        oc_l[ri] = cnt > CTX_INLINE ? cnt - CTX_INLINE : 0u;
        ovf_l[ri] = a.y;
      }
      const bool live = part >= 1 && part - 1 < cnt && !(dbg & 1);
      const unsigned long long vote = __ballot(live);
      if (vote == 0) return;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      if (live) {
        const uint32_t gene = part == 1 ? g0 : (part == 2 ? g1 : g2);
        const uint32_t jx = part == 1 ? j0 : (part == 2 ? j1 : j2);
        uint4* dst = reinterpret_cast<uint4*>(&q_l[(qn + below) * MATCH_QW]);
        dst[0] = a;
        dst[1] = b2;
        dst[2] = make_uint4(gene, jx, ri, 0u);
      }
      qn += (uint32_t)__popcll(vote);
    };

    // a comparison pass over the top min(qn, 64) entries of the stack, all of window k
    auto pass = [&](auto ulen_tag, uint32_t k, int q1, uint32_t sh, const uint32_t (&lm)[8]) {
      constexpr bool ULEN = decltype(ulen_tag)::value;
      const uint32_t m = qn < 64u ? qn : 64u;
      qn -= m;
      const bool have = lane < m;
      const uint4* src = reinterpret_cast<const uint4*>(&q_l[(qn + (have ? lane : 0u)) * MATCH_QW]);
      const uint4 ca = src[0], cb = src[1], cc = src[2];
      uint32_t c[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
      const uint32_t gene = cc.x, jx = cc.y, ri = cc.z & 63u;
      const uint32_t meta = s_meta[wid][ri];
      const int rlen = (int)REC_LEN(meta);
      uint32_t z = 0;
      bool ok;
      if (__any(have && jx <= (uint32_t)q1)) ok = have & ctx_fit(jx, c[7] >> 16, q1, ww, rlen, &z);
      else ok = have & (rlen - q1 <= (int)(c[7] >> 16));
      {
        const unsigned long long okv = __ballot(ok);
        ncmp += lane == 0 ? (uint32_t)__popcll(okv) : 0u;
      }
      uint32_t w = NX_REJECT;
      if (ok) {
        // the read's image for this window: its record shifted left by sh bits (wave-uniform)
        const uint32_t wo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sh >> 5)), bs = sh & 31u;
        const uint32_t* rp = rec_l + ri * RW - wo;
        uint32_t x[9];
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = rp[j - 1];
        uint32_t img[8];
#pragma unroll
        for (int j = 0; j < 8; j++) img[j] = bs ? __builtin_amdgcn_alignbit(x[j + 1], x[j], 32u - bs) : x[j + 1];
        c[7] &= 0xFFFFu;
        const uint32_t exact0 = REC_VALID(meta) & (z ? ~q1zero : 0xFFFFFFFFu);
        uint32_t xm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (RX) {
          // the read's X in the image's coordinates (only passes that hold such a read pay for it)
          const uint32_t xw = s_xp[wid][ri];
          if (__any(xw != 0)) {
            const uint32_t xc = XPOS_CNT(xw);
#pragma unroll
            for (int q = 0; q < XPOS_MAX; q++) {
              const uint32_t b = sh + 2u * XPOS_AT(xw, q);
              const uint32_t bit = (uint32_t)q < xc ? 1u << (b & 31u) : 0u;
#pragma unroll
              for (int j = 0; j < 8; j++) xm[j] |= (b >> 5) == (uint32_t)j ? bit : 0u;
            }
          }
        }
        w = ctx_score<ULEN, RX>(img, c, sh, k, mp, W, exact0, REC_BUDGET(meta), ri, ULEN ? ulen : (uint32_t)rlen, lm, xm);
      }
      report(w, gene, jx - (uint32_t)q1);
    };

    // ---- phase B: per window four arrival steps, its overflow entries, and the passes
    {
      auto window_loop = [&](auto ulen_tag) {
        constexpr bool ULEN = decltype(ulen_tag)::value;
#pragma unroll 1
        for (int k = 0; k < W; k++) {
          const int q1 = W2 ? (k == 0 ? win0 : win1) : mp->win[k];
          const uint32_t sh = opaque_s(2u * (uint32_t)(CL - q1));
          uint32_t lm[8];
          {
            const uint32_t* __restrict__ row = mp->lm[ULEN ? __builtin_amdgcn_readfirstlane((int)ulen) : 0][k];
#pragma unroll
            for (int j = 0; j < 8; j++) lm[j] = ULEN ? row[j] : 0u;
          }
          // before the last window: the next wave-tile's buckets (a1), so that this window's
          // arrival steps can refill the ring with the next tile's first steps
          if (k + 1 == W && have_next) {
            fetch(wt + nw, nrec);
            nvalid_next = phase_a1(wt + nw, par ^ 1u, nrec, nxw);
            wave_lds_sync();
          }
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            arrive(rr, va[rr % MATCH_RING], vb[rr % MATCH_RING]);
            // refill the slot with the step MATCH_RING ahead: this window's, the next window's, or
            // the next wave-tile's
            if (rr + MATCH_RING < 4) issue(par, k, rr + MATCH_RING, va[rr % MATCH_RING], vb[rr % MATCH_RING]);
            else if (k + 1 < W) issue(par, k + 1, rr + MATCH_RING - 4, va[rr % MATCH_RING], vb[rr % MATCH_RING]);
            else if (have_next) issue(par ^ 1u, 0, rr + MATCH_RING - 4, va[rr % MATCH_RING], vb[rr % MATCH_RING]);
            wave_lds_sync();
            if (qn >= (uint32_t)MATCHD_QTHR) pass(ulen_tag, (uint32_t)k, q1, sh, lm);
          }
          // the entries beyond a bucket's third, MATCH_DOWN per step, a lane per entry
          if (!(dbg & 4)) {
            const uint32_t oc = oc_l[lane];
            const uint32_t inc = wave_scan_incl(oc);
            const uint32_t pre = inc - oc;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            novf += oc;
            for (uint32_t c0 = 0; c0 < total; c0 += MATCH_DOWN) {
              // this lane's items that fall into [c0, c0 + MATCH_DOWN)
              const uint32_t e_lo = c0 > pre ? c0 - pre : 0u;
              const uint32_t e_hi = pre + oc > c0 + MATCH_DOWN ? (c0 + MATCH_DOWN > pre ? c0 + MATCH_DOWN - pre : 0u) : oc;
              for (uint32_t e = e_lo; e < e_hi; e++) {
                s_own[wid][pre + e - c0] = (uint8_t)lane;
                s_oix[wid][pre + e - c0] = e;
              }
              wave_lds_sync();
              const uint32_t cntc = total - c0 < MATCH_DOWN ? total - c0 : MATCH_DOWN;
              const bool have = lane < cntc && !(dbg & 1);
              if (have) {
                const uint32_t seg = s_own[wid][lane];
                const uint32_t* __restrict__ pe =
                    reinterpret_cast<const uint32_t*>(E + ((uint64_t)ovf_l[seg] + s_oix[wid][lane]));
                const uint2 hd = *reinterpret_cast<const uint2*>(pe);
                const u32x4_u x = *reinterpret_cast<const u32x4_u*>(pe + 2);
                const u32x4_u y = *reinterpret_cast<const u32x4_u*>(pe + 6);
                uint4* dst = reinterpret_cast<uint4*>(&q_l[(qn + lane) * MATCH_QW]);
                dst[0] = make_uint4(x.x, x.y, x.z, x.w);
                dst[1] = make_uint4(y.x, y.y, y.z, y.w);
                dst[2] = make_uint4(hd.x, hd.y, seg, 0u);
              }
              if (!(dbg & 1)) qn += cntc;
              wave_lds_sync();
              if (qn >= (uint32_t)MATCHD_QTHR) pass(ulen_tag, (uint32_t)k, q1, sh, lm);
            }
          }
          if (qn) pass(ulen_tag, (uint32_t)k, q1, sh, lm);
          wave_lds_sync();  // the next window's arrival steps rewrite s_oc / s_ovf and the stack
        }
      };
      if (ulen != 0xFFFFFFFFu) window_loop(std::true_type{}); else window_loop(std::false_type{});
    }
    wave_lds_sync();

    // ---- phase D: per-read selection and the tuples (as in k_match)
    {
      const uint32_t nl = nlist;
      const uint32_t nspill = nl > MATCHD_WLIST ? nl - MATCHD_WLIST : 0u;
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
      wave_lds_sync();  // phases A..B of this tile are done with the per-tile LDS state
      // the next wave-tile's records and meta words take the place of this one's (its first bucket
      // lines have been on their way since this tile's last window)
      uint32_t ulen_next = 0xFFFFFFFFu;
      if (have_next) ulen_next = phase_a2(wt + nw, nrec, nvalid_next, nxw);
      pcopy_end(cpv, cpm, cpd);
      cnt_l[lane] = 0;  // the stack becomes cnt / base
      wave_lds_sync();
      auto item = [&](uint32_t j, uint32_t* gene, uint32_t* pos) -> uint32_t {
        if (j < MATCHD_WLIST) {
          const uint3 it = s_list[wid][j];
          *gene = it.y;
          *pos = it.z;
          return it.x;
        }
        // written by other lanes of this wave a moment ago: read past the L1
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(spill + sregion0 + (j - MATCHD_WLIST));
        *gene = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *pos = __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      const uint32_t nuse = spill_ok ? nl : (nl < MATCHD_WLIST ? nl : MATCHD_WLIST);
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
      best_l[lane] = 0xFFFFFFFFu;  // for the next wave-tile
      used += total;
      nrep += lane == 0 ? nl : 0u;
      ulen = ulen_next;
      par ^= 1u;
      wave_lds_sync();  // the next wave-tile's phase B rewrites the stack / the candidate list
    }
  }
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
