// k_match_t -- the fused screen + confirm + select kernel on context buckets with the comparisons done
// IN THE LANE THAT OWNS THE READ (device code, included by muscato_hip.hip after kernels_match.hpp and
// kernels_match_dense.hpp, whose bucket layout, parameter block, fit rules, cdiff and tuple staging
// protocol it shares).
//
// k_match compares where a bucket line arrives (a quad of lanes per probe) and k_match_d moves the
// entries that exist onto a stack in LDS and pops 64 of them per comparison pass.  Both spend most
// of their instructions on getting an entry and "its" read into the same lane: DPP moves, votes,
// stack writes, per-entry record reads and funnel shifts, owner tables -- 2 640 instructions per
// wave-tile for 237 comparisons that need about 50 instructions each.  Here the bucket line is
// TRANSPOSED instead: the quad that fetched a line (coalesced: 32 bytes per lane) writes it into a
// per-wave line buffer in LDS as it arrived, and once the 64 lines of a window are there, lane p
// reads line p -- all 128 bytes, eight ds_read_b128 -- into its own registers, where the read's
// record already is.  Everything cdiff needs is then in one lane:
//   * the read's image for the window is built once per (read, window) from registers
//     (read_image), not once per entry from LDS;
//   * the three inline entries of the line are compared by three straight-line blocks with no
//     memory access in them (the lane's registers hold count, target numbers, window positions and
//     contexts): no stack, no entry copies, no per-entry meta / record / mask lookups;
//   * best-per-read and the MaxMatches counters of (window, read) are registers of that lane.
// 56 % of the slots of those blocks hold an entry (1.67 inline entries per probe on cfg3) -- less
// than a dense pass, but a slot costs ~55 vector instructions and nothing else.
// The line buffer is XOR-swizzled at 16-byte granularity (chunk c of line p sits at slot
// c ^ ((p >> 1) & 7) ^ (p & 1)), which makes both the quads' writes and the lanes' whole-line reads
// bank-conflict free.
//
// Loads in flight: a ring of 4 x W register slots of 16 lines each -- one per (window, step) of a
// wave-tile; the slot of step (k, rr) is refilled with the same step of the wave's NEXT wave-tile
// the moment its lines have been written to LDS, so a wave has up to 64 x W lines in flight and
// the buckets of a wave-tile are requested one whole wave-tile ahead (phase A of tile t + 1 runs at
// the start of tile t, on records that were fetched during tile t - 1).
//
// The entries beyond a bucket's third (CtxEntry in E, 10 % of the entries walked on cfg3) take one
// generic pass per wave-tile, a lane per entry, with the read's record from LDS (as k_match_d's
// passes); their loads are issued before the last window's comparisons and consumed after them.
//
// LDS per workgroup of four waves: 4 x 14.2 KB + the MaxMatches sketch = 63 KB: two workgroups
// per CU, two waves per SIMD -- what in-lane comparisons need is registers (ring 64, line 32,
// records 24), not occupancy: their instruction stream has no LDS or memory latency to hide.
#pragma once
#include "kernels_match_lane_inst.hpp"

template <int RW, int W>
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
  constexpr int NWAVE = TILE / 64;
  constexpr int RD = 4 * W;  // ring slots: one per (window, step of 16 probes)
  constexpr int RPAD = 8;    // words in front of and behind a wave's records that a shifted window may touch
  extern __shared__ uint32_t s_dyn[];  // block_mode != 0: NWAVE x W x 64 counters of the overflow pass, then (mode 1) the sketch
  __shared__ uint4 s_line[NWAVE][64 * 8];                 // the window's 64 bucket lines, swizzled
  __shared__ __attribute__((aligned(16))) uint32_t s_rec[NWAVE][WT * RW + 2 * RPAD];  // the wave-tile's records (overflow pass)
  __shared__ uint32_t s_meta[NWAVE][WT];                  // length | budget << 17 | valid windows << 24
  __shared__ uint32_t s_bb[2][NWAVE][W * WT];             // bucket of (window, read), WB_NONE when the window takes no part; this tile's and the next one's
  __shared__ uint32_t s_best[NWAVE][WT];                  // smallest mismatch count reported per read
  __shared__ uint3 s_list[NWAVE][MATCH_WLIST];            // reported candidates: result word, gene, position
  __shared__ uint32_t s_cb[NWAVE][2 * WT];                // phase D: cnt[64], base[64]
  __shared__ uint32_t s_oix[NWAVE][WT];                   // overflow pass: item -> its entry in E
  __shared__ uint8_t s_own[NWAVE][WT];                    //                item -> window * 64 + read slot
  __shared__ uint16_t s_nm[CONF_NM];
  __shared__ uint32_t s_xp[NWAVE][WT];                    // reads with X (rdx != nullptr): their xpos words (overflow pass)

  const int ww = mp->ww, CL = mp->CL;
  const uint32_t q1zero = mp->q1zero_mask;
  const bool RX = rdx != nullptr;  // some read of the batch holds an X (wave-uniform)
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

  auto fetch = [&](uint32_t wt, Rec<RW>& rec) {
    const uint32_t i = wt * WT + (opaque(threadIdx.x) & 63);
    rec.load(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW, RW);
  };
  // phase A of wave-tile wt, a lane per read: per window the length gate + CountDinuc >= MinDinuc
  // (cmd/muscato_window_reads/main.go:106-118 == cmd/muscato_screen/main.go:174-185) and the bucket
  // of the window key -> s_bb[par].  Returns the windows that take part.
  auto phase_a = [&](uint32_t wt, uint32_t par, const Rec<RW>& rec, uint32_t& xw) -> uint32_t {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint32_t* const bb_l = s_bb[par][wid];
    const bool active = wt * WT + lane < n;
    const int len = (int)rec.len();
    uint32_t valid = 0;
    // RX: windows that hold an X never probe; a read with more X than fit the word has no tuples
    xw = 0;
    uint32_t xwin = 0;  // windows barred by an X
    if (RX) {
      if (active && rec.has_x()) xw = rdx[r0 + wt * WT + lane];
      const uint32_t xc = XPOS_CNT(xw);
      if (xc > XPOS_MAX) xwin = 0xFFFFFFFFu;
#pragma unroll
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
#pragma unroll
      for (int k = 0; k < W; k++) {
        const uint32_t q1 = (uint32_t)mp->win[k], q2 = q1 + (uint32_t)ww;
        const uint32_t key = (uint32_t)rec.ext(2 * q1) & kmask;
        bool pt = active && (uint32_t)len >= q2 && !((xwin >> k) & 1u);
        if (mp->min_dinuc > 0) pt = pt && key_count_dinuc16(key, ww) >= mp->min_dinuc;
        bb_l[k * WT + lane] = pt ? __brev(key) >> (32 - 2 * ww) : WB_NONE;
        valid |= pt ? 1u << k : 0u;
      }
    } else {
#pragma unroll
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
  // the bucket loads of one step (16 probes, a quad each: 32 bytes per lane)
  auto issue = [&](uint32_t bpar, int k, int rr, uint4& a, uint4& b2) {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6, part = lane & 3;
    const uint32_t b = s_bb[bpar][wid][k * WT + rr * 16 + (lane >> 2)];
    a.x = 0;  // a probe that takes no part reads as an empty bucket (count 0 in the quad's first lane)
    if (b != WB_NONE) {
      const u32x4_v* p = reinterpret_cast<const u32x4_v*>(T + b) + 2 * part;
      const u32x4_v x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1);
      a = make_uint4(x.x, x.y, x.z, x.w);
      b2 = make_uint4(y.x, y.y, y.z, y.w);
    }
  };

  // The tuples the PREVIOUS batch's launch staged (pstage != nullptr: same grid, same regions, the
  // other stage buffer) move to their final place in `hits` from inside this launch (as in k_match_d)
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

  uint4 va[RD], vb[RD];
#pragma unroll
  for (int s = 0; s < RD; s++) va[s] = vb[s] = make_uint4(0, 0, 0, 0);
  uint32_t par = 0;
  Rec<RW> rec_cur, rec_nx;
  rec_cur.zero();
  rec_nx.zero();
  uint32_t valid_cur = 0, xw_cur = 0;
  s_best[threadIdx.x >> 6][threadIdx.x & 63] = 0xFFFFFFFFu;
  if (gw < nwt) {
    fetch(gw, rec_cur);
    valid_cur = phase_a(gw, 0, rec_cur, xw_cur);
    if (gw + nw < nwt) fetch(gw + nw, rec_nx);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < W; k++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) issue(0, k, rr, va[k * 4 + rr], vb[k * 4 + rr]);
  }
  uint32_t wt = gw;
  for (; wt < nwt; wt += nw) {
    uint4 cpv = make_uint4(0, 0, 0, 0);
    uint32_t cpm, cpd;
    pcopy_begin(wt, cpv, cpm, cpd);
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6, part = lane & 3;
    uint32_t* const wcnt_l = s_dyn + wid * WT * W;
    uint32_t* const best_l = s_best[wid];
    uint32_t* const cnt_l = s_cb[wid];
    uint32_t* const base_l = s_cb[wid] + WT;
    const uint32_t* const rec_l = s_rec[wid] + RPAD;
    uint4* const line_l = s_line[wid];
    const bool have_next = wt + nw < nwt;
    const bool active = wt * WT + lane < n;

    // ---- the next wave-tile's buckets (its records were fetched during the previous wave-tile), then
    // the fetch of the records after those
    uint32_t valid_nx = 0, xw_nx = 0;
    Rec<RW> rec_pre;
    rec_pre.zero();
    if (have_next) {
      valid_nx = phase_a(wt + nw, par ^ 1u, rec_nx, xw_nx);
      if (wt + 2 * nw < nwt) fetch(wt + 2 * nw, rec_pre);
    }
    // this wave-tile's per-read state: registers of the read's lane; the record and the meta word
    // also go to LDS for the overflow pass
    const int rlen = (int)rec_cur.len();
    const uint32_t budget = rlen < CONF_NM ? s_nm[rlen] : 0u;  // (reads on this path are at most 120 bases)
    {
      uint4* dst = reinterpret_cast<uint4*>(&s_rec[wid][RPAD + lane * RW]);
#pragma unroll
      for (int q = 0; q < RW / 4; q++)
        dst[q] = make_uint4(rec_cur.w[4 * q], rec_cur.w[4 * q + 1], rec_cur.w[4 * q + 2], rec_cur.w[4 * q + 3]);
      s_meta[wid][lane] = (uint32_t)rlen | ((budget > 127u ? 127u : budget) << 17) | (valid_cur << 24);
      if (RX) s_xp[wid][lane] = xw_cur;
      if (block_mode)
        for (uint32_t t = lane; t < WT * (uint32_t)W; t += 64) wcnt_l[t] = 0;
    }
    // every read of the wave-tile of one length: the comparisons use scalar length masks
    const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(rlen);
    const uint32_t ulen = __ballot(active && (uint32_t)rlen != len0) == 0 ? len0 : 0xFFFFFFFFu;
    wave_lds_sync();

    uint32_t nlist = 0;          // reported candidates of this wave-tile so far (wave-uniform)
    uint32_t best = 0xFFFFFFFFu; // smallest mismatch count reported for this lane's read by the in-lane comparisons
    uint32_t wc[W];              // accepted pairs of (window, this lane's read): MaxMatches accounting
    uint32_t oc[W], ovf[W];      // entries beyond the third of this lane's probe of window k, and where in E
#pragma unroll
    for (int k = 0; k < W; k++) wc[k] = oc[k] = ovf[k] = 0;

    // a reported candidate of the lane's own read (in-lane comparisons): appended in lane order
    auto report_own = [&](uint32_t w, uint32_t gene, uint32_t pos, uint32_t& wck) {
      const bool acc = w != NX_REJECT;
      wck += acc ? 1u : 0u;
      const bool rep = acc && !(w & NX_DUP);
      const unsigned long long vote = __ballot(rep);
      if (vote == 0) return;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      const uint32_t slot = nlist + below;
      nlist += (uint32_t)__popcll(vote);
      if (!rep) return;
      const uint32_t v = w & 0xFFFFu;
      best = v < best ? v : best;
      if (slot < MATCH_WLIST) {
        s_list[wid][slot] = make_uint3(w, gene, pos);
      } else if (slot - MATCH_WLIST < sregion) {
        spill[sregion0 + (slot - MATCH_WLIST)] = make_uint4(w, gene, pos, 0u);
      }
    };
    // a reported candidate of any read of the wave-tile (overflow pass)
    auto report_any = [&](uint32_t w, uint32_t gene, uint32_t pos) {
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

    // what the overflow pass keeps between the issue of its loads and their use
    uint32_t o_total = 0, o_k = 0, o_seg = 0, o_gene = 0, o_jx = 0;
    uint32_t o_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t o_pre[W];
#pragma unroll
    for (int k = 0; k < W; k++) o_pre[k] = 0;
    // chunk c0 of the overflow items: owner tables, then a lane per item loads its entry
    auto overflow_issue = [&](uint32_t c0) {
#pragma unroll
      for (int k = 0; k < W; k++) {
        // this lane's items of window k that fall into [c0, c0 + 64)
        const uint32_t e_lo = c0 > o_pre[k] ? c0 - o_pre[k] : 0u;
        const uint32_t e_hi = o_pre[k] + oc[k] > c0 + WT ? (c0 + WT > o_pre[k] ? c0 + WT - o_pre[k] : 0u) : oc[k];
        for (uint32_t e = e_lo; e < e_hi; e++) {
          s_own[wid][o_pre[k] + e - c0] = (uint8_t)(k * WT + lane);
          s_oix[wid][o_pre[k] + e - c0] = ovf[k] + e;
        }
      }
      wave_lds_sync();
      if (c0 + lane < o_total) {
        const uint32_t probe = s_own[wid][lane];
        o_k = probe >> 6;
        o_seg = probe & 63u;
        const uint32_t* __restrict__ pe = reinterpret_cast<const uint32_t*>(E + (uint64_t)s_oix[wid][lane]);
        const uint2 hd = *reinterpret_cast<const uint2*>(pe);
        const u32x4_u x = *reinterpret_cast<const u32x4_u*>(pe + 2);
        const u32x4_u y = *reinterpret_cast<const u32x4_u*>(pe + 6);
        o_gene = hd.x;
        o_jx = hd.y;
        o_c[0] = x.x; o_c[1] = x.y; o_c[2] = x.z; o_c[3] = x.w; o_c[4] = y.x; o_c[5] = y.y; o_c[6] = y.z; o_c[7] = y.w;
      }
      wave_lds_sync();  // the owner tables may be rewritten
    };

    // the length mask of a read of `len` bases placed through window k (one bit per base that takes part
    // in cdiff, in the coordinates of the context stream): a row of the host's table when every read
    // of the wave-tile has the same length (scalar loads), per-lane arithmetic otherwise
    auto length_mask = [&](int k, uint32_t sh, uint32_t len, uint32_t (&lm)[8]) {
      if (ulen != 0xFFFFFFFFu) {
        const uint32_t* __restrict__ row = mp->lm[__builtin_amdgcn_readfirstlane((int)ulen)][k];
#pragma unroll
        for (int j = 0; j < 8; j++) lm[j] = row[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) lm[j] = 0x55555555u & bit_range_mask((int)sh - 32 * j, (int)sh + 2 * (int)len - 32 * j);
      }
    };
    // the X of a read in the image's coordinates (RX)
    auto x_mask = [&](uint32_t xw, uint32_t sh, uint32_t (&xm)[8]) {
#pragma unroll
      for (int j = 0; j < 8; j++) xm[j] = 0;
      if (RX) {
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
    };
    // a generic comparison of the overflow pass: the entry is in this lane, the read (slot seg) in LDS
    auto compare_any = [&](uint32_t k, int q1, uint32_t sh, uint32_t seg, uint32_t jx, bool live, const uint32_t (&c)[8]) -> uint32_t {
      const uint32_t meta = s_meta[wid][seg];
      const int len_s = (int)REC_LEN(meta);
      uint32_t z = 0;
      bool ok;
      if (__any(live && jx <= (uint32_t)q1)) ok = live & ctx_fit(jx, c[7] >> 16, q1, ww, len_s, &z);
      else ok = live & (len_s - q1 <= (int)(c[7] >> 16));
      {
        const unsigned long long okv = __ballot(ok);
        ncmp += lane == 0 ? (uint32_t)__popcll(okv) : 0u;
      }
      uint32_t w = NX_REJECT;
      if (ok) {
        const uint32_t wo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sh >> 5)), bs = sh & 31u;
        const uint32_t* rp = rec_l + seg * RW - wo;
        uint32_t x[9];
#pragma unroll
        for (int j = 0; j < 9; j++) x[j] = rp[j - 1];
        uint32_t img[8];
#pragma unroll
        for (int j = 0; j < 8; j++) img[j] = bs ? __builtin_amdgcn_alignbit(x[j + 1], x[j], 32u - bs) : x[j + 1];
        uint32_t cc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) cc[j] = c[j];
        cc[7] &= 0xFFFFu;
        const uint32_t exact0 = REC_VALID(meta) & (z ? ~q1zero : 0xFFFFFFFFu);
        uint32_t lm[8], xm[8];
        length_mask((int)k, sh, (uint32_t)len_s, lm);
        uint32_t xw = 0;
        if (RX) xw = s_xp[wid][seg];
        x_mask(xw, sh, xm);
        w = ctx_score<true, true>(img, cc, sh, k, mp, W, exact0, REC_BUDGET(meta), seg, (uint32_t)len_s, lm, xm);
      }
      return w;
    };

#pragma unroll
    for (int k = 0; k < W; k++) {
      const int q1 = mp->win[k];
      const uint32_t sh = opaque_s(2u * (uint32_t)(CL - q1));
      // ---- arrival: the quads write the lines they fetched into the line buffer (swizzled), and
      // each ring slot is refilled with the same step of the next wave-tile
      {
        const uint32_t q = lane >> 2;
        const uint32_t sw = ((q >> 1) & 7u) ^ (q & 1u);
        const uint32_t wb0 = q * 8u + ((2u * part) ^ sw);  // in units of 16 bytes; the line's second half: ^ 1
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          line_l[rr * 128 + wb0] = va[k * 4 + rr];
          line_l[rr * 128 + (wb0 ^ 1u)] = vb[k * 4 + rr];
          if (have_next) issue(par ^ 1u, k, rr, va[k * 4 + rr], vb[k * 4 + rr]);
        }
      }
      wave_lds_sync();
      // ---- transpose: lane p takes line p -- its header now, an entry's context when its turn comes
      const uint32_t rb = lane * 8u + (((lane >> 1) & 7u) ^ (lane & 1u));
      const uint4 h0 = line_l[rb], h1 = line_l[rb ^ 1u];
      uint4 ca = line_l[rb ^ 2u], cb = line_l[rb ^ 3u];
      const uint32_t cnt = h0.x;
      ncand += cnt;
      oc[k] = cnt > CTX_INLINE ? cnt - CTX_INLINE : 0u;
      ovf[k] = h0.y;
      novf += oc[k];
      // the entries beyond the third: their loads go out before the last window's comparisons
      if (k + 1 == W) {
        uint32_t total = 0;
#pragma unroll
        for (int kk = 0; kk < W; kk++) {
          const uint32_t inc = wave_scan_incl(oc[kk]);
          o_pre[kk] = total + inc - oc[kk];
          total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        o_total = total;
        if (total) overflow_issue(0);
      }
      // ---- the read's image for this window, then the line's three entries, in this lane
      uint32_t img[8], lm[8], xm[8];
      read_image<RW>(rec_cur, sh, img);
      length_mask(k, sh, (uint32_t)rlen, lm);
      x_mask(xw_cur, sh, xm);
#pragma unroll
      for (int s = 0; s < CTX_INLINE; s++) {
        const bool live = (uint32_t)s < cnt;
        if (!__any(live)) break;
        uint4 na = ca, nb = cb;
        if (s + 1 < CTX_INLINE) {  // the next entry's context is on its way while this one is compared
          na = line_l[rb ^ (uint32_t)(2 * s + 4)];
          nb = line_l[rb ^ (uint32_t)(2 * s + 5)];
        }
        const uint32_t gene = s == 0 ? h0.z : (s == 1 ? h0.w : h1.x);
        const uint32_t jx = s == 0 ? h1.y : (s == 1 ? h1.z : h1.w);
        uint32_t c[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
        // placements past the target's first bases (p = jx - q1 > 0) only have to end inside the
        // target; the pos-0 rules are evaluated only when some lane of the wave is at p <= 0
        uint32_t z = 0;
        bool ok;
        if (__any(live && jx <= (uint32_t)q1)) ok = live & ctx_fit(jx, c[7] >> 16, q1, ww, rlen, &z);
        else ok = live & (rlen - q1 <= (int)(c[7] >> 16));
        {
          const unsigned long long okv = __ballot(ok);
          ncmp += lane == 0 ? (uint32_t)__popcll(okv) : 0u;
        }
        uint32_t w = NX_REJECT;
        if (ok) {
          c[7] &= 0xFFFFu;
          const uint32_t exact0 = valid_cur & (z ? ~q1zero : 0xFFFFFFFFu);
          w = ctx_score<true, true>(img, c, sh, (uint32_t)k, mp, W, exact0, budget > 127u ? 127u : budget, lane, (uint32_t)rlen, lm, xm);
        }
        report_own(w, gene, jx - (uint32_t)q1, wc[k]);
        ca = na;
        cb = nb;
      }
      wave_lds_sync();  // (the next window's arrival writes come after this window's reads)
    }

    // ---- the overflow pass: a lane per entry beyond a bucket's third, all windows of the wave-tile
    for (uint32_t c0 = 0; c0 < o_total; c0 += WT) {
      if (c0) overflow_issue(c0);
      const bool have = c0 + lane < o_total;
      uint32_t w = NX_REJECT;
      int q1 = 0;
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        const bool mine = have && o_k == (uint32_t)kk;
        if (!__any(mine)) continue;
        const int q1k = mp->win[kk];
        const uint32_t shk = opaque_s(2u * (uint32_t)(CL - q1k));
        const uint32_t w2 = compare_any((uint32_t)kk, q1k, shk, o_seg, o_jx, mine, o_c);
        if (mine) {
          w = w2;
          q1 = q1k;
        }
      }
      report_any(w, o_gene, o_jx - (uint32_t)q1);
    }
    wave_lds_sync();

    // ---- phase D: per-read selection and the tuples (as in k_match_d)
    {
      const uint32_t nl = nlist;
      const uint32_t nspill = nl > MATCH_WLIST ? nl - MATCH_WLIST : 0u;
      const bool spill_ok = nspill <= sregion;
      if (nspill > maxspill) maxspill = nspill;
      if (nspill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's spilled candidates have landed
      if (block_mode) {
        const uint32_t* const bb_l = s_bb[par][wid];
#pragma unroll
        for (int k = 0; k < W; k++) {
          const uint32_t cw = wc[k] + wcnt_l[k * WT + lane];
          if (!cw) continue;
          const uint64_t h = mix64(((uint64_t)k << 32) | bb_l[k * WT + lane]);
          if (block_mode == 1) atomicAdd(&s_sketch[h >> (64 - MATCH_SKETCH_BITS)], cw);
          else atomicAdd(&block_table[h >> (64 - BLOCK_TABLE_BITS)], cw);
        }
      }
      pcopy_end(cpv, cpm, cpd);
      // the lane's own best joins what the overflow pass found for its read
      {
        const uint32_t b0 = best_l[lane];
        best_l[lane] = best < b0 ? best : b0;
      }
      cnt_l[lane] = 0;
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
      wave_lds_sync();
      best_l[lane] = 0xFFFFFFFFu;  // for the next wave-tile
      used += total;
      nrep += lane == 0 ? nl : 0u;
      par ^= 1u;
      rec_cur = rec_nx;
      rec_nx = rec_pre;
      valid_cur = valid_nx;
      xw_cur = xw_nx;
      wave_lds_sync();  // the next wave-tile rewrites the records, the candidate list and cnt / base
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
