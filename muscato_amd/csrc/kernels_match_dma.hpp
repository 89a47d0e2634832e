// k_match_g -- the fused screen + confirm + select kernel at THREE TO FOUR waves per SIMD (round 4; device code, compiled
// in match_dma_rw8.hip, declared in kernels_match_lane_inst.hpp; bucket layout, parameter block and fit rules of
// kernels_match.hpp; the comparison itself is k_match_t's: in the lane that owns the read).
//
// k_match_t (kernels_match_lane.hpp) lands a window's 64 bucket lines in 32 registers, writes them to a line buffer,
// keeps two generations of records, candidate lists and per-read tables in LDS and carries a wave-tile's overflow
// entries in registers across the tile boundary: 211-225 VGPRs and 16 KB of LDS per wave = two waves per SIMD, and
// r03's counters said what that costs: 43 % of a wave's cycles execute, 39 % wait, with ONE other wave to cover them.
// Here a wave owns ONE 8 KB buffer and everything that comes from memory lands in it by LDS-DMA
// (global_load_lds_dwordx4: L2 -> LDS, no destination register, no ds_write pass):
//   * the 64 lines of a window: eight instructions of EIGHT WHOLE LINES each (eight lanes per line, 16 contiguous
//     bytes per lane; the per-lane SOURCE address picks the chunk that belongs at the lane's place in the swizzled
//     buffer -- chunk c of line p at slot c ^ ((p >> 1) & 7) ^ (p & 1) -- so lane p reads line p without bank
//     conflicts, exactly as in k_match_t);
//   * then, in the same buffer, the wave-tile's overflow entries (beyond a bucket's third, 40 bytes each: three
//     16-byte pieces per entry -- the pieces' tails belong to the entry's line, ctx_entry_word -- 64 entries per
//     round) next to the NEXT wave-tile's records (64 * RW words in a row = RW / 4 instructions of 1 KB);
//   * and phase D's per-read tables (best, count, base) once the entries have been compared.
// No state crosses a wave-tile boundary but the next tile's record, buckets and meta word: no second generation of
// anything.  A lane keeps its read's record in registers; an overflow entry's lane gets its read's record and meta
// word by ds_bpermute.  9 KB of LDS per wave, <= 128 VGPRs: FOUR workgroups of four waves per CU.
//
// The schedule of a wave-tile (two windows; "flight" = LDS-DMA issued ... s_waitcnt vmcnt(0)):
//   F1  lines of window 0 -> buffer | the read's image and the mask tables for window 0 | wait | header + the three
//       inline entries, in the lane that owns the read
//   F2  the same for window 1; then the overflow entries of both windows are listed (a lane per entry)
//   F3  overflow entries + the next wave-tile's records -> buffer | wait | the entries' comparison pass, phase D
//       (per-read best + MMTol selection, tuples), the next wave-tile's phase A
// A wave exposes its three memory round trips -- by design: fifteen other waves on its CU cover them and keep the
// memory system full, where k_match_t had seven and hid one round trip behind its own arithmetic.  All vector-memory
// traffic of the loop is LDS-DMA or a store (the compiler's waitcnt insertion cannot see an LDS-DMA: every wait for
// one is written out here), except the rare paths that say so.
//
// Built for two windows on 120-base buckets, no X on either side (BASELINE configs 2-4); every other run keeps
// k_match_t.
#pragma once
#include "kernels_match_lane_inst.hpp"


// one LDS-DMA instruction: every active lane's 16 bytes at gsrc land at lds_dst + 16 * lane (lds_dst wave-uniform)
DEV void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// eight of them, 1 KB apart in LDS: the 64 lines of a window
DEV void glds16x8(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, const void* p5, const void* p6,
                  const void* p7, uint32_t lds_dst) {
  uint32_t keep;
#ifdef MATCHG_NT
#define MUSC_GLDS_LINE(N) "global_load_lds_dwordx4 %" #N ", off nt\n\t"
#else
#define MUSC_GLDS_LINE(N) "global_load_lds_dwordx4 %" #N ", off\n\t"
#endif
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %9\n\ts_nop 0\n\t" MUSC_GLDS_LINE(1)
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t" MUSC_GLDS_LINE(2)
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t" MUSC_GLDS_LINE(3)
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t" MUSC_GLDS_LINE(4)
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t" MUSC_GLDS_LINE(5)
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t" MUSC_GLDS_LINE(6)
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t" MUSC_GLDS_LINE(7)
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t" MUSC_GLDS_LINE(8)
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7), "s"(lds_dst)
               : "memory", "scc");
#undef MUSC_GLDS_LINE
}
DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the byte address of a __shared__ object within the workgroup's LDS, wave-uniform (what M0 wants)
template <class T>
DEV uint32_t lds_addr(const T* p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p); }

template <int RW, int SG>
__global__ __launch_bounds__(TILE, MATCHG_WAVES_OF(SG)) void k_match_g(const uint32_t* __restrict__ rd, uint64_t r0, uint32_t n,
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
  (void)rdx;  // (reads with X stay with k_match_t)
  constexpr int W = 2, NW = 8, NIN = CTX_INLINE;
  constexpr int NWAVE = TILE / 64;
  constexpr uint32_t WLIST = (uint32_t)MATCHG_WLIST;
  static_assert(RW % 4 == 0 && RW <= 8, "the next wave-tile's records land in 2 KB of the wave's buffer");
  extern __shared__ uint32_t s_dyn[];                      // block_mode == 1: the MaxMatches sketch of the workgroup
  // The wave's buffer, 512 x 16 bytes, by phase:
  //   F1 / F2  [0, 512)    the window's 64 bucket lines, swizzled
  //   F3       [0, 192)    the overflow entries in hand: three planes of 64 x 16 bytes
  //            [192, 256)  the first 64 tuples the PREVIOUS batch's launch staged for this wave-tile, on their way to `hits`
  //            [256, 384)  the next wave-tile's records
  //            [384, 400)  best[64]: smallest mismatch count the entries' pass reported per read
  //            [400, 416)  phase D: cnt[64]      [416, 432)  base[64]
  //            [432, 448)  item -> its entry in E (64 words)   [448, 452)  item -> window * 64 + read slot (64 bytes)
  __shared__ uint4 s_line[NWAVE][64 * 8];
  __shared__ uint3 s_list[NWAVE][WLIST];                   // reported candidates: result word, gene, position
  __shared__ uint16_t s_nm[CONF_NM / 2];                   // (reads on this path are at most 112 bases)

  typedef SpecGeom<SG> SGm;
  constexpr bool SPEC = SGm::on;
  static_assert(!SPEC || (SGm::nwin == W && SGm::ww <= 15), "a specialised instance is built for its geometry's window count and a direct table of one-word keys");
  constexpr int S_WW = SGm::ww, S_CL = SGm::CL, S_L = SGm::L, S_MIND = SGm::min_dinuc;
  constexpr int S_WIN[CTX_MAX_W] = {SGm::win[0], SGm::win[1], SGm::win[2], SGm::win[3]};
  if constexpr (SPEC) {
    // (the host compared the geometry before it chose this instance: spec_geom_matches; k_match_t has the same guard)
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
  uint32_t* const s_sketch = s_dyn;
  for (uint32_t t = threadIdx.x; t < CONF_NM / 2; t += TILE) s_nm[t] = t <= (uint32_t)mp->max_len ? nmiss_tab[t] : (uint16_t)0;
  if (block_mode == 1)
    for (uint32_t t = threadIdx.x; t < (1u << MATCHG_SKETCH_BITS); t += TILE) s_sketch[t] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) tcount2[(n + WT - 1) / WT] = 0;
  __syncthreads();

  const uint32_t nwt = (n + WT - 1) / WT;
  const uint32_t gw = blockIdx.x * NWAVE + (threadIdx.x >> 6), nw = gridDim.x * NWAVE;
  const uint64_t region = stage_cap / nw, region0 = region * gw;
  const uint64_t sregion = spill_cap / nw, sregion_w = sregion * gw;
  uint64_t used = 0;      // tuples this wave has staged so far (wave-uniform)
  uint32_t maxspill = 0;  // largest spill a wave-tile of this wave needed
  uint32_t nvalid = 0, ncand = 0, ncmp = 0, novf = 0, nrep = 0;  // per lane: far below 2^32
  const uint32_t mmtol = (uint32_t)mp->mmtol;
  const bool apply = mp->apply_mmtol != 0;

  // phase A of wave-tile wt, a lane per read: per window the length gate + CountDinuc >= MinDinuc
  // (cmd/muscato_window_reads/main.go:106-118 == cmd/muscato_screen/main.go:174-185) and the bucket of the window
  // key -> bb[] (WB_NONE when the window takes no part); returns the meta word
  auto phase_a = [&](uint32_t wt, const Rec<RW>& rec, uint32_t (&bb)[W]) __attribute__((always_inline)) -> uint32_t {
    const uint32_t lane = opaque(threadIdx.x) & 63;
    const bool active = wt * WT + lane < n;
    const int len = (int)rec.len();
    uint32_t valid = 0;
    if (ww <= 16 && direct) {
      const uint32_t kmask = ww == 16 ? 0xFFFFFFFFu : ((1u << (2 * ww)) - 1u);
#pragma unroll
      for (int k = 0; k < W; k++) {
        const uint32_t q1 = (uint32_t)win[k], q2 = q1 + (uint32_t)ww;
        const uint32_t key = (uint32_t)rec.ext(2 * q1) & kmask;
        bool pt = active && (uint32_t)len >= q2;
        if (min_dinuc > 0) pt = pt && key_dinucs16(key, ww) >= min_dinuc;
        bb[k] = pt ? __brev(key) >> (32 - 2 * ww) : WB_NONE;
        valid |= pt ? 1u << k : 0u;
      }
    } else {
#pragma unroll
      for (int k = 0; k < W; k++) {
        uint32_t b = WB_NONE;
        const uint32_t q1 = (uint32_t)win[k], q2 = q1 + (uint32_t)ww;
        if (active) {
          bool pt = (uint32_t)len >= q2;
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
    const uint32_t budget0 = (uint32_t)len < CONF_NM / 2 ? s_nm[len] : 0u;
    return (uint32_t)len | ((budget0 > 127u ? 127u : budget0) << 17) | (valid << 24);
  };
  // the LDS-DMA of one window's 64 lines into this wave's line buffer: bbk = this lane's read's bucket for that window;
  // the eight lanes of a line get it from the lane that owns the read (ds_bpermute), all eight before the first
  // address is formed.  Always eight instructions: a probe that takes no part fetches bucket 0 and its owner
  // ignores the line (the waits count instructions).
  auto issue_window = [&](uint32_t bbk) __attribute__((always_inline)) {
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint32_t bq[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bq[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((i * 8 + (lane >> 3)) * 4), (int)bbk);
    const uint4* pp[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      // the chunk: (lane & 7) ^ sw(p) for line p = 8 i + (lane >> 3), sw(p) = ((p >> 1) & 7) ^ (p & 1); (p >> 1) & 7 =
      // (lane >> 4) | ((i & 1) << 2) and p & 1 = (lane >> 3) & 1: one lane constant, bit 2 flipped on the odd instructions
      const uint32_t c = ((lane & 7u) ^ (lane >> 4) ^ ((lane >> 3) & 1u)) ^ (((uint32_t)i & 1u) << 2);
      pp[i] = reinterpret_cast<const uint4*>(T + (bq[i] != WB_NONE ? bq[i] : 0u)) + c;
    }
    glds16x8(pp[0], pp[1], pp[2], pp[3], pp[4], pp[5], pp[6], pp[7], lds_addr(&s_line[wid][0]));
  };

  // What a comparison through window k reads of the host's tables (k_match_t's WinTab)
  struct WinTab {
    uint32_t lm[NW];
    uint32_t need;
    uint32_t wm[W][NW];
  };
  auto win_tab = [&](uint32_t ul, int k, uint32_t sh, uint32_t len, WinTab& tb) __attribute__((always_inline)) {
    if constexpr (SPEC) {
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
      for (int j = 0; j < NW; j++) tb.wm[kk][j] = kk <= k ? (uint32_t)__builtin_amdgcn_readfirstlane((int)row[j]) : 0u;
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
  // one entry (context words c, c[7] still carrying the distance to the target end in its high half) against a read
  // whose image for window k is img: the fit rules (ctx_fit), cdiff (cmd/muscato_confirm/main.go:151-159, 205-211)
  // and, from the same mismatch mask, which windows of the read match the target exactly here (first-window rule).
  // Returns the pair's result word (NX_REJECT, or nmiss | NX_DUP | NX_ACC0 | window << 20 | slot << 24).
  auto score = [&](bool live, int k, int q1, const uint32_t (&img)[NW], const WinTab& tb, uint32_t jx, const uint32_t (&c)[NW], int len,
                   uint32_t budget, uint32_t valid, uint32_t slot) __attribute__((always_inline)) -> uint32_t {
    uint32_t z = 0;
    bool ok;
    const uint32_t rem = c[NW - 1] >> 16;
    if (__any(live && (jx == (uint32_t)q1 || jx == 0u))) ok = live & ctx_fit(jx, rem, q1, ww, len, &z);
    else ok = live & (jx >= (uint32_t)q1) & (len - q1 <= (int)rem);
    ncmp += ok ? 1u : 0u;
    uint32_t w = NX_REJECT;
    if (ok) {
      uint32_t d[NW], nx = 0;
#pragma unroll
      for (int j = 0; j < NW; j++) {
        // SPEC: a word no read of this record stride can reach through window k -- below the read's first bit 2 (CL - q1),
        // or at and beyond 2 (CL - q1) + 2 x 16 (RW - 1) -- is left out AT COMPILE TIME (a test of the run-time mask word
        // makes every word an exec-masked region of its own: kernels_match_lane.hpp)
        // (k and j are constants once the window lambda is instantiated and the loop unrolled)
        const int lo_k = SPEC ? 2 * (S_CL - S_WIN[k >= 0 && k < CTX_MAX_W ? k : 0]) : 0;
        const bool reached = !SPEC || (32 * (j + 1) > lo_k && 32 * j < lo_k + 32 * (RW - 1));
        const uint32_t x = img[j] ^ c[j];
        if constexpr (SPEC) d[j] = reached ? (x | (x >> 1)) & tb.lm[j] : 0u;
        else d[j] = base_diff(x, tb.lm[j]);
        if (reached) nx = bcnt_add(d[j], nx);
      }
      uint32_t exact = valid & (z ? ~q1zero : 0xFFFFFFFFu);
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        if (kk > k || !((tb.need >> kk) & 1u)) continue;  // wave-uniform
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < NW; j++) {
          if constexpr (SPEC) acc |= d[j] & tb.wm[kk][j];
          else acc = and_or_s(d[j], tb.wm[kk][j], acc);
        }
        if (acc) exact &= ~(1u << kk);
      }
      if (nx <= budget && ((exact >> k) & 1u)) {
        const bool first = (uint32_t)(__ffs(exact) - 1) == (uint32_t)k;
        w = (first ? nx : (nx | NX_DUP)) | NX_ACC0 | ((uint32_t)k << 20) | (slot << 24);
      }
    }
    return w;
  };

  // The tuples the PREVIOUS batch's launch staged (pstage != nullptr: same grid, same regions, the other stage buffer)
  // move to their final place in `hits` from inside this launch (the protocol of match_ctx_pass, as in k_match_t): a
  // wave-tile's first 64 ride in F3's flight through the buffer, the rest (rare) by plain loads behind its wait
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
  // ---- state that crosses a wave-tile boundary: the record, buckets and meta word of the wave-tile in hand
  Rec<RW> rec_cur;
  rec_cur.zero();
  uint32_t meta_cur = 0, bb_cur[W];
#pragma unroll
  for (int k = 0; k < W; k++) bb_cur[k] = WB_NONE;
  if (gw < nwt) {
    const uint32_t i = gw * WT + (opaque(threadIdx.x) & 63);
    rec_cur.load(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW, RW);  // (the one plain load of the loop-free part: nothing is in flight yet)
    meta_cur = phase_a(gw, rec_cur, bb_cur);
  }
#ifdef MUSC_LANE_PROF
  unsigned long long pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime(), pstart = pt0;
  uint32_t pf_tiles = 0;
#define PF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pf[i] += t_ - pt0; pt0 = t_; }
#else
#define PF(i)
#endif
  for (uint32_t wt = gw; wt < nwt; wt += nw) {
    const bool have_next = wt + nw < nwt;
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint4* const line_l = s_line[wid];
    uint32_t* const best_l = reinterpret_cast<uint32_t*>(line_l + 384);
    uint32_t* const cnt_l = reinterpret_cast<uint32_t*>(line_l + 400);
    uint32_t* const base_l = reinterpret_cast<uint32_t*>(line_l + 416);
    uint32_t* const oix_l = reinterpret_cast<uint32_t*>(line_l + 432);
    uint8_t* const own_l = reinterpret_cast<uint8_t*>(line_l + 448);
    uint3* const list_l = s_list[wid];

    uint32_t nlist = 0;           // reported candidates so far (wave-uniform)
    uint32_t best = 0xFFFFFFFFu;  // smallest mismatch count reported for this lane's read by the in-lane comparisons
    uint32_t wc[W], oc[W], ovf[W];
#pragma unroll
    for (int k = 0; k < W; k++) wc[k] = oc[k] = ovf[k] = 0;
    const bool active = wt * WT + lane < n;
    const int rlen = (int)REC_LEN(meta_cur);
    const uint32_t budget = REC_BUDGET(meta_cur), valid_cur = REC_VALID(meta_cur);
    // every read of the wave-tile of one length: the comparisons use scalar length masks
    const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(rlen);
    const uint32_t ulen = __ballot(active && (uint32_t)rlen != len0) == 0 ? len0 : 0xFFFFFFFFu;

    // a reported candidate: appended in lane order
    auto append = [&](bool rep, uint32_t w, uint32_t gene, uint32_t pos) __attribute__((always_inline)) -> bool {
      const unsigned long long vote = __ballot(rep);
      if (vote == 0) return false;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      const uint32_t slot = nlist + below;
      nlist += (uint32_t)__popcll(vote);
      if (!rep) return false;
      if (slot < WLIST) {
        list_l[slot] = make_uint3(w, gene, pos);
      } else if (slot - WLIST < sregion) {
        spill[sregion_w + (slot - WLIST)] = make_uint4(w, gene, pos, 0u);
      }
      return true;
    };
    // ... of the lane's own read (in-lane comparisons): the first two stay in the lane's registers (k0 / k1: result word,
    // gene, position) -- nine tenths of the candidates of a wave-tile never see the list, nor phase D its LDS round trips
    // for them -- a third and later ones go to the list like the overflow entries' (their best joins best_l there)
    uint32_t kc = 0, k0w = NX_REJECT, k0g = 0, k0p = 0, k1w = NX_REJECT, k1g = 0, k1p = 0;
    auto report_own = [&](uint32_t w, uint32_t gene, uint32_t pos, uint32_t& wck) __attribute__((always_inline)) {
      const bool acc = w != NX_REJECT;
      wck += acc ? 1u : 0u;
      const bool rep = acc && !(w & NX_DUP);
      const bool first = rep && kc == 0, second = rep && kc == 1, later = rep && kc >= 2;
      k0w = first ? w : k0w; k0g = first ? gene : k0g; k0p = first ? pos : k0p;
      k1w = second ? w : k1w; k1g = second ? gene : k1g; k1p = second ? pos : k1p;
      kc += rep ? 1u : 0u;
      const uint32_t v = w & 0xFFFFu;
      best = rep && v < best ? v : best;
      if (__any(later)) append(later, w, gene, pos);  // (its nmiss is in `best` already: phase D merges the lane's best into best_l)
    };
    const uint32_t rb = lane * 8u + (((lane >> 1) & 7u) ^ (lane & 1u));
    // the part of a window that needs its lines: the header, then the three inline entries, in this lane
    auto slots = [&](auto kc, const uint32_t (&img)[NW], const WinTab& tb) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      const int q1 = win[k];
      const uint4 h0 = line_l[rb], h1 = line_l[rb ^ 1u];
      const uint32_t cnt = bb_cur[k] != WB_NONE ? h0.x : 0u;  // (a probe that takes no part fetched bucket 0)
      ncand += cnt;
      oc[k] = cnt > (uint32_t)NIN ? cnt - (uint32_t)NIN : 0u;
      ovf[k] = h0.y;
      novf += oc[k];
      uint4 ca = line_l[rb ^ 2u], cb = line_l[rb ^ 3u];
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
        const uint32_t c[NW] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
        const uint32_t w = score(live, k, q1, img, tb, jx, c, rlen, budget, valid_cur, lane);
        report_own(w, gene, jx - (uint32_t)q1, wc[k]);
        ca = na;
        cb = nb;
      }
    };
    // ===================================================================== F1, F2: the two windows
    {
      wave_lds_sync();  // (everything this wave has read from or written to its buffer is done)
      issue_window(bb_cur[0]);
      PF(0)
      const uint32_t sh0 = 2u * (uint32_t)(CL - win[0]);
      WinTab tb;
      win_tab(ulen, 0, sh0, (uint32_t)rlen, tb);
      uint32_t img[NW];
      read_image_n<RW, NW>(rec_cur, sh0, img);
      PF(1)
      wait_vm0();
      PF(2)
      slots(std::integral_constant<int, 0>{}, img, tb);
      PF(3)
    }
    {
      wave_lds_sync();
      issue_window(bb_cur[1]);
      PF(4)
      const uint32_t sh1 = 2u * (uint32_t)(CL - win[1]);
      WinTab tb;
      win_tab(ulen, 1, sh1, (uint32_t)rlen, tb);
      uint32_t img[NW];
      read_image_n<RW, NW>(rec_cur, sh1, img);
      wait_vm0();
      PF(5)
      slots(std::integral_constant<int, 1>{}, img, tb);
      PF(6)
    }
    // ===================================================================== F3: overflow entries, the next records
    // the entries beyond the buckets' third: a lane per entry, listed window-major, lane-major
    uint32_t pre[W], total = 0;
#pragma unroll
    for (int kk = 0; kk < W; kk++) {
      const uint32_t inc = wave_scan_incl(oc[kk]);
      pre[kk] = total + inc - oc[kk];
      total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    // items [c0, c0 + 64): item -> (window, read slot) and its place in E, from the lanes that own the probes
    auto owner_tables = [&](uint32_t c0) __attribute__((always_inline)) {
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        const uint32_t e_lo = c0 > pre[kk] ? c0 - pre[kk] : 0u;
        const uint32_t e_hi = pre[kk] + oc[kk] > c0 + WT ? (c0 + WT > pre[kk] ? c0 + WT - pre[kk] : 0u) : oc[kk];
#pragma unroll 1
        for (uint32_t e = e_lo; e < e_hi; e++) {
          own_l[pre[kk] + e - c0] = (uint8_t)(kk * WT + lane);
          oix_l[pre[kk] + e - c0] = ovf[kk] + e;
        }
      }
      wave_lds_sync();
    };
    // three LDS-DMA instructions: the entries of the lanes' items -> planes 0 .. 2 of the buffer (entry 0 for a lane without an item)
    auto issue_entries = [&](uint32_t eix) __attribute__((always_inline)) {
      const uint32_t* __restrict__ pe = reinterpret_cast<const uint32_t*>(E) + ctx_entry_word<false>(eix);
      const uint32_t zb = lds_addr(line_l);
      glds16(pe, zb);
      glds16(pe + 4, zb + 1024u);
      glds16(pe + 8, zb + 2048u);
    };
    // The comparison pass over overflow entries, a lane per entry: the entry's words (gene, jx, ctx[8]) from the buffer,
    // the probe it belongs to (window, read slot) and that read's meta word and record from the lane that owns the read
    // (ds_bpermute).  ONE pass with the window per lane (k_match_t's form for two windows; a hashed table needs the
    // lane's own window exact as well: the key is not the bucket).  c0 = the first item of the round.
    auto entry_round = [&](uint32_t c0) __attribute__((always_inline)) {
      const uint32_t n_items = total - c0 < 64u ? total - c0 : 64u;
      const bool have = lane < n_items;
      const uint32_t ks = have ? (uint32_t)own_l[lane] : 0u;
      const uint32_t seg = ks & 63u;
      const uint4 e0 = line_l[lane], e1 = line_l[64 + lane], e2 = line_l[128 + lane];
      const uint32_t meta = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(seg * 4u), (int)meta_cur);
      Rec<RW> rec;
#pragma unroll
      for (int q = 0; q < RW; q++) rec.w[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(seg * 4u), (int)rec_cur.w[q]);
      const int len = (int)REC_LEN(meta);
      const uint32_t gene = e0.x, jx = e0.y;
      const uint32_t c[NW] = {e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y};
      const bool k1 = have && (ks >> 6) != 0;
      const int q1a = win[0], q1b = win[1];
      const uint32_t sha = 2u * (uint32_t)(CL - q1a), shb = 2u * (uint32_t)(CL - q1b);
      const uint32_t sh = k1 ? shb : sha;
      const int q1 = k1 ? q1b : q1a;
      uint32_t lm[NW];
      if (SPEC && ulen == (uint32_t)S_L) {
        // (the geometry's masks are constants: no table rows from scalar memory in front of the pass)
#pragma unroll
        for (int j = 0; j < NW; j++)
          lm[j] = k1 ? 0x55555555u & bit_range_mask(2 * (S_CL - S_WIN[1]) - 32 * j, 2 * (S_CL - S_WIN[1]) + 2 * S_L - 32 * j)
                     : 0x55555555u & bit_range_mask(2 * (S_CL - S_WIN[0]) - 32 * j, 2 * (S_CL - S_WIN[0]) + 2 * S_L - 32 * j);
      } else if (ulen != 0xFFFFFFFFu) {
        const uint32_t (*rows)[CTXW_WORDS] = mp->lm[__builtin_amdgcn_readfirstlane((int)ulen)];
#pragma unroll
        for (int j = 0; j < NW; j++) lm[j] = k1 ? rows[1][j] : rows[0][j];
      } else {
#pragma unroll
        for (int j = 0; j < NW; j++) lm[j] = 0x55555555u & bit_range_mask((int)sh - 32 * j, (int)sh + 2 * len - 32 * j);
      }
      uint32_t z = 0;
      bool ok;
      const uint32_t rem = c[NW - 1] >> 16;
      if (__any(have && (jx == (uint32_t)q1 || jx == 0u))) ok = have & ctx_fit(jx, rem, q1, ww, len, &z);
      else ok = have & (jx >= (uint32_t)q1) & (len - q1 <= (int)rem);
      ncmp += ok ? 1u : 0u;
      uint32_t w = NX_REJECT;
      if (ok) {
        uint32_t ia[NW], ib[NW];
        read_image_n<RW, NW>(rec, sha, ia);
        read_image_n<RW, NW>(rec, shb, ib);
        uint32_t d[NW], nx = 0, acc0 = 0, acc1 = 0;
        // window 0 of the read in the coordinates of a comparison through window 1, and (hashed tables) the lane's own window
        const uint32_t* __restrict__ row10 = mp->wm[1][0];
        const uint32_t* __restrict__ row00 = mp->wm[0][0];
        const uint32_t* __restrict__ row11 = mp->wm[1][1];
#pragma unroll
        for (int j = 0; j < NW; j++) {
          d[j] = base_diff((k1 ? ib[j] : ia[j]) ^ c[j], lm[j]);
          nx = bcnt_add(d[j], nx);
          if constexpr (SPEC) {
            // window 0 of the read in the coordinates of a comparison through window 1: a constant (words it misses fold away)
            constexpr int sh1c = 2 * (S_CL - S_WIN[1]);
            acc0 |= d[j] & bit_range_mask(sh1c + 2 * S_WIN[0] - 32 * j, sh1c + 2 * (S_WIN[0] + S_WW) - 32 * j);
          } else {
            acc0 = and_or_s(d[j], row10[j], acc0);
            if (!direct) acc1 |= d[j] & (k1 ? row11[j] : row00[j]);
          }
        }
        uint32_t exact = REC_VALID(meta) & (z ? ~q1zero : 0xFFFFFFFFu);
        if (k1 && acc0) exact &= ~1u;
        const uint32_t kbit = k1 ? 2u : 1u;
        if (!direct && acc1) exact &= ~kbit;
        if (nx <= REC_BUDGET(meta) && (exact & kbit)) {
          const bool first = (exact & (kbit - 1u)) == 0;
          w = (first ? nx : (nx | NX_DUP)) | NX_ACC0 | (k1 ? 1u << 20 : 0u) | (seg << 24);
        }
      }
      // the accepted pairs go to their probes' MaxMatches counters: a probe's items are consecutive lanes, so its
      // owner counts the votes in its range
      if (block_mode) {
        const unsigned long long acc_vote = __ballot(w != NX_REJECT);
#pragma unroll
        for (int kk = 0; kk < W; kk++) {
          const uint32_t lo = pre[kk] > c0 ? pre[kk] - c0 : 0u;
          const uint32_t hi_abs = pre[kk] + oc[kk];
          const uint32_t hi = hi_abs > c0 ? (hi_abs - c0 < 64u ? hi_abs - c0 : 64u) : 0u;
          if (hi > lo && lo < 64u) {
            const unsigned long long m = (hi >= 64u ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
            wc[kk] += (uint32_t)__popcll(acc_vote & m);
          }
        }
      }
      if (append(w != NX_REJECT && !(w & NX_DUP), w, gene, jx - (uint32_t)q1)) atomicMin(&best_l[seg], w & 0xFFFFu);
    };
    Rec<RW> rec_nx;
    rec_nx.zero();
    {
      wave_lds_sync();  // (every lane has read its line of window 1)
      best_l[lane] = 0xFFFFFFFFu;
      uint32_t eix = 0;
      if (total) {
        owner_tables(0);
        eix = lane < total ? oix_l[lane] : 0u;
        wave_lds_sync();
        issue_entries(eix);
      }
      if (have_next) {
        // the records of the next wave-tile, RW / 4 instructions of 1 KB contiguous (chunks of reads past the batch's end
        // come from read 0)
        const uint32_t zb = lds_addr(line_l + 256);
#pragma unroll
        for (int q = 0; q < RW / 4; q++) {
          const uint32_t g = (uint32_t)q * 64u + lane;
          const uint32_t i = (wt + nw) * WT + g / (uint32_t)(RW / 4);
          glds16(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW + 4u * (g % (uint32_t)(RW / 4)), zb + (uint32_t)q * 1024u);
        }
      }
      uint32_t cpm = 0, cpd = 0;
      if (pcopy && wt < pnwt) {
        const uint32_t wtu = (uint32_t)__builtin_amdgcn_readfirstlane((int)wt);
        cpm = ptcount2[wtu];
        cpd = ptpre[wtu];
        if (cpm) glds16(pstage + region0 + pused + (lane < cpm ? lane : 0u), lds_addr(line_l + 192));
      }
      PF(7)
      if (total || have_next || cpm) wait_vm0();
      PF(8)
      if (cpm) {
        uint4* __restrict__ dst = hits + pbase + cpd;
        if (lane < cpm) dst[lane] = line_l[192 + lane];
        for (uint32_t i = 64 + lane; i < cpm; i += 64) dst[i] = pstage[region0 + pused + i];  // a tile with more than 64 tuples
        pused += cpm;
      }
      if (have_next) {
        const uint4* src = line_l + 256 + lane * (RW / 4);
#pragma unroll
        for (int q = 0; q < RW / 4; q++) {
          const uint4 v = src[q];
          rec_nx.w[4 * q] = v.x; rec_nx.w[4 * q + 1] = v.y; rec_nx.w[4 * q + 2] = v.z; rec_nx.w[4 * q + 3] = v.w;
        }
      }
      if (total) {
        entry_round(0);
        // beyond the first 64 (families of near-identical targets, low-complexity keys): listed, fetched and compared on
        // the spot, 64 at a time -- every round a memory round trip of its own
        for (uint32_t c0 = 64; c0 < total; c0 += 64) {
          wave_lds_sync();
          owner_tables(c0);
          const uint32_t e2 = lane < total - c0 ? oix_l[lane] : 0u;
          wave_lds_sync();
          issue_entries(e2);
          wait_vm0();
          entry_round(c0);
        }
      }
      PF(9)
    }
    // ===================================================================== phase D: per-read selection, the tuples
    // best[read] = min nmiss over its reported pairs, tuples with nmiss <= best + MMTol per read
    // (cmd/muscato_combine_windows/main.go:36-60; all of them when apply_mmtol == 0), a wave scan over the 64 reads,
    // and the tuples, read-major, into the wave's region of `stage`: a read's own first two from its lane's registers,
    // then the list's (overflow entries, third and later own candidates)
    {
      const uint32_t nl = nlist;
      const uint32_t nspill = nl > WLIST ? nl - WLIST : 0u;
      const bool spill_ok = nspill <= sregion;
      if (nspill > maxspill) maxspill = nspill;
      if (nspill) wait_vm0();  // this wave's spilled candidates have landed
      if (block_mode) {
#pragma unroll
        for (int k = 0; k < W; k++) {
          const uint32_t cw = wc[k];
          if (!cw) continue;
          const uint32_t h = block_hash32((uint32_t)k, bb_cur[k]);
          if (block_mode == 1) atomicAdd(&s_sketch[h >> (32 - MATCHG_SKETCH_BITS)], cw);
          else atomicAdd(&block_table[h >> (32 - BLOCK_TABLE_BITS)], cw);
        }
      }
      // the read's best: the lane's own (every own candidate, listed or not, went into `best`) joins what the list's
      // candidates of OTHER lanes (overflow entries) reported for it
      const uint32_t b0 = total ? best_l[lane] : 0xFFFFFFFFu;  // (only the overflow entries' pass writes it)
      const uint32_t bestr = best < b0 ? best : b0;
      const uint32_t thr_own = apply ? bestr + mmtol : 0xFFFFu;
      const bool s0 = kc > 0 && (k0w & 0xFFFFu) <= thr_own, s1 = kc > 1 && (k1w & 0xFFFFu) <= thr_own;
      const uint32_t nown = (s0 ? 1u : 0u) + (s1 ? 1u : 0u);
      uint32_t cnum = nown;
      uint32_t kw = NX_REJECT, kg = 0, kp = 0, ko = 0;
      const uint32_t nuse = spill_ok ? nl : (nl < WLIST ? nl : WLIST);
      auto item = [&](uint32_t j, uint32_t* gene, uint32_t* pos) __attribute__((always_inline)) -> uint32_t {
        if (j < WLIST) {
          const uint3 it = list_l[j];
          *gene = it.y;
          *pos = it.z;
          return it.x;
        }
        // written by other lanes of this wave a moment ago: read past the L1
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(spill + sregion_w + (j - WLIST));
        *gene = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *pos = __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      if (nuse) {  // (wave-uniform: most wave-tiles of a sparse database have no listed candidate at all)
        best_l[lane] = bestr;
        cnt_l[lane] = 0;
        wave_lds_sync();
        // the first round of listed candidates (a lane each) stays in registers from the count to the store: the counting
        // atomic also hands out the tuple's place among its read's listed ones; further rounds are walked twice
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
      const uint32_t tot = __builtin_amdgcn_readlane(inc, 63);
      const uint32_t mybase = inc - cnum;
      const uint64_t base = region0 + used;
      const bool fits = spill_ok && used + tot <= region;
      if (lane == 0) {
        tbase[wt] = (uint32_t)base;
        tcount2[wt] = fits ? tot : 0u;
      }
      if (fits && tot) {
        const uint32_t rid = (uint32_t)(r0 + wt * WT + lane);
        if (s0) stage[base + mybase] = make_uint4(rid, k0g, k0p, k0w & 0xFFFFu);
        if (s1) stage[base + mybase + (s0 ? 1u : 0u)] = make_uint4(rid, k1g, k1p, k1w & 0xFFFFu);
        if (nuse) {
          // a listed tuple's place: behind its read's own ones (base_l = where the read's listed tuples start)
          base_l[lane] = mybase + nown;
          wave_lds_sync();
          if (kw != NX_REJECT) {
            const uint32_t rl = kw >> 24;
            stage[base + base_l[rl] + ko] = make_uint4((uint32_t)(r0 + wt * WT + rl), kg, kp, kw & 0xFFFFu);
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
              stage[base + base_l[rl] + ord] = make_uint4((uint32_t)(r0 + wt * WT + rl), g, p, v);
            }
          }
        }
      }
      used += tot;
      nrep += kc < 2u ? kc : 2u;  // (the listed ones below, from one lane)
      nrep += lane == 0 ? nl : 0u;
      PF(10)
    }
    // ---- the next wave-tile's phase A; its record, buckets and meta word become the ones in hand
    if (have_next) {
      meta_cur = phase_a(wt + nw, rec_nx, bb_cur);
      rec_cur = rec_nx;
    }
    PF(11)
#ifdef MUSC_LANE_PROF
    pf_tiles++;
#endif
  }
#ifdef MUSC_LANE_PROF
  if ((gw == 0 || gw == 1001) && (threadIdx.x & 63) == 0 && pf_tiles > 4)
    printf("wave %u: %u tiles, cycles/tile: total %llu | F1 issue %llu image0 %llu wait0 %llu slots0 %llu | F2 issue %llu image1+wait1 %llu slots1 %llu | F3 list+issue %llu wait %llu entries %llu | phase D %llu phase A next %llu\n",
           gw, pf_tiles, (__builtin_amdgcn_s_memtime() - pstart) / pf_tiles, pf[0] / pf_tiles, pf[1] / pf_tiles, pf[2] / pf_tiles, pf[3] / pf_tiles,
           pf[4] / pf_tiles, pf[5] / pf_tiles, pf[6] / pf_tiles, pf[7] / pf_tiles, pf[8] / pf_tiles, pf[9] / pf_tiles, pf[10] / pf_tiles, pf[11] / pf_tiles);
#endif
  if (pcopy) {
    const uint32_t mine = gw < nwt ? (nwt - gw + nw - 1) / nw : 0u;  // wave-tiles of this batch this wave has walked
    for (uint32_t wt = gw + mine * nw; wt < pnwt; wt += nw) {
      const uint32_t wtu = (uint32_t)__builtin_amdgcn_readfirstlane((int)wt);
      const uint32_t m = ptcount2[wtu], d = ptpre[wtu];
      const uint32_t lane = opaque(threadIdx.x) & 63;
      for (uint32_t i = lane; i < m; i += 64) hits[pbase + d + i] = pstage[region0 + pused + i];
      pused += m;
    }
  }
  // one reduction per workgroup and a handful of atomics from its first thread (as in k_match_t)
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
    for (uint32_t t = threadIdx.x; t < (1u << MATCHG_SKETCH_BITS); t += TILE) hot |= s_sketch[t] >= block_thr;
    if (__any(hot) && (threadIdx.x & 63) == 0) atomicOr(&counters[6], 1ull);
  }
}
