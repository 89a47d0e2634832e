// k_match_g -- the fused screen + confirm + select kernel at THREE waves per SIMD (round 4; device code, compiled in
// match_dma_rw8.hip, declared in kernels_match_lane_inst.hpp; bucket layout, parameter block and fit rules of
// kernels_match.hpp; the comparison itself is k_match_t's: in the lane that owns the read).
//
// k_match_t (kernels_match_lane.hpp) lands a window's 64 bucket lines in 32 registers, writes them to a line buffer,
// keeps two generations of records, candidate lists and per-read tables in LDS and carries a wave-tile's overflow
// entries in registers across the tile boundary: 211-225 VGPRs and 16 KB of LDS per wave = two waves per SIMD, and
// r03's counters said what that costs: 43 % of a wave's cycles execute, 39 % wait, with ONE other wave to cover them.
// Here everything that comes from memory travels by LDS-DMA (global_load_lds_dwordx4: the data goes from the L2 to
// LDS, no destination register, no ds_write pass):
//   * the 64 lines of a window: eight instructions of EIGHT WHOLE LINES each (eight lanes per line, 16 contiguous
//     bytes per lane; the per-lane SOURCE address picks the chunk that belongs at the lane's place in the swizzled
//     line buffer -- chunk c of line p at slot c ^ ((p >> 1) & 7) ^ (p & 1) -- so lane p reads line p without bank
//     conflicts, exactly as in k_match_t);
//   * the overflow entries of the wave-tile before (beyond a bucket's third, 40 bytes each): three 16-byte pieces per
//     entry into a landing zone Z of 48 entries (the pieces' tails belong to the entry's line: ctx_entry_word);
//   * the NEXT wave-tile's records: 64 * RW words in a row = RW / 4 instructions of 1 KB, into the same zone.
// ONE line buffer per wave (8 KB), ONE generation of candidate list / best / meta, no records in LDS (a lane keeps
// its read's record in registers; an overflow entry's lane gets its read's record by ds_bpermute when it is listed):
// 11.8 KB of LDS per wave and ~130-160 VGPRs = three workgroups of four waves per CU.
//
// The schedule of a wave-tile t (two windows; "flight" = LDS-DMA issued ... s_waitcnt):
//   F1  issue [overflow entries of t-1 -> Z] then [lines of window 0 of t -> line buffer]
//       s_waitcnt vmcnt(8): the entries are there, the eight line instructions still fly --
//         compare the overflow entries of t-1 (a lane per entry), phase D of t-1 (per-read selection, tuples)
//         the read's image for window 0 of t
//       s_waitcnt vmcnt(0): header + the three inline entries of window 0, in the lane that owns the read
//   F2  issue [lines of window 1 of t -> line buffer] and [records of t+1 -> Z]
//         the read's image for window 1
//       s_waitcnt vmcnt(0): records of t+1 -> registers, its phase A (window gates, buckets);
//       header + inline entries of window 1; the overflow entries of t are listed (item tables -> Z)
// A wave exposes most of two memory round trips per wave-tile -- by design: the third wave of its SIMD (and the
// fourth ... twelfth of its CU) is what covers them, and with twelve waves per CU the memory system stays full.
// All vector-memory traffic of the loop is LDS-DMA or a store (the compiler's waitcnt insertion cannot see an
// LDS-DMA; every wait for one is written out here, and nothing the compiler waits for with vmcnt(0) -- it always
// drains -- is in flight while a line flight is, except on the rare paths that say so).
//
// Built for two windows on 120-base buckets, no X on either side (BASELINE configs 2-4); every other run keeps
// k_match_t.  The tuples of a batch are moved into `hits` by k_compact_w (no in-launch move of the previous batch).
#pragma once
#include "kernels_match_lane_inst.hpp"

#ifndef MATCHG_ZITEMS
#define MATCHG_ZITEMS 48  // overflow entries per landing round (48 bytes each)
#endif

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
DEV void wait_vm8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
// the byte address of a __shared__ object within the workgroup's LDS, wave-uniform (what M0 wants)
template <class T>
DEV uint32_t lds_addr(const T* p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p); }

template <int RW, int SG>
__global__ __launch_bounds__(TILE, MATCHG_WAVES) void k_match_g(const uint32_t* __restrict__ rd, uint64_t r0, uint32_t n,
                                                                const MatchParams* __restrict__ mp,
                                                                const uint16_t* __restrict__ nmiss_tab,
                                                                const CtxBucket* __restrict__ T, const CtxEntry* __restrict__ E,
                                                                uint4* __restrict__ stage, uint64_t stage_cap,
                                                                uint4* __restrict__ spill, uint64_t spill_cap,
                                                                uint32_t* __restrict__ tbase, uint32_t* __restrict__ tcount2,
                                                                int block_mode, uint32_t block_thr,
                                                                uint32_t* __restrict__ block_table,
                                                                unsigned long long* __restrict__ counters) {
  constexpr int W = 2, NW = 8, NIN = CTX_INLINE;
  constexpr int NWAVE = TILE / 64;
  constexpr uint32_t WLIST = (uint32_t)MATCHG_WLIST, ZI = (uint32_t)MATCHG_ZITEMS;
  static_assert(RW % 4 == 0 && RW / 4 * 64 <= 3 * MATCHG_ZITEMS, "the next wave-tile's records land in the overflow entries' zone");
  static_assert(MATCHG_ZITEMS % 4 == 0 && MATCHG_ZITEMS <= 64 && 3 * MATCHG_ZITEMS >= 32 + 16 + 4, "the zone (3 ZI uint4) also holds phase D's cnt / base (32 uint4) and the item tables (16 + 4)");
  extern __shared__ uint32_t s_dyn[];                      // block_mode == 1: the MaxMatches sketch of the workgroup
  __shared__ uint4 s_line[NWAVE][64 * 8];                  // the window's 64 bucket lines, swizzled
  // the landing zone, by phase: overflow entries (three planes of ZI x 16 bytes) | the next wave-tile's records |
  // phase D's cnt[64], base[64] | the item tables of the overflow entries just listed (s_oix at uint4 32.., s_own at 48..)
  __shared__ uint4 s_z[NWAVE][3 * MATCHG_ZITEMS];
  __shared__ uint3 s_list[NWAVE][WLIST];                   // reported candidates: result word, gene, position
  __shared__ uint32_t s_best[NWAVE][WT];                   // smallest mismatch count the overflow pass reported per read
  __shared__ uint32_t s_meta[NWAVE][WT];                   // length | budget << 17 | valid windows << 24
  __shared__ uint16_t s_nm[CONF_NM];

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
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)mp->max_len ? nmiss_tab[t] : (uint16_t)0;
  if (block_mode == 1)
    for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) s_sketch[t] = 0;
  s_best[threadIdx.x >> 6][threadIdx.x & 63] = 0xFFFFFFFFu;
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
    const uint32_t budget0 = (uint32_t)len < CONF_NM ? s_nm[len] : 0u;  // (reads on this path are at most 120 bases)
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
      const uint32_t pl = (uint32_t)i * 8u + (lane >> 3);
      const uint32_t c = (lane & 7u) ^ (((pl >> 1) & 7u) ^ (pl & 1u));
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
        const uint32_t x = img[j] ^ c[j];
        if constexpr (SPEC) d[j] = (x | (x >> 1)) & tb.lm[j];
        else d[j] = base_diff(x, tb.lm[j]);
        if (!SPEC || tb.lm[j] != 0u) nx = bcnt_add(d[j], nx);
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

  // ---- state: the wave-tile in hand ("cur": record, buckets, meta word in registers), the next one (filled by F2)
  // and the one before ("prev": its first ZI overflow entries are listed in the zone, each with its read's record
  // and meta word in its lane's registers; its candidates wait in s_list / s_best for its phase D in cur's F1)
  Rec<RW> rec_cur;
  rec_cur.zero();
  uint32_t meta_cur = 0, bb_cur[W];
#pragma unroll
  for (int k = 0; k < W; k++) bb_cur[k] = WB_NONE;
  uint32_t wt_prev = 0, nlist_prev = 0, best_prev = 0xFFFFFFFFu, ulen_prev = 0xFFFFFFFFu, total_prev = 0;
  uint32_t wc_prev[W], bb_prev[W], oc_prev[W], ovf_prev[W];
#pragma unroll
  for (int k = 0; k < W; k++) wc_prev[k] = oc_prev[k] = ovf_prev[k] = 0, bb_prev[k] = WB_NONE;
  uint32_t o_n = 0, o_ks = 0, o_meta = 0, o_eix = 0;  // this lane's item: window << 6 | read slot, the read's meta word, its entry in E
  Rec<RW> o_rec;
  o_rec.zero();

  if (gw < nwt) {
    const uint32_t i = gw * WT + (opaque(threadIdx.x) & 63);
    rec_cur.load(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW, RW);  // (the one plain load of the kernel: nothing is in flight yet)
    meta_cur = phase_a(gw, rec_cur, bb_cur);
  }
  uint32_t wt = gw;
  bool have_prev = false;
#ifdef MUSC_LANE_PROF
  unsigned long long pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime(), pstart = pt0;
  uint32_t pf_tiles = 0;
#define PF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pf[i] += t_ - pt0; pt0 = t_; }
#else
#define PF(i)
#endif
  while (wt < nwt || have_prev) {
    const bool have_cur = wt < nwt;
    const bool have_next = wt + nw < nwt;
    const uint32_t tid = opaque(threadIdx.x);
    const uint32_t lane = tid & 63, wid = tid >> 6;
    uint4* const line_l = s_line[wid];
    uint4* const z_l = s_z[wid];
    uint32_t* const cnt_l = reinterpret_cast<uint32_t*>(z_l);          // phase D: cnt[64] ...
    uint32_t* const base_l = reinterpret_cast<uint32_t*>(z_l) + WT;    // ... base[64]
    uint32_t* const oix_l = reinterpret_cast<uint32_t*>(z_l + 32);     // item -> its entry in E (64 words)
    uint8_t* const own_l = reinterpret_cast<uint8_t*>(z_l + 48);       // item -> window * 64 + read slot (64 bytes)
    uint3* const list_l = s_list[wid];
    uint32_t* const best_l = s_best[wid];

    // a reported candidate of any read of prev (overflow entries): prev's list / best
    auto report_any = [&](uint32_t w, uint32_t gene, uint32_t pos) __attribute__((always_inline)) {
      const bool rep = w != NX_REJECT && !(w & NX_DUP);
      const unsigned long long vote = __ballot(rep);
      if (vote == 0) return;
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
      const uint32_t slot = nlist_prev + below;
      nlist_prev += (uint32_t)__popcll(vote);
      if (!rep) return;
      atomicMin(&best_l[w >> 24], w & 0xFFFFu);
      if (slot < WLIST) {
        list_l[slot] = make_uint3(w, gene, pos);
      } else if (slot - WLIST < sregion) {
        spill[sregion_w + (slot - WLIST)] = make_uint4(w, gene, pos, 0u);
      }
    };
    // the accepted pairs among the items [c0, c0 + 64) go to their probes' MaxMatches counters: a probe's items
    // are consecutive lanes (the listing is window-major, lane-major), so its owner counts the votes in its range
    auto count_accepted = [&](unsigned long long acc_vote, uint32_t c0, const uint32_t (&pre)[W]) __attribute__((always_inline)) {
      if (!block_mode) return;
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        const uint32_t lo = pre[kk] > c0 ? pre[kk] - c0 : 0u;
        const uint32_t hi_abs = pre[kk] + oc_prev[kk];
        const uint32_t hi = hi_abs > c0 ? (hi_abs - c0 < 64u ? hi_abs - c0 : 64u) : 0u;
        if (hi > lo && lo < 64u) {
          const unsigned long long m = (hi >= 64u ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
          wc_prev[kk] += (uint32_t)__popcll(acc_vote & m);
        }
      }
    };
    // The comparison pass over overflow entries of prev, a lane per entry: the entry's words (gene, jx, ctx[8]) from
    // the zone, the probe it belongs to (window, read slot) and that read's meta word and record from the lane's
    // registers.  Two windows on 120-base buckets: ONE pass with the window per lane (k_match_t's form; a hashed
    // table needs the lane's own window exact as well: the key is not the bucket).
    auto entry_compare = [&](uint32_t n_items, uint32_t ks, uint32_t meta, const Rec<RW>& rec, const uint4& e0, const uint4& e1, const uint4& e2,
                             unsigned long long& acc_vote) __attribute__((always_inline)) {
      acc_vote = 0;
      const bool have = lane < n_items;
      if (!__any(have)) return;
      const int len = (int)REC_LEN(meta);
      const uint32_t gene = e0.x, jx = e0.y;
      const uint32_t c[NW] = {e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y};
      const uint32_t seg = ks & 63u;
      const bool k1 = have && (ks >> 6) != 0;
      const int q1a = win[0], q1b = win[1];
      const uint32_t sha = 2u * (uint32_t)(CL - q1a), shb = 2u * (uint32_t)(CL - q1b);
      const uint32_t sh = k1 ? shb : sha;
      const int q1 = k1 ? q1b : q1a;
      uint32_t lm[NW];
      if (ulen_prev != 0xFFFFFFFFu) {
        const uint32_t (*rows)[CTXW_WORDS] = mp->lm[__builtin_amdgcn_readfirstlane((int)ulen_prev)];
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
          acc0 = and_or_s(d[j], row10[j], acc0);
          if (!direct) acc1 |= d[j] & (k1 ? row11[j] : row00[j]);
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
      acc_vote = __ballot(w != NX_REJECT);
      report_any(w, gene, jx - (uint32_t)q1);
    };
    // the items [c0, c0 + 64) of prev: item -> (window, read slot) and its place in E, from the lanes that own the probes
    auto owner_tables = [&](uint32_t c0, const uint32_t (&oc)[W], const uint32_t (&ovf)[W], const uint32_t (&pre)[W]) __attribute__((always_inline)) {
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
    // three LDS-DMA instructions: the entries of the lanes' items -> the zone (lanes 0 .. ZI - 1; entry 0 for a lane without an item)
    auto issue_entries = [&](uint32_t eix) __attribute__((always_inline)) {
      if (lane < ZI) {
        const uint32_t* __restrict__ pe = reinterpret_cast<const uint32_t*>(E) + ctx_entry_word<false>(eix);
        const uint32_t zb = lds_addr(z_l);
        glds16(pe, zb);
        glds16(pe + 4, zb + 16u * ZI);
        glds16(pe + 8, zb + 32u * ZI);
      }
    };
    // prev's overflow entries beyond the first ZI (families of near-identical targets, low-complexity keys): listed,
    // fetched and compared on the spot, ZI at a time -- every round a memory round trip of its own, and the flight
    // of cur's window 0 ends with the first of them.  The reads' records come from global memory.
    auto overflow_rest_prev = [&](const uint32_t (&pre)[W]) __attribute__((always_inline)) {
      for (uint32_t c0 = ZI; c0 < total_prev; c0 += ZI) {
        wave_lds_sync();
        owner_tables(c0, oc_prev, ovf_prev, pre);
        const uint32_t cnt = total_prev - c0 < ZI ? total_prev - c0 : ZI;
        const bool mine = lane < cnt;
        const uint32_t ks = mine ? (uint32_t)own_l[lane] : 0u;
        const uint32_t eix = mine ? oix_l[lane] : 0u;
        const uint32_t tmeta = s_meta[wid][ks & 63u];
        wave_lds_sync();  // (the tables are read: the entries may land on them)
        issue_entries(eix);
        Rec<RW> trec;
        trec.load(rd + (r0 + (uint64_t)wt_prev * WT + (ks & 63u)) * (uint64_t)RW, RW);
        wait_vm0();
        const uint4 e0 = z_l[lane < ZI ? lane : 0], e1 = z_l[ZI + (lane < ZI ? lane : 0)], e2 = z_l[2 * ZI + (lane < ZI ? lane : 0)];
        unsigned long long av;
        entry_compare(cnt, ks, tmeta, trec, e0, e1, e2, av);
        count_accepted(av, c0, pre);
      }
    };
    // phase D for prev: per-read selection and the tuples (the protocol of match_ctx_pass; k_match_t's phase D with one
    // generation of lists)
    auto phase_d_prev = [&]() __attribute__((always_inline)) {
      const uint32_t nl = nlist_prev;
      const uint32_t nspill = nl > WLIST ? nl - WLIST : 0u;
      const bool spill_ok = nspill <= sregion;
      if (nspill > maxspill) maxspill = nspill;
      if (nspill) wait_vm0();  // this wave's spilled candidates have landed (and cur's window 0: a heavy tile pays for it)
      if (block_mode) {
#pragma unroll
        for (int k = 0; k < W; k++) {
          const uint32_t cw = wc_prev[k];
          if (!cw) continue;
          const uint32_t h = block_hash32((uint32_t)k, bb_prev[k]);
          if (block_mode == 1) atomicAdd(&s_sketch[h >> (32 - MATCH_SKETCH_BITS)], cw);
          else atomicAdd(&block_table[h >> (32 - BLOCK_TABLE_BITS)], cw);
        }
      }
      {
        const uint32_t b0 = best_l[lane];
        best_l[lane] = best_prev < b0 ? best_prev : b0;
      }
      cnt_l[lane] = 0;
      wave_lds_sync();
      auto item = [&](uint32_t j, uint32_t* gene, uint32_t* pos) __attribute__((always_inline)) -> uint32_t {
        if (j < WLIST) {
          const uint3 it = list_l[j];
          *gene = it.y;
          *pos = it.z;
          return it.x;
        }
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(spill + sregion_w + (j - WLIST));
        *gene = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *pos = __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      const uint32_t nuse = spill_ok ? nl : (nl < WLIST ? nl : WLIST);
      // the first round of candidates (a lane each; cfg3 has ~53 per wave-tile) stays in registers from the count to
      // the store; further rounds are walked twice
      uint32_t kw = NX_REJECT, kg = 0, kp = 0, ko = 0;
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
      const uint32_t cnum = cnt_l[lane];
      const uint32_t inc = wave_scan_incl(cnum);
      const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
      base_l[lane] = inc - cnum;
      const uint64_t base = region0 + used;
      const bool fits = spill_ok && used + total <= region;
      if (lane == 0) {
        tbase[wt_prev] = (uint32_t)base;
        tcount2[wt_prev] = fits ? total : 0u;
      }
      wave_lds_sync();
      if (fits && total) {
        if (kw != NX_REJECT) {
          const uint32_t rl = kw >> 24;
          stage[base + base_l[rl] + ko] = make_uint4((uint32_t)(r0 + wt_prev * WT + rl), kg, kp, kw & 0xFFFFu);
        }
        if (nuse > 64) {
          for (uint32_t j = 64 + lane; j < nuse; j += 64) {
            uint32_t g, p;
            const uint32_t w = item(j, &g, &p);
            const uint32_t rl = w >> 24, v = w & 0xFFFFu;
            const uint32_t thr = apply ? best_l[rl] + mmtol : 0xFFFFu;
            if (v > thr) continue;
            const uint32_t ord = atomicAdd(&cnt_l[rl], 0xFFFFFFFFu) - 1u;
            stage[base + base_l[rl] + ord] = make_uint4((uint32_t)(r0 + wt_prev * WT + rl), g, p, v);
          }
        }
      }
      wave_lds_sync();
      best_l[lane] = 0xFFFFFFFFu;  // for cur
      used += total;
      nrep += lane == 0 ? nl : 0u;
    };

    // ===================================================================== F1
    // this lane's item of prev (tables in the zone, written when prev's last window was compared): read BEFORE the
    // entries land on them
    const bool ent = have_prev && o_n > 0;
    wave_lds_sync();  // (everything this wave has read from or written to the zone and the line buffer is done)
    if (ent) issue_entries(o_eix);
    if (have_cur) issue_window(bb_cur[0]);
    PF(0)
    uint32_t nlist = 0;           // reported candidates of cur so far (wave-uniform)
    uint32_t best = 0xFFFFFFFFu;  // smallest mismatch count reported for this lane's read by the in-lane comparisons
    uint32_t ulen = 0xFFFFFFFFu, total_cur = 0;
    uint32_t wc[W], oc[W], ovf[W];
    uint32_t meta_nx = 0, bb_nx[W];
    Rec<RW> rec_nx;
    rec_nx.zero();
#pragma unroll
    for (int k = 0; k < W; k++) wc[k] = oc[k] = ovf[k] = 0, bb_nx[k] = WB_NONE;
    if (have_prev) {
      uint32_t pre[W], total = 0;
#pragma unroll
      for (int kk = 0; kk < W; kk++) {
        const uint32_t inc = wave_scan_incl(oc_prev[kk]);
        pre[kk] = total + inc - oc_prev[kk];
        total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      }
      if (ent) {
        if (have_cur) wait_vm8();  // the entries are there; the eight line instructions behind them still fly
        else wait_vm0();
        const uint32_t li = lane < ZI ? lane : 0u;
        const uint4 e0 = z_l[li], e1 = z_l[ZI + li], e2 = z_l[2 * ZI + li];
        unsigned long long av;
        entry_compare(o_n, o_ks, o_meta, o_rec, e0, e1, e2, av);
        count_accepted(av, 0, pre);
        PF(1)
        if (total_prev > ZI) overflow_rest_prev(pre);
      }
      wave_lds_sync();
      phase_d_prev();
      PF(2)
    }
    if (have_cur) {
      const bool active = wt * WT + lane < n;
      const int rlen = (int)REC_LEN(meta_cur);
      const uint32_t budget = REC_BUDGET(meta_cur), valid_cur = REC_VALID(meta_cur);
      s_meta[wid][lane] = meta_cur;
      const uint32_t len0 = (uint32_t)__builtin_amdgcn_readfirstlane(rlen);
      ulen = __ballot(active && (uint32_t)rlen != len0) == 0 ? len0 : 0xFFFFFFFFu;

      // a reported candidate of the lane's own read (in-lane comparisons): appended in lane order
      auto report_own = [&](uint32_t w, uint32_t gene, uint32_t pos, uint32_t& wck) __attribute__((always_inline)) {
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
        if (slot < WLIST) {
          list_l[slot] = make_uint3(w, gene, pos);
        } else if (slot - WLIST < sregion) {
          spill[sregion_w + (slot - WLIST)] = make_uint4(w, gene, pos, 0u);
        }
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
      {
        // ---- window 0: its image and tables while its lines fly
        const uint32_t sh0 = 2u * (uint32_t)(CL - win[0]);
        WinTab tb;
        win_tab(ulen, 0, sh0, (uint32_t)rlen, tb);
        uint32_t img[NW];
        read_image_n<RW, NW>(rec_cur, sh0, img);
        PF(3)
        wait_vm0();
        PF(4)
        slots(std::integral_constant<int, 0>{}, img, tb);
        PF(5)
      }
      // ================================================================= F2
      wave_lds_sync();  // (the lane has read its line of window 0; the zone's last readers were phase D / the entries)
      issue_window(bb_cur[1]);
      if (have_next) {
        // the records of the next wave-tile, RW / 4 instructions of 1 KB contiguous (chunks of reads past the batch's end
        // come from read 0), into the zone
        const uint32_t zb = lds_addr(z_l);
#pragma unroll
        for (int q = 0; q < RW / 4; q++) {
          const uint32_t g = (uint32_t)q * 64u + lane;
          const uint32_t i = (wt + nw) * WT + g / (uint32_t)(RW / 4);
          glds16(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW + 4u * (g % (uint32_t)(RW / 4)), zb + (uint32_t)q * 1024u);
        }
      }
      PF(6)
      {
        const uint32_t sh1 = 2u * (uint32_t)(CL - win[1]);
        WinTab tb;
        win_tab(ulen, 1, sh1, (uint32_t)rlen, tb);
        uint32_t img[NW];
        read_image_n<RW, NW>(rec_cur, sh1, img);
        PF(7)
        wait_vm0();
        PF(8)
        if (have_next) {
          // the next wave-tile's records -> registers, its phase A
          const uint4* src = z_l + lane * (RW / 4);
#pragma unroll
          for (int q = 0; q < RW / 4; q++) {
            const uint4 v = src[q];
            rec_nx.w[4 * q] = v.x; rec_nx.w[4 * q + 1] = v.y; rec_nx.w[4 * q + 2] = v.z; rec_nx.w[4 * q + 3] = v.w;
          }
          meta_nx = phase_a(wt + nw, rec_nx, bb_nx);
        }
        PF(9)
        slots(std::integral_constant<int, 1>{}, img, tb);
        PF(10)
      }
      // ---- this wave-tile's overflow entries: the first ZI are listed (item tables -> the zone, which the next
      // wave-tile's records have left) and their lanes take the reads' meta words and records along
      {
        uint32_t pre[W], total = 0;
#pragma unroll
        for (int kk = 0; kk < W; kk++) {
          const uint32_t inc = wave_scan_incl(oc[kk]);
          pre[kk] = total + inc - oc[kk];
          total += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        total_cur = total;
        wave_lds_sync();
        owner_tables(0, oc, ovf, pre);
        o_n = total < ZI ? total : ZI;
        const bool mine = lane < o_n;
        o_ks = mine ? (uint32_t)own_l[lane] : 0u;
        o_eix = mine ? oix_l[lane] : 0u;
        o_meta = s_meta[wid][o_ks & 63u];
        // the item's read's record: from the lane that owns the read (the LDS crossbar, no memory)
#pragma unroll
        for (int q = 0; q < RW; q++) o_rec.w[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((o_ks & 63u) * 4u), (int)rec_cur.w[q]);
      }
      PF(11)
#ifdef MUSC_LANE_PROF
      pf_tiles++;
#endif
    }
    // ---- cur becomes prev
    have_prev = have_cur;
    if (have_cur) {
      wt_prev = wt;
      nlist_prev = nlist;
      best_prev = best;
      ulen_prev = ulen;
      total_prev = total_cur;
#pragma unroll
      for (int k = 0; k < W; k++) wc_prev[k] = wc[k], bb_prev[k] = bb_cur[k], bb_cur[k] = bb_nx[k], oc_prev[k] = oc[k], ovf_prev[k] = ovf[k];
      meta_cur = meta_nx;
      rec_cur = rec_nx;
      wt += nw;
    }
  }
#ifdef MUSC_LANE_PROF
  if ((gw == 0 || gw == 1001) && (threadIdx.x & 63) == 0 && pf_tiles > 4)
    printf("wave %u: %u tiles, cycles/tile: total %llu | F1 issue %llu ovf compare %llu phase D %llu image0 %llu wait0 %llu slots0 %llu | F2 issue %llu image1 %llu wait1 %llu phase A next %llu slots1 %llu listing %llu\n",
           gw, pf_tiles, (__builtin_amdgcn_s_memtime() - pstart) / pf_tiles, pf[0] / pf_tiles, pf[1] / pf_tiles, pf[2] / pf_tiles, pf[3] / pf_tiles,
           pf[4] / pf_tiles, pf[5] / pf_tiles, pf[6] / pf_tiles, pf[7] / pf_tiles, pf[8] / pf_tiles, pf[9] / pf_tiles, pf[10] / pf_tiles, pf[11] / pf_tiles);
#endif
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
    for (uint32_t t = threadIdx.x; t < (1u << MATCH_SKETCH_BITS); t += TILE) hot |= s_sketch[t] >= block_thr;
    if (__any(hot) && (threadIdx.x & 63) == 0) atomicOr(&counters[6], 1ull);
  }
}
