// kernels_confirm.hpp -- k_confirm (cdiff + per-read selection + MaxMatches accounting), k_compact, tuple packing
// Part of libmuscato_hip.so: included by muscato_hip.hip (one translation unit).
#pragma once

// What one pair brings in from memory (static stride): descriptor, record, target span, masks.
template <int RW, bool MASK>
struct PairRegs {
  uint4 ds;
  uint32_t exact;  // rvalid of the read
  uint32_t r[RW ? RW : 1], t[RW ? RW : 1], rm[(RW && MASK) ? RW : 1], tm[(RW && MASK) ? RW : 1];
};

// pair_issue: every load of the pair, nothing that needs their results -- so that a caller can
// put other work between issue and finish
template <int RW, bool MASK>
DEV void pair_issue(PairRegs<RW, MASK>& P, const uint4 ds, const uint32_t* __restrict__ rd,
                    const uint32_t* __restrict__ rdm, const uint32_t* __restrict__ db2,
                    const uint32_t* __restrict__ dbm2, const uint32_t* __restrict__ dbx, uint64_t r0,
                    const uint32_t* __restrict__ rvalid) {
  static_assert(RW != 0, "static stride only");
  P.ds = ds;
  const uint32_t ri = ds.x & 0xFFFFFFu;
  const uint64_t gpos = (uint64_t)ds.y | ((uint64_t)(ds.x >> 24) << 32);
  const uint32_t* __restrict__ rec = rd + (r0 + ri) * (uint64_t)RW;
  const uint64_t widx = gpos >> 4;
  P.exact = rvalid[ri];
  // Mask words are fetched only where an X can be: the read says so in its descriptor, the
  // database in a bitmap of 64-base blocks that stays in L2 (X is rare and clustered; most
  // pairs then cost what they cost without mask planes).
  bool rx = false, tx = false;
  if constexpr (MASK) {
    rx = (ds.z & DESC_RX) != 0;
    tx = db_span_has_x(dbx, gpos, RW * 16);
  }
  // read records stream through once: non-temporal, so that the database -- the only operand
  // with reuse -- keeps the Infinity Cache
#pragma unroll
  for (int q = 0; q < RW / 4; q++) {
    const u32x4_v a = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(rec) + q);
    P.r[4 * q] = a.x; P.r[4 * q + 1] = a.y; P.r[4 * q + 2] = a.z; P.r[4 * q + 3] = a.w;
    const u32x4_u b = *reinterpret_cast<const u32x4_u*>(db2 + widx + 4 * q);
    P.t[4 * q] = b.x; P.t[4 * q + 1] = b.y; P.t[4 * q + 2] = b.z; P.t[4 * q + 3] = b.w;
    if constexpr (MASK) {
      uint4 c = make_uint4(0, 0, 0, 0);
      if (rx) c = *reinterpret_cast<const uint4*>(rdm + (r0 + ri) * (uint64_t)RW + 4 * q);
      P.rm[4 * q] = c.x; P.rm[4 * q + 1] = c.y; P.rm[4 * q + 2] = c.z; P.rm[4 * q + 3] = c.w;
      u32x4_u d = {0, 0, 0, 0};
      if (tx) d = *reinterpret_cast<const u32x4_u*>(dbm2 + widx + 4 * q);
      P.tm[4 * q] = d.x; P.tm[4 * q + 1] = d.y; P.tm[4 * q + 2] = d.z; P.tm[4 * q + 3] = d.w;
    }
  }
}

// The reference's confirm for window k (and for k + 1 when the descriptor stands for both)
// accepts this pair -- it counts towards that window-key block's MaxMatches; the tuple is
// reported here only if the first window that accepts it is one of this descriptor's.
DEV uint32_t pair_code(const uint4 ds, uint32_t nx, uint32_t budget, uint32_t exact) {
  const uint32_t ri = ds.x & 0xFFFFFFu, k = ds.z & 15u;
  const bool within = nx <= budget;
  const bool a0 = within && ((exact >> k) & 1u);
  const bool a1 = within && (ds.z & DESC_TWO) && ((exact >> (k + 1)) & 1u);
  if (!(a0 || a1)) return NX_REJECT;
  const uint32_t kmin = (uint32_t)(__ffs(exact) - 1);
  const bool first = (a0 && kmin == k) || (a1 && kmin == k + 1);
  return (first ? nx : (nx | NX_DUP)) | (a0 ? NX_ACC0 : 0u) | (a1 ? NX_ACC1 : 0u) | (k << 20) | ((ri & (TILE - 1)) << 24);
}

// pair_finish: XOR + popcount = cdiff over the whole read (cmd/muscato_confirm/main.go:151-159,
// 205-211; X==X through the mask plane) and, from the same mismatch mask, which windows of the
// read match the target exactly here (the first-window rule that makes the union over windows a
// set without a sort).  Returns the pair's result word.  budget_of(len) = the read's mismatch
// budget.
// W2: at most two windows -- the per-window loop is then two unrolled steps (measured: 4 % of
// the kernel; with a run-time bound the scalar unit recomputes the window masks per word).
template <int RW, bool MASK, bool W2, class BudgetOf>
DEV uint32_t pair_finish(const PairRegs<RW, MASK>& P, const PathParams& pp, BudgetOf budget_of) {
  const uint64_t gpos = (uint64_t)P.ds.y | ((uint64_t)(P.ds.x >> 24) << 32);
  const uint32_t sh = ((uint32_t)gpos & 15u) * 2u;
  uint32_t exact = P.exact;
  if ((P.ds.z >> 4) & 1u) exact &= ~pp.q1zero_mask;
  const uint32_t len = P.r[RW - 1] & 0xFFFFu;  // (bit 16 of the word is READ_HAS_X)
  const int len2 = 2 * (int)len;
  uint32_t nx = 0;
#pragma unroll
  for (int j = 0; j < RW - 1; j++) {
    const uint32_t tj = __funnelshift_r(P.t[j], P.t[j + 1], sh);
    const uint32_t x = P.r[j] ^ tj;
    uint32_t d = (x | (x >> 1)) & 0x55555555u;
    if constexpr (MASK) d |= (P.rm[j] ^ __funnelshift_r(P.tm[j], P.tm[j + 1], sh)) & 0x55555555u;
    const int rem = len2 - 32 * j;
    d &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
    nx += __popc(d);
    if constexpr (W2) {
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
        if (kk < pp.W && (d & window_word_mask(pp.win[kk], pp.ww, j))) exact &= ~(1u << kk);
    } else {
      for (int kk = 0; kk < pp.W; kk++)
        if (d & window_word_mask(pp.win[kk], pp.ww, j)) exact &= ~(1u << kk);
    }
  }
  return pair_code(P.ds, nx, budget_of(len), exact);
}

// confirm_pair -- cdiff for one candidate pair in one go.  Loads the 2-bit read record (aligned,
// neighbouring lanes mostly share it) and the target span at an arbitrary base offset
// (dword-aligned 16-byte gathers + funnel shift).  RW = record words (compile time) or 0 =
// runtime stride.  Returns the pair's result word (NX_REJECT, or nmiss | flags | window << 20 |
// slot << 24).
template <int RW, bool MASK, bool W2, class BudgetOf>
DEV uint32_t confirm_pair(const uint4 ds, const uint32_t* __restrict__ rd, const uint32_t* __restrict__ rdm,
                          const uint32_t* __restrict__ db2, const uint32_t* __restrict__ dbm2,
                          const uint32_t* __restrict__ dbx, uint64_t r0, int rw_rt, const PathParams& pp,
                          BudgetOf budget_of, const uint32_t* __restrict__ rvalid) {
  if constexpr (RW != 0) {
    PairRegs<RW, MASK> P;
    pair_issue<RW, MASK>(P, ds, rd, rdm, db2, dbm2, dbx, r0, rvalid);
    return pair_finish<RW, MASK, W2>(P, pp, budget_of);
  } else {
    // ---- runtime stride (reads longer than the compiled strides): streaming words
    const uint32_t ri = ds.x & 0xFFFFFFu;
    const int rw = rw_rt;
    const uint64_t gpos = (uint64_t)ds.y | ((uint64_t)(ds.x >> 24) << 32);
    const uint32_t* __restrict__ rec = rd + (r0 + ri) * (uint64_t)rw;
    const uint64_t widx = gpos >> 4;
    const uint32_t sh = ((uint32_t)gpos & 15u) * 2u;
    uint32_t exact = rvalid[ri];
    if ((ds.z >> 4) & 1u) exact &= ~pp.q1zero_mask;
    uint32_t nx = 0;
    const uint32_t len = rec[rw - 1] & 0xFFFFu;
    const int len2 = 2 * (int)len;
    const uint32_t* __restrict__ recm = MASK ? rdm + (r0 + ri) * (uint64_t)rw : nullptr;
    uint32_t tlo = db2[widx], tmlo = MASK ? dbm2[widx] : 0u;
    for (int j = 0; j < rw - 1; j++) {
      const uint32_t thi = db2[widx + j + 1];
      const uint32_t x = rec[j] ^ __funnelshift_r(tlo, thi, sh);
      tlo = thi;
      uint32_t d = (x | (x >> 1)) & 0x55555555u;
      if (MASK) {
        const uint32_t tmhi = dbm2[widx + j + 1];
        d |= (recm[j] ^ __funnelshift_r(tmlo, tmhi, sh)) & 0x55555555u;
        tmlo = tmhi;
      }
      const int rem = len2 - 32 * j;
      d &= rem >= 32 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
      nx += __popc(d);
      for (int kk = 0; kk < pp.W; kk++)
        if (d & window_word_mask(pp.win[kk], pp.ww, j)) exact &= ~(1u << kk);
    }
    return pair_code(ds, nx, budget_of(len), exact);
  }
}

#define BLOCK_LDS_BITS 11  // sketch size
#define CONF_TILES 32      // tiles per k_confirm workgroup (65536 tiles per batch / MAX_GRID = 16)


#define CODE_CAP 1024      // result words beyond a lane's first kept in LDS; a larger tile spills the rest to p_nx

// k_confirm -- muscato_confirm for one tile of k_screen per workgroup iteration, followed in
// the same workgroup by the per-read best + MMTol filter
// (cmd/muscato_combine_windows/main.go:36-60) and the MaxMatches block accounting: a tile's
// pairs are desc[tbase[tile] .. +tcount[tile]) in any order, every per-read quantity lives in
// LDS and the result words never travel through HBM (tiles of more than CODE_CAP pairs spill).
//   pass 1: one lane per pair: cdiff; best[read] = min nmiss over its reported pairs;
//           wcnt[read][window] = pairs that window's confirm accepts (its share of the
//           (window,key) block, which cmd/muscato_confirm/main.go:233-242, 424-448 truncate
//           at MaxMatches)
//   pass 2: cnt[read] = pairs with nmiss <= best + MMTol (all accepted pairs when
//           apply_mmtol == 0); scan over the tile's reads
//   pass 3: the surviving tuples go to stage[tbase[tile] + ...], reads in order, a read's
//           tuples contiguous (arrival order within one read); tcount2[tile] = how many.
//           k_compact then closes the gaps between tiles.
// block_mode 0: no MaxMatches accounting.
// block_mode 1: screening -- each workgroup keeps a count-min sketch of (window, key) -> accepted
//   pairs in LDS across all its tiles; if no sketch cell of any workgroup of any launch reaches
//   `block_thr` = floor(MaxMatches / number of workgroup-launches), then by pigeonhole no block can
//   hold more than MaxMatches pairs (cells only over-estimate).  Otherwise counters[6] is raised
//   and the host repeats the pass in mode 2.
// block_mode 2: exact -- one global atomic per (read, window) into a 2^22-cell table.
// (Eight waves per SIMD where the record fits 64 registers without spilling: measured 1.82 ms
// per cfg3 pass against 1.95 ms at the compiler's own choice.  Prefetching the next tile's gathers across the select passes was tried and lost --
// 2.1 ms: the registers it holds cost more waves than the overlap wins.)
template <int RW, bool MASK, bool W2>
__global__ __launch_bounds__(TILE, (RW <= 8 && !(RW == 8 && MASK)) ? 8 : 4) void k_confirm(
    const uint32_t* __restrict__ rd, const uint32_t* __restrict__ rdm,
    const uint32_t* __restrict__ db2, const uint32_t* __restrict__ dbm2, const uint32_t* __restrict__ dbx,
    uint64_t r0, uint32_t n, int rw_rt,
    const PathParams* __restrict__ ppp, const uint16_t* __restrict__ nmiss_tab, const uint4* __restrict__ cdesc,
    const uint32_t* __restrict__ rvalid, uint32_t* __restrict__ p_nx,
    const uint32_t* __restrict__ tbase, const uint32_t* __restrict__ tcount,
    const uint32_t* __restrict__ wb, int block_mode, uint32_t block_thr, uint32_t* __restrict__ block_table,
    const uint64_t* __restrict__ seq_off, uint4* __restrict__ stage, uint32_t* __restrict__ tcount2,
    unsigned long long* __restrict__ counters) {
  const PathParams& pp = *ppp;  // read where used (scalar loads), see k_screen
  extern __shared__ uint32_t s_wcnt[];  // TILE * W counters when block_mode != 0
  __shared__ uint32_t s_best[TILE], s_cnt[TILE], s_base[TILE];
  __shared__ uint32_t s_code[CODE_CAP];
  __shared__ uint32_t s_sketch[1 << BLOCK_LDS_BITS];
  __shared__ uint32_t s_wsum[TILE / 64];
  __shared__ uint16_t s_nm[CONF_NM];                 // mismatch budget of the short read lengths
  __shared__ uint32_t s_tb[CONF_TILES], s_tn[CONF_TILES];  // this workgroup's tiles: descriptor range
  const uint32_t ntiles = (n + TILE - 1) / TILE;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (block_mode == 1)
    for (uint32_t t = threadIdx.x; t < (1u << BLOCK_LDS_BITS); t += TILE) s_sketch[t] = 0;
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += TILE) s_nm[t] = t <= (uint32_t)pp.max_len ? nmiss_tab[t] : (uint16_t)0;
  // tile j of this workgroup = blockIdx.x + j * gridDim.x (the host keeps it to CONF_TILES)
  const uint32_t my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  for (uint32_t j = threadIdx.x; j < my_tiles && j < CONF_TILES; j += TILE) {
    s_tb[j] = tbase[blockIdx.x + j * gridDim.x];
    s_tn[j] = tcount[blockIdx.x + j * gridDim.x];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) tcount2[ntiles] = 0;
  unsigned long long acc = 0;
  auto budget_of = [&](uint32_t len) -> uint32_t { return len < CONF_NM ? s_nm[len] : nmiss_tab[len]; };
  __syncthreads();

  for (uint32_t j = 0; j < my_tiles; j++) {
    const uint32_t tile = blockIdx.x + j * gridDim.x;
    lds_barrier();  // the previous tile is done with the LDS state
    s_best[threadIdx.x] = 0xFFFFFFFFu;
    s_cnt[threadIdx.x] = 0;
    if (block_mode)
      for (uint32_t t = threadIdx.x; t < TILE * (uint32_t)pp.W; t += TILE) s_wcnt[t] = 0;
    lds_barrier();
    const uint32_t tn = s_tn[j];
    const uint64_t tb = s_tb[j];
    // ---- pass 1.  The lane's first pair (five tiles in six have no second) stays in
    // registers through all three passes; later ones park their result word in LDS.
    auto tally = [&](uint32_t w) {
      if (w == NX_REJECT) return;
      const uint32_t rl = w >> 24;
      if (block_mode) {
        const uint32_t k = (w >> 20) & 15u;
        if (w & NX_ACC0) atomicAdd(&s_wcnt[rl * pp.W + k], 1u);
        if (w & NX_ACC1) atomicAdd(&s_wcnt[rl * pp.W + k + 1], 1u);
      }
      if (w & NX_DUP) return;
      atomicMin(&s_best[rl], w & 0xFFFFu);
      acc++;
    };
    auto confirm_at = [&](uint32_t tj, uint32_t* gene, uint32_t* zword) -> uint32_t {
      // descriptors stream through once: non-temporal
      const u32x4_v dsv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_v*>(cdesc) + tb + tj);
      *gene = dsv.w;
      *zword = dsv.z;
      return confirm_pair<RW, MASK, W2>(make_uint4(dsv.x, dsv.y, dsv.z, dsv.w), rd, rdm, db2, dbm2, dbx, r0, rw_rt, pp,
                                    budget_of, rvalid);
    };
    uint32_t w0 = NX_REJECT, gene0 = 0, z0 = 0;
    if (threadIdx.x < tn) w0 = confirm_at(threadIdx.x, &gene0, &z0);
    tally(w0);
    for (uint32_t tj = threadIdx.x + TILE; tj < tn; tj += TILE) {
      uint32_t g, z;
      const uint32_t w = confirm_at(tj, &g, &z);
      if (tj - TILE < CODE_CAP) s_code[tj - TILE] = w; else p_nx[tb + tj] = w;
      tally(w);
    }
    lds_barrier();
    // ---- pass 2
    auto count = [&](uint32_t w) {
      if (w == NX_REJECT || (w & NX_DUP)) return;
      const uint32_t rl = w >> 24;
      const uint32_t thr = pp.apply_mmtol ? s_best[rl] + (uint32_t)pp.mmtol : 0xFFFFu;
      if ((w & 0xFFFFu) <= thr) atomicAdd(&s_cnt[rl], 1u);
    };
    count(w0);
    for (uint32_t tj = threadIdx.x + TILE; tj < tn; tj += TILE)
      count(tj - TILE < CODE_CAP ? s_code[tj - TILE] : p_nx[tb + tj]);
    if (block_mode) {
      for (uint32_t t = threadIdx.x; t < TILE * (uint32_t)pp.W; t += TILE) {
        const uint32_t cw = s_wcnt[t];
        if (!cw) continue;
        const uint32_t rl = t / pp.W, k = t % pp.W;
        const uint32_t h = block_hash32((uint32_t)k, wb[((uint64_t)tile * TILE + rl) * pp.W + k]);
        if (block_mode == 1) atomicAdd(&s_sketch[h >> (32 - BLOCK_LDS_BITS)], cw);
        else atomicAdd(&block_table[h >> (32 - BLOCK_TABLE_BITS)], cw);
      }
    }
    lds_barrier();
    // ---- scan of the per-read counts
    const uint32_t c = s_cnt[threadIdx.x];
    uint32_t inc = c;
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
    s_base[threadIdx.x] = woff + inc - c;
    s_cnt[threadIdx.x] = 0;  // now the arrival counter of the read
    if (threadIdx.x == 0) tcount2[tile] = total;
    lds_barrier();
    // ---- pass 3
    if (total == 0) continue;
    auto emit = [&](uint32_t w, uint32_t tj, uint32_t gene, uint32_t zword) {
      if (w == NX_REJECT || (w & NX_DUP)) return;
      const uint32_t rl = w >> 24, v = w & 0xFFFFu;
      const uint32_t thr = pp.apply_mmtol ? s_best[rl] + (uint32_t)pp.mmtol : 0xFFFFu;
      if (v > thr) return;
      const uint32_t ord = atomicAdd(&s_cnt[rl], 1u);
      // position in the target: carried in the descriptor unless the target is so long that
      // the entry's 16-bit distance saturated (then the placement's offset minus the gene's)
      uint32_t pos = (zword >> 6) & 0xFFFFu;
      if (!((zword >> 5) & 1u)) {
        const uint4 ds = cdesc[tb + tj];
        pos = (uint32_t)(((uint64_t)ds.y | ((uint64_t)(ds.x >> 24) << 32)) - seq_off[gene]);
      }
      stage[tb + s_base[rl] + ord] = make_uint4((uint32_t)(r0 + tile * TILE + rl), gene, pos, v);
    };
    emit(w0, threadIdx.x, gene0, z0);
    for (uint32_t tj = threadIdx.x + TILE; tj < tn; tj += TILE) {
      const uint32_t w = tj - TILE < CODE_CAP ? s_code[tj - TILE] : p_nx[tb + tj];
      if (w == NX_REJECT || (w & NX_DUP)) continue;
      const uint4 ds = cdesc[tb + tj];
      emit(w, tj, ds.w, ds.z);
    }
  }
  block_add_u64(acc, &counters[1]);
  if (block_mode == 1) {
    lds_barrier();
    uint32_t hot = 0;
    for (uint32_t t = threadIdx.x; t < (1u << BLOCK_LDS_BITS); t += TILE) hot |= s_sketch[t] >= block_thr;
    if (__any(hot) && (threadIdx.x & 63) == 0) atomicOr(&counters[6], 1ull);
  }
}

// k_compact -- hits[counters[2] + tpre[tile] ...] = the tile's staged tuples (tpre = scan of
// tcount2): plain 16-byte copies, a tile's run is contiguous on both sides.
MUSC_KERNEL __launch_bounds__(256) void k_compact(uint32_t ntiles, const uint32_t* __restrict__ tbase,
                                                 const uint32_t* __restrict__ tcount2,
                                                 const uint32_t* __restrict__ tpre,
                                                 const uint4* __restrict__ stage, uint4* __restrict__ hits,
                                                 uint64_t hits_cap, unsigned long long* __restrict__ counters) {
  const unsigned long long base = counters[2];
  if (base + tpre[ntiles] > hits_cap) {  // cannot happen on a sized pass; a sync-free pass re-runs sized
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&counters[3], 2ull);
    return;
  }
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t m = tcount2[tile];
    const uint4* __restrict__ src = stage + tbase[tile];
    uint4* __restrict__ dst = hits + base + tpre[tile];
    for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) dst[j] = src[j];
  }
}

// After an exact (mode 2) pass: the (read, window) probes whose (window, key) block counter is
// above MaxMatches -- the blocks the reference would have truncated.  One thread per read.
template <int RW>
__global__ __launch_bounds__(256) void k_hot_probes(const uint32_t* __restrict__ rd,
                                                    const uint32_t* __restrict__ rdm, uint64_t nreads,
                                                    int rw_rt, const PathParams* __restrict__ ppp,
                                                    const uint32_t* __restrict__ block_table,
                                                    uint32_t max_matches, uint2* __restrict__ out,
                                                    uint64_t cap, unsigned long long* __restrict__ cursor) {
  const PathParams& pp = *ppp;
  const int rw = RW ? RW : rw_rt;
  const bool has_m = rdm != nullptr;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nreads; i += (uint64_t)gridDim.x * blockDim.x) {
    Rec<RW> rec;
    rec.load(rd + i * (uint64_t)rw, rw);
    Rec<RW> recm = rec;
    if (has_m) recm.load(rdm + i * (uint64_t)rw, rw);
    const uint32_t len = rec.len();
    for (int k = 0; k < pp.W; k++) {
      const uint32_t q1 = (uint32_t)pp.win[k], q2 = q1 + (uint32_t)pp.ww;
      if (len < q2) continue;
      if (pp.min_dinuc > 0 && rec_count_dinuc(rec, recm, has_m, q1, pp.ww) < pp.min_dinuc) continue;
      const uint32_t b = rec_bucket(rec, recm, has_m, q1, pp.ww, pp.bits, pp.direct);
      const uint32_t h = block_hash32((uint32_t)k, b);
      if (block_table[h >> (32 - BLOCK_TABLE_BITS)] > max_matches) {
        const unsigned long long slot = atomicAdd(cursor, 1ull);
        if (slot < cap) out[slot] = make_uint2((uint32_t)i, (uint32_t)k);
      }
    }
  }
}

// counters[2] (hits so far) += tpre[ntiles] (hits of this batch)
MUSC_KERNEL void k_advance(const uint32_t* __restrict__ tpre, uint32_t ntiles, unsigned long long* counters) {
  if (threadIdx.x == 0 && blockIdx.x == 0) counters[2] += tpre[ntiles];
}

// number of block counters above MaxMatches (hash collisions only inflate counters, so 0 is
// a proof that no window-key block overflowed)
MUSC_KERNEL void k_block_overflow(const uint32_t* __restrict__ block_table, uint32_t max_matches,
                                 unsigned long long* __restrict__ counters) {
  unsigned long long c = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << BLOCK_TABLE_BITS); i += gridDim.x * blockDim.x)
    c += block_table[i] > max_matches;
  block_add_u64(c, &counters[5]);
}

// Tuples as one u64 each for the wire (RCCL gather to rank 0): read index (+ the shard's base)
// in the top bits, then gene, position, mismatch count with caller-chosen widths; numeric order
// of the words = lexicographic order of the tuples.  *bad is raised if a field does not fit.
struct PackBits {
  int32_t read, gene, pos, nmiss;
};

MUSC_KERNEL __launch_bounds__(256) void k_pack_hits(const uint4* __restrict__ hits, uint64_t n, uint64_t read_base,
                                                   PackBits b, uint64_t* __restrict__ out, uint32_t* __restrict__ bad) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 h = hits[i];
    const uint64_t r = (uint64_t)h.x + read_base;
    if ((b.read < 64 && (r >> b.read)) || ((uint64_t)h.y >> b.gene) || ((uint64_t)h.z >> b.pos) || ((uint64_t)h.w >> b.nmiss))
      atomicOr(bad, 1u);
    out[i] = (((((r << b.gene) | h.y) << b.pos) | h.z) << b.nmiss) | h.w;
  }
}

MUSC_KERNEL __launch_bounds__(256) void k_unpack_hits(const uint64_t* __restrict__ in, uint64_t n, PackBits b,
                                                     uint4* __restrict__ hits) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v = in[i];
    const uint32_t nm = (uint32_t)(v & ((1ull << b.nmiss) - 1ull));
    v >>= b.nmiss;
    const uint32_t pos = (uint32_t)(v & ((1ull << b.pos) - 1ull));
    v >>= b.pos;
    const uint32_t gene = (uint32_t)(v & ((1ull << b.gene) - 1ull));
    v >>= b.gene;
    hits[i] = make_uint4((uint32_t)v, gene, pos, nm);
  }
}

// Tuples in their most compact wire form.  The hit list is read-major (a read's tuples are
// contiguous, reads in increasing order: both match paths stage them that way), so the read index
// travels as one byte per read -- counts[r] = tuples of read r -- and a tuple is ONE u32 word:
// gene | pos | nmiss with caller-chosen widths.  *bad: 1 a field does not fit, 2 a read has more
// than 255 tuples, 4 the list is not read-major (cannot happen; checked because the format rests on it).
MUSC_KERNEL __launch_bounds__(256) void k_pack_compact(const uint4* __restrict__ hits, uint64_t n, PackBits b,
                                                      uint32_t* __restrict__ words, uint8_t* __restrict__ counts,
                                                      uint32_t* __restrict__ bad) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 h = hits[i];
    if (((uint64_t)h.y >> b.gene) || ((uint64_t)h.z >> b.pos) || ((uint64_t)h.w >> b.nmiss)) atomicOr(bad, 1u);
    words[i] = (((h.y << b.pos) | h.z) << b.nmiss) | h.w;
    const uint32_t prev = i ? hits[i - 1].x : 0xFFFFFFFFu;
    if (i && prev > h.x) atomicOr(bad, 4u);
    if (i == 0 || prev != h.x) {  // the first tuple of its read counts the run
      uint32_t cnt = 1;
      while (i + cnt < n && cnt < 256 && hits[i + cnt].x == h.x) cnt++;
      if (cnt > 255) atomicOr(bad, 2u);
      counts[h.x] = (uint8_t)cnt;
    }
  }
}
