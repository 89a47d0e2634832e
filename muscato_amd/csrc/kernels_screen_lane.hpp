// k_screen_t -- k_screen for line buckets (kernels_index.hpp) with the tests done IN THE LANE THAT OWNS
// THE READ: the screen of the two-kernel path for dense databases (BASELINE config 5: 10 Gbp, ten
// index entries per probe).  Device code, included by muscato_hip.hip.
//
// k_screen walks a tile of 256 reads through workgroup-wide phases separated by barriers (buckets by
// quads, then the overflow entries as one flat list): on the cfg5 shard a workgroup needs 58 us per
// tile -- 140 000 cycles for 7 680 sixteen-byte entries -- because every phase waits for the slowest
// wave's memory round trip and four workgroups per CU cannot hide twenty of those per tile.  Neither
// fewer lines per probe (line buckets: 1.75 instead of 3) nor more loads in flight per lane moved
// it.  Here a WAVE works alone on its 64 reads of the tile (k_match_t's way, kernels_match_lane.hpp):
//   * a probe's line -- 16-byte header + seven entries -- is fetched by EIGHT lanes (16 contiguous bytes per
//     lane: a load instruction brings eight whole lines; r03 used a quad and two loads per line), written to a
//     per-wave line buffer in LDS and read back whole by the lane that owns the read (the buffer is
//     XOR-swizzled and the swizzle is applied to the SOURCE chunk: no bank conflicts either way); that lane tests the
//     seven entries against its own read's flanks and length (screen_entry_ok: the fit rules of
//     cmd/muscato_screen/main.go:294-316, 335-363 + cmd/muscato_confirm/main.go:201-203 and the 8+8
//     flank pre-filter) straight from registers;
//   * the entries beyond the seventh are 128-byte-aligned lines of eight: the wave lists the lines
//     its probes need (72 % of the probes need one at cfg5), fetches up to 64 of them the same way,
//     and a lane per LINE tests its eight entries with the owner's flanks from LDS;
//   * the next window's lines are requested before the current window is tested.
// Survivors become descriptors exactly as in k_screen (same format, same per-workgroup regions, a
// tile's descriptors contiguous, a probe's next to each other), so k_confirm takes over unchanged.
// A workgroup is ONE wave that walks the tile's 256 reads as four sub-tiles of 64: no barrier, no
// shared counter, and the next sub-tile's lines are on their way across the seam.  No two-window descriptors (k_confirm's first-window rule keeps the
// tuple set the same).  Reads or databases with X, and record strides beyond 16 words, stay with
// k_screen.
#pragma once

#ifndef SCRT_WAVES
#define SCRT_WAVES 3
#endif

template <int RW>
__global__ __launch_bounds__(64, SCRT_WAVES) void k_screen_t(const uint32_t* __restrict__ rd, uint64_t r0, uint32_t n,
                                                             const PathParams* __restrict__ ppp,
                                                             const uint16_t* __restrict__ nmiss_tab,
                                                             const LineBucket* __restrict__ T, const uint4* __restrict__ E,
                                                             uint4* __restrict__ desc, uint64_t desc_cap,
                                                             uint32_t* __restrict__ rvalid, uint32_t* __restrict__ wb,
                                                             uint32_t* __restrict__ tbase, uint32_t* __restrict__ tcount,
                                                             unsigned long long* __restrict__ counters,
                                                             unsigned long long* __restrict__ pass_flags) {
  const PathParams& pp = *ppp;
  __shared__ uint4 s_line[64 * 8];      // 64 lines (a window's buckets, or overflow lines), swizzled
  __shared__ uint32_t s_rfl[64];        // the window's probes: the read's own 8+8 flanking bases
  __shared__ uint32_t s_lenbud[64];     //                      read length | mismatch budget << 16
  __shared__ uint32_t s_oln[64];        // overflow lines in hand: which line of E
  __shared__ uint8_t s_own[64];         //                         the read (lane) that owns it
  __shared__ uint8_t s_ocn[64];         //                         entries of it that exist (1..8)
  __shared__ uint16_t s_nm[CONF_NM];    // mismatch budget of the short read lengths
  for (uint32_t t = threadIdx.x; t < CONF_NM; t += 64) s_nm[t] = t <= (uint32_t)pp.max_len ? nmiss_tab[t] : (uint16_t)0;
  wave_lds_sync();
  const int W = pp.W, ww = pp.ww, min_dinuc = pp.min_dinuc, direct = pp.direct;
  // the window starts and the position width, once: read where they are used they are scalar loads inside the step
  // and inside the survivors' loop, each with its own wait (a spilled scalar register comes back by v_readlane)
  const bool wide = pp.wide != 0;
  const int win0 = pp.win[0], win1 = pp.win[1], win2 = pp.win[2], win3 = pp.win[3];
  auto win_of = [&](int k) __attribute__((always_inline)) -> int { return k == 0 ? win0 : k == 1 ? win1 : k == 2 ? win2 : k == 3 ? win3 : pp.win[k]; };
  const uint32_t ntiles = (n + TILE - 1) / TILE;
  unsigned long long nvalid = 0, ncand = 0;  // wave-uniform (scalar registers): the wave's totals
  const uint64_t region = desc_cap / gridDim.x;
  const uint64_t region0 = region * blockIdx.x;
  uint64_t used = 0;  // descriptors this wave has needed so far
  const uint4* __restrict__ TL = reinterpret_cast<const uint4*>(T);
  const uint32_t lane = opaque(threadIdx.x) & 63;
  const uint32_t rb = lane * 8u + (((lane >> 1) & 7u) ^ (lane & 1u));  // this lane's line in the buffer, chunk c at rb ^ c

  // the loads of 64 lines: eight loads of EIGHT WHOLE LINES each (eight lanes per line, 16 contiguous bytes per
  // lane: the access shape of k_match_t's issue_window -- a quad per line with 32 bytes per lane in two loads made
  // every line two strided half-requests).  lid = this lane's line (an index into `base` in lines of eight uint4,
  // or WB_NONE: the lane that asked then sees line 0 of `base` and must ignore it); the eight lanes of a line get
  // its number from the lane that names it (ds_bpermute), all eight before the first address is formed.  Lane l
  // of load i takes the chunk that belongs at slot l & 7 of line 8 i + (l >> 3) in the swizzled line buffer
  // (chunk c of line p at slot c ^ ((p >> 1) & 7) ^ (p & 1)), so land64 writes 1 KB contiguously per load.
  auto issue64 = [&](const uint4* __restrict__ base, uint32_t lid, uint4 (&a)[4], uint4 (&b2)[4]) __attribute__((always_inline)) {
    uint32_t bq[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bq[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((i * 8 + (lane >> 3)) * 4), (int)lid);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      // the chunk: (lane & 7) ^ sw(p), p = 8 i + (lane >> 3): one lane constant, bit 2 flipped on the odd loads
      const uint32_t c = ((lane & 7u) ^ (lane >> 4) ^ ((lane >> 3) & 1u)) ^ (((uint32_t)i & 1u) << 2);
      const u32x4_v* p = reinterpret_cast<const u32x4_v*>(base + (uint64_t)(bq[i] != WB_NONE ? bq[i] : 0u) * 8u) + c;
      const u32x4_v x = __builtin_nontemporal_load(p);
      if (i < 4) a[i] = make_uint4(x.x, x.y, x.z, x.w);
      else b2[i - 4] = make_uint4(x.x, x.y, x.z, x.w);
    }
  };
  auto land64 = [&](const uint4 (&a)[4], const uint4 (&b2)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; i++) s_line[i * 64 + lane] = i < 4 ? a[i] : b2[i - 4];
  };
  // the bucket of a read's window k (cmd/muscato_window_reads/main.go:106-118 ==
  // cmd/muscato_screen/main.go:174-185: long enough, CountDinuc >= MinDinuc), WB_NONE when it takes no part
  auto bucket_k = [&](const Rec<RW>& rec, bool active, int k) __attribute__((always_inline)) -> uint32_t {
    const uint32_t q1 = (uint32_t)win_of(k), q2 = q1 + (uint32_t)ww;
    bool pt = active && rec.len() >= q2;
    if (ww <= 16 && direct) {
      // the usual case: the window key is one 32-bit word, the bucket its bases in reading order
      const uint32_t key = rec.ext32(2 * q1) & (ww == 16 ? 0xFFFFFFFFu : ((1u << (2 * ww)) - 1u));
      if (min_dinuc > 0) pt = pt && key_dinucs16(key, ww) >= min_dinuc;
      return pt ? __brev(key) >> (32 - 2 * ww) : WB_NONE;
    }
    if (pt && min_dinuc > 0) pt = rec_count_dinuc(rec, rec, false, q1, ww) >= min_dinuc;
    return pt ? rec_bucket(rec, rec, false, q1, ww, pp.bits, direct) : WB_NONE;
  };
  auto rec_at = [&](uint32_t i, Rec<RW>& rec) __attribute__((always_inline)) {
    rec.load_nt(rd + (r0 + (i < n ? i : 0)) * (uint64_t)RW);
  };

#ifdef SCRT_PROF
  unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime(), pstart = pt0;
  uint32_t pf_steps = 0;
#define SPF(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pf[i] += t_ - pt0; pt0 = t_; }
#else
#define SPF(i)
#endif
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t base = region0 + used;
    const uint64_t room = region > used ? region - used : 0;  // descriptors this tile may still write
    uint32_t tilecnt = 0;  // survivors of the tile so far (wave-uniform)
    // The survivors of up to eight entries per lane (okm: one bit per entry of this lane's line that
    // passed; zm: its z flag; entry s sits at chunk c0 + s of the lane's line in the line buffer):
    // every lane writes its survivors next to each other, the lanes' runs in lane order.
    auto append_all = [&](uint32_t okm, uint32_t zm, uint32_t c0, uint32_t rit, int k, int q1) __attribute__((always_inline)) {
      const uint32_t nmine = (uint32_t)__popc(okm);
      const uint32_t inc = wave_scan_incl(nmine);
      const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      if (tot == 0) return;
      uint32_t slot = tilecnt + inc - nmine;
      tilecnt += tot;
      while (okm) {
        const uint32_t e = (uint32_t)__ffs(okm) - 1u;
        okm &= okm - 1u;
        const uint4 ent = s_line[rb ^ (c0 + e)];
        if (slot < room) {
          const uint32_t left = ent.z & 0xFFFFu;
          const uint32_t pos_ok = left < 65535u ? 1u : 0u;
          // global offset of the placement (40 bits in wide mode: the high byte rides in x)
          const uint64_t gp = (((uint64_t)(wide ? ent.x >> 24 : 0u) << 32) | ent.y) - (uint64_t)q1;
          desc[base + slot] = make_uint4((tile * TILE + rit) | ((uint32_t)(gp >> 32) << 24), (uint32_t)gp,
                                         (uint32_t)k | (((zm >> e) & 1u) << 4) | (pos_ok << 5) | ((left - (uint32_t)q1) << 6),
                                         wide ? (ent.x & 0xFFFFFFu) : ent.x);
        }
        slot++;
      }
    };

    // a tile = four sub-tiles of 64 reads, one after the other; the lines of the next step -- the
    // sub-tile's next window, or the next sub-tile's first -- are requested before the current step
    // is tested (the next sub-tile's records were fetched a sub-tile earlier)
    uint4 va[4], vb[4], oa[4], ob[4];
    Rec<RW> rec, rec_nx;
    rec_at(tile * TILE + lane, rec);
    rec_nx = rec;
    uint32_t b_cur = bucket_k(rec, tile * TILE + lane < n, 0);
    issue64(TL, b_cur, va, vb);
#pragma unroll 1
    for (uint32_t sub = 0; sub < TILE / 64; sub++) {
      const uint32_t i = tile * TILE + sub * 64 + lane;
      const bool active = i < n;
      const int len = (int)rec.len();
      const uint32_t budget = len < CONF_NM ? s_nm[len] : nmiss_tab[len];
      const uint32_t lenbud = (uint32_t)len | (budget << 16);
      uint32_t valid = 0;
      if (sub + 1 < TILE / 64) rec_at(i + 64, rec_nx);
#pragma unroll 1
      for (int k = 0; k < W; k++) {
        const int q1 = win_of(k);
        const uint32_t q2 = (uint32_t)q1 + (uint32_t)ww;
        SPF(0)
        // ---- the step's 64 lines arrive; the next step's are requested at once
        land64(va, vb);
        SPF(1)
        uint32_t b_nx = WB_NONE;
        auto issue_next = [&]() __attribute__((always_inline)) {
          if (k + 1 < W) {
            b_nx = bucket_k(rec, active, k + 1);
            issue64(TL, b_nx, va, vb);
          } else if (sub + 1 < TILE / 64) {
            b_nx = bucket_k(rec_nx, i + 64 < n, 0);
            issue64(TL, b_nx, va, vb);
          }
        };
#ifndef SCRT_OVF_FIRST
        issue_next();
#endif
        const bool pv = b_cur != WB_NONE;
        const uint32_t rfl = pv ? (rec_flank_left(rec, (uint32_t)q1) | ((rec.ext32(2u * q2) & 0xFFFFu) << 16)) : 0u;
        s_rfl[lane] = rfl;
        s_lenbud[lane] = lenbud;
        if (active) wb[(uint64_t)i * W + k] = b_cur;
        valid |= pv ? 1u << k : 0u;
        nvalid += (uint32_t)__popcll(__ballot(pv));
        SPF(2)
        wave_lds_sync();
        // ---- lane p takes line p: its header first -- the lines of eight that hold the entries beyond the
        // seventh are listed and requested at once (the first 64 of them), so that they are on their way,
        // like the next step's lines, while the seven inline entries are tested
        const uint4 h = s_line[rb];
        const uint32_t cnt = pv ? h.x : 0u;
        ncand += (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(cnt < 0xFFFFFFu ? cnt : 0xFFFFFFu), 63);
#if defined(SCRT_DBG) && (SCRT_DBG & 2)
        const uint32_t nl = 0;
#else
        const uint32_t nl = cnt > LINE_INLINE ? (cnt - LINE_INLINE + 7u) / 8u : 0u;  // lines this probe needs
#endif
        const uint32_t inc = wave_scan_incl(nl);
        const uint32_t pre = inc - nl;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        // this lane's lines that fall into [c0, c0 + 64): who owns them, where they are, how many entries they hold
        auto list_lines = [&](uint32_t c0) __attribute__((always_inline)) {
          const uint32_t j_lo = c0 > pre ? c0 - pre : 0u;
          const uint32_t j_hi = pre + nl > c0 + 64u ? (c0 + 64u > pre ? c0 + 64u - pre : 0u) : nl;
#pragma unroll 1
          for (uint32_t j = j_lo; j < j_hi; j++) {
            const uint32_t rest = cnt - LINE_INLINE - 8u * j;
            s_own[pre + j - c0] = (uint8_t)lane;
            s_oln[pre + j - c0] = h.y + j;
            s_ocn[pre + j - c0] = (uint8_t)(rest < 8u ? rest : 8u);
          }
          wave_lds_sync();
        };
        if (total) {
          list_lines(0);
          issue64(E, lane < (total < 64u ? total : 64u) ? s_oln[lane] : WB_NONE, oa, ob);
        }
#ifdef SCRT_OVF_FIRST
        issue_next();
#endif
        SPF(3)
        {
          uint32_t okm = 0, zm = 0;
#pragma unroll
          // (straight-line: the seven entries are read from the lane's own line whether they exist or not, and every
          // test is evaluated -- no exec-mask regions, the LDS reads of all seven issued ahead of the first test; four
          // probes in five have more than seven entries, so almost nothing was skipped by testing `live` first)
          for (int s = 0; s < LINE_INLINE; s++) {
            const uint4 ent = s_line[rb ^ (uint32_t)(s + 1)];
            uint32_t z = 0;
            bool ok = screen_entry_ok(ent, q1, ww, rfl, lenbud, &z) & ((uint32_t)s < cnt);
#if defined(SCRT_DBG) && (SCRT_DBG & 1)
            ok = false;
#endif
            okm |= ok ? 1u << s : 0u;
            zm |= (ok ? z : 0u) << s;
          }
          append_all(okm, zm, 1u, sub * 64 + lane, k, q1);
        }
        SPF(4)
        // ---- the lines of eight, 64 at a time, a lane per line with its owner's flanks from LDS
        for (uint32_t c0 = 0; c0 < total; c0 += 64) {
          wave_lds_sync();  // (the line buffer is free: the inline entries / the previous round are done)
          if (c0) {
            list_lines(c0);
            issue64(E, lane < (total - c0 < 64u ? total - c0 : 64u) ? s_oln[lane] : WB_NONE, oa, ob);
          }
          const uint32_t m = total - c0 < 64u ? total - c0 : 64u;
          land64(oa, ob);
          wave_lds_sync();
          SPF(5)
          const uint32_t owner = lane < m ? (uint32_t)s_own[lane] : 0u;
          const uint32_t ocn = lane < m ? (uint32_t)s_ocn[lane] : 0u;
          const uint32_t orfl = s_rfl[owner], olb = s_lenbud[owner];
          uint32_t okm = 0, zm = 0;
#pragma unroll
          for (int s = 0; s < 8; s++) {
            const uint4 ent = s_line[rb ^ (uint32_t)s];
            uint32_t z = 0;
            bool ok = screen_entry_ok(ent, q1, ww, orfl, olb, &z) & ((uint32_t)s < ocn);
#if defined(SCRT_DBG) && (SCRT_DBG & 1)
            ok = false;
#endif
            okm |= ok ? 1u << s : 0u;
            zm |= (ok ? z : 0u) << s;
          }
          append_all(okm, zm, 0u, sub * 64 + owner, k, q1);
          SPF(6)
        }
        wave_lds_sync();  // the next step rewrites the line buffer and the per-read tables
#ifdef SCRT_PROF
        pf_steps++;
#endif
        b_cur = b_nx;
      }
      if (active) rvalid[i] = valid;
      rec = rec_nx;
    }
    const bool fits = tilecnt <= room;  // else: the host grows desc and repeats the batch
    used += tilecnt;
    if (lane == 0) {
      tbase[tile] = (uint32_t)base;
      tcount[tile] = fits ? tilecnt : 0u;
    }
  }
#ifdef SCRT_PROF
  if ((blockIdx.x == 0 || blockIdx.x == 1001) && lane == 0 && pf_steps > 8)
    printf("wave %u: %u steps, cycles/step: total %llu | between %llu land(wait) %llu next-bucket+issue %llu hdr+list+ovf-issue %llu inline tests+append %llu ovf land(wait) %llu ovf tests+append %llu\n",
           blockIdx.x, pf_steps, (__builtin_amdgcn_s_memtime() - pstart) / pf_steps, pf[0] / pf_steps, pf[1] / pf_steps, pf[2] / pf_steps,
           pf[3] / pf_steps, pf[4] / pf_steps, pf[5] / pf_steps, pf[6] / pf_steps);
#endif
  // (a wave is a workgroup here: one reduction per wave and a handful of atomics from its first lane)
  const unsigned long long v0 = nvalid, v1 = ncand;
  if (lane == 0) {
    if (v0) atomicAdd(&counters[0], v0);
    if (v1) atomicAdd(&counters[3], v1);
    atomicAdd(&counters[4], (unsigned long long)used);
    atomicMax(&counters[7], (unsigned long long)used);
    if (used > region) atomicOr(pass_flags, 1ull);  // pass-level flag: descriptor space ran out
  }
}
