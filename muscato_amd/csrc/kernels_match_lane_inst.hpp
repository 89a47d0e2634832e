// k_match_t is compiled in translation units of its own (match_lane_rw*.hip, one per record stride:
// twelve instantiations take minutes in one unit and build side by side in three).  muscato_hip.hip,
// which launches them, sees only this declaration and the list of instances.
#pragma once
#ifndef MATCHT_WAVES
#define MATCHT_WAVES 2  // waves per SIMD the register allocator leaves room for (256 VGPRs)
#endif
#ifndef MATCHT_WLIST
#define MATCHT_WLIST 96  // reported candidates of a wave-tile kept in LDS (cfg3: ~53); more spill to HBM
#endif
template <int RW, int W, int XM, bool WIDE>
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
                                                                const uint32_t* __restrict__ rdx);
#define MUSC_LANE_ARGS                                                                                              \
  (const uint32_t*, uint64_t, uint32_t, const MatchParams*, const uint16_t*, const CtxBucket*, const CtxEntry*, uint4*, \
   uint64_t, uint4*, uint64_t, uint32_t*, uint32_t*, int, uint32_t, uint32_t*, unsigned long long*, const uint4*,   \
   const uint32_t*, const uint32_t*, uint32_t, uint4*, uint64_t, const uint32_t*)
#define MUSC_LANE_INSTANCES_WD(X, RW, WD)                                \
  X template __global__ void k_match_t<RW, 1, 0, WD> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 2, 0, WD> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 3, 0, WD> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 4, 0, WD> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 1, 1, WD> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 2, 1, WD> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 3, 1, WD> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 4, 1, WD> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 1, 2, WD> MUSC_LANE_ARGS;     \
  X template __global__ void k_match_t<RW, 2, 2, WD> MUSC_LANE_ARGS;     \
  X template __global__ void k_match_t<RW, 3, 2, WD> MUSC_LANE_ARGS;     \
  X template __global__ void k_match_t<RW, 4, 2, WD> MUSC_LANE_ARGS;
// 120-base context buckets: records of 4, 8, 12 words; wide (200 bases): 4 (distant windows), 8, 12, 16.
// One translation unit per line.
#define MUSC_LANE_INSTANCES_4(X) MUSC_LANE_INSTANCES_WD(X, 4, false) MUSC_LANE_INSTANCES_WD(X, 4, true)
#define MUSC_LANE_INSTANCES_8(X) MUSC_LANE_INSTANCES_WD(X, 8, false)
#define MUSC_LANE_INSTANCES_12(X) MUSC_LANE_INSTANCES_WD(X, 12, false)
#define MUSC_LANE_INSTANCES_8W(X) MUSC_LANE_INSTANCES_WD(X, 8, true)
#define MUSC_LANE_INSTANCES_12W(X) MUSC_LANE_INSTANCES_WD(X, 12, true)
#define MUSC_LANE_INSTANCES_16W(X) MUSC_LANE_INSTANCES_WD(X, 16, true)
