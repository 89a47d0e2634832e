// k_match_t is compiled in translation units of its own (match_lane_rw*.hip, one per record stride:
// twelve instantiations take minutes in one unit and build side by side in three).  muscato_hip.hip,
// which launches them, sees only this declaration and the list of instances.
#pragma once
#ifndef MATCHT_WAVES
#define MATCHT_WAVES 2  // waves per SIMD the register allocator leaves room for (256 VGPRs)
#endif
#define MATCHT_WLIST 96  // reported candidates of a wave-tile kept in LDS (cfg3: ~53); more spill to HBM
template <int RW, int W, bool RX>
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
#define MUSC_LANE_INSTANCES(X, RW) \
  X template __global__ void k_match_t<RW, 1, false> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 2, false> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 3, false> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 4, false> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 1, true> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 2, true> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 3, true> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 4, true> MUSC_LANE_ARGS;
