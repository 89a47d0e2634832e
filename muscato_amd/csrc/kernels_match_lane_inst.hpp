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
// A kernel instance may be compiled for ONE run geometry (SG != 0): WindowWidth, window starts, context offset,
// MinDinuc and the table kind become compile-time constants -- the mask tables turn into immediates, the image
// shift into a constant, a third of the scalar instructions and nine tenths of the spilled scalars go away
// (profiles/r03_spec_variant.txt: 3-4 % of the launch).  SpecGeom<SG> names the geometry; the host launches such an
// instance ONLY after comparing every one of these quantities with the run's (spec_geom_matches, muscato_hip.hip),
// and the kernel checks them again at entry and refuses to touch the table on a mismatch (flag 8 of counters[3]).
template <int SG>
struct SpecGeom {  // SG = 0: the general kernel, everything from MatchParams
  static constexpr bool on = false;
  static constexpr int ww = 0, CL = 0, L = 0, min_dinuc = 0, nwin = 0;
  static constexpr int win[CTX_MAX_W] = {0, 0, 0, 0};
};
template <>
struct SpecGeom<1> {  // BASELINE configs 2-4: WindowWidth 15, Windows 0,20, 100-base reads, MinDinuc 5, a direct table
  static constexpr bool on = true;
  static constexpr int ww = 15, CL = 20, L = 100, min_dinuc = 5, nwin = 2;
  static constexpr int win[CTX_MAX_W] = {0, 20, 0, 0};
};
#define MUSC_SPEC_GEOMS 1  // geometries 1 .. MUSC_SPEC_GEOMS exist

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
                                                                const uint32_t* __restrict__ rdx);
// k_match_g (kernels_match_dma.hpp): the same kernel at three waves per SIMD -- lines, overflow entries and records by
// LDS-DMA -- for two windows on 120-base buckets without X (BASELINE configs 2-4); general (SG = 0) and specialised
// waves per SIMD the register allocator leaves room for: four (128 VGPRs) for a geometry-specialised instance, three
// (168) for the general one, which would spill four registers into scratch at 128; LDS allows four either way
#ifndef MATCHG_WAVES_OF
#define MATCHG_WAVES_OF(SG) ((SG) ? 4 : 3)
#endif
#ifndef MATCHG_SKETCH_BITS
#define MATCHG_SKETCH_BITS 9  // cells of the workgroup's MaxMatches sketch (2 KB: four workgroups of 38 KB fit a CU)
#endif
#ifndef MATCHG_WLIST
#define MATCHG_WLIST 64  // reported candidates of a wave-tile kept in LDS (cfg3: ~53); more spill to HBM
#endif
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
                                                                const uint32_t* __restrict__ rdx);
#define MUSC_DMA_ARGS                                                                                                  \
  (const uint32_t*, uint64_t, uint32_t, const MatchParams*, const uint16_t*, const CtxBucket*, const CtxEntry*, uint4*, \
   uint64_t, uint4*, uint64_t, uint32_t*, uint32_t*, int, uint32_t, uint32_t*, unsigned long long*, const uint4*,   \
   const uint32_t*, const uint32_t*, uint32_t, uint4*, uint64_t, const uint32_t*)
#define MUSC_DMA_INSTANCES(X)                            \
  X template __global__ void k_match_g<8, 0> MUSC_DMA_ARGS; \
  X template __global__ void k_match_g<8, 1> MUSC_DMA_ARGS;
#define MUSC_LANE_ARGS                                                                                              \
  (const uint32_t*, uint64_t, uint32_t, const MatchParams*, const uint16_t*, const CtxBucket*, const CtxEntry*, uint4*, \
   uint64_t, uint4*, uint64_t, uint32_t*, uint32_t*, int, uint32_t, uint32_t*, unsigned long long*, const uint4*,   \
   const uint32_t*, const uint32_t*, uint32_t, uint4*, uint64_t, const uint32_t*)
#define MUSC_LANE_INSTANCES_WD(X, RW, WD)                                \
  X template __global__ void k_match_t<RW, 1, 0, WD, 0> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 2, 0, WD, 0> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 3, 0, WD, 0> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 4, 0, WD, 0> MUSC_LANE_ARGS; \
  X template __global__ void k_match_t<RW, 1, 1, WD, 0> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 2, 1, WD, 0> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 3, 1, WD, 0> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 4, 1, WD, 0> MUSC_LANE_ARGS;  \
  X template __global__ void k_match_t<RW, 1, 2, WD, 0> MUSC_LANE_ARGS;     \
  X template __global__ void k_match_t<RW, 2, 2, WD, 0> MUSC_LANE_ARGS;     \
  X template __global__ void k_match_t<RW, 3, 2, WD, 0> MUSC_LANE_ARGS;     \
  X template __global__ void k_match_t<RW, 4, 2, WD, 0> MUSC_LANE_ARGS;
// 120-base context buckets: records of 4, 8, 12 words; wide (200 bases): 4 (distant windows), 8, 12, 16.
// One translation unit per line.
#define MUSC_LANE_INSTANCES_4(X) MUSC_LANE_INSTANCES_WD(X, 4, false) MUSC_LANE_INSTANCES_WD(X, 4, true)
#define MUSC_LANE_INSTANCES_8(X) MUSC_LANE_INSTANCES_WD(X, 8, false)
#define MUSC_LANE_INSTANCES_12(X) MUSC_LANE_INSTANCES_WD(X, 12, false)
#define MUSC_LANE_INSTANCES_8W(X) MUSC_LANE_INSTANCES_WD(X, 8, true)
#define MUSC_LANE_INSTANCES_12W(X) MUSC_LANE_INSTANCES_WD(X, 12, true)
#define MUSC_LANE_INSTANCES_16W(X) MUSC_LANE_INSTANCES_WD(X, 16, true)
// the geometry-specialised instances (one translation unit: match_lane_rw8s.hip)
#define MUSC_LANE_INSTANCES_SPEC(X) X template __global__ void k_match_t<8, 2, 0, false, 1> MUSC_LANE_ARGS;
