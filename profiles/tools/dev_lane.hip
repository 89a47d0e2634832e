// development translation unit: ONE instantiation of k_match_t, for quick resource / ISA checks
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -DDEV_RW=8 -DDEV_W=2 profiles/tools/dev_lane.hip
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/muscato_hip.h"
#include "../../muscato_amd/csrc/kernels_common.hpp"
#include "../../muscato_amd/csrc/kernels_index.hpp"
#include "../../muscato_amd/csrc/kernels_screen.hpp"
#include "../../muscato_amd/csrc/kernels_match.hpp"
#include "../../muscato_amd/csrc/kernels_match_lane.hpp"
#ifndef DEV_RW
#define DEV_RW 8
#endif
#ifndef DEV_W
#define DEV_W 2
#endif
#include "../../muscato_amd/csrc/kernels_match_lane_inst.hpp"
#ifndef DEV_RX
#define DEV_RX false
#endif
#ifndef DEV_WIDE
#define DEV_WIDE false
#endif
#ifndef DEV_SG
#define DEV_SG 0  // 1: the geometry-specialised instance (SpecGeom<1>)
#endif
template __global__ void k_match_t<DEV_RW, DEV_W, DEV_RX, DEV_WIDE, DEV_SG> MUSC_LANE_ARGS;
