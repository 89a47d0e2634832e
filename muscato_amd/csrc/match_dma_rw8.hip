// k_match_g's instances (kernels_match_dma.hpp): their own translation unit
#include <hip/hip_runtime.h>
#define MUSC_KERNEL static __global__
#include <type_traits>
#include "../../include/muscato_hip.h"
#include "kernels_common.hpp"
#include "kernels_index.hpp"
#include "kernels_screen.hpp"
#include "kernels_match.hpp"
#include "kernels_match_lane.hpp"
#include "kernels_match_dma.hpp"
MUSC_DMA_INSTANCES()
