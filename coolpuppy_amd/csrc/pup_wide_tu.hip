// pup_wide_tu.hip — one of the translation units holding the instantiations of the wide-window staged kernel K1w
// (pup_wide.hpp): the four kernels (observed over expected x factorised counts) of ONE cells-per-lane value.
// Compiled with -DPUP_TU_PART=CH (7..13); gfx950 only.
#include "../../include/pup_hip.h"
#define PUP_KERNEL static __global__      // the headers' plain kernels belong to the engine's unit
#include "pup_wide.hpp"

#ifndef PUP_TU_PART
#error "compile with -DPUP_TU_PART=7..13"
#endif

namespace pup {

#define PUP_PASTE_(a, b) a##b
#define PUP_PASTE(a, b) PUP_PASTE_(a, b)
bool PUP_PASTE(launch_wide_part, PUP_TU_PART)(const K1Args& a, const WideArgs& wa, int G, bool ooe, bool fact, hipStream_t s) {
    constexpr int CH = PUP_TU_PART;
    const dim3 grid((unsigned)G), block(kWave * 16);
    if (ooe) {
        if (fact) hipLaunchKernelGGL((pileup_wide_kernel<CH, true, true>), grid, block, 0, s, a, wa);
        else      hipLaunchKernelGGL((pileup_wide_kernel<CH, true, false>), grid, block, 0, s, a, wa);
    } else {
        if (fact) hipLaunchKernelGGL((pileup_wide_kernel<CH, false, true>), grid, block, 0, s, a, wa);
        else      hipLaunchKernelGGL((pileup_wide_kernel<CH, false, false>), grid, block, 0, s, a, wa);
    }
    return true;
}

}  // namespace pup
