// pup_wide_tu.hip — one of the translation units holding the instantiations of the wide-window staged kernel K1w
// (pup_wide.hpp): the four kernels (observed over expected x factorised counts) of ONE lane shape (cells per lane, column chunks per row).
// Compiled with -DPUP_TU_PART=shape (0..8, see wide_shape_ch / wide_shape_nch); gfx950 only.
#include "../../include/pup_hip.h"
#define PUP_KERNEL static __global__      // the headers' plain kernels belong to the engine's unit
#include "pup_wide.hpp"

#ifndef PUP_TU_PART
#error "compile with -DPUP_TU_PART=0..8"
#endif

namespace pup {

#define PUP_PASTE_(a, b) a##b
#define PUP_PASTE(a, b) PUP_PASTE_(a, b)
bool PUP_PASTE(launch_wide_part, PUP_TU_PART)(const K1Args& a, const WideArgs& wa, int G, bool ooe, bool fact, hipStream_t s) {
    constexpr int CH = wide_shape_ch(PUP_TU_PART), NCH = wide_shape_nch(PUP_TU_PART);
    const dim3 grid((unsigned)G), block(kWave * kWideWaves);
    if (ooe) {
        if (fact) hipLaunchKernelGGL((pileup_wide_kernel<CH, NCH, true, true>), grid, block, 0, s, a, wa);
        else      hipLaunchKernelGGL((pileup_wide_kernel<CH, NCH, true, false>), grid, block, 0, s, a, wa);
    } else {
        if (fact) hipLaunchKernelGGL((pileup_wide_kernel<CH, NCH, false, true>), grid, block, 0, s, a, wa);
        else      hipLaunchKernelGGL((pileup_wide_kernel<CH, NCH, false, false>), grid, block, 0, s, a, wa);
    }
    return true;
}

}  // namespace pup
