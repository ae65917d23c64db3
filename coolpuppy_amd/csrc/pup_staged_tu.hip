// pup_staged_tu.hip — one of the kStagedParts translation units holding the instantiations of the workgroup-staged kernel
// (see pup_staged_launch.hpp).  Compiled with -DPUP_TU_PART=k; gfx950 only.
#include "../../include/pup_hip.h"
#define PUP_KERNEL static __global__      // the headers' plain kernels belong to the engine's unit
#include "pup_staged_launch.hpp"

#ifndef PUP_TU_PART
#error "compile with -DPUP_TU_PART=0..7"
#endif

namespace pup {
namespace {

template <int W, bool OOE, int ACC, bool FACT, bool EXTRA>
void launch_kernel(const StagedLaunch& l, const K1Args& a, const StagedArgs& sa, hipStream_t s) {
    using Geo = StagedGeom<W, OOE, EXTRA, false, FACT>;
    if constexpr (!EXTRA && FACT && !OOE && Geo::big && W == 21) {      // (an experiment kept for the record: instantiated for the bench's width only)
        if (l.band && l.prog && !l.small21) {                // progressive staging (pup_staged.hpp: PROG), tuning bit 22
            hipLaunchKernelGGL((pileup_staged_kernel<W, OOE, Geo::RSR, Geo::RSC, Geo::NW, ACC, FACT, EXTRA, true, false, true>), dim3(l.G),
                               dim3(kWave * Geo::NW), 0, s, a, sa);
            return;
        }
    }
    if constexpr (!EXTRA) {
        if (l.band && !(W == 21 && !OOE && l.small21)) {
            hipLaunchKernelGGL((pileup_staged_kernel<W, OOE, Geo::RSR, Geo::RSC, Geo::NW, ACC, FACT, EXTRA, true>), dim3(l.G),
                               dim3(kWave * Geo::NW), 0, s, a, sa);
            return;
        }
    }
    if constexpr (W == 21 && !OOE && !EXTRA && ACC <= 2) {
        if (l.small21) {
            using GeoS = StagedGeom<W, OOE, EXTRA, true, FACT>;
            if constexpr (FACT) {
                if (l.band) {                                // two buffers of 64 rows, sixteen waves (pup_staged.hpp: DB)
                    hipLaunchKernelGGL((pileup_staged_kernel<W, OOE, GeoS::RSR, GeoS::RSC, GeoS::NW, ACC, FACT, EXTRA, true, true>), dim3(l.G),
                                       dim3(kWave * GeoS::NW), 0, s, a, sa);
                    return;
                }
            }
            if (l.band)
                hipLaunchKernelGGL((pileup_staged_kernel<W, OOE, GeoS::RSR, GeoS::RSC, GeoS::NW, ACC, FACT, EXTRA, true>), dim3(l.G),
                                   dim3(kWave * GeoS::NW), 0, s, a, sa);
            else
                hipLaunchKernelGGL((pileup_staged_kernel<W, OOE, GeoS::RSR, GeoS::RSC, GeoS::NW, ACC, FACT, EXTRA>), dim3(l.G),
                                   dim3(kWave * GeoS::NW), 0, s, a, sa);
            return;
        }
    }
    hipLaunchKernelGGL((pileup_staged_kernel<W, OOE, Geo::RSR, Geo::RSC, Geo::NW, ACC, FACT, EXTRA>), dim3(l.G),
                       dim3(kWave * Geo::NW), 0, s, a, sa);
}

template <int W, int ACC, bool EXTRA>
bool pick_mode(const StagedLaunch& l, const K1Args& a, const StagedArgs& sa, hipStream_t s) {
    if (a.mode & PUP_MODE_OOE) {
        if constexpr (!EXTRA) { if (l.fact) launch_kernel<W, true, ACC, true, EXTRA>(l, a, sa, s); else launch_kernel<W, true, ACC, false, EXTRA>(l, a, sa, s); }
        else launch_kernel<W, true, ACC, false, EXTRA>(l, a, sa, s);
    } else if (l.fact) launch_kernel<W, false, ACC, true, EXTRA>(l, a, sa, s);
    else launch_kernel<W, false, ACC, false, EXTRA>(l, a, sa, s);
    return true;
}

template <int W>
bool pick(const StagedLaunch& l, const K1Args& a, const StagedArgs& sa, hipStream_t s) {
    if (l.slots == 8) {                                  // sets of four tile pairs: plain pile-ups on the big regions only
        if constexpr (staged_big<W>()) { if (!l.extra && !l.small21) return pick_mode<W, 8, false>(l, a, sa, s); }
        return false;
    }
    if (l.extra) return l.slots == 2 ? pick_mode<W, 2, true>(l, a, sa, s) : pick_mode<W, 1, true>(l, a, sa, s);
    return l.slots == 2 ? pick_mode<W, 2, false>(l, a, sa, s) : pick_mode<W, 1, false>(l, a, sa, s);
}

}  // namespace

#define PUP_PASTE_(a, b) a##b
#define PUP_PASTE(a, b) PUP_PASTE_(a, b)
bool PUP_PASTE(launch_staged_part, PUP_TU_PART)(const StagedLaunch& l, const K1Args& a, const StagedArgs& sa, hipStream_t s) {
    constexpr int W0 = 3 + 4 * PUP_TU_PART, W1 = W0 + 2;
    if (l.W == W0) return pick<W0>(l, a, sa, s);
    if constexpr (W1 <= 31) { if (l.W == W1) return pick<W1>(l, a, sa, s); }
    return false;
}

}  // namespace pup
