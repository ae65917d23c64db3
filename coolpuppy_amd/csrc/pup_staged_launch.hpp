// pup_staged_launch.hpp — how the engine reaches the instantiations of the workgroup-staged kernel (pup_staged.hpp).
// They are compiled in kStagedParts translation units (pup_staged_tu.hip with -DPUP_TU_PART=k: window widths 3 + 4k and
// 5 + 4k), side by side: one unit with every width took seven minutes of hipcc, eight units take one.
#pragma once
#include "pup_staged.hpp"

namespace pup {

struct StagedLaunch {
    int  W;              // window width (odd, 3..31)
    int  G;              // persistent workgroups
    int  slots;          // accumulator slots of a pass: 1 (a tile), 2 (a tile pair), 8 (four pairs; plain pile-ups up to W = 21)
    bool fact;           // every window clear of the diagonal mask, nothing divided by expected: validity factorises
    bool extra;          // coverage vectors and / or pixel statistics ride along
    bool small21;        // tuning probe: the plain 21-bin kernel on 64 x 128 regions
    bool band;           // regions staged from the dense band of counts
    bool prog;           // progressive staging where the instantiation exists (band, factorised counts, no expected, 128 x 128 regions): no barrier between blocks
};

constexpr int kStagedParts = 8;
#define PUP_STAGED_PART_DECL(k) bool launch_staged_part##k(const StagedLaunch&, const K1Args&, const StagedArgs&, hipStream_t);
PUP_STAGED_PART_DECL(0) PUP_STAGED_PART_DECL(1) PUP_STAGED_PART_DECL(2) PUP_STAGED_PART_DECL(3)
PUP_STAGED_PART_DECL(4) PUP_STAGED_PART_DECL(5) PUP_STAGED_PART_DECL(6) PUP_STAGED_PART_DECL(7)
#undef PUP_STAGED_PART_DECL

// false: no instantiation for this width / slot count
inline bool launch_staged(const StagedLaunch& l, const K1Args& a, const StagedArgs& sa, hipStream_t s) {
    if (l.W < 3 || l.W > 31 || !(l.W & 1)) return false;
    switch ((l.W - 3) / 4) {
        case 0: return launch_staged_part0(l, a, sa, s);
        case 1: return launch_staged_part1(l, a, sa, s);
        case 2: return launch_staged_part2(l, a, sa, s);
        case 3: return launch_staged_part3(l, a, sa, s);
        case 4: return launch_staged_part4(l, a, sa, s);
        case 5: return launch_staged_part5(l, a, sa, s);
        case 6: return launch_staged_part6(l, a, sa, s);
        default: return launch_staged_part7(l, a, sa, s);
    }
}

}  // namespace pup
