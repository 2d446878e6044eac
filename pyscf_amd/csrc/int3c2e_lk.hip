// Instantiates the int3c2e kernel family for one aux angular momentum (compile with
// -DPAMD_LK=<0..6>); see int3c2e_kernel.h.
#include "int3c2e_kernel.h"

#ifndef PAMD_LK
#error "compile with -DPAMD_LK=<l_aux>"
#endif

namespace pamd {

#define PAMD_CAT2(a, b) a##b
#define PAMD_CAT(a, b) PAMD_CAT2(a, b)

int PAMD_CAT(launch_int3c2e_lk, PAMD_LK)(int li, int lj, const Int3c2eArgs &a, hipStream_t st)
{
    constexpr int LK = PAMD_LK;
    switch (li * 8 + lj) {
    case 0 * 8 + 0: return launch_class<0, 0, LK>(a, st);
    case 1 * 8 + 0: return launch_class<1, 0, LK>(a, st);
    case 1 * 8 + 1: return launch_class<1, 1, LK>(a, st);
    case 2 * 8 + 0: return launch_class<2, 0, LK>(a, st);
    case 2 * 8 + 1: return launch_class<2, 1, LK>(a, st);
    case 2 * 8 + 2: return launch_class<2, 2, LK>(a, st);
    case 3 * 8 + 0: return launch_class<3, 0, LK>(a, st);
    case 3 * 8 + 1: return launch_class<3, 1, LK>(a, st);
    case 3 * 8 + 2: return launch_class<3, 2, LK>(a, st);
    case 3 * 8 + 3: return launch_class<3, 3, LK>(a, st);
    case 4 * 8 + 0: return launch_class<4, 0, LK>(a, st);   // also the 2-centre (P|Q) with l_P = 4
#if PAMD_LK <= 5              // g AO shells go with fitting shells up to h (quadruple-zeta sets); i fitting shells (def2 3d metals) with AO l <= 3
    case 4 * 8 + 1: return launch_class<4, 1, LK>(a, st);
    case 4 * 8 + 2: return launch_class<4, 2, LK>(a, st);
    case 4 * 8 + 3: return launch_class<4, 3, LK>(a, st);
    case 4 * 8 + 4: return launch_class<4, 4, LK>(a, st);
#endif
    case 5 * 8 + 0: return launch_class<5, 0, LK>(a, st);   // 2-centre (P|Q) with l_P = 5
    case 6 * 8 + 0: return launch_class<6, 0, LK>(a, st);   // 2-centre (P|Q) with l_P = 6
    default:
        return set_error(-2, "int3c2e: unsupported (l_i, l_j) class (AO l <= 4 with aux l <= 5, AO l <= 3 with aux l = 6)", __FILE__, __LINE__);
    }
}

}  // namespace pamd
