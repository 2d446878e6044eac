// Argument block of one 4-centre (l_i >= l_j | l_k >= l_l) class launch (plain C layout, mirrored by PAMD_int2e_args in
// include/pyscf_amd.h and by ctypes in pyscf_amd/scf/_vhf.py).
#pragma once
extern "C" {
typedef struct PAMD_int2e_args {
    // bra shell pairs (same records as PAMD_int3c2e_args: zeta, P, K_ab c_i c_j, P - A)
    const int *bra_ish;
    const int *bra_jsh;
    const int *bra_pp0;
    const int *bra_npp;
    const double *bra_pp;
    // ket shell pairs
    const int *ket_ish;
    const int *ket_jsh;
    const int *ket_pp0;
    const int *ket_npp;
    const double *ket_pp;
    const double *shell_xyz;    // [nshell][3]
    const int *shell_ao0;       // [nshell]
    const double *rys_table;
    const double *c2s;
    const int *c2s_off;
    double *eri;                // [nao][nao][nao][nao], all 8 permutational images are written
    int nbra, nket;
    int li, lj, lk, ll;
    int nao;
    int same_class;             // bra and ket lists are the same list: only ket <= bra is computed
    double omega;               // > 0: erf(omega r12)/r12; 0: 1/r12
} PAMD_int2e_args;
}
namespace pamd { typedef PAMD_int2e_args Int2eArgs; }
