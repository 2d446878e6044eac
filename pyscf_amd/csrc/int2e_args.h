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
// Integral-direct use of the same kernel (pyscf/scf/_vhf.py:370-429 direct -> CVHFnr_direct_drv, lib/vhf/nr_direct.c:361-489):
// base.eri is ignored; either q_out (Schwarz pass, bra list == ket list: q_out[pair] = sqrt(max |(ij|ij)|), the q_cond of
// lib/vhf/optimizer.c:90-117) or vj / vk (contraction with nset densities, the role of CVHFdot_nrs8, nr_direct.c:183-231).
typedef struct PAMD_int2e_direct_args {
    PAMD_int2e_args base;
    const double *dm;           // [nset][nao][nao]
    double *vj;                 // [nset][nao][nao] += ; nullable
    double *vk;                 // [nset][nao][nao] += ; nullable
    int nset;
    const double *q_bra;        // [nbra] Schwarz factors of the bra pairs (nullable: no screening)
    const double *q_ket;        // [nket]
    double cutoff;              // skip a shell quartet when q_bra q_ket dm_max < cutoff (direct_scf_tol)
    double dm_max;              // max |dm|
    double *q_out;              // [nbra]: Schwarz pass
} PAMD_int2e_direct_args;
}
namespace pamd { typedef PAMD_int2e_args Int2eArgs; typedef PAMD_int2e_direct_args Int2eDirectArgs; }
