// Argument block of one int3c2e class launch (plain C layout, mirrored by
// PAMD_int3c2e_args in include/pyscf_amd.h and by ctypes in pyscf_amd/gto/moleintor.py).
#pragma once
extern "C" {
typedef struct PAMD_int3c2e_args {
    // shell pairs of this class
    const int *pair_ish;        // [npairs] shell a (l = LI)
    const int *pair_jsh;        // [npairs] shell b (l = LJ)
    const int *pair_pp0;        // [npairs] first primitive-pair record
    const int *pair_npp;        // [npairs] number of primitive-pair records
    const double *pp;           // [npp_total][8]: zeta, Px,Py,Pz, cc, PAx,PAy,PAz
    const double *shell_xyz;    // [nshell_ao][3]
    const int *shell_ao0;       // [nshell_ao] first AO function of the shell
    // aux shells of class LK (sorted list)
    const int *aux_f0;          // [naux_cls] first aux function index (column of T)
    const double *aux_xyz;      // [naux_cls][3]
    const double *aux_exp;      // [naux_cls][npk]
    const double *aux_coef;     // [naux_cls][npk]  (zero padded)
    int naux_cls;
    int npk;                    // primitives per aux shell (padded, uniform over the class)
    const double *rys_table;    // Chebyshev table (device copy of RYS_TABLE)
    // cart->sph matrices: c2s[l] = [(2l+1)][ncart(l)] at c2s + c2s_off[l]
    const double *c2s;
    const int *c2s_off;
    // output
    double *T;
    long ldT;
    long row_offset;            // subtracted from the packed-tril row index
    int tril;                   // 1: rows are packed-tril AO pairs; 0: row = ao0_i + fi (2-centre)
    int npairs;
    double omega;               // > 0: long-range operator erf(omega r12)/r12 (env[PTR_RANGE_OMEGA]); 0: Coulomb
} PAMD_int3c2e_args;
}
extern "C" {
// Argument block of one gradient-contraction class launch (int3c2e_grad_kernel.h).
typedef struct PAMD_int3c2e_grad_args {
    PAMD_int3c2e_args base;      // base.T = Z (read only): same row/column addressing as the integral kernel
    const double *pp_ab;         // [npp_total][2]: primitive exponents (alpha_i, alpha_j) of every primitive-pair record
    const int *shell_atom;       // [nshell_ao]  atom of each AO-side shell
    const int *aux_atom;         // [naux_cls]   atom of each aux shell of the class
    double *grad;                // [nrep][natm][3] accumulators (atomicAdd; the caller sums the replicas)
    int nrep;
    int natm;
    int aux_response;            // 0: leave out the derivative of the third centre
} PAMD_int3c2e_grad_args;
}

namespace pamd { typedef PAMD_int3c2e_args Int3c2eArgs; typedef PAMD_int3c2e_grad_args Int3c2eGradArgs; }
