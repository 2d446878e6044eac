/*
 * pyscf_amd.h - C ABI of libpyscf_amd.so (MI355X / gfx950 density-fitted Fock-build engine).
 *
 * Drop-in boundary: these are the entry points a PySCF maintainer would bind with ctypes in
 * place of the reference's CPU entry points on the DF J/K hot path.  Every function
 *   - is extern "C", takes plain pointers and sizes (no torch / C++ types),
 *   - takes DEVICE pointers (d_ prefix) owned by the caller, plus the HIP stream to launch on
 *     (hipStream_t passed as void*; NULL = default stream); launches are asynchronous,
 *   - returns 0 on success or a negative error code (never exits); PAMD_last_error() gives the
 *     message of the last failure in the calling thread.
 *
 * Reference interfaces replaced (paths relative to the reference tree pyscf/):
 *   PAMD_int3c2e_class   lib/gto/fill_nr_3c.c:196-225 GTOnr3c_drv + :127-185 GTOnr3c_fill_s2ij
 *                        (+ libcint int3c2e_sph), and lib/gto/fill_int2c.c GTOint2c (int2c2e_sph)
 *   PAMD_int2e_class     lib/gto/fill_int2e.c:538 GTOnr2e_fill_drv (+ libcint int2e_sph) as called through
 *                        mol.intor('int2e') by RHF.get_jk, scf/hf.py:2499-2511 (in-core 4-centre path, config 1)
 *   PAMD_int1e_ovlp_kin  lib/gto/fill_int2c.c GTOint2c with int1e_ovlp_sph / int1e_kin_sph
 *   PAMD_cderi_solve     df/incore.py:204-213 (BLAS trsm), :216 (lib.dot, eig fallback)
 *   PAMD_pack_dm_tril    df/df_jk.py:329-332 (lib.pack_tril + halved diagonal; lib/np_helper/pack_tril.c:59-112)
 *   PAMD_df_vj_pass1/2   df/df_jk.py:367 `vj += dmtril.dot(eri1.T).dot(eri1)`
 *   PAMD_nr_e2_symm      lib/ao2mo/nr_ao2mo.c:1240-1266 AO2MOnr_e2_drv with
 *                        ftrans = AO2MOtranse2_nr_s2 (:1026-1031), fmmm = AO2MOmmm_bra_nr_s2 (:399-419)
 *   PAMD_dgemm_tn        lib/np_helper/npdot.c:32 NPdgemm as used by lib.dot(buf1.T, buf1), df/df_jk.py:380,407
 *   PAMD_unpack_tril     lib/np_helper/pack_tril.c:150-273 NPdunpack_tril_2d
 *   PAMD_becke_partition lib/dft/grid_basis.c:32-101 VXCgen_grid
 *   PAMD_grid_partition  dft/gen_grid.py:341-419 get_partition with becke_scheme = original_becke / stratmann (:203-212) /
 *                        becke_lko (lib/dft/grid_basis.c:266-384 VXCgen_grid_lko)
 *   PAMD_eval_ao         gto/eval_gto.py:31-144 -> lib/gto/grid_ao_drv.c:222-284,415-459 (GTOval_sph_deriv0/1)
 *   PAMD_rho_from_mo/_dm dft/numint.py:116-469 eval_rho / eval_rho2 (VXCdot_ao_dm, VXCdcontract_rho)
 *   PAMD_eval_xc         lib/dft/libxc_itrf.c:968-1024 LIBXC_eval_xc (+ libxc 7.1.2) and dft/xc_deriv.py:32-85
 *   PAMD_eval_fxc        the fxc part of the same (deriv = 2) contracted with a first-order density, dft/numint.py:1418-1576
 *   PAMD_scale_ao        dft/numint.py:803-834 (VXCdscale_ao_sparse, lib/dft/nr_numint_sparse.c:1103)
 *   PAMD_dgemm_nt        dft/numint.py:836-874 (VXCdot_ao_ao_sparse, lib/dft/nr_numint_sparse.c:890-973)
 *
 * Array conventions (identical to the reference): cderi is (naux, nao_pair) row-major f64 with
 * pq = p(p+1)/2+q, p >= q (df/df.py:59-72); density matrices are (nset, nao, nao) row-major f64.
 */
#ifndef PYSCF_AMD_H
#define PYSCF_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------ */
const char *PAMD_last_error(void);
int PAMD_version(void);
int PAMD_device_count(void);                 /* 0 when no HIP device / driver is present */
int PAMD_set_device(int dev);
int PAMD_stream_synchronize(void *stream);
/* micro-benchmarks of the FP64 matrix pipe (bench.py / tools/mfma_peak.py; not on the product path): a register-only MFMA stream,
 * and the J/K kernels' k-loop without its memory system (fresh LDS operands every k-group, 20 MFMAs per 9 fragment reads) */
int PAMD_mfma_f64_peak(double *d_out, int nblocks, int iters, int nacc, double scale, void *stream);
int PAMD_mfma_f64_live(double *d_out, int nblocks, int iters, void *stream);

/* ---- integral generation -------------------------------------------------------------------- */
/* Argument block of one (l_i >= l_j | l_k) class launch.  All pointers are device pointers. */
typedef struct PAMD_int3c2e_args {
    const int *pair_ish;        /* [npairs] shell a (l = l_i) of each shell pair                */
    const int *pair_jsh;        /* [npairs] shell b (l = l_j)                                   */
    const int *pair_pp0;        /* [npairs] first primitive-pair record                         */
    const int *pair_npp;        /* [npairs] number of primitive-pair records                    */
    const double *pp;           /* [][8]: zeta, Px,Py,Pz, K_ab c_a c_b, (P-A)x,(P-A)y,(P-A)z     */
    const double *shell_xyz;    /* [nshell][3] centres of the segmented AO shells (Bohr)        */
    const int *shell_ao0;       /* [nshell] first AO function of each shell                     */
    const int *aux_f0;          /* [naux_cls] first aux function of each aux shell of class l_k */
    const double *aux_xyz;      /* [naux_cls][3]                                                */
    const double *aux_exp;      /* [naux_cls][npk]                                              */
    const double *aux_coef;     /* [naux_cls][npk] (zero padded)                                */
    int naux_cls;
    int npk;
    const double *rys_table;    /* device copy of the Rys Chebyshev table (PAMD_rys_table_upload) */
    const double *c2s;          /* cart->sph matrices, c2s + c2s_off[l] = [(2l+1)][ncart(l)]     */
    const int *c2s_off;
    double *T;                  /* output T[row][aux function], leading dimension ldT            */
    long ldT;
    long row_offset;            /* subtracted from the packed-tril row index                     */
    int tril;                   /* 1: rows = packed-tril AO pairs; 0: row = AO index (2-centre)  */
    int npairs;
    double omega;               /* > 0: erf(omega r12)/r12 (range-separated LR, gto/mole.py:76-84 PTR_RANGE_OMEGA); 0: 1/r12 */
} PAMD_int3c2e_args;

long PAMD_rys_table_len(void);
int PAMD_rys_table_upload(double *d_dst, void *stream);
int PAMD_rys_table_host(double *h_dst, int *offsets, int *nint, double *herm_u, double *herm_w);
int PAMD_int3c2e_class(int li, int lj, int lk, const PAMD_int3c2e_args *args, void *stream);

/* Nuclear-gradient contraction, generate-and-contract in place (no derivative tensor):
 *   grad[rep][atom][3] += sum_{pq,Q} Z[row(pq)][Q] d(pq|Q)/dR_atom        for one angular class.
 * Replaces the int3c2e_ip1 / int3c2e_ip2 / int2c2e_ip1 blocks that pyscf/df/grad/rhf.py:117-199 (get_jk) forms with
 * libcint and contracts on the host; with point-charge aux shells also the int1e_ipnuc / int1e_iprinv terms of
 * pyscf/grad/rhf.py:91-146.  The third-centre derivative is -(d/dA + d/dB) (translational invariance). */
typedef struct PAMD_int3c2e_grad_args {
    PAMD_int3c2e_args base;      /* base.T = Z (read only), addressed exactly like the integral output T        */
    const double *pp_ab;         /* [npp_total][2] primitive exponents (alpha_i, alpha_j) of each pp record      */
    const int *shell_atom;       /* [nshell_ao] atom of each AO-side shell                                       */
    const int *aux_atom;         /* [naux_cls]  atom of each aux shell of the class                              */
    double *grad;                /* [nrep][natm][3] FP64 atomicAdd accumulators; the caller sums the replicas   */
    int nrep;
    int natm;
    int aux_response;            /* 0: omit the third-centre derivative (auxbasis_response=False, df/grad/rhf.py:133) */
} PAMD_int3c2e_grad_args;
int PAMD_int3c2e_grad_class(int li, int lj, int lk, const PAMD_int3c2e_grad_args *args, void *stream);

/* 4-centre integrals (ij|kl) of one angular class (l_i >= l_j | l_k >= l_l), AO l <= 3, into the dense
 * d_eri[nao][nao][nao][nao] (all 8 permutational images).  Bra and ket lists are shell-pair tables with the same
 * records as PAMD_int3c2e_args.  In-core path of small molecules: scf/hf.py:2499-2511 builds `_eri` once with
 * mol.intor('int2e') and contracts it with dot_eri_dm (:902) every iteration. */
typedef struct PAMD_int2e_args {
    const int *bra_ish;         /* [nbra] shell i of the pair (l = li)                           */
    const int *bra_jsh;         /* [nbra] shell j (l = lj)                                       */
    const int *bra_pp0;         /* [nbra] first primitive-pair record                            */
    const int *bra_npp;         /* [nbra] number of primitive-pair records                       */
    const double *bra_pp;       /* [..][8]: zeta, Px,Py,Pz, K_ab c_i c_j, PAx,PAy,PAz            */
    const int *ket_ish;
    const int *ket_jsh;
    const int *ket_pp0;
    const int *ket_npp;
    const double *ket_pp;
    const double *shell_xyz;    /* [nshell][3]                                                   */
    const int *shell_ao0;       /* [nshell] first AO function of the shell                       */
    const double *rys_table;
    const double *c2s;
    const int *c2s_off;
    double *eri;                /* [nao][nao][nao][nao]                                          */
    int nbra, nket;
    int li, lj, lk, ll;
    int nao;
    int same_class;             /* bra list == ket list: only ket <= bra is computed             */
    double omega;               /* > 0: erf(omega r12)/r12; 0: 1/r12                             */
} PAMD_int2e_args;
int PAMD_int2e_class(const PAMD_int2e_args *args, void *stream);
/* Integral-direct form of the same kernel family: no tensor, the integrals of every shell quartet are contracted with the
 * densities as they are produced (pyscf/scf/_vhf.py:370-429 direct -> CVHFnr_direct_drv, lib/vhf/nr_direct.c:361-489, with
 * CVHFdot_nrs8 :183-231 and the Schwarz prescreen CVHFnrs8_prescreen, lib/vhf/optimizer.c:90-117).  base.eri is ignored.
 * q_out != NULL: Schwarz pass over the bra list (base.ket_* = base.bra_*), q_out[pair] = sqrt(max |(ij|ij)|). */
typedef struct PAMD_int2e_direct_args {
    PAMD_int2e_args base;
    const double *dm;           /* [nset][nao][nao]                                            */
    double *vj;                 /* [nset][nao][nao] accumulated (FP64 atomics); nullable       */
    double *vk;                 /* [nset][nao][nao] accumulated; nullable                      */
    int nset;
    const double *q_bra;        /* [nbra] Schwarz factors (nullable: no screening)             */
    const double *q_ket;        /* [nket]                                                      */
    double cutoff;              /* skip a quartet when q_bra q_ket dm_max < cutoff             */
    double dm_max;              /* max |dm|                                                    */
    double *q_out;              /* [nbra]: Schwarz pass                                        */
} PAMD_int2e_direct_args;
int PAMD_int2e_direct_class(const PAMD_int2e_direct_args *args, void *stream);
/* d_grad[natm][3] += Tr(Dt dT/dR) - Tr(Ws dS/dR): int1e_ipkin / int1e_ipovlp contractions of
 * pyscf/grad/rhf.py:62-75; Dt, Ws symmetric (nao, nao) */
int PAMD_int1e_grad(const int *d_l, const int *d_ao0, const int *d_prim0, const int *d_nprim,
                    const double *d_xyz, const double *d_exps, const double *d_coefs, const int *d_sh_atom,
                    int nsh, int nao, const double *d_c2s, const int *d_c2s_off, const double *d_Dt,
                    const double *d_Ws, double *d_grad, void *stream);
int PAMD_int1e_ovlp_kin(const int *d_l, const int *d_ao0, const int *d_prim0, const int *d_nprim,
                        const double *d_xyz, const double *d_exps, const double *d_coefs, int nsh,
                        int nao, const double *d_c2s, const int *d_c2s_off, double *d_S, double *d_K,
                        void *stream);

/* cderi[l_off + m][pq] = sum_Q linvT[Q][m] * T[pq][Q]   (m < nL; Q <= l_off + m when triangular) */
int PAMD_cderi_solve(const double *d_linvT, int lda, const double *d_T, long ldT, double *d_cderi,
                     long ldc, int nL, long npq, int naux, int l_off, int triangular, void *stream);

/* integral-direct J (df/df_jk.py:415-506 get_j; screens of lib/vhf/optimizer.c:305-349 replaced by the
 * primitive-pair screening of the pair tables): contract a generated slab T[row][Q] in place */
long PAMD_vj_direct_pass1_worksize(long nrows, int naux);
int PAMD_vj_direct_pass1(const double *d_T, long ldT, long nrows, int naux, const double *d_dmtril_rows,
                         double *d_part, void *stream);                 /* part[chunk][Q] = sum_r T[r][Q] d[r] */
int PAMD_vj_direct_pass2(const double *d_T, long ldT, long nrows, int naux, const double *d_rho,
                         double *d_vj_rows, void *stream);              /* vj[r] = sum_Q T[r][Q] rho[Q]        */

/* ---- J/K contraction ------------------------------------------------------------------------ */
int PAMD_pack_dm_tril(const double *d_dm, int nset, int nao, double *d_tril, void *stream);
long PAMD_df_vj_pass1_worksize(long npair, int naux, int nset);           /* doubles of d_work */
int PAMD_df_vj_pass1(const double *d_cderi, long npair, int naux, const double *d_dmtril, int nset,
                     double *d_rho, double *d_work, void *stream);        /* rho[s][L] = B_L . dmtril_s */
int PAMD_df_vj_pass2(const double *d_cderi, long npair, int naux, const double *d_rho, int nset,
                     double *d_vjtril, void *stream);                     /* vjtril[s] += rho_s^T B  */
/* r06 - the SQUARE layout d_sq[naux][rows][ld] (both triangles of every B_L, rows = ld = round_up(nao, 16), pads zero; lstride =
 * rows * ld) as the ONLY resident copy of the tensor (2x the packed bytes instead of packed + image = 3x): the same two J passes
 * (pyscf/df/df_jk.py:329-337,367) reading the p >= q run of every square row; dmtril / vjtril stay packed */
int PAMD_df_vj_pass1_sq(const double *d_sq, long lstride, int ld, int nao, int naux, const double *d_dmtril, int nset,
                        double *d_rho, double *d_work, void *stream);
int PAMD_df_vj_pass2_sq(const double *d_sq, long lstride, int ld, int nao, int naux, const double *d_rho, int nset,
                        double *d_vjtril, void *stream);
/* one column slab of the build (the AO rows [p0, p1) of every aux row as PAMD_cderi_solve leaves them in d_slab[nL][ncol];
 * pyscf/df/incore.py:189-217) written into both triangles of the square layout */
int PAMD_unpack_tril_slab(const double *d_slab, long ncol, int nL, int p0, int p1, double *d_sq, int ld, long lstride, void *stream);
/* packed rows of the reference's `_cderi` format (pyscf/df/df.py:59-72, lib/np_helper/pack_tril.c:59-112) out of the square layout */
int PAMD_pack_tril_rows(const double *d_sq, long lstride, int ld, int nao, int count, double *d_tril, void *stream);
/* Half transform X[L][i][p] = sum_q B_L[p][q] orb[q][i] (AO2MOnr_e2_drv + AO2MOmmm_bra_nr_s2, nr_ao2mo.c:399-419,1240-1266):
 * d_orb [orb_rows][ldo], columns beyond the orbitals zero up to ldo = PAMD_e2_orb_ld(nocc_pad); d_out [nL][nocc_pad][ldx].
 * nocc_pad = rows of d_out per aux index: any value >= the number of orbitals (r04: no longer a multiple of 16 - with exactly
 * norb rows the K = X^T X that follows contracts nL * norb rows; pad the END of the block with zero rows to a multiple of 16). */
int PAMD_nr_e2_symm(const double *d_cderi, long npair, int nL, int nao, const double *d_orb, int ldo,
                    int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work,
                    void *stream);
long PAMD_nr_e2_rho_worksize(int nL, int ldx, int nocc_pad);             /* doubles of d_rho_work */
int  PAMD_e2_orb_ld(int nocc_pad);                                       /* ldo that lets every half-transform kernel tile nocc_pad columns */
/* r06: tile shape and k splits of the K = X^T X product that follows the half transform (lib.dot(buf1.T, buf1), pyscf/df/df_jk.py:367,380)
 * - the one rule both host layers use.  flags_in < 0 / nsplit_in <= 0: defaults; reserve: workgroup slots left to a co-running pass. */
long PAMD_k_block_rows(long naux, int rows_per_aux, int ldx, long long budget_bytes);   /* aux rows per X block (`blksize`, df_jk.py:359-360), equal blocks */
int  PAMD_j2_schedule_pick(const double *ms, int ncand);                  /* 0 overlap / 1 serial / 2 fused from the candidates' best times (1 % margin) */
int  PAMD_df_layout_pick(long long need_packed_image, long long need_square_build, long long need_square_after, long long free_bytes, int prefer_image);   /* 2 packed + full image / 1 square rows / 0 packed */
int  PAMD_syrk_item_count(int nao);                                      /* work items of the re-tiled triangle, 0 = 128 x 128 tiles */
int  PAMD_syrk_plan(int nao, int reserve, int flags_in, int nsplit_in, int *flags_out, int *nsplit_out);
/* packed-operand transform with the diagonal-block side image d_diag[nL][ceil(ldx/128)][128][128] of the same aux rows
 * (both triangles of the 128 x 128 blocks on the diagonal of every B_L, 0 beyond nao; 14 % of the packed size at nao 1856):
 * the k-tiles that cross the diagonal are then read once, unmasked - what AO2MOtranse2_nr_s2's per-row NPdunpack_tril
 * (nr_ao2mo.c:1026-1031) does for the whole matrix, done for the blocks where the packed layout changes direction.
 * d_diag NULL = PAMD_nr_e2_symm.  Build the image once per tensor with PAMD_e2_diag_blocks (PAMD_e2_diag_size doubles). */
int PAMD_nr_e2_symm_diag(const double *d_cderi, long npair, int nL, int nao, const double *d_orb, int ldo,
                         int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work,
                         const double *d_diag, void *stream);
long PAMD_e2_diag_size(int nL, int ldx);
int PAMD_e2_diag_blocks(const double *d_cderi, long npair, int nL, int nao, int ldx, double *d_diag, void *stream);
/* the same contraction on the unpacked image sq[nL][rows][ld] (PAMD_unpack_tril into a zeroed buffer, rows = ld =
 * round_up(nao,16)): both operands stream by LDS-DMA; spends 2x the packed size of HBM to take the symmetric unpack out
 * of the hot loop.  d_orb, nocc_pad as for PAMD_nr_e2_symm */
int PAMD_nr_e2_square(const double *d_sq, long ld, int rows, int nL, int nao, const double *d_orb, int ldo,
                      int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work,
                      void *stream);
/* r06: the same with an explicit stride (doubles, even, >= rows * ld) between consecutive aux rows: the square LAYOUT pads it so that
 * equal (p, q) of consecutive aux rows do not fall on one HBM channel (rows * ld * 8 is a multiple of 32 KB .. 8 MB at the named configs) */
int PAMD_nr_e2_square_ls(const double *d_sq, long ld, int rows, long lstride, int nL, int nao, const double *d_orb, int ldo,
                         int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work, void *stream);
/* d_rho (nullable) [nL]: d_rho[L] += sum_{i,p} X[L][i][p] orb[p][i] = sum_pq B_L[p][q] (orb orb^T)[p][q]: the first J pass
 * (df_jk.py:367) of the density the orbitals stand for, taken from the accumulators in the epilogue.  Deterministic: every
 * wave leaves one partial in d_rho_work (PAMD_nr_e2_rho_worksize doubles, required with d_rho) and a fixed-order reduction
 * adds them to d_rho - no floating-point atomics on the energy path */  /* out[L][i][p] */
/* out[y][i][n] = sum_k src_y[n][k] orb[k][i] (plain-operand mode of the e2_symm MFMA kernel) */
int PAMD_orb_dot_rows(const double *d_src, long lds, long src_stride, int ny, long nrows, int kdim,
                      const double *d_orb, int ldo, int nocc_pad, double *d_out, long ldout,
                      const unsigned char *d_kmask, void *stream);   /* kmask nullable: [ny][nrows/128][kdim/16] */
int PAMD_dgemm_tn(const double *d_A, int lda, const double *d_B, int ldb, double *d_C, int ldc, int m,
                  int n, long k, int lower_only, int nsplit, void *stream); /* C[s] += A^T B (k split s) */
/* screened GEMM (VXCdot_ao_ao_sparse role): maskA[k/16][m/128], maskB[k/16][n/128] bytes, operands as flag 2 */
/* r05: the SYRK of PAMD_dgemm_tn (A = B = X, lower triangle, re-tiled) with the second J pass of the same tensor rows folded into
 * the kernel: d_vj[pq] += sum_L d_rho[L] d_B[L][pq] (pyscf/df/df_jk.py:367 and :380 in one launch).  Returns 1 (nothing launched)
 * when the shape has no fused form. */
int PAMD_syrk_jfused(const double *d_X, int ldx, double *d_C, int ldc, int m, long k, int flags, int nsplit, const double *d_B,
                     long npair, int nb, const double *d_rho, double *d_vj, void *stream);
int PAMD_dgemm_tn_masked(const double *d_A, int lda, const double *d_B, int ldb, double *d_C, int ldc, int m,
                         int n, long k, int nsplit, const unsigned char *d_maskA, const unsigned char *d_maskB,
                         void *stream);
/* flags[nrows/16][ld/16] = any |src| > thr in the 16 x 16 tile (value-based screen index) */
int PAMD_tile_mask(const double *d_src, long ld, long nrows, double thr, unsigned char *d_flags, void *stream);
int PAMD_reduce_splits(const double *d_part, int nsplit, int m, int ldc, double *d_out, int ldo,
                       int symmetrize, void *stream);
int PAMD_unpack_tril(const double *d_tril, long npair, int count, int nao, double *d_full, int ld,
                     int rows, void *stream);

int PAMD_set_tuning(const char *key, int value);      /* benchmarking switches, e.g. "glds" 0/1 */

/* ---- DFT grid path -------------------------------------------------------------------------- */
/* pbecke[natm][ngrids]: unnormalised Becke cell functions; radii table a[i][j] nullable */
int PAMD_becke_partition(double *d_out, const double *d_coords, const double *d_atm_coords,
                         const double *d_radii_table, int natm, long ngrids, void *stream);
/* same, cell function by scheme: 0 original Becke, 1 Stratmann-Scuseria-Frisch, 2 Laqua-Kussmann-Ochsenfeld */
int PAMD_grid_partition(double *d_out, const double *d_coords, const double *d_atm_coords,
                        const double *d_radii_table, int natm, long ngrids, int scheme, void *stream);
/* ao[comp][ldg_rows][ldao] (AO index fastest, columns nao..ldao-1 zero), comp = 1 (deriv 0), 4 (deriv 1) or 10 (deriv 2: 1, x, y, z, xx, xy, xz, yy, yz, zz),
 * grid points [g0, g0+ng) of d_coords; d_fn2sh[mu] = segmented shell of AO mu.
 * d_flags (nullable; caller zeroes it): [ceil(ldg_rows/16)][ldao/16] bytes <- 1 where the 16 x 16 (grid x AO) tile
 * has a value of any component above thr: the screening table of GTO_screen_index (lib/gto/grid_ao_drv.c:32-123) */
int PAMD_eval_ao(int deriv, const int *d_l, const int *d_ao0, const int *d_prim0, const int *d_nprim,
                 const double *d_xyz, const double *d_exps, const double *d_coefs, int nsh, const int *d_fn2sh,
                 int nao, const double *d_coords, long g0, long ng, const double *d_c2s, const int *d_c2s_off,
                 double *d_ao, long ldg_rows, int ldao, double thr, unsigned char *d_flags, void *stream);
/* rho[4][ldg] (rho, grad rho) from c[comp][i][ldc] = C_occ^T ao_comp^T (orbital rows), or from ao and
 * c0t[mu][ldc] = (D ao0^T) */
int PAMD_rho_from_mo(const double *d_c, long comp_stride, long ldc, int nocc, int ncomp, long ng,
                     double *d_rho, long ldg, const double *d_occ_sign /* nullable [nocc]: +-1 weights of the rows,
                     for a symmetric matrix factorised as C diag(sign) C^T */, void *stream);
/* first-order density of a factorised matrix A B^T (rank-nocc trial densities of the response solvers): rho = coef sum_i
 * a_i b_i, grad rho = coef sum_i (grad a_i b_i + a_i grad b_i); d_ca / d_cb in the layout of PAMD_rho_from_mo */
int PAMD_rho_from_mo_pair(const double *d_ca, const double *d_cb, long comp_stride, long ldc, int nocc, int ncomp, long ng,
                          double coef, double *d_rho, long ldg, void *stream);
int PAMD_rho_from_dm(const double *d_ao, const double *d_c0t, int nao, int ldao, long ldg_rows, long ldc,
                     int ncomp, long ng, double *d_rho, long ldg, void *stream);
/* fac[PAMD_XC_NFAC = 10]: weights of {Slater, VWN5, VWN_RPA, B88, LYP, PBE_X, PBE_C, ITYH = short-range B88 (libxc gga_x_ityh),
 * WB97 = the omega-B97 exchange-correlation functional}, then the omega of ITYH / WB97.  wv[4][ldg] = w (vrho/2, 2 vsigma grad rho);
 * d_acc[0] += sum w rho, d_acc[1] += sum w e_xc; d_exc nullable */
int PAMD_eval_xc(const double *fac, int gga, const double *d_rho, const double *d_weights, long ng,
                 long ldg, double *d_wv, double *d_exc, double *d_acc, void *stream);
/* closed-shell response kernel, numint.nr_rks_fxc (dft/numint.py:1418-1530, weights of _rks_gga_wv1 :1560-1576):
 * wv1[4][ldg] = w (d vrho / 2, 2 [d vsigma grad rho0 + vsigma grad rho1]) along the first-order density d_rho1[4][ldg] */
int PAMD_eval_fxc(const double *fac, int gga, const double *d_rho0, const double *d_rho1, const double *d_weights,
                  long ng, long ldg, double *d_wv1, void *stream);
/* spin-polarised response kernel, numint.nr_uks_fxc (dft/numint.py:1690-1832, weights of _uks_gga_wv1 :1834-1915) */
int PAMD_eval_fxc_pol(const double *fac, int gga, const double *d_rho0_a, const double *d_rho0_b, const double *d_rho1_a,
                      const double *d_rho1_b, const double *d_weights, long ng, long ldg, double *d_wv1_a,
                      double *d_wv1_b, void *stream);
/* spin-polarised variant for nr_uks (dft/numint.py:1192-1324): d_acc3 = {nelec_a, nelec_b, exc} */
int PAMD_eval_xc_pol(const double *fac, int gga, const double *d_rho_a, const double *d_rho_b,
                     const double *d_weights, long ng, long ldg, double *d_wv_a, double *d_wv_b, double *d_acc3,
                     double *d_evol /* nullable: energy density per volume */, void *stream);
/* XC nuclear gradient of one grid block (pyscf/grad/rks.py:197-255 get_vxc/_gga_grad_sum_/_make_dR_dao_w contracted with
 * the density on the fly): d_out[3][nao] += sum_g {...}, d_c[k][g][mu] = sum_nu ao_k[g][nu] D[nu][mu] */
int PAMD_xc_grad(const double *d_ao, const double *d_c, const double *d_wv, int ldao, long ldg_rows, long ldg,
                 int gga, long ng, int nao, double *d_out, void *stream);
/* grid response of the XC gradient (pyscf/grad/rks.py:257-340 get_vxc_full_response): per-point row sums of the same
 * integrand (the points' own motion) and the Becke weight derivatives contracted with the energy density */
int PAMD_xc_grad_rows(const double *d_ao, const double *d_c, const double *d_wv, int ldao, long ldg_rows, long ldg,
                      int gga, long ng, int nao, double *d_rows, void *stream);
int PAMD_becke_response(const double *d_coords, const int *d_owner, const double *d_weights, const double *d_e,
                        const double *d_pb, const double *d_atm_coords, const double *d_radii_table, int natm,
                        long ng, double *d_out, void *stream);
/* same for the cell function `scheme` of PAMD_grid_partition: grad/rks.py grids_response_becke (0, 1 with the Stratmann
 * switch) / grids_response_lko (2; lib/dft/grid_basis.c:386-560 VXCgen_grid_lko_deriv) */
int PAMD_grid_response(const double *d_coords, const int *d_owner, const double *d_weights, const double *d_e,
                       const double *d_pb, const double *d_atm_coords, const double *d_radii_table, int natm,
                       long ng, int scheme, double *d_out, void *stream);
int PAMD_scale_ao(const double *d_ao, const double *d_wv, int ldao, long ldg_rows, long ldg, int ncomp,
                  long ng, long nrows, double *d_aow, void *stream);     /* aow[g][ldao], rows ng..nrows-1 zero */
int PAMD_dgemm_nt(const double *d_A, long lda, const double *d_B, long ldb, double *d_C, int ldc, int m,
                  int n, long k, int nsplit, void *stream);                /* C[s] += A B^T (k split s) */
int PAMD_reduce_sym(const double *d_part, int nsplit, int m, int ldc, double *d_out, void *stream);

/* ---- block-sparse XC on compact AO subsets (csrc/xc_sparse.hip) ------------------------------------------------------
 * The role of numint's non0tab / screen_index shell lists (dft/numint.py:2845, lib/gto/grid_ao_drv.c:32-123
 * GTO_screen_index) and of the sparse products VXCdot_ao_dm_sparse / VXCdot_ao_ao_sparse
 * (lib/dft/nr_numint_sparse.c:226-304, :890-973).  The grid is cut into tiles of G points; tile t stores the AO values of
 * its active functions compacted, ao_c[t] = [ncomp][G][ld_t] at d_ao_c + d_ao_off[t] (ld_t = d_ld[t], a multiple of 16),
 * and d_idx[d_idx_off[t] + mu] is the AO index of compact column mu (>= nao for padding).  One launch per product covers
 * all `ntile` tiles handed to the call. */
int PAMD_set_tuning_xc(const char *key, int value);      /* benchmarking switch: "orbdotdma" = 0/1 */
int PAMD_sub_gather_ao(const double *d_dense, long dense_rows, int ldao, int ncomp, long row0, long nrows_valid,
                       const long *d_ao_off, const long *d_idx_off, const int *d_ld, const int *d_idx, int ntile, int G,
                       int ld_max, int nao, double *d_ao_c, void *stream);     /* ao_c <- columns of a PAMD_eval_ao block */
int PAMD_sub_orb_dot(const double *d_ao_c, const long *d_ao_off, const long *d_idx_off, const int *d_ld, const int *d_idx,
                     int ntile, int G, int ncomp, const double *d_orb, int ldo, int nocc_pad, double *d_cmo,
                     long comp_stride, long ldc, void *stream);   /* cmo[c][i][t G + g] = sum_mu orb[idx[mu]][i] ao_c[t][c][g][mu] */
/* r04: rho[4][ldg] (rho, grad rho) of sum_i sign_i c_i c_i^T on the GGA image in one kernel (= PAMD_sub_orb_dot + PAMD_rho_from_mo,
 * numint.py:328-469); returns 1 without launching when the shape has no fused kernel (the caller then uses the two calls) */
int PAMD_sub_orb_rho(const double *d_ao_c, const long *d_ao_off, const long *d_idx_off, const int *d_ld, const int *d_idx,
                     int ntile, int G, const double *d_orb, int ldo, int nocc, int nocc_pad, const double *d_sign, double *d_rho,
                     long ldg, void *stream);
int PAMD_sub_scale_ao(const double *d_ao_c, const long *d_ao_off, const long *d_aow_off, const int *d_ld, int ntile, int G,
                      int ncomp, int ld_max, const double *d_wv, long ldg, double *d_aow_c, void *stream);
int PAMD_sub_vmat(const double *d_ao_c, const long *d_ao_off, const double *d_aow_c, const long *d_aow_off,
                  const long *d_idx_off, const int *d_ld, const int *d_idx, const int *d_work /* {tile, tm, tn} x nwork */,
                  int nwork, int G, int nao, double *d_vmat, long ldv, void *stream);  /* vmat[idx][idx] += ao_c[0]^T aow_c */

/* r04: the symmetrised product V = M + M^T on balanced blocks, lower triangle only (two products per block, half the atomics);
 * d_work = {tile, p0, gp, q0, gq, diag} x nwork from PAMD_sub_vmat_work (host; work NULL -> count); PAMD_mirror_tril completes V */
int PAMD_sub_vmat_sym(const double *d_ao_c, const long *d_ao_off, const double *d_aow_c, const long *d_aow_off,
                      const long *d_idx_off, const int *d_ld, const int *d_idx, const int *d_work, int nwork, int G, int nao,
                      double *d_vmat, long ldv, void *stream);
long PAMD_sub_vmat_work(const int *ld, int ntile, int *work);
int PAMD_mirror_tril(const double *d_part, int m, int ldc, double *d_out, void *stream);

/* ---- host-array, opaque-handle form of the DF J/K path (csrc/df_handle.hip) -------------------------------------------
 * The reference's convention: plain C functions taking raw HOST pointers and ints, the caller owns every buffer
 * (pyscf/df/df_jk.py:373-379 fdrv(..., buf1.ctypes.data_as(c_void_p), eri1..., orbo...), pyscf/gto/moleintor.py:590-596).
 * These entry points need no device-side runtime in the caller (no torch): the handle owns all HBM (hipMalloc), calls
 * return when the results are in the caller's arrays.  Return 0 or a negative code (PAMD_last_error() has the message).
 *   PAMD_df_create        replaces DF.build (pyscf/df/df.py:147-199) -> incore.cholesky_eri (pyscf/df/incore.py:129-220):
 *                         atm[natm][6], bas[nbas_ao + nbas_aux][8] (AO rows, then aux rows: gto.conc_env), env[nenv] in libcint's
 *                         format (pyscf/gto/mole.py:58-88); lindep = LINEAR_DEP_THR of the eigen-decomposition fallback (:263-270)
 *   PAMD_df_get_jk        replaces df_jk.get_jk (pyscf/df/df_jk.py:280-413): dm[nset][nao][nao]; orbo = NULL -> general-DM
 *                         branch (:382-408); else the sqrt(occ)-scaled occupied orbitals of every density, (nao, nocc[s]) C order
 *                         one after the other -> MO branch (:339-381).  flags bit 0: dm[s] == orbo_s orbo_s^T is guaranteed
 *                         (the first J pass then comes out of the half transform's epilogue); flags bit 1 (r06): NOT guaranteed -
 *                         verify it inside the call: one D v = C (C^T v) probe per density on the calling thread while the queued
 *                         kernels run (hidden), and when it fails J is recomputed from the matrix before the call returns - what
 *                         the reference always does (df_jk.py:367), K keeps following the tag (df_jk.py:340);
 *                         PAMD_df_last_mismatch(h) = the probe's result
 *                         flags bit 3 (r06): dm, orbo, vj, vk are DEVICE pointers on the handle's device (one-part handles): the
 *                         HBM-resident SCF loop over a handle-held tensor - nothing crosses PCIe; the tag then counts as
 *                         promised (bit 0) or absent (the host cannot probe device memory)
 *   PAMD_df_export_cderi  rows [l0, l1) of `_cderi` (naux, nao_pair) into out (DF.loop, pyscf/df/df.py:214-242)
 *   PAMD_df_naux          rows of the tensor (get_naoaux, :248-257: fewer than the aux functions after an eigen-decomposition) */
typedef struct PAMD_df PAMD_df;
/* r04 - the same handle over SEVERAL devices of the node, in ONE process (SURVEY.md 8(b) `mi_ctx_create(const int *devices, int ndev, ...)`;
 * the caller is the single Python thread of hf.kernel, pyscf/df/df_jk.py:175-176): the auxiliary index L is cut into ndev contiguous,
 * row-balanced shards (the serial loop over L blocks this replaces: pyscf/df/df_jk.py:362-381), part p lives on devices[p] (a device
 * may be listed more than once), PAMD_df_get_jk runs one host thread per part and sums the partial [J~ | K] on devices[0] (direct
 * peer copies over xGMI where hipDeviceCanAccessPeer allows, through the host otherwise; K of the MO branch travels packed).
 *   PAMD_df_create_multi  = PAMD_df_create with a device list
 *   PAMD_df_create_ex     every option:  omega != 0 -> the tensor of erf(omega r12)/r12 (omega > 0) or erfc(|omega| r12)/r12 (omega < 0),
 *                         what DF.range_coulomb(omega) holds (pyscf/df/df.py:298-333);  max_device_bytes > 0 caps the HBM one part
 *                         may take - rows that do not fit (with or without a cap) stay in page-locked HOST memory and are streamed
 *                         through two staging buffers under the kernels in every PAMD_df_get_jk: the out-of-core twin of the
 *                         reference (pyscf/df/outcore.py:109-232, pyscf/df/df.py:167,214-242), PCIe-bound for those rows.
 *                         flags bit 0: keep the sharded code path even for a one-entry device list
 *                         flags bit 1 (r05): ONE RANK of a multi-process job - the handle holds only the rows of shard `part` of
 *                         `nparts` (contiguous, row-balanced: DF.shard_range) on devices[0], out of core where they do not fit;
 *                         PAMD_df_get_jk returns that shard's PARTIAL J/K, the caller sums over the ranks (the accumulation over
 *                         `dfobj.loop()` blocks of pyscf/df/df_jk.py:362-381, spread over processes)
 *                         flags bit 2 (r06): reserve_bytes is valid - what the XC leg of the same calculation will cache per device
 *                         (one budget: tensor, X block, XC compact image, work space).  The rows are held in the SQUARE layout
 *                         (both triangles, padded aux-row stride, 2x the packed bytes and NO second copy) when that fits with the
 *                         reserve, else packed as the reference's `_cderi` (pyscf/df/df.py:59-72) with the optional image
 *   PAMD_df_shard_info    info[4] = {first global row, rows held, rows of the whole tensor, 1 = partial sums}
 *   PAMD_df_last_timing   timings of the last PAMD_df_get_jk: out[3 + 5 parts] = {parts, host ms of sum + download on part 0, peer
 *                         copies used 0/1, then per part: host ms contraction, host ms push into the gather buffer, bytes pushed
 *                         across devices, HIP-event ms of the half-transform launches, of the SYRK launches}
 *   PAMD_df_layout        layout[5] = {parts, rows resident in HBM, rows in host memory, rows with a square image, peer copies 0 / 1},
 *                         part_rows[parts] (nullable) = rows per part */
typedef struct PAMD_df_options {
    double lindep;              /* LINEAR_DEP_THR of the eigen-decomposition fallback (pyscf/df/incore.py:263-270) */
    double omega;               /* range-separation parameter, 0 = Coulomb */
    const int *devices;         /* [ndev] HIP device indices (NULL with ndev <= 0: device 0) */
    int ndev;
    int flags;
    long long max_device_bytes; /* 0: whatever the device has free */
    int part;                   /* flags bit 1 only: this handle holds shard `part` ... */
    int nparts;                 /* ... of `nparts` contiguous row-balanced shards (one rank of a multi-process job) */
    long long reserve_bytes;    /* flags bit 2 (r06): HBM to leave free per device besides the J/K work space (see above) */
} PAMD_df_options;
int PAMD_df_create_ex(const int *atm, int natm, const int *bas, int nbas_ao, int nbas_aux, const double *env, int nenv,
                      const PAMD_df_options *opt, PAMD_df **out);
int PAMD_df_create_multi(const int *atm, int natm, const int *bas, int nbas_ao, int nbas_aux, const double *env, int nenv,
                         double lindep, const int *devices, int ndev, PAMD_df **out);
/* The tag probe of a density that carries its orbitals (lib.tag_array, pyscf/scf/hf.py:855-868; the reference trusts the tag,
 * "#TODO: test whether dm.mo_coeff matching dm", pyscf/df/df_jk.py:340): max_s |D_s v - C_s (C_s^T v)| / max(1, |D_s v|) on host arrays. */
int PAMD_dm_orbital_mismatch(const double *dm, const double *orbo, const int *nocc, int nset, int nao, double *out);
/* r05 - a handle over tensor rows the caller already holds: `DF._cderi` given as an array or as PySCF's own HDF5 file
 * (pyscf/df/df.py:153-155; dataset 'j3c', outcore.py:217-221; read back block by block in DF.loop, df.py:214-242).
 * rows[nrows][nao (nao + 1) / 2] f64 in HOST memory (a numpy array or an mmap of the contiguous dataset); what fits the device is
 * uploaded once, the rest is streamed under the kernels in every PAMD_df_get_jk.  flags bit 0: stream straight from the caller's
 * memory (must outlive the handle) instead of a page-locked copy. */
int PAMD_df_create_from_rows(const double *rows, int nrows, int nao, int device, long long max_device_bytes, int flags, PAMD_df **out);
int PAMD_df_layout(const PAMD_df *h, long *layout, int *part_rows);
double PAMD_df_last_mismatch(const PAMD_df *h);
int PAMD_df_tensor_layout(const PAMD_df *h);     /* 1: every part holds its rows in the square layout only (r06); 0: packed rows */
int PAMD_df_shard_info(const PAMD_df *h, int *info);
int PAMD_df_last_timing(const PAMD_df *h, double *out, int nout);
/* the metric factorisation alone (df/incore.py:150-158, :263-270; decompose_j2c = 'ED' with force_ed): host j2c[naux][naux] ->
 * host m[nrow][naux] (caller provides naux x naux doubles), cderi = m (Q|pq); *tri = 1: rows of L^-1 */
int PAMD_metric_decompose(const double *j2c, int naux, double lindep, int force_ed, int device, double *m, int *nrow, int *tri);
int PAMD_df_create(const int *atm, int natm, const int *bas, int nbas_ao, int nbas_aux, const double *env, int nenv,
                   double lindep, int device, PAMD_df **out);
void PAMD_df_destroy(PAMD_df *h);
/* page-locked (portable) host memory for result / input arrays of the host-array entry points: copies at the PCIe rate instead of
 * the pageable ~10 GB/s.  Optional - every entry point takes any host pointer. */
int PAMD_host_alloc(long long nbytes, void **out);
int PAMD_host_free(void *p);
int PAMD_df_naux(const PAMD_df *h, int *naux);
int PAMD_df_nao(const PAMD_df *h, int *nao);
int PAMD_df_export_cderi(PAMD_df *h, int l0, int l1, double *out);
int PAMD_df_get_jk(PAMD_df *h, const double *dm, const double *orbo, const int *nocc, int nset, int nao, int hermi, int with_j,
                   int with_k, int flags, double *vj, double *vk);

/* ---- host-array, opaque-handle form of the XC quadrature (csrc/xc_handle.hip; SURVEY.md 8(b) mi_xc_build_grids / mi_nr_rks) ----
 * Same convention as the PAMD_df_* handle: raw HOST pointers, the caller owns every array, no device runtime needed in the caller.
 *   PAMD_grid_weights_host  replaces VXCgen_grid (pyscf/lib/dft/grid_basis.c:32-101) + the normalisation loop of
 *                           gen_grid.get_partition (pyscf/dft/gen_grid.py:341-419) for the points of atom `ia`: coords[ngrids][3]
 *                           (already shifted to the atom), vol[ngrids] the atomic quadrature weights, radii_table[natm][natm]
 *                           (nullable: radii_adjust = None), scheme 0 Becke / 1 Stratmann / 2 LKO  ->  weights[ngrids]
 *   PAMD_xc_create          atm / bas / env of the molecule (libcint format) + coords[ngrids][3], weights[ngrids] of a built Grids
 *                           object (pyscf/dft/gen_grid.py:487-744) -> handle; the block-sparse plan (tiles of 512 points, active
 *                           shells per tile, compact AO image cached in HBM) is built at the first call per functional type
 *   PAMD_xc_nr_rks          replaces numint.nr_rks (pyscf/dft/numint.py:1074-1190).  fac[PAMD_XC_NFAC] = the component weights the
 *                           host-side parser produces (libxc.parse_xc -> LIBXC_eval_xc(ids, facs), pyscf/dft/libxc.py:496-720,
 *                           pyscf/lib/dft/libxc_itrf.c:968-1024); xctype 0 LDA / 1 GGA; every density s of the nset is given by
 *                           orbital factors D_s = sum_i signs[i] c_i c_i^T: orbs = the (nao, nocc[s]) blocks one after the other
 *                           (C order, row = AO; occupied orbitals scaled by sqrt(occ) as numint's _gen_rho_evaluator MO branch,
 *                           :2930-2994), signs nullable (all +1) -> nelec[nset], exc[nset], vmat[nset][nao][nao]
 *   PAMD_xc_nr_uks          replaces numint.nr_uks (:1192-1324): orbs = alpha block then beta block, nocc[2] -> nelec[2], exc[1],
 *                           vmat[2][nao][nao]
 *   PAMD_xc_plan_info       info[3] = {tiles, mean fraction of AO functions active per tile, GB of the cached compact image} */
#define PAMD_XC_NFAC 10
typedef struct PAMD_xc PAMD_xc;
int PAMD_grid_weights_host(const double *coords, long ngrids, const double *atm_coords, int natm, const double *radii_table, int scheme,
                           int ia, const double *vol, int device, double *weights);
int PAMD_xc_create(const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, const double *coords,
                   const double *weights, long ngrids, int device, PAMD_xc **out);
/* the same over a device list in one process: grid tiles (512 consecutive points) dealt round-robin over the parts, one host
 * thread per part, nelec / exc summed on the host and vmat on devices[0] (peer copies or host bounce) */
int PAMD_xc_create_multi(const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, const double *coords,
                         const double *weights, long ngrids, const int *devices, int ndev, PAMD_xc **out);
void PAMD_xc_destroy(PAMD_xc *h);
int PAMD_xc_nao(const PAMD_xc *h, int *nao);
int PAMD_xc_plan_info(PAMD_xc *h, int xctype, double *info);
/* r05: HIP-event timings of the last PAMD_xc_nr_rks / _nr_uks: out[10] = {parts, ms of the slowest part's orbital product (+ densities),
 * eval_xc, scale, vmat; sum of the tiles' compact widths, of their squares; points per tile; components; padded orbital count} */
int PAMD_xc_last_timing(const PAMD_xc *h, double *out, int nout);
int PAMD_xc_nr_rks(PAMD_xc *h, const double *fac, int xctype, int nset, const double *orbs, const int *nocc, const double *signs,
                   double *nelec, double *exc, double *vmat);
int PAMD_xc_nr_uks(PAMD_xc *h, const double *fac, int xctype, const double *orbs, const int *nocc, const double *signs, double *nelec,
                   double *exc, double *vmat);

#ifdef __cplusplus
}
#endif
#endif /* PYSCF_AMD_H */
