// One-electron overlap and kinetic-energy integrals (real spherical), one thread per
// segmented shell pair.  Not a hot kernel: it feeds get_ovlp / get_hcore of the SCF driver
// (pyscf/scf/hf.py:322-345; libcint int1e_ovlp_sph / int1e_kin_sph via GTOint2c,
// pyscf/lib/gto/fill_int2c.c).  Obara-Saika 1-D overlap recurrence (J. Chem. Phys. 84, 3963).
// Nuclear attraction is obtained from the int3c2e family with point-charge "aux shells"
// (pyscf_amd/scf/hf.py:get_hcore).
#include "common.h"

using namespace pamd;

namespace {

constexpr int LMAX = 4;
constexpr int NC = (LMAX + 1) * (LMAX + 2) / 2;

__device__ inline void cart_exps(int l, int c, int &lx, int &ly, int &lz)
{
    int x = l, rem = c;
    while (rem > l - x) { rem -= (l - x + 1); x--; }
    lx = x; ly = (l - x) - rem; lz = rem;
}

struct ShellTab {
    const int *l;          // [nsh]
    const int *ao0;        // [nsh]
    const int *prim0;      // [nsh] offset into exps/coefs
    const int *nprim;      // [nsh]
    const double *xyz;     // [nsh][3]
    const double *exps;
    const double *coefs;
};

__global__ void int1e_ovlp_kin_kernel(ShellTab t, int nsh, int nao, const double *__restrict__ c2s,
                                      const int *__restrict__ c2s_off, double *__restrict__ S,
                                      double *__restrict__ K)
{
    long pid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long npairs = (long)nsh * (nsh + 1) / 2;
    if (pid >= npairs) return;
    int ish = (int)((sqrt(8.0 * pid + 1.0) - 1.0) * 0.5);
    while ((long)(ish + 1) * (ish + 2) / 2 <= pid) ish++;
    while ((long)ish * (ish + 1) / 2 > pid) ish--;
    int jsh = (int)(pid - (long)ish * (ish + 1) / 2);
    const int li = t.l[ish], lj = t.l[jsh];
    const int nci = (li + 1) * (li + 2) / 2, ncj = (lj + 1) * (lj + 2) / 2;
    double sc[NC * NC], kc[NC * NC];
    for (int e = 0; e < nci * ncj; e++) { sc[e] = 0; kc[e] = 0; }
    const double *A = t.xyz + 3 * ish, *B = t.xyz + 3 * jsh;
    for (int pa = 0; pa < t.nprim[ish]; pa++)
    for (int pb = 0; pb < t.nprim[jsh]; pb++) {
        const double a = t.exps[t.prim0[ish] + pa], b = t.exps[t.prim0[jsh] + pb];
        const double cc = t.coefs[t.prim0[ish] + pa] * t.coefs[t.prim0[jsh] + pb];
        const double p = a + b, mu = a * b / p, hp = 0.5 / p;
        double s[3][LMAX + 1][LMAX + 3];
        for (int d = 0; d < 3; d++) {
            const double ab = A[d] - B[d];
            const double P = (a * A[d] + b * B[d]) / p;
            const double pa_ = P - A[d], pb_ = P - B[d];
            s[d][0][0] = sqrt(M_PI / p) * exp(-mu * ab * ab);
            for (int i = 0; i < li; i++)
                s[d][i + 1][0] = pa_ * s[d][i][0] + (i ? hp * i * s[d][i - 1][0] : 0.0);
            for (int j = 0; j < lj + 2; j++)
                for (int i = 0; i <= li; i++) {
                    double v = pb_ * s[d][i][j];
                    if (i) v += hp * i * s[d][i - 1][j];
                    if (j) v += hp * j * s[d][i][j - 1];
                    s[d][i][j + 1] = v;
                }
        }
        for (int ci = 0; ci < nci; ci++) {
            int ix[3];
            cart_exps(li, ci, ix[0], ix[1], ix[2]);
            for (int cj = 0; cj < ncj; cj++) {
                int jx[3];
                cart_exps(lj, cj, jx[0], jx[1], jx[2]);
                double s1[3], t1[3];
                for (int d = 0; d < 3; d++) {
                    const int i = ix[d], j = jx[d];
                    s1[d] = s[d][i][j];
                    double v = -2 * b * (2 * j + 1) * s[d][i][j] + 4 * b * b * s[d][i][j + 2];
                    if (j >= 2) v += j * (j - 1) * s[d][i][j - 2];
                    t1[d] = -0.5 * v;
                }
                sc[ci * ncj + cj] += cc * s1[0] * s1[1] * s1[2];
                kc[ci * ncj + cj] += cc * (t1[0] * s1[1] * s1[2] + s1[0] * t1[1] * s1[2] + s1[0] * s1[1] * t1[2]);
            }
        }
    }
    // cart -> sph on both indices, scatter symmetric
    const double *ci_m = c2s + c2s_off[li], *cj_m = c2s + c2s_off[lj];
    const int nsi = 2 * li + 1, nsj = 2 * lj + 1;
    for (int mi = 0; mi < nsi; mi++)
        for (int mj = 0; mj < nsj; mj++) {
            double vs = 0, vk = 0;
            for (int ci = 0; ci < nci; ci++) {
                const double f = ci_m[mi * nci + ci];
                if (f == 0) continue;
                for (int cj = 0; cj < ncj; cj++) {
                    const double g = f * cj_m[mj * ncj + cj];
                    vs += g * sc[ci * ncj + cj];
                    vk += g * kc[ci * ncj + cj];
                }
            }
            const long p = t.ao0[ish] + mi, q = t.ao0[jsh] + mj;
            S[p * nao + q] = vs; S[q * nao + p] = vs;
            K[p * nao + q] = vk; K[q * nao + p] = vk;
        }
}


// grad[atom] += sum_pq ( Dt[p][q] d<p|T|q>/dR_atom - Ws[p][q] d<p|q>/dR_atom ): the int1e_ipkin and
// int1e_ipovlp contractions of pyscf/grad/rhf.py:62-75 (grad_elec: h1ao . dm0, s1 . dme0), one thread per
// shell pair i > j on different atoms.  d/dA = -grad_r on function i; (grad i|O|j) = -(i|O|grad j) for
// these two-centre operators, so one derivative block serves both atoms.
__global__ void int1e_grad_kernel(ShellTab t, const int *__restrict__ sh_atom, int nsh, int nao,
                                  const double *__restrict__ c2s, const int *__restrict__ c2s_off,
                                  const double *__restrict__ Dt, const double *__restrict__ Ws,
                                  double *__restrict__ grad)
{
    long pid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long npairs = (long)nsh * (nsh + 1) / 2;
    if (pid >= npairs) return;
    int ish = (int)((sqrt(8.0 * pid + 1.0) - 1.0) * 0.5);
    while ((long)(ish + 1) * (ish + 2) / 2 <= pid) ish++;
    while ((long)ish * (ish + 1) / 2 > pid) ish--;
    int jsh = (int)(pid - (long)ish * (ish + 1) / 2);
    const int ia = sh_atom[ish], ja = sh_atom[jsh];
    if (ia == ja) return;
    const int li = t.l[ish], lj = t.l[jsh];
    const int nci = (li + 1) * (li + 2) / 2, ncj = (lj + 1) * (lj + 2) / 2;
    const int nsi = 2 * li + 1, nsj = 2 * lj + 1;
    // contraction weights of the block in the Cartesian basis: xk for the kinetic, xs for the overlap derivative
    double xk[NC * NC], xs[NC * NC];
    const double *ci_m = c2s + c2s_off[li], *cj_m = c2s + c2s_off[lj];
    for (int ci = 0; ci < nci; ci++)
        for (int cj = 0; cj < ncj; cj++) {
            double vk = 0, vs = 0;
            for (int mi = 0; mi < nsi; mi++) {
                const double f = ci_m[mi * nci + ci];
                if (f == 0) continue;
                for (int mj = 0; mj < nsj; mj++) {
                    const double g = f * cj_m[mj * ncj + cj];
                    const long p = t.ao0[ish] + mi, q = t.ao0[jsh] + mj;
                    vk += g * Dt[p * nao + q];
                    vs += g * Ws[p * nao + q];
                }
            }
            xk[ci * ncj + cj] = vk;
            xs[ci * ncj + cj] = vs;
        }
    double gr[3] = {0, 0, 0};
    const double *A = t.xyz + 3 * ish, *B = t.xyz + 3 * jsh;
    for (int pa = 0; pa < t.nprim[ish]; pa++)
    for (int pb = 0; pb < t.nprim[jsh]; pb++) {
        const double a = t.exps[t.prim0[ish] + pa], b = t.exps[t.prim0[jsh] + pb];
        const double cc = t.coefs[t.prim0[ish] + pa] * t.coefs[t.prim0[jsh] + pb];
        const double p = a + b, mu = a * b / p, hp = 0.5 / p;
        double s[3][LMAX + 2][LMAX + 3];
        for (int d = 0; d < 3; d++) {
            const double ab = A[d] - B[d];
            const double P = (a * A[d] + b * B[d]) / p;
            const double pa_ = P - A[d], pb_ = P - B[d];
            s[d][0][0] = sqrt(M_PI / p) * exp(-mu * ab * ab);
            for (int i = 0; i < li + 1; i++)
                s[d][i + 1][0] = pa_ * s[d][i][0] + (i ? hp * i * s[d][i - 1][0] : 0.0);
            for (int j = 0; j < lj + 2; j++)
                for (int i = 0; i <= li + 1; i++) {
                    double v = pb_ * s[d][i][j];
                    if (i) v += hp * i * s[d][i - 1][j];
                    if (j) v += hp * j * s[d][i][j - 1];
                    s[d][i][j + 1] = v;
                }
        }
        auto kin1 = [&](int d, int i, int j) {           // 1-D kinetic factor T(i,j)
            double v = -2 * b * (2 * j + 1) * s[d][i][j] + 4 * b * b * s[d][i][j + 2];
            if (j >= 2) v += j * (j - 1) * s[d][i][j - 2];
            return -0.5 * v;
        };
        for (int ci = 0; ci < nci; ci++) {
            int ix[3];
            cart_exps(li, ci, ix[0], ix[1], ix[2]);
            for (int cj = 0; cj < ncj; cj++) {
                int jx[3];
                cart_exps(lj, cj, jx[0], jx[1], jx[2]);
                double s1[3], t1[3], ds[3], dt[3];
                for (int d = 0; d < 3; d++) {
                    const int i = ix[d], j = jx[d];
                    s1[d] = s[d][i][j];
                    t1[d] = kin1(d, i, j);
                    ds[d] = -2 * a * s[d][i + 1][j] + (i ? i * s[d][i - 1][j] : 0.0);
                    dt[d] = -2 * a * kin1(d, i + 1, j) + (i ? i * kin1(d, i - 1, j) : 0.0);
                }
                const double wk = cc * xk[ci * ncj + cj], ws = cc * xs[ci * ncj + cj];
                for (int d = 0; d < 3; d++) {
                    const int e = (d + 1) % 3, f = (d + 2) % 3;
                    const double dS = ds[d] * s1[e] * s1[f];
                    const double dT = dt[d] * s1[e] * s1[f] + ds[d] * (t1[e] * s1[f] + s1[e] * t1[f]);
                    gr[d] += wk * dT - ws * dS;
                }
            }
        }
    }
    // the block (p in i, q in j) and its transpose: factor 2; nuclear derivative on A = -(grad i|..), on B = +(grad i|..)
    for (int d = 0; d < 3; d++) {
        atomicAdd(grad + ia * 3 + d, -2.0 * gr[d]);
        atomicAdd(grad + ja * 3 + d, 2.0 * gr[d]);
    }
}

}  // namespace

extern "C" {

// shells: segmented (nctr = 1) shell table on the device; S, K: (nao, nao) outputs.
int PAMD_int1e_ovlp_kin(const int *d_l, const int *d_ao0, const int *d_prim0, const int *d_nprim,
                        const double *d_xyz, const double *d_exps, const double *d_coefs, int nsh, int nao,
                        const double *d_c2s, const int *d_c2s_off, double *d_S, double *d_K, void *stream)
{
    if (nsh == 0) return 0;
    ShellTab t{d_l, d_ao0, d_prim0, d_nprim, d_xyz, d_exps, d_coefs};
    long npairs = (long)nsh * (nsh + 1) / 2;
    int1e_ovlp_kin_kernel<<<ceil_div(npairs, 64), 64, 0, (hipStream_t)stream>>>(t, nsh, nao, d_c2s, d_c2s_off,
                                                                               d_S, d_K);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// d_grad[natm][3] += Tr(Dt dT/dR) - Tr(Ws dS/dR); Dt, Ws symmetric (nao, nao) device matrices
int PAMD_int1e_grad(const int *d_l, const int *d_ao0, const int *d_prim0, const int *d_nprim,
                    const double *d_xyz, const double *d_exps, const double *d_coefs, const int *d_sh_atom,
                    int nsh, int nao, const double *d_c2s, const int *d_c2s_off, const double *d_Dt,
                    const double *d_Ws, double *d_grad, void *stream)
{
    if (nsh == 0) return 0;
    ShellTab t{d_l, d_ao0, d_prim0, d_nprim, d_xyz, d_exps, d_coefs};
    long npairs = (long)nsh * (nsh + 1) / 2;
    int1e_grad_kernel<<<ceil_div(npairs, 64), 64, 0, (hipStream_t)stream>>>(t, d_sh_atom, nsh, nao, d_c2s, d_c2s_off,
                                                                           d_Dt, d_Ws, d_grad);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
