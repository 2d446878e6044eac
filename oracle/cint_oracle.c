/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Nothing under pyscf_amd/ may link or call this.
 *
 * CPU restatement of the Gaussian integrals the density-fitting hot path consumes.
 * The reference (pyscf/pyscf v2.14.0) obtains them from the third-party library
 * libcint v6.1.3 (pinned at pyscf/lib/CMakeLists.txt:176-209), which is NOT in the
 * reference tree.  Call sites that define the conventions restated here:
 *   - 3-centre fill  pyscf/lib/gto/fill_nr_3c.c:31-225 (int3c2e_sph, buf[k][j][i])
 *   - 2-centre fill  pyscf/lib/gto/fill_int2c.c          (int2c2e_sph, int1e_*_sph)
 *   - 4-centre fill  pyscf/lib/vhf/fill_nr_s8.c:110-139   (int2e_sph)
 *   - s/p factors and the cart->real-spherical convention
 *       pyscf/gto/mole.py:159-189, pyscf/lib/gto/grid_ao_drv.c:250-277,
 *       pyscf/lib/parameters.py:69-77, pyscf/symm/sph.py:24-56
 *   - basis normalisation (coefficients already carry it) pyscf/gto/mole.py:122-157
 *
 * Algorithm: McMurchie-Davidson Hermite expansion with Boys functions (published
 * algorithm; J. Comput. Phys. 26, 218 (1978)) - deliberately DIFFERENT from the Rys
 * quadrature used by the HIP kernels so that the two implementations are independent.
 * Parity is pinned against the reference's own golden values (tests/test_oracle_golden.py:
 * lib.fp(int3c2e) = 45.27912877994409 / 12.407403711205063 from
 * pyscf/df/test/test_incore.py:67,72, J/K fingerprints test_df_jk.py:144-152, energies).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LMAX 7           /* max angular momentum of one shell */
#define LMAX1 (LMAX + 1)
#define NCART(l) (((l) + 1) * ((l) + 2) / 2)
#define ATOM_OF 0
#define ANG_OF 1
#define NPRIM_OF 2
#define NCTR_OF 3
#define PTR_EXP 5
#define PTR_COEFF 6
#define BAS_SLOTS 8
#define ATM_SLOTS 6
#define PTR_COORD 1

/* ---------------------------------------------------------------- Boys function */
static void boys(int nmax, double x, double *f)
{
    if (x < 1e-14) {
        for (int n = 0; n <= nmax; n++) f[n] = 1.0 / (2 * n + 1);
        return;
    }
    if (x < nmax + 36.0) {
        /* series for F_nmax, then downward recursion (stable) */
        double ex = exp(-x);
        double term = 1.0 / (2 * nmax + 1), sum = term;
        for (int k = 1; k < 400; k++) {
            term *= 2 * x / (2 * nmax + 2 * k + 1);
            sum += term;
            if (term < 1e-17 * sum) break;
        }
        f[nmax] = ex * sum;
        for (int n = nmax - 1; n >= 0; n--) f[n] = (2 * x * f[n + 1] + ex) / (2 * n + 1);
    } else {
        double ex = exp(-x);
        f[0] = 0.5 * sqrt(M_PI / x) * erf(sqrt(x));
        for (int n = 0; n < nmax; n++) f[n + 1] = ((2 * n + 1) * f[n] - ex) / (2 * x);
    }
}

/* ---------------------------------------------------------------- shells */
typedef struct {
    int l, nprim, nctr, dummy;
    const double *exps;  /* [nprim] */
    const double *coef;  /* [nctr][nprim] */
    double r[3];
} Shell;

static const double ONE = 1.0, ZERO = 0.0;

static Shell get_shell(const int *atm, const int *bas, const double *env, int ib)
{
    Shell s;
    const int *b = bas + BAS_SLOTS * ib;
    s.l = b[ANG_OF];
    s.nprim = b[NPRIM_OF];
    s.nctr = b[NCTR_OF];
    s.exps = env + b[PTR_EXP];
    s.coef = env + b[PTR_COEFF];
    s.dummy = 0;
    const double *r = env + atm[ATM_SLOTS * b[ATOM_OF] + PTR_COORD];
    s.r[0] = r[0]; s.r[1] = r[1]; s.r[2] = r[2];
    return s;
}

static Shell dummy_shell(const double *r)
{
    Shell s;
    s.l = 0; s.nprim = 1; s.nctr = 1; s.dummy = 1;
    s.exps = &ZERO; s.coef = &ONE;
    s.r[0] = r[0]; s.r[1] = r[1]; s.r[2] = r[2];
    return s;
}

/* ------------------------------------------------ real solid harmonics (cart->sph)
 * c2s[m + l][icart], m = -l..l, cartesian order lx descending then ly descending.
 * Standard real solid harmonics (Helgaker, Jorgensen, Olsen, "Molecular Electronic-
 * Structure Theory", eq. 6.4.47) scaled by sqrt((2l+1)/4pi) so that the angular part
 * is normalised on the unit sphere.  l=1 is ordered x,y,z (pyscf/lib/parameters.py:69-77).
 */
static double binom(int n, int k)
{
    if (k < 0 || k > n) return 0;
    double r = 1;
    for (int i = 1; i <= k; i++) r = r * (n - k + i) / i;
    return r;
}
static double fact(int n) { double r = 1; for (int i = 2; i <= n; i++) r *= i; return r; }

static int cart_index(int l, int lx, int ly)
{
    /* lx descending, then ly descending */
    int idx = 0;
    for (int x = l; x > lx; x--) idx += (l - x + 1);
    return idx + (l - lx - ly);
}

void oracle_c2s_matrix(int l, double *c2s /* [(2l+1)][ncart] */)
{
    int nc = NCART(l);
    memset(c2s, 0, sizeof(double) * (2 * l + 1) * nc);
    for (int m = -l; m <= l; m++) {
        int am = abs(m);
        double N = 1.0 / (pow(2.0, am) * fact(l)) *
                   sqrt(2.0 * fact(l + am) * fact(l - am) / (m == 0 ? 2.0 : 1.0));
        N *= sqrt((2 * l + 1) / (4 * M_PI));
        int row;
        if (l == 1) row = (m == 1) ? 0 : (m == -1 ? 1 : 2);   /* x, y, z */
        else row = m + l;
        for (int t = 0; t <= (l - am) / 2; t++)
            for (int u = 0; u <= t; u++) {
                int kmax = (m >= 0) ? am / 2 : (am - 1) / 2;
                for (int k = 0; k <= kmax; k++) {
                    int twov = (m >= 0) ? 2 * k : 2 * k + 1;
                    double c = ((t + k) % 2 ? -1.0 : 1.0) * pow(0.25, t) * binom(l, t) *
                               binom(l - t, am + t) * binom(t, u) * binom(am, twov);
                    int ly = 2 * u + twov;
                    int lx = 2 * t + am - ly;
                    int lz = l - 2 * t - am;
                    if (lx < 0 || lz < 0) continue;
                    c2s[row * nc + cart_index(l, lx, ly)] += N * c;
                }
            }
    }
}

/* ---------------------------------------------------------------- Hermite E coefficients
 * E[i][j][t], 0<=i<=la, 0<=j<=lb, 0<=t<=i+j for one Cartesian direction.
 */
typedef double Earr[LMAX1][LMAX1][2 * LMAX1];

static void hermite_E(int la, int lb, double a, double b, double Ax, double Bx, Earr E)
{
    double p = a + b;
    double Px = (p > 0) ? (a * Ax + b * Bx) / p : Ax;
    double XPA = Px - Ax, XPB = Px - Bx, XAB = Ax - Bx;
    double mu = (p > 0) ? a * b / p : 0;
    double hp = 0.5 / p;
    memset(E, 0, sizeof(Earr));
    E[0][0][0] = exp(-mu * XAB * XAB);
    for (int i = 0; i < la; i++)
        for (int t = 0; t <= i + 1; t++) {
            double v = XPA * E[i][0][t] + (t + 1) * E[i][0][t + 1];
            if (t > 0) v += hp * E[i][0][t - 1];
            E[i + 1][0][t] = v;
        }
    for (int i = 0; i <= la; i++)
        for (int j = 0; j < lb; j++)
            for (int t = 0; t <= i + j + 1; t++) {
                double v = XPB * E[i][j][t] + (t + 1) * E[i][j][t + 1];
                if (t > 0) v += hp * E[i][j][t - 1];
                E[i][j + 1][t] = v;
            }
}

/* ---------------------------------------------------------------- Hermite Coulomb R_tuv
 * R[t][u][v] for t+u+v <= N, from R^n_000 = (-2 alpha)^n F_n(alpha |PQ|^2).
 */
#define NR (4 * LMAX + 1)
typedef struct { double v[NR][NR][NR]; } Rarr;

/* long-range attenuation erf(omega r12)/r12 (env[PTR_RANGE_OMEGA], pyscf/gto/mole.py:76-84): the Boys
 * moments become theta^(n+1/2) F_n(theta x) with theta = omega^2/(omega^2 + alpha); omega < 0 selects the
 * short-range complement erfc(|omega| r12)/r12 (libcint's sign convention), F_n(x) minus the above.  Set only around
 * the 2e integral calls by oracle/ref.py (never for the nuclear attraction). */
static double g_omega = 0.0;
void oracle_set_omega(double omega) { g_omega = omega; }

static void hermite_R(int N, double alpha, const double *PQ, Rarr *R)
{
    int n1 = N + 1;
    double *tmp = malloc(sizeof(double) * (size_t)(n1 + 1) * n1 * n1 * n1);
#define TMP(n, t, u, v) tmp[((((size_t)(n)) * n1 + (t)) * n1 + (u)) * n1 + (v)]
    double f[NR + 2];
    double x = alpha * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]);
    if (g_omega != 0) {
        double theta = g_omega * g_omega / (g_omega * g_omega + alpha);
        boys(N, x * theta, f);
        double th = sqrt(theta);
        for (int n = 0; n <= N; n++) { f[n] *= th; th *= theta; }
        if (g_omega < 0) {                  /* short range erfc(|omega| r12)/r12 = Coulomb - long range */
            double fc[NR + 2];
            boys(N, x, fc);
            for (int n = 0; n <= N; n++) f[n] = fc[n] - f[n];
        }
    } else {
        boys(N, x, f);
    }
    double m2a = 1;
    for (int n = 0; n <= N; n++) { TMP(n, 0, 0, 0) = m2a * f[n]; m2a *= -2 * alpha; }
    /* increasing total order L = t+u+v:  R^n_{t+1,u,v} = t R^{n+1}_{t-1,u,v} + X R^{n+1}_{t,u,v} */
    for (int L = 1; L <= N; L++)
        for (int n = 0; n <= N - L; n++)
            for (int t = 0; t <= L; t++)
                for (int u = 0; u <= L - t; u++) {
                    int v = L - t - u;
                    double val;
                    if (t > 0) {
                        val = PQ[0] * TMP(n + 1, t - 1, u, v);
                        if (t > 1) val += (t - 1) * TMP(n + 1, t - 2, u, v);
                    } else if (u > 0) {
                        val = PQ[1] * TMP(n + 1, t, u - 1, v);
                        if (u > 1) val += (u - 1) * TMP(n + 1, t, u - 2, v);
                    } else {
                        val = PQ[2] * TMP(n + 1, t, u, v - 1);
                        if (v > 1) val += (v - 1) * TMP(n + 1, t, u, v - 2);
                    }
                    TMP(n, t, u, v) = val;
                }
    for (int t = 0; t <= N; t++)
        for (int u = 0; u <= N - t; u++)
            for (int v = 0; v <= N - t - u; v++) R->v[t][u][v] = TMP(0, t, u, v);
#undef TMP
    free(tmp);
}

static void cart_list(int l, int *lx, int *ly, int *lz)
{
    int n = 0;
    for (int x = l; x >= 0; x--)
        for (int y = l - x; y >= 0; y--) { lx[n] = x; ly[n] = y; lz[n] = l - x - y; n++; }
}

/* ------------------------------------------------ generic contracted 2e block, cartesian
 * out[d][c][b][a] (a fastest), each index = ctr*ncart + cart.
 * Shells may be "dummy" (s, exponent 0) to obtain 3- and 2-centre integrals.
 */
static void eri_cart(const Shell *A, const Shell *B, const Shell *C, const Shell *D, double *out)
{
    int la = A->l, lb = B->l, lc = C->l, ld = D->l;
    int na = NCART(la), nb = NCART(lb), nc = NCART(lc), nd = NCART(ld);
    int da = na * A->nctr, db = nb * B->nctr, dc = nc * C->nctr, dd = nd * D->nctr;
    size_t ntot = (size_t)da * db * dc * dd;
    memset(out, 0, sizeof(double) * ntot);
    int ax[64], ay[64], az[64], bx[64], by[64], bz[64], cx[64], cy[64], cz[64], dx[64], dy[64], dz[64];
    cart_list(la, ax, ay, az); cart_list(lb, bx, by, bz);
    cart_list(lc, cx, cy, cz); cart_list(ld, dx, dy, dz);
    int Lab = la + lb, Lcd = lc + ld, N = Lab + Lcd;
    int nh = (Lab + 1);
    Earr *Eab = malloc(3 * sizeof(Earr)), *Ecd = malloc(3 * sizeof(Earr));
    Rarr *R = malloc(sizeof(Rarr));
    size_t ncd = (size_t)nc * nd;
    /* G[t][u][v][cd] */
    double *G = malloc(sizeof(double) * nh * nh * nh * ncd);
    double *prim = malloc(sizeof(double) * (size_t)na * nb * nc * nd);

    for (int pa = 0; pa < A->nprim; pa++)
    for (int pb = 0; pb < B->nprim; pb++) {
        double a = A->exps[pa], b = B->exps[pb], p = a + b;
        double P[3];
        for (int k = 0; k < 3; k++) {
            P[k] = (a * A->r[k] + b * B->r[k]) / p;
            hermite_E(la, lb, a, b, A->r[k], B->r[k], Eab[k]);
        }
        for (int pc = 0; pc < C->nprim; pc++)
        for (int pd = 0; pd < D->nprim; pd++) {
            double c = C->exps[pc], d = D->exps[pd], q = c + d;
            double Q[3], PQ[3];
            for (int k = 0; k < 3; k++) {
                Q[k] = (c * C->r[k] + d * D->r[k]) / q;
                PQ[k] = P[k] - Q[k];
                hermite_E(lc, ld, c, d, C->r[k], D->r[k], Ecd[k]);
            }
            double alpha = p * q / (p + q);
            hermite_R(N, alpha, PQ, R);
            double pref = 2 * pow(M_PI, 2.5) / (p * q * sqrt(p + q));
            /* step 1: contract ket Hermite expansion */
            for (int t = 0; t <= Lab; t++)
            for (int u = 0; u <= Lab - t; u++)
            for (int v = 0; v <= Lab - t - u; v++) {
                double *g = G + (((size_t)t * nh + u) * nh + v) * ncd;
                for (int ic = 0; ic < nc; ic++)
                for (int id = 0; id < nd; id++) {
                    double s = 0;
                    for (int tt = 0; tt <= cx[ic] + dx[id]; tt++) {
                        double ex = Ecd[0][cx[ic]][dx[id]][tt];
                        for (int uu = 0; uu <= cy[ic] + dy[id]; uu++) {
                            double exy = ex * Ecd[1][cy[ic]][dy[id]][uu];
                            for (int vv = 0; vv <= cz[ic] + dz[id]; vv++) {
                                double e = exy * Ecd[2][cz[ic]][dz[id]][vv];
                                double sg = ((tt + uu + vv) & 1) ? -1.0 : 1.0;
                                s += sg * e * R->v[t + tt][u + uu][v + vv];
                            }
                        }
                    }
                    g[ic * nd + id] = s;
                }
            }
            /* step 2: contract bra */
            for (int ia = 0; ia < na; ia++)
            for (int ib = 0; ib < nb; ib++) {
                double *o = prim + ((size_t)ia * nb + ib) * ncd;
                for (size_t k = 0; k < ncd; k++) o[k] = 0;
                for (int t = 0; t <= ax[ia] + bx[ib]; t++) {
                    double ex = Eab[0][ax[ia]][bx[ib]][t];
                    for (int u = 0; u <= ay[ia] + by[ib]; u++) {
                        double exy = ex * Eab[1][ay[ia]][by[ib]][u];
                        for (int v = 0; v <= az[ia] + bz[ib]; v++) {
                            double e = exy * Eab[2][az[ia]][bz[ib]][v];
                            const double *g = G + (((size_t)t * nh + u) * nh + v) * ncd;
                            for (size_t k = 0; k < ncd; k++) o[k] += e * g[k];
                        }
                    }
                }
            }
            /* contraction */
            for (int kd = 0; kd < D->nctr; kd++)
            for (int kc = 0; kc < C->nctr; kc++)
            for (int kb = 0; kb < B->nctr; kb++)
            for (int ka = 0; ka < A->nctr; ka++) {
                double cc = pref * A->coef[ka * A->nprim + pa] * B->coef[kb * B->nprim + pb] *
                            C->coef[kc * C->nprim + pc] * D->coef[kd * D->nprim + pd];
                if (cc == 0) continue;
                for (int id = 0; id < nd; id++)
                for (int ic = 0; ic < nc; ic++)
                for (int ib = 0; ib < nb; ib++)
                for (int ia = 0; ia < na; ia++) {
                    size_t o = ((((size_t)(kd * nd + id) * dc + (kc * nc + ic)) * db + (kb * nb + ib)) * da) +
                               (ka * na + ia);
                    out[o] += cc * prim[(((size_t)ia * nb + ib) * nc + ic) * nd + id];
                }
            }
        }
    }
    free(Eab); free(Ecd); free(R); free(G); free(prim);
}

/* transform one index of a tensor from cartesian to real spherical.
 * in : [nouter][nctr*ncart][ninner]  ->  out: [nouter][nctr*nsph][ninner] */
static void c2s_index(const Shell *S, const double *in, double *out, size_t nouter, size_t ninner)
{
    int l = S->l, nc = NCART(l), ns = 2 * l + 1;
    if (S->dummy) { memcpy(out, in, sizeof(double) * nouter * ninner); return; }
    double *mat = malloc(sizeof(double) * ns * nc);
    oracle_c2s_matrix(l, mat);
    for (size_t o = 0; o < nouter; o++)
        for (int k = 0; k < S->nctr; k++)
            for (int m = 0; m < ns; m++) {
                double *dst = out + ((o * S->nctr + k) * ns + m) * ninner;
                for (size_t i = 0; i < ninner; i++) dst[i] = 0;
                for (int c = 0; c < nc; c++) {
                    double f = mat[m * nc + c];
                    if (f == 0) continue;
                    const double *src = in + ((o * S->nctr + k) * nc + c) * ninner;
                    for (size_t i = 0; i < ninner; i++) dst[i] += f * src[i];
                }
            }
    free(mat);
}

static int nsph(const Shell *S) { return S->dummy ? 1 : (2 * S->l + 1) * S->nctr; }
static int ncartf(const Shell *S) { return NCART(S->l) * S->nctr; }

/* spherical block out[d][c][b][a] */
static void eri_sph(const Shell *A, const Shell *B, const Shell *C, const Shell *D, double *out)
{
    size_t ca = ncartf(A), cb = ncartf(B), cc = ncartf(C), cd = ncartf(D);
    size_t sa = nsph(A), sb = nsph(B), sc = nsph(C), sd = nsph(D);
    double *b0 = malloc(sizeof(double) * ca * cb * cc * cd);
    double *b1 = malloc(sizeof(double) * ca * cb * cc * cd);
    eri_cart(A, B, C, D, b0);
    c2s_index(A, b0, b1, cd * cc * cb, 1);           /* [d][c][b][a->sa] */
    c2s_index(B, b1, b0, cd * cc, sa);               /* [d][c][sb][sa] */
    c2s_index(C, b0, b1, cd, sb * sa);
    c2s_index(D, b1, b0, 1, sc * sb * sa);
    memcpy(out, b0, sizeof(double) * sa * sb * sc * sd);
    free(b0); free(b1);
}

static void make_ao_loc(const int *bas, int nbas, int *loc)
{
    loc[0] = 0;
    for (int i = 0; i < nbas; i++)
        loc[i + 1] = loc[i] + (2 * bas[i * BAS_SLOTS + ANG_OF] + 1) * bas[i * BAS_SLOTS + NCTR_OF];
}

/* (ij|k): i,j in [0,nbas_ao), k in [nbas_ao, nbas_ao+nbas_aux) of a concatenated bas table.
 * out[k][i][j]  (naux, nao, nao) C-order, aosym s1. */
void oracle_int3c2e(double *out, const int *atm, int natm, const int *bas, int nbas_ao,
                    int nbas_aux, const double *env)
{
    int nbas = nbas_ao + nbas_aux;
    int *loc = malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(bas, nbas, loc);
    int nao = loc[nbas_ao];
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int ks = 0; ks < nbas_aux; ks++)
    for (int is = 0; is < nbas_ao; is++) {
        Shell K = get_shell(atm, bas, env, nbas_ao + ks);
        Shell Kd = dummy_shell(K.r);
        Shell I = get_shell(atm, bas, env, is);
        int dk = nsph(&K), di = nsph(&I);
        int k0 = loc[nbas_ao + ks] - nao, i0 = loc[is];
        for (int js = 0; js <= is; js++) {
            Shell J = get_shell(atm, bas, env, js);
            int dj = nsph(&J), j0 = loc[js];
            double *buf = malloc(sizeof(double) * di * dj * dk);
            eri_sph(&I, &J, &K, &Kd, buf);   /* buf[k][j][i] */
            for (int k = 0; k < dk; k++)
                for (int j = 0; j < dj; j++)
                    for (int i = 0; i < di; i++) {
                        double v = buf[((size_t)k * dj + j) * di + i];
                        out[((size_t)(k0 + k) * nao + (i0 + i)) * nao + (j0 + j)] = v;
                        out[((size_t)(k0 + k) * nao + (j0 + j)) * nao + (i0 + i)] = v;
                    }
            free(buf);
        }
    }
    free(loc);
}

/* Packed (s2ij) slab of the same integrals: AO shells i in [ish0, ish1), every j <= i, every aux function.
 * out[k][pq - pq0], pq = p(p+1)/2 + q (p >= q), pq0 = p0(p0+1)/2 with p0 the first function of shell ish0; the row length
 * `ld` must be >= p1(p1+1)/2 - pq0.  This is the layout of the reference's `_cderi` rows before the metric solve
 * (pyscf/df/incore.py:178-199: getints3c with aosym='s2ij' over shell-range slabs). */
void oracle_int3c2e_slab(double *out, long ld, const int *atm, int natm, const int *bas, int nbas_ao,
                         int nbas_aux, const double *env, int ish0, int ish1)
{
    int nbas = nbas_ao + nbas_aux;
    int *loc = malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(bas, nbas, loc);
    int nao = loc[nbas_ao];
    long p0 = loc[ish0];
    long pq0 = p0 * (p0 + 1) / 2;
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int ks = 0; ks < nbas_aux; ks++)
    for (int is = ish0; is < ish1; is++) {
        Shell K = get_shell(atm, bas, env, nbas_ao + ks);
        Shell Kd = dummy_shell(K.r);
        Shell I = get_shell(atm, bas, env, is);
        int dk = nsph(&K), di = nsph(&I);
        int k0 = loc[nbas_ao + ks] - nao, i0 = loc[is];
        for (int js = 0; js <= is; js++) {
            Shell J = get_shell(atm, bas, env, js);
            int dj = nsph(&J), j0 = loc[js];
            double *buf = malloc(sizeof(double) * di * dj * dk);
            eri_sph(&I, &J, &K, &Kd, buf);   /* buf[k][j][i] */
            for (int k = 0; k < dk; k++)
                for (int j = 0; j < dj; j++)
                    for (int i = 0; i < di; i++) {
                        long p = i0 + i, q = j0 + j;
                        if (q > p) continue;             /* diagonal shell block: lower triangle only */
                        out[(size_t)(k0 + k) * ld + (p * (p + 1) / 2 + q - pq0)] = buf[((size_t)k * dj + j) * di + i];
                    }
            free(buf);
        }
    }
    free(loc);
}

/* Rectangular (s1) block of the same integrals: AO shells i in [ish0, ish1) x AO shells j in [jsh0, jsh1), every aux function.
 * out[k][(p - p0) * nj + (q - q0)] with p0 / q0 the first functions of shells ish0 / jsh0 and nj the functions of the j range.
 * Used by tools/gen_golden_shard_local.py (pairs (p, q in a local support) of a tensor too large to generate whole). */
void oracle_int3c2e_block(double *out, const int *atm, int natm, const int *bas, int nbas_ao,
                          int nbas_aux, const double *env, int ish0, int ish1, int jsh0, int jsh1)
{
    int nbas = nbas_ao + nbas_aux;
    int *loc = malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(bas, nbas, loc);
    int nao = loc[nbas_ao];
    long p0 = loc[ish0], q0 = loc[jsh0];
    long ni = loc[ish1] - p0, nj = loc[jsh1] - q0;
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int ks = 0; ks < nbas_aux; ks++)
    for (int is = ish0; is < ish1; is++) {
        Shell K = get_shell(atm, bas, env, nbas_ao + ks);
        Shell Kd = dummy_shell(K.r);
        Shell I = get_shell(atm, bas, env, is);
        int dk = nsph(&K), di = nsph(&I);
        int k0 = loc[nbas_ao + ks] - nao, i0 = loc[is];
        for (int js = jsh0; js < jsh1; js++) {
            Shell J = get_shell(atm, bas, env, js);
            int dj = nsph(&J), j0 = loc[js];
            double *buf = malloc(sizeof(double) * di * dj * dk);
            eri_sph(&I, &J, &K, &Kd, buf);   /* buf[k][j][i] */
            for (int k = 0; k < dk; k++)
                for (int j = 0; j < dj; j++)
                    for (int i = 0; i < di; i++)
                        out[(size_t)(k0 + k) * ni * nj + (size_t)(i0 + i - p0) * nj + (j0 + j - q0)] =
                            buf[((size_t)k * dj + j) * di + i];
            free(buf);
        }
    }
    free(loc);
}

/* (i|k) over the shells [sh0, sh1) of a bas table: out[i][k] symmetric, n x n */
void oracle_int2c2e(double *out, const int *atm, int natm, const int *bas, int sh0, int sh1,
                    const double *env)
{
    int nb = sh1 - sh0;
    int *loc = malloc(sizeof(int) * (nb + 1));
    make_ao_loc(bas + sh0 * BAS_SLOTS, nb, loc);
    int n = loc[nb];
#pragma omp parallel for schedule(dynamic)
    for (int is = 0; is < nb; is++) {
        Shell I = get_shell(atm, bas, env, sh0 + is);
        Shell Id = dummy_shell(I.r);
        int di = nsph(&I);
        for (int ks = 0; ks <= is; ks++) {
            Shell K = get_shell(atm, bas, env, sh0 + ks);
            Shell Kd = dummy_shell(K.r);
            int dk = nsph(&K);
            double *buf = malloc(sizeof(double) * di * dk);
            eri_sph(&I, &Id, &K, &Kd, buf);  /* buf[k][i] */
            for (int k = 0; k < dk; k++)
                for (int i = 0; i < di; i++) {
                    double v = buf[k * di + i];
                    out[(size_t)(loc[is] + i) * n + loc[ks] + k] = v;
                    out[(size_t)(loc[ks] + k) * n + loc[is] + i] = v;
                }
            free(buf);
        }
    }
    free(loc);
}

/* full (ij|kl), out[i][j][k][l], nao^4 - small molecules only */
void oracle_int2e(double *out, const int *atm, int natm, const int *bas, int nbas, const double *env)
{
    int *loc = malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(bas, nbas, loc);
    size_t n = loc[nbas];
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int is = 0; is < nbas; is++)
    for (int js = 0; js < nbas; js++) {
        if (js > is) continue;
        Shell I = get_shell(atm, bas, env, is), J = get_shell(atm, bas, env, js);
        int di = nsph(&I), dj = nsph(&J);
        for (int ks = 0; ks < nbas; ks++)
        for (int ls = 0; ls <= ks; ls++) {
            Shell K = get_shell(atm, bas, env, ks), L = get_shell(atm, bas, env, ls);
            int dk = nsph(&K), dl = nsph(&L);
            double *buf = malloc(sizeof(double) * di * dj * dk * dl);
            eri_sph(&I, &J, &K, &L, buf);  /* buf[l][k][j][i] */
            for (int l = 0; l < dl; l++)
            for (int k = 0; k < dk; k++)
            for (int j = 0; j < dj; j++)
            for (int i = 0; i < di; i++) {
                double v = buf[(((size_t)l * dk + k) * dj + j) * di + i];
                size_t a = loc[is] + i, b = loc[js] + j, c = loc[ks] + k, d = loc[ls] + l;
                out[((a * n + b) * n + c) * n + d] = v;
                out[((b * n + a) * n + c) * n + d] = v;
                out[((a * n + b) * n + d) * n + c] = v;
                out[((b * n + a) * n + d) * n + c] = v;
            }
            free(buf);
        }
    }
    free(loc);
}

/* ---------------------------------------------------------------- one-electron integrals
 * type 0: overlap, 1: kinetic, 2: nuclear attraction (sum_C -Z_C / |r-C|)   out[i][j] */
static void int1e_cart(int type, const Shell *A, const Shell *B, const int *atm, int natm,
                       const double *env, double *out /* [b][a] contracted cart */)
{
    int la = A->l, lb = B->l, na = NCART(la), nb = NCART(lb);
    int da = na * A->nctr, db = nb * B->nctr;
    memset(out, 0, sizeof(double) * da * db);
    int ax[64], ay[64], az[64], bx[64], by[64], bz[64];
    cart_list(la, ax, ay, az); cart_list(lb, bx, by, bz);
    Earr *E = malloc(3 * sizeof(Earr));
    Rarr *R = malloc(sizeof(Rarr));
    double *prim = malloc(sizeof(double) * na * nb);
    for (int pa = 0; pa < A->nprim; pa++)
    for (int pb = 0; pb < B->nprim; pb++) {
        double a = A->exps[pa], b = B->exps[pb], p = a + b, P[3];
        for (int k = 0; k < 3; k++) {
            P[k] = (a * A->r[k] + b * B->r[k]) / p;
            hermite_E(la, lb + 2, a, b, A->r[k], B->r[k], E[k]);
        }
        memset(prim, 0, sizeof(double) * na * nb);
        if (type == 0 || type == 1) {
            double f = pow(M_PI / p, 1.5);
            for (int ia = 0; ia < na; ia++)
            for (int ib = 0; ib < nb; ib++) {
                int i[3] = {ax[ia], ay[ia], az[ia]}, j[3] = {bx[ib], by[ib], bz[ib]};
                double S[3], T[3];
                for (int k = 0; k < 3; k++) {
                    S[k] = E[k][i[k]][j[k]][0];
                    double t = -2 * b * (2 * j[k] + 1) * E[k][i[k]][j[k]][0] +
                               4 * b * b * E[k][i[k]][j[k] + 2][0];
                    if (j[k] >= 2) t += j[k] * (j[k] - 1) * E[k][i[k]][j[k] - 2][0];
                    T[k] = -0.5 * t;
                }
                prim[ia * nb + ib] = (type == 0) ? f * S[0] * S[1] * S[2]
                    : f * (T[0] * S[1] * S[2] + S[0] * T[1] * S[2] + S[0] * S[1] * T[2]);
            }
        } else {
            for (int ic = 0; ic < natm; ic++) {
                const double *rc = env + atm[ic * ATM_SLOTS + PTR_COORD];
                double Z = atm[ic * ATM_SLOTS + 0];
                double PC[3] = {P[0] - rc[0], P[1] - rc[1], P[2] - rc[2]};
                hermite_R(la + lb, p, PC, R);
                double f = -Z * 2 * M_PI / p;
                for (int ia = 0; ia < na; ia++)
                for (int ib = 0; ib < nb; ib++) {
                    double s = 0;
                    for (int t = 0; t <= ax[ia] + bx[ib]; t++)
                    for (int u = 0; u <= ay[ia] + by[ib]; u++)
                    for (int v = 0; v <= az[ia] + bz[ib]; v++)
                        s += E[0][ax[ia]][bx[ib]][t] * E[1][ay[ia]][by[ib]][u] *
                             E[2][az[ia]][bz[ib]][v] * R->v[t][u][v];
                    prim[ia * nb + ib] += f * s;
                }
            }
        }
        for (int kb = 0; kb < B->nctr; kb++)
        for (int ka = 0; ka < A->nctr; ka++) {
            double cc = A->coef[ka * A->nprim + pa] * B->coef[kb * B->nprim + pb];
            for (int ib = 0; ib < nb; ib++)
            for (int ia = 0; ia < na; ia++)
                out[(kb * nb + ib) * da + ka * na + ia] += cc * prim[ia * nb + ib];
        }
    }
    free(E); free(R); free(prim);
}

void oracle_int1e(int type, double *out, const int *atm, int natm, const int *bas, int nbas,
                  const double *env)
{
    int *loc = malloc(sizeof(int) * (nbas + 1));
    make_ao_loc(bas, nbas, loc);
    int n = loc[nbas];
#pragma omp parallel for schedule(dynamic)
    for (int is = 0; is < nbas; is++) {
        Shell I = get_shell(atm, bas, env, is);
        for (int js = 0; js <= is; js++) {
            Shell J = get_shell(atm, bas, env, js);
            size_t ci = ncartf(&I), cj = ncartf(&J), si = nsph(&I), sj = nsph(&J);
            double *b0 = malloc(sizeof(double) * ci * cj), *b1 = malloc(sizeof(double) * ci * cj);
            int1e_cart(type, &I, &J, atm, natm, env, b0);   /* [j][i] */
            c2s_index(&I, b0, b1, cj, 1);
            c2s_index(&J, b1, b0, 1, si);
            for (size_t j = 0; j < sj; j++)
                for (size_t i = 0; i < si; i++) {
                    out[(size_t)(loc[is] + i) * n + loc[js] + j] = b0[j * si + i];
                    out[(size_t)(loc[js] + j) * n + loc[is] + i] = b0[j * si + i];
                }
            free(b0); free(b1);
        }
    }
    free(loc);
}

/* NPdunpack_tril analogue (pyscf/lib/np_helper/pack_tril.c:150-273), hermitian fill, OpenMP
 * over the leading index; used by the CPU-baseline leg so that the timed port does its unpack
 * in C like the reference does (AO2MOtranse2_nr_s2, nr_ao2mo.c:1016-1031). */
void oracle_unpack_tril(const double *tril, double *full, int count, int n)
{
    size_t npair = (size_t)n * (n + 1) / 2;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < count; c++) {
        const double *t = tril + c * npair;
        double *f = full + (size_t)c * n * n;
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++) {
                double v = t[(size_t)i * (i + 1) / 2 + j];
                f[(size_t)i * n + j] = v;
                f[(size_t)j * n + i] = v;
            }
    }
}
