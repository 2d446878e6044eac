// 4-centre 2-electron Coulomb integrals (ij|kl) on gfx950 for the in-core (non density-fitted) J/K of small
// molecules: BASELINE config 1 (H2O, `scf.RHF(mol)` without `.density_fit()`).
//
// What it replaces: `mol.intor('int2e', aosym='s8')` as called by RHF.get_jk (pyscf/scf/hf.py:2499-2511) -
// GTOnr2e_fill_drv (pyscf/lib/gto/fill_int2e.c:538) over libcint's int2e_sph; the tensor is then contracted by
// dot_eri_dm (pyscf/scf/hf.py:902, CVHFnrs8_incore_drv in pyscf/lib/vhf/nr_incore.c).  The arithmetic (libcint) is
// not in the reference tree: this is the same from-scratch Rys quadrature as int3c2e_kernel.h with the second
// electron carried by a shell PAIR, i.e. one more horizontal recurrence on the ket side:
//   (ab|cd) = 2 pi^{5/2} / (zeta eta sqrt(zeta+eta)) K_ab K_cd  sum_r  Ix(r) Iy(r) Iz(r)
//   2-D integrals: VRR on (n = 0..la+lb at A, m = 0..lc+ld at C), HRR A->B, HRR C->D.
//
// This is not a hot path (the north-star path is the density-fitted build): ONE kernel with run-time angular momenta
// instead of a template family.  Workgroup = one (bra shell pair, ket shell pair); the 3*NR two-dimensional
// recurrences are dealt to lanes, their results go through LDS, and every lane owns a strided set of the
// ncart^4 Cartesian components (accumulators in LDS).  The epilogue carries the four indices Cartesian -> real
// spherical one after the other through two LDS buffers and writes all 8 permutational images into the dense
// [nao][nao][nao][nao] tensor.
#include "common.h"
#include "rys_device.h"
#include "int2e_args.h"

namespace pamd {
namespace {

__host__ __device__ inline int ncart_rt(int l) { return (l + 1) * (l + 2) / 2; }

__device__ inline void cart_exps_rt(int l, int c, int &lx, int &ly, int &lz)
{
    int x = l;
    int rem = c;
    while (rem > l - x) { rem -= (l - x + 1); x--; }
    lx = x; ly = (l - x) - rem; lz = rem;
}

__device__ inline double rys_rw(int nr, const double *table, double x, int q)
{
    switch (nr) {
    case 1: return rys_root_or_weight<1>(table, x, q);
    case 2: return rys_root_or_weight<2>(table, x, q);
    case 3: return rys_root_or_weight<3>(table, x, q);
    case 4: return rys_root_or_weight<4>(table, x, q);
    case 5: return rys_root_or_weight<5>(table, x, q);
    case 6: return rys_root_or_weight<6>(table, x, q);
    case 7: return rys_root_or_weight<7>(table, x, q);
    default: return rys_root_or_weight<8>(table, x, q);
    }
}

// LDS plan of one class (doubles), shared by the kernel and the launcher
struct Plan {
    int nr, N, M, di, dj, dk, dl;
    int g2u, hbu, g4u;               // per (root, direction) unit: VRR table, bra-HRR table, final table
    int nci, ncj, nck, ncl, nsi, nsj, nsk, nsl;
    int o_rw, o_tab, o_g, o_hb, o_acc, total;
};

__host__ __device__ inline Plan make_plan(int li, int lj, int lk, int ll)
{
    Plan p;
    p.nr = (li + lj + lk + ll) / 2 + 1;
    p.N = li + lj; p.M = lk + ll;
    p.di = li + 1; p.dj = lj + 1; p.dk = lk + 1; p.dl = ll + 1;
    p.g2u = (p.N + 1) * (p.M + 1);
    p.hbu = p.di * p.dj * (p.M + 1);
    p.g4u = p.di * p.dj * p.dk * p.dl;
    p.nci = ncart_rt(li); p.ncj = ncart_rt(lj); p.nck = ncart_rt(lk); p.ncl = ncart_rt(ll);
    p.nsi = 2 * li + 1; p.nsj = 2 * lj + 1; p.nsk = 2 * lk + 1; p.nsl = 2 * ll + 1;
    const int gmax = (p.g2u > p.g4u ? p.g2u : p.g4u) * 3 * p.nr;       // the VRR tables are dead when g4 is written
    const int stage = p.nsi * p.ncj * p.nck * p.ncl;                   // first epilogue buffer (aliases g / hb)
    int work = gmax + 3 * p.nr * p.hbu;
    if (stage > work) work = stage;
    p.o_rw = 0;
    p.o_tab = 16;                                                      // 12 small int tables of 16 entries (as doubles: 96)
    p.o_g = p.o_tab + 96;
    p.o_hb = p.o_g + gmax;
    p.o_acc = p.o_g + work;
    p.total = p.o_acc + p.nci * p.ncj * p.nck * p.ncl;
    return p;
}

// MODE 0: write the 8 images into the dense tensor; 1: Schwarz pass (diagonal quartets only); 2: contract with densities
template <int MODE>
__global__ void int2e_kernel(Int2eDirectArgs xa)
{
    extern __shared__ double smem[];
    const Int2eArgs &a = xa.base;
    const int bra = blockIdx.x, ket = MODE == 1 ? blockIdx.x : blockIdx.y;
    if (MODE != 1 && a.same_class && ket > bra) return;
    if (MODE == 2 && xa.q_bra && xa.q_bra[bra] * xa.q_ket[ket] * xa.dm_max < xa.cutoff) return;      // CVHFnrs8_prescreen
    const Plan p = make_plan(a.li, a.lj, a.lk, a.ll);
    const int tid = threadIdx.x, nth = blockDim.x;
    const int NR = p.nr;
    double *rw = smem + p.o_rw;
    int *tab = reinterpret_cast<int *>(smem + p.o_tab);      // [4 indices][3 directions][16]: LDS offset of the exponent
    double *g = smem + p.o_g;
    double *hb = smem + p.o_hb;
    double *acc = smem + p.o_acc;

    const int ish = a.bra_ish[bra], jsh = a.bra_jsh[bra], ksh = a.ket_ish[ket], lsh = a.ket_jsh[ket];
    const int bp0 = a.bra_pp0[bra], bnp = a.bra_npp[bra], kp0 = a.ket_pp0[ket], knp = a.ket_npp[ket];
    double ab[3], cd[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        ab[d] = a.shell_xyz[ish * 3 + d] - a.shell_xyz[jsh * 3 + d];
        cd[d] = a.shell_xyz[ksh * 3 + d] - a.shell_xyz[lsh * 3 + d];
    }
    const int ncart4 = p.nci * p.ncj * p.nck * p.ncl;
    // exponent -> offset tables: offset(direction d) = T_i[ci] + T_j[cj] + T_k[ck] + T_l[cl] into one unit of g4
    if (tid < 64) {
        const int which = tid >> 4, c = tid & 15;
        const int l = which == 0 ? a.li : which == 1 ? a.lj : which == 2 ? a.lk : a.ll;
        const int stride = which == 0 ? p.dj * p.dk * p.dl : which == 1 ? p.dk * p.dl : which == 2 ? p.dl : 1;
        if (c < ncart_rt(l)) {
            int ex[3];
            cart_exps_rt(l, c, ex[0], ex[1], ex[2]);
#pragma unroll
            for (int d = 0; d < 3; d++) tab[(which * 3 + d) * 16 + c] = ex[d] * stride;
        }
    }
    for (int c = tid; c < ncart4; c += nth) acc[c] = 0.0;
    __syncthreads();

    for (int ib = 0; ib < bnp; ib++) {
        const double *pb = a.bra_pp + (long)(bp0 + ib) * 8;
        const double zeta = pb[0], px = pb[1], py = pb[2], pz = pb[3], ccb = pb[4];
        const double pa[3] = {pb[5], pb[6], pb[7]};
        for (int ik = 0; ik < knp; ik++) {
            const double *pk = a.ket_pp + (long)(kp0 + ik) * 8;
            const double eta = pk[0], qx = pk[1], qy = pk[2], qz = pk[3], cck = pk[4];
            const double qc[3] = {pk[5], pk[6], pk[7]};
            const double pq[3] = {px - qx, py - qy, pz - qz};
            const double ze = zeta + eta;
            const double rho = zeta * eta / ze;
            const double theta = (a.omega > 0) ? a.omega * a.omega / (a.omega * a.omega + rho) : 1.0;
            const double x = rho * (pq[0] * pq[0] + pq[1] * pq[1] + pq[2] * pq[2]) * theta;
            for (int q = tid; q < 2 * NR; q += nth) {
                const double v = rys_rw(NR, a.rys_table, x, q);
                rw[q] = (q < NR) ? v * theta : v * sqrt(theta);
            }
            __syncthreads();
            const double fac = 2.0 * 17.493418327624862846 /* pi^2.5 */ / (zeta * eta * sqrt(ze)) * ccb * cck;
            // ---- VRR + bra HRR, one (root, direction) unit per lane
            for (int un = tid; un < 3 * NR; un += nth) {
                const int r = un / 3, d = un - 3 * r;
                const double u = rw[r], w = rw[NR + r];
                const double ue = u * eta / ze, uz = u * zeta / ze;
                const double b00 = 0.5 * u / ze;
                const double b10 = (1.0 - ue) * 0.5 / zeta;
                const double b01 = (1.0 - uz) * 0.5 / eta;
                const double c00 = pa[d] - ue * pq[d];
                const double c0p = qc[d] + uz * pq[d];
                const int M1 = p.M + 1;
                double *t = g + un * p.g2u;                       // t[n][m]
                t[0] = (d == 2) ? w * fac : 1.0;
                for (int n = 0; n < p.N; n++)
                    t[(n + 1) * M1] = c00 * t[n * M1] + (n > 0 ? n * b10 * t[(n - 1) * M1] : 0.0);
                for (int m = 0; m < p.M; m++)
                    for (int n = 0; n <= p.N; n++) {
                        double v = c0p * t[n * M1 + m];
                        if (m > 0) v += m * b01 * t[n * M1 + m - 1];
                        if (n > 0) v += n * b00 * t[(n - 1) * M1 + m];
                        t[n * M1 + m + 1] = v;
                    }
                // bra HRR in place along n, (i, j, m) for i <= li saved after every step
                double *h = hb + un * p.hbu;                      // h[(i * dj + j) * M1 + m]
                for (int j = 0; j <= a.lj; j++) {
                    for (int i = 0; i <= a.li; i++)
                        for (int m = 0; m <= p.M; m++) h[(i * p.dj + j) * M1 + m] = t[i * M1 + m];
                    if (j < a.lj)
                        for (int n = 0; n < p.N - j; n++)
                            for (int m = 0; m <= p.M; m++) t[n * M1 + m] = t[(n + 1) * M1 + m] + ab[d] * t[n * M1 + m];
                }
            }
            __syncthreads();                                      // the g4 tables overwrite other units' VRR tables
            // ---- ket HRR: g4[((i * dj + j) * dk + k) * dl + l]
            for (int un = tid; un < 3 * NR; un += nth) {
                const int d = un % 3;
                const int M1 = p.M + 1;
                double *h = hb + un * p.hbu;
                double *o = g + un * p.g4u;
                for (int ij = 0; ij < p.di * p.dj; ij++) {
                    double *c = h + ij * M1;
                    for (int l = 0; l <= a.ll; l++) {
                        for (int k = 0; k <= a.lk; k++) o[(ij * p.dk + k) * p.dl + l] = c[k];
                        if (l < a.ll)
                            for (int m = 0; m < p.M - l; m++) c[m] = c[m + 1] + cd[d] * c[m];
                    }
                }
            }
            __syncthreads();
            // ---- contract the roots into the Cartesian components of this lane
            for (int c = tid; c < ncart4; c += nth) {
                int rem = c;
                const int cl = rem % p.ncl; rem /= p.ncl;
                const int ck = rem % p.nck; rem /= p.nck;
                const int cj = rem % p.ncj;
                const int ci = rem / p.ncj;
                const int ox = tab[0 * 16 + ci] + tab[3 * 16 + cj] + tab[6 * 16 + ck] + tab[9 * 16 + cl];
                const int oy = tab[1 * 16 + ci] + tab[4 * 16 + cj] + tab[7 * 16 + ck] + tab[10 * 16 + cl] + p.g4u;
                const int oz = tab[2 * 16 + ci] + tab[5 * 16 + cj] + tab[8 * 16 + ck] + tab[11 * 16 + cl] + 2 * p.g4u;
                double s = 0;
                for (int r = 0; r < NR; r++) {
                    const double *gr = g + r * 3 * p.g4u;
                    s += gr[ox] * gr[oy] * gr[oz];
                }
                acc[c] += s;
            }
            __syncthreads();
        }
    }

    // ================= epilogue: the four indices Cartesian -> real spherical =========================
    const double *c2s_i = a.c2s + a.c2s_off[a.li];
    const double *c2s_j = a.c2s + a.c2s_off[a.lj];
    const double *c2s_k = a.c2s + a.c2s_off[a.lk];
    const double *c2s_l = a.c2s + a.c2s_off[a.ll];
    double *b1 = g;                                              // [mi][cj][ck][cl]
    {
        const int inner = p.ncj * p.nck * p.ncl;
        for (int o = tid; o < p.nsi * inner; o += nth) {
            const int mi = o / inner, rest = o - mi * inner;
            double v = 0;
            for (int ci = 0; ci < p.nci; ci++) v += c2s_i[mi * p.nci + ci] * acc[ci * inner + rest];
            b1[o] = v;
        }
    }
    __syncthreads();
    double *b2 = acc;                                            // [mi][mj][ck][cl]
    {
        const int inner = p.nck * p.ncl;
        for (int o = tid; o < p.nsi * p.nsj * inner; o += nth) {
            const int rest = o % inner, mj = (o / inner) % p.nsj, mi = o / (inner * p.nsj);
            double v = 0;
            for (int cj = 0; cj < p.ncj; cj++) v += c2s_j[mj * p.ncj + cj] * b1[(mi * p.ncj + cj) * inner + rest];
            b2[o] = v;
        }
    }
    __syncthreads();
    double *b3 = g;                                              // [mi][mj][mk][cl]
    {
        const int nij = p.nsi * p.nsj;
        for (int o = tid; o < nij * p.nsk * p.ncl; o += nth) {
            const int cl = o % p.ncl, mk = (o / p.ncl) % p.nsk, ij = o / (p.ncl * p.nsk);
            double v = 0;
            for (int ck = 0; ck < p.nck; ck++) v += c2s_k[mk * p.nck + ck] * b2[(ij * p.nck + ck) * p.ncl + cl];
            b3[o] = v;
        }
    }
    __syncthreads();
    {
        const long n = a.nao;
        const long p0 = a.shell_ao0[ish], q0 = a.shell_ao0[jsh], r0 = a.shell_ao0[ksh], s0 = a.shell_ao0[lsh];
        double qmax = 0;
        for (int o = tid; o < p.nsi * p.nsj * p.nsk * p.nsl; o += nth) {
            const int ml = o % p.nsl, ijk = o / p.nsl;
            const int mk = ijk % p.nsk, mj = (ijk / p.nsk) % p.nsj, mi = ijk / (p.nsk * p.nsj);
            if (MODE == 1) {
                if (mi != mk || mj != ml) continue;               // (ij|ij): the Schwarz diagonal
            } else {
                // an element that is its own permutational image of another one of this block is written once (from the
                // canonical member), so the tensor is exactly symmetric and independent of the write order
                if ((ish == jsh && mj > mi) || (ksh == lsh && ml > mk) ||
                    (ish == ksh && jsh == lsh && mi * p.nsj + mj < mk * p.nsl + ml)) continue;
            }
            double v = 0;
            for (int cl = 0; cl < p.ncl; cl++) v += c2s_l[ml * p.ncl + cl] * b3[ijk * p.ncl + cl];
            const long P = p0 + mi, Q = q0 + mj, R = r0 + mk, S = s0 + ml;
            if (MODE == 1) {
                qmax = fmax(qmax, fabs(v));
            } else if (MODE == 0) {
                double *e = a.eri;
                e[((P * n + Q) * n + R) * n + S] = v;
                e[((Q * n + P) * n + R) * n + S] = v;
                e[((P * n + Q) * n + S) * n + R] = v;
                e[((Q * n + P) * n + S) * n + R] = v;
                e[((R * n + S) * n + P) * n + Q] = v;
                e[((S * n + R) * n + P) * n + Q] = v;
                e[((R * n + S) * n + Q) * n + P] = v;
                e[((S * n + R) * n + Q) * n + P] = v;
            } else {
                // the distinct permutational images (a b | c d) of this canonical integral, each contracted once:
                //   J[a][b] += v D[c][d],  K[a][c] += v D[b][d]     (vj_ij = sum_kl (ij|kl) D_lk, vk_il = sum_jk (ij|kl) D_jk with
                //   symmetric handling as in CVHFdot_nrs8; for a non-symmetric D the images use D as stored: hermi = 0 safe
                //   because every image (a b|c d) and its partner (a b|d c) are both visited)
                const long im[8][4] = {{P, Q, R, S}, {Q, P, R, S}, {P, Q, S, R}, {Q, P, S, R},
                                       {R, S, P, Q}, {S, R, P, Q}, {R, S, Q, P}, {S, R, Q, P}};
                for (int t = 0; t < 8; t++) {
                    bool dup = false;
                    for (int u = 0; u < t; u++)
                        dup = dup || (im[u][0] == im[t][0] && im[u][1] == im[t][1] && im[u][2] == im[t][2] && im[u][3] == im[t][3]);
                    if (dup) continue;
                    const long A_ = im[t][0], B_ = im[t][1], C_ = im[t][2], D_ = im[t][3];
                    for (int sset = 0; sset < xa.nset; sset++) {
                        const double *dm = xa.dm + (long)sset * n * n;
                        if (xa.vj) unsafeAtomicAdd(xa.vj + (long)sset * n * n + A_ * n + B_, v * dm[D_ * n + C_]);
                        if (xa.vk) unsafeAtomicAdd(xa.vk + (long)sset * n * n + A_ * n + D_, v * dm[B_ * n + C_]);
                    }
                }
            }
        }
        if (MODE == 1) {
            // block maximum -> q_out[bra] = sqrt(max |(ij|ij)|)
            double *red = smem;                                   // everything else in LDS is dead now
            __syncthreads();
            red[tid] = qmax;
            __syncthreads();
            if (tid == 0) {
                double mx = 0;
                for (int t = 0; t < nth; t++) mx = fmax(mx, red[t]);
                xa.q_out[bra] = sqrt(mx);
            }
        }
    }
}

}  // namespace
}  // namespace pamd

static int launch_int2e(const pamd::Int2eDirectArgs &xa, int mode, void *stream)
{
    using namespace pamd;
    const Int2eArgs &a = xa.base;
    PAMD_REQUIRE(a.li >= a.lj && a.lk >= a.ll && a.lj >= 0 && a.ll >= 0, "int2e class needs l_i >= l_j and l_k >= l_l");
    PAMD_REQUIRE(a.li <= 3 && a.lk <= 3, "int2e: AO angular momentum > 3 unsupported on the 4-centre path");
    const Plan p = make_plan(a.li, a.lj, a.lk, a.ll);
    PAMD_REQUIRE(p.nr <= RYS_NMAX, "int2e: more Rys roots than the tables hold");
    const size_t lds = (size_t)p.total * sizeof(double);
    PAMD_REQUIRE(lds <= 160 * 1024, "int2e: LDS budget exceeded");
    if (a.nbra == 0 || a.nket == 0) return 0;
    static bool attr_done[64] = {false};        // per device (the attribute lives in the device's code object); setting it twice is harmless
    int dev = 0;
    PAMD_CHECK_HIP(hipGetDevice(&dev));
    bool &attr_set = attr_done[dev & 63];
    if (!attr_set) {
        PAMD_CHECK_HIP(hipFuncSetAttribute((const void *)int2e_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PAMD_CHECK_HIP(hipFuncSetAttribute((const void *)int2e_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PAMD_CHECK_HIP(hipFuncSetAttribute((const void *)int2e_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int ncart4 = p.nci * p.ncj * p.nck * p.ncl;
    const int nth = ncart4 <= 64 ? 64 : ncart4 <= 1024 ? 128 : 256;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(int2e_kernel<0>, dim3(a.nbra, a.nket), dim3(nth), lds, st, xa);
    else if (mode == 1) hipLaunchKernelGGL(int2e_kernel<1>, dim3(a.nbra, 1), dim3(nth), lds, st, xa);
    else hipLaunchKernelGGL(int2e_kernel<2>, dim3(a.nbra, a.nket), dim3(nth), lds, st, xa);
    PAMD_CHECK_LAUNCH();
    return 0;
}

extern "C" {

int PAMD_int2e_class(const PAMD_int2e_args *args, void *stream)
{
    pamd::Int2eDirectArgs xa;
    memset(&xa, 0, sizeof(xa));
    xa.base = *args;
    return launch_int2e(xa, 0, stream);
}

// Integral-direct J/K (or, with q_out set, the Schwarz factors of the bra list; then base.ket_* must repeat base.bra_*)
int PAMD_int2e_direct_class(const PAMD_int2e_direct_args *args, void *stream)
{
    PAMD_REQUIRE(args->q_out || ((args->vj || args->vk) && args->dm && args->nset > 0), "int2e direct: nothing to compute");
    return launch_int2e(*args, args->q_out ? 1 : 2, stream);
}

}  // extern "C"
