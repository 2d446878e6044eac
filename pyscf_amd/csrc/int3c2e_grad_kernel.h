// Nuclear-gradient contraction of the 3-centre Coulomb integrals, generate-and-contract in place:
//
//   grad[A] += sum_{pq,Q} Z[pq][Q] * d(pq|Q)/dA
//
// What it replaces: the derivative-integral tensors int3c2e_ip1 / int3c2e_ip2 / int2c2e_ip1 that
// pyscf/df/grad/rhf.py:117-199 (get_jk) materialises block by block through `_int3c_wrapper`
// (libcint) and then contracts with rhoj / rhok on the host.  Here no derivative tensor exists:
// every shell triple's derivative block is formed in registers and dotted with the matching block
// of the (back-transformed) two-particle density Z, and six numbers per triple leave the kernel.
//
//   d/dA_x [x_A^a e^{-alpha r_A^2}] = -( a x_A^{a-1} - 2 alpha x_A^{a+1} ) e^{-alpha r_A^2}
// acts on the 2-D Rys integrals: gx(i,j,k) -> i gx(i-1,j,k) - 2 alpha_i gx(i+1,j,k), so the only
// change to the integral kernel is one more level of the recurrences (i <= LI+1, j <= LJ+1, one more
// Rys root when L is even).  The derivative with respect to the third centre follows from
// translational invariance: d/dC = -(d/dA + d/dB).
//
// Same work decomposition as int3c2e_kernel (workgroup = shell pair x NT aux shells, S lanes per
// triple).  Prologue: the Z block is carried spherical -> Cartesian through LDS (the transpose of the
// integral kernel's epilogue), so the primitive loop works on Cartesian components only.
#pragma once
#include "int3c2e_kernel.h"

namespace pamd {

template <int LI, int LJ, int LK>
struct GG3 {
    using G = G3<LI, LJ, LK>;
    static constexpr int NR = (LI + LJ + LK + 1) / 2 + 1;
    static constexpr int DI = LI + 2, DJ = LJ + 2, DK = LK + 1;
    static constexpr int GU = DI * DJ * DK;
    static constexpr int GT = 3 * NR * GU;
    static constexpr int TSTRIDE = (GT + 2 * NR) | 1;
    static constexpr int STAGE1 = G::NIJ * G::NSK;               // [ci][cj][mk]  (also holds [mi][mj][mk])
    static constexpr int STAGE2 = G::NCI * G::NSJ * G::NSK;      // [ci][mj][mk]
    static constexpr int ESTRIDE = ((STAGE1 + STAGE2) > TSTRIDE ? (STAGE1 + STAGE2) : TSTRIDE) | 1;
};

template <int LI, int LJ, int LK, int S, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void int3c2e_grad_kernel(Int3c2eGradArgs ga)
{
    const Int3c2eArgs &a = ga.base;
    using G = G3<LI, LJ, LK>;
    using GG = GG3<LI, LJ, LK>;
    constexpr int NR = GG::NR;
    constexpr int NT = NTHREADS / S;
    constexpr int NE = (G::NIJ + S - 1) / S;
    constexpr int MAXPP = 64;
    constexpr int DJ = GG::DJ, DK = GG::DK, GU = GG::GU;

    extern __shared__ double smem[];
    double *s_pp = smem;                                   // [MAXPP][8]
    double *s_ab = smem + MAXPP * 8;                       // [MAXPP][2]
    double *s_t = smem + MAXPP * 10;                       // [NT][ESTRIDE]

    const int tid = threadIdx.x;
    const int t = tid / S;
    const int s = tid - t * S;
    const int ipair = blockIdx.x;
    const int kidx = blockIdx.y * NT + t;
    const bool kvalid = kidx < a.naux_cls;
    const int kk = kvalid ? kidx : a.naux_cls - 1;

    const int ish = a.pair_ish[ipair], jsh = a.pair_jsh[ipair];
    const int pp0 = a.pair_pp0[ipair], npp = a.pair_npp[ipair];
    const double ax_ = a.shell_xyz[ish * 3 + 0], ay_ = a.shell_xyz[ish * 3 + 1], az_ = a.shell_xyz[ish * 3 + 2];
    const double abx = ax_ - a.shell_xyz[jsh * 3 + 0];
    const double aby = ay_ - a.shell_xyz[jsh * 3 + 1];
    const double abz = az_ - a.shell_xyz[jsh * 3 + 2];
    const double cx = a.aux_xyz[kk * 3 + 0], cy = a.aux_xyz[kk * 3 + 1], cz = a.aux_xyz[kk * 3 + 2];

    double *my = s_t + t * GG::ESTRIDE;
    double *rw = my;
    double *gbuf = my + 2 * NR;

    int ix[NE], iy[NE], iz[NE], jx[NE], jy[NE], jz[NE];
#pragma unroll
    for (int el = 0; el < NE; el++) {
        int e = s + el * S;
        if (e >= G::NIJ) e = 0;
        int ci = e / G::NCJ, cj = e - ci * G::NCJ;
        cart_exps(LI, ci, ix[el], iy[el], iz[el]);
        cart_exps(LJ, cj, jx[el], jy[el], jz[el]);
    }

    // ================= prologue: Z block, spherical -> Cartesian ====================================
    const double *c2s_i = a.c2s + a.c2s_off[LI];
    const double *c2s_j = a.c2s + a.c2s_off[LJ];
    const double *c2s_k = a.c2s + a.c2s_off[LK];
    double zc[NE][G::NCK];
    {
        double *st0 = my;                          // [mi][mj][mk], later [ci][cj][mk]
        double *st1 = my + GG::STAGE1;             // [ci][mj][mk]
        const int p0 = a.shell_ao0[ish], q0 = a.shell_ao0[jsh];
        const int f0 = a.aux_f0[kk];
        const double pairfac = (a.tril && ish != jsh) ? 2.0 : 1.0;
        for (int o = s; o < G::NSI * G::NSJ * G::NSK; o += S) {
            int mk = o % G::NSK;
            int mj = (o / G::NSK) % G::NSJ;
            int mi = o / (G::NSK * G::NSJ);
            long p = p0 + mi, q = q0 + mj;
            long row = a.tril ? ((p >= q) ? p * (p + 1) / 2 + q : q * (q + 1) / 2 + p) : p;
            st0[o] = kvalid ? pairfac * a.T[(row - a.row_offset) * a.ldT + f0 + mk] : 0.0;
        }
        __syncthreads();
        for (int o = s; o < G::NCI * G::NSJ * G::NSK; o += S) {
            int mk = o % G::NSK;
            int mj = (o / G::NSK) % G::NSJ;
            int ci = o / (G::NSK * G::NSJ);
            double v = 0;
            for (int mi = 0; mi < G::NSI; mi++) v += c2s_i[mi * G::NCI + ci] * st0[(mi * G::NSJ + mj) * G::NSK + mk];
            st1[o] = v;
        }
        __syncthreads();
        for (int o = s; o < G::NIJ * G::NSK; o += S) {
            int mk = o % G::NSK;
            int e = o / G::NSK;
            int ci = e / G::NCJ, cj = e - ci * G::NCJ;
            double v = 0;
            for (int mj = 0; mj < G::NSJ; mj++) v += c2s_j[mj * G::NCJ + cj] * st1[(ci * G::NSJ + mj) * G::NSK + mk];
            st0[o] = v;
        }
        __syncthreads();
#pragma unroll
        for (int el = 0; el < NE; el++) {
            int e = s + el * S;
#pragma unroll
            for (int c = 0; c < G::NCK; c++) {
                double v = 0;
                if (e < G::NIJ)
                    for (int mk = 0; mk < G::NSK; mk++) v += c2s_k[mk * G::NCK + c] * st0[e * G::NSK + mk];
                zc[el][c] = v;
            }
        }
    }
    double gA[3] = {0, 0, 0}, gB[3] = {0, 0, 0};

    for (int ppb = 0; ppb < npp; ppb += MAXPP) {
        const int nppb = (npp - ppb < MAXPP) ? npp - ppb : MAXPP;
        __syncthreads();
        for (int e = tid; e < nppb * 8; e += NTHREADS) s_pp[e] = a.pp[(long)(pp0 + ppb) * 8 + e];
        for (int e = tid; e < nppb * 2; e += NTHREADS) s_ab[e] = ga.pp_ab[(long)(pp0 + ppb) * 2 + e];
        __syncthreads();
        for (int ip = 0; ip < nppb; ip++) {
            const double zeta = s_pp[ip * 8 + 0];
            const double cc = s_pp[ip * 8 + 4];
            const double pax = s_pp[ip * 8 + 5], pay = s_pp[ip * 8 + 6], paz = s_pp[ip * 8 + 7];
            const double ai2 = 2.0 * s_ab[ip * 2 + 0], aj2 = 2.0 * s_ab[ip * 2 + 1];
            const double pqx = ax_ + pax - cx, pqy = ay_ + pay - cy, pqz = az_ + paz - cz;
            const double r2 = pqx * pqx + pqy * pqy + pqz * pqz;
            for (int kp = 0; kp < a.npk; kp++) {
                const double eta = a.aux_exp[kk * a.npk + kp];
                const double ck = a.aux_coef[kk * a.npk + kp];
                const double ze = zeta + eta;
                const double rho = zeta * eta / ze;
                const double theta = (a.omega > 0) ? a.omega * a.omega / (a.omega * a.omega + rho) : 1.0;
                const double x = rho * r2 * theta;
                for (int q = s; q < 2 * NR; q += S) {
                    const double v = rys_root_or_weight<NR>(a.rys_table, x, q);
                    rw[q] = (q < NR) ? v * theta : v * sqrt(theta);
                }
                __syncthreads();
                const double fac = 2.0 * 17.493418327624862846 /* pi^2.5 */ / (zeta * eta * sqrt(ze)) * cc * ck;
                for (int un = s; un < 3 * NR; un += S) {
                    const int r = un / 3, d = un - 3 * r;
                    const double u = rw[r];
                    const double w = rw[NR + r];
                    const double ue = u * eta / ze;
                    const double uz = u * zeta / ze;
                    const double b00 = 0.5 * u / ze;
                    const double b10 = (1.0 - ue) * 0.5 / zeta;
                    const double b01 = (1.0 - uz) * 0.5 / eta;
                    const double pa = (d == 0) ? pax : (d == 1 ? pay : paz);
                    const double pq = (d == 0) ? pqx : (d == 1 ? pqy : pqz);
                    const double ab = (d == 0) ? abx : (d == 1 ? aby : abz);
                    const double c00 = pa - ue * pq;
                    const double c0p = uz * pq;
                    const double g00 = (d == 2) ? w * fac : 1.0;
                    rys_2d_unit<LI + 1, LJ + 1, LK>(g00, c00, c0p, b00, b10, b01, ab, gbuf + (r * 3 + d) * GU);
                }
                __syncthreads();
#pragma unroll
                for (int el = 0; el < NE; el++) {
                    if (s + el * S < G::NIJ) {
                        for (int r = 0; r < NR; r++) {
                            const double *gx = gbuf + r * 3 * GU;
                            const double *gy = gx + GU;
                            const double *gz = gy + GU;
                            double f[3][DK], da[3][DK], db[3][DK];
                            const int ii[3] = {ix[el], iy[el], iz[el]};
                            const int jj[3] = {jx[el], jy[el], jz[el]};
#pragma unroll
                            for (int d = 0; d < 3; d++) {
                                const double *g = (d == 0) ? gx : (d == 1 ? gy : gz);
                                const int i = ii[d], j = jj[d];
                                const double *g0 = g + (i * DJ + j) * DK;
#pragma unroll
                                for (int m = 0; m < DK; m++) {
                                    f[d][m] = g0[m];
                                    double va = -ai2 * g0[DJ * DK + m];
                                    if (i > 0) va += i * g0[m - DJ * DK];
                                    da[d][m] = va;
                                    double vb = -aj2 * g0[DK + m];
                                    if (j > 0) vb += j * g0[m - DK];
                                    db[d][m] = vb;
                                }
                            }
                            int c = 0;
#pragma unroll
                            for (int kx = LK; kx >= 0; kx--)
#pragma unroll
                                for (int ky = LK - kx; ky >= 0; ky--) {
                                    const int kz = LK - kx - ky;
                                    const double z = zc[el][c];
                                    const double fyz = z * f[1][ky] * f[2][kz];
                                    const double fxz = z * f[0][kx] * f[2][kz];
                                    const double fxy = z * f[0][kx] * f[1][ky];
                                    gA[0] += da[0][kx] * fyz; gA[1] += da[1][ky] * fxz; gA[2] += da[2][kz] * fxy;
                                    gB[0] += db[0][kx] * fyz; gB[1] += db[1][ky] * fxz; gB[2] += db[2][kz] * fxy;
                                    c++;
                                }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }

    // (grad_r i j | k) summed over the lanes of the triple; d/dA = -grad_r on function i
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
        for (int off = S / 2; off > 0; off >>= 1) {
            gA[d] += __shfl_xor(gA[d], off, 64);
            gB[d] += __shfl_xor(gB[d], off, 64);
        }
    }
    if (s == 0 && kvalid) {
        const int rep = (int)((blockIdx.x + blockIdx.y * 7u) % (unsigned)ga.nrep);
        double *g = ga.grad + (long)rep * ga.natm * 3;
        const int ia = ga.shell_atom[ish], ja = ga.shell_atom[jsh], ka = ga.aux_atom[kk];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            // da/db hold (i g(i-1) - 2 alpha g(i+1)) = the electron-coordinate derivative; nuclear derivative = minus that
            atomicAdd(g + ia * 3 + d, -gA[d]);
            atomicAdd(g + ja * 3 + d, -gB[d]);
            if (ga.aux_response) atomicAdd(g + ka * 3 + d, gA[d] + gB[d]);
        }
    }
}

template <int LI, int LJ, int LK>
struct GradCfg {
    using GG = GG3<LI, LJ, LK>;
    static constexpr int NIJ = ncart(LI) * ncart(LJ);
    static constexpr int S0 = NIJ >= 150 ? 64 : NIJ >= 100 ? 32 : NIJ >= 36 ? 16 : NIJ >= 18 ? 8 : NIJ >= 9 ? 4 : NIJ >= 6 ? 2 : 1;
    static constexpr int ES = GG::ESTRIDE;
    static constexpr int NTHREADS = ((256 / S0) * ES * 8 <= 60 * 1024) ? 256 : (((128 / S0) * ES * 8 <= 60 * 1024) ? 128 : 64);
    // more lanes per triple (fewer triples per workgroup) until the workgroup's LDS fits
    static constexpr int fit(int s) { return ((NTHREADS / s) * ES * 8 <= 120 * 1024 || s >= NTHREADS) ? s : fit(2 * s); }
    static constexpr int S = fit(S0 > NTHREADS ? NTHREADS : S0);
};

template <int LI, int LJ, int LK>
int launch_grad_class(const Int3c2eGradArgs &ga, hipStream_t st)
{
    using C = GradCfg<LI, LJ, LK>;
    constexpr int NT = C::NTHREADS / C::S;
    constexpr size_t lds = (size_t)(64 * 10 + NT * C::ES) * sizeof(double);
    static_assert(lds <= 160 * 1024, "LDS budget exceeded");
    if (ga.base.npairs == 0 || ga.base.naux_cls == 0) return 0;
    auto kern = int3c2e_grad_kernel<LI, LJ, LK, C::S, C::NTHREADS>;
    static bool attr_done[64] = {false};       // per device: the attribute belongs to the device's code object (multi-GPU handles)
    int dev_id = 0;
    if (lds > 64 * 1024) PAMD_CHECK_HIP(hipGetDevice(&dev_id));
    bool &attr_set = attr_done[dev_id & 63];
    if (!attr_set && lds > 64 * 1024) {
        PAMD_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(ga.base.npairs, ceil_div(ga.base.naux_cls, NT));
    hipLaunchKernelGGL(kern, grid, dim3(C::NTHREADS), lds, st, ga);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // namespace pamd
