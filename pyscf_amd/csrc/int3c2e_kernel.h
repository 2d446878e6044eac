// 3-centre 2-electron Coulomb integrals (ij|k) on gfx950, one kernel per angular-momentum
// class (LI >= LJ, LK), Rys quadrature.
//
// What it replaces: the libcint call `int3c2e_sph(buf, NULL, shls, atm, natm, bas, nbas, env,
// cintopt, cache)` made per shell triple by GTOnr3c_fill_s2ij (pyscf/lib/gto/fill_nr_3c.c:127-185)
// under the OpenMP job loop of GTOnr3c_drv (:196-225), plus the scatter into the packed-tril
// layout (dcopy_s2_igtj/ieqj :74-117).  The arithmetic itself (libcint v6.1.3) is not in the
// reference tree; this is a from-scratch Rys-quadrature implementation:
//   (ab|c) = 2 pi^{5/2} / (zeta eta sqrt(zeta+eta)) K_ab  sum_r  Ix(r) Iy(r) Iz(r)
//   2-D integrals by the standard VRR on centre A / C and HRR A->B.
//
// Work decomposition (wave64-first):
//   workgroup  = one contracted shell pair (i,j)  x  NT consecutive aux shells of class LK
//   S lanes cooperate on one shell triple (S | 64): the 2*NR root/weight evaluations, the
//   3*NR two-dimensional recurrences and the ncart_i*ncart_j Cartesian output columns are each
//   dealt round-robin to the S lanes; 2-D integrals are exchanged through LDS.
//   Shell-pair primitive data (zeta, P, K_ab c_i c_j, P-A) is read once per workgroup from a
//   coalesced SoA in HBM into LDS.
// Output: T[row(p,q)][Q]  (row = packed-tril index of the AO pair minus the slab offset,
//   Q = aux function index) - aux index fastest, so each lane group writes runs of
//   consecutive aux functions; the L^-1 GEMM (cderi_solve) re-lays it out as cderi[L][pq].
#pragma once
#include "common.h"
#include "rys_device.h"
#include "int3c2e_args.h"

namespace pamd {


__host__ __device__ constexpr int ncart(int l) { return (l + 1) * (l + 2) / 2; }

// cartesian exponents of component c of shell l (lx descending, then ly descending)
__device__ __forceinline__ void cart_exps(int l, int c, int &lx, int &ly, int &lz)
{
    int x = l;
    int rem = c;
    while (rem > l - x) { rem -= (l - x + 1); x--; }
    lx = x; ly = (l - x) - rem; lz = rem;
}

template <int LI, int LJ, int LK>
struct G3 {
    static constexpr int NR = (LI + LJ + LK) / 2 + 1;
    static constexpr int DI = LI + 1, DJ = LJ + 1, DK = LK + 1;
    static constexpr int GU = DI * DJ * DK;          // one unit (root, direction)
    static constexpr int GT = 3 * NR * GU;           // all units of one triple
    static constexpr int NCI = ncart(LI), NCJ = ncart(LJ), NCK = ncart(LK);
    static constexpr int NIJ = NCI * NCJ;
    static constexpr int NSI = 2 * LI + 1, NSJ = 2 * LJ + 1, NSK = 2 * LK + 1;
};

// One 2-D recurrence: VRR in registers on (n = 0..LI+LJ, m = 0..LK), then HRR A->B, result
// g[i][j][k] (k fastest) written to LDS.
template <int LI, int LJ, int LK>
__device__ __forceinline__ void rys_2d_unit(double g00, double c00, double c0p, double b00, double b10,
                                            double b01, double ab, double *__restrict__ gout)
{
    constexpr int N = LI + LJ;
    double g[N + 1][LK + 1];
    g[0][0] = g00;
    if constexpr (N > 0) {
        g[1][0] = c00 * g00;
#pragma unroll
        for (int n = 1; n < N; n++) g[n + 1][0] = c00 * g[n][0] + n * b10 * g[n - 1][0];
    }
    if constexpr (LK > 0) {
#pragma unroll
        for (int n = 0; n <= N; n++) {
            double v = c0p * g[n][0];
            if (n > 0) v += n * b00 * g[n - 1][0];
            g[n][1] = v;
        }
#pragma unroll
        for (int m = 1; m < LK; m++) {
#pragma unroll
            for (int n = 0; n <= N; n++) {
                double v = c0p * g[n][m] + m * b01 * g[n][m - 1];
                if (n > 0) v += n * b00 * g[n - 1][m];
                g[n][m + 1] = v;
            }
        }
    }
    // HRR: g(i, j, k) = g(i+1, j-1, k) + ab * g(i, j-1, k); store i <= LI for every j
#pragma unroll
    for (int m = 0; m <= LK; m++) {
        double col[N + 1];
#pragma unroll
        for (int n = 0; n <= N; n++) col[n] = g[n][m];
#pragma unroll
        for (int j = 0; j <= LJ; j++) {
#pragma unroll
            for (int i = 0; i <= LI; i++) gout[(i * (LJ + 1) + j) * (LK + 1) + m] = col[i];
            if (j < LJ) {
#pragma unroll
                for (int n = 0; n < N - j; n++) col[n] = col[n + 1] + ab * col[n];
            }
        }
    }
}

template <int LI, int LJ, int LK, int S, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void int3c2e_kernel(Int3c2eArgs a)
{
    using G = G3<LI, LJ, LK>;
    constexpr int NR = G::NR;
    constexpr int NT = NTHREADS / S;                       // shell triples per workgroup
    constexpr int TSTRIDE = (G::GT + 2 * NR) | 1;          // odd stride: triples on distinct banks
    constexpr int NE = (G::NIJ + S - 1) / S;               // (ci,cj) columns per lane
    // staging sizes for the cart->sph epilogue (aliases the g area)
    constexpr int STAGE1 = G::NIJ * G::NSK;                // [ci][cj][mk]
    constexpr int STAGE2 = G::NCI * G::NSJ * G::NSK;       // [ci][mj][mk]
    constexpr int ESTRIDE = ((STAGE1 + STAGE2) > TSTRIDE ? (STAGE1 + STAGE2) : TSTRIDE) | 1;
    constexpr int MAXPP = 64;                              // primitive pairs staged per pass

    extern __shared__ double smem[];
    double *s_pp = smem;                                   // [MAXPP][8]
    double *s_t = smem + MAXPP * 8;                        // [NT][ESTRIDE]

    const int tid = threadIdx.x;
    const int t = tid / S;                                 // triple slot in the workgroup
    const int s = tid - t * S;                             // lane within the triple group
    const int ipair = blockIdx.x;
    const int kidx = blockIdx.y * NT + t;
    const bool kvalid = kidx < a.naux_cls;
    const int kk = kvalid ? kidx : a.naux_cls - 1;

    const int ish = a.pair_ish[ipair], jsh = a.pair_jsh[ipair];
    const int pp0 = a.pair_pp0[ipair], npp = a.pair_npp[ipair];
    const double abx = a.shell_xyz[ish * 3 + 0] - a.shell_xyz[jsh * 3 + 0];
    const double aby = a.shell_xyz[ish * 3 + 1] - a.shell_xyz[jsh * 3 + 1];
    const double abz = a.shell_xyz[ish * 3 + 2] - a.shell_xyz[jsh * 3 + 2];
    const double cx = a.aux_xyz[kk * 3 + 0], cy = a.aux_xyz[kk * 3 + 1], cz = a.aux_xyz[kk * 3 + 2];

    double *my = s_t + t * ESTRIDE;
    double *rw = my;                                       // [2*NR]
    double *gbuf = my + 2 * NR;                            // [3*NR][GU]

    // cartesian columns of this lane
    int ex[NE], ey[NE], ez[NE];                            // LDS offsets of gx/gy/gz for column e
#pragma unroll
    for (int el = 0; el < NE; el++) {
        int e = s + el * S;
        if (e >= G::NIJ) e = 0;
        int ci = e / G::NCJ, cj = e - ci * G::NCJ;
        int ix, iy, iz, jx, jy, jz;
        cart_exps(LI, ci, ix, iy, iz);
        cart_exps(LJ, cj, jx, jy, jz);
        ex[el] = (ix * G::DJ + jx) * G::DK;
        ey[el] = (iy * G::DJ + jy) * G::DK + G::GU;
        ez[el] = (iz * G::DJ + jz) * G::DK + 2 * G::GU;
    }
    double acc[NE][G::NCK];
#pragma unroll
    for (int el = 0; el < NE; el++)
#pragma unroll
        for (int c = 0; c < G::NCK; c++) acc[el][c] = 0.0;

    for (int ppb = 0; ppb < npp; ppb += MAXPP) {
        const int nppb = (npp - ppb < MAXPP) ? npp - ppb : MAXPP;
        __syncthreads();
        for (int e = tid; e < nppb * 8; e += NTHREADS) s_pp[e] = a.pp[(long)(pp0 + ppb) * 8 + e];
        __syncthreads();
        for (int ip = 0; ip < nppb; ip++) {
            const double zeta = s_pp[ip * 8 + 0];
            const double px = s_pp[ip * 8 + 1], py = s_pp[ip * 8 + 2], pz = s_pp[ip * 8 + 3];
            const double cc = s_pp[ip * 8 + 4];
            const double pax = s_pp[ip * 8 + 5], pay = s_pp[ip * 8 + 6], paz = s_pp[ip * 8 + 7];
            const double pqx = px - cx, pqy = py - cy, pqz = pz - cz;
            const double r2 = pqx * pqx + pqy * pqy + pqz * pqz;
            for (int kp = 0; kp < a.npk; kp++) {
                const double eta = a.aux_exp[kk * a.npk + kp];
                const double ck = a.aux_coef[kk * a.npk + kp];
                const double ze = zeta + eta;
                const double rho = zeta * eta / ze;
                // long-range attenuation erf(omega r)/r: the t-integral stops at t^2 = theta = omega^2/(omega^2+rho),
                // i.e. x -> theta x, roots -> theta u, weights -> sqrt(theta) w
                const double theta = (a.omega > 0) ? a.omega * a.omega / (a.omega * a.omega + rho) : 1.0;
                const double x = rho * r2 * theta;
                // ---- phase 1: roots and weights, dealt over the S lanes
                for (int q = s; q < 2 * NR; q += S) {
                    const double v = rys_root_or_weight<NR>(a.rys_table, x, q);
                    rw[q] = (q < NR) ? v * theta : v * sqrt(theta);
                }
                __syncthreads();
                // ---- phase 2: 2-D integrals, unit = (root r, direction d)
                const double fac = 2.0 * 17.493418327624862846 /* pi^2.5 */ / (zeta * eta * sqrt(ze)) * cc * ck;
                for (int un = s; un < 3 * NR; un += S) {
                    const int r = un / 3, d = un - 3 * r;
                    const double u = rw[r];
                    const double w = rw[NR + r];
                    const double ue = u * eta / ze;            // t^2 eta/(zeta+eta)
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
                    rys_2d_unit<LI, LJ, LK>(g00, c00, c0p, b00, b10, b01, ab, gbuf + (r * 3 + d) * G::GU);
                }
                __syncthreads();
                // ---- phase 3: contract the roots into the Cartesian columns of this lane
#pragma unroll
                for (int el = 0; el < NE; el++) {
                    if (s + el * S < G::NIJ) {
                        for (int r = 0; r < NR; r++) {
                            const double *gx = gbuf + r * 3 * G::GU + ex[el];
                            const double *gy = gbuf + r * 3 * G::GU + ey[el];
                            const double *gz = gbuf + r * 3 * G::GU + ez[el];
                            double vx[G::DK], vy[G::DK], vz[G::DK];
#pragma unroll
                            for (int m = 0; m < G::DK; m++) { vx[m] = gx[m]; vy[m] = gy[m]; vz[m] = gz[m]; }
                            int c = 0;
#pragma unroll
                            for (int kx = LK; kx >= 0; kx--)
#pragma unroll
                                for (int ky = LK - kx; ky >= 0; ky--) {
                                    const int kz = LK - kx - ky;
                                    acc[el][c] += vx[kx] * vy[ky] * vz[kz];
                                    c++;
                                }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }

    // ================= epilogue: cart -> real spherical for k (registers), j and i (LDS) ========
    const double *c2s_i = a.c2s + a.c2s_off[LI];
    const double *c2s_j = a.c2s + a.c2s_off[LJ];
    const double *c2s_k = a.c2s + a.c2s_off[LK];
    double *st1 = my;                  // [ci][cj][mk]
    double *st2 = my + STAGE1;         // [ci][mj][mk]
#pragma unroll
    for (int el = 0; el < NE; el++) {
        int e = s + el * S;
        if (e < G::NIJ) {
            for (int mk = 0; mk < G::NSK; mk++) {
                double v = 0;
#pragma unroll
                for (int c = 0; c < G::NCK; c++) v += c2s_k[mk * G::NCK + c] * acc[el][c];
                st1[e * G::NSK + mk] = v;
            }
        }
    }
    __syncthreads();
    // j transform: st2[ci][mj][mk] = sum_cj c2s_j[mj][cj] st1[ci][cj][mk]
    for (int o = s; o < G::NCI * G::NSJ * G::NSK; o += S) {
        int mk = o % G::NSK;
        int mj = (o / G::NSK) % G::NSJ;
        int ci = o / (G::NSK * G::NSJ);
        double v = 0;
        for (int cj = 0; cj < G::NCJ; cj++) v += c2s_j[mj * G::NCJ + cj] * st1[(ci * G::NCJ + cj) * G::NSK + mk];
        st2[o] = v;
    }
    __syncthreads();
    // i transform + store
    if (kvalid) {
        const int p0 = a.shell_ao0[ish], q0 = a.shell_ao0[jsh];
        const int f0 = a.aux_f0[kk];
        for (int o = s; o < G::NSI * G::NSJ * G::NSK; o += S) {
            int mk = o % G::NSK;
            int mj = (o / G::NSK) % G::NSJ;
            int mi = o / (G::NSK * G::NSJ);
            double v = 0;
            for (int ci = 0; ci < G::NCI; ci++) v += c2s_i[mi * G::NCI + ci] * st2[(ci * G::NSJ + mj) * G::NSK + mk];
            long p = p0 + mi, q = q0 + mj;
            long row;
            if (a.tril) {
                if (ish == jsh && q > p) continue;
                row = (p >= q) ? p * (p + 1) / 2 + q : q * (q + 1) / 2 + p;
            } else {
                row = p;
            }
            a.T[(row - a.row_offset) * a.ldT + f0 + mk] = v;
        }
    }
}

template <int LI, int LJ, int LK, int S, int NTHREADS>
int launch_int3c2e(const Int3c2eArgs &a, hipStream_t st)
{
    using G = G3<LI, LJ, LK>;
    constexpr int NR = G::NR;
    constexpr int NT = NTHREADS / S;
    constexpr int TSTRIDE = (G::GT + 2 * NR) | 1;
    constexpr int STAGE = G::NIJ * G::NSK + G::NCI * G::NSJ * G::NSK;
    constexpr int ESTRIDE = (STAGE > TSTRIDE ? STAGE : TSTRIDE) | 1;
    constexpr size_t lds = (size_t)(64 * 8 + NT * ESTRIDE) * sizeof(double);
    static_assert(lds <= 160 * 1024, "LDS budget exceeded");
    if (a.npairs == 0 || a.naux_cls == 0) return 0;
    auto kern = int3c2e_kernel<LI, LJ, LK, S, NTHREADS>;
    static bool attr_done[64] = {false};       // per device: the attribute belongs to the device's code object (multi-GPU handles)
    int dev_id = 0;
    if (lds > 64 * 1024) PAMD_CHECK_HIP(hipGetDevice(&dev_id));
    bool &attr_set = attr_done[dev_id & 63];
    if (!attr_set && lds > 64 * 1024) {
        PAMD_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid(a.npairs, ceil_div(a.naux_cls, NT));
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, a);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// lanes per triple and workgroup size per class
template <int LI, int LJ, int LK>
struct ClassCfg {
    static constexpr int NIJ = ncart(LI) * ncart(LJ);
    static constexpr int S = NIJ >= 150 ? 64 : NIJ >= 100 ? 32 : NIJ >= 36 ? 16 : NIJ >= 18 ? 8 : NIJ >= 9 ? 4 : NIJ >= 6 ? 2 : 1;
    using G = G3<LI, LJ, LK>;
    static constexpr int TS = (G::GT + 2 * G::NR) | 1;
    static constexpr int ST = G::NIJ * G::NSK + G::NCI * G::NSJ * G::NSK;
    static constexpr int ES = (ST > TS ? ST : TS) | 1;
    // keep the workgroup's LDS under ~64 KB
    static constexpr int NTHREADS = ((256 / S) * ES * 8 <= 60 * 1024) ? 256 : (((128 / S) * ES * 8 <= 60 * 1024) ? 128 : 64);
};

template <int LI, int LJ, int LK>
int launch_class(const Int3c2eArgs &a, hipStream_t st)
{
    using C = ClassCfg<LI, LJ, LK>;
    constexpr int S = (C::S > C::NTHREADS) ? C::NTHREADS : C::S;
    return launch_int3c2e<LI, LJ, LK, S, C::NTHREADS>(a, st);
}

}  // namespace pamd
