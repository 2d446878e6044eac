// DFT grid kernels: Becke partition weights and AO (+ gradient) values on grid points.
//   PAMD_becke_partition <- VXCgen_grid  (pyscf/lib/dft/grid_basis.c:32-101)
//   PAMD_grid_partition  <- get_partition's three cell functions (pyscf/dft/gen_grid.py:341-419)
//   PAMD_eval_ao         <- GTOval_sph_deriv0 / GTOval_sph_deriv1 (pyscf/gto/eval_gto.py:31-144;
//                           pyscf/lib/gto/grid_ao_drv.c:222-284 GTOeval_sph_iter, :125-141 GTOnabla1,
//                           pyscf/lib/gto/deriv1.c:129-520)
// Output layout of PAMD_eval_ao: ao[comp][grid][ldao] (AO index fastest = numint.eval_ao's
// (comp, ngrids, nao) view, numint.py:51-114), comp = 1 (value) or 4 (value, d/dx, d/dy, d/dz).
#include "common.h"

using namespace pamd;

namespace {

// Saturated interatomic distance of the Laqua-Kussmann-Ochsenfeld partition, R_c (1 - exp(-sum_{m<=12} (R/R_c)^m / m)),
// R_c = 5 Bohr (grid_basis.c:236-247).
__device__ inline double lko_saturate(double r)
{
    const double x = r / 5.0;
    double xm = 1, tot = 0;
    for (int m = 1; m <= 12; m++) {
        xm *= x;
        tot += xm / m;
    }
    return 5.0 * (1 - exp(-tot));
}

// out[ia][g] = prod_{j != ia} s(mu_ij)-type cell function of atom ia (unnormalised)
// SCHEME 0: Becke's thrice-iterated polynomial (VXCgen_grid, grid_basis.c:32-101); 1: the Stratmann-Scuseria-Frisch
// piecewise septic with a = 0.64 (gen_grid.py:203-212 through the generic branch of get_partition :388-404);
// 2: Becke's polynomial on mu clamped to [-1, 1] with the saturated distance (VXCgen_grid_lko, grid_basis.c:266-384).
template <int SCHEME>
__global__ __launch_bounds__(256) void becke_kernel(double *__restrict__ out, const double *__restrict__ coords,
                                                    const double *__restrict__ atm, const double *__restrict__ radii,
                                                    int natm, long ngrids)
{
    long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngrids) return;
    const double x = coords[g * 3 + 0], y = coords[g * 3 + 1], z = coords[g * 3 + 2];
    for (int i = 0; i < natm; i++) out[(long)i * ngrids + g] = 1.0;
    for (int i = 0; i < natm; i++) {
        const double dxi = x - atm[i * 3], dyi = y - atm[i * 3 + 1], dzi = z - atm[i * 3 + 2];
        const double di = sqrt(dxi * dxi + dyi * dyi + dzi * dzi);
        double pi = out[(long)i * ngrids + g];
        for (int j = 0; j < i; j++) {
            const double dxj = x - atm[j * 3], dyj = y - atm[j * 3 + 1], dzj = z - atm[j * 3 + 2];
            const double dj = sqrt(dxj * dxj + dyj * dyj + dzj * dzj);
            const double ax = atm[i * 3] - atm[j * 3], ay = atm[i * 3 + 1] - atm[j * 3 + 1], az = atm[i * 3 + 2] - atm[j * 3 + 2];
            const double rij = sqrt(ax * ax + ay * ay + az * az);
            double s = (di - dj) / (SCHEME == 2 ? lko_saturate(rij) : rij);
            if (SCHEME == 2) s = fmin(1.0, fmax(-1.0, s));
            if (radii) s += radii[i * natm + j] * (1 - s * s);
            if (SCHEME == 1) {
                const double ma = s / .64, ma2 = ma * ma;
                const double g1 = (1 / 16.) * (ma * (35 + ma2 * (-35 + ma2 * (21 - 5 * ma2))));
                s = .5 * (s <= -.64 ? -1.0 : (s >= .64 ? 1.0 : g1));
            } else {
                s = (3 - s * s) * s * .5;
                s = (3 - s * s) * s * .5;
                s = ((3 - s * s) * s * .5) * .5;
            }
            pi *= .5 - s;
            out[(long)j * ngrids + g] *= .5 + s;
        }
        out[(long)i * ngrids + g] = pi;
    }
}

// Grid response of the Becke weights contracted with a per-point scalar e[g] (the XC energy density per volume):
//   out[C][x] += sum_g e[g] d w_g / d R_C,x        (pyscf/grad/rks.py get_vxc_full_response: excsum += exc rho weight1)
// the point moving rigidly with its owner atom.  For C != owner (point fixed)
//   d w/dR_C = w [ d ln P_own/dR_C - sum_B (P_B/Z) d ln P_B/dR_C ],
//   d ln P_B/dR_C = [B != C] g_BC dmu_BC/dR_C + [B == C] sum_{D != C} g_CD dmu_CD/dR_C,
//   g_BD = -p3'(nu_BD) (1 - 2 a_BD mu_BD) / (2 f_BD),  f_BD = (1 - p3(nu_BD))/2,  nu = mu + a (1 - mu^2),
// and the owner's derivative is minus the sum over the other atoms (translational invariance).
// One thread per grid point, O(natm^2) pair terms recomputed on the fly; P_B/Z from the partition kernel's output.
// With SCHEME 1 the cell polynomial p3 is the Stratmann septic h (h' = 35/(16 a) (1 - (nu/a)^2)^3 inside |nu| < a);
// with SCHEME 2 mu is scaled by the saturated distance S(R_BD) instead of R_BD, so the atom derivatives of mu become
// -+ u/S -+ mu S'(R)/S n/R, and mu is clamped to [-1, 1] (no derivative outside) - VXCgen_grid_lko_deriv,
// grid_basis.c:386-560.
template <int SCHEME>
__device__ inline void becke_pair(double dB, double dD, double sBD, double a, double &mu, double &gfac)
{
    mu = (dB - dD) / sBD;
    bool inside = true;
    if (SCHEME == 2) {
        inside = fabs(mu) < 1.0;
        mu = fmin(1.0, fmax(-1.0, mu));
    }
    const double nu = mu + a * (1 - mu * mu);
    double h, dh;
    if (SCHEME == 1) {
        const double ma = nu / .64, ma2 = ma * ma;
        const bool out = fabs(nu) >= .64;
        h = out ? (nu > 0 ? 1.0 : -1.0) : (1 / 16.) * (ma * (35 + ma2 * (-35 + ma2 * (21 - 5 * ma2))));
        dh = out ? 0.0 : (35 / 16.) * (1 - ma2) * (1 - ma2) * (1 - ma2) / .64;
    } else {
        const double p1 = (3 - nu * nu) * nu * .5;
        const double p2 = (3 - p1 * p1) * p1 * .5;
        h = (3 - p2 * p2) * p2 * .5;
        dh = 3.375 * (1 - p2 * p2) * (1 - p1 * p1) * (1 - nu * nu);
    }
    const double f = .5 * (1 - h);
    gfac = inside ? -.5 * dh * (1 - 2 * a * mu) / (f + 1e-200) : 0.0;
}

__device__ inline double lko_saturate_deriv(double r)
{
    const double x = r / 5.0;
    double xm = 1, tot = 0, dtot = 0;
    for (int m = 1; m <= 12; m++) {
        dtot += xm;
        xm *= x;
        tot += xm / m;
    }
    return exp(-tot) * dtot;
}

template <int SCHEME>
__global__ __launch_bounds__(256) void becke_response_kernel(
    const double *__restrict__ coords, const int *__restrict__ owner, const double *__restrict__ weights,
    const double *__restrict__ e, const double *__restrict__ pb, const double *__restrict__ atm,
    const double *__restrict__ radii, int natm, long ng, double *__restrict__ out)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = g < ng;
    const long gg = valid ? g : 0;
    const double x = coords[gg * 3], y = coords[gg * 3 + 1], z = coords[gg * 3 + 2];
    const int own = owner[gg];
    const double we = valid ? weights[gg] * e[gg] : 0.0;
    double zsum = 0;
    for (int b = 0; b < natm; b++) zsum += pb[(long)b * ng + gg];
    const double zinv = 1.0 / (zsum + 1e-300);
    double acc_own[3] = {0, 0, 0};
    __shared__ double red[4][3];
    for (int c = 0; c < natm; c++) {
        const double cx = atm[c * 3], cy = atm[c * 3 + 1], cz = atm[c * 3 + 2];
        const double ucx = x - cx, ucy = y - cy, ucz = z - cz;
        const double dC = sqrt(ucx * ucx + ucy * ucy + ucz * ucz) + 1e-200;
        const double uhx = ucx / dC, uhy = ucy / dC, uhz = ucz / dC;
        double ownt[3] = {0, 0, 0}, avg[3] = {0, 0, 0}, self[3] = {0, 0, 0};
        for (int b = 0; b < natm; b++) {
            if (b == c) continue;
            const double bx = atm[b * 3], by = atm[b * 3 + 1], bz = atm[b * 3 + 2];
            const double dB = sqrt((x - bx) * (x - bx) + (y - by) * (y - by) + (z - bz) * (z - bz));
            const double nx = bx - cx, ny = by - cy, nz = bz - cz;             // R_B - R_C
            const double rbc = sqrt(nx * nx + ny * ny + nz * nz);
            const double sbc = (SCHEME == 2) ? lko_saturate(rbc) : rbc;        // S(R)
            const double rinv = 1.0 / sbc;                                     // 1 / S
            const double kinv = ((SCHEME == 2) ? lko_saturate_deriv(rbc) : 1.0) / (sbc * rbc);   // S'(R) / (S R)
            // B's cell function seen from C: d mu_BC / dR_C = u_C / S + mu_BC S'/S n_BC / R
            double mu, gf;
            becke_pair<SCHEME>(dB, dC, sbc, radii ? radii[b * natm + c] : 0.0, mu, gf);
            const double tbx = gf * (uhx * rinv + mu * nx * kinv);
            const double tby = gf * (uhy * rinv + mu * ny * kinv);
            const double tbz = gf * (uhz * rinv + mu * nz * kinv);
            const double wb = pb[(long)b * ng + gg] * zinv;
            avg[0] += wb * tbx; avg[1] += wb * tby; avg[2] += wb * tbz;
            if (b == own) { ownt[0] = tbx; ownt[1] = tby; ownt[2] = tbz; }
            // C's own cell function: d mu_CB / dR_C = -u_C / R - mu_CB n_CB / R^2, n_CB = R_C - R_B = -n
            becke_pair<SCHEME>(dC, dB, sbc, radii ? radii[c * natm + b] : 0.0, mu, gf);
            self[0] += gf * (-uhx * rinv + mu * nx * kinv);
            self[1] += gf * (-uhy * rinv + mu * ny * kinv);
            self[2] += gf * (-uhz * rinv + mu * nz * kinv);
        }
        const double wc = pb[(long)c * ng + gg] * zinv;
        double v[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            v[k] = (c == own) ? 0.0 : we * (ownt[k] - avg[k] - wc * self[k]);
            acc_own[k] -= v[k];
        }
        // block reduction of the contribution to atom c
#pragma unroll
        for (int k = 0; k < 3; k++)
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = v[0]; red[threadIdx.x >> 6][1] = v[1]; red[threadIdx.x >> 6][2] = v[2]; }
        __syncthreads();
        if (threadIdx.x < 3)
            atomicAdd(out + c * 3 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
    // owner: minus the sum over the other atoms (different owners within the block: per-thread atomics)
    if (valid && we != 0.0)
#pragma unroll
        for (int k = 0; k < 3; k++) atomicAdd(out + own * 3 + k, acc_own[k]);
}

constexpr int AO_LMAX = 4;
constexpr int AO_NC = (AO_LMAX + 1) * (AO_LMAX + 2) / 2;

struct AOShells {
    const int *l, *ao0, *prim0, *nprim;
    const double *xyz, *exps, *coefs;
};

// value and gradient of  R(r) * sum_c m[c] x^lx y^ly z^lz  with compile-time L (all monomial
// indices become register indices after unrolling)
template <int L, int DERIV>
__device__ __forceinline__ void ao_angular(double x, double y, double z, double rad, double rad1,
                                           const double *__restrict__ m, double &v, double &vx, double &vy, double &vz)
{
    double px[L + 2], py[L + 2], pz[L + 2];
    px[0] = py[0] = pz[0] = 1;
#pragma unroll
    for (int i = 1; i <= L + 1; i++) { px[i] = px[i - 1] * x; py[i] = py[i - 1] * y; pz[i] = pz[i - 1] * z; }
    int c = 0;
#pragma unroll
    for (int lx = L; lx >= 0; lx--)
#pragma unroll
        for (int ly = L - lx; ly >= 0; ly--, c++) {
            const int lz = L - lx - ly;
            const double f = m[c];
            v += f * px[lx] * py[ly] * pz[lz] * rad;
            if (DERIV) {
                vx += f * (rad1 * px[lx + 1] * py[ly] * pz[lz] + (lx ? lx * px[lx ? lx - 1 : 0] * py[ly] * pz[lz] * rad : 0.0));
                vy += f * (rad1 * px[lx] * py[ly + 1] * pz[lz] + (ly ? ly * px[lx] * py[ly ? ly - 1 : 0] * pz[lz] * rad : 0.0));
                vz += f * (rad1 * px[lx] * py[ly] * pz[lz + 1] + (lz ? lz * px[lx] * py[ly] * pz[lz ? lz - 1 : 0] * rad : 0.0));
            }
        }
}

// value, gradient and Hessian (order: 1, x, y, z, xx, xy, xz, yy, yz, zz = numint.eval_ao deriv=2,
// pyscf/dft/numint.py:51-114) of R(r^2) * sum_c m[c] x^a y^b z^c with dR/dx = x R1, d2R/dxdy = x y R2 (+ R1 on the diagonal)
template <int L>
__device__ __forceinline__ void ao_angular2(double x, double y, double z, double r0, double r1, double r2,
                                            const double *__restrict__ m, double *__restrict__ o)
{
    double px[L + 1], py[L + 1], pz[L + 1];
    px[0] = py[0] = pz[0] = 1;
#pragma unroll
    for (int i = 1; i <= L; i++) { px[i] = px[i - 1] * x; py[i] = py[i - 1] * y; pz[i] = pz[i - 1] * z; }
#pragma unroll
    for (int k = 0; k < 10; k++) o[k] = 0;
    int c = 0;
#pragma unroll
    for (int a = L; a >= 0; a--)
#pragma unroll
        for (int b = L - a; b >= 0; b--, c++) {
            const int cc = L - a - b;
            const double f = m[c];
            const double X0 = px[a], Y0 = py[b], Z0 = pz[cc];
            const double X1 = a ? a * px[a ? a - 1 : 0] : 0.0, Y1 = b ? b * py[b ? b - 1 : 0] : 0.0, Z1 = cc ? cc * pz[cc ? cc - 1 : 0] : 0.0;
            const double X2 = a > 1 ? a * (a - 1) * px[a > 1 ? a - 2 : 0] : 0.0;
            const double Y2 = b > 1 ? b * (b - 1) * py[b > 1 ? b - 2 : 0] : 0.0;
            const double Z2 = cc > 1 ? cc * (cc - 1) * pz[cc > 1 ? cc - 2 : 0] : 0.0;
            const double P = X0 * Y0 * Z0, Px = X1 * Y0 * Z0, Py = X0 * Y1 * Z0, Pz = X0 * Y0 * Z1;
            o[0] += f * P * r0;
            o[1] += f * (Px * r0 + P * x * r1);
            o[2] += f * (Py * r0 + P * y * r1);
            o[3] += f * (Pz * r0 + P * z * r1);
            o[4] += f * (X2 * Y0 * Z0 * r0 + 2 * Px * x * r1 + P * (r1 + x * x * r2));
            o[5] += f * (X1 * Y1 * Z0 * r0 + (Px * y + Py * x) * r1 + P * x * y * r2);
            o[6] += f * (X1 * Y0 * Z1 * r0 + (Px * z + Pz * x) * r1 + P * x * z * r2);
            o[7] += f * (X0 * Y2 * Z0 * r0 + 2 * Py * y * r1 + P * (r1 + y * y * r2));
            o[8] += f * (X0 * Y1 * Z1 * r0 + (Py * z + Pz * y) * r1 + P * y * z * r2);
            o[9] += f * (X0 * Y0 * Z2 * r0 + 2 * Pz * z * r1 + P * (r1 + z * z * r2));
        }
}

// Grid-major output ao[comp][g][ldao] (AO index fastest): one wave per grid point.  Phase 1: lanes
// over shells evaluate the contracted radial sums (one exp per primitive, not per function) into
// LDS; phase 2: lanes over AO functions combine them with the angular polynomials, so stores are
// fully coalesced and the (ngrids x nao) blocks feed the FP64 MFMA GEMMs as [k = grid][m = AO] panels.
constexpr int AO_SC = 256;            // shells per chunk (LDS: 4 waves x 256 x 2 doubles = 16 KB)

template <int DERIV>
__global__ __launch_bounds__(256) void eval_ao_kernel(AOShells sh, int nsh, const int *__restrict__ fn2sh,
                                                      const double *__restrict__ coords, long g0, long ng,
                                                      const double *__restrict__ c2s, const int *__restrict__ c2s_off,
                                                      double *__restrict__ ao, long ldg_rows, int ldao, int nao,
                                                      double thr, unsigned char *__restrict__ flags)
{
    __shared__ double s_rad[4][AO_SC][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gl = (long)blockIdx.x * 4 + wave;
    const bool valid = gl < ng;
    const long g = g0 + (valid ? gl : 0);
    const double gx = coords[g * 3 + 0], gy = coords[g * 3 + 1], gz = coords[g * 3 + 2];
    const long comp_stride = ldg_rows * ldao;
    for (int s0 = 0; s0 < nsh; s0 += AO_SC) {
        const int s1 = (s0 + AO_SC < nsh) ? s0 + AO_SC : nsh;
        __syncthreads();
        for (int s = s0 + lane; s < s1; s += 64) {
            const double x = gx - sh.xyz[s * 3], y = gy - sh.xyz[s * 3 + 1], z = gz - sh.xyz[s * 3 + 2];
            const double r2 = x * x + y * y + z * z;
            double rad = 0, rad1 = 0, rad2 = 0;
            for (int p = 0; p < sh.nprim[s]; p++) {
                const double a = sh.exps[sh.prim0[s] + p];
                const double e = sh.coefs[sh.prim0[s] + p] * exp(-a * r2);
                rad += e;
                rad1 += -2 * a * e;
                if (DERIV == 2) rad2 += 4 * a * a * e;
            }
            s_rad[wave][s - s0][0] = rad;
            s_rad[wave][s - s0][1] = rad1;
            s_rad[wave][s - s0][2] = rad2;
        }
        __syncthreads();
        const int mu0 = sh.ao0[s0];
        const int mu1 = (s1 < nsh) ? sh.ao0[s1] : ldao;     // last chunk also zero-fills the padding
        for (int mu = mu0 + lane; mu < mu1; mu += 64) {
            if (DERIV == 2) {
                double o[10];
#pragma unroll
                for (int k = 0; k < 10; k++) o[k] = 0;
                if (mu < nao) {
                    const int s = fn2sh[mu];
                    const int l = sh.l[s];
                    const int k = mu - sh.ao0[s];
                    const double x = gx - sh.xyz[s * 3], y = gy - sh.xyz[s * 3 + 1], z = gz - sh.xyz[s * 3 + 2];
                    const double r0 = s_rad[wave][s - s0][0], r1 = s_rad[wave][s - s0][1], r2 = s_rad[wave][s - s0][2];
                    const double *m = c2s + c2s_off[l] + k * ((l + 1) * (l + 2) / 2);
                    switch (l) {
                    case 0: ao_angular2<0>(x, y, z, r0, r1, r2, m, o); break;
                    case 1: ao_angular2<1>(x, y, z, r0, r1, r2, m, o); break;
                    case 2: ao_angular2<2>(x, y, z, r0, r1, r2, m, o); break;
                    case 3: ao_angular2<3>(x, y, z, r0, r1, r2, m, o); break;
                    default: ao_angular2<4>(x, y, z, r0, r1, r2, m, o); break;
                    }
                }
                if (valid) {
                    double *p = ao + gl * ldao + mu;
#pragma unroll
                    for (int k = 0; k < 10; k++) p[k * comp_stride] = o[k];
                }
                continue;
            }
            double v = 0, vx = 0, vy = 0, vz = 0;
            if (mu < nao) {
                const int s = fn2sh[mu];
                const int l = sh.l[s];
                const int k = mu - sh.ao0[s];
                const double x = gx - sh.xyz[s * 3], y = gy - sh.xyz[s * 3 + 1], z = gz - sh.xyz[s * 3 + 2];
                const double rad = s_rad[wave][s - s0][0], rad1 = s_rad[wave][s - s0][1];
                const int nc = (l + 1) * (l + 2) / 2;
                const double *m = c2s + c2s_off[l] + k * nc;
                switch (l) {
                case 0: ao_angular<0, DERIV>(x, y, z, rad, rad1, m, v, vx, vy, vz); break;
                case 1: ao_angular<1, DERIV>(x, y, z, rad, rad1, m, v, vx, vy, vz); break;
                case 2: ao_angular<2, DERIV>(x, y, z, rad, rad1, m, v, vx, vy, vz); break;
                case 3: ao_angular<3, DERIV>(x, y, z, rad, rad1, m, v, vx, vy, vz); break;
                default: ao_angular<4, DERIV>(x, y, z, rad, rad1, m, v, vx, vy, vz); break;
                }
            }
            if (valid) {
                double *o = ao + gl * ldao + mu;
                o[0] = v;
                if (DERIV) { o[comp_stride] = vx; o[2 * comp_stride] = vy; o[3 * comp_stride] = vz; }
            }
            if (flags) {
                // 16 x 16 (grid x AO) tile flag: some component above thr.  One store per tile and wave:
                // the lowest flagged lane among the lanes that share this lane's column tile writes it.
                bool ex = valid && fabs(v) > thr;
                if (DERIV) ex = ex || (valid && (fabs(vx) > thr || fabs(vy) > thr || fabs(vz) > thr));
                const unsigned long long b = __ballot(ex);
                if (ex) {
                    int start = (mu & ~15) - (mu - lane);
                    if (start < 0) start = 0;
                    const unsigned long long earlier = b & ((1ull << lane) - 1) & ~((1ull << start) - 1);
                    if (!earlier) flags[(gl >> 4) * (ldao >> 4) + (mu >> 4)] = 1;
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int PAMD_grid_partition(double *d_out, const double *d_coords, const double *d_atm_coords,
                        const double *d_radii_table, int natm, long ngrids, int scheme, void *stream);
int PAMD_grid_response(const double *d_coords, const int *d_owner, const double *d_weights, const double *d_e,
                       const double *d_pb, const double *d_atm_coords, const double *d_radii_table, int natm,
                       long ng, int scheme, double *d_out, void *stream);

int PAMD_becke_partition(double *d_out, const double *d_coords, const double *d_atm_coords,
                         const double *d_radii_table, int natm, long ngrids, void *stream)
{
    return PAMD_grid_partition(d_out, d_coords, d_atm_coords, d_radii_table, natm, ngrids, 0, stream);
}

// Same with the cell function chosen by scheme: 0 original Becke, 1 Stratmann-Scuseria-Frisch, 2 Laqua-Kussmann-
// Ochsenfeld (gen_grid.py:203-234 stratmann / original_becke / becke_lko).
int PAMD_grid_partition(double *d_out, const double *d_coords, const double *d_atm_coords,
                        const double *d_radii_table, int natm, long ngrids, int scheme, void *stream)
{
    PAMD_REQUIRE(scheme >= 0 && scheme <= 2, "grid_partition: scheme must be 0 (becke), 1 (stratmann) or 2 (lko)");
    if (ngrids == 0) return 0;
    const dim3 grid(ceil_div(ngrids, 256));
    hipStream_t st = (hipStream_t)stream;
    if (scheme == 0) becke_kernel<0><<<grid, 256, 0, st>>>(d_out, d_coords, d_atm_coords, d_radii_table, natm, ngrids);
    else if (scheme == 1) becke_kernel<1><<<grid, 256, 0, st>>>(d_out, d_coords, d_atm_coords, d_radii_table, natm, ngrids);
    else becke_kernel<2><<<grid, 256, 0, st>>>(d_out, d_coords, d_atm_coords, d_radii_table, natm, ngrids);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// d_out[natm][3] += sum_g e[g] d w_g/dR: weight response of the Becke partition (points follow their owner atoms);
// d_pb[natm][ng] from PAMD_becke_partition on the same points, d_owner[ng] atom index of every point, d_e[ng] per-point
// scalar (XC energy per volume).  pyscf/grad/rks.py:257-340 (get_vxc_full_response, grids_response_cc).
int PAMD_becke_response(const double *d_coords, const int *d_owner, const double *d_weights, const double *d_e,
                        const double *d_pb, const double *d_atm_coords, const double *d_radii_table, int natm,
                        long ng, double *d_out, void *stream)
{
    return PAMD_grid_response(d_coords, d_owner, d_weights, d_e, d_pb, d_atm_coords, d_radii_table, natm, ng, 0, d_out, stream);
}

// Same for the cell function `scheme` of PAMD_grid_partition (d_pb from that call with the same scheme): 0 Becke
// (grids_response_becke), 1 Stratmann (same routine with the Stratmann switch), 2 LKO (grids_response_lko).
int PAMD_grid_response(const double *d_coords, const int *d_owner, const double *d_weights, const double *d_e,
                       const double *d_pb, const double *d_atm_coords, const double *d_radii_table, int natm,
                       long ng, int scheme, double *d_out, void *stream)
{
    PAMD_REQUIRE(scheme >= 0 && scheme <= 2, "grid_response: scheme must be 0 (becke), 1 (stratmann) or 2 (lko)");
    if (ng == 0 || natm == 0) return 0;
    const dim3 grid(ceil_div(ng, 256));
    hipStream_t st = (hipStream_t)stream;
    if (scheme == 0)
        becke_response_kernel<0><<<grid, 256, 0, st>>>(d_coords, d_owner, d_weights, d_e, d_pb, d_atm_coords, d_radii_table, natm, ng, d_out);
    else if (scheme == 1)
        becke_response_kernel<1><<<grid, 256, 0, st>>>(d_coords, d_owner, d_weights, d_e, d_pb, d_atm_coords, d_radii_table, natm, ng, d_out);
    else
        becke_response_kernel<2><<<grid, 256, 0, st>>>(d_coords, d_owner, d_weights, d_e, d_pb, d_atm_coords, d_radii_table, natm, ng, d_out);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// ao[comp][ldg_rows][ldao] (AO index fastest; columns nao..ldao-1 zero), grid points
// [g0, g0+ng) of d_coords[][3]; deriv = 0 (comp 1) or 1 (comp 4); d_fn2sh[mu] = shell of AO mu.
// d_flags (nullable, zeroed by the caller): [ceil(ldg_rows/16)][ldao/16] bytes, set to 1 where the
// 16 x 16 (grid x AO) tile holds a value (any component) above thr -- the screening table that
// GTO_screen_index / make_mask (pyscf/lib/gto/grid_ao_drv.c:32-123) estimate from exponents.
int PAMD_eval_ao(int deriv, const int *d_l, const int *d_ao0, const int *d_prim0, const int *d_nprim,
                 const double *d_xyz, const double *d_exps, const double *d_coefs, int nsh, const int *d_fn2sh,
                 int nao, const double *d_coords, long g0, long ng, const double *d_c2s, const int *d_c2s_off,
                 double *d_ao, long ldg_rows, int ldao, double thr, unsigned char *d_flags, void *stream)
{
    PAMD_REQUIRE(d_flags == nullptr || ldao % 16 == 0, "eval_ao: tile flags need ldao % 16 == 0");
    PAMD_REQUIRE(deriv >= 0 && deriv <= 2, "eval_ao: deriv must be 0, 1 or 2");
    PAMD_REQUIRE(deriv < 2 || d_flags == nullptr, "eval_ao: tile flags are not produced for deriv = 2");
    PAMD_REQUIRE(ldao >= nao, "eval_ao: ldao < nao");
    if (ng == 0 || nao == 0) return 0;
    AOShells sh{d_l, d_ao0, d_prim0, d_nprim, d_xyz, d_exps, d_coefs};
    dim3 grid(ceil_div(ng, 4));
    if (deriv == 2)
        eval_ao_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(sh, nsh, d_fn2sh, d_coords, g0, ng, d_c2s, d_c2s_off, d_ao, ldg_rows, ldao, nao, thr, d_flags);
    else if (deriv)
        eval_ao_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(sh, nsh, d_fn2sh, d_coords, g0, ng, d_c2s, d_c2s_off, d_ao, ldg_rows, ldao, nao, thr, d_flags);
    else
        eval_ao_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(sh, nsh, d_fn2sh, d_coords, g0, ng, d_c2s, d_c2s_off, d_ao, ldg_rows, ldao, nao, thr, d_flags);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
