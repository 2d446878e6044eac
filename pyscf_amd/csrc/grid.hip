// DFT grid kernels: Becke partition weights and AO (+ gradient) values on grid points.
//   PAMD_becke_partition <- VXCgen_grid  (pyscf/lib/dft/grid_basis.c:32-101)
//   PAMD_eval_ao         <- GTOval_sph_deriv0 / GTOval_sph_deriv1 (pyscf/gto/eval_gto.py:31-144;
//                           pyscf/lib/gto/grid_ao_drv.c:222-284 GTOeval_sph_iter, :125-141 GTOnabla1,
//                           pyscf/lib/gto/deriv1.c:129-520)
// Output layout of PAMD_eval_ao is the reference's: ao[comp][nao][ldg] (grid index fastest,
// eval_gto.py:123-127), comp = 1 (value) or 4 (value, d/dx, d/dy, d/dz).
#include "common.h"

using namespace pamd;

namespace {

// out[ia][g] = prod_{j != ia} s(mu_ij)-type cell function of atom ia (unnormalised)
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
            double s = (di - dj) / sqrt(ax * ax + ay * ay + az * az);
            if (radii) s += radii[i * natm + j] * (1 - s * s);
            s = (3 - s * s) * s * .5;
            s = (3 - s * s) * s * .5;
            s = ((3 - s * s) * s * .5) * .5;
            pi *= .5 - s;
            out[(long)j * ngrids + g] *= .5 + s;
        }
        out[(long)i * ngrids + g] = pi;
    }
}

constexpr int AO_LMAX = 4;
constexpr int AO_NC = (AO_LMAX + 1) * (AO_LMAX + 2) / 2;

struct AOShells {
    const int *l, *ao0, *prim0, *nprim;
    const double *xyz, *exps, *coefs;
};

__device__ inline void cart_exps(int l, int c, int &lx, int &ly, int &lz)
{
    int x = l, rem = c;
    while (rem > l - x) { rem -= (l - x + 1); x--; }
    lx = x; ly = (l - x) - rem; lz = rem;
}

// one thread per grid point, blockIdx.y strides over shells
template <int DERIV>
__global__ __launch_bounds__(256) void eval_ao_kernel(AOShells sh, int nsh, const double *__restrict__ coords,
                                                      long g0, long ng, const double *__restrict__ c2s,
                                                      const int *__restrict__ c2s_off, double *__restrict__ ao,
                                                      long ldg, int nao)
{
    const long gl = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = gl < ng;
    const long g = g0 + (valid ? gl : 0);
    const double gx = coords[g * 3 + 0], gy = coords[g * 3 + 1], gz = coords[g * 3 + 2];
    const long comp_stride = (long)nao * ldg;
    for (int s = blockIdx.y; s < nsh; s += gridDim.y) {
        const int l = sh.l[s];
        const double x = gx - sh.xyz[s * 3], y = gy - sh.xyz[s * 3 + 1], z = gz - sh.xyz[s * 3 + 2];
        const double r2 = x * x + y * y + z * z;
        double rad = 0, rad1 = 0;                        // sum c e^{-a r2},  sum -2 a c e^{-a r2}
        for (int p = 0; p < sh.nprim[s]; p++) {
            const double a = sh.exps[sh.prim0[s] + p];
            const double e = sh.coefs[sh.prim0[s] + p] * exp(-a * r2);
            rad += e;
            rad1 += -2 * a * e;
        }
        // monomial powers
        double px[AO_LMAX + 2], py[AO_LMAX + 2], pz[AO_LMAX + 2];
        px[0] = py[0] = pz[0] = 1;
        for (int i = 1; i <= l + 1; i++) { px[i] = px[i - 1] * x; py[i] = py[i - 1] * y; pz[i] = pz[i - 1] * z; }
        const int nc = (l + 1) * (l + 2) / 2;
        double cv[AO_NC], cdx[AO_NC], cdy[AO_NC], cdz[AO_NC];
        for (int c = 0; c < nc; c++) {
            int lx, ly, lz;
            cart_exps(l, c, lx, ly, lz);
            const double poly = px[lx] * py[ly] * pz[lz];
            cv[c] = poly * rad;
            if (DERIV) {
                cdx[c] = rad1 * px[lx + 1] * py[ly] * pz[lz] + (lx ? lx * px[lx - 1] * py[ly] * pz[lz] * rad : 0.0);
                cdy[c] = rad1 * px[lx] * py[ly + 1] * pz[lz] + (ly ? ly * px[lx] * py[ly - 1] * pz[lz] * rad : 0.0);
                cdz[c] = rad1 * px[lx] * py[ly] * pz[lz + 1] + (lz ? lz * px[lx] * py[ly] * pz[lz - 1] * rad : 0.0);
            }
        }
        const double *m = c2s + c2s_off[l];
        for (int k = 0; k < 2 * l + 1; k++) {
            double v = 0, vx = 0, vy = 0, vz = 0;
            for (int c = 0; c < nc; c++) {
                const double f = m[k * nc + c];
                v += f * cv[c];
                if (DERIV) { vx += f * cdx[c]; vy += f * cdy[c]; vz += f * cdz[c]; }
            }
            if (valid) {
                double *o = ao + (long)(sh.ao0[s] + k) * ldg + gl;
                o[0] = v;
                if (DERIV) { o[comp_stride] = vx; o[2 * comp_stride] = vy; o[3 * comp_stride] = vz; }
            }
        }
    }
}

}  // namespace

extern "C" {

int PAMD_becke_partition(double *d_out, const double *d_coords, const double *d_atm_coords,
                         const double *d_radii_table, int natm, long ngrids, void *stream)
{
    if (ngrids == 0) return 0;
    becke_kernel<<<ceil_div(ngrids, 256), 256, 0, (hipStream_t)stream>>>(d_out, d_coords, d_atm_coords,
                                                                         d_radii_table, natm, ngrids);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// ao[comp][nao][ldg], grid points [g0, g0+ng) of d_coords[][3]; deriv = 0 or 1
int PAMD_eval_ao(int deriv, const int *d_l, const int *d_ao0, const int *d_prim0, const int *d_nprim,
                 const double *d_xyz, const double *d_exps, const double *d_coefs, int nsh, int nao,
                 const double *d_coords, long g0, long ng, const double *d_c2s, const int *d_c2s_off,
                 double *d_ao, long ldg, void *stream)
{
    PAMD_REQUIRE(deriv == 0 || deriv == 1, "eval_ao: deriv must be 0 or 1");
    if (ng == 0 || nsh == 0) return 0;
    AOShells sh{d_l, d_ao0, d_prim0, d_nprim, d_xyz, d_exps, d_coefs};
    int ny = nsh < 64 ? nsh : 64;
    dim3 grid(ceil_div(ng, 256), ny);
    if (deriv)
        eval_ao_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(sh, nsh, d_coords, g0, ng, d_c2s, d_c2s_off, d_ao, ldg, nao);
    else
        eval_ao_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(sh, nsh, d_coords, g0, ng, d_c2s, d_c2s_off, d_ao, ldg, nao);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
