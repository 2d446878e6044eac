// Exchange-correlation grid kernels (closed shell, LDA / GGA).
//   PAMD_rho_from_mo      <- numint.eval_rho2 (pyscf/dft/numint.py:328-469): rho, grad rho from c = ao . C_occ
//   PAMD_rho_from_dm      <- numint.eval_rho  (:116-225): rho, grad rho from ao and ao . D
//   PAMD_eval_xc          <- LIBXC_eval_xc (pyscf/lib/dft/libxc_itrf.c:968-1024; libxc 7.1.2 is not in the
//                            reference tree) + xc_deriv.transform_vxc (pyscf/dft/xc_deriv.py:32-85) + the
//                            weighting of numint.nr_rks (:1132-1153).  Functionals are restated from
//                            the literature and differentiated by forward-mode automatic differentiation
//                            (dual numbers), not by hand:
//                              Slater exchange; VWN5 and VWN-RPA correlation (Vosko, Wilk, Nusair 1980);
//                              B88 exchange (Becke 1988); LYP correlation (Lee, Yang, Parr 1988 in the
//                              Miehlich-Savin-Stoll-Preuss 1989 form); PBE exchange and correlation.
//   PAMD_scale_ao         <- numint._scale_ao (:803-834 / VXCdscale_ao_sparse)
//   PAMD_dgemm_nt         <- numint._dot_ao_ao (:836-874 / VXCdot_ao_ao_sparse): vmat = ao0 . aow^T
#include "common.h"

using namespace pamd;

namespace {

// ---------------------------------------------------------------------------- dual numbers
// Scalar with a first-order part along one direction (the first-order density of the response kernel, PAMD_eval_fxc).
struct Eps {
    double v, e;
};
__device__ inline Eps operator+(Eps a, Eps b) { return {a.v + b.v, a.e + b.e}; }
__device__ inline Eps operator-(Eps a, Eps b) { return {a.v - b.v, a.e - b.e}; }
__device__ inline Eps operator-(Eps a) { return {-a.v, -a.e}; }
__device__ inline Eps operator*(Eps a, Eps b) { return {a.v * b.v, a.e * b.v + a.v * b.e}; }
__device__ inline Eps operator/(Eps a, Eps b) { double iv = 1.0 / b.v, q = a.v * iv; return {q, (a.e - q * b.e) * iv}; }
__device__ inline Eps operator+(Eps a, double b) { return {a.v + b, a.e}; }
__device__ inline Eps operator+(double a, Eps b) { return {a + b.v, b.e}; }
__device__ inline Eps operator-(Eps a, double b) { return {a.v - b, a.e}; }
__device__ inline Eps operator-(double a, Eps b) { return {a - b.v, -b.e}; }
__device__ inline Eps operator*(Eps a, double b) { return {a.v * b, a.e * b}; }
__device__ inline Eps operator*(double a, Eps b) { return {a * b.v, a * b.e}; }
__device__ inline Eps operator/(Eps a, double b) { return {a.v / b, a.e / b}; }
__device__ inline Eps operator/(double a, Eps b) { double q = a / b.v; return {q, -q * b.e / b.v}; }
__device__ inline double s_pow(double a, double p) { return pow(a, p); }
__device__ inline double s_sqrt(double a) { return sqrt(a); }
__device__ inline double s_log(double a) { return log(a); }
__device__ inline double s_exp(double a) { return exp(a); }
__device__ inline double s_atan(double a) { return atan(a); }
__device__ inline double s_asinh(double a) { return asinh(a); }
__device__ inline Eps s_pow(Eps a, double p) { double f = pow(a.v, p); return {f, p * f / a.v * a.e}; }
__device__ inline Eps s_sqrt(Eps a) { double f = sqrt(a.v); return {f, 0.5 / f * a.e}; }
__device__ inline Eps s_log(Eps a) { return {log(a.v), a.e / a.v}; }
__device__ inline Eps s_exp(Eps a) { double f = exp(a.v); return {f, f * a.e}; }
__device__ inline Eps s_atan(Eps a) { return {atan(a.v), a.e / (1.0 + a.v * a.v)}; }
__device__ inline Eps s_asinh(Eps a) { return {asinh(a.v), a.e / sqrt(1.0 + a.v * a.v)}; }
__device__ inline double s_erf(double a) { return erf(a); }
__device__ inline Eps s_erf(Eps a) { return {erf(a.v), 1.1283791670955126 * exp(-a.v * a.v) * a.e}; }
__device__ inline double s_val(double a) { return a; }
__device__ inline double s_val(Eps a) { return a.v; }
template <class S> __device__ inline S lift(double v);
template <> __device__ inline double lift<double>(double v) { return v; }
template <> __device__ inline Eps lift<Eps>(double v) { return Eps{v, 0.0}; }

// value, d/d rho, d/d sigma over the scalar S: double (energy and potential) or Eps (their first-order change)
template <class S>
struct DualT {
    S v, r, s;
};
template <class S> __device__ inline DualT<S> mk(double v) { return DualT<S>{lift<S>(v), lift<S>(0.0), lift<S>(0.0)}; }
template <class S> __device__ inline DualT<S> operator+(DualT<S> a, DualT<S> b) { return {a.v + b.v, a.r + b.r, a.s + b.s}; }
template <class S> __device__ inline DualT<S> operator-(DualT<S> a, DualT<S> b) { return {a.v - b.v, a.r - b.r, a.s - b.s}; }
template <class S> __device__ inline DualT<S> operator-(DualT<S> a) { return {-a.v, -a.r, -a.s}; }
template <class S> __device__ inline DualT<S> operator*(DualT<S> a, DualT<S> b) { return {a.v * b.v, a.r * b.v + a.v * b.r, a.s * b.v + a.v * b.s}; }
template <class S> __device__ inline DualT<S> operator/(DualT<S> a, DualT<S> b)
{
    S iv = 1.0 / b.v, q = a.v * iv;
    return {q, (a.r - q * b.r) * iv, (a.s - q * b.s) * iv};
}
template <class S> __device__ inline DualT<S> operator+(DualT<S> a, double b) { return {a.v + b, a.r, a.s}; }
template <class S> __device__ inline DualT<S> operator+(double a, DualT<S> b) { return {a + b.v, b.r, b.s}; }
template <class S> __device__ inline DualT<S> operator-(DualT<S> a, double b) { return {a.v - b, a.r, a.s}; }
template <class S> __device__ inline DualT<S> operator-(double a, DualT<S> b) { return {a - b.v, -b.r, -b.s}; }
template <class S> __device__ inline DualT<S> operator*(DualT<S> a, double b) { return {a.v * b, a.r * b, a.s * b}; }
template <class S> __device__ inline DualT<S> operator*(double a, DualT<S> b) { return b * a; }
template <class S> __device__ inline DualT<S> operator/(DualT<S> a, double b) { return a * (1.0 / b); }
template <class S> __device__ inline DualT<S> operator/(double a, DualT<S> b) { return DualT<S>{lift<S>(a), lift<S>(0.0), lift<S>(0.0)} / b; }
template <class S> __device__ inline DualT<S> chain(DualT<S> a, S f, S df) { return {f, df * a.r, df * a.s}; }
template <class S> __device__ inline DualT<S> dpow(DualT<S> a, double p) { S f = s_pow(a.v, p); return chain(a, f, p * f / a.v); }
template <class S> __device__ inline DualT<S> dsqrt(DualT<S> a) { S f = s_sqrt(a.v); return chain(a, f, 0.5 / f); }
template <class S> __device__ inline DualT<S> dlog(DualT<S> a) { return chain(a, s_log(a.v), 1.0 / a.v); }
template <class S> __device__ inline DualT<S> dexp(DualT<S> a) { S f = s_exp(a.v); return chain(a, f, f); }
template <class S> __device__ inline DualT<S> datan(DualT<S> a) { return chain(a, s_atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
template <class S> __device__ inline DualT<S> dasinh(DualT<S> a) { return chain(a, s_asinh(a.v), 1.0 / s_sqrt(1.0 + a.v * a.v)); }

// ---------------------------------------------------------------------------- functionals
// All return the energy density per unit volume e(rho, sigma) for a closed-shell density.
constexpr double PI = 3.14159265358979323846;

template <class S> __device__ inline DualT<S> slater_x(DualT<S> rho)
{
    const double cx = 0.75 * 0.98474502184269654;     // (3/4)(3/pi)^(1/3)
    return -cx * dpow(rho, 4.0 / 3.0);
}

// VWN paramagnetic correlation energy per particle; params (A, x0, b, c)
template <class S> __device__ inline DualT<S> vwn_eps(DualT<S> rho, double A, double x0, double b, double c)
{
    DualT<S> rs = dpow(3.0 / (4.0 * PI) / rho, 1.0 / 3.0);
    DualT<S> x = dsqrt(rs);
    const double Q = sqrt(4 * c - b * b);
    DualT<S> X = x * x + b * x + c;
    const double X0 = x0 * x0 + b * x0 + c;
    DualT<S> at = datan(Q / (2.0 * x + b));
    DualT<S> t1 = dlog(x * x / X) + (2 * b / Q) * at;
    DualT<S> t2 = dlog((x - x0) * (x - x0) / X) + (2 * (b + 2 * x0) / Q) * at;
    return A * (t1 - (b * x0 / X0) * t2);
}
template <class S> __device__ inline DualT<S> vwn5_c(DualT<S> rho) { return rho * vwn_eps(rho, 0.0310907, -0.10498, 3.72744, 12.9352); }
template <class S> __device__ inline DualT<S> vwnrpa_c(DualT<S> rho) { return rho * vwn_eps(rho, 0.0310907, -0.409286, 13.0720, 42.7198); }

// B88 exchange, spin-scaled to the closed-shell case (rho_s = rho/2, sigma_ss = sigma/4)
template <class S> __device__ inline DualT<S> b88_x(DualT<S> rho, DualT<S> sigma)
{
    const double beta = 0.0042;
    const double cx = 1.5 * 0.62035049089940001;      // (3/2)(3/(4 pi))^(1/3)
    DualT<S> rs = 0.5 * rho;
    DualT<S> r43 = dpow(rs, 4.0 / 3.0);
    DualT<S> g = dsqrt(0.25 * sigma + 1e-300);
    DualT<S> x = g / r43;
    DualT<S> e = -cx * r43 - beta * r43 * x * x / (1.0 + 6.0 * beta * x * dasinh(x));
    return 2.0 * e;
}

// LYP correlation, closed shell (Miehlich et al. form)
template <class S> __device__ inline DualT<S> lyp_c(DualT<S> rho, DualT<S> sigma)
{
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    const double CF = 0.3 * 9.5707800006273392;       // (3/10)(3 pi^2)^(2/3)
    DualT<S> rm13 = dpow(rho, -1.0 / 3.0);
    DualT<S> den = 1.0 + d * rm13;
    DualT<S> omega = dexp(-c * rm13) / den * dpow(rho, -11.0 / 3.0);
    DualT<S> delta = c * rm13 + d * rm13 / den;
    DualT<S> ra = 0.5 * rho;                               // = rb
    DualT<S> saa = 0.25 * sigma;                           // |grad rho_a|^2 = |grad rho_b|^2, total sigma
    DualT<S> rab = ra * ra;
    DualT<S> t1 = -a * 4.0 / den * rab / rho;
    DualT<S> br = rab * (pow(2.0, 11.0 / 3.0) * CF * 2.0 * dpow(ra, 8.0 / 3.0)
                     + (47.0 / 18.0 - 7.0 / 18.0 * delta) * sigma
                     - (2.5 - delta / 18.0) * (2.0 * saa)
                     - (delta - 11.0) / 9.0 * (saa))       // (ra/rho + rb/rho) saa = saa
              - (2.0 / 3.0) * rho * rho * sigma
              + 2.0 * ((2.0 / 3.0) * rho * rho - ra * ra) * saa;
    return t1 - a * b * omega * br;
}

// PBE exchange (closed shell) and correlation (zeta = 0)
template <class S> __device__ inline DualT<S> pbe_x(DualT<S> rho, DualT<S> sigma)
{
    const double kappa = 0.804, mu = 0.2195149727645171;
    DualT<S> ex_lda = slater_x(rho);
    DualT<S> kf = dpow(3.0 * PI * PI * rho, 1.0 / 3.0);
    DualT<S> s2 = sigma / (4.0 * kf * kf * rho * rho);
    DualT<S> fx = 1.0 + kappa - kappa / (1.0 + mu / kappa * s2);
    return ex_lda * fx;
}
template <class S> __device__ inline DualT<S> pw92_eps(DualT<S> rs)
{   // PW92 paramagnetic, libxc "pw_mod" parameters
    const double A = 0.0310907, a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    DualT<S> srs = dsqrt(rs);
    DualT<S> q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
    return -2.0 * A * (1.0 + a1 * rs) * dlog(1.0 + 1.0 / q);
}
template <class S> __device__ inline DualT<S> pbe_c(DualT<S> rho, DualT<S> sigma)
{
    const double beta = 0.06672455060314922, gamma = 0.031090690869654895;   // (1 - ln 2)/pi^2
    DualT<S> rs = dpow(3.0 / (4.0 * PI) / rho, 1.0 / 3.0);
    DualT<S> ec = pw92_eps(rs);
    DualT<S> kf = dpow(3.0 * PI * PI * rho, 1.0 / 3.0);
    DualT<S> ks = dsqrt(4.0 * kf / PI);
    DualT<S> t2 = sigma / (4.0 * ks * ks * rho * rho);
    DualT<S> Aa = beta / gamma / (dexp(-ec / gamma) - 1.0);
    DualT<S> num = 1.0 + Aa * t2;
    DualT<S> H = gamma * dlog(1.0 + beta / gamma * t2 * num / (1.0 + Aa * t2 + Aa * Aa * t2 * t2));
    return rho * (ec + H);
}

using Dual = DualT<double>;

// attenuated (short-range) B88 exchange of one spin channel, defined with the spin-polarised functionals below
template <class T> __device__ inline T ityh_spin(T r, T s, double omega);
template <class T> __device__ inline T wb97_xc(T ra, T rb, T saa, T sbb, double omega);

enum { F_SLATER = 0, F_VWN5, F_VWNRPA, F_B88, F_LYP, F_PBEX, F_PBEC, F_ITYH, F_WB97, F_NUM };

struct XCSpec {
    double fac[F_NUM];
    double omega;              // range-separation parameter of the attenuated exchange (F_ITYH)
};

// rho[4][ldg] (rho, dx, dy, dz; only row 0 used for LDA), weights[ng]
// wv[4][ldg]: wv0 = 0.5 w vrho, wv1..3 = w 2 vsigma grad rho     (numint.py:1139-1153, xc_deriv.py:81-84)
// acc[0] += sum w rho, acc[1] += sum w e
__global__ __launch_bounds__(256) void eval_xc_kernel(XCSpec spec, int gga, const double *__restrict__ rho,
                                                      const double *__restrict__ weights, long ng, long ldg,
                                                      double *__restrict__ wv, double *__restrict__ exc_out,
                                                      double *__restrict__ acc)
{
    long g = (long)blockIdx.x * 256 + threadIdx.x;
    double nel = 0, exc = 0;
    if (g < ng) {
        const double r = rho[g];
        const double w = weights[g];
        double gxv = 0, gyv = 0, gzv = 0, sig = 0;
        if (gga) {
            gxv = rho[ldg + g]; gyv = rho[2 * ldg + g]; gzv = rho[3 * ldg + g];
            sig = gxv * gxv + gyv * gyv + gzv * gzv;
        }
        double e = 0, vr = 0, vs = 0;
        if (r > 1e-14) {
            Dual dr{r, 1.0, 0.0}, ds{sig, 0.0, 1.0};
            Dual tot = mk<double>(0);
            if (spec.fac[F_SLATER] != 0) tot = tot + spec.fac[F_SLATER] * slater_x(dr);
            if (spec.fac[F_VWN5] != 0) tot = tot + spec.fac[F_VWN5] * vwn5_c(dr);
            if (spec.fac[F_VWNRPA] != 0) tot = tot + spec.fac[F_VWNRPA] * vwnrpa_c(dr);
            if (spec.fac[F_B88] != 0) tot = tot + spec.fac[F_B88] * b88_x(dr, ds);
            if (spec.fac[F_LYP] != 0) tot = tot + spec.fac[F_LYP] * lyp_c(dr, ds);
            if (spec.fac[F_PBEX] != 0) tot = tot + spec.fac[F_PBEX] * pbe_x(dr, ds);
            if (spec.fac[F_PBEC] != 0) tot = tot + spec.fac[F_PBEC] * pbe_c(dr, ds);
            if (spec.fac[F_ITYH] != 0) tot = tot + spec.fac[F_ITYH] * (2.0 * ityh_spin(0.5 * dr, 0.25 * ds, spec.omega));
            if (spec.fac[F_WB97] != 0) tot = tot + spec.fac[F_WB97] * wb97_xc(0.5 * dr, 0.5 * dr, 0.25 * ds, 0.25 * ds, spec.omega);
            e = tot.v; vr = tot.r; vs = tot.s;
        }
        nel = w * r;
        exc = w * e;
        wv[g] = 0.5 * w * vr;
        if (gga) {
            const double f = 2.0 * w * vs;
            wv[ldg + g] = f * gxv; wv[2 * ldg + g] = f * gyv; wv[3 * ldg + g] = f * gzv;
        }
        if (exc_out) exc_out[g] = (r > 1e-14) ? e / r : 0.0;
    }
    for (int off = 32; off > 0; off >>= 1) { nel += __shfl_down(nel, off, 64); exc += __shfl_down(exc, off, 64); }
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = nel; red[1][threadIdx.x >> 6] = exc; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(acc + threadIdx.x, v);
    }
}

// First-order change of the XC potential weights along a first-order density rho1 (numint.nr_rks_fxc, numint.py:1418-1530,
// weights of _rks_gga_wv1 :1560-1576): the functionals are evaluated on DualT<Eps>, value and (d/d rho, d/d sigma) parts
// each carrying the derivative along (rho1, sigma1 = 2 grad rho0 . grad rho1) - forward over forward AD, no hand-written
// second derivatives.
//   wv1[0] = 0.5 w d(vrho),   wv1[k] = 2 w [ d(vsigma) grad_k rho0 + vsigma grad_k rho1 ]
__global__ __launch_bounds__(256) void eval_fxc_kernel(XCSpec spec, int gga, const double *__restrict__ rho0,
                                                       const double *__restrict__ rho1,
                                                       const double *__restrict__ weights, long ng, long ldg,
                                                       double *__restrict__ wv)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= ng) return;
    const double r = rho0[g], r1 = rho1[g];
    const double w = weights[g];
    double gx = 0, gy = 0, gz = 0, hx = 0, hy = 0, hz = 0, sig = 0, sig1 = 0;
    if (gga) {
        gx = rho0[ldg + g]; gy = rho0[2 * ldg + g]; gz = rho0[3 * ldg + g];
        hx = rho1[ldg + g]; hy = rho1[2 * ldg + g]; hz = rho1[3 * ldg + g];
        sig = gx * gx + gy * gy + gz * gz;
        sig1 = 2 * (gx * hx + gy * hy + gz * hz);
    }
    double dvr = 0, vs = 0, dvs = 0;
    if (r > 1e-14) {
        DualT<Eps> dr{{r, r1}, {1.0, 0.0}, {0.0, 0.0}}, ds{{sig, sig1}, {0.0, 0.0}, {1.0, 0.0}};
        DualT<Eps> tot = mk<Eps>(0);
        if (spec.fac[F_SLATER] != 0) tot = tot + spec.fac[F_SLATER] * slater_x(dr);
        if (spec.fac[F_VWN5] != 0) tot = tot + spec.fac[F_VWN5] * vwn5_c(dr);
        if (spec.fac[F_VWNRPA] != 0) tot = tot + spec.fac[F_VWNRPA] * vwnrpa_c(dr);
        if (spec.fac[F_B88] != 0) tot = tot + spec.fac[F_B88] * b88_x(dr, ds);
        if (spec.fac[F_LYP] != 0) tot = tot + spec.fac[F_LYP] * lyp_c(dr, ds);
        if (spec.fac[F_PBEX] != 0) tot = tot + spec.fac[F_PBEX] * pbe_x(dr, ds);
        if (spec.fac[F_PBEC] != 0) tot = tot + spec.fac[F_PBEC] * pbe_c(dr, ds);
        if (spec.fac[F_ITYH] != 0) tot = tot + spec.fac[F_ITYH] * (2.0 * ityh_spin(0.5 * dr, 0.25 * ds, spec.omega));
        if (spec.fac[F_WB97] != 0) tot = tot + spec.fac[F_WB97] * wb97_xc(0.5 * dr, 0.5 * dr, 0.25 * ds, 0.25 * ds, spec.omega);
        dvr = tot.r.e; vs = tot.s.v; dvs = tot.s.e;
    }
    wv[g] = 0.5 * w * dvr;
    if (gga) {
        wv[ldg + g] = 2.0 * w * (dvs * gx + vs * hx);
        wv[2 * ldg + g] = 2.0 * w * (dvs * gy + vs * hy);
        wv[3 * ldg + g] = 2.0 * w * (dvs * gz + vs * hz);
    }
}

// ============================================================================ spin-polarised (UKS)
// forward-mode AD with 5 directions: d/d(rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb)
template <class S>
struct D5T {
    S v, d[5];
};
template <class S> __device__ inline D5T<S> c5(double v) { D5T<S> r; r.v = lift<S>(v); for (int i = 0; i < 5; i++) r.d[i] = lift<S>(0.0); return r; }
__device__ inline void floor_at(double &x, double f) { if (x < f) x = f; }
__device__ inline void floor_at(Eps &x, double f) { if (x.v < f) { x.v = f; x.e = 0.0; } }
template <class S> __device__ inline D5T<S> operator+(D5T<S> a, D5T<S> b) { D5T<S> r; r.v = a.v + b.v; for (int i = 0; i < 5; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <class S> __device__ inline D5T<S> operator-(D5T<S> a, D5T<S> b) { D5T<S> r; r.v = a.v - b.v; for (int i = 0; i < 5; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <class S> __device__ inline D5T<S> operator-(D5T<S> a) { D5T<S> r; r.v = -a.v; for (int i = 0; i < 5; i++) r.d[i] = -a.d[i]; return r; }
template <class S> __device__ inline D5T<S> operator*(D5T<S> a, D5T<S> b) { D5T<S> r; r.v = a.v * b.v; for (int i = 0; i < 5; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <class S> __device__ inline D5T<S> operator/(D5T<S> a, D5T<S> b)
{
    D5T<S> r; S iv = 1.0 / b.v; r.v = a.v * iv;
    for (int i = 0; i < 5; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * iv;
    return r;
}
template <class S> __device__ inline D5T<S> operator+(D5T<S> a, double b) { a.v = a.v + b; return a; }
template <class S> __device__ inline D5T<S> operator+(double a, D5T<S> b) { b.v = b.v + a; return b; }
template <class S> __device__ inline D5T<S> operator-(D5T<S> a, double b) { a.v = a.v - b; return a; }
template <class S> __device__ inline D5T<S> operator-(double a, D5T<S> b) { return c5<S>(a) - b; }
template <class S> __device__ inline D5T<S> operator*(D5T<S> a, double b) { a.v = a.v * b; for (int i = 0; i < 5; i++) a.d[i] = a.d[i] * b; return a; }
template <class S> __device__ inline D5T<S> operator*(double a, D5T<S> b) { return b * a; }
template <class S> __device__ inline D5T<S> operator/(D5T<S> a, double b) { return a * (1.0 / b); }
template <class S> __device__ inline D5T<S> operator/(double a, D5T<S> b) { return c5<S>(a) / b; }
template <class S> __device__ inline D5T<S> chain5(D5T<S> a, S f, S df) { D5T<S> r; r.v = f; for (int i = 0; i < 5; i++) r.d[i] = df * a.d[i]; return r; }
template <class S> __device__ inline D5T<S> pow5(D5T<S> a, double p) { S f = s_pow(a.v, p); return chain5(a, f, p * f / a.v); }
template <class S> __device__ inline D5T<S> sqrt5(D5T<S> a) { S f = s_sqrt(a.v); return chain5(a, f, 0.5 / f); }
template <class S> __device__ inline D5T<S> log5(D5T<S> a) { return chain5(a, s_log(a.v), 1.0 / a.v); }
template <class S> __device__ inline D5T<S> exp5(D5T<S> a) { S f = s_exp(a.v); return chain5(a, f, f); }
template <class S> __device__ inline D5T<S> atan5(D5T<S> a) { return chain5(a, s_atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
template <class S> __device__ inline D5T<S> asinh5(D5T<S> a) { return chain5(a, s_asinh(a.v), 1.0 / s_sqrt(1.0 + a.v * a.v)); }

template <class S> __device__ inline D5T<S> slater_pol(D5T<S> ra, D5T<S> rb)
{
    const double cx = 1.5 * 0.62035049089940001;       // (3/2)(3/(4 pi))^(1/3)
    return -cx * (pow5(ra, 4.0 / 3.0) + pow5(rb, 4.0 / 3.0));
}
template <class S> __device__ inline D5T<S> vwn_eps5(D5T<S> rho, double A, double x0, double b, double c)
{
    D5T<S> rs = pow5(3.0 / (4.0 * PI) / rho, 1.0 / 3.0);
    D5T<S> x = sqrt5(rs);
    const double Q = sqrt(4 * c - b * b);
    D5T<S> X = x * x + b * x + c;
    const double X0 = x0 * x0 + b * x0 + c;
    D5T<S> at = atan5(Q / (2.0 * x + b));
    D5T<S> t1 = log5(x * x / X) + (2 * b / Q) * at;
    D5T<S> t2 = log5((x - x0) * (x - x0) / X) + (2 * (b + 2 * x0) / Q) * at;
    return A * (t1 - (b * x0 / X0) * t2);
}
template <class S> __device__ inline D5T<S> fzeta5(D5T<S> zeta)
{
    // guard the fully polarised limit: (1 -+ zeta)^(4/3) with a tiny floor keeps the derivative finite
    D5T<S> p = 1.0 + zeta, m = 1.0 - zeta;
    floor_at(p.v, 1e-14);
    floor_at(m.v, 1e-14);
    return (pow5(p, 4.0 / 3.0) + pow5(m, 4.0 / 3.0) - 2.0) / (2.5198420997897464 - 2.0);
}
// VWN5: para + spin stiffness + ferro (Vosko, Wilk, Nusair 1980, eq. 4.4 interpolation)
template <class S> __device__ inline D5T<S> vwn5_pol(D5T<S> rho, D5T<S> zeta)
{
    D5T<S> eP = vwn_eps5(rho, 0.0310907, -0.10498, 3.72744, 12.9352);
    D5T<S> eF = vwn_eps5(rho, 0.01554535, -0.32500, 7.06042, 18.0578);
    D5T<S> eA = vwn_eps5(rho, -1.0 / (6.0 * PI * PI), -0.0047584, 1.13107, 13.0045);
    const double fpp = 4.0 / (9.0 * (1.2599210498948732 - 1.0));
    D5T<S> f = fzeta5(zeta);
    D5T<S> z4 = zeta * zeta * zeta * zeta;
    return rho * (eP + eA * f / fpp * (1.0 - z4) + (eF - eP) * f * z4);
}
// VWN-RPA (libxc LDA_C_VWN_RPA): linear interpolation in f(zeta) between para and ferro RPA fits
template <class S> __device__ inline D5T<S> vwnrpa_pol(D5T<S> rho, D5T<S> zeta)
{
    D5T<S> eP = vwn_eps5(rho, 0.0310907, -0.409286, 13.0720, 42.7198);
    D5T<S> eF = vwn_eps5(rho, 0.01554535, -0.743294, 20.1231, 101.578);
    D5T<S> f = fzeta5(zeta);
    return rho * (eP * (1.0 - f) + eF * f);
}
template <class S> __device__ inline D5T<S> b88_spin(D5T<S> r, D5T<S> s)
{
    const double beta = 0.0042, cx = 1.5 * 0.62035049089940001;
    D5T<S> r43 = pow5(r, 4.0 / 3.0);
    D5T<S> x = sqrt5(s + 1e-300) / r43;
    return -cx * r43 - beta * r43 * x * x / (1.0 + 6.0 * beta * x * asinh5(x));
}
// ---- one spelling of the elementary functions for both AD types (DualT: closed shell, D5T: spin-polarised)
template <class S> __device__ inline DualT<S> ad_pow(DualT<S> a, double p) { return dpow(a, p); }
template <class S> __device__ inline D5T<S> ad_pow(D5T<S> a, double p) { return pow5(a, p); }
template <class S> __device__ inline DualT<S> ad_sqrt(DualT<S> a) { return dsqrt(a); }
template <class S> __device__ inline D5T<S> ad_sqrt(D5T<S> a) { return sqrt5(a); }
template <class S> __device__ inline DualT<S> ad_exp(DualT<S> a) { return dexp(a); }
template <class S> __device__ inline D5T<S> ad_exp(D5T<S> a) { return exp5(a); }
template <class S> __device__ inline DualT<S> ad_asinh(DualT<S> a) { return dasinh(a); }
template <class S> __device__ inline D5T<S> ad_asinh(D5T<S> a) { return asinh5(a); }
template <class S> __device__ inline DualT<S> ad_erf(DualT<S> a) { return chain(a, s_erf(a.v), 1.1283791670955126 * s_exp(-(a.v * a.v))); }
template <class S> __device__ inline D5T<S> ad_erf(D5T<S> a) { return chain5(a, s_erf(a.v), 1.1283791670955126 * s_exp(-(a.v * a.v))); }
template <class S> __device__ inline DualT<S> ad_log(DualT<S> a) { return dlog(a); }
template <class S> __device__ inline D5T<S> ad_log(D5T<S> a) { return log5(a); }
template <class S> __device__ inline double ad_val(DualT<S> a) { return s_val(a.v); }
template <class S> __device__ inline double ad_val(D5T<S> a) { return s_val(a.v); }

// Attenuation function of the erf-screened exchange hole (Iikura, Tsuneda, Yanai, Hirao, JCP 115, 3540 (2001), eq. 12; libxc
// attenuation_erf):  att(a) = 1 - 8/3 a [sqrt(pi) erf(1/(2a)) + 2a (b - c)],  b = exp(-1/(4a^2)) - 1,  c = 2a^2 b + 1/2.
// The closed form cancels to ~1/(36 a^2) for large a (5e-13 relative error at a = 2 in FP64), so from a = 1.4 on its
// expansion in 1/a^2 is used (8 terms: 6e-16 at a = 1.5).
template <class T> __device__ inline T att_erf(T a)
{
    if (ad_val(a) > 1.4) {
        T i2 = 1.0 / (a * a);
        return i2 * (1.0 / 36.0 + i2 * (-1.0 / 960.0 + i2 * (1.0 / 26880.0 + i2 * (-1.0 / 829440.0 + i2 * (1.0 / 28385280.0 +
               i2 * (-1.0 / 1073479680.0 + i2 * (1.0 / 44590694400.0 + i2 * (-1.0 / 2021444812800.0))))))));
    }
    T b = ad_exp(-1.0 / (4.0 * a * a)) - 1.0;
    T c = 2.0 * a * a * b + 0.5;
    return 1.0 - (8.0 / 3.0) * a * (1.7724538509055159 * ad_erf(1.0 / (2.0 * a)) + 2.0 * a * (b - c));
}

// Short-range B88 exchange of one spin channel (density r, |grad r|^2 = s), ITYH scheme as libxc's gga_x_ityh:
//   e = -cx r^(4/3) F(x) att(a),  F = B88 enhancement,  a = omega / (2 k),  k = sqrt(9 pi / (2 cx F)) r^(1/3)
// (the long-range part of the exchange is the exact one, K_LR(omega); CAM-B3LYP = 0.35 B88 + 0.46 ITYH(0.33) + ...)
template <class T> __device__ inline T ityh_spin(T r, T s, double omega)
{
    const double beta = 0.0042, cx = 1.5 * 0.62035049089940001;
    T r43 = ad_pow(r, 4.0 / 3.0);
    T x = ad_sqrt(s + 1e-300) / r43;
    T F = 1.0 + (beta / cx) * x * x / (1.0 + 6.0 * beta * x * ad_asinh(x));
    T k = ad_sqrt((9.0 * PI / (2.0 * cx)) / F) * ad_pow(r, 1.0 / 3.0);
    T a = omega / (2.0 * k);
    return -cx * r43 * F * att_erf(a);
}
// ---- omega-B97 (Chai, Head-Gordon, JCP 128, 084106 (2008), Table I; libxc hyb_gga_xc_wb97), one template for both AD types.
// B97 series  g(u) = sum_i c_i u^i,  u = gamma s^2 / (1 + gamma s^2)  with  s^2 = sigma / r^(8/3)  written as
// gamma sigma / (r^(8/3) + gamma sigma)  so that a vanishing spin density cannot overflow.
template <class T> __device__ inline T b97_series(const double *c, T u)
{
    return c[0] + u * (c[1] + u * (c[2] + u * (c[3] + u * c[4])));
}
// Perdew-Wang 1992 G function with the ORIGINAL published digits (B97-type functionals; PBE uses the "pw_mod" digits above)
template <class T> __device__ inline T pw92_g_orig(T rs, double A, double a1, double b1, double b2, double b3, double b4)
{
    T srs = ad_sqrt(rs);
    T q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
    return -2.0 * A * (1.0 + a1 * rs) * ad_log(1.0 + 1.0 / q);
}
// e_xc per volume from the spin densities ra, rb and |grad ra|^2 = saa, |grad rb|^2 = sbb:
//   exchange      sum_s  e_x^LSDA(r_s) att(omega / (2 k_F,s)) g_x(u_s),          gamma_x  = 0.004, k_F,s = (6 pi^2 r_s)^(1/3)
//   correlation   sum_s  e_c^PW92(r_s, 0) g_ss(u_s)  +  [e_c^PW92(ra, rb) - same-spin parts] g_os(u_av)   (Stoll partition),
//                 gamma_ss = 0.2, gamma_os = 0.006 on the mean of the two s^2
// The long-range exchange is exact (alpha = 1, omega = 0.4): K_LR on the host side.
template <class T> __device__ inline T wb97_xc(T ra, T rb, T saa, T sbb, double omega)
{
    const double cx[5] = {1.00000, 1.13116, -2.74915, 12.0900, -5.71642};
    const double css[5] = {1.00000, -2.55352, 11.8926, -26.9452, 17.0927};
    const double cos_[5] = {1.00000, 3.99051, -17.0066, 1.07292, 8.88211};
    const double clda = 1.5 * 0.62035049089940001;
    T ra83 = ad_pow(ra, 8.0 / 3.0), rb83 = ad_pow(rb, 8.0 / 3.0);
    T ua = 0.004 * saa / (ra83 + 0.004 * saa + 1e-300), ub = 0.004 * sbb / (rb83 + 0.004 * sbb + 1e-300);
    T kfa = ad_pow(6.0 * PI * PI * ra, 1.0 / 3.0), kfb = ad_pow(6.0 * PI * PI * rb, 1.0 / 3.0);
    T ex = -clda * (ad_pow(ra, 4.0 / 3.0) * att_erf(omega / (2.0 * kfa)) * b97_series(cx, ua) +
                    ad_pow(rb, 4.0 / 3.0) * att_erf(omega / (2.0 * kfb)) * b97_series(cx, ub));
    // PW92 (original digits): paramagnetic, ferromagnetic, spin stiffness
    T n = ra + rb;
    T zeta = (ra - rb) / n;
    T rs = ad_pow(3.0 / (4.0 * PI) / n, 1.0 / 3.0);
    T e0 = pw92_g_orig(rs, 0.031091, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    T e1 = pw92_g_orig(rs, 0.015545, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    T mac = pw92_g_orig(rs, 0.016887, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);
    T p = 1.0 + zeta, m = 1.0 - zeta;
    floor_at(p.v, 1e-14);
    floor_at(m.v, 1e-14);
    T fz = (ad_pow(p, 4.0 / 3.0) + ad_pow(m, 4.0 / 3.0) - 2.0) / (2.5198420997897464 - 2.0);
    T z4 = zeta * zeta * zeta * zeta;
    T ec = n * (e0 - mac * fz / 1.709921 * (1.0 - z4) + (e1 - e0) * fz * z4);
    T rsa = ad_pow(3.0 / (4.0 * PI) / ra, 1.0 / 3.0), rsb = ad_pow(3.0 / (4.0 * PI) / rb, 1.0 / 3.0);
    T eaa = ra * pw92_g_orig(rsa, 0.015545, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    T ebb = rb * pw92_g_orig(rsb, 0.015545, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    T uss_a = 0.2 * saa / (ra83 + 0.2 * saa + 1e-300), uss_b = 0.2 * sbb / (rb83 + 0.2 * sbb + 1e-300);
    T s_av = 0.5 * (saa * rb83 + sbb * ra83);
    T uos = 0.006 * s_av / (ra83 * rb83 + 0.006 * s_av + 1e-300);
    return ex + eaa * b97_series(css, uss_a) + ebb * b97_series(css, uss_b) + (ec - eaa - ebb) * b97_series(cos_, uos);
}
template <class S> __device__ inline D5T<S> lyp_pol(D5T<S> ra, D5T<S> rb, D5T<S> saa, D5T<S> sab, D5T<S> sbb)
{
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    const double CF = 0.3 * 9.5707800006273392;
    D5T<S> rho = ra + rb;
    D5T<S> sig = saa + 2.0 * sab + sbb;
    D5T<S> rm13 = pow5(rho, -1.0 / 3.0);
    D5T<S> den = 1.0 + d * rm13;
    D5T<S> omega = exp5(-c * rm13) / den * pow5(rho, -11.0 / 3.0);
    D5T<S> delta = c * rm13 + d * rm13 / den;
    D5T<S> rab = ra * rb;
    D5T<S> br = rab * (pow(2.0, 11.0 / 3.0) * CF * (pow5(ra, 8.0 / 3.0) + pow5(rb, 8.0 / 3.0))
                   + (47.0 / 18.0 - 7.0 / 18.0 * delta) * sig
                   - (2.5 - delta / 18.0) * (saa + sbb)
                   - (delta - 11.0) / 9.0 * (ra / rho * saa + rb / rho * sbb))
            - (2.0 / 3.0) * rho * rho * sig
            + ((2.0 / 3.0) * rho * rho - ra * ra) * sbb + ((2.0 / 3.0) * rho * rho - rb * rb) * saa;
    return -a * 4.0 / den * rab / rho - a * b * omega * br;
}

// PBE exchange of one spin channel: E_x[rho_a, rho_b] = (E_x[2 rho_a] + E_x[2 rho_b]) / 2 (spin-scaling relation)
template <class S> __device__ inline D5T<S> pbe_x_spin(D5T<S> r, D5T<S> s)
{
    const double kappa = 0.804, mu = 0.2195149727645171;
    const double cx = 0.75 * 0.98474502184269654;     // (3/4)(3/pi)^(1/3)
    D5T<S> rho = 2.0 * r, sigma = 4.0 * s;
    D5T<S> ex_lda = -cx * pow5(rho, 4.0 / 3.0);
    D5T<S> kf = pow5(3.0 * PI * PI * rho, 1.0 / 3.0);
    D5T<S> s2 = sigma / (4.0 * kf * kf * rho * rho);
    D5T<S> fx = 1.0 + kappa - kappa / (1.0 + mu / kappa * s2);
    return 0.5 * ex_lda * fx;
}
// Perdew-Wang 1992 G function, "pw_mod" digits (the variant PBE is built on)
template <class S> __device__ inline D5T<S> pw92_g5(D5T<S> rs, double A, double a1, double b1, double b2, double b3, double b4)
{
    D5T<S> srs = sqrt5(rs);
    D5T<S> q = 2.0 * A * (b1 * srs + b2 * rs + b3 * rs * srs + b4 * rs * rs);
    return -2.0 * A * (1.0 + a1 * rs) * log5(1.0 + 1.0 / q);
}
// PBE correlation for a spin-polarised density (Perdew, Burke, Ernzerhof 1996, eqs. 3-8)
template <class S> __device__ inline D5T<S> pbe_c_pol(D5T<S> rho, D5T<S> zeta, D5T<S> sigma)
{
    const double beta = 0.06672455060314922, gamma = 0.031090690869654895;
    const double fz20 = 1.709920934161365617563962776245;
    D5T<S> rs = pow5(3.0 / (4.0 * PI) / rho, 1.0 / 3.0);
    D5T<S> e0 = pw92_g5(rs, 0.0310907, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    D5T<S> e1 = pw92_g5(rs, 0.01554535, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    D5T<S> mac = pw92_g5(rs, 0.0168869, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);      // -alpha_c
    D5T<S> f = fzeta5(zeta);
    D5T<S> z4 = zeta * zeta * zeta * zeta;
    D5T<S> ec = e0 - mac * f / fz20 * (1.0 - z4) + (e1 - e0) * f * z4;
    D5T<S> p = 1.0 + zeta, m = 1.0 - zeta;
    floor_at(p.v, 1e-14);
    floor_at(m.v, 1e-14);
    D5T<S> phi = 0.5 * (pow5(p, 2.0 / 3.0) + pow5(m, 2.0 / 3.0));
    D5T<S> phi3 = phi * phi * phi;
    D5T<S> kf = pow5(3.0 * PI * PI * rho, 1.0 / 3.0);
    D5T<S> ks2 = 4.0 * kf / PI;
    D5T<S> t2 = sigma / (4.0 * phi * phi * ks2 * rho * rho);
    D5T<S> Aa = beta / gamma / (exp5(-ec / (gamma * phi3)) - 1.0);
    D5T<S> num = 1.0 + Aa * t2;
    D5T<S> H = gamma * phi3 * log5(1.0 + beta / gamma * t2 * num / (1.0 + Aa * t2 + Aa * Aa * t2 * t2));
    return rho * (ec + H);
}

using D5 = D5T<double>;
__device__ inline D5 var5(double v, int k) { D5 r = c5<double>(v); r.d[k] = 1; return r; }

// rho_a / rho_b [4][ldg]; wv_a / wv_b [4][ldg]: wv_s0 = 0.5 w vrho_s, wv_s(1..3) = w (2 vsigma_ss grad rho_s +
// vsigma_ab grad rho_other)  (pyscf/dft/numint.py:1192-1324, xc_deriv.transform_vxc for spin = 1)
// acc[0] += sum w rho_a, acc[1] += sum w rho_b, acc[2] += sum w e
__global__ __launch_bounds__(256) void eval_xc_pol_kernel(XCSpec spec, int gga, const double *__restrict__ rho_a,
                                                          const double *__restrict__ rho_b,
                                                          const double *__restrict__ weights, long ng, long ldg,
                                                          double *__restrict__ wv_a, double *__restrict__ wv_b,
                                                          double *__restrict__ acc, double *__restrict__ evol_out)
{
    long g = (long)blockIdx.x * 256 + threadIdx.x;
    double na = 0, nb = 0, exc = 0;
    if (g < ng) {
        const double w = weights[g];
        double ra = rho_a[g], rb = rho_b[g];
        double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0};
        if (gga)
            for (int x = 0; x < 3; x++) { ga[x] = rho_a[(x + 1) * ldg + g]; gb[x] = rho_b[(x + 1) * ldg + g]; }
        na = w * ra; nb = w * rb;
        double dv[5] = {0, 0, 0, 0, 0};
        if (ra + rb > 1e-14) {
            if (ra < 1e-30) ra = 1e-30;
            if (rb < 1e-30) rb = 1e-30;
            D5 Ra = var5(ra, 0), Rb = var5(rb, 1);
            D5 Saa = var5(ga[0] * ga[0] + ga[1] * ga[1] + ga[2] * ga[2], 2);
            D5 Sab = var5(ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2], 3);
            D5 Sbb = var5(gb[0] * gb[0] + gb[1] * gb[1] + gb[2] * gb[2], 4);
            D5 rho = Ra + Rb;
            D5 zeta = (Ra - Rb) / rho;
            D5 tot = c5<double>(0);
            if (spec.fac[F_SLATER] != 0) tot = tot + spec.fac[F_SLATER] * slater_pol(Ra, Rb);
            if (spec.fac[F_VWN5] != 0) tot = tot + spec.fac[F_VWN5] * vwn5_pol(rho, zeta);
            if (spec.fac[F_VWNRPA] != 0) tot = tot + spec.fac[F_VWNRPA] * vwnrpa_pol(rho, zeta);
            if (spec.fac[F_B88] != 0) tot = tot + spec.fac[F_B88] * (b88_spin(Ra, Saa) + b88_spin(Rb, Sbb));
            if (spec.fac[F_LYP] != 0) tot = tot + spec.fac[F_LYP] * lyp_pol(Ra, Rb, Saa, Sab, Sbb);
            if (spec.fac[F_PBEX] != 0) tot = tot + spec.fac[F_PBEX] * (pbe_x_spin(Ra, Saa) + pbe_x_spin(Rb, Sbb));
            if (spec.fac[F_PBEC] != 0) tot = tot + spec.fac[F_PBEC] * pbe_c_pol(rho, zeta, Saa + 2.0 * Sab + Sbb);
            if (spec.fac[F_ITYH] != 0) tot = tot + spec.fac[F_ITYH] * (ityh_spin(Ra, Saa, spec.omega) + ityh_spin(Rb, Sbb, spec.omega));
            if (spec.fac[F_WB97] != 0) tot = tot + spec.fac[F_WB97] * wb97_xc(Ra, Rb, Saa, Sbb, spec.omega);
            exc = w * tot.v;
            for (int k = 0; k < 5; k++) dv[k] = tot.d[k];
        }
        if (evol_out) evol_out[g] = (w != 0.0) ? exc / w : 0.0;      // energy density per unit volume
        wv_a[g] = 0.5 * w * dv[0];
        wv_b[g] = 0.5 * w * dv[1];
        if (gga)
            for (int x = 0; x < 3; x++) {
                wv_a[(x + 1) * ldg + g] = w * (2.0 * dv[2] * ga[x] + dv[3] * gb[x]);
                wv_b[(x + 1) * ldg + g] = w * (2.0 * dv[4] * gb[x] + dv[3] * ga[x]);
            }
    }
    for (int off = 32; off > 0; off >>= 1) {
        na += __shfl_down(na, off, 64); nb += __shfl_down(nb, off, 64); exc += __shfl_down(exc, off, 64);
    }
    __shared__ double red[3][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = na; red[1][threadIdx.x >> 6] = nb; red[2][threadIdx.x >> 6] = exc; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(acc + threadIdx.x, v);
    }
}

// Spin-polarised response weights (numint.nr_uks_fxc, numint.py:1690-1832, weights of _uks_gga_wv1 :1834-1915): the
// first-order change of (v_rho_a, v_rho_b, v_sigma_aa, v_sigma_ab, v_sigma_bb) along the first-order spin densities,
// from the functionals evaluated on D5T<Eps>;
//   wv1_a[0] = 0.5 w d v_rho_a,  wv1_a[k] = w [2 d v_aa grad_k rho0_a + d v_ab grad_k rho0_b + 2 v_aa grad_k rho1_a + v_ab grad_k rho1_b]
__global__ __launch_bounds__(256) void eval_fxc_pol_kernel(XCSpec spec, int gga, const double *__restrict__ rho0_a,
                                                           const double *__restrict__ rho0_b,
                                                           const double *__restrict__ rho1_a,
                                                           const double *__restrict__ rho1_b,
                                                           const double *__restrict__ weights, long ng, long ldg,
                                                           double *__restrict__ wv_a, double *__restrict__ wv_b)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= ng) return;
    const double w = weights[g];
    double ra = rho0_a[g], rb = rho0_b[g];
    const double ta = rho1_a[g], tb = rho1_b[g];
    double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0}, ha[3] = {0, 0, 0}, hb[3] = {0, 0, 0};
    if (gga)
        for (int x = 0; x < 3; x++) {
            ga[x] = rho0_a[(x + 1) * ldg + g]; gb[x] = rho0_b[(x + 1) * ldg + g];
            ha[x] = rho1_a[(x + 1) * ldg + g]; hb[x] = rho1_b[(x + 1) * ldg + g];
        }
    double dv[5] = {0, 0, 0, 0, 0}, v[5] = {0, 0, 0, 0, 0};
    if (ra + rb > 1e-14) {
        if (ra < 1e-30) ra = 1e-30;
        if (rb < 1e-30) rb = 1e-30;
        const double val[5] = {ra, rb, ga[0] * ga[0] + ga[1] * ga[1] + ga[2] * ga[2],
                               ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2], gb[0] * gb[0] + gb[1] * gb[1] + gb[2] * gb[2]};
        const double dir[5] = {ta, tb, 2 * (ga[0] * ha[0] + ga[1] * ha[1] + ga[2] * ha[2]),
                               ga[0] * hb[0] + ga[1] * hb[1] + ga[2] * hb[2] + gb[0] * ha[0] + gb[1] * ha[1] + gb[2] * ha[2],
                               2 * (gb[0] * hb[0] + gb[1] * hb[1] + gb[2] * hb[2])};
        D5T<Eps> X[5];
        for (int k = 0; k < 5; k++) {
            X[k] = c5<Eps>(val[k]);
            X[k].v.e = dir[k];
            X[k].d[k].v = 1.0;
        }
        D5T<Eps> rho = X[0] + X[1];
        D5T<Eps> zeta = (X[0] - X[1]) / rho;
        D5T<Eps> tot = c5<Eps>(0);
        if (spec.fac[F_SLATER] != 0) tot = tot + spec.fac[F_SLATER] * slater_pol(X[0], X[1]);
        if (spec.fac[F_VWN5] != 0) tot = tot + spec.fac[F_VWN5] * vwn5_pol(rho, zeta);
        if (spec.fac[F_VWNRPA] != 0) tot = tot + spec.fac[F_VWNRPA] * vwnrpa_pol(rho, zeta);
        if (spec.fac[F_B88] != 0) tot = tot + spec.fac[F_B88] * (b88_spin(X[0], X[2]) + b88_spin(X[1], X[4]));
        if (spec.fac[F_LYP] != 0) tot = tot + spec.fac[F_LYP] * lyp_pol(X[0], X[1], X[2], X[3], X[4]);
        if (spec.fac[F_PBEX] != 0) tot = tot + spec.fac[F_PBEX] * (pbe_x_spin(X[0], X[2]) + pbe_x_spin(X[1], X[4]));
        if (spec.fac[F_PBEC] != 0) tot = tot + spec.fac[F_PBEC] * pbe_c_pol(rho, zeta, X[2] + 2.0 * X[3] + X[4]);
        if (spec.fac[F_ITYH] != 0) tot = tot + spec.fac[F_ITYH] * (ityh_spin(X[0], X[2], spec.omega) + ityh_spin(X[1], X[4], spec.omega));
        if (spec.fac[F_WB97] != 0) tot = tot + spec.fac[F_WB97] * wb97_xc(X[0], X[1], X[2], X[4], spec.omega);
        for (int k = 0; k < 5; k++) { v[k] = tot.d[k].v; dv[k] = tot.d[k].e; }
    }
    wv_a[g] = 0.5 * w * dv[0];
    wv_b[g] = 0.5 * w * dv[1];
    if (gga)
        for (int x = 0; x < 3; x++) {
            wv_a[(x + 1) * ldg + g] = w * (2.0 * dv[2] * ga[x] + dv[3] * gb[x] + 2.0 * v[2] * ha[x] + v[3] * hb[x]);
            wv_b[(x + 1) * ldg + g] = w * (2.0 * dv[4] * gb[x] + dv[3] * ga[x] + 2.0 * v[4] * hb[x] + v[3] * ha[x]);
        }
}

// rho from c[comp][i][ldc] = sum_mu C_occ[mu][i] ao_comp[g][mu]   (orbital rows, grid index fastest)
// sign (nullable): +-1 per row - a symmetric matrix D = C diag(sign) C^T (eigen-factorised, not positive) as "orbitals"
__global__ __launch_bounds__(256) void rho_from_mo_kernel(const double *__restrict__ c, long comp_stride, long ldc,
                                                          int nocc, int ncomp, long ng, double *__restrict__ rho,
                                                          long ldg, const double *__restrict__ sign)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= ng) return;
    double s0 = 0, sx = 0, sy = 0, sz = 0;
    for (int i = 0; i < nocc; i++) {
        const double cv = c[(long)i * ldc + g];
        const double c0 = sign ? sign[i] * cv : cv;
        s0 += c0 * cv;
        if (ncomp == 4) {
            sx += c0 * c[comp_stride + (long)i * ldc + g];
            sy += c0 * c[2 * comp_stride + (long)i * ldc + g];
            sz += c0 * c[3 * comp_stride + (long)i * ldc + g];
        }
    }
    rho[g] = s0;
    if (ncomp == 4) { rho[ldg + g] = 2 * sx; rho[2 * ldg + g] = 2 * sy; rho[3 * ldg + g] = 2 * sz; }
}

// first-order density of a factorised matrix D = A B^T: rho = coef sum_i a_i b_i, grad rho = coef sum_i (grad a_i b_i + a_i grad b_i)
// with a = A^T ao, b = B^T ao in the [comp][i][ldc] layout of rho_from_mo (a trial density C_occ x C_vir^T of the response
// solvers has rank nocc: two orbital products replace the nao^2 matrix, numint.py:1418-1530 eval_rho on dm1)
__global__ __launch_bounds__(256) void rho_from_mo_pair_kernel(const double *__restrict__ ca, const double *__restrict__ cb,
                                                               long comp_stride, long ldc, int nocc, int ncomp, long ng,
                                                               double coef, double *__restrict__ rho, long ldg)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= ng) return;
    double s0 = 0, sx = 0, sy = 0, sz = 0;
    for (int i = 0; i < nocc; i++) {
        const double a0 = ca[(long)i * ldc + g], b0 = cb[(long)i * ldc + g];
        s0 += a0 * b0;
        if (ncomp == 4) {
            sx += a0 * cb[comp_stride + (long)i * ldc + g] + ca[comp_stride + (long)i * ldc + g] * b0;
            sy += a0 * cb[2 * comp_stride + (long)i * ldc + g] + ca[2 * comp_stride + (long)i * ldc + g] * b0;
            sz += a0 * cb[3 * comp_stride + (long)i * ldc + g] + ca[3 * comp_stride + (long)i * ldc + g] * b0;
        }
    }
    rho[g] = coef * s0;
    if (ncomp == 4) { rho[ldg + g] = coef * sx; rho[2 * ldg + g] = coef * sy; rho[3 * ldg + g] = coef * sz; }
}

// rho[g] = sum_mu ao0[g][mu] c0t[mu][g], grad = 2 sum_mu ao_x[g][mu] c0t[mu][g]   (c0t = (D ao0^T), [mu][ldc])
// one wave per grid point (general-DM branch; not a hot path)
__global__ __launch_bounds__(256) void rho_from_dm_kernel(const double *__restrict__ ao, const double *__restrict__ c0t,
                                                          int nao, int ldao, long ldg_rows, long ldc, int ncomp,
                                                          long ng, double *__restrict__ rho, long ldg)
{
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ng) return;
    const int lane = threadIdx.x & 63;
    const long cs = ldg_rows * ldao;
    double s0 = 0, sx = 0, sy = 0, sz = 0;
    for (int m = lane; m < nao; m += 64) {
        const double c = c0t[(long)m * ldc + g];
        s0 += ao[g * ldao + m] * c;
        if (ncomp == 4) {
            sx += ao[cs + g * ldao + m] * c;
            sy += ao[2 * cs + g * ldao + m] * c;
            sz += ao[3 * cs + g * ldao + m] * c;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64); sx += __shfl_down(sx, off, 64);
        sy += __shfl_down(sy, off, 64); sz += __shfl_down(sz, off, 64);
    }
    if (lane == 0) {
        rho[g] = s0;
        if (ncomp == 4) { rho[ldg + g] = 2 * sx; rho[2 * ldg + g] = 2 * sy; rho[3 * ldg + g] = 2 * sz; }
    }
}

// aow[g][mu] = sum_c wv[c][g] ao[c][g][mu]   (rows g >= ng are zeroed)
__global__ __launch_bounds__(256) void scale_ao_kernel(const double *__restrict__ ao, const double *__restrict__ wv,
                                                       int ldao, long ldg_rows, long ldg, int ncomp, long ng,
                                                       double *__restrict__ aow)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    const long g = blockIdx.y;
    if (m >= ldao) return;
    double v = 0;
    if (g < ng) {
        const long cs = ldg_rows * ldao;
        v = wv[g] * ao[g * ldao + m];
        if (ncomp == 4)
            v += wv[ldg + g] * ao[cs + g * ldao + m] + wv[2 * ldg + g] * ao[2 * cs + g * ldao + m] +
                 wv[3 * ldg + g] * ao[3 * cs + g * ldao + m];
    }
    aow[g * ldao + m] = v;
}


// XC nuclear-gradient reduction (grid response left out, as pyscf/grad/rks.py:get_vxc with grid_response=False):
//   out[x][mu] += sum_g { d_x ao_mu (wv0 c0_mu + sum_k wv_k ck_mu) + (sum_k wv_k d_x d_k ao_mu) c0_mu }
// with c_k[g][mu] = sum_nu ao_k[g][nu] D[nu][mu]; the caller turns it into -2 sum_{mu on A} out[x][mu]
// (_d1_dot_ + _gga_grad_sum_ + _make_dR_dao_w of pyscf/grad/rks.py:197-255, contracted with D on the fly).
// wv follows PAMD_eval_xc's convention (wv[0] = w vrho / 2), ao holds 4 (LDA) or 10 (GGA) components.
__global__ __launch_bounds__(256) void xc_grad_kernel(const double *__restrict__ ao, const double *__restrict__ c,
                                                      const double *__restrict__ wv, int ldao, long ldg_rows,
                                                      long ldg, int gga, long ng, int nao, long rows_per_block,
                                                      double *__restrict__ out)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= nao) return;
    const long g0 = (long)blockIdx.y * rows_per_block;
    const long g1 = (g0 + rows_per_block < ng) ? g0 + rows_per_block : ng;
    const long cs = ldg_rows * ldao;
    double sx = 0, sy = 0, sz = 0;
    // Hessian component (x,k): xx xy xz / xy yy yz / xz yz zz = components 4 5 6 / 5 7 8 / 6 8 9
    for (long g = g0; g < g1; g++) {
        const long o = g * ldao + m;
        const double w0 = 2 * wv[g];
        const double c0 = c[o];
        const double ax = ao[cs + o], ay = ao[2 * cs + o], az = ao[3 * cs + o];
        double t = w0 * c0;
        if (gga) {
            const double w1 = wv[ldg + g], w2 = wv[2 * ldg + g], w3 = wv[3 * ldg + g];
            t += w1 * c[cs + o] + w2 * c[2 * cs + o] + w3 * c[3 * cs + o];
            const double hxx = ao[4 * cs + o], hxy = ao[5 * cs + o], hxz = ao[6 * cs + o];
            const double hyy = ao[7 * cs + o], hyz = ao[8 * cs + o], hzz = ao[9 * cs + o];
            sx += (w1 * hxx + w2 * hxy + w3 * hxz) * c0;
            sy += (w1 * hxy + w2 * hyy + w3 * hyz) * c0;
            sz += (w1 * hxz + w2 * hyz + w3 * hzz) * c0;
        }
        sx += ax * t; sy += ay * t; sz += az * t;
    }
    atomicAdd(out + m, sx);
    atomicAdd(out + nao + m, sy);
    atomicAdd(out + 2 * nao + m, sz);
}

// Per-point row sums of the same integrand: rows[x][g] = sum_mu { d_x ao_mu (w0 c0 + sum_k w_k c_k) + (sum_k w_k d_x d_k ao_mu) c0 }
// (the motion of a grid point with its owner atom, pyscf/grad/rks.py:303-318: excsum[atom] += 2 vtmp . dm on the
// atom's own points).  One workgroup per point and 256 AO columns.
__global__ __launch_bounds__(256) void xc_grad_rows_kernel(const double *__restrict__ ao, const double *__restrict__ c,
                                                           const double *__restrict__ wv, int ldao, long ldg_rows,
                                                           long ldg, int gga, long ng, int nao, double *__restrict__ rows)
{
    const int m = blockIdx.y * 256 + threadIdx.x;
    const long g = blockIdx.x;                           // grid.x: up to 2^31 points per block
    double s[3] = {0, 0, 0};
    if (m < nao) {
        const long cs = ldg_rows * ldao;
        const long o = g * ldao + m;
        const double w0 = 2 * wv[g];
        const double c0 = c[o];
        double t = w0 * c0;
        if (gga) {
            const double w1 = wv[ldg + g], w2 = wv[2 * ldg + g], w3 = wv[3 * ldg + g];
            t += w1 * c[cs + o] + w2 * c[2 * cs + o] + w3 * c[3 * cs + o];
            s[0] = (w1 * ao[4 * cs + o] + w2 * ao[5 * cs + o] + w3 * ao[6 * cs + o]) * c0;
            s[1] = (w1 * ao[5 * cs + o] + w2 * ao[7 * cs + o] + w3 * ao[8 * cs + o]) * c0;
            s[2] = (w1 * ao[6 * cs + o] + w2 * ao[8 * cs + o] + w3 * ao[9 * cs + o]) * c0;
        }
        s[0] += ao[cs + o] * t; s[1] += ao[2 * cs + o] * t; s[2] += ao[3 * cs + o] * t;
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, 64);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; k++) atomicAdd(rows + k * ldg + g, s[k]);
}

// C[m][n] += sum_k A[m][k] B[n][k]   (both operands k-contiguous), split-K over gridDim.y
constexpr int KB = 16, NT = 128, LDT = KB + 1;
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const double *__restrict__ A, long lda,
                                                         const double *__restrict__ B, long ldb,
                                                         double *__restrict__ C, int ldc, int m, int n, long kdim,
                                                         int ntile_n)
{
    __shared__ double sA[NT * LDT];
    __shared__ double sB[NT * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / ntile_n, tn = blockIdx.x - tm * ntile_n;
    const int p0 = tm * NT, q0 = tn * NT;
    const int nsplit = gridDim.y;
    const long kchunk = ((kdim + nsplit - 1) / nsplit + KB - 1) / KB * KB;
    const long kbeg = (long)blockIdx.y * kchunk;
    const long kend = (kbeg + kchunk < kdim) ? kbeg + kchunk : kdim;
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int row = tid >> 1, kq = (tid & 1) * 8;       // each thread stages 8 consecutive k of one row
    double pa[8], pb[8];
    auto fetch = [&](long k0) {
        const double *ga = A + (long)(p0 + row) * lda + k0 + kq;
        const double *gb = B + (long)(q0 + row) * ldb + k0 + kq;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            pa[j] = (p0 + row < m && k0 + kq + j < kend) ? ga[j] : 0.0;
            pb[j] = (q0 + row < n && k0 + kq + j < kend) ? gb[j] : 0.0;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (long k0 = kbeg; k0 < kend; k0 += KB) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            sA[row * LDT + kq + j] = pa[j];
            sB[row * LDT + kq + j] = pb[j];
        }
        __syncthreads();
        if (k0 + KB < kend) fetch(k0 + KB);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = sA[(wr * 64 + a * 16 + fn) * LDT + kk + fk];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = sB[(wc * 64 + b * 16 + fn) * LDT + kk + fk];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
        __syncthreads();
    }
    double *out = C + (long)blockIdx.y * m * ldc;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int col = q0 + wc * 64 + b * 16 + fn;
            if (col >= n) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int rowi = p0 + wr * 64 + a * 16 + fk + 4 * r;
                if (rowi < m) out[(long)rowi * ldc + col] += acc[a][b][r];
            }
        }
}

// out[i][j] = sum_s (part[s][i][j] + part[s][j][i])     (vmat = M + M^T, numint.py:1157)
__global__ void reduce_sym_kernel(const double *__restrict__ part, int nsplit, int m, int ldc,
                                  double *__restrict__ out)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= m) return;
    double v = 0;
    for (int s = 0; s < nsplit; s++)
        v += part[((long)s * m + i) * ldc + j] + part[((long)s * m + j) * ldc + i];
    out[(long)i * m + j] = v;
}

}  // namespace

extern "C" {

// fac[10]: weights of {Slater, VWN5, VWN_RPA, B88, LYP, PBE_X, PBE_C, ITYH (short-range B88), WB97 (whole xc)}, then omega of ITYH / WB97; gga = 1 if any GGA term.
// d_acc[0] += sum w rho (nelec), d_acc[1] += sum w e_xc.  d_exc (nullable): e_xc per particle.
int PAMD_eval_xc(const double *fac, int gga, const double *d_rho, const double *d_weights, long ng, long ldg,
                 double *d_wv, double *d_exc, double *d_acc, void *stream)
{
    if (ng == 0) return 0;
    XCSpec spec;
    for (int i = 0; i < F_NUM; i++) spec.fac[i] = fac[i];
    spec.omega = fac[F_NUM];
    eval_xc_kernel<<<ceil_div(ng, 256), 256, 0, (hipStream_t)stream>>>(spec, gga, d_rho, d_weights, ng, ldg, d_wv,
                                                                       d_exc, d_acc);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// Response kernel: d_wv1[4][ldg] from the zeroth- and first-order densities d_rho0 / d_rho1 [4][ldg] (rho, grad rho)
// of grid points [0, ng); same fac / gga as PAMD_eval_xc.  numint.nr_rks_fxc (dft/numint.py:1418-1530).
int PAMD_eval_fxc(const double *fac, int gga, const double *d_rho0, const double *d_rho1, const double *d_weights,
                  long ng, long ldg, double *d_wv1, void *stream)
{
    if (ng == 0) return 0;
    XCSpec spec;
    for (int i = 0; i < F_NUM; i++) spec.fac[i] = fac[i];
    spec.omega = fac[F_NUM];
    eval_fxc_kernel<<<ceil_div(ng, 256), 256, 0, (hipStream_t)stream>>>(spec, gga, d_rho0, d_rho1, d_weights, ng, ldg,
                                                                        d_wv1);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// Spin-polarised response kernel: first-order weights d_wv1_a / d_wv1_b [4][ldg] from the zeroth-order (d_rho0_a/b) and
// first-order (d_rho1_a/b) spin densities [4][ldg].  numint.nr_uks_fxc (dft/numint.py:1690-1915).
int PAMD_eval_fxc_pol(const double *fac, int gga, const double *d_rho0_a, const double *d_rho0_b, const double *d_rho1_a,
                      const double *d_rho1_b, const double *d_weights, long ng, long ldg, double *d_wv1_a,
                      double *d_wv1_b, void *stream)
{
    if (ng == 0) return 0;
    XCSpec spec;
    for (int i = 0; i < F_NUM; i++) spec.fac[i] = fac[i];
    spec.omega = fac[F_NUM];
    eval_fxc_pol_kernel<<<ceil_div(ng, 256), 256, 0, (hipStream_t)stream>>>(spec, gga, d_rho0_a, d_rho0_b, d_rho1_a,
                                                                            d_rho1_b, d_weights, ng, ldg, d_wv1_a, d_wv1_b);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// spin-polarised variant (numint.nr_uks)
// d_evol (nullable) [ng]: XC energy density per unit volume (for the grid-response term of the gradient)
int PAMD_eval_xc_pol(const double *fac, int gga, const double *d_rho_a, const double *d_rho_b,
                     const double *d_weights, long ng, long ldg, double *d_wv_a, double *d_wv_b, double *d_acc3,
                     double *d_evol, void *stream)
{
    if (ng == 0) return 0;
    XCSpec spec;
    for (int i = 0; i < F_NUM; i++) spec.fac[i] = fac[i];
    spec.omega = fac[F_NUM];
    eval_xc_pol_kernel<<<ceil_div(ng, 256), 256, 0, (hipStream_t)stream>>>(spec, gga, d_rho_a, d_rho_b, d_weights,
                                                                           ng, ldg, d_wv_a, d_wv_b, d_acc3, d_evol);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_rho_from_mo(const double *d_c, long comp_stride, long ldc, int nocc, int ncomp, long ng, double *d_rho,
                     long ldg, const double *d_occ_sign, void *stream)
{
    if (ng == 0) return 0;
    rho_from_mo_kernel<<<ceil_div(ng, 256), 256, 0, (hipStream_t)stream>>>(d_c, comp_stride, ldc, nocc, ncomp, ng,
                                                                           d_rho, ldg, d_occ_sign);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_rho_from_mo_pair(const double *d_ca, const double *d_cb, long comp_stride, long ldc, int nocc, int ncomp, long ng,
                          double coef, double *d_rho, long ldg, void *stream)
{
    if (ng == 0) return 0;
    rho_from_mo_pair_kernel<<<ceil_div(ng, 256), 256, 0, (hipStream_t)stream>>>(d_ca, d_cb, comp_stride, ldc, nocc, ncomp,
                                                                                ng, coef, d_rho, ldg);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_rho_from_dm(const double *d_ao, const double *d_c0t, int nao, int ldao, long ldg_rows, long ldc, int ncomp,
                     long ng, double *d_rho, long ldg, void *stream)
{
    if (ng == 0) return 0;
    rho_from_dm_kernel<<<ceil_div(ng, 4), 256, 0, (hipStream_t)stream>>>(d_ao, d_c0t, nao, ldao, ldg_rows, ldc, ncomp,
                                                                         ng, d_rho, ldg);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// aow[g][ldao] for g < nrows (rows ng..nrows-1 zeroed so that padded k ranges contribute nothing)
int PAMD_scale_ao(const double *d_ao, const double *d_wv, int ldao, long ldg_rows, long ldg, int ncomp, long ng,
                  long nrows, double *d_aow, void *stream)
{
    if (nrows == 0) return 0;
    dim3 grid(ceil_div(ldao, 256), nrows);
    scale_ao_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_ao, d_wv, ldao, ldg_rows, ldg, ncomp, ng, d_aow);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// d_out[3][nao] += per-AO XC gradient sums of one grid block; d_c[ncomp_c][ldg_rows][ldao] (1 or 4 components,
// same strides as d_ao), d_ao[4 or 10][ldg_rows][ldao]
int PAMD_xc_grad(const double *d_ao, const double *d_c, const double *d_wv, int ldao, long ldg_rows, long ldg,
                 int gga, long ng, int nao, double *d_out, void *stream)
{
    if (ng == 0 || nao == 0) return 0;
    const long rows = 512;
    dim3 grid(ceil_div(nao, 256), ceil_div(ng, rows));
    xc_grad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_ao, d_c, d_wv, ldao, ldg_rows, ldg, gga, ng, nao, rows, d_out);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// d_rows[3][ldg] += per-point sums over the AO index of the XC gradient integrand (grid-response term of the points'
// own motion); same operands as PAMD_xc_grad
int PAMD_xc_grad_rows(const double *d_ao, const double *d_c, const double *d_wv, int ldao, long ldg_rows, long ldg,
                      int gga, long ng, int nao, double *d_rows, void *stream)
{
    if (ng == 0 || nao == 0) return 0;
    dim3 grid(ng, ceil_div(nao, 256));
    xc_grad_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_ao, d_c, d_wv, ldao, ldg_rows, ldg, gga, ng, nao, d_rows);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// C[s][m][ldc] += A[m][k] B[n][k]^T over the s-th k range
int PAMD_dgemm_nt(const double *d_A, long lda, const double *d_B, long ldb, double *d_C, int ldc, int m, int n,
                  long k, int nsplit, void *stream)
{
    if (m == 0 || n == 0 || k == 0) return 0;
    int tm = ceil_div(m, NT), tn = ceil_div(n, NT);
    dim3 grid(tm * tn, nsplit);
    gemm_nt_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_A, lda, d_B, ldb, d_C, ldc, m, n, k, tn);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_reduce_sym(const double *d_part, int nsplit, int m, int ldc, double *d_out, void *stream)
{
    dim3 grid(ceil_div(m, 256), m);
    reduce_sym_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_part, nsplit, m, ldc, d_out);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
