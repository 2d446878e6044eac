// Host-side tables shared by the opaque-handle entry points (df_handle.hip, xc_handle.hip): libcint-format bas / atm / env ->
// segmented shells, cart -> sph matrices, a pool of device allocations.  (pyscf_amd/gto/moleintor.py does the same in numpy.)
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "common.h"

namespace pamd {
namespace host {

constexpr int ATOM_OF = 0, ANG_OF = 1, NPRIM_OF = 2, NCTR_OF = 3, PTR_EXP = 5, PTR_COEFF = 6, BAS_SLOTS = 8;
constexpr int PTR_COORD = 1, ATM_SLOTS = 6;
constexpr double EXPCUTOFF = 60.0;          // primitive-pair screening (pyscf_amd/gto/moleintor.py)
constexpr int LMAX_TAB = 6;

inline long round_up(long x, long m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------ cart -> sph
inline double binom(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    double r = 1;
    for (int i = 1; i <= k; i++) r = r * (n - k + i) / i;
    return r;
}
inline double fact(int n) { double r = 1; for (int i = 2; i <= n; i++) r *= i; return r; }

// real solid harmonics in Cartesian monomials (Helgaker/Jorgensen/Olsen 6.4.47), ordering of pyscf/lib/parameters.py:69-77
inline std::vector<double> c2s_matrix(int l)
{
    std::vector<int> cx, cy, cz;
    for (int x = l; x >= 0; x--)
        for (int y = l - x; y >= 0; y--) { cx.push_back(x); cy.push_back(y); cz.push_back(l - x - y); }
    const int nc = (int)cx.size();
    auto idx = [&](int lx, int ly, int lz) {
        for (int i = 0; i < nc; i++) if (cx[i] == lx && cy[i] == ly && cz[i] == lz) return i;
        return -1;
    };
    std::vector<double> out((2 * l + 1) * nc, 0.0);
    for (int m = -l; m <= l; m++) {
        const int am = std::abs(m);
        double N = 1.0 / (std::pow(2.0, am) * fact(l)) * std::sqrt(2.0 * fact(l + am) * fact(l - am) / (m == 0 ? 2.0 : 1.0));
        N *= std::sqrt((2 * l + 1) / (4 * M_PI));
        int row = m + l;
        if (l == 1) row = (m == 1) ? 0 : (m == -1 ? 1 : 2);
        for (int t = 0; t <= (l - am) / 2; t++)
            for (int u = 0; u <= t; u++) {
                const int kmax = (m >= 0) ? am / 2 : (am - 1) / 2;
                for (int k = 0; k <= kmax; k++) {
                    const int twov = (m >= 0) ? 2 * k : 2 * k + 1;
                    const double c = (((t + k) % 2) ? -1.0 : 1.0) * std::pow(0.25, t) * binom(l, t) * binom(l - t, am + t) *
                                     binom(t, u) * binom(am, twov);
                    const int ly = 2 * u + twov, lx = 2 * t + am - ly, lz = l - 2 * t - am;
                    if (lx < 0 || lz < 0) continue;
                    out[row * nc + idx(lx, ly, lz)] += N * c;
                }
            }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ shell tables
struct Shells {                         // segmented shells of a bas table (general contractions split, zero coefficients dropped)
    std::vector<int> l, ao0, atom;
    std::vector<double> xyz;            // [n][3]
    std::vector<std::vector<double>> exps, coefs;
    int nao = 0, n = 0;
};

inline Shells make_shells(const int *atm, const int *bas, int b0, int b1, const double *env)
{
    Shells s;
    int off = 0;
    for (int ib = b0; ib < b1; ib++) {
        const int *b = bas + (long)ib * BAS_SLOTS;
        const int ia = b[ATOM_OF], ll = b[ANG_OF], nprim = b[NPRIM_OF], nctr = b[NCTR_OF];
        const double *r = env + atm[ia * ATM_SLOTS + PTR_COORD];
        const double *e = env + b[PTR_EXP], *c = env + b[PTR_COEFF];
        for (int k = 0; k < nctr; k++) {
            std::vector<double> ee, cc;
            for (int p = 0; p < nprim; p++)
                if (c[k * nprim + p] != 0.0) { ee.push_back(e[p]); cc.push_back(c[k * nprim + p]); }
            s.l.push_back(ll);
            s.xyz.insert(s.xyz.end(), r, r + 3);
            s.exps.push_back(ee);
            s.coefs.push_back(cc);
            s.ao0.push_back(off);
            s.atom.push_back(ia);
            off += 2 * ll + 1;
        }
    }
    s.nao = off;
    s.n = (int)s.l.size();
    return s;
}

struct DevPool {                        // every device allocation of a handle, freed together
    std::vector<void *> ptrs;
    int alloc(void **p, size_t bytes)
    {
        *p = nullptr;
        if (bytes == 0) bytes = 8;
        PAMD_CHECK_HIP(hipMalloc(p, bytes));
        ptrs.push_back(*p);
        return 0;
    }
    void release(void *p)
    {
        auto it = std::find(ptrs.begin(), ptrs.end(), p);
        if (it != ptrs.end()) { (void)hipFree(p); ptrs.erase(it); }
    }
    ~DevPool() { for (void *p : ptrs) (void)hipFree(p); }
};

template <class T>
inline int upload(DevPool &pool, const std::vector<T> &h, T **d)
{
    int rc = pool.alloc((void **)d, h.size() * sizeof(T));
    if (rc) return rc;
    if (!h.empty()) PAMD_CHECK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}


}  // namespace host
}  // namespace pamd
