"""ORACLE (DFT part) - TEST INFRASTRUCTURE ONLY.

CPU restatement of the XC grid path:
* Becke partition            <- pyscf/lib/dft/grid_basis.c:32-101 (VXCgen_grid) in numpy
* AO values on a grid        <- pyscf/lib/gto/grid_ao_drv.c:222-284, deriv1.c (GTOval_sph_deriv0/1) in numpy
* nr_rks                     <- pyscf/dft/numint.py:1074-1190 (dense path)
* XC functionals: the reference calls libxc 7.1.2 (not in tree, pyscf/lib/CMakeLists.txt:229-250).
  Here the published energy expressions (Slater; VWN5 / VWN-RPA; B88; LYP) are written in sympy and
  differentiated SYMBOLICALLY - independent of the forward-mode AD used by the HIP kernel.
* RKS energy                 <- pyscf/dft/rks.py:37-142,228-258

Parity status: PINNED through the reference's SCF energies (tests/test_oracle_dft_golden.py):
LDA,VWN_RPA -76.01330948329084; B88,VWN -76.690247578608236; B3LYPG -76.384928891413438
(pyscf/dft/test/test_h2o.py:95-115), DF B88,VWN -76.690346887915879 (:236-240), grid norms
(pyscf/dft/test/test_grids.py:54-65).  Pointwise libxc values themselves are not available here.
PBE: pinned (to the 2e-5 the reference itself uses) by the DF-RKS energy -75.2497029684 of H2O + ghost:H
(pyscf/dft/test/test_h2o.py:721-784).
"""
import ctypes

import numpy as np

from . import ref

# ----------------------------------------------------------------------------- grids


def _lko_saturate(r):
    """grid_basis.c:236-247: R_c (1 - exp(-sum_{m=1..12} (r/R_c)^m / m)), R_c = 5."""
    x = r / 5.0
    return 5.0 * (1 - np.exp(-sum(x ** m / m for m in range(1, 13))))


def becke_partition(coords, atm_coords, radii_table, scheme='becke'):
    """pbecke[natm][ngrids]: original Becke cell functions (grid_basis.c:32-101), 'stratmann' (gen_grid.py:203-212 in the
    generic loop :388-404) or 'lko' (grid_basis.c:266-384)."""
    natm = len(atm_coords)
    d = np.linalg.norm(coords[None, :, :] - atm_coords[:, None, :], axis=2)   # [natm][ng]
    pb = np.ones((natm, len(coords)))
    for i in range(natm):
        for j in range(i):
            rij = np.linalg.norm(atm_coords[i] - atm_coords[j])
            g = (d[i] - d[j]) / (_lko_saturate(rij) if scheme == 'lko' else rij)
            if scheme == 'lko':
                g = np.clip(g, -1, 1)
            if radii_table is not None:
                g = g + radii_table[i, j] * (1 - g * g)
            if scheme == 'stratmann':
                ma = g / .64
                ma2 = ma * ma
                s = (1 / 16.) * (ma * (35 + ma2 * (-35 + ma2 * (21 - 5 * ma2))))
                s[g <= -.64] = -1
                s[g >= .64] = 1
                s = .5 * s
            else:
                s = g
                s = (3 - s * s) * s * .5
                s = (3 - s * s) * s * .5
                s = ((3 - s * s) * s * .5) * .5
            pb[i] *= .5 - s
            pb[j] *= .5 + s
    return pb


def _lko_saturate_deriv(r):
    """d/dr of _lko_saturate (grid_basis.c:249-264)."""
    x = r / 5.0
    tot = sum(x ** m / m for m in range(1, 13))
    return np.exp(-tot) * sum(x ** (m - 1) for m in range(1, 13))


def _cell_function(nu, scheme):
    """h(nu) in [-1, 1] and h'(nu): P_B = prod_D (1 - h(nu_BD)) / 2."""
    if scheme == 'stratmann':
        ma = nu / .64
        ma2 = ma * ma
        h = (1 / 16.) * (ma * (35 + ma2 * (-35 + ma2 * (21 - 5 * ma2))))
        dh = (35 / 16.) * (1 - ma2) ** 3 / .64
        out = np.abs(nu) >= .64
        return np.where(out, np.sign(nu), h), np.where(out, 0.0, dh)
    p1 = (3 - nu * nu) * nu * .5
    p2 = (3 - p1 * p1) * p1 * .5
    p3 = (3 - p2 * p2) * p2 * .5
    return p3, 1.5 * (1 - p2 * p2) * 1.5 * (1 - p1 * p1) * 1.5 * (1 - nu * nu)


def becke_weight_response(coords, owner, weights, atm_coords, radii_table, scheme='becke'):
    """dw[natm][3][ngrids] = d w_g / d R_C of the Becke quadrature weights, the grid point moving rigidly with its
    owner atom (restatement of pyscf/grad/rks.py:grids_response_cc / get_vxc_full_response weight1; formulas of
    Johnson, Gill, Pople, JCP 98, 5612).  For C != owner the point is fixed:
        d ln P_B / dR_C = sum_{D != B} g_BD d mu_BD/dR_C,   g_BD = -p3'(nu_BD) (1 - 2 a_BD mu_BD) / (2 f_BD),
        d w / dR_C = w [ d ln P_owner/dR_C - sum_B (P_B / Z) d ln P_B/dR_C ];
    the owner's own derivative follows from translational invariance.  scheme 'stratmann' swaps the cell polynomial,
    'lko' divides by the saturated distance S(R_BD) (so mu also moves through S'(R)) and clamps mu to [-1, 1]
    (grids_response_lko, VXCgen_grid_lko_deriv grid_basis.c:386-560)."""
    natm = len(atm_coords)
    ng = len(coords)
    vec = coords[None, :, :] - atm_coords[:, None, :]                          # [natm][ng][3]
    d = np.linalg.norm(vec, axis=2) + 1e-200
    uhat = vec / d[:, :, None]
    f = np.ones((natm, natm, ng))
    gfac = np.zeros((natm, natm, ng))
    mu = np.zeros((natm, natm, ng))
    sdist = np.ones((natm, natm))               # S(R_BD): the distance mu is scaled by
    sfac = np.zeros((natm, natm))               # S'(R_BD) / S(R_BD)
    for b in range(natm):
        for dd in range(natm):
            if b == dd:
                continue
            rbd = np.linalg.norm(atm_coords[b] - atm_coords[dd])
            sdist[b, dd] = _lko_saturate(rbd) if scheme == 'lko' else rbd
            sfac[b, dd] = (_lko_saturate_deriv(rbd) if scheme == 'lko' else 1.0) / sdist[b, dd]
            m = (d[b] - d[dd]) / sdist[b, dd]
            inside = 1.0
            if scheme == 'lko':
                inside = (np.abs(m) < 1).astype(float)
                m = np.clip(m, -1, 1)
            a = 0.0 if radii_table is None else radii_table[b, dd]
            h, dh = _cell_function(m + a * (1 - m * m), scheme)
            f[b, dd] = .5 * (1 - h)
            gfac[b, dd] = -.5 * dh * (1 - 2 * a * m) / (f[b, dd] + 1e-200) * inside
            mu[b, dd] = m
    P = np.prod(f, axis=1)                                                      # [natm][ng]
    Z = P.sum(axis=0)
    dw = np.zeros((natm, 3, ng))
    for c in range(natm):
        dlnP = np.zeros((natm, ng, 3))
        for b in range(natm):
            if b == c:
                for dd in range(natm):
                    if dd == c:
                        continue
                    n_cd = atm_coords[c] - atm_coords[dd]
                    rcd = np.linalg.norm(n_cd)
                    dmu = -uhat[c] / sdist[c, dd] - mu[c, dd][:, None] * (n_cd / rcd)[None, :] * sfac[c, dd]   # d mu_CD / dR_C
                    dlnP[c] += gfac[c, dd][:, None] * dmu
            else:
                n_bc = atm_coords[b] - atm_coords[c]
                rbc = np.linalg.norm(n_bc)
                dmu = uhat[c] / sdist[b, c] + mu[b, c][:, None] * (n_bc / rbc)[None, :] * sfac[b, c]         # d mu_BC / dR_C
                dlnP[b] = gfac[b, c][:, None] * dmu
        avg = np.einsum('bg,bgx->gx', P / Z, dlnP)
        own = dlnP[owner, np.arange(ng)]
        dw[c] = (weights[:, None] * (own - avg)).T
    fixed = dw.copy()
    for a in range(natm):
        sel = owner == a
        dw[a][:, sel] = -(fixed.sum(axis=0) - fixed[a])[:, sel]
    return dw


def build_grids(mol, atom_grid=None, radi_method=None, prune='nwchem', radii_adjust='treutler', level=3,
                sort_grids=True, alignment=8, atomic_radii=None, scheme='becke'):
    """coords, weights (host).  Atomic (radial x Lebedev) tables come from the host-side generator
    (pure numpy restatement of gen_grid.gen_atomic_grids, pinned by the grid-norm goldens); the
    partition is done here in numpy."""
    from pyscf_amd.dft import gen_grid, radi
    prune_fn = {'nwchem': gen_grid.nwchem_prune, 'treutler': gen_grid.treutler_prune, 'sg1': gen_grid.sg1_prune,
                None: None}[prune]
    tab = gen_grid.gen_atomic_grids(mol, atom_grid or {}, radi_method or radi.treutler, level, prune_fn)
    if atomic_radii is None:
        atomic_radii = radi.BRAGG_RADII
    table = None
    if radii_adjust == 'treutler':
        table = radi.treutler_atomic_radii_adjust(mol, atomic_radii)
    elif radii_adjust == 'becke':
        table = radi.becke_atomic_radii_adjust(mol, atomic_radii)
    atm = mol.atom_coords()
    cs, ws = [], []
    for ia in range(mol.natm):
        c, vol = tab[mol.atom_symbol(ia)]
        c = c + atm[ia]
        pb = becke_partition(c, atm, table, scheme)
        cs.append(c)
        ws.append(vol * pb[ia] / pb.sum(axis=0))
    coords, weights = np.vstack(cs), np.hstack(ws)
    if sort_grids:
        idx = gen_grid.arg_group_grids(mol, coords)
        coords, weights = coords[idx], weights[idx]
    if alignment > 1:
        pad = (-len(weights)) % alignment
        if pad:
            coords = np.vstack([coords, np.repeat([[1e-4] * 3], pad, axis=0)])
            weights = np.hstack([weights, np.zeros(pad)])
    return coords, weights


# ----------------------------------------------------------------------------- AO values
def _cart_list(l):
    return [(x, y, l - x - y) for x in range(l, -1, -1) for y in range(l - x, -1, -1)]


def _c2s(l):
    nc = (l + 1) * (l + 2) // 2
    m = np.zeros((2 * l + 1, nc))
    ref.lib().oracle_c2s_matrix(ctypes.c_int(l), m.ctypes.data_as(ctypes.c_void_p))
    return m


def eval_ao(mol, coords, deriv=0):
    """(ngrids, nao) or (4, ngrids, nao) like numint.eval_ao (numint.py:51-114)."""
    ng = len(coords)
    nao = mol.nao_nr()
    out = np.zeros((4 if deriv else 1, ng, nao))
    loc = mol.ao_loc_nr()
    for ib in range(mol.nbas):
        l = mol.bas_angular(ib)
        r = coords - mol.atom_coords()[mol.bas_atom(ib)]
        r2 = np.einsum('ij,ij->i', r, r)
        es = mol.bas_exp(ib)
        cs = mol.bas_ctr_coeff_raw(ib)          # (nprim, nctr), normalised
        ex = np.exp(-np.outer(r2, es))          # (ng, nprim)
        rad = ex.dot(cs)                        # (ng, nctr)
        rad1 = (ex * (-2 * es)).dot(cs)
        carts = _cart_list(l)
        x, y, z = r[:, 0], r[:, 1], r[:, 2]
        poly = np.array([x ** a * y ** b * z ** c for a, b, c in carts])          # (nc, ng)
        c2s = _c2s(l)
        nctr = cs.shape[1]
        for k in range(nctr):
            p0 = loc[ib] + k * (2 * l + 1)
            out[0, :, p0:p0 + 2 * l + 1] = (c2s.dot(poly) * rad[:, k]).T
            if deriv:
                for d in range(3):
                    dp = []
                    for (a, b, c) in carts:
                        e = [a, b, c]
                        t = np.array([x, y, z][d]) * (x ** a * y ** b * z ** c) * rad1[:, k]
                        if e[d] > 0:
                            e2 = list(e)
                            e2[d] -= 1
                            t = t + e[d] * (x ** e2[0] * y ** e2[1] * z ** e2[2]) * rad[:, k]
                        dp.append(t)
                    out[1 + d, :, p0:p0 + 2 * l + 1] = c2s.dot(np.array(dp)).T
    return out[0] if deriv == 0 else out


# ----------------------------------------------------------------------------- functionals (sympy)
_FUNCS = None


def _build_functionals():
    import sympy as sp
    rho, sigma = sp.symbols('rho sigma', positive=True)
    pi = sp.pi
    out = {}
    out['slater'] = -sp.Rational(3, 4) * (3 / pi) ** sp.Rational(1, 3) * rho ** sp.Rational(4, 3)

    def vwn(A, x0, b, c):
        rs = (3 / (4 * pi * rho)) ** sp.Rational(1, 3)
        x = sp.sqrt(rs)
        Q = sp.sqrt(4 * c - b * b)
        X = x * x + b * x + c
        X0 = x0 * x0 + b * x0 + c
        at = sp.atan(Q / (2 * x + b))
        eps = A * (sp.log(x * x / X) + 2 * b / Q * at -
                   b * x0 / X0 * (sp.log((x - x0) ** 2 / X) + 2 * (b + 2 * x0) / Q * at))
        return rho * eps
    out['vwn5'] = vwn(sp.Float('0.0310907', 20), sp.Float('-0.10498', 20), sp.Float('3.72744', 20),
                      sp.Float('12.9352', 20))
    out['vwnrpa'] = vwn(sp.Float('0.0310907', 20), sp.Float('-0.409286', 20), sp.Float('13.0720', 20),
                        sp.Float('42.7198', 20))
    # B88, closed shell
    beta = sp.Float('0.0042', 20)
    cx = sp.Rational(3, 2) * (3 / (4 * pi)) ** sp.Rational(1, 3)
    rs_ = rho / 2
    x = sp.sqrt(sigma / 4) / rs_ ** sp.Rational(4, 3)
    out['b88'] = 2 * (-cx * rs_ ** sp.Rational(4, 3) -
                      beta * rs_ ** sp.Rational(4, 3) * x * x / (1 + 6 * beta * x * sp.asinh(x)))
    # LYP, closed shell (Miehlich, Savin, Stoll, Preuss, CPL 157, 200 (1989) eq. 2)
    a, b, c, d = [sp.Float(v, 20) for v in ('0.04918', '0.132', '0.2533', '0.349')]
    CF = sp.Rational(3, 10) * (3 * pi ** 2) ** sp.Rational(2, 3)
    ra = rho / 2
    gaa = sigma / 4
    rm13 = rho ** sp.Rational(-1, 3)
    den = 1 + d * rm13
    omega = sp.exp(-c * rm13) / den * rho ** sp.Rational(-11, 3)
    delta = c * rm13 + d * rm13 / den
    t1 = -a * 4 / den * ra * ra / rho
    br = (ra * ra * (2 ** sp.Rational(11, 3) * CF * 2 * ra ** sp.Rational(8, 3)
                     + (sp.Rational(47, 18) - sp.Rational(7, 18) * delta) * sigma
                     - (sp.Rational(5, 2) - delta / 18) * 2 * gaa
                     - (delta - 11) / 9 * gaa)
          - sp.Rational(2, 3) * rho ** 2 * sigma + 2 * (sp.Rational(2, 3) * rho ** 2 - ra ** 2) * gaa)
    out['lyp'] = t1 - a * b * omega * br
    # PBE (pinned through one reference energy, 2e-5)
    kappa, mu = sp.Float('0.804', 20), sp.Float('0.2195149727645171', 20)
    kf = (3 * pi ** 2 * rho) ** sp.Rational(1, 3)
    s2 = sigma / (4 * kf ** 2 * rho ** 2)
    out['pbex'] = out['slater'] * (1 + kappa - kappa / (1 + mu / kappa * s2))
    A, a1, b1, b2, b3, b4 = [sp.Float(v, 20) for v in ('0.0310907', '0.21370', '7.5957', '3.5876', '1.6382', '0.49294')]
    rs = (3 / (4 * pi * rho)) ** sp.Rational(1, 3)
    ec = -2 * A * (1 + a1 * rs) * sp.log(1 + 1 / (2 * A * (b1 * sp.sqrt(rs) + b2 * rs + b3 * rs ** sp.Rational(3, 2) + b4 * rs ** 2)))
    betap, gamma = sp.Float('0.06672455060314922', 20), (1 - sp.log(2)) / pi ** 2
    ks = sp.sqrt(4 * kf / pi)
    t2 = sigma / (4 * ks ** 2 * rho ** 2)
    Aa = betap / gamma / (sp.exp(-ec / gamma) - 1)
    H = gamma * sp.log(1 + betap / gamma * t2 * (1 + Aa * t2) / (1 + Aa * t2 + Aa ** 2 * t2 ** 2))
    out['pbec'] = rho * (ec + H)
    # short-range B88 exchange of the ITYH scheme (Iikura, Tsuneda, Yanai, Hirao, JCP 115, 3540 (2001)) as libxc's
    # gga_x_ityh defines it: e_s = -cx rho_s^(4/3) F(x) att(a), a = omega / (2 k), k = sqrt(9 pi / (2 cx F)) rho_s^(1/3)
    om = sp.Symbol('omega', positive=True)
    out['ityh'] = 2 * _ityh_spin(sp, rs_, sigma / 4, om)
    out['wb97'] = _wb97(sp, rho / 2, rho / 2, sigma / 4, sigma / 4, om)
    fns = {}
    mods = [{'erf': _erf}, 'numpy']
    for k, e in out.items():
        v = (rho, sigma, om) if k in _WITH_OMEGA else (rho, sigma)
        # lambdas (and the symbolic derivatives behind them) are made when first asked for
        fns[k] = _LazySeq([lambda e=e, v=v: sp.lambdify(v, e, mods),
                           lambda e=e, v=v: sp.lambdify(v, sp.diff(e, rho), mods),
                           lambda e=e, v=v: sp.lambdify(v, sp.diff(e, sigma), mods),
                           lambda e=e, v=v: sp.lambdify(v, sp.diff(e, rho, 2), mods),
                           lambda e=e, v=v: sp.lambdify(v, sp.diff(e, rho, sigma), mods),
                           lambda e=e, v=v: sp.lambdify(v, sp.diff(e, sigma, 2), mods)])
    return fns


_WITH_OMEGA = ('ityh', 'wb97')


class _LazySeq:
    """Sequence of callables built on first access (int index or slice)."""

    def __init__(self, makers):
        self._makers = makers
        self._made = {}

    def _get(self, i):
        if i not in self._made:
            self._made[i] = self._makers[i]()
        return self._made[i]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self._makers)))]
        return self._get(i if i >= 0 else len(self._makers) + i)


def _wb97(sp, ra, rb, saa, sbb, om):
    """omega-B97 exchange-correlation energy per volume (Chai, Head-Gordon, JCP 128, 084106 (2008), Table I; libxc
    hyb_gga_xc_wb97): short-range LSDA exchange (the erf attenuation of _ityh_spin with k_F of the spin density) times the B97
    series in u = g s^2 / (1 + g s^2), g = 0.004; B97 correlation on the Stoll partition of the ORIGINAL PW92 parametrisation
    (not the extra-digit 'pw_mod' one PBE uses), same-spin g = 0.2, opposite-spin g = 0.006 on the mean of the two s^2.  The
    long-range exchange is the exact one (alpha = 1 at omega = 0.4).  Pinned by He / cc-pVDZ -2.89430888240579
    (pyscf/dft/test/test_he.py:92-95)."""
    pi = sp.pi
    F = lambda v: sp.Float(v, 20)
    cx = [F(v) for v in ('1.00000', '1.13116', '-2.74915', '12.0900', '-5.71642')]
    css = [F(v) for v in ('1.00000', '-2.55352', '11.8926', '-26.9452', '17.0927')]
    cos = [F(v) for v in ('1.00000', '3.99051', '-17.0066', '1.07292', '8.88211')]

    def series(c, g, s, r83):
        u = g * s / (r83 + g * s)                              # = g s^2 / (1 + g s^2) without forming s^2 = s / r^(8/3)
        return sum(ci * u ** i for i, ci in enumerate(c))

    def att(a):
        b = sp.exp(-1 / (4 * a * a)) - 1
        c = 2 * a * a * b + sp.Rational(1, 2)
        closed = 1 - sp.Rational(8, 3) * a * (sp.sqrt(pi) * sp.erf(1 / (2 * a)) + 2 * a * (b - c))
        coef = [36, -960, 26880, -829440, 28385280, -1073479680, 44590694400, -2021444812800]
        ser = sum(sp.Integer(1) / (sp.Integer(cf) * a ** (2 * n + 2)) for n, cf in enumerate(coef))
        return sp.Piecewise((ser, a > 2), (closed, True))

    def ex_spin(r, s):
        clda = sp.Rational(3, 2) * (3 / (4 * pi)) ** sp.Rational(1, 3)
        kf = (6 * pi ** 2 * r) ** sp.Rational(1, 3)
        return -clda * r ** sp.Rational(4, 3) * att(om / (2 * kf)) * series(cx, F('0.004'), s, r ** sp.Rational(8, 3))

    def pw_g(rs, A, a1, b1, b2, b3, b4):
        q = 2 * A * (b1 * sp.sqrt(rs) + b2 * rs + b3 * rs ** sp.Rational(3, 2) + b4 * rs ** 2)
        return -2 * A * (1 + a1 * rs) * sp.log(1 + 1 / q)
    para = [F(x) for x in ('0.031091', '0.21370', '7.5957', '3.5876', '1.6382', '0.49294')]
    ferro = [F(x) for x in ('0.015545', '0.20548', '14.1189', '6.1977', '3.3662', '0.62517')]
    stiff = [F(x) for x in ('0.016887', '0.11125', '10.357', '3.6231', '0.88026', '0.49671')]

    def ec_pw(na, nb):
        n = na + nb
        zeta = (na - nb) / n
        rs = (3 / (4 * pi * n)) ** sp.Rational(1, 3)
        fz = ((1 + zeta) ** sp.Rational(4, 3) + (1 - zeta) ** sp.Rational(4, 3) - 2) / (2 ** sp.Rational(4, 3) - 2)
        e0, e1, mac = pw_g(rs, *para), pw_g(rs, *ferro), pw_g(rs, *stiff)
        return n * (e0 - mac * fz / F('1.709921') * (1 - zeta ** 4) + (e1 - e0) * fz * zeta ** 4)

    def ec_ferro(n):
        return n * pw_g((3 / (4 * pi * n)) ** sp.Rational(1, 3), *ferro)
    ra83, rb83 = ra ** sp.Rational(8, 3), rb ** sp.Rational(8, 3)
    eaa, ebb = ec_ferro(ra), ec_ferro(rb)
    eab = ec_pw(ra, rb) - eaa - ebb
    # opposite spin: u of the mean s^2 = (saa / ra^(8/3) + sbb / rb^(8/3)) / 2
    g_os = F('0.006')
    s_av = (saa * rb83 + sbb * ra83) / 2
    u_os = g_os * s_av / (ra83 * rb83 + g_os * s_av)
    return (ex_spin(ra, saa) + ex_spin(rb, sbb) + eaa * series(css, F('0.2'), saa, ra83) + ebb * series(css, F('0.2'), sbb, rb83) +
            eab * sum(ci * u_os ** i for i, ci in enumerate(cos)))


def _erf(x):
    from scipy.special import erf
    x = np.asarray(x)
    if x.dtype == np.longdouble:                       # scipy has no 80-bit erf; its argument 1/(2a) is harmless in float64
        return erf(x.astype(np.float64)).astype(np.longdouble)
    return erf(x)


def _ityh_spin(sp, r, s, om):
    """Attenuated B88 exchange energy (per volume) of one spin channel with density r and |grad r|^2 = s.  att(a) = 1 - 8/3 a
    (sqrt(pi) erf(1/(2a)) + 2a (b - c)), b = exp(-1/(4a^2)) - 1, c = 2a^2 b + 1/2; for a > 2 its expansion in 1/a^2 (the
    closed form loses digits by cancellation there: 5e-13 relative at a = 2 in float64)."""
    pi = sp.pi
    beta = sp.Float('0.0042', 20)
    cx = sp.Rational(3, 2) * (3 / (4 * pi)) ** sp.Rational(1, 3)
    x = sp.sqrt(s) / r ** sp.Rational(4, 3)
    F = 1 + beta / cx * x * x / (1 + 6 * beta * x * sp.asinh(x))
    k = sp.sqrt(9 * pi / (2 * cx * F)) * r ** sp.Rational(1, 3)
    a = om / (2 * k)
    b = sp.exp(-1 / (4 * a * a)) - 1
    c = 2 * a * a * b + sp.Rational(1, 2)
    closed = 1 - sp.Rational(8, 3) * a * (sp.sqrt(pi) * sp.erf(1 / (2 * a)) + 2 * a * (b - c))
    coef = [36, -960, 26880, -829440, 28385280, -1073479680, 44590694400, -2021444812800]
    series = sum(sp.Integer(1) / (sp.Integer(cf) * a ** (2 * n + 2)) for n, cf in enumerate(coef))
    att = sp.Piecewise((series, a > 2), (closed, True))
    return -cx * r ** sp.Rational(4, 3) * F * att


_ORDER = ['slater', 'vwn5', 'vwnrpa', 'b88', 'lyp', 'pbex', 'pbec', 'ityh', 'wb97']      # fac[9] = omega of 'ityh' / 'wb97'


def _args(fac, name, *v):
    if name == 'wb97':
        # the expanded derivatives of the B97 series hold powers like rho^(-40/3) sigma^4: evaluated in 80-bit floats so that
        # they neither overflow nor underflow on the low-density tail (results are cast back by the callers' += into float64)
        v = tuple(np.asarray(x, dtype=np.longdouble) for x in v)
    return v + (float(fac[9]),) if name in _WITH_OMEGA else v


def eval_xc(fac, rho, sigma):
    """-> e (per volume), vrho, vsigma for component weights fac[7] (same order as the device)."""
    global _FUNCS
    if _FUNCS is None:
        _FUNCS = _build_functionals()
    e = np.zeros_like(rho)
    vr = np.zeros_like(rho)
    vs = np.zeros_like(rho)
    ok = rho > 1e-14
    r = rho[ok]
    s = np.maximum(sigma[ok], 1e-300)
    for w, name in zip(fac, _ORDER):
        if w == 0:
            continue
        f, fr, fs = _FUNCS[name][:3]
        a = _args(fac, name, r, s)
        e[ok] += w * f(*a)
        vr[ok] += w * fr(*a)
        vs[ok] += w * fs(*a) * np.ones_like(r)
    return e, vr, vs


def eval_fxc(fac, rho, sigma):
    """-> v2rho2, v2rhosigma, v2sigma2 (second derivatives of the energy per volume; zero where rho <= 1e-14)."""
    global _FUNCS
    if _FUNCS is None:
        _FUNCS = _build_functionals()
    out = [np.zeros_like(rho) for _ in range(3)]
    ok = rho > 1e-14
    r = rho[ok]
    s = np.maximum(sigma[ok], 1e-300)
    for w, name in zip(fac, _ORDER):
        if w == 0:
            continue
        for k in range(3):
            out[k][ok] += w * _FUNCS[name][3 + k](*_args(fac, name, r, s)) * np.ones_like(r)
    return out


def nr_rks_fxc(mol, coords, weights, fac, gga, dm0, dm1):
    """Closed-shell XC kernel contracted with a first-order density matrix: dense restatement of numint.nr_rks_fxc
    (numint.py:1418-1530) with the weights of _rks_gga_wv1 (:1560-1576).  Only the symmetric part of dm1 has a density."""
    dm0 = (dm0 + dm0.T) * .5
    dm1 = (dm1 + dm1.T) * .5
    ao = eval_ao(mol, coords, 1 if gga else 0)
    if not gga:
        ao = ao[None] if ao.ndim == 2 else ao
    c0, c1 = ao[0].dot(dm0), ao[0].dot(dm1)
    rho0 = np.einsum('gi,gi->g', ao[0], c0)
    rho1 = np.einsum('gi,gi->g', ao[0], c1)
    if gga:
        g0 = 2 * np.einsum('xgi,gi->xg', ao[1:4], c0)
        g1 = 2 * np.einsum('xgi,gi->xg', ao[1:4], c1)
        sigma = np.einsum('xg,xg->g', g0, g0)
        sig1 = 2 * np.einsum('xg,xg->g', g0, g1)
    else:
        sigma = np.zeros_like(rho0)
    vs = eval_xc(fac, rho0, sigma)[2]
    frr, frs, fss = eval_fxc(fac, rho0, sigma)
    if not gga:
        aow = ao[0] * (.5 * weights * frr * rho1)[:, None]
    else:
        aow = ao[0] * (.5 * weights * (frr * rho1 + frs * sig1))[:, None]
        for d in range(3):
            aow += ao[1 + d] * (2 * weights * ((frs * rho1 + fss * sig1) * g0[d] + vs * g1[d]))[:, None]
    m = ao[0].T.dot(aow)
    return m + m.T


def nr_rks(mol, coords, weights, fac, gga, dm):
    """(nelec, excsum, vmat) - dense restatement of numint.nr_rks (numint.py:1116-1157)."""
    dm = (dm + dm.T) * .5
    if gga:
        ao = eval_ao(mol, coords, 1)
        c0 = ao[0].dot(dm)
        rho = np.einsum('gi,gi->g', ao[0], c0)
        grad = 2 * np.einsum('xgi,gi->xg', ao[1:], c0)
        sigma = np.einsum('xg,xg->g', grad, grad)
    else:
        ao0 = eval_ao(mol, coords, 0)
        ao = ao0[None]
        rho = np.einsum('gi,ij,gj->g', ao0, dm, ao0)
        sigma = np.zeros_like(rho)
    e, vr, vs = eval_xc(fac, rho, sigma)
    nelec = np.dot(weights, rho)
    exc = np.dot(weights, e)
    wv0 = .5 * weights * vr
    aow = ao[0] * wv0[:, None]
    if gga:
        for d in range(3):
            aow += ao[1 + d] * (2 * weights * vs * grad[d])[:, None]
    m = ao[0].T.dot(aow)
    return nelec, exc, m + m.T


def rks_energy(mol, xc_fac, hyb, gga, coords, weights, get_jk, conv_tol=1e-10, verbose=False, dm0=None, h1e=None,
               s1e=None):
    """RKS SCF with the oracle pieces.  get_jk(dm, c, occ, with_k) -> (vj, vk|None).  dm0: optional start density (the
    orbitals handed to get_jk are then None on the first call)."""
    state = {}

    def veff(dm, c, occ):
        n, exc, vxc = nr_rks(mol, coords, weights, xc_fac, gga, dm)
        vj, vk = get_jk(dm, c, occ, hyb != 0)
        v = vxc + vj
        e2 = .5 * np.einsum('ij,ji', dm, vj) + exc
        if hyb != 0:
            v = v - .5 * hyb * vk
            e2 -= .25 * hyb * np.einsum('ij,ji', dm, vk)
        state['e2'] = e2
        state['nelec'] = n
        return v
    return ref.rhf_kernel(mol, veff, conv_tol=conv_tol, verbose=verbose, e2_fn=lambda: state['e2'], dm0=dm0, h1e=h1e,
                          s1e=s1e)


def eval_ao_hess(mol, coords, h=1e-4):
    """(3, 3, ngrids, nao) second derivatives of the AOs: central differences (step h, Richardson with 2h) of
    the analytic first derivatives of eval_ao - accurate to ~1e-10 relative, enough for a 1e-8 gradient check
    without a second hand-written derivative formula in the oracle."""
    ng = len(coords)
    nao = mol.nao_nr()
    out = np.zeros((3, 3, ng, nao))
    for d in range(3):
        def g1(step):
            cp, cm = coords.copy(), coords.copy()
            cp[:, d] += step
            cm[:, d] -= step
            return (eval_ao(mol, cp, 1)[1:] - eval_ao(mol, cm, 1)[1:]) / (2 * step)
        out[d] = (4 * g1(h) - g1(2 * h)) / 3
    return out


def _vmat_grad(mol, coords, weights, fac, gga, dm):
    """(vmat[3][nao][nao], e[g]): the derivative matrices of pyscf/grad/rks.py get_vxc (:119-195; vmat[x] =
    -(nabla_x ao)^T (wv0 ao) for LDA, _gga_grad_sum_ / _make_dR_dao_w :197-255 for GGA) on the given points, and the
    XC energy density per unit volume at those points."""
    ao = eval_ao(mol, coords, 1)
    c0 = ao[0].dot(dm)
    rho = np.einsum('gi,gi->g', ao[0], c0)
    if gga:
        grad = 2 * np.einsum('xgi,gi->xg', ao[1:], c0)
        sigma = np.einsum('xg,xg->g', grad, grad)
    else:
        grad = np.zeros((3, len(rho)))
        sigma = np.zeros_like(rho)
    e, vr, vs = eval_xc(fac, rho, sigma)
    nao = mol.nao_nr()
    vmat = np.zeros((3, nao, nao))
    if not gga:
        aow = ao[0] * (weights * vr)[:, None]
        for x in range(3):
            vmat[x] = ao[1 + x].T.dot(aow)
    else:
        hess = eval_ao_hess(mol, coords)
        wv = np.empty((4, len(rho)))
        wv[0] = weights * vr * .5
        wv[1:] = 2 * weights * vs * grad
        aow = np.einsum('cgi,cg->gi', ao, wv)
        for x in range(3):
            vmat[x] = ao[1 + x].T.dot(aow)
            aow2 = ao[1 + x] * wv[0][:, None] + np.einsum('kgi,kg->gi', hess[x], wv[1:])
            vmat[x] += aow2.T.dot(ao[0])
    return -vmat, e


def nr_rks_grad(mol, coords, weights, fac, gga, dm):
    """XC nuclear gradient (natm, 3) without grid response: numpy restatement of pyscf/grad/rks.py get_vxc
    (:119-195) + the contraction de[A] = 2 sum_{mu on A, nu} vmat[x]_{mu nu} D_{mu nu} of grad/rhf.py:80-84."""
    dm = (dm + dm.T) * .5
    vmat, _ = _vmat_grad(mol, coords, weights, fac, gga, dm)
    aoslices = mol.aoslice_by_atom()
    de = np.zeros((mol.natm, 3))
    for ia in range(mol.natm):
        p0, p1 = aoslices[ia][2], aoslices[ia][3]
        de[ia] = 2 * np.einsum('xij,ij->x', vmat[:, p0:p1], dm[p0:p1])
    return de


def nr_rks_grad_response(mol, coords, weights, owner, radii_table, fac, gga, dm, scheme='becke'):
    """The two grid-response terms of pyscf/grad/rks.py get_vxc_full_response (:257-340) to be added to nr_rks_grad:
    sum_g e_g dw_g/dR_C (Becke weights follow the nuclei) and, for the points owned by atom C, the motion of the point
    itself: -2 sum_{mu nu} vmat_C[x]_{mu nu} D_{mu nu} with vmat_C restricted to C's points."""
    dm = (dm + dm.T) * .5
    de = np.zeros((mol.natm, 3))
    _, e = _vmat_grad(mol, coords, weights, fac, gga, dm)
    dw = becke_weight_response(coords, owner, weights, mol.atom_coords(), radii_table, scheme)
    de += np.einsum('g,cxg->cx', e, dw)
    for ia in range(mol.natm):
        sel = owner == ia
        vmat_c, _ = _vmat_grad(mol, coords[sel], weights[sel], fac, gga, dm)
        de[ia] -= 2 * np.einsum('xij,ij->x', vmat_c, dm)
    return de


# ============================================================================ spin-polarised (UKS)
_FUNCS_POL = None


def _build_functionals_pol():
    """e(rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb) per unit volume and its 5 first derivatives."""
    import sympy as sp
    ra, rb, saa, sab, sbb = sp.symbols('ra rb saa sab sbb', positive=True)
    pi = sp.pi
    rho = ra + rb
    sig = saa + 2 * sab + sbb
    zeta = (ra - rb) / rho
    out = {}
    cx = sp.Rational(3, 2) * (3 / (4 * pi)) ** sp.Rational(1, 3)
    out['slater'] = -cx * (ra ** sp.Rational(4, 3) + rb ** sp.Rational(4, 3))

    def vwn_eps(A, x0, b, c):
        rs = (3 / (4 * pi * rho)) ** sp.Rational(1, 3)
        x = sp.sqrt(rs)
        Q = sp.sqrt(4 * c - b * b)
        X = x * x + b * x + c
        X0 = x0 * x0 + b * x0 + c
        at = sp.atan(Q / (2 * x + b))
        return A * (sp.log(x * x / X) + 2 * b / Q * at -
                    b * x0 / X0 * (sp.log((x - x0) ** 2 / X) + 2 * (b + 2 * x0) / Q * at))
    F = lambda v: sp.Float(v, 20)
    fz = ((1 + zeta) ** sp.Rational(4, 3) + (1 - zeta) ** sp.Rational(4, 3) - 2) / (2 ** sp.Rational(4, 3) - 2)
    fpp = sp.Rational(4, 9) / (2 ** sp.Rational(1, 3) - 1)
    A_alpha = -1 / (6 * pi ** 2)

    def vwn(para, ferro, alpha):
        eP, eF, eA = vwn_eps(*para), vwn_eps(*ferro), vwn_eps(*alpha)
        return rho * (eP + eA * fz / fpp * (1 - zeta ** 4) + (eF - eP) * fz * zeta ** 4)
    out['vwn5'] = vwn((F('0.0310907'), F('-0.10498'), F('3.72744'), F('12.9352')),
                      (F('0.01554535'), F('-0.32500'), F('7.06042'), F('18.0578')),
                      (A_alpha, F('-0.0047584'), F('1.13107'), F('13.0045')))
    # libxc LDA_C_VWN_RPA interpolates para / ferro linearly in f(zeta) (no spin-stiffness term); established
    # against the reference's UKS B3LYPG golden (tests/test_oracle_dft_golden.py)
    out['vwnrpa'] = rho * (vwn_eps(F('0.0310907'), F('-0.409286'), F('13.0720'), F('42.7198')) * (1 - fz) +
                           vwn_eps(F('0.01554535'), F('-0.743294'), F('20.1231'), F('101.578')) * fz)
    beta = F('0.0042')

    def b88s(r, s):
        x = sp.sqrt(s) / r ** sp.Rational(4, 3)
        return -cx * r ** sp.Rational(4, 3) - beta * r ** sp.Rational(4, 3) * x * x / (1 + 6 * beta * x * sp.asinh(x))
    out['b88'] = b88s(ra, saa) + b88s(rb, sbb)
    a, b, c, d = [F(v) for v in ('0.04918', '0.132', '0.2533', '0.349')]
    CF = sp.Rational(3, 10) * (3 * pi ** 2) ** sp.Rational(2, 3)
    rm13 = rho ** sp.Rational(-1, 3)
    den = 1 + d * rm13
    omega = sp.exp(-c * rm13) / den * rho ** sp.Rational(-11, 3)
    delta = c * rm13 + d * rm13 / den
    br = (ra * rb * (2 ** sp.Rational(11, 3) * CF * (ra ** sp.Rational(8, 3) + rb ** sp.Rational(8, 3))
                     + (sp.Rational(47, 18) - sp.Rational(7, 18) * delta) * sig
                     - (sp.Rational(5, 2) - delta / 18) * (saa + sbb)
                     - (delta - 11) / 9 * (ra / rho * saa + rb / rho * sbb))
          - sp.Rational(2, 3) * rho ** 2 * sig + (sp.Rational(2, 3) * rho ** 2 - ra ** 2) * sbb
          + (sp.Rational(2, 3) * rho ** 2 - rb ** 2) * saa)
    out['lyp'] = -a * 4 / den * ra * rb / rho - a * b * omega * br
    # PBE (Perdew, Burke, Ernzerhof, PRL 77, 3865): exchange by the spin-scaling relation, correlation with the
    # PW92 ("pw_mod" digits) uniform-gas energy e_c(rs, zeta), phi(zeta) and H(rs, zeta, t)
    kappa, mu = F('0.804'), F('0.2195149727645171')
    cx_u = sp.Rational(3, 4) * (3 / pi) ** sp.Rational(1, 3)

    def pbex_unpol(r, s_):
        kf = (3 * pi ** 2 * r) ** sp.Rational(1, 3)
        s2 = s_ / (4 * kf ** 2 * r ** 2)
        return -cx_u * r ** sp.Rational(4, 3) * (1 + kappa - kappa / (1 + mu / kappa * s2))
    out['pbex'] = (pbex_unpol(2 * ra, 4 * saa) + pbex_unpol(2 * rb, 4 * sbb)) / 2
    rs = (3 / (4 * pi * rho)) ** sp.Rational(1, 3)

    def pw_g(A, a1, b1, b2, b3, b4):
        q = 2 * A * (b1 * sp.sqrt(rs) + b2 * rs + b3 * rs ** sp.Rational(3, 2) + b4 * rs ** 2)
        return -2 * A * (1 + a1 * rs) * sp.log(1 + 1 / q)
    e0 = pw_g(*[F(x) for x in ('0.0310907', '0.21370', '7.5957', '3.5876', '1.6382', '0.49294')])
    e1 = pw_g(*[F(x) for x in ('0.01554535', '0.20548', '14.1189', '6.1977', '3.3662', '0.62517')])
    mac = pw_g(*[F(x) for x in ('0.0168869', '0.11125', '10.357', '3.6231', '0.88026', '0.49671')])
    fz20 = F('1.709920934161365617563962776245')
    ec = e0 - mac * fz / fz20 * (1 - zeta ** 4) + (e1 - e0) * fz * zeta ** 4
    beta_c, gamma_c = F('0.06672455060314922'), (1 - sp.log(2)) / pi ** 2
    phi = ((1 + zeta) ** sp.Rational(2, 3) + (1 - zeta) ** sp.Rational(2, 3)) / 2
    kf = (3 * pi ** 2 * rho) ** sp.Rational(1, 3)
    t2 = sig / (4 * phi ** 2 * (4 * kf / pi) * rho ** 2)
    Aa = beta_c / gamma_c / (sp.exp(-ec / (gamma_c * phi ** 3)) - 1)
    H = gamma_c * phi ** 3 * sp.log(1 + beta_c / gamma_c * t2 * (1 + Aa * t2) / (1 + Aa * t2 + Aa ** 2 * t2 ** 2))
    out['pbec'] = rho * (ec + H)
    om = sp.Symbol('omega', positive=True)
    out['ityh'] = _ityh_spin(sp, ra, saa, om) + _ityh_spin(sp, rb, sbb, om)
    out['wb97'] = _wb97(sp, ra, rb, saa, sbb, om)
    fns = {}
    v = (ra, rb, saa, sab, sbb)
    mods = [{'erf': _erf}, 'numpy']
    for k, e in out.items():
        vv = v + (om,) if k in _WITH_OMEGA else v
        fns[k] = _LazySeq([lambda e=e, vv=vv: sp.lambdify(vv, e, mods)] +
                          [lambda e=e, vv=vv, x=x: sp.lambdify(vv, sp.diff(e, x), mods) for x in v])
    return fns


def eval_xc_pol(fac, ra, rb, saa, sab, sbb):
    """-> e, (vra, vrb, vsaa, vsab, vsbb) for the component weights fac[7] (order of _ORDER)."""
    global _FUNCS_POL
    if _FUNCS_POL is None:
        _FUNCS_POL = _build_functionals_pol()
    n = len(ra)
    e = np.zeros(n)
    dv = [np.zeros(n) for _ in range(5)]
    ok = (ra + rb) > 1e-14
    args = [np.maximum(ra[ok], 1e-30), np.maximum(rb[ok], 1e-30), np.maximum(saa[ok], 1e-300), sab[ok],
            np.maximum(sbb[ok], 1e-300)]
    # sympy symbols are declared positive; sab may be negative: shift-free evaluation is fine numerically
    for w, name in zip(fac, _ORDER):
        if w == 0:
            continue
        f = _FUNCS_POL[name]
        aa = _args(fac, name, *args)
        e[ok] += w * f[0](*aa)
        for k in range(5):
            dv[k][ok] += w * f[1 + k](*aa) * np.ones(ok.sum())
    return e, dv


def nr_uks(mol, coords, weights, fac, gga, dma, dmb):
    """(nelec[2], excsum, vmat[2]) - dense restatement of numint.nr_uks (pyscf/dft/numint.py:1192-1324)."""
    ao = eval_ao(mol, coords, 1) if gga else eval_ao(mol, coords, 0)[None]
    rho, grad = [], []
    for dm in (dma, dmb):
        dm = (dm + dm.T) * .5
        c0 = ao[0].dot(dm)
        rho.append(np.einsum('gi,gi->g', ao[0], c0))
        grad.append(2 * np.einsum('xgi,gi->xg', ao[1:], c0) if gga else np.zeros((3, len(coords))))
    saa = np.einsum('xg,xg->g', grad[0], grad[0])
    sab = np.einsum('xg,xg->g', grad[0], grad[1])
    sbb = np.einsum('xg,xg->g', grad[1], grad[1])
    e, (vra, vrb, vsaa, vsab, vsbb) = eval_xc_pol(fac, rho[0], rho[1], saa, sab, sbb)
    nelec = (np.dot(weights, rho[0]), np.dot(weights, rho[1]))
    exc = np.dot(weights, e)
    vmat = []
    for s, (vr, vss, g_same, g_other) in enumerate(((vra, vsaa, grad[0], grad[1]), (vrb, vsbb, grad[1], grad[0]))):
        aow = ao[0] * (.5 * weights * vr)[:, None]
        if gga:
            for d in range(3):
                aow += ao[1 + d] * (weights * (2 * vss * g_same[d] + vsab * g_other[d]))[:, None]
        m = ao[0].T.dot(aow)
        vmat.append(m + m.T)
    return nelec, exc, np.array(vmat)


def nr_uks_fxc(mol, coords, weights, fac, gga, dm0a, dm0b, dm1a, dm1b, h=2e-4):
    """(2, nao, nao): the spin-polarised XC kernel contracted with first-order spin density matrices, numint.nr_uks_fxc
    (numint.py:1690-1832), as what it is - the derivative of the nr_uks potential along (dm1a, dm1b) - by central
    differences with Richardson extrapolation (h, 2h -> O(h^4)); no second hand-written set of second derivatives.
    Valid where the first-order densities are small against the zeroth-order ones (physical perturbations)."""
    def v(t):
        return nr_uks(mol, coords, weights, fac, gga, dm0a + t * dm1a, dm0b + t * dm1b)[2]
    d1 = (v(h) - v(-h)) / (2 * h)
    d2 = (v(2 * h) - v(-2 * h)) / (4 * h)
    return (4 * d1 - d2) / 3


def nr_uks_grad(mol, coords, weights, fac, gga, dma, dmb):
    """XC nuclear gradient (natm, 3) of a spin-polarised density, grid response left out: numpy restatement of
    pyscf/grad/uks.py get_vxc (:100-190) + the contraction with (D_alpha, D_beta) of grad/uhf.py:72-76."""
    ao = eval_ao(mol, coords, 1)
    dms = [(dma + dma.T) * .5, (dmb + dmb.T) * .5]
    c0 = [ao[0].dot(d) for d in dms]
    rho = [np.einsum('gi,gi->g', ao[0], c) for c in c0]
    grad = [2 * np.einsum('xgi,gi->xg', ao[1:], c) if gga else np.zeros((3, len(coords))) for c in c0]
    saa = np.einsum('xg,xg->g', grad[0], grad[0])
    sab = np.einsum('xg,xg->g', grad[0], grad[1])
    sbb = np.einsum('xg,xg->g', grad[1], grad[1])
    e, (vra, vrb, vsaa, vsab, vsbb) = eval_xc_pol(fac, rho[0], rho[1], saa, sab, sbb)
    nao = mol.nao_nr()
    hess = eval_ao_hess(mol, coords) if gga else None
    aoslices = mol.aoslice_by_atom()
    de = np.zeros((mol.natm, 3))
    for s, (vr, vss, g_same, g_other) in enumerate(((vra, vsaa, grad[0], grad[1]), (vrb, vsbb, grad[1], grad[0]))):
        vmat = np.zeros((3, nao, nao))
        if not gga:
            aow = ao[0] * (weights * vr)[:, None]
            for x in range(3):
                vmat[x] = ao[1 + x].T.dot(aow)
        else:
            wv = np.empty((4, len(weights)))
            wv[0] = weights * vr * .5
            wv[1:] = weights * (2 * vss * g_same + vsab * g_other)
            aow = np.einsum('cgi,cg->gi', ao, wv)
            for x in range(3):
                vmat[x] = ao[1 + x].T.dot(aow)
                aow2 = ao[1 + x] * wv[0][:, None] + np.einsum('kgi,kg->gi', hess[x], wv[1:])
                vmat[x] += aow2.T.dot(ao[0])
        vmat = -vmat
        for ia in range(mol.natm):
            p0, p1 = aoslices[ia][2], aoslices[ia][3]
            de[ia] += 2 * np.einsum('xij,ij->x', vmat[:, p0:p1], dms[s][p0:p1])
    return de


def uks_energy(mol, xc_fac, hyb, gga, coords, weights, eri, nelec, conv_tol=1e-10, max_cycle=100):
    """UKS SCF with exact 4-centre J/K (for the reference's non-DF goldens)."""
    import scipy.linalg
    h1e = ref.int1e(mol, 'kin') + ref.int1e(mol, 'nuc')
    s1e = ref.int1e(mol, 'ovlp')
    enuc = mol.energy_nuc()
    w, v = scipy.linalg.eigh(s1e)
    x = v[:, w > 1e-6] / np.sqrt(w[w > 1e-6])

    def eig(f):
        e, c = scipy.linalg.eigh(x.T.dot(f).dot(x))
        return x.dot(c)
    c0 = eig(h1e)
    cs = [c0, c0]
    fs, es = [], []
    e_tot = 0
    for cycle in range(max_cycle):
        dms = np.array([cs[s][:, :nelec[s]].dot(cs[s][:, :nelec[s]].T) for s in range(2)])
        n, exc, vxc = nr_uks(mol, coords, weights, xc_fac, gga, dms[0], dms[1])
        vj, vk = ref.get_jk_exact(eri, dms)
        vjt = vj[0] + vj[1]
        f = h1e + vxc + vjt - hyb * vk
        e_last = e_tot
        e_tot = (np.einsum('ij,ji', h1e, dms[0] + dms[1]) + .5 * np.einsum('ij,ji', vjt, dms[0] + dms[1]) + exc
                 - .5 * hyb * sum(np.einsum('ij,ji', vk[s], dms[s]) for s in range(2)) + enuc)
        err = np.hstack([x.T.dot(f[s].dot(dms[s]).dot(s1e) - s1e.dot(dms[s]).dot(f[s])).dot(x).ravel() for s in range(2)])
        if abs(e_tot - e_last) < conv_tol and np.linalg.norm(err) < 1e-5:
            return True, e_tot
        fs.append(f); es.append(err)
        fs, es = fs[-8:], es[-8:]
        m = len(fs)
        h = np.zeros((m + 1, m + 1)); h[0, 1:] = h[1:, 0] = 1
        for i in range(m):
            for j in range(m):
                h[i + 1, j + 1] = es[i].dot(es[j])
        g = np.zeros(m + 1); g[0] = 1
        ww, vv = scipy.linalg.eigh(h)
        idx = abs(ww) > 1e-14
        c = np.dot(vv[:, idx] * (1. / ww[idx]), vv[:, idx].T.dot(g))
        f = sum(ci * fi for ci, fi in zip(c[1:], fs))
        cs = [eig(f[s]) for s in range(2)]
    return False, e_tot
