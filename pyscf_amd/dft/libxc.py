"""XC functional description parser (the subset of ``pyscf/dft/libxc.py`` the path needs:
``parse_xc`` :496-720, ``XC_CODES`` :60-210, ``hybrid_coeff``/``rsh_coeff``/``xc_type``).

A functional is reduced to the weights of the building blocks implemented by the device kernel
``PAMD_eval_xc`` (order: Slater, VWN5, VWN_RPA, B88, LYP, PBE_X, PBE_C, ITYH = short-range B88 of the
Iikura-Tsuneda-Yanai-Hirao scheme, libxc gga_x_ityh; WB97 = the whole omega-B97 exchange-correlation functional; the last
slot of the array is the omega of the attenuated exchange) plus the exact-exchange fractions.  Names follow PySCF: 'LDA' / 'SLATER' = Slater exchange, 'VWN' = 'VWN5' (libxc id 7),
'VWN_RPA' = 'VWNRPA' = 'VWN3' (id 8, libxc.py:168-169), 'B3LYP' = 'B3LYPG' = id 402 (VWN_RPA,
libxc.py:175), 'B3LYP5' = VWN5 flavour (:177)."""
import re

import numpy as np

F_SLATER, F_VWN5, F_VWNRPA, F_B88, F_LYP, F_PBEX, F_PBEC, F_ITYH, F_WB97 = range(9)
F_OMEGA = 9                  # fac[F_OMEGA]: range-separation parameter of the attenuated exchange (F_ITYH, F_WB97)
NFAC = 10
_GGA = {F_B88, F_LYP, F_PBEX, F_PBEC, F_ITYH, F_WB97}

_X = {'LDA': {F_SLATER: 1.}, 'SLATER': {F_SLATER: 1.}, 'LDA_X': {F_SLATER: 1.}, 'S': {F_SLATER: 1.},
      'B88': {F_B88: 1.}, 'B': {F_B88: 1.}, 'PBE': {F_PBEX: 1.}, 'HF': {}, 'ITYH': {F_ITYH: 1.}}
_C = {'VWN': {F_VWN5: 1.}, 'VWN5': {F_VWN5: 1.}, 'VWN_RPA': {F_VWNRPA: 1.}, 'VWNRPA': {F_VWNRPA: 1.},
      'VWN3': {F_VWNRPA: 1.}, 'LYP': {F_LYP: 1.}, 'PBE': {F_PBEC: 1.}}
# compound names: (hyb, {component: weight})
_XC = {
    'B3LYP': (0.2, {F_SLATER: 0.08, F_B88: 0.72, F_VWNRPA: 0.19, F_LYP: 0.81}),
    'B3LYPG': (0.2, {F_SLATER: 0.08, F_B88: 0.72, F_VWNRPA: 0.19, F_LYP: 0.81}),
    'B3LYP5': (0.2, {F_SLATER: 0.08, F_B88: 0.72, F_VWN5: 0.19, F_LYP: 0.81}),
    'BLYP': (0.0, {F_B88: 1., F_LYP: 1.}),
    'PBE': (0.0, {F_PBEX: 1., F_PBEC: 1.}),
    'PBE0': (0.25, {F_PBEX: 0.75, F_PBEC: 1.}),
    'LDA': (0.0, {F_SLATER: 1.}), 'SVWN': (0.0, {F_SLATER: 1., F_VWN5: 1.}),
    'LSDA': (0.0, {F_SLATER: 1., F_VWN5: 1.}),
    'HF': (1.0, {}),
    # hyb_gga_xc_cam_b3lyp (Yanai, Tew, Handy, CPL 393, 51): (1 - 0.65) B88 + 0.46 ITYH(0.33) + 0.19 VWN5 + 0.81 LYP,
    # exact exchange 0.19 short range / 0.65 long range.  4-tuples: (short-range HF, components, long-range HF, omega)
    'CAMB3LYP': (0.19, {F_B88: 0.35, F_ITYH: 0.46, F_VWN5: 0.19, F_LYP: 0.81}, 0.65, 0.33),
    # hyb_gga_xc_wb97 (Chai, Head-Gordon, JCP 128, 084106): attenuated-LSDA B97 exchange + B97 correlation, no short-range and
    # full long-range exact exchange at omega = 0.4
    'WB97': (0.0, {F_WB97: 1.0}, 1.0, 0.4),
}


def parse_xc(description):
    """-> (hyb, fac[NFAC]); see parse_xc_rsh for the range-separated exact-exchange terms."""
    hyb, alpha, omega, fac = parse_xc_rsh(description)
    return hyb, fac


def parse_xc_rsh(description):
    """-> (hyb, alpha, omega, fac[NFAC]).  'RSH(omega,alpha,beta)' (libxc.py:640-660) sets omega and adds alpha to the long-range and
    alpha + beta to the short-range exact exchange.  Grammar subset of libxc.parse_xc (:496-720): 'X,C' with '+'-separated,
    optionally 'w*name'-weighted terms, or a single compound name; exact exchange as 'HF' (full range: counts for
    hyb and alpha), 'SR_HF(omega)' (hyb only) and 'LR_HF(omega)' (alpha only), so that
    K = hyb K_full + (alpha - hyb) K_LR(omega)   (pyscf/dft/rks.py:110-127)."""
    name = re.sub(r'(?<=[A-Z0-9])-(?=[A-Z])', '', description.upper().replace(' ', ''))     # 'CAM-B3LYP' = 'CAMB3LYP'
    name = _protect_rsh(name)
    fac = np.zeros(NFAC)
    hyb = 0.0
    alpha = 0.0
    omega = 0.0

    def add(table, token, allow_compound):
        nonlocal hyb, alpha, omega
        w = 1.0
        if '*' in token:
            a, b = token.split('*')
            try:
                w, token = float(a), b
            except ValueError:
                w, token = float(b), a
        if token in ('', 'NONE'):
            return
        if token == 'HF':
            hyb += w
            alpha += w
            return
        if token.startswith('RSH('):
            om, a_, b_ = [float(v) for v in token[4:-1].split(';')]
            if omega not in (0.0, om):
                raise ValueError('different values of omega in one functional')
            omega = om
            alpha += w * a_
            hyb += w * (a_ + b_)
            return
        if token.startswith(('SR_HF', 'LR_HF')):
            if '(' in token:
                om = float(token[token.index('(') + 1:token.index(')')])
                if omega not in (0.0, om):
                    raise ValueError('different values of omega in one functional')
                omega = om
            if token.startswith('SR_HF'):
                hyb += w
            else:
                alpha += w
            return
        if allow_compound and token in _XC and token not in table:
            h, comps = _XC[token][:2]
            hyb += w * h
            if len(_XC[token]) == 4:
                alpha += w * _XC[token][2]
                if omega not in (0.0, _XC[token][3]):
                    raise ValueError('different values of omega in one functional')
                omega = _XC[token][3]
            else:
                alpha += w * h
            for k, v in comps.items():
                fac[k] += w * v
            return
        if token not in table and table is _X and token in _C:
            table = _C                   # an unambiguous correlation name in the exchange part (libxc.parse_xc accepts it)
        if token not in table:
            raise NotImplementedError('XC component %s is not implemented on the device' % token)
        for k, v in table[token].items():
            fac[k] += w * v

    if ',' in name:
        xs, cs = name.split(',')
        for t in xs.split('+'):
            add(_X, t, True)
        for t in cs.split('+'):
            add(_C, t, False)
    else:
        for t in name.split('+'):
            if t in _XC or '*' in t and t.split('*')[1] in _XC:
                add({}, t, True)
            else:
                add(_X, t, True)
    if omega == 0.0:
        alpha = hyb                      # no range separation: one full-range coefficient
    if (fac[F_ITYH] != 0 or fac[F_WB97] != 0) and omega == 0.0:
        raise ValueError('the attenuated exchange (ITYH, WB97) needs a range-separation parameter: RSH(omega,alpha,beta)')
    fac[F_OMEGA] = omega
    return hyb, alpha, omega, fac


def _protect_rsh(name):
    """The commas inside RSH(omega,alpha,beta) would be taken for the exchange,correlation separator."""
    out, depth = [], 0
    for ch in name:
        depth += ch == '('
        depth -= ch == ')'
        out.append(';' if ch == ',' and depth else ch)
    return ''.join(out)


def xc_type(description):
    _, fac = parse_xc(description)
    if any(fac[k] != 0 for k in _GGA):
        return 'GGA'
    return 'LDA' if np.any(fac[:F_OMEGA] != 0) else 'HF'


def hybrid_coeff(description, spin=0):
    return parse_xc(description)[0]


def rsh_coeff(description):
    """(omega, alpha, beta) with hyb = alpha + beta: alpha weights the long-range, hyb the short-range exact exchange
    (pyscf/dft/libxc.py rsh_coeff)."""
    hyb, alpha, omega, _ = parse_xc_rsh(description)
    if omega == 0.0:
        return 0.0, 0.0, 0.0
    return omega, alpha, hyb - alpha


def is_hybrid_xc(description):
    hyb, alpha, omega, _ = parse_xc_rsh(description)
    return hyb != 0 or (omega != 0 and alpha != 0)
