"""XC functional description parser (the subset of ``pyscf/dft/libxc.py`` the path needs:
``parse_xc`` :496-720, ``XC_CODES`` :60-210, ``hybrid_coeff``/``rsh_coeff``/``xc_type``).

A functional is reduced to the weights of the building blocks implemented by the device kernel
``PAMD_eval_xc`` (order: Slater, VWN5, VWN_RPA, B88, LYP, PBE_X, PBE_C) plus the exact-exchange
fraction.  Names follow PySCF: 'LDA' / 'SLATER' = Slater exchange, 'VWN' = 'VWN5' (libxc id 7),
'VWN_RPA' = 'VWNRPA' = 'VWN3' (id 8, libxc.py:168-169), 'B3LYP' = 'B3LYPG' = id 402 (VWN_RPA,
libxc.py:175), 'B3LYP5' = VWN5 flavour (:177)."""
import numpy as np

F_SLATER, F_VWN5, F_VWNRPA, F_B88, F_LYP, F_PBEX, F_PBEC = range(7)
_GGA = {F_B88, F_LYP, F_PBEX, F_PBEC}

_X = {'LDA': {F_SLATER: 1.}, 'SLATER': {F_SLATER: 1.}, 'LDA_X': {F_SLATER: 1.}, 'S': {F_SLATER: 1.},
      'B88': {F_B88: 1.}, 'B': {F_B88: 1.}, 'PBE': {F_PBEX: 1.}, 'HF': {}}
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
}


def parse_xc(description):
    """-> (hyb, fac[7]); see parse_xc_rsh for the range-separated exact-exchange terms."""
    hyb, alpha, omega, fac = parse_xc_rsh(description)
    return hyb, fac


def parse_xc_rsh(description):
    """-> (hyb, alpha, omega, fac[7]).  Grammar subset of libxc.parse_xc (:496-720): 'X,C' with '+'-separated,
    optionally 'w*name'-weighted terms, or a single compound name; exact exchange as 'HF' (full range: counts for
    hyb and alpha), 'SR_HF(omega)' (hyb only) and 'LR_HF(omega)' (alpha only), so that
    K = hyb K_full + (alpha - hyb) K_LR(omega)   (pyscf/dft/rks.py:110-127)."""
    name = description.upper().replace(' ', '')
    fac = np.zeros(7)
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
            h, comps = _XC[token]
            hyb += w * h
            alpha += w * h
            for k, v in comps.items():
                fac[k] += w * v
            return
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
    return hyb, alpha, omega, fac


def xc_type(description):
    _, fac = parse_xc(description)
    if any(fac[k] != 0 for k in _GGA):
        return 'GGA'
    return 'LDA' if np.any(fac != 0) else 'HF'


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
