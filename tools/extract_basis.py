"""Build pyscf_amd/gto/basis/data.json from the NWChem-format basis tables that
ship with the reference (pyscf/gto/basis/*.dat, public Basis-Set-Exchange data).

Run in the authoring container only (needs /root/reference):
    python tools/extract_basis.py
Only numeric exponent/coefficient tables of H-Kr (all-electron sets) are kept: the basis sets the hot path's
configurations need (sto-3g, 6-31g, cc-pvdz, cc-pvtz, def2-svp, def2-tzvp and their J/JK
fitting sets) plus a few common neighbours (cc-pvqz, aug-cc-pvdz/tz, Pople polarised sets).
"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from pyscf_amd.gto.basis import parse_nwchem

REF = '/root/reference/pyscf/gto/basis'
FILES = {
    'sto3g': 'sto-3g.dat', '631g': 'pople-basis/6-31G.dat',
    'ccpvdz': 'cc-pvdz.dat', 'ccpvtz': 'cc-pvtz.dat',
    'def2svp': 'def2-svp.dat', 'def2tzvp': 'def2-tzvp.dat',
    'ccpvdzjkfit': 'cc-pvdz-jkfit.dat', 'ccpvtzjkfit': 'cc-pvtz-jkfit.dat',
    'def2universaljkfit': 'def2-universal-jkfit.dat',
    'def2universaljfit': 'def2-universal-jfit.dat',
    'ccpvdzri': 'cc-pvdz-ri.dat',
    'ano': 'ano.dat',      # ANO-RCC tables: source of the MINAO initial guess (scf/hf.py:354-488)
    # widely used neighbours of the benchmark sets (AO shells up to g, fitting shells up to h are integral-ready)
    'ccpvqz': 'cc-pvqz.dat', 'augccpvdz': 'aug-cc-pvdz.dat', 'augccpvtz': 'aug-cc-pvtz.dat',
    'augccpvdzjkfit': 'aug-cc-pvdz-jkfit.dat', 'augccpvtzjkfit': 'aug-cc-pvtz-jkfit.dat',
    'ccpvtzri': 'cc-pvtz-ri.dat',
    '321g': 'pople-basis/3-21G.dat', '631gs': 'pople-basis/6-31Gs.dat', '631gss': 'pople-basis/6-31Gss.dat',
    '6311g': 'pople-basis/6-311G.dat', '6311gss': 'pople-basis/6-311Gss.dat',
    # quadruple-zeta sets: g AO shells / h fitting shells (3-centre classes up to (gg|h))
    'ccpvqzjkfit': 'cc-pvqz-jkfit.dat', 'augccpvqz': 'aug-cc-pvqz.dat', 'augccpvqzjkfit': 'aug-cc-pvqz-jkfit.dat',
    'ccpvqzri': 'cc-pvqz-ri.dat', 'def2qzvp': 'def2-qzvp.dat', 'def2qzvpp': 'def2-qzvpp.dat', 'def2tzvpp': 'def2-tzvpp.dat',
    'def2svpd': 'def2-svpd.dat', 'def2tzvpd': 'def2-tzvpd.dat',
}
ELEMENTS = ['H', 'He', 'Li', 'Be', 'B', 'C', 'N', 'O', 'F', 'Ne', 'Na', 'Mg', 'Al', 'Si', 'P', 'S', 'Cl', 'Ar',
            'K', 'Ca', 'Sc', 'Ti', 'V', 'Cr', 'Mn', 'Fe', 'Co', 'Ni', 'Cu', 'Zn', 'Ga', 'Ge', 'As', 'Se', 'Br', 'Kr']

out = {}
for name, fn in FILES.items():
    text = open(os.path.join(REF, fn)).read()
    out[name] = {}
    for el in ELEMENTS:
        try:
            out[name][el] = parse_nwchem.parse(text, el)
        except KeyError:
            pass
dst = os.path.join(os.path.dirname(__file__), '..', 'pyscf_amd', 'gto', 'basis', 'data.json')
json.dump(out, open(dst, 'w'), separators=(',', ':'))
print({k: sorted(v) for k, v in out.items()})
