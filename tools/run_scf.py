"""End-to-end DF-SCF on a water cluster through the reference-style API.
    python tools/run_scf.py --nwater 32 --basis cc-pvtz --xc b3lyp     (xc '' -> RHF)
    python tools/run_scf.py --molecule taxol --basis def2-tzvp --xc b3lyp"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyscf_amd import gto, scf, dft
from pyscf_amd.data import clusters
ap = argparse.ArgumentParser()
ap.add_argument('--nwater', type=int, default=32)
ap.add_argument('--basis', default=None, help='default: cc-pvtz (water), def2-tzvp (taxol)')
ap.add_argument('--xc', default='b3lyp')
ap.add_argument('--conv-tol', type=float, default=1e-9)
ap.add_argument('--molecule', default='water', choices=['water', 'taxol'])
ap.add_argument('--level-shift', type=float, default=0.0)
ap.add_argument('--max-cycle', type=int, default=50)
ap.add_argument('--dump-orbitals', default='', help='.npz: orbo = C_occ sqrt(occ) and e_tot of the converged state (input of tools/gen_golden_streaming.py)')
ap.add_argument('--host-loop', action='store_true', help='the numpy SCF loop instead of the HBM-resident one')
ap.add_argument('--native', action='store_true', help='J/K through the host-array C handle (NativeDF): rows that do not fit HBM are streamed from '
                'page-locked host memory - e.g. (H2O)_128 cc-pVDZ (560 GB tensor) on ONE GPU')
ap.add_argument('--layout', default='', help="tensor layout of df.DF: '' (the budget's choice) | packed | square; --no-image: square rows whenever 2x fits")
ap.add_argument('--no-image', action='store_true')
ap.add_argument('--devices', default='', help='with --native: comma list of HIP devices for the handle')
a = ap.parse_args()
if a.basis is None:
    a.basis = 'def2-tzvp' if a.molecule == 'taxol' else 'cc-pvtz'
mol = gto.M(atom=clusters.taxol() if a.molecule == 'taxol' else clusters.water_cluster(a.nwater), basis=a.basis, verbose=4)
mf = (dft.RKS(mol, xc=a.xc) if a.xc else scf.RHF(mol))
if a.native:
    from pyscf_amd.df.native import NativeDF
    devs = [int(d) for d in a.devices.split(',')] if a.devices else None
    t0 = time.perf_counter()
    mf = mf.density_fit(with_df=NativeDF(mol, devices=devs).build())
    print('NativeDF built in %.1f s: layout %s' % (time.perf_counter() - t0, mf.with_df.layout()), flush=True)
else:
    mf = mf.density_fit()
    if a.layout:
        mf.with_df.layout = a.layout
    if a.no_image:
        mf.with_df.prefer_image = False
mf.conv_tol = a.conv_tol
mf.level_shift = a.level_shift
mf.max_cycle = a.max_cycle
if a.host_loop:
    mf.device_scf = False
t0 = time.perf_counter()
e = mf.kernel()
print('converged=%s cycles=%d E=%.10f wall=%.1f s (nao=%d naux=%d) layout=%s j2=%s' %
      (mf.converged, mf.cycles, e, time.perf_counter() - t0, mol.nao, mf.with_df.get_naoaux(), getattr(mf.with_df, '_layout', None),
       getattr(mf.with_df, '_j2_policy_times', None)), flush=True)
if a.dump_orbitals:
    import numpy as np
    occ = mf.mo_occ > 0
    np.savez(a.dump_orbitals, orbo=mf.mo_coeff[:, occ] * np.sqrt(mf.mo_occ[occ]), e_tot=e, converged=mf.converged,
             mo_energy=mf.mo_energy)
