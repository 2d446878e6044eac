cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize_scf.py 2>&1 | tail -40 > gpurun_out/pytest_r02f.log
tail -6 gpurun_out/pytest_r02f.log
timeout 300 python tools/run_scf.py --nwater 32 --xc b3lyp > gpurun_out/scf_h2o32_b3lyp_r02f.log 2>&1
grep "cycle= 6\|cycle= 7\|converged" gpurun_out/scf_h2o32_b3lyp_r02f.log
timeout 300 python -X importtime -c "pass" 2>/dev/null
timeout 300 python - <<'PY' > gpurun_out/scf_profile.log 2>&1
import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
from pyscf_amd import gto, dft
from pyscf_amd.data import clusters
mol = gto.M(atom=clusters.water_cluster(32), basis='cc-pvtz')
mf = dft.RKS(mol, xc='b3lyp').density_fit()
mf.max_cycle = 4
mf.kernel()
mf.max_cycle = 6
pr = cProfile.Profile(); pr.enable()
mf.kernel(dm0=mf.make_rdm1())
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
PY
grep -A60 "cumulative" gpurun_out/scf_profile.log | cut -c1-150 | head -70
