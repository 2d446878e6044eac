#!/bin/bash
# The one GPU-box launcher:  gpurun --timeout T -- 'bash tools/gpu_job.sh <job> [args]'
# Every job writes under gpurun_out/<job>/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
cd ${GRAFT_REPO_ROOT:-.}
JOB=${1:-suite}; shift
O=gpurun_out/$JOB; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
case $JOB in
suite)      # whole GPU suite, default bench line, smoke
  timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > $O/pytest_gpu.log 2>&1; tail -18 $O/pytest_gpu.log
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 800 $O/bench_default.json; tail -3 $O/bench_default.err
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
tests)      # selected tests: gpu_job.sh tests <pytest args>
  timeout 1800 python -m pytest -q -x --durations=8 "$@" > $O/pytest.log 2>&1; tail -25 $O/pytest.log ;;
taxol_dump) # converged DF-RKS B3LYP orbitals of config 4 (input of the oracle functional golden) + VALU counters of the build
  timeout 900 python tools/run_scf.py --molecule taxol --xc b3lyp --conv-tol 1e-10 --dump-orbitals $O/taxol_b3lyp_orbitals.npz > $O/scf_taxol_b3lyp.log 2>&1
  tail -4 $O/scf_taxol_b3lyp.log
  cd /tmp
  B="python $R/tools/build_only.py"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/build_stats -o b -- $B > $R/$O/build_stats.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/$O/build_valu -o b -- $B > $R/$O/build_valu.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/build_mfma -o b -- $B > $R/$O/build_mfma.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/build_write -o b -- $B > $R/$O/build_write.log 2>&1
  cd $R; tail -2 $O/build_valu.log; find $O -name "*.csv" | head; find $O -name "*.db" -delete ;;
run)        # arbitrary command line, logged: gpu_job.sh run <tag> <cmd...>
  T=$1; shift; timeout 1500 "$@" > $O/$T.log 2>&1; tail -30 $O/$T.log ;;
*) echo "unknown job $JOB"; exit 2 ;;
esac
