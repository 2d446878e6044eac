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
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  timeout 900 python bench.py --molecule taxol --no-cpu-baseline > $O/bench_taxol_1gpu.json 2> $O/bench_taxol.err; cut -c1-240 $O/bench_taxol_1gpu.json; tail -2 $O/bench_taxol.err ;;
suiteA)     # GPU suite in three parts (a lost box takes its logs with it): A = bench_launch .. fullsize
  timeout 2000 python -m pytest -q -x --durations=8 -m gpu tests/test_gpu_bench_launch.py tests/test_gpu_cabi_kernels.py tests/test_gpu_device_scf.py tests/test_gpu_df_jk.py tests/test_gpu_dft.py tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; tail -14 $O/pytest.log ;;
suiteB)     # B = fullsize_cfg45, grad, int3c2e
  timeout 2000 python -m pytest -q -x --durations=8 -m gpu tests/test_gpu_fullsize_cfg45.py tests/test_gpu_fullsize_scf.py tests/test_gpu_grad.py tests/test_gpu_int3c2e.py > $O/pytest.log 2>&1; tail -14 $O/pytest.log ;;
suiteC)     # C = native_abi .. xc_sparse, then the default bench line and smoke
  timeout 2400 python -m pytest -q -x --durations=8 -m gpu tests/test_gpu_native_abi.py tests/test_gpu_native_r04.py tests/test_gpu_rccl.py tests/test_gpu_response.py tests/test_gpu_scf.py tests/test_gpu_soscf.py tests/test_gpu_tdscf.py tests/test_gpu_vhf.py tests/test_gpu_xc_sparse.py > $O/pytest.log 2>&1; tail -14 $O/pytest.log
  tail -8 gpurun_out/_native_cfg45_worker_config5.log
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_digest.py $O/bench_default.json; tail -3 $O/bench_default.err
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
xcab)       # XC leg A/B: r03 sub_vmat vs r04 sub_vmat_sym, with / without the XCD-aware work order
  : > $O/xcbench.log
  timeout 600 python -m pytest -q -x tests/test_gpu_xc_sparse.py tests/test_gpu_dft.py -m gpu -k "not fxc" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
  for v in "--tune-xc orbrho=0" "--tune-xc orbrho=1"; do
    echo "== $v" >> $O/xcbench.log
    timeout 400 python tools/xcbench.py --steps 5 $v 2>/dev/null | tail -1 >> $O/xcbench.log
  done
  python - <<'PY'
import json
for l in open('gpurun_out/xcab/xcbench.log'):
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print('   wall', d['wall_ms_per_call'], 'nelec %.12f exc %.12f' % (d['nelec'], d['exc']), d['kernel_ms'], d['plan']['density'], d['plan']['density2'])
PY
  ;;
xcpmc)      # counters of the XC kernels for the sub_vmat variants: gpu_job.sh xcpmc
  cd /tmp
  for tag in "r03:--vmat-sym 0 --tune-xc vmatxcd=0" "r03x:--vmat-sym 0 --tune-xc vmatxcd=1" "sym:--vmat-sym 1 --tune-xc vmatxcd=0" "symx:--vmat-sym 1 --tune-xc vmatxcd=1"; do
    T=${tag%%:*}; V=${tag#*:}
    B="python $R/tools/xcbench.py --steps 2 $V"
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/${T}_fetch -o x -- $B > $R/$O/${T}_fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/${T}_write -o x -- $B > $R/$O/${T}_write.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/${T}_sq -o x -- $B > $R/$O/${T}_sq.log 2>&1
  done
  cd $R
  python tools/pmc_table.py $O > $O/summary.txt; cat $O/summary.txt
  find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete ;;
e2w)        # wide last orbital chunk of the half transform (taxol shape): kernel tests, then kbench with / without
  timeout 900 python -m pytest -q -x tests/test_gpu_cabi_kernels.py tests/test_gpu_df_jk.py tests/test_gpu_native_abi.py tests/test_gpu_device_scf.py tests/test_gpu_fullsize_cfg45.py::test_config4_taxol_full_jk_and_energy_vs_oracle_golden -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
  : > $O/kbench.log
  for t in "$@"; do
    [ "$t" = "-" ] && T="" || T="$t"
    echo "== tune=$T" >> $O/kbench.log
    timeout 400 python tools/kbench.py --steps 4 --nao 2228 --naux 5598 --nocc 226 --j2-policy serial ${T:+--tune $T} 2>&1 | tail -1 >> $O/kbench.log
  done
  cut -c1-600 $O/kbench.log ;;
cfg5)       # BASELINE config 5 whole (560 GB) on one GPU through the out-of-core handle (opt-in test)
  timeout 1500 python -m pytest -q -x tests/test_gpu_00_config5_whole_tensor.py -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
  tail -4 gpurun_out/_native_cfg45_worker_config5.log ;;
cfg5scf)    # BASELINE config 5 converged on ONE GPU through the out-of-core handle; dumps the occupied orbitals (oracle energy golden input)
  nproc > $O/host.txt; grep -E "MemTotal|MemAvailable" /proc/meminfo >> $O/host.txt; df -h /tmp . >> $O/host.txt; cat $O/host.txt
  timeout 1500 python tools/run_scf.py --nwater 128 --basis cc-pvdz --xc '' --native --conv-tol ${1:-1e-10} --dump-orbitals $O/h2o128_rhf_orbitals.npz > $O/scf_h2o128_rhf_native.log 2>&1
  grep -E "NativeDF built|cycle=|converged" $O/scf_h2o128_rhf_native.log | tail -40 ;;
probe)      # one-off hardware probes
  ./tools/probe/cu_mask_probe.bin 2>&1 | tee $O/cu_mask_probe.log ;;
kab)        # kbench A/B of tuning keys on the config-3 shape: gpu_job.sh kab "<tune1>" "<tune2>" ...  (use - for none)
  : > $O/kbench.log
  for t in "$@"; do
    [ "$t" = "-" ] && T="" || T="$t"
    for rep in 1 2; do
      echo "== tune=$T" >> $O/kbench.log
      timeout 300 python tools/kbench.py --steps 5 $KBENCH_ARGS ${T:+--tune $T} 2>&1 | tail -1 >> $O/kbench.log
    done
  done
  cut -c1-420 $O/kbench.log ;;
kpol)       # kbench over J-pass-2 schedules: gpu_job.sh kpol "<kbench args 1>" "<kbench args 2>" ...
  : > $O/kbench.log
  for t in "$@"; do
    for rep in 1 2; do
      echo "== $t" >> $O/kbench.log
      timeout 300 python tools/kbench.py --steps 5 $t 2>&1 | tail -1 >> $O/kbench.log
    done
  done
  cut -c1-330 $O/kbench.log ;;
r04a)       # new tests of the round + bench modes
  timeout 1200 python -m pytest -q -x --durations=6 tests/test_gpu_native_r04.py::test_stock_script_with_a_device_list_reaches_the_reference_golden tests/test_gpu_fullsize_cfg45.py::test_config4_taxol_df_rks_xc_and_energy_vs_oracle_golden > $O/pytest.log 2>&1; tail -12 $O/pytest.log
  timeout 600 python bench.py --gpus 2 --single-process --steps 3 > $O/bench_single_process_2parts.json 2> $O/bench_sp.err; cut -c1-700 $O/bench_single_process_2parts.json; tail -2 $O/bench_sp.err
  timeout 1200 python bench.py --pmc --steps 10 --warmup 2 > $O/bench_pmc.json 2> $O/bench_pmc.err; tail -c 1500 $O/bench_pmc.json; tail -3 $O/bench_pmc.err ;;
r05a)       # r05 validation: new handle paths, un-gated config 5, rank shard out of core, bench modes with parity_golden
  timeout 2400 python -m pytest -q -x --durations=8 tests/test_gpu_native_abi.py tests/test_gpu_native_r04.py tests/test_gpu_fullsize_cfg45.py::test_config5_rank_shard_out_of_core_behind_DF_vs_oracle_golden tests/test_gpu_scf.py tests/test_gpu_grad.py -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
  tail -6 gpurun_out/_native_cfg45_worker_config5.log
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_digest.py $O/bench_default.json; tail -3 $O/bench_default.err
  timeout 600 python bench.py --gpus 1 --single-process --steps 5 > $O/bench_sp1.json 2> $O/bench_sp1.err; python tools/bench_digest.py $O/bench_sp1.json; tail -3 $O/bench_sp1.err
  timeout 600 python bench.py --gpus 2 --single-process --steps 5 --no-cpu-baseline --xc '' > $O/bench_sp2.json 2> $O/bench_sp2.err; python tools/bench_digest.py $O/bench_sp2.json; tail -3 $O/bench_sp2.err
  timeout 600 python bench.py --gpus 2 --backend gloo --steps 3 --no-cpu-baseline --xc '' > $O/bench_gloo2.json 2> $O/bench_gloo2.err; python tools/bench_digest.py $O/bench_gloo2.json; tail -3 $O/bench_gloo2.err
  timeout 300 python tools/native_bench.py > $O/native_bench.log 2>&1; tail -4 $O/native_bench.log ;;
r05b)       # r05: threaded probe + overlapped J download + calibrated CPU baseline + default pmc; mmap streaming test
  timeout 1500 python -m pytest -q -x --durations=6 tests/test_gpu_native_abi.py tests/test_gpu_native_r04.py::test_cderi_file_is_streamed_from_an_mmap_when_it_does_not_fit tests/test_gpu_native_r04.py::test_torch_resident_df_falls_back_to_the_streaming_handle_when_the_tensor_does_not_fit tests/test_gpu_native_r04.py::test_multi_device_handle_streaming_and_omega_without_torch tests/test_gpu_native_r04.py::test_config3_through_the_native_handle_vs_oracle_golden tests/test_gpu_bench_launch.py tests/test_gpu_device_scf.py -m gpu > $O/pytest.log 2>&1; tail -12 $O/pytest.log
  T0=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench wall: $(( $(date +%s) - T0 )) s"; python tools/bench_digest.py $O/bench_default.json; tail -3 $O/bench_default.err
  timeout 600 python bench.py --gpus 1 --single-process --steps 10 --xc '' > $O/bench_sp1.json 2> $O/bench_sp1.err; python tools/bench_digest.py $O/bench_sp1.json; tail -3 $O/bench_sp1.err
  timeout 600 python bench.py --gpus 2 --single-process --steps 5 --no-cpu-baseline --xc '' > $O/bench_sp2.json 2> $O/bench_sp2.err; python tools/bench_digest.py $O/bench_sp2.json; tail -3 $O/bench_sp2.err
  timeout 300 python tools/native_bench.py > $O/native_bench.log 2>&1; tail -4 $O/native_bench.log ;;
jf)         # r05: second J pass inside the SYRK kernel (PAMD_syrk_jfused) - tests, then kbench / bench A/B at config 3 and taxol shape
  timeout 600 python -m pytest -q -x tests/test_gpu_df_jk.py -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
  : > $O/kbench.log
  for pol in overlap fused serial overlap fused; do
    echo "== config3 $pol" >> $O/kbench.log
    timeout 300 python tools/kbench.py --steps 5 --j2-policy $pol --syrk-reserve 16 2>&1 | tail -1 >> $O/kbench.log
  done
  for pol in serial fused; do
    echo "== taxol $pol" >> $O/kbench.log
    timeout 400 python tools/kbench.py --steps 4 --nao 2228 --naux 5598 --nocc 226 --j2-policy $pol 2>&1 | tail -1 >> $O/kbench.log
  done
  cut -c1-420 $O/kbench.log
  timeout 600 python bench.py --j2-policy fused --no-cpu-baseline --no-pmc --xc '' --steps 10 > $O/bench_fused.json 2> $O/bench_fused.err; python tools/bench_digest.py $O/bench_fused.json; tail -3 $O/bench_fused.err
  timeout 600 python bench.py --j2-policy overlap --no-cpu-baseline --no-pmc --xc '' --steps 10 > $O/bench_overlap.json 2> $O/bench_overlap.err; python tools/bench_digest.py $O/bench_overlap.json; tail -3 $O/bench_overlap.err ;;
jf2)        # host API under the fused policy (diagnostic), df_jk tests, taxol bench with the fused policy
  timeout 600 python -m pytest -q -x tests/test_gpu_df_jk.py -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
  for pol in overlap fused; do timeout 300 python tools/host_api_probe.py --j2-policy $pol > $O/host_api_$pol.log 2>&1; grep -E "host API|cumtime|get_jk|_to_host|download|mismatch|dot|synchronize|run_fused|_vk_mo" $O/host_api_$pol.log | head -24; done
  timeout 900 python bench.py --molecule taxol --j2-policy fused --no-cpu-baseline --no-pmc --xc '' > $O/bench_taxol_fused.json 2> $O/bench_taxol_fused.err; python tools/bench_digest.py $O/bench_taxol_fused.json; tail -3 $O/bench_taxol_fused.err ;;
evidence)   # the round's measured evidence (everything except the test suite): gpu_job.sh evidence <tag>
  TAG=${1:-r05}
  bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
  timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_h2o32_1gpu_steps20.json 2> $O/bench20.err; python tools/bench_digest.py $O/bench_h2o32_1gpu_steps20.json; tail -2 $O/bench20.err
  timeout 600 python bench.py --k-square off --no-cpu-baseline --no-pmc --xc '' > $O/bench_h2o32_1gpu_ksquare_off.json 2> $O/bench_ksq.err; cut -c1-200 $O/bench_h2o32_1gpu_ksquare_off.json
  timeout 900 python bench.py --molecule taxol --no-cpu-baseline > $O/bench_taxol_1gpu.json 2> $O/bench_taxol.err; python tools/bench_digest.py $O/bench_taxol_1gpu.json; python -c "import json; d=json.loads(open('$O/bench_taxol_1gpu.json').read().strip().splitlines()[-1]); print('   taxol schedule', d['jk_schedule'])"
  timeout 600 python bench.py --gpus 1 --single-process --steps 10 > $O/bench_single_process_1part.json 2> $O/bench_sp1.err; python tools/bench_digest.py $O/bench_single_process_1part.json
  timeout 600 python bench.py --gpus 2 --single-process --steps 5 --no-cpu-baseline > $O/bench_single_process_2parts_1gpu.json 2> $O/bench_sp2.err; python tools/bench_digest.py $O/bench_single_process_2parts_1gpu.json
  timeout 300 python tools/native_bench.py > $O/native_bench.log 2>&1; tail -5 $O/native_bench.log
  for pol in overlap; do timeout 300 python tools/host_api_probe.py --j2-policy $pol > $O/host_api_$pol.log 2>&1; grep -E "host API|gc gen" $O/host_api_$pol.log | head -12; done
  timeout 600 python tools/run_scf.py --nwater 32 --xc b3lyp --conv-tol 1e-10 > $O/scf_h2o32_b3lyp.log 2>&1; tail -2 $O/scf_h2o32_b3lyp.log
  timeout 600 python tools/run_scf.py --nwater 32 --xc '' --conv-tol 1e-10 > $O/scf_h2o32_rhf.log 2>&1; tail -1 $O/scf_h2o32_rhf.log
  timeout 600 python tools/run_scf.py --molecule taxol --xc b3lyp --conv-tol 1e-9 > $O/scf_taxol_b3lyp.log 2>&1; tail -1 $O/scf_taxol_b3lyp.log
  timeout 600 python tools/grad_bench.py --nwater 32 > $O/grad_h2o32_rhf.json 2> $O/grad.err; cat $O/grad_h2o32_rhf.json | cut -c1-300
  # r06: a complete N = 2 line (two gloo ranks on the one GPU: cpu_baseline on rank 0's shard, live PMC of the shard, parity_golden)
  timeout 900 python bench.py --gpus 2 --backend gloo --pmc on --steps 5 > $O/bench_h2o32_2rank_gloo_1gpu.json 2> $O/bench_gloo2.err; python tools/bench_digest.py $O/bench_h2o32_2rank_gloo_1gpu.json; tail -2 $O/bench_gloo2.err
  # r06: the compute partition mode of the device, READ ONLY (a CPX run of 8 RCCL ranks on one MI355X was not attempted: see profiles/r06/README.md)
  ( rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20; amd-smi partition 2>&1 | head -30 ) > $O/partition_query.txt 2>&1; head -12 $O/partition_query.txt
  find gpurun_out -name "*.db" -delete ;;
syrkfetch)  # r06: SYRK dispatch orders x FETCH_SIZE x ms.  gpu_job.sh syrkfetch "<tune1>" "<tune2>" ... (- = none)
  : > $O/kbench.log
  for t in "$@"; do
    [ "$t" = "-" ] && T="" || T="$t"
    TAG=$(echo "${T:-base}" | tr '=,' '__')
    for sc in "jk:--j2-policy overlap --syrk-reserve 16" "konly:--no-j"; do
      S=${sc%%:*}; A=${sc#*:}
      for rep in 1 2; do
        echo "== $S tune=$T" >> $O/kbench.log
        timeout 300 python tools/kbench.py --steps 5 $A ${T:+--tune $T} 2>&1 | tail -1 >> $O/kbench.log
      done
      ( cd /tmp; timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/${S}_${TAG} -o x -- python $R/tools/kbench.py --steps 2 $A ${T:+--tune $T} > $R/$O/${S}_${TAG}.log 2>&1 )
    done
  done
  cut -c1-560 $O/kbench.log
  python tools/pmc_table.py $O syrk_slots gemm_tn_glds2 vj_pass2 e2_sq2 > $O/summary.txt; cat $O/summary.txt
  find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete ;;
sq1)        # r06: the square layout - tests, then kbench packed + image (3x) vs square rows only (2x) at config-3 and taxol shape
  timeout 1500 python -m pytest -q -x --durations=6 -m gpu tests/test_gpu_square_layout.py tests/test_gpu_df_jk.py tests/test_gpu_device_scf.py tests/test_gpu_scf.py tests/test_gpu_grad.py > $O/pytest.log 2>&1; tail -12 $O/pytest.log
  : > $O/kbench.log
  for A in "--layout packed --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy overlap --syrk-reserve 16" "--layout packed --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy serial" "--layout square --no-j"; do
    echo "== config3 $A" >> $O/kbench.log
    timeout 300 python tools/kbench.py --steps 5 $A 2>&1 | tail -1 >> $O/kbench.log
  done
  for A in "--layout square --j2-policy serial" "--layout square --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy overlap" "--layout packed --j2-policy serial"; do
    echo "== taxol $A" >> $O/kbench.log
    timeout 500 python tools/kbench.py --steps 4 --nao 2228 --naux 5598 --nocc 226 $A 2>&1 | tail -1 >> $O/kbench.log
  done
  cut -c1-420 $O/kbench.log ;;
sq2)        # r06: square layout with / without the padded aux-row stride; fullsize tests; bench default + taxol
  timeout 1800 python -m pytest -q -x --durations=6 -m gpu tests/test_gpu_square_layout.py tests/test_gpu_df_jk.py tests/test_gpu_native_abi.py tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; tail -12 $O/pytest.log
  : > $O/kbench.log
  for A in "--layout packed --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy overlap --syrk-reserve 16" "--layout square --sq-contiguous --j2-policy overlap --syrk-reserve 16" "--layout packed --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy overlap --syrk-reserve 16" "--layout square --sq-contiguous --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy serial" "--layout square --sq-contiguous --j2-policy serial"; do
    echo "== config3 $A" >> $O/kbench.log
    timeout 300 python tools/kbench.py --steps 5 $A 2>&1 | tail -1 >> $O/kbench.log
  done
  for A in "--layout square --j2-policy overlap --syrk-reserve 16" "--layout square --sq-contiguous --j2-policy overlap --syrk-reserve 16" "--layout square --j2-policy serial"; do
    echo "== taxol $A" >> $O/kbench.log
    timeout 500 python tools/kbench.py --steps 4 --nao 2228 --naux 5598 --nocc 226 $A 2>&1 | tail -1 >> $O/kbench.log
  done
  cut -c1-420 $O/kbench.log
  timeout 900 python bench.py --no-cpu-baseline --no-pmc > $O/bench_default.json 2> $O/bench_default.err; python tools/bench_digest.py $O/bench_default.json; tail -3 $O/bench_default.err
  timeout 900 python bench.py --molecule taxol --no-cpu-baseline --no-pmc > $O/bench_taxol_1gpu.json 2> $O/bench_taxol.err; python tools/bench_digest.py $O/bench_taxol_1gpu.json; tail -3 $O/bench_taxol.err ;;
r06b)       # r06: rest of the suite after the layout change, gradient Z-slab GEMM A/B, layout A/B of the bench line on ONE box
  timeout 2400 python -m pytest -q -x --durations=8 -m gpu tests/test_gpu_fullsize_cfg45.py tests/test_gpu_fullsize_scf.py tests/test_gpu_grad.py tests/test_gpu_int3c2e.py tests/test_gpu_xc_sparse.py tests/test_gpu_native_r04.py > $O/pytest.log 2>&1; tail -14 $O/pytest.log
  for z in hip torch; do
    PAMD_GRAD_ZGEMM=$z timeout 600 python tools/grad_bench.py --nwater 32 > $O/grad_h2o32_rhf_$z.json 2> $O/grad_$z.err; cut -c1-900 $O/grad_h2o32_rhf_$z.json; tail -2 $O/grad_$z.err
  done
  for lay in packed auto packed auto; do
    timeout 600 python bench.py --layout $lay --no-cpu-baseline --no-pmc --xc '' --steps 10 > $O/bench_layout_$lay.json 2> $O/bench_$lay.err
    python - <<PY
import json
d=json.loads(open('$O/bench_layout_$lay.json').read().strip().splitlines()[-1])
print('layout $lay ->', d['config']['tensor_layout'], 'ms', d['value'], 'host', d['value_host_api_ms'], d['host_api_ms_calls'], {k: v['ms_total'] for k, v in d['kernels'].items()}, d['config']['hbm_after_build_GB'], d['jk_schedule'])
PY
  done
  timeout 900 python bench.py --molecule taxol --no-cpu-baseline --no-pmc > $O/bench_taxol_1gpu.json 2> $O/bench_taxol.err; python tools/bench_digest.py $O/bench_taxol_1gpu.json; tail -3 $O/bench_taxol.err ;;
r06c)       # r06: the tests the -x stop of the suite did not reach + the all-DMA cderi_solve (tensor tests, build A/B, counters)
  timeout 2400 python -m pytest -q -x --durations=6 -m gpu tests/test_gpu_square_layout.py tests/test_gpu_tdscf.py tests/test_gpu_vhf.py tests/test_gpu_xc_sparse.py tests/test_gpu_int3c2e.py tests/test_gpu_fullsize.py tests/test_gpu_scf.py tests/test_gpu_native_abi.py > $O/pytest.log 2>&1; tail -8 $O/pytest.log
  for v in 0 1 0 1; do PAMD_SOLVE_V2=$v timeout 300 python tools/build_only.py --time --layout packed 2>/dev/null | tail -1; done | tee $O/build_ab.jsonl
  PAMD_SOLVE_V2=1 timeout 300 python tools/build_only.py --time --layout square 2>/dev/null | tail -1 | tee -a $O/build_ab.jsonl
  ( cd /tmp; for v in 0 1; do PAMD_SOLVE_V2=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/solve_stats_v$v -o b -- python $R/tools/build_only.py > $R/$O/solve_stats_v$v.log 2>&1; done
    PAMD_SOLVE_V2=1 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/solve_mfma -o b -- python $R/tools/build_only.py > $R/$O/solve_mfma.log 2>&1 )
  for v in 0 1; do f=$(find $O/solve_stats_v$v -name "*kernel_stats.csv" | head -1); echo "== PAMD_SOLVE_V2=$v"; grep -E "cderi_solve|unpack_slab|int3c2e_kernel<3, 3, 4>" $f | cut -c1-220 | head -5; done
  python tools/pmc_table.py $O cderi_solve | tee $O/solve_pmc.txt
  find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete ;;
r06e)       # r06: sub_vmat_sym wave-role flip A/B + the wave-balance model of the plan's ld distribution; host profile of a whole SCF
  : > $O/xcbench.log
  for v in "vmatflip=0" "vmatflip=1" "vmatflip=2" "vmatflip=0" "vmatflip=1" "vmatflip=2"; do
    echo "== $v" >> $O/xcbench.log
    timeout 400 python tools/xcbench.py --steps 5 --tune-xc $v 2>/dev/null | tail -1 >> $O/xcbench.log
  done
  python - <<'PY'
import json
for l in open('gpurun_out/r06e/xcbench.log'):
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print('   wall', d['wall_ms_per_call'], d['kernel_ms'], d['executed']['ao_dot_aow'], 'balance model', d.get('vmat_sym_wave_balance_model'))
print('ld / 16 histogram', d.get('ld_groups_hist'))
PY
  for A in "" "--no-image" "" "--no-image"; do timeout 600 python tools/run_scf.py --nwater 32 --xc b3lyp --conv-tol 1e-10 $A 2>&1 | grep -E "df vj|init E|cycle= [12] |converged" | cut -c1-260; done | tee $O/scf_layouts.log
  timeout 600 python tools/prof_scf.py --max-cycle 4 > $O/prof_scf.log 2>&1; grep -E "^clocks|cycle=|converged" $O/prof_scf.log | head; sed -n '/cumulative/,+45p' $O/prof_scf.log | cut -c1-170 | head -60 ;;
r06f)       # r06: even pieces in the sub_vmat_sym work list (A/B + XC tests); whole-SCF wall, packed / square alternating, with the build clock
  timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_xc_sparse.py tests/test_gpu_dft.py -k "not fxc" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
  : > $O/xcbench.log
  for v in "vmateven=0" "vmateven=1" "vmateven=0" "vmateven=1"; do
    echo "== $v" >> $O/xcbench.log
    timeout 400 python tools/xcbench.py --steps 5 --tune-xc $v 2>/dev/null | tail -1 >> $O/xcbench.log
  done
  python - <<'PY'
import json
for l in open('gpurun_out/r06f/xcbench.log'):
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print('   wall', d['wall_ms_per_call'], 'nelec %.10f exc %.10f' % (d['nelec'], d['exc']), d['kernel_ms'], d['executed']['ao_dot_aow'])
PY
  for A in "" "--no-image" "" "--no-image"; do timeout 600 python tools/run_scf.py --nwater 32 --xc b3lyp --conv-tol 1e-10 $A 2>&1 | grep -E "DF tensor|one-electron|setting up|df vj|init E|cycle= [12] |converged" | cut -c1-260; done | tee $O/scf_layouts.log ;;
r06g)       # r06 (second half): int3c2e classes - LDS bank-conflict share and time per class after the odd unit stride
  timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_int3c2e.py tests/test_gpu_cabi_kernels.py > $O/pytest.log 2>&1; tail -3 $O/pytest.log
  ( cd /tmp; B="python $R/tools/build_only.py"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/build_stats -o b -- $B > $R/$O/build_stats.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/build_lds -o b -- $B > $R/$O/build_lds.log 2>&1 )
  python - <<'PY'
import csv, glob, re, json
st = glob.glob('gpurun_out/r06g/build_stats/**/*kernel_stats.csv', recursive=True)[0]
ms = {}
for r in csv.DictReader(open(st)):
    m = re.search(r'int3c2e_kernel<(\d+), (\d+), (\d+)', r['Name'])
    if m: ms['%s,%s|%s' % m.groups()] = (float(r['TotalDurationNs']) * 1e-6, int(r['Calls']))
pc = glob.glob('gpurun_out/r06g/build_lds/**/*counter_collection.csv', recursive=True)[0]
agg = {}
for r in csv.DictReader(open(pc)):
    m = re.search(r'int3c2e_kernel<(\d+), (\d+), (\d+)', r['Kernel_Name'])
    if not m: continue
    d = agg.setdefault('%s,%s|%s' % m.groups(), {})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
out = []
for k, (t, n) in sorted(ms.items(), key=lambda kv: -kv[1][0]):
    d = agg.get(k, {})
    out.append({'kernel': 'int3c2e<%s>' % k, 'launches': n, 'ms': round(t, 2),
                'lds_bank_conflict_frac': round(d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 0), 1), 3)})
json.dump(out, open('gpurun_out/r06g/pmc_build_path_int3c2e.json', 'w'), indent=0)
print('total int3c2e ms', round(sum(o['ms'] for o in out), 1))
for o in out[:22]: print(o)
PY
  find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete ;;
kfetch)     # r06: kbench argument strings x FETCH_SIZE x ms: gpu_job.sh kfetch "<kbench args 1>" "<kbench args 2>" ...
  : > $O/kbench.log
  i=0
  for A in "$@"; do
    i=$((i+1))
    for rep in 1 2; do
      echo "== [$i] $A" >> $O/kbench.log
      timeout 300 python tools/kbench.py --steps 5 $A 2>&1 | tail -1 >> $O/kbench.log
    done
    ( cd /tmp; timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/v$i -o x -- python $R/tools/kbench.py --steps 2 $A > $R/$O/v$i.log 2>&1 )
  done
  cut -c1-330 $O/kbench.log
  python tools/pmc_table.py $O syrk_slots gemm_tn_glds2 vj_pass2 e2_sq2 > $O/summary.txt; cat $O/summary.txt
  find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -delete ;;
sparsity)   # r06: static tile sparsity of the tensor at configs 3, 4 and one config-5 shard
  timeout 600 python tools/tile_sparsity.py --nwater 32 --basis cc-pvtz 2>&1 | tail -1 > $O/tile_sparsity.jsonl
  timeout 600 python tools/tile_sparsity.py --molecule taxol --basis def2-tzvp 2>&1 | tail -1 >> $O/tile_sparsity.jsonl
  timeout 900 python tools/tile_sparsity.py --nwater 128 --basis cc-pvdz --world 8 --rank 3 2>&1 | tail -1 >> $O/tile_sparsity.jsonl
  cat $O/tile_sparsity.jsonl ;;
run)        # arbitrary command line, logged: gpu_job.sh run <tag> <cmd...>
  T=$1; shift; timeout 1500 "$@" > $O/$T.log 2>&1; tail -30 $O/$T.log ;;
*) echo "unknown job $JOB"; exit 2 ;;
esac
