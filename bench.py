#!/usr/bin/env python
"""Headline benchmark: ms per SCF iteration of the density-fitted J/K build.

Workload (BASELINE.json metric): (H2O)_32 cc-pVTZ, aux cc-pvtz-jkfit (nao 1856, naux 4448,
nocc 160; B = 61.3 GB FP64 resident in HBM), one `with_df.get_jk(dm, hermi=1)` (J and K, MO
branch) per step with the DM / orbitals already resident on the device.  With N GPUs the aux
index is sharded over the ranks and the partial J/K are all-reduced (RCCL): strong scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nwater 32] [--basis cc-pvtz]
    python bench.py --molecule taxol          # BASELINE config 4 on one GPU: C47H51NO14 def2-TZVP (data/taxol.xyz)

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel: HIP events around every launch of the TIMED steps,
each on the stream the kernel is launched on (`kernels`); `kernels_serial_pass` is one extra untimed step with J and K on
one stream (every kernel alone on the chip).  `cpu_baseline` times the reference's own C (oracle/_ref, built from the
sources under /root/reference where they exist; else the oracle's numpy restatement of pyscf/df/df_jk.py:329-381) on the
same tensor.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix peak (vendor spec; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec


def _aux_label(b):
    if isinstance(b, dict):
        names = sorted({v if isinstance(v, str) else 'even-tempered' for v in b.values()})
        return '/'.join(names)
    return str(b)


def _golden_case(args, nao, nocc):
    """(file, key prefix, rank of the seeded density) of the oracle-only J/K golden that matches this workload, or None."""
    if args.molecule == 'water' and args.nwater == 32 and args.basis == 'cc-pvtz':
        return 'h2o32_ccpvtz_oracle.json', '', nocc
    if args.molecule == 'water' and args.nwater == 8 and args.basis == 'cc-pvtz':       # the small case of the launch tests
        return 'h2o8_ccpvtz_oracle.json', '', nocc
    if args.molecule == 'taxol' and args.basis == 'def2-tzvp':
        return 'taxol_def2tzvp_oracle.json', 'syn_', 32
    return None


def _golden_density(nao, rank):
    """The seeded density of tools/gen_golden_*.py: D = 2 C C^T, C uniform in [-0.5, 0.5) / sqrt(nao) from RandomState(7)
    (oracle/golden_util.synthetic_orbitals restated: three lines of numpy, so that an N > 1 run imports nothing from oracle/)."""
    rng = np.random.RandomState(7)
    c = (rng.random_sample((nao, rank)) - 0.5) / np.sqrt(nao) * np.sqrt(2.0)
    cfull = np.zeros((nao, nao))
    cfull[:, :rank] = c
    occ = np.zeros(nao)
    occ[:rank] = 1.0
    return c.dot(c.T), cfull, occ


def _parity_golden(args, nao, nocc, get_jk):
    """In-run correctness evidence at ANY N, both launch modes: J and K of the seeded density of the committed
    oracle-only golden (tests/golden/, computed by the CPU oracle alone at full size) through the same product path that was
    timed - sharded, all-reduced / gathered - against the golden's fingerprints and 4096 sampled elements.  `get_jk(dm_tag)` is
    called on EVERY rank (it contains the collective); the comparison is returned for rank 0 to print."""
    case = _golden_case(args, nao, nocc)
    if case is None:
        return {'golden': None, 'why': 'no oracle-only J/K golden is committed for this workload (configs 3 and 4 have one)'}
    fname, pre, rank_d = case
    path = os.path.join(ROOT, 'tests', 'golden', fname)
    if not os.path.exists(path):
        return {'golden': None, 'why': '%s is missing' % fname}
    g = json.load(open(path))
    from pyscf_amd import lib
    dm, cfull, occ = _golden_density(nao, rank_d)
    vj, vk = get_jk(lib.tag_array(dm, mo_coeff=cfull, mo_occ=occ))
    vj, vk = np.asarray(vj).reshape(nao, nao), np.asarray(vk).reshape(nao, nao)
    rng = np.random.RandomState(int(g.get('sample_seed', 11)))
    n = len(g[pre + 'vk_sample'])
    ri, ci = rng.randint(0, nao, size=n), rng.randint(0, nao, size=n)
    out = {'golden': 'tests/golden/' + fname, 'density': g.get(pre + 'density', g.get('jk_density')), 'tol': 1e-9}
    ok = True
    for name, m in (('vj', vj), ('vk', vk)):
        fp = float(np.dot(np.cos(np.arange(m.size)), m.ravel()))
        err_s = float(np.abs(m[ri, ci] - np.array(g[pre + name + '_sample'])).max() / g[pre + name + '_absmax'])
        err_n = float(abs(np.linalg.norm(m) - g[pre + name + '_norm']) / g[pre + name + '_norm'])
        err_f = float(abs(fp - g[pre + name + '_fp']) / g[pre + name + '_norm'])
        out.update({'fp_' + name: fp, 'fp_%s_golden' % name: g[pre + name + '_fp'], 'max_rel_err_' + name: max(err_s, err_n, err_f)})
        ok = ok and max(err_s, err_n, err_f) < 1e-9
    out['max_rel_err'] = max(out['max_rel_err_vj'], out['max_rel_err_vk'])
    out['ok'] = bool(ok)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--nwater', type=int, default=32)
    ap.add_argument('--basis', default=None, help='default: cc-pvtz (water clusters), def2-tzvp (taxol)')
    ap.add_argument('--molecule', default='water', choices=['water', 'taxol'],
                    help="'water': (H2O)_nwater (configs 3 and 5); 'taxol': C47H51NO14 of config 4")
    ap.add_argument('--cpu-sample-rows', type=int, default=0, help='aux rows of the CPU baseline (0: all rows if host RAM allows, else half)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-threads', type=int, default=0, help='threads of the CPU baseline (default: all host cores)')
    ap.add_argument('--xc', default='b3lyp', help="XC functional of the secondary nr_rks timing ('' to skip)")
    ap.add_argument('--backend', default=None, help="torch.distributed backend ('nccl' = RCCL; 'gloo' for the "
                    "single-device self-test where several ranks share one GPU)")
    ap.add_argument('--tune', default='', help='comma list key=value for PAMD_set_tuning (A/B runs)')
    ap.add_argument('--k-square', default='auto', choices=['auto', 'off'], help="'off': no unpacked image - the K half transform "
                    "reads the packed rows (+ the diagonal-block side image, 14 %% of the tensor): what a rank without 2x the "
                    "tensor size of spare HBM runs")
    ap.add_argument('--layout', default='auto', choices=['auto', 'square', 'packed'], help="DF.layout: 'square' rows as the only copy of "
                    "the tensor (2x the packed bytes, r06), 'packed' rows (+ optional image), 'auto': square when the HBM budget allows")
    ap.add_argument('--j2-policy', default='auto', choices=['auto', 'overlap', 'serial', 'fused'], help='second J pass beside the SYRK on a '
                    'side stream, or in line before a re-tiled SYRK; auto: both timed once in the first (warm-up) build')
    ap.add_argument('--syrk-flags', type=int, default=-1, help='override DF.k_syrk_flags (A/B runs): 0 plain, 12 re-tiled + balanced')
    ap.add_argument('--single-process', action='store_true', help='N GPUs from ONE process through the C handle (PAMD_df_create_multi: '
                    'one host thread per device, peer gather + sum on device 0) instead of one rank per GPU under torch.distributed')
    ap.add_argument('--pmc', nargs='?', const='on', default='auto', choices=['auto', 'on', 'off'],
                    help="roofline.traffic measured IN THIS RUN: two extra rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters "
                    "alone with --kernel-trace) of a short child bench.py before the timed run (N = 1 only; adds ~1-2 minutes).  "
                    "'auto' (default, r05): on when rocprofv3 is on PATH; 'off': the committed profiles/<round>/pmc_summary.json")
    ap.add_argument('--no-pmc', dest='pmc', action='store_const', const='off')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--pmc-shard', default='', help=argparse.SUPPRESS)      # 'r,w': the child builds and contracts rank r's shard of w only
    args = ap.parse_args()
    if args.single_process:
        return single_process_main(args)
    if args.backend:
        os.environ['PAMD_DIST_BACKEND'] = args.backend

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Launched as plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run, rendezvous on 127.0.0.1) so that a scaling run can never silently fall back to one GPU.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: launch with `python bench.py --gpus N` (self-launching) or '
                         '`torchrun --nproc-per-node N bench.py --gpus N`' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)       # (several ranks may share a device only in the gloo self-test)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    backend = None
    grouped = world > 1 or 'RANK' in os.environ           # under torch.distributed.run a group exists even for one rank
    if grouped:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = os.environ.get('PAMD_DIST_BACKEND', 'nccl')        # 'nccl' = RCCL on ROCm
        if backend == 'nccl':
            if ndev < world:
                raise SystemExit('bench.py: %d ranks but %d visible GPUs (RCCL needs one device per rank; '
                                 '--backend gloo shares a device for the self-test)' % (world, ndev))
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        world = dist.get_world_size()          # n_gpus below = the ranks the process group really has
        # pre-flight: the first collective of the run is a 1-element all-reduce (RCCL communicator set-up happens here, with
        # a readable error, not inside the first J/K build) and every rank reports its free HBM
        one = torch.ones(1, dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(one)
        if int(one.item()) != world:
            raise SystemExit('bench.py: pre-flight all-reduce returned %r, expected %d' % (one.item(), world))

    pmc_live = None
    import shutil
    want_pmc = args.pmc == 'on' or (args.pmc == 'auto' and shutil.which('rocprofv3') is not None)
    if want_pmc and world == 1 and not args.pmc_child and not grouped:
        pmc_live = _pmc_passes(args)         # before this process holds any HBM: the child runs need the whole device
        if not pmc_live.get('FETCH_SIZE'):
            pmc_live = None                  # profiler missing / failed: the committed summary is used (and named) instead
        # the driver hands a finished child's HBM back with a delay: wait for it (the tensor's layout decision reads the free memory)
        for _ in range(100):
            f_, t_ = torch.cuda.mem_get_info(dev)
            if f_ > 0.97 * t_:
                break
            time.sleep(0.2)
    elif args.pmc == 'on' and world > 1 and rank == 0 and not args.pmc_child:
        # r06: at N > 1 rank 0 measures the traffic of ITS shard's launch on ITS device - a child run outside
        # the process group that builds and contracts rank 0's rows only (--pmc-shard 0,N) - while the other ranks build and wait at
        # the first barrier.  Opt-in (--pmc on): the default at N > 1 is the committed figure scaled by rows, labelled as such.
        pmc_live = _pmc_passes(args, shard=(0, world), device=dev_index)
        if not pmc_live.get('FETCH_SIZE'):
            pmc_live = None

    from pyscf_amd import gto, df, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df import df_jk
    from pyscf_amd.scf import hf

    if args.basis is None:
        args.basis = 'def2-tzvp' if args.molecule == 'taxol' else 'cc-pvtz'
    label = 'taxol C47H51NO14' if args.molecule == 'taxol' else '(H2O)_%d' % args.nwater
    mol = gto.M(atom=clusters.taxol() if args.molecule == 'taxol' else clusters.water_cluster(args.nwater), basis=args.basis)
    nao, nocc = mol.nao, mol.nelectron // 2
    dfobj = df.DF(mol)                      # aux basis by the reference's rule (cc-pvtz -> cc-pvtz-jkfit)
    if args.pmc_child and args.pmc_shard:
        dfobj._shard_override = tuple(int(v) for v in args.pmc_shard.split(','))     # one rank's rows, no collectives
    if args.syrk_flags >= 0:
        dfobj.k_syrk_flags = args.syrk_flags
    if args.k_square == 'off':
        dfobj.k_square = False
    if args.layout != 'auto':
        dfobj.layout = args.layout
    if args.xc:
        from pyscf_amd.dft.numint import estimate_ao_image_bytes
        dfobj.xc_image_hint = estimate_ao_image_bytes(mol)      # what RKS.density_fit() tells the tensor object (one HBM budget)
    dfobj.j2_policy = args.j2_policy
    dfobj.j2_tune = 'eager'             # the schedule is settled by trial builds in the set-up call, before the warm-up and the timed steps
    for kv in filter(None, args.tune.split(',')):
        k_, v_ = kv.split('=')
        lib.check(lib.load_library().PAMD_set_tuning(k_.encode(), int(v_)))
    # pre-flight, memory: this rank's packed shard must fit (the square image is optional: DF.k_square='auto' builds it only
    # when HBM allows); refuse with a clear message instead of an out-of-memory error inside the build
    from pyscf_amd.lib import comm
    naux_all = df.make_auxmol(mol, dfobj.auxbasis).nao_nr()
    l0_, l1_ = dfobj.shard_range(naux_all, dfobj.rank, dfobj.world_size)
    shard_gb = 8e-9 * (l1_ - l0_) * (nao * (nao + 1) // 2)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    preflight = {'device_count': ndev, 'hbm_free_GB': round(free_b / 1e9, 1), 'hbm_total_GB': round(total_b / 1e9, 1),
                 'shard_GB': round(float(shard_gb), 1), 'square_image_GB': round(2.0 * float(shard_gb), 1)}
    if shard_gb * 1e9 * 1.15 + (8 << 30) > free_b:
        raise SystemExit('bench.py: rank %d needs %.1f GB for its %d aux rows (+ work space) but only %.1f GB of HBM are free; '
                         'use more ranks (--gpus N)' % (rank, shard_gb, l1_ - l0_, free_b / 1e9))
    t0 = time.perf_counter()
    dfobj.build()
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    hbm_after = {'allocated': round(torch.cuda.memory_allocated(dev) * 1e-9, 1), 'free': round(torch.cuda.mem_get_info(dev)[0] * 1e-9, 1)}
    naux = dfobj.get_naoaux()
    npair = nao * (nao + 1) // 2
    naux_local = dfobj.tensor_shape()[0]
    square_layout = getattr(dfobj, '_layout', None) == 'square'

    # converged-like idempotent DM: Loewdin-orthogonalised random orbitals, seed 1
    s1e = hf.int1e_gpu(mol, dev)[0]
    rng = np.random.RandomState(1)
    x = rng.random_sample((nao, nao))
    w, v = np.linalg.eigh(x.T.dot(s1e).dot(x))
    c = x.dot(v / np.sqrt(w)).dot(v.T)
    mo_occ = np.zeros(nao)
    mo_occ[:nocc] = 2
    orbo = c[:, :nocc] * np.sqrt(2.0)
    dm = orbo.dot(orbo.T)
    dms_dev = torch.from_numpy(dm[None]).to(dev)
    orb_list = [df_jk.pad_orbitals(orbo, dev)]

    def step():
        # D was built as orbo orbo^T (as make_rdm1 does, pyscf/scf/hf.py:855-868): the first J pass is fused into the half transform
        return df_jk.get_jk_device(dfobj, dms_dev, orb_list, True, True, dm_from_orbitals=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()           # set-up, not a warm-up step: builds the lazy images of the tensor and lets DF.j2_policy = 'auto' time its two schedules
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    # a generation-2 garbage collection of this process (basis tables, torch: ~1 M container objects) takes 50 ms - measured in
    # r05 (profiles/r05/host_api_overlap.log), one about every eighth J/K call; it is interpreter jitter, not part of the step:
    # collect now and move what survives out of the collector's way (gc stays enabled)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()                          # ... and, as `timeit` does, no cyclic collection inside the timed regions (re-enabled below)
    ctimer = comm.CommTimer()
    comm.set_timer(ctimer)                # HIP events around every collective of the timed steps
    live_timer = df_jk.KernelTimer()      # ... and around every kernel launch, each on the stream it is launched on: the
    dfobj.kernel_timer = live_timer       # `roofline` durations are those of the TIMED steps (J overlapped with K as they run)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vjtril, vk = step()
    fence()
    dt = time.perf_counter() - t0
    comm.set_timer(None)
    dfobj.kernel_timer = None
    ksum_live = {k_: (tot_ / args.steps, cnt_ / float(args.steps)) for k_, (tot_, cnt_) in live_timer.summary().items()}
    comm_ms, comm_bytes = ctimer.total_ms() if ctimer.records else (0.0, 0)
    comm_info = {'backend': backend, 'collectives_per_step': len(ctimer.records) / max(args.steps, 1),
                 'comm_ms_per_step': round(comm_ms / max(args.steps, 1), 4) if grouped else None,
                 'bytes_per_step': int(comm_bytes / max(args.steps, 1)),
                 'what': '[J~ || K] packed f64 all-reduce (2 nao_pair doubles per density), HIP events on the launch stream'}
    if grouped and ctimer.records and comm_ms > 0:
        # algorithm bandwidth = message / time; bus bandwidth = what a ring moves over every link, 2 (N - 1) / N x that (the RCCL
        # convention); the yardstick: 7 xGMI links x ~153 GB/s per GPU, point to point - a ring is bound by ONE link per hop
        algbw = comm_bytes / (comm_ms * 1e-3) / 1e9
        comm_info.update({'algorithm_GBs': round(algbw, 2), 'bus_GBs': round(algbw * 2.0 * (world - 1) / max(world, 1), 2),
                          'xgmi_link_peak_GBs': 153.0, 'xgmi_links_per_gpu': 7})
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    if args.pmc_child:
        # the counters of the timed launches are all the parent wants from this run
        print(json.dumps({'pmc_child': True, 'ms_per_step': round(ms_per_step, 3), 'naux_local': int(naux_local)}))
        return
    jk_schedule = dict(getattr(dfobj, '_j2_policy_times', {'chosen': dfobj.j2_policy}))     # of the TIMED workload (the golden density
                                                                                          # below has its own shape and its own trial)

    # the reference's own timer ('df vj and vk', pyscf/df/df_jk.py:412) brackets with_df.get_jk(dm) with numpy in/out:
    # dm upload, J/K build, download and host unpack_tril.  Reported beside `value` (device-resident), never as it.
    dm_tag = lib.tag_array(dm, mo_coeff=c, mo_occ=mo_occ)
    dfobj.get_jk(dm_tag, hermi=1)
    fence()
    nh = max(1, min(args.steps, 5))
    host_calls = []
    dfobj.host_timing = []                    # r06: where each call spends its host time (df_jk.get_jk), printed beside the list
    for _ in range(nh):                       # every call fenced and timed by itself: the MEDIAN is reported (a single call now
        t0 = time.perf_counter()              # and then carries a 40 ms host hiccup - the list is in the line as well)
        vj_h, vk_h = dfobj.get_jk(dm_tag, hermi=1)
        fence()
        host_calls.append((time.perf_counter() - t0) * 1e3)
    host_api_ms = float(np.median(host_calls))
    host_breakdown, dfobj.host_timing = dfobj.host_timing, None
    gc.enable()
    host_fused = getattr(dfobj, '_last_fused', None)       # was the first J pass fused for the foreign (unpromised) tag?
    if world > 1:
        tmax = torch.tensor([host_api_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        host_api_ms = float(tmax.item())
        rows_t = torch.zeros(world, dtype=torch.int64, device=dev)
        rows_t[rank] = naux_local
        dist.all_reduce(rows_t)
        naux_per_rank = [int(v) for v in rows_t.cpu()]
    else:
        naux_per_rank = [int(naux_local)]
    if sum(naux_per_rank) != naux:
        raise SystemExit('bench.py: the aux rows of the ranks %r do not add up to naux = %d' % (naux_per_rank, naux))
    # in-run correctness evidence at any N: the committed oracle-only golden through the sharded + all-reduced product path
    parity_golden = _parity_golden(args, nao, nocc, lambda d: dfobj.get_jk(d, hermi=1))
    fence()

    # per-kernel durations (one extra, untimed step with HIP events around every launch)
    # (J and K are issued back-to-back on one stream for this pass so that each kernel is timed alone;
    # the timed steps above overlap the HBM-bound J kernels with the MFMA-bound SYRK on a second stream)
    dfobj.kernel_timer = df_jk.KernelTimer()
    dfobj.overlap_jk = False
    step()
    ksum_serial = dfobj.kernel_timer.summary()
    dfobj.kernel_timer = None
    dfobj.overlap_jk = True
    ksum = ksum_live                      # everything below (dominant kernel, roofline, TF/s) is priced on the timed steps

    # secondary figure (config 3 is DF-RKS B3LYP): one numint.nr_rks call per SCF iteration, grid blocks
    # dealt round-robin over the ranks; not part of `value`
    xc_info = None
    if args.xc:
        from pyscf_amd import dft
        t0 = time.perf_counter()
        grids = dft.Grids(mol).build()
        grid_s = time.perf_counter() - t0
        ni = dft.NumInt()
        dm_tag = lib.tag_array(dm, mo_coeff=c, mo_occ=mo_occ)
        ni.nr_rks(mol, grids, args.xc, dm_tag)
        fence()
        xc_times = []
        for _ in range(max(5, min(args.steps, 10))):           # median of >= 5 calls (one shot is dominated by noise)
            t0 = time.perf_counter()
            nel, exc, _ = ni.nr_rks(mol, grids, args.xc, dm_tag)
            fence()
            xc_times.append((time.perf_counter() - t0) * 1e3)
        xc_ms = float(np.median(xc_times))
        ni.kernel_timer = df_jk.KernelTimer()
        ni.nr_rks(mol, grids, args.xc, dm_tag)
        xs = ni.kernel_timer.summary()
        xc_info = {'xc': args.xc, 'nr_rks_ms_per_call': round(xc_ms, 1), 'nr_rks_ms_calls': [round(t, 1) for t in xc_times],
                   'nr_rks_kernels_ms_sum': round(sum(t for t, n_ in xs.values()), 2), 'ngrids': int(grids.size),
                   'grid_build_s': round(grid_s, 2), 'nelec': float(nel),
                   'kernels_ms': {k: round(t, 2) for k, (t, n_) in xs.items()}}
        if ni.sparse:
            plan = ni.sparse_plan(mol, grids, dft.libxc.xc_type(args.xc) == 'GGA')
            # roofline of the XC leg: flops the two MFMA products EXECUTE on the compact operands (16-column groups incl. their
            # padding, from the plan's own tile table) over their HIP-event time; sub_scale is the HBM-bound kernel of the leg
            ldh = plan.ld_host.astype(np.float64)
            npad = (nocc + 15) // 16 * 16
            fl = {'ao_dot_mo': 2.0 * plan.ncomp * plan.G * float(ldh.sum()) * npad, 'ao_dot_aow': 2.0 * plan.G * float((ldh ** 2).sum())}
            xr = {}
            for k_, f_ in fl.items():
                if k_ in xs and xs[k_][0] > 0:
                    xr[k_] = {'bound': 'mfma', 'executed_TFLOP': round(f_ * 1e-12, 4), 'ms': round(xs[k_][0], 3),
                              'achieved': round(f_ / xs[k_][0] * 1e-9, 2), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': round(f_ / xs[k_][0] * 1e-9 / FP64_MFMA_PEAK_TFLOPS, 4)}
            if 'scale_ao' in xs and xs['scale_ao'][0] > 0:
                by = 8.0 * (plan.ncomp + 1) * plan.G * float(ldh.sum())
                xr['scale_ao'] = {'bound': 'hbm', 'bytes': by, 'ms': round(xs['scale_ao'][0], 3), 'achieved': round(by / xs['scale_ao'][0] * 1e-6, 1),
                                  'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(by / xs['scale_ao'][0] * 1e-6 / HBM_PEAK_GBS, 4)}
            tot_fl = sum(fl.values())
            xr['leg'] = {'executed_TFLOP': round(tot_fl * 1e-12, 4), 'kernels_ms': round(sum(t for t, n_ in xs.values()), 2),
                         'ideal_ms_at_mfma_peak': round(tot_fl / (FP64_MFMA_PEAK_TFLOPS * 1e12) * 1e3, 2),
                         'frac_of_mfma_peak_over_kernel_time': round(tot_fl / (sum(t for t, n_ in xs.values()) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)}
            xc_info['roofline'] = xr
            xc_info['block_sparse'] = {'tile_points': plan.G, 'tiles': int(plan.nloc), 'cutoff': ni.sparse_cutoff,
                                       'ao_density_mean': round(plan.density, 4),
                                       'ao_density_sq_mean': round(plan.density2, 4),
                                       'compact_ao_GB': round(plan.ao_total * 8e-9, 2),
                                       'ao_cached_in_hbm': plan.ao_c is not None}

    if rank != 0:
        if grouped:
            dist.destroy_process_group()
        return

    # what the FP64 matrix pipe of THIS chip sustains under its power limit (measured here, same run): the kernels' k-loop without
    # its memory system on live operands (PAMD_mfma_f64_live) and a register-only stream on zero operands (the data-sheet case)
    practical = None
    try:
        import ctypes as _ct
        _so = lib.load_library()
        _out = torch.zeros(4, dtype=torch.float64, device=dev)
        _st = _ct.c_void_p(torch.cuda.current_stream().cuda_stream)

        def _timed(fn, flops):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            return flops / (e0.elapsed_time(e1) * 1e-3) / 1e12
        it_live, it_reg = 90000, 110000
        live = _timed(lambda: _so.PAMD_mfma_f64_live(_ct.c_void_p(_out.data_ptr()), 512, it_live, _st), 512 * 4 * it_live * 20 * 2048.0)
        zero = _timed(lambda: _so.PAMD_mfma_f64_peak(_ct.c_void_p(_out.data_ptr()), 512, it_reg // 20, 20, _ct.c_double(0.0), _st),
                      512 * 4 * (it_reg // 20) * 20 * 2048.0)
        practical = {'mfma_live_operands_TFLOPs': round(live, 2), 'mfma_register_stream_zero_operands_TFLOPs': round(zero, 2),
                     'what': 'PAMD_mfma_f64_live: 20 v_mfma_f64_16x16x4 per 9 fresh LDS fragment reads, no global memory, no '
                             'barriers, 2 x 256 workgroups resident, ~0.1 s; the chip lowers its clock under live FP64 matrix load'}
    except Exception as e:                       # never let a micro-benchmark break the metric line
        practical = {'error': str(e)}

    nocc_pad = orb_list[0][1]
    # algorithmic work of this rank's shard
    flops_e2 = 2.0 * naux_local * nao * nao * nocc          # X_L = B_L C
    flops_syrk = 2.0 * naux_local * nao * nao * nocc        # K += X^T X (full-square figure)
    bytes_j = 8.0 * naux_local * npair
    kern = {}
    for name, (tot, cnt) in ksum.items():
        kern[name] = {'ms_total': round(tot, 4), 'launches': cnt, 'ms_avg': round(tot / cnt, 4)}
    kern_serial = {name: {'ms_total': round(tot, 4), 'launches': cnt, 'ms_avg': round(tot / cnt, 4)}
                   for name, (tot, cnt) in ksum_serial.items()}
    dom = max(ksum, key=lambda k: ksum[k][0])
    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # profiles/r01/pmc_summary.json; KiB per launch as reported, corrected below where the access width needs it)
    traffic = None
    traffic_src = None
    try:
        pmc_dirs = sorted(d for d in os.listdir(os.path.join(ROOT, 'profiles'))
                          if os.path.exists(os.path.join(ROOT, 'profiles', d, 'pmc_summary.json')))
        traffic_src = 'profiles/%s/pmc_summary.json' % pmc_dirs[-1]
        pm_doc = json.load(open(os.path.join(ROOT, traffic_src)))
        pm = pm_doc['kernels']
        square = getattr(dfobj, '_cderi_sq', None) is not None
        cands = {'e2_symm': ['e2_sq2_kernel', 'e2_sq_kernel'] if square else ['e2_pk_kernel', 'e2_symm_kernel', 'e2_symm'],
                 'dgemm_tn': ['syrk_slots_kernel', 'gemm_tn_glds2_kernel', 'gemm_tn_glds_kernel'], 'vj_pass1': ['vj_pass1_rows_kernel'],
                 'vj_pass2': ['vj_pass2_sq_kernel', 'vj_pass2_kernel']}[dom]
        pk = [k for k in cands if k in pm][0]
        # gfx950: FETCH_SIZE reports half the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section).
        # Calibrated on kernels whose byte count is known (profiles/r01): vj_pass1 0.53x and vj_pass2 0.50x of their
        # 61.3 GB, e2_sq 0.54x; the packed-operand e2_symm (8-B/lane scattered reads) reports 1.00x
        # the profiled command also launches the kernel on fewer rows (parity sample, warm-up of the XC leg): the largest
        # launch is the full 2224-row one of the timed region
        rd = 1.0 if pk in ('e2_symm', 'e2_symm_kernel') else 2.0
        traffic = (rd * pm[pk]['FETCH_SIZE_KiB_per_launch_max'] + pm[pk]['WRITE_SIZE_KiB_per_launch_max']) * 1024.0
    except Exception:
        pass
    dtot, dcnt = ksum[dom]
    if pmc_live is not None:
        square = getattr(dfobj, '_cderi_sq', None) is not None
        names = {'e2_symm': ['e2_sq2_kernel', 'e2_sq_kernel'] if square else ['e2_pk_kernel', 'e2_symm_kernel'],
                 'dgemm_tn': ['syrk_slots_kernel', 'gemm_tn_glds2_kernel', 'gemm_tn_glds_kernel'], 'vj_pass1': ['vj_pass1_rows_kernel'],
                 'vj_pass2': ['vj_pass2_sq_kernel', 'vj_pass2_kernel', 'vj_pass2_wide_kernel']}[dom]
        hit = [k for k in names if k in pmc_live.get('FETCH_SIZE', {})]
        if hit:
            k0 = hit[0]
            rd = 1.0 if k0 == 'e2_symm_kernel' else 2.0          # gfx950 reports half the bytes of coalesced streaming reads
            traffic = (rd * pmc_live['FETCH_SIZE'][k0]['max'] + pmc_live.get('WRITE_SIZE', {}).get(k0, {'max': 0.0})['max']) * 1024.0
            traffic_src = ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 2 --pmc-child` '
                           '(largest launch of %s; FETCH_SIZE x %.0f: the gfx950 half-count of coalesced reads)' % (k0, rd))
            pm_doc = {'profiled_run': {'rows_per_e2_launch': float(naux_local / max(dcnt, 1))}}
    if traffic is not None and dom == 'e2_symm':
        # the PMC pass ran at N = 1; a rank's launch moves bytes in proportion to its aux rows per launch (recorded with the pass)
        traffic *= (naux_local / max(dcnt, 1)) / float(pm_doc.get('profiled_run', {}).get('rows_per_e2_launch', 2224.0))
    if dom in ('e2_symm', 'dgemm_tn'):
        fl = flops_e2 if dom == 'e2_symm' else flops_syrk
        ach = fl / (dtot * 1e-3) / 1e12
        kname = {'e2_symm': ('e2_sq2 (half transform on the square rows: the only copy of the tensor)' if square_layout else
                             'e2_sq2 (half transform on the unpacked image)') if getattr(dfobj, '_cderi_sq', None) is not None
                 else 'e2_symm (half transform)', 'dgemm_tn': 'gemm_tn_glds (SYRK)'}[dom]
        roofline = {'bound': 'mfma', 'kernel': kname, 'achieved': round(ach, 3), 'peak': FP64_MFMA_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(ach / FP64_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic,
                    'traffic_source': (traffic_src if pmc_live is not None else
                                       '%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not re-measured '
                                       'in this run - bench.py --pmc does)' % traffic_src) if traffic is not None else None,
                    'traffic_measured_in_run': bool(pmc_live is not None and traffic is not None),
                    'avg_launch_ms': round(dtot / dcnt, 4), 'launches_per_step': round(dcnt, 2),
                    'flops_per_step': fl}
        if traffic is not None and pmc_live is None and world > 1:
            roofline['traffic_source'] += ('; SCALED from the N = 1 launch by aux rows per launch (%d here): not measured at this N - '
                                           '`bench.py --gpus N --pmc on` measures it on rank 0' % int(naux_local / max(dcnt, 1)))
    else:
        ach = bytes_j / (dtot * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'traffic_source': traffic_src if traffic is not None else None,
                    'avg_launch_ms': round(dtot / dcnt, 4), 'launches_per_step': round(dcnt, 2)}
    # HBM GB/s of the J kernels (the metric's second figure): one algorithmic read of B per pass
    j_gbs = {}
    for name in ('vj_pass1', 'vj_pass2'):
        if name in ksum_serial:            # alone on the chip (the extra pass): the kernel's own HBM rate
            j_gbs[name] = round(bytes_j / (ksum_serial[name][0] * 1e-3) / 1e9, 1)
        if name in ksum:                   # as it runs in the timed steps, beside the MFMA-bound SYRK
            j_gbs[name + '_in_timed_steps'] = round(bytes_j / (ksum[name][0] * 1e-3) / 1e9, 1)
    k_tflops = {}
    if 'e2_symm' in ksum:
        k_tflops['e2_symm'] = round(flops_e2 / (ksum['e2_symm'][0] * 1e-3) / 1e12, 2)
    if 'dgemm_tn' in ksum:
        # the SYRK computes only the lower-triangular 128 x 128 tiles of K = X^T X: 'charged' is the SURVEY 8(d) full-square
        # figure 2 naux nao^2 nocc over its time (may exceed the 78.6 TF/s peak: symmetry, not speed); 'executed' counts the
        # tiles it really multiplies
        nt = -(-nao // 128)
        syrk_exec = flops_syrk * (nt * (nt + 1) / 2) / (float(nt) * nt)     # useful share: lower-triangular tiles of the nt^2
        k_tflops['dgemm_tn_charged'] = round(flops_syrk / (ksum['dgemm_tn'][0] * 1e-3) / 1e12, 2)
        k_tflops['dgemm_tn_executed'] = round(syrk_exec / (ksum['dgemm_tn'][0] * 1e-3) / 1e12, 2)
    else:
        syrk_exec = 0.0
    # whole-step roofline: the flops the step really executes (half transform + lower-triangular SYRK tiles) over ms_per_step
    step_exec = flops_e2 + syrk_exec
    step_roof = {'executed_TFLOP': round(step_exec / 1e12, 3), 'charged_TFLOP': round((flops_e2 + flops_syrk) / 1e12, 3),
                 'achieved_TFLOPs': round(step_exec / (ms_per_step * 1e-3) / 1e12, 2), 'peak_TFLOPs': FP64_MFMA_PEAK_TFLOPS,
                 'frac': round(step_exec / (ms_per_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                 'ideal_ms_at_peak': round(step_exec / (FP64_MFMA_PEAK_TFLOPS * 1e12) * 1e3, 2)}
    if practical and 'mfma_live_operands_TFLOPs' in practical:
        step_roof['frac_of_measured_mfma_ceiling'] = round(step_roof['achieved_TFLOPs'] / practical['mfma_live_operands_TFLOPs'], 4)
    step_roof['measured_mfma_ceiling'] = practical

    cpu = None
    parity = None
    if not args.no_cpu_baseline:
        # r06: at N > 1 rank 0 times the reference C on ITS SHARD's rows (a bounded sample of the same
        # workload: 1 / N of the rows) - `value` is extrapolated to all aux rows and says so, `shard_value` is what was timed
        from oracle import ref, ref_c
        ncore = args.cpu_threads or os.cpu_count()
        use_ref = ref_c.available()
        nrow = args.cpu_sample_rows
        if nrow <= 0:
            # all aux rows when the host has room for the tensor (+ the reference's 240-row work buffers), else half of them
            try:
                import psutil
                avail = psutil.virtual_memory().available
            except ImportError:
                avail = 0
                for line in open('/proc/meminfo'):
                    if line.startswith('MemAvailable'):
                        avail = int(line.split()[1]) * 1024
            nrow = naux_local if avail > 1.5 * 8.0 * naux_local * npair + (32 << 30) else -(-naux_local // 2)
        nrow = min(nrow, naux_local)
        sample = np.empty((nrow, npair))
        for r0 in range(0, nrow, 240):                   # download in the reference's block size (no 2x staging copy)
            sample[r0:r0 + 240] = dfobj.packed_rows(r0, min(r0 + 240, nrow)).cpu().numpy()
        if use_ref:
            # oracle/_ref/libref_dfjk.so = pyscf/lib/ao2mo/nr_ao2mo.c + pyscf/lib/np_helper/*.c compiled as they are, driven
            # call for call like pyscf/df/df_jk.py:329-381 (oracle/ref_c.get_jk): the reference's CPU path minus libcint
            ref_c.calibrate(sample, dm, c, mo_occ, nthreads=ncore)      # warms the thread pools; picks the faster BLAS build
            t0 = time.perf_counter()
            vj0, vk0, cpu_flops = ref_c.get_jk(sample, dm, c, mo_occ, blockdim=240, nthreads=ncore)
            cpu_s = time.perf_counter() - t0
            phases, kind = ref_c.get_jk.last_phases, 'reference'
            ncore = ref_c.get_jk.last_threads            # the threads really used (all host cores on the MKL variant, <= 64 on OpenBLAS)
            what = ("the reference's own C (AO2MOnr_e2_drv / AO2MOtranse2_nr_s2 / AO2MOmmm_bra_nr_s2 of lib/ao2mo/nr_ao2mo.c, NPdgemm / "
                    "NPdunpack_tril of lib/np_helper) compiled into oracle/_ref and called as pyscf/df/df_jk.py:329-381 does, blocks "
                    "of 240 aux rows; %s; the J line is numpy.matmul as in the reference; integral generation (libcint) is not "
                    "part of it" % ref_c.describe())
        else:
            ref.get_jk_rows_parallel(sample[:min(nrow, 64)], dm, c, mo_occ, nthreads=ncore)      # warm the BLAS threads
            t0 = time.perf_counter()
            vj0, vk0, cpu_flops = ref.get_jk_rows_parallel(sample, dm, c, mo_occ, nthreads=ncore)
            cpu_s = time.perf_counter() - t0
            phases, kind = getattr(ref.get_jk_rows_parallel, 'last_phases', None), 'port'
            what = ('oracle/ref.get_jk_rows_parallel (numpy restatement of df_jk.py:329-381; oracle/_ref was not built)')
        cpu = {'value': round(cpu_s / nrow * naux * 1e3, 1),
               'unit': 'ms/iter' + ('' if nrow == naux else ' (extrapolated from %d to all %d aux rows)' % (nrow, naux)),
               'cores': ncore, 'host_cores': os.cpu_count(), 'kind': kind, 'host_gflops': round(cpu_flops / cpu_s / 1e9, 1), 'phases_s': phases,
               'blas_calibration': getattr(ref_c, 'calibration', None) if use_ref else None,
               'sample': '%s; %d of %d aux rows of the GPU-built tensor%s: %.2f s' % (
                   what, nrow, naux, '' if world == 1 else " (rank 0's shard of %d ranks)" % world, cpu_s)}
        if world > 1:
            cpu.update({'shard_value': round(cpu_s * 1e3, 1), 'shard_rows': int(nrow), 'shard_unit': "ms for rank 0's rows, as timed",
                        'extrapolation': 'value = shard_value x naux / shard_rows (x %.3f): the whole iteration on the host cores of '
                                         "rank 0's box" % (naux / float(nrow))})
        # parity at full size: the same rows through the HIP path
        sub = df.DF(mol)
        if square_layout:
            sub._cderi_sq, sub._sq_nao, sub._layout = dfobj._cderi_sq[:nrow], nao, 'square'
        else:
            sub._cderi_dev = dfobj._packed[:nrow]
        vjt, vkd = df_jk.get_jk_device(_Single(sub), dms_dev, orb_list, True, True)
        vj1 = lib.unpack_tril(vjt.cpu().numpy(), 1)[0]
        vk1 = vkd.cpu().numpy()[0]
        parity = {'rows': nrow, 'checker': kind, 'max_abs_err_vj': float(np.abs(vj1 - vj0).max()),
                  'max_abs_err_vk': float(np.abs(vk1 - vk0).max())}

    out = {
        'metric': 'ms per SCF iter (DF J/K build)', 'value': round(ms_per_step, 3), 'unit': 'ms',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s %s DF J/K build (aux %s), nao=%d naux=%d nocc=%d, B=%.1f GB in HBM%s'
                               % (label, args.basis, _aux_label(getattr(dfobj.auxmol, 'basis', 'auto')),
                                  nao, naux, nocc, 8e-9 * naux * npair,
                                  (' held as SQUARE rows (%.1f GB: the only copy, r06)' % (8e-9 * dfobj._cderi_sq.numel()) if square_layout else
                                   '' if getattr(dfobj, '_cderi_sq', None) is None else
                                   ' + unpacked image for the K half transform' + ('' if dfobj._cderi_sq.shape[0] == naux_local else
                                   ' (of %d of the %d aux rows: what fits)' % (dfobj._cderi_sq.shape[0], naux_local))) +
                                  ('' if getattr(dfobj, '_cderi_diag', None) is None else
                                   ' + diagonal-block image of %d packed rows (%.1f GB)' % (dfobj._cderi_diag.shape[0],
                                                                                            8e-9 * dfobj._cderi_diag.numel()))),
                   'parallelism': 'aux-index shards x%d + RCCL all-reduce' % world if world > 1 else 'single GPU',
                   'naux_local': naux_local, 'naux_per_rank': naux_per_rank, 'tensor_layout': getattr(dfobj, '_layout', None),
                   'hbm_after_build_GB': hbm_after},
        'value_host_api_ms': round(host_api_ms, 3), 'host_api_ms_calls': [round(t, 2) for t in host_calls], 'host_api_fused_j': host_fused,
        'host_api_breakdown_ms': host_breakdown,
        'roofline': roofline, 'roofline_step': step_roof,
        'cpu_baseline': cpu, 'comm': comm_info, 'preflight': preflight,
        'jk_schedule': jk_schedule,
        'kernels': kern, 'kernels_what': 'HIP events around every launch of the %d timed steps, per step (ms_total) and per launch '
                                         '(ms_avg); J runs overlapped with K there' % args.steps,
        'kernels_serial_pass': kern_serial, 'j_hbm_GBs': j_gbs, 'k_mfma_TFLOPs': k_tflops,
        'build_s': round(build_s, 2), 'parity_golden': parity_golden, 'parity_sample': parity, 'xc_path': xc_info,
    }
    print(json.dumps(out))
    if grouped:
        dist.destroy_process_group()
    if parity_golden.get('golden') and not parity_golden['ok']:
        raise SystemExit('bench.py: J/K of the golden density differ from tests/golden by %.3e (relative): the number above is '
                         'not a valid measurement' % parity_golden['max_rel_err'])


def _pmc_passes(args, shard=None, device=None):
    """{'FETCH_SIZE': {kernel: {'max', 'mean', 'n'}}, 'WRITE_SIZE': ...} in KiB per launch from two rocprofv3 --pmc passes of a short
    child run of this file (counters alone with --kernel-trace, as the profiling guide prescribes; never with other trace domains).
    shard = (r, w): the child builds and contracts rank r's rows of w only (what a rank of an N > 1 job launches), on HIP device
    `device`, outside any process group (the launcher's rendezvous variables are removed from its environment)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    short = ('e2_sq2_kernel', 'e2_sq_kernel', 'e2_pk_kernel', 'e2_symm_kernel', 'syrk_slots_kernel', 'gemm_tn_glds2_kernel',
             'gemm_tn_glds_kernel', 'vj_pass1_rows_kernel', 'vj_pass2_wide_kernel', 'vj_pass2_kernel', 'vj_pass2_sq_kernel', 'vj_pass1_sq_kernel')
    out = {}
    env = dict(os.environ, TMPDIR='/tmp')
    for k in list(env):
        if k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'ROLE_WORLD_SIZE', 'MASTER_ADDR',
                 'MASTER_PORT', 'PAMD_DIST_BACKEND') or k.startswith('TORCHELASTIC_'):
            del env[k]
    if device is not None:
        vis = [v for v in env.get('HIP_VISIBLE_DEVICES', '').split(',') if v != '']
        env['HIP_VISIBLE_DEVICES'] = vis[device] if device < len(vis) else str(device)
    child = [sys.executable, os.path.abspath(__file__), '--steps', '2', '--warmup', '0', '--no-cpu-baseline', '--xc', '', '--pmc-child', '--no-pmc',
             '--nwater', str(args.nwater), '--molecule', args.molecule, '--k-square', args.k_square, '--j2-policy', args.j2_policy, '--layout', args.layout]
    if shard is not None:
        child += ['--pmc-shard', '%d,%d' % tuple(shard)]
    if args.basis:
        child += ['--basis', args.basis]
    if args.tune:
        child += ['--tune', args.tune]
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='pamd_pmc_', dir='/tmp')
        try:
            # own session: a timeout ends the profiler AND the python under it (a survivor would keep its HBM on this device)
            pr = subprocess.Popen(['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--'] + child,
                                  cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=180)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                raise
            agg = {}
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r['Counter_Name'] != counter:
                        continue
                    k = next((n for n in short if n in r['Kernel_Name']), None)
                    if k:
                        agg.setdefault((k, r['Dispatch_Id']), 0.0)
                        agg[(k, r['Dispatch_Id'])] += float(r['Counter_Value'])
            per = {}
            for (k, _), v in agg.items():
                per.setdefault(k, []).append(v)
            out[counter] = {k: {'max': max(v), 'mean': sum(v) / len(v), 'n': len(v)} for k, v in per.items()}
        except Exception as e:                         # a missing profiler must not break the metric line
            out[counter] = {}
            out['error'] = str(e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def single_process_main(args):
    """`bench.py --gpus N --single-process`: the same workload through the host-array C handle with a device LIST
    (PAMD_df_create_multi via pyscf_amd.df.native.NativeDF(devices=range(N))) - no torch, no torch.distributed, one process.
    A step is one `with_df.get_jk(dm)` with numpy arrays in and out, so `value` here INCLUDES the PCIe transfers of D, the
    orbitals, J and K (the device-resident figure is the default mode's `value`)."""
    os.environ.setdefault('PAMD_DF_J2_TUNE', 'eager')      # the handle settles its second-J-pass schedule by trial builds in the first (untimed) call

    from pyscf_amd import gto, lib
    from pyscf_amd.data import clusters
    from pyscf_amd.df.native import NativeDF
    if args.basis is None:
        args.basis = 'def2-tzvp' if args.molecule == 'taxol' else 'cc-pvtz'
    label = 'taxol C47H51NO14' if args.molecule == 'taxol' else '(H2O)_%d' % args.nwater
    mol = gto.M(atom=clusters.taxol() if args.molecule == 'taxol' else clusters.water_cluster(args.nwater), basis=args.basis)
    nao, nocc = mol.nao, mol.nelectron // 2
    ndev_visible = lib.load_library().PAMD_device_count()
    if ndev_visible < 1:
        raise SystemExit('bench.py --single-process: no HIP device')
    devices = [i % ndev_visible for i in range(args.gpus)]
    if ndev_visible < args.gpus:
        print('bench.py --single-process: %d parts on %d visible device(s): parts share devices (self-test layout)' % (args.gpus, ndev_visible),
              file=sys.stderr)
    t0 = time.perf_counter()
    obj = NativeDF(mol, devices=devices).build()
    build_s = time.perf_counter() - t0
    naux = obj.get_naoaux()
    npair = nao * (nao + 1) // 2
    rng = np.random.RandomState(1)
    c = np.linalg.qr(rng.random_sample((nao, nao)))[0]    # orthonormal columns: a density of the right rank (no overlap matrix
                                                          # without a device runtime in this process; the timing does not care)
    mo_occ = np.zeros(nao)
    mo_occ[:nocc] = 2
    dm = lib.tag_array((c[:, :nocc] * 2).dot(c[:, :nocc].T), mo_coeff=c, mo_occ=mo_occ, dm_from_orbitals=True)
    obj.get_jk(dm, hermi=1)                               # set-up (schedule timing inside the handle), then warm-up
    for _ in range(args.warmup):
        obj.get_jk(dm, hermi=1)
    import gc
    gc.collect()
    gc.freeze()                                           # (see main(): a 50 ms generation-2 collection is not part of a step)
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vj, vk = obj.get_jk(dm, hermi=1)
    dt = time.perf_counter() - t0
    gc.enable()
    ms = dt / args.steps * 1e3
    lay = obj.layout()
    tm = obj.last_timing()                                # of the last timed call: host clocks + HIP events inside the handle
    if sum(lay['part_rows']) != naux:
        raise SystemExit('bench.py --single-process: the aux rows of the parts %r do not add up to naux = %d' % (lay['part_rows'], naux))
    ndist = len(set(devices))
    if ndist > 1 and not lay['peer'] and os.environ.get('PAMD_DF_PEER', '') != '0':
        raise SystemExit('bench.py --single-process: %d distinct devices but the partial J/K travel through the HOST (peer access was '
                         'refused): this is not the xGMI path the number would be quoted for; set PAMD_DF_PEER=0 to time the host '
                         'bounce on purpose' % ndist)
    parity_golden = _parity_golden(args, nao, nocc, lambda d: obj.get_jk(d, hermi=1))
    # roofline of the dominant kernel (the half transform) per part, from the HIP events the handle records on its launch stream
    e2 = [t for t in tm['e2_ms']]
    worst = int(np.argmax(e2)) if e2 else 0
    roofline = None
    if e2 and e2[worst] > 0:
        fl = 2.0 * lay['part_rows'][worst] * nao * nao * nocc
        ach = fl / (e2[worst] * 1e-3) / 1e12
        roofline = {'bound': 'mfma', 'kernel': 'half transform of part %d (the slowest part), HIP events on its launch stream inside the '
                    'handle (PAMD_df_last_timing)' % worst, 'achieved': round(ach, 3), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / FP64_MFMA_PEAK_TFLOPS, 4), 'traffic': None, 'ms': round(e2[worst], 3),
                    'flops_per_step': fl, 'e2_ms_per_part': [round(t, 3) for t in tm['e2_ms']],
                    'syrk_ms_per_part': [round(t, 3) for t in tm['syrk_ms']]}
    moved = [b for b in tm['push_bytes'] if b > 0]
    comm = {'what': 'every part pushes its packed [J~ | K] into its slot of the gather buffer on device %d from its own thread and '
                    'stream (hipMemcpyPeerAsync over xGMI where peer access is granted), fixed-order sum there, one download' % devices[0],
            'peer_copies': bool(lay['peer']) and ndist > 1, 'distinct_devices': ndist,
            'bytes_per_part': tm['push_bytes'], 'push_ms_per_part': [round(t, 3) for t in tm['push_ms']],
            'sum_download_ms': round(tm['sum_download_ms'], 3), 'compute_ms_per_part': [round(t, 2) for t in tm['compute_ms']],
            'algorithm_GBs_per_link': round(max(moved) / (max(p for p, b in zip(tm['push_ms'], tm['push_bytes']) if b > 0) * 1e-3) / 1e9, 2)
            if moved else None,
            'xgmi_link_peak_GBs': 153.0, 'xgmi_links_per_gpu': 7,
            'note': 'push_ms is a host clock around pack + copy + stream synchronisation of one part (it includes the wait for the '
                    "part's own kernels to drain); with all parts on ONE device nothing crosses a link and bytes are 0"}
    cpu = None
    parity = None
    if not args.no_cpu_baseline and args.gpus == 1:
        cpu, parity = _cpu_baseline_native(args, obj, dm, c, mo_occ, nao, naux, npair)
    xc_info = None
    if args.xc:
        # the XC leg through the host-array handle over the same device list (grid tiles dealt round-robin over the parts)
        from pyscf_amd.dft.native import NativeGrids, NativeNumInt
        t0 = time.perf_counter()
        grids = NativeGrids(mol, device=devices[0]).build()
        grid_s = time.perf_counter() - t0
        ni = NativeNumInt(devices=devices)
        t0 = time.perf_counter()
        ni.nr_rks(mol, grids, args.xc, dm)                # builds the block-sparse plans of all parts
        plan_s = time.perf_counter() - t0
        xt = []
        for _ in range(max(5, min(args.steps, 10))):
            t0 = time.perf_counter()
            nel, exc, _v = ni.nr_rks(mol, grids, args.xc, dm)
            xt.append((time.perf_counter() - t0) * 1e3)
        xc_info = {'xc': args.xc, 'nr_rks_ms_per_call': round(float(np.median(xt)), 1), 'nr_rks_ms_calls': [round(t, 1) for t in xt],
                   'ngrids': int(grids.size), 'grid_build_s': round(grid_s, 2), 'plan_build_s': round(plan_s, 2), 'nelec': float(nel),
                   'plan': ni.plan_info(mol, grids, args.xc), 'what': 'NativeNumInt(devices=...).nr_rks with numpy in / out (PAMD_xc_nr_rks)'}
        # roofline of the XC leg from the HIP events the handle records around its kernels (slowest part); with several parts the
        # executed flops are those of ALL parts, so the fraction is priced against parts x the peak
        xt_ = ni.last_timing(mol, grids)
        nd_ = len(set(devices))
        xr = {}
        for k_ in ('ao_dot_mo', 'ao_dot_aow'):
            if xt_['ms'][k_] > 0:
                ach_ = xt_['flops'][k_] / xt_['ms'][k_] * 1e-9
                xr[k_] = {'bound': 'mfma', 'executed_TFLOP': round(xt_['flops'][k_] * 1e-12, 4), 'ms': round(xt_['ms'][k_], 3),
                          'achieved': round(ach_, 2), 'peak': FP64_MFMA_PEAK_TFLOPS * nd_, 'unit': 'TFLOP/s',
                          'frac': round(ach_ / (FP64_MFMA_PEAK_TFLOPS * nd_), 4)}
        if xt_['ms']['scale_ao'] > 0:
            gbs_ = xt_['scale_bytes'] / xt_['ms']['scale_ao'] * 1e-6
            xr['scale_ao'] = {'bound': 'hbm', 'bytes': xt_['scale_bytes'], 'ms': round(xt_['ms']['scale_ao'], 3), 'achieved': round(gbs_, 1),
                              'peak': HBM_PEAK_GBS * nd_, 'unit': 'GB/s', 'frac': round(gbs_ / (HBM_PEAK_GBS * nd_), 4)}
        if nd_ < args.gpus:
            xr['note'] = ('%d parts share %d device(s) (self-test layout): the executed flops are those of ALL parts, the time is one '
                          "part's - fractions above are not meaningful here" % (args.gpus, nd_))
        xc_info['kernels_ms'] = {k_: round(v_, 3) for k_, v_ in xt_['ms'].items()}
        xc_info['roofline'] = xr
        ni.reset()
    nt = -(-nao // 128)
    step_exec = 2.0 * naux * nao * nao * nocc * (1.0 + (nt * (nt + 1) / 2) / (float(nt) * nt))
    out = {'metric': 'ms per SCF iter (DF J/K build)', 'value': round(ms, 3), 'unit': 'ms', 'n_gpus': len(set(devices)),
           'parts': args.gpus, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': False,
           'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'launch': 'single process: PAMD_df_create_multi, one host thread per part, peer gather + fixed-order sum on device %d' % devices[0],
           'value_includes': 'host -> device copies of D and the orbitals and device -> host copies of J and K (numpy in / out)',
           'config': {'workload': '%s %s DF J/K build, nao=%d naux=%d nocc=%d, B=%.1f GB over %d part(s)' % (
                          label, args.basis, nao, naux, nocc, 8e-9 * naux * npair, args.gpus),
                      'parallelism': 'aux-index shards x%d in one process (C handle)' % args.gpus, 'devices': devices,
                      'naux_per_rank': lay['part_rows'], 'layout': lay},
           'roofline': roofline,
           'roofline_step': {'executed_TFLOP': round(step_exec / 1e12, 3), 'achieved_TFLOPs': round(step_exec / (ms * 1e-3) / 1e12, 2),
                             'peak_TFLOPs': FP64_MFMA_PEAK_TFLOPS * len(set(devices)),
                             'frac': round(step_exec / (ms * 1e-3) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * len(set(devices))), 4)},
           'comm': comm, 'cpu_baseline': cpu, 'parity_golden': parity_golden, 'parity_sample': parity,
           'build_s': round(build_s, 2), 'xc_path': xc_info,
           'checksum': {'fp_vj': lib.fp(vj), 'fp_vk': lib.fp(vk)}}
    print(json.dumps(out))
    obj.reset()
    if parity_golden.get('golden') and not parity_golden['ok']:
        raise SystemExit('bench.py: J/K of the golden density differ from tests/golden by %.3e (relative): the number above is '
                         'not a valid measurement' % parity_golden['max_rel_err'])


def _cpu_baseline_native(args, obj, dm, c, mo_occ, nao, naux, npair):
    """The reference's own C (oracle/_ref) on the host cores over the rows exported from the handle (single-process mode, N = 1):
    the same leg as the default mode's `cpu_baseline` + `parity_sample`."""
    from oracle import ref, ref_c
    ncore = args.cpu_threads or os.cpu_count()
    nrow = args.cpu_sample_rows
    if nrow <= 0:
        avail = 0
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                avail = int(line.split()[1]) * 1024
        nrow = naux if avail > 1.5 * 8.0 * naux * npair + (32 << 30) else -(-naux // 2)
    nrow = min(nrow, naux)
    sample = np.empty((nrow, npair))
    from pyscf_amd.df import native as _nat
    import ctypes as _ct
    for r0 in range(0, nrow, 240):
        r1 = min(r0 + 240, nrow)
        _nat._check(_nat.load().PAMD_df_export_cderi(obj._h, _ct.c_int(r0), _ct.c_int(r1), sample[r0:r1].ctypes.data_as(_ct.c_void_p)))
    if ref_c.available():
        ref_c.calibrate(sample, dm, c, mo_occ, nthreads=ncore)
        t0 = time.perf_counter()
        vj0, vk0, cpu_flops = ref_c.get_jk(sample, dm, c, mo_occ, blockdim=240, nthreads=ncore)
        cpu_s = time.perf_counter() - t0
        phases, kind, ncore = ref_c.get_jk.last_phases, 'reference', ref_c.get_jk.last_threads
        what = "the reference's own C (oracle/_ref, pyscf/lib/ao2mo/nr_ao2mo.c + np_helper), %s" % ref_c.describe()
    else:
        ref.get_jk_rows_parallel(sample[:min(nrow, 64)], dm, c, mo_occ, nthreads=ncore)
        t0 = time.perf_counter()
        vj0, vk0, cpu_flops = ref.get_jk_rows_parallel(sample, dm, c, mo_occ, nthreads=ncore)
        cpu_s = time.perf_counter() - t0
        phases, kind = getattr(ref.get_jk_rows_parallel, 'last_phases', None), 'port'
        what = 'oracle/ref.get_jk_rows_parallel (numpy restatement of df_jk.py:329-381; oracle/_ref was not built)'
    cpu = {'value': round(cpu_s / nrow * naux * 1e3, 1),
           'unit': 'ms/iter' + ('' if nrow == naux else ' (extrapolated from %d to all %d aux rows)' % (nrow, naux)),
           'cores': ncore, 'host_cores': os.cpu_count(), 'kind': kind, 'host_gflops': round(cpu_flops / cpu_s / 1e9, 1), 'phases_s': phases,
           'blas_calibration': getattr(ref_c, 'calibration', None) if kind == 'reference' else None,
           'sample': '%s; %d of %d aux rows of the GPU-built tensor: %.2f s' % (what, nrow, naux, cpu_s)}
    parity = None
    if nrow == naux:
        vj1, vk1 = obj.get_jk(dm, hermi=1)
        parity = {'rows': nrow, 'checker': kind, 'max_abs_err_vj': float(np.abs(vj1 - vj0).max()),
                  'max_abs_err_vk': float(np.abs(vk1 - vk0).max())}
    return cpu, parity


class _Single:
    """View of a DF object that never all-reduces (used for the single-rank parity sample)."""

    def __init__(self, obj):
        self.__dict__['_o'] = obj

    def __getattr__(self, k):
        if k == 'world_size':
            return 1
        return getattr(self.__dict__['_o'], k)


if __name__ == '__main__':
    main()
