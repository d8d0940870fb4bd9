#!/usr/bin/env python3
"""bench.py — DQMC walker-sweeps/s of the hot path, one rank per GPU.

Workloads (BASELINE.json `configs`), selected with --config:
  3 (default, the configuration the metric is quoted on): attractive Hubbard 16x16, beta=8, dtau=0.1
    (n=256, M=80, K=8), 32 walkers per MI355X — weak scaling (N GPUs run N x 32 walkers);
  4: repulsive Hubbard 16x16, beta=8: 256 walkers in total, split over WORLD_SIZE — strong scaling;
  5: attractive Hubbard 24x24, beta=20, dtau=0.05 (n=576, M=400, K=40), 256 walkers per GPU (64 GB of the 288 GB:
     one workgroup per matrix in the n > 256 QR fills the 256 CUs) — weak scaling.

One "step" = one sweep (2*slices `update` calls, src/flavors/DQMC/DQMC.jl:422-437) of every walker resident on the
rank.  Walkers are independent Markov chains: there is no data-path collective, only the RCCL all-reduce of the
measurement accumulators every `measure_rate` sweeps (inside the library: dqmc_reduce).

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline:     whole-sweep algorithmic flops (F = n^3 (12M + 48K - 4) + 4 a M n^3 per block) x rate against
                the fp64 MFMA peak, with a `kernels` list (device time of every kernel family taken with HIP events
                attached to the launches of the TIMED region, its share, bound and recomputable fraction)
  cpu_baseline: the CPU oracle (a restatement of MonteCarlo.jl's algorithm, NOT the Julia package itself) timed on all
                host cores of this box on a bounded sample, plus the same with its dense products routed to OpenBLAS
"""
import argparse
import json
import os
import sys
import threading
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")  # the "strong CPU" leg runs one single-threaded chain per core
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASE_SEED = 123
FP64_PEAK_TFLOPS = 78.6  # AMD spec, vector = matrix; reproduced by the in-library probe at full clock (64-cycle MFMA)
HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    3: dict(model="attractive", L=16, beta=8.0, dtau=0.1, walkers=32, total=None, scaling="weak", steps=10, warmup=2,
            name="attractive Hubbard 16x16, beta=8, dtau=0.1 (n=256, M=80, safe_mult=10)"),
    4: dict(model="repulsive", L=16, beta=8.0, dtau=0.1, walkers=None, total=256, scaling="strong", steps=4, warmup=1,
            name="repulsive Hubbard 16x16, beta=8, dtau=0.1 (2 blocks of n=256, M=80, safe_mult=10), 256 walkers in total"),
    5: dict(model="attractive", L=24, beta=20.0, dtau=0.05, walkers=256, total=None, scaling="weak", steps=2, warmup=1,
            name="attractive Hubbard 24x24, beta=20, dtau=0.05 (n=576, M=400, safe_mult=10)"),
}
SAFE_MULT = 10


def flops_per_sweep(n, nb, M, K, acc_rate):
    """algorithmic flops of ONE walker-sweep of the reference algorithm, split by the kernel
    family that serves them here (the four shares add up to F = nb n^3 (12M + 48K - 4 + 4 a M))"""
    n3 = float(nb) * n ** 3
    gemm = n3 * (12 * M + 24 * K - 4) + n3 * 8 * K   # reference GEMMs + explicit Q of the 6K UDTs (4/3 n^3 each)
    qr = n3 * 12 * K                                 # Householder factorisation + column norms, (4/3 + 2/3) n^3 x 6K
    trsm = n3 * 4 * K                                # rdivp!, n^3 x 4K
    rank1 = 4 * acc_rate * M * n3                    # accept_local!, 2 n^2 per accepted site
    return dict(total=gemm + qr + trsm + rank1, gemm=gemm, qr=qr, trsm=trsm, flush=rank1, sweep=0.0, misc=0.0)


def effective_cores():
    """host cores this process may actually use: the cgroup CPU quota when there is one (a GPU box hands out a share of
    its cores per GPU), else the affinity mask"""
    aff = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, min(aff, int(float(q) / float(p)))), aff
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, min(aff, q // p)), aff
    except Exception:
        pass
    return aff, aff


def cpu_baseline(cfg, n_threads, sweeps_each, use_blas, partial_updates=None):
    """Oracle chains, one per host core (the reference is one chain per core by construction).  Returns
    (walker-sweeps/s, seconds, description of the sample)."""
    import subprocess
    import tempfile
    from oracle import oracle as O
    path = None
    try:  # host-tuned build for the timing leg; falls back to the portable in-tree .so
        out = os.path.join(tempfile.gettempdir(), "libdqmc_oracle_native_%d.so" % os.getpid())
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native", "OUT=" + out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        path = out
    except Exception:
        path = None
    O.lib(path)
    if use_blas and not O.use_openblas_dgemm(True):
        return None
    chains = []
    for w in range(n_threads):
        mc = O.OracleDQMC(cfg["L"], cfg["model"], beta=cfg["beta"], delta_tau=cfg["dtau"], safe_mult=SAFE_MULT)
        mc.set_conf(O.random_conf(BASE_SEED + w, mc.N, mc.slices))
        mc.seed(BASE_SEED + w)
        chains.append(mc)

    def par(fn, *a):
        ts = [threading.Thread(target=lambda m=mc: getattr(m, fn)(*a)) for mc in chains]
        [t.start() for t in ts]
        [t.join() for t in ts]

    def updates(mc, k):
        for _ in range(k):
            mc.update()

    par("prepare")          # ctypes releases the GIL: the chains really run on separate cores
    t0 = time.time()
    if partial_updates:     # a bounded part of one sweep (config 5: one sweep is minutes of CPU time)
        ts = [threading.Thread(target=updates, args=(mc, partial_updates)) for mc in chains]
        [t.start() for t in ts]
        [t.join() for t in ts]
        dt = time.time() - t0
        frac = partial_updates / (2.0 * chains[0].slices)
        rate, what = n_threads * frac / dt, "%d of the %d updates of one sweep" % (partial_updates, 2 * chains[0].slices)
    else:
        par("sweeps", sweeps_each)
        dt = time.time() - t0
        rate, what = n_threads * sweeps_each / dt, "%d sweeps" % sweeps_each
    if use_blas:
        O.use_openblas_dgemm(False)
    return rate, dt, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU (overrides the configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweeps", type=int, default=5)
    ap.add_argument("--no-kernel-timing", action="store_true", help="no HIP events on the launches of the timed region")
    ap.add_argument("--cpu-leg", choices=("port", "blas"), default=None,
                    help="internal: run only this CPU baseline leg (child process, never touches the GPU)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.cpu_leg:  # child process of the GPU run: isolates the BLAS runtime (and any crash of it) from the bench
        r = cpu_baseline(cfg, args.cpu_threads, args.cpu_sweeps, args.cpu_leg == "blas", 20 if args.config == 5 else None)
        print(json.dumps(None if r is None else {"value": r[0], "secs": r[1], "what": r[2]}))
        return
    steps = args.steps if args.steps is not None else cfg["steps"]
    warmup = args.warmup if args.warmup is not None else cfg["warmup"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `bench.py --gpus N` (no launcher): start N fresh ranks, one per GPU, BEFORE anything in this process
        # touches the GPU (this parent never does), and pass rank 0's JSON line through
        import subprocess
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        p = subprocess.run(cmd, stdout=subprocess.PIPE)
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith('{"metric"')]
        if lines:
            print(lines[-1], flush=True)
        sys.exit(p.returncode if p.returncode else (0 if lines else 1))

    # stdout must carry exactly ONE JSON line: native libraries (the RCCL banner, HIP warnings) write
    # to fd 1 as well, so fd 1 is pointed at stderr for the run and the JSON goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = max(world, 1)
    if args.gpus != n_gpus and "WORLD_SIZE" in os.environ and rank == 0:
        print("bench.py: --gpus %d ignored, the launcher started %d rank(s)" % (args.gpus, n_gpus), file=sys.stderr)
    # BENCH_ONE_DEVICE: rehearsal of the N-rank path on a one-GPU box - every rank uses device 0, the process group is
    # gloo and the measurement reduction goes through the host-mediated export / import pair (RCCL needs one device
    # per rank); the JSON has the N-rank shape, the number means nothing
    one_dev = bool(os.environ.get("BENCH_ONE_DEVICE"))
    dev = 0 if one_dev else local_rank
    import torch
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # the env var exercises the RCCL path on one rank
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)  # torchrun sets these; a bare single-rank rehearsal does not
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    import __graft_entry__ as g
    mc_amd = g.load_package()
    if args.walkers is not None:
        walkers, first = args.walkers, rank * args.walkers
        total_walkers = walkers * n_gpus
    elif cfg["total"] is not None:  # fixed total, contiguous blocks of global walker ids
        total_walkers = cfg["total"]
        lo, hi = mc_amd.walker_block(rank, n_gpus, total_walkers)
        walkers, first = hi - lo, lo
    else:
        walkers, first = cfg["walkers"], rank * cfg["walkers"]
        total_walkers = walkers * n_gpus
    Model = mc_amd.HubbardModelAttractive if cfg["model"] == "attractive" else mc_amd.HubbardModelRepulsive
    model = Model(cfg["L"], 2)
    mc = mc_amd.DQMC(model, beta=cfg["beta"], delta_tau=cfg["dtau"], safe_mult=SAFE_MULT, n_walkers=walkers,
                     device_id=dev, seed=BASE_SEED, first_walker=first)
    n, M = mc.N, mc.p.slices
    K, nb = M // SAFE_MULT, (2 if cfg["model"] == "repulsive" else 1)
    mc.prepare()
    # measurement reduction: RCCL inside the library (dqmc_reduce).  Should the library's own communicator not come
    # up on this node, the reduction falls back to torch.distributed (same RCCL, same packed buffer on the device)
    # so that the run still completes; which path ran is reported in config.reduction.
    comm, reduction = None, "none (single rank)"
    acc_dev = None
    if dist is not None and one_dev:
        reduction = "dqmc_reduce_export -> gloo all_reduce -> dqmc_reduce_import (BENCH_ONE_DEVICE rehearsal)"
    elif dist is not None:
        try:
            comm = mc_amd.Communicator(dist, device_id=dev)
            reduction = "dqmc_reduce (ncclAllReduce inside libdqmc_hip.so)"
        except Exception as e:  # noqa: BLE001
            comm = None
            acc_dev = torch.zeros(mc.accumulator_size(), dtype=torch.float64, device="cuda:%d" % dev)
            reduction = "torch.distributed all_reduce of the exported accumulators (library communicator failed: %s)" % e

    def reduce_measurements():
        if one_dev and dist is not None:
            mc.reduce_host(dist)
        elif acc_dev is not None:
            mc.export_accumulators(acc_dev.data_ptr())
            mc_amd.reduce_accumulators(acc_dev, dist)
        else:
            mc.reduce(comm)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        mc.sweep(1)
    a0 = mc.analysis_sum()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        mc.sweep(1)
        if (i + 1) % mc.p.measure_rate == 0:  # measurement sums + RCCL reduction (DQMC.jl:429-436)
            mc.accumulate_greens()
            reduce_measurements()
    barrier()
    dt = time.perf_counter() - t0
    a1 = mc.analysis_sum()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_dev else "cuda:%d" % dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # Kernel pass: the same K steps once more with HIP events attached to every launch (drained after its own timer
    # stops).  Kept apart from the region `value` is taken from, because the events themselves cost ~4 us per launch
    # (about 2900 launches per sweep); the family times add up to less than THIS pass's time per step.
    tim, kp_ms = {}, None
    if not args.no_kernel_timing:
        mc.timing_enable(True)
        barrier()
        t1 = time.perf_counter()
        mc.sweep(steps)
        barrier()
        kp_ms = (time.perf_counter() - t1) / steps * 1e3
        tim = mc.timing()
        mc.timing_enable(False)
    acc_rate = (a1[1] - a0[1]) / max(1, a1[0] - a0[0])
    qr_fallbacks = mc.qr_fallbacks()
    device_errors = mc.device_errors()  # the device error word as read after the timed regions (0: no bounded wait ran out)

    value = total_walkers * steps / dt
    ms_per_step = dt / steps * 1e3
    F = flops_per_sweep(n, nb, M, K, acc_rate)
    whole_tflops = F["total"] * value / n_gpus / 1e12  # per GPU
    bounds = {"gemm": "mfma", "flush": "mfma", "qr": "latency", "trsm": "latency", "sweep": "latency", "misc": "hbm"}
    notes = {
        "gemm": "slab_chain_kernel (the safe_mult slice products of a stack interval in one launch, wrap_greens in one "
                "launch) + gemm_kernel<TA,TB> (the GEMMs of calculate_greens; of compact-WY Q only without the one-launch UDT)",
        "flush": "sweep_flush_lu_kernel, last chunk of a slice only (the other chunks are applied inside the fused "
                 "sweep launches): block-triangular solves + a K=64 update of G (MFMA); HBM side 16 n^2 B per unit "
                 "and launch",
        "qr": "udt_AVX_pivot! - n = 256, <= 32 units: qrb_udt_kernel, ONE launch per UDT (pre-pivoted blocked Householder QR, 8 "
              "workgroups per matrix own a panel each, block reflectors on MFMA, Q, D and T included; its flops here = "
              "factorisation + norms + explicit Q); otherwise qr_coop_kernel + qr_tail_kernel (pivoted, Q as GEMMs)",
        "trsm": "rdivp! and the compact-WY triangle (blocked substitution, MFMA)",
        "sweep": "sweep_lu4_kernel / sweep_fused_kernel: Metropolis decisions = conditional elimination of G[c,c] "
                 "(sequential site chain), fused with the flush of the previous chunk when the grid is co-resident; "
                 "its flops are the rank-1 updates applied inside the fused launches",
        "misc": "udt_finish, copies, propagation-error check",
    }
    # rank-1 flops by where they are applied: of the 2M * ceil(N/64) chunk flushes per sweep the `flush` family runs
    # only the stand-alone launches (the last chunk of every slice when the fused form is active); the others happen
    # inside sweep_fused_kernel and are credited to `sweep`
    # the one-launch UDT forms Q inside the factorisation kernel: the 4/3 n^3 of "Calculate Q" (UDT.jl:250-266) per UDT then
    # belong to the qr family (2K slice-sequence UDTs at call site 0, 2K + 2K inside calculate_greens at sites 1 and 2)
    udt_sites = mc.udt_one_launch_sites()
    qshare = sum(1 for b_ in range(3) if (udt_sites >> b_) & 1) / 3.0
    if qshare > 0:
        qflops = float(nb) * n ** 3 * 8 * K * qshare
        # ... and inside calculate_greens_AVX! the launch also forms the product that follows the decomposition (Ul Q, stack.jl:360;
        # Tl Q, :378) by carrying Ul' / Tl' instead of the identity through the reflectors: 2K products of 2 n^3 per call site
        qflops += float(nb) * n ** 3 * 4 * K * (((udt_sites >> 1) & 1) + ((udt_sites >> 2) & 1))
        F["gemm"] -= qflops
        F["qr"] += qflops
    if tim:
        chunks = 2.0 * M * ((n + 63) // 64) * steps
        share = min(1.0, tim.get("flush", (0.0, 0))[1] / chunks) if chunks else 1.0
        F["sweep"] = F["flush"] * (1.0 - share)
        F["flush"] = F["flush"] * share
    # Fields named pmc_* / mfma_busy_frac / traffic are NOT measured by this run: hardware counters need rocprofv3 passes of
    # their own, so they come from the committed profile files.  Each carries the commit its file was taken at
    # (roofline.pmc_commit) and `stale: true` when that is not the commit this library was built from.
    try:
        lib_commit = mc_amd.lib().dqmc_build_commit().decode()
        lib_hash = mc_amd.lib().dqmc_build_source_hash().decode()
    except Exception:
        lib_commit = lib_hash = "unknown"

    def stale(file_hash):
        """a committed profile is stale when the kernel sources it was taken from (csrc/Makefile: SOURCE_HASH, recorded in the
        file) are not the ones this library was built from; commits that touch only documents / tests / profiles keep it fresh"""
        return not (file_hash and lib_hash != "unknown" and str(file_hash) == lib_hash)
    pmc = {}
    try:  # PMC passes cannot run inside this process: committed rocprofv3 --pmc summary of the same workload
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        if pm and args.config == 3 and walkers == 32:
            pmc = json.load(open(pm[-1]))
            pmc["_file"] = os.path.basename(pm[-1])
    except Exception:
        pmc = {}

    def pmc_sum(prefixes, counter):
        tot, hit = 0.0, False
        for k, v in pmc.items():
            if isinstance(v, dict) and any(k.startswith(q) for q in prefixes) and counter in v:
                tot += v[counter]["mean_per_launch"]
                hit = True
        return tot if hit else None
    kernels = []
    sum_ms = 0.0
    for fam, (ms, launches) in sorted(tim.items(), key=lambda kv: -kv[1][0]):
        ms_sweep = ms / steps
        sum_ms += ms_sweep
        ent = {"family": fam, "ms_per_sweep": ms_sweep, "share_of_kernel_pass": ms_sweep / kp_ms if kp_ms else 0.0,
               "launches_per_sweep": launches / steps, "avg_launch_us": ms * 1e3 / max(1, launches),
               "bound": bounds.get(fam, "latency"), "note": notes.get(fam, "")}
        fl = F.get(fam, 0.0)
        if fl > 0 and ms > 0:
            ent["algorithmic_gflop_per_walker_sweep"] = fl / 1e9
            ent["achieved_tflops"] = fl * walkers * steps / (ms * 1e-3) / 1e12
            ent["frac_of_fp64_peak"] = ent["achieved_tflops"] / FP64_PEAK_TFLOPS
        if fam == "flush" and ms > 0:
            hb = 16.0 * n * n * walkers * nb * launches  # read + write of G per launch
            ent["algorithmic_GBps"] = hb / (ms * 1e-3) / 1e9
            ent["frac_of_hbm_peak"] = ent["algorithmic_GBps"] / HBM_PEAK_GBS
        if fam == "qr" and ms > 0:
            # north_star: achieved HBM GB/s on the QR panels.  SURVEY 8(d): 24 n^2 + 8 n bytes per UDT (read A; write
            # U, T, D); one "launch" of this family = one batched factorisation of all units
            hb = (24.0 * n * n + 8.0 * n) * walkers * nb * launches
            ent["algorithmic_GBps"] = hb / (ms * 1e-3) / 1e9
            ent["frac_of_hbm_peak"] = ent["algorithmic_GBps"] / HBM_PEAK_GBS
            qk = ("qrb_udt",) if pmc_sum(("qrb_udt",), "FETCH_SIZE") is not None else ("qr_coop", "qr_tail")
            fe, wr = pmc_sum(qk, "FETCH_SIZE"), pmc_sum(qk, "WRITE_SIZE")
            if fe is not None and wr is not None:  # KB per launch; gfx950 FETCH_SIZE x2 correction as an upper bound
                ent["pmc_hbm_bytes_per_batch"] = (2.0 * fe + wr) * 1024.0
                ent["pmc_GBps"] = ent["pmc_hbm_bytes_per_batch"] / (ms * 1e-3 / launches) / 1e9
                ent["pmc_frac_of_hbm_peak"] = ent["pmc_GBps"] / HBM_PEAK_GBS
                ent["pmc_source"] = pmc.get("_file")
                ent["pmc_commit"] = pmc.get("_commit")
                ent["stale"] = stale(pmc.get("_source_hash"))
        if fam == "gemm" and pmc:
            busy, cyc = pmc_sum(("slab_chain",), "SQ_VALU_MFMA_BUSY_CYCLES"), pmc_sum(("slab_chain",), "SQ_BUSY_CYCLES")
            if busy and cyc:
                # slab kernel: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader
                # engines (8 XCDs x 4): (busy / 1024) / (cyc / 32).  Cross-check: 0.74 for the slab kernel's 52 TF/s
                # against the 71 - 77 TF/s of the MFMA-only probe.  (Round-3 profiles before this fix divided by 4: > 1.)
                ent["mfma_busy_frac"] = busy / (32.0 * cyc)
                ent["pmc_source"] = pmc.get("_file")
                ent["pmc_commit"] = pmc.get("_commit")
                ent["stale"] = stale(pmc.get("_source_hash"))
        kernels.append(ent)
    # HBM-side traffic of one full GEMM launch: PMC passes cannot run inside this process; the number comes from the
    # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE profile (gfx950 x2 FETCH correction applied there)
    traffic, traffic_commit, traffic_hash = None, None, None
    try:
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm.json")))
        if pm and args.config == 3 and walkers == 32:
            tj = json.load(open(pm[-1]))
            traffic = tj["traffic_bytes_per_launch"]
            traffic_commit, traffic_hash = tj.get("commit"), tj.get("source_hash")
    except Exception:
        traffic = None
    dom = kernels[0] if kernels else None
    out = {
        "metric": "DQMC sweeps/sec, 16x16 Hubbard beta=8 dtau=0.1; achieved % fp64 MFMA roofline",
        "value": value, "unit": "walker-sweeps/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s, %d walkers per MI355X" % (cfg["name"], walkers), "config": args.config,
                   "walkers_per_gpu": walkers, "total_walkers": total_walkers,
                   "parallelism": "walkers sharded, %d rank(s)" % n_gpus, "acceptance_rate": acc_rate,
                   "reduction": reduction,
                   # a cooperative-QR launch that timed out and was redone by the guarded kernel is a FAULT of the fast
                   # path, not a slow run: must be 0; device_errors is the device error word read back after the timed
                   # regions (sweep-elimination and one-launch-UDT bounded waits)
                   "qr_fallbacks": qr_fallbacks, "device_errors": device_errors, "library_commit": lib_commit, "library_source_hash": lib_hash,
                   "udt_one_launch_sites": udt_sites},
        "roofline": {"bound": "mfma", "achieved": whole_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": whole_tflops / FP64_PEAK_TFLOPS,
                     "definition": "SURVEY 8(d): algorithmic flops of the reference algorithm per walker-sweep "
                                   "(%.3f GFLOP at the measured acceptance) x walker-sweeps/s per GPU" % (F["total"] / 1e9),
                     "traffic": traffic,
                     "pmc_commit": traffic_commit, "stale": (stale(traffic_hash) if traffic is not None else None),
                     "traffic_note": "HBM bytes (PMC FETCH_SIZE + WRITE_SIZE, guide corrections) of one launch of the most "
                                     "frequent MFMA kernel, see profiles/*_pmc_gemm.json (kernel, algorithmic bytes)",
                     "kernels": kernels, "kernel_ms_sum": sum_ms, "kernel_pass_ms_per_step": kp_ms,
                     "kernel_pass_note": "family times from HIP events attached to the launches of a second region of "
                                         "the same K steps; ms_per_step / value come from the event-free region",
                     "dominant_kernel": None if dom is None else {k: dom.get(k) for k in (
                         "family", "avg_launch_us", "launches_per_sweep", "bound", "achieved_tflops",
                         "frac_of_fp64_peak", "algorithmic_GBps", "frac_of_hbm_peak", "pmc_GBps",
                         "pmc_frac_of_hbm_peak")}},
    }
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        import subprocess
        cores, visible = effective_cores()

        def leg(kind, threads):
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", str(args.config), "--cpu-leg",
                                    kind, "--cpu-threads", str(threads), "--cpu-sweeps", str(args.cpu_sweeps)],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
                return json.loads(p.stdout.decode().strip().splitlines()[-1])
            except Exception:
                return None

        r = leg("port", cores)
        if r is not None:
            out["cpu_baseline"] = {"value": r["value"], "unit": "walker-sweeps/s", "cores": cores,
                                   "nproc": visible, "kind": "port",
                                   "cores_note": "cores = cgroup CPU quota of this box (cpu.max); nproc = visible CPUs",
                                   "sample": "%d oracle chains (one per usable core) x %s of the same "
                                             "workload, %.1f s; restatement of MonteCarlo.jl's algorithm, not the "
                                             "Julia package" % (cores, r["what"], r["secs"])}
            bl = min(cores, 64)  # the image's OpenBLAS serves at most 128 concurrent callers
            rb = leg("blas", bl)
            if rb is not None:
                out["cpu_baseline"]["strong_cpu"] = {
                    "value": rb["value"] * cores / bl, "unit": "walker-sweeps/s", "cores": cores,
                    "measured_on_cores": bl, "measured_value": rb["value"],
                    "sample": "%d chains with the dense products routed to OpenBLAS dgemm (scipy's BLAS, 1 thread per "
                              "chain), %s, %.1f s; QR / rank-1 updates stay literal; scaled linearly to %d cores"
                              % (bl, rb["what"], rb["secs"], cores)}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    mc.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
