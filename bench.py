#!/usr/bin/env python3
"""bench.py — DQMC walker-sweeps/s on BASELINE config 3 (attractive Hubbard 16x16, beta=8,
dtau=0.1: n=256, M=80, K=8; 32 walkers per MI355X), one rank per GPU.

One "step" = one sweep (2*slices `update` calls, src/flavors/DQMC/DQMC.jl:422-437) of every
walker resident on the rank.  Walkers are independent Markov chains, so N GPUs run N x 32
walkers (weak scaling) with no data-path collective; the only collective is the RCCL
all-reduce of the measurement accumulators every `measure_rate` sweeps.

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline:     fp64 MFMA roofline of the dominant kernel family (the batched GEMM), from
                per-launch HIP-event timings on the engine's stream
  cpu_baseline: the CPU oracle (a restatement of MonteCarlo.jl's algorithm, NOT the Julia
                package itself) timed on this host's cores on a bounded sample
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L, BETA, DTAU, SAFE_MULT, WALKERS_PER_GPU, BASE_SEED = 16, 8.0, 0.1, 10, 32, 123
FP64_PEAK_TFLOPS = 78.6  # AMD spec (vector = matrix); not in the in-container guides, see DESIGN.md


def flops_per_sweep(n, M, K, acc_rate):
    """algorithmic flops of ONE walker-sweep of the reference algorithm, per block"""
    gemm = n ** 3 * (12 * M + 24 * K - 4)
    qr_trsm = n ** 3 * 24 * K
    rank1 = 4 * acc_rate * M * n ** 3
    return dict(total=gemm + qr_trsm + rank1, gemm=gemm, qr_trsm=qr_trsm, rank1=rank1)


def cpu_baseline(n_threads, sweeps_each, conf_seed=BASE_SEED):
    """Oracle chains, one per host core (the reference is one chain per core by construction)."""
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
    chains = []
    for w in range(n_threads):
        mc = O.OracleDQMC(L, "attractive", beta=BETA, delta_tau=DTAU, safe_mult=SAFE_MULT)
        mc.set_conf(O.random_conf(conf_seed + w, mc.N, mc.slices))
        mc.seed(conf_seed + w)
        chains.append(mc)

    def run(mc, fn, *a):
        getattr(mc, fn)(*a)

    def par(fn, *a):
        ts = [threading.Thread(target=run, args=(mc, fn) + a) for mc in chains]
        [t.start() for t in ts]
        [t.join() for t in ts]

    par("prepare")          # ctypes releases the GIL: the chains really run on separate cores
    t0 = time.time()
    par("sweeps", sweeps_each)
    dt = time.time() - t0
    return n_threads * sweeps_each / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--walkers", type=int, default=WALKERS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweeps", type=int, default=2)
    args = ap.parse_args()

    # stdout must carry exactly ONE JSON line: native libraries (the RCCL banner, HIP warnings) write
    # to fd 1 as well, so fd 1 is pointed at stderr for the run and the JSON goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # the env var exercises the RCCL path on one rank
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)  # torchrun sets these; a bare single-rank rehearsal does not
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = max(world, 1)

    import __graft_entry__ as g
    mc_amd = g.load_package()
    model = mc_amd.HubbardModelAttractive(L, 2)
    mc = mc_amd.DQMC(model, beta=BETA, delta_tau=DTAU, safe_mult=SAFE_MULT, n_walkers=args.walkers,
                     device_id=local_rank, seed=BASE_SEED,
                     first_walker=mc_amd.walker_range(rank, n_gpus, args.walkers)[0])
    n, M, K = mc.N, mc.p.slices, mc.p.slices // mc.p.safe_mult
    mc.prepare()
    acc_dev = torch.zeros(mc.accumulator_size(), dtype=torch.float64, device="cuda:%d" % local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        mc.sweep(1)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        mc.sweep(1)
        if (i + 1) % mc.p.measure_rate == 0:  # measurement sums + RCCL reduction (DQMC.jl:429-436)
            mc.accumulate_greens()
            mc.export_accumulators(acc_dev.data_ptr())
            mc_amd.reduce_accumulators(acc_dev, dist)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-kernel-family device time: same steps replayed with HIP events around every launch
    a0 = mc.analysis(0)
    mc.timing_enable(True)
    t_steps = min(args.steps, 3)
    mc.sweep(t_steps)
    tim = mc.timing()
    mc.timing_enable(False)
    a1 = mc.analysis(0)
    acc_rate = (a1.acc_local - a0.acc_local) / max(1, a1.prop_local - a0.prop_local)

    value = n_gpus * args.walkers * args.steps / dt
    F = flops_per_sweep(n, M, K, acc_rate)
    # algorithmic flops served by the GEMM kernel family per walker-sweep: the reference's GEMMs,
    # the rank-1 updates (flushed as GEMMs here) and the explicit-Q part of the UDTs (4/3 n^3 each,
    # 6K per sweep, formed here with compact-WY GEMMs)
    gemm_alg = F["gemm"] + F["rank1"] + (4.0 / 3.0) * n ** 3 * 6 * K
    gemm_ms, gemm_launches = tim["gemm"]
    # HBM-side traffic of one full GEMM launch: PMC passes cannot run inside this process; the
    # number comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE profile (gfx950 x2
    # FETCH correction applied there), see profiles/*_pmc_gemm.json
    traffic = None
    try:
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_gemm.json")))
        if pm and args.walkers == WALKERS_PER_GPU:
            traffic = json.load(open(pm[-1]))["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    achieved = gemm_alg * args.walkers * t_steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    out = {
        "metric": "DQMC sweeps/sec, 16x16 Hubbard beta=8 dtau=0.1; achieved % fp64 MFMA roofline",
        "value": value, "unit": "walker-sweeps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "attractive Hubbard 16x16, beta=8, dtau=0.1 (n=256, M=80, safe_mult=10), "
                               "%d walkers per MI355X" % args.walkers,
                   "walkers_per_gpu": args.walkers, "parallelism": "walkers sharded, %d rank(s)" % n_gpus,
                   "acceptance_rate": acc_rate},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_note": "bytes per full 256^3 x 32 GEMM launch, from profiles/*_pmc_gemm.json",
                     "kernel": "GEMM family: gemm_kernel<TA,TB> + gemm_flush_kernel (v_mfma_f64_16x16x4_f64)", "launches_per_sweep": gemm_launches / t_steps,
                     "avg_launch_us": gemm_ms * 1e3 / max(1, gemm_launches)},
        "whole_sweep": {"algorithmic_gflop_per_walker_sweep": F["total"] / 1e9,
                        "achieved_tflops": F["total"] * value / n_gpus / 1e12,
                        "frac_of_fp64_peak": F["total"] * value / n_gpus / 1e12 / FP64_PEAK_TFLOPS},
        "device_ms_per_sweep": {k: v[0] / t_steps for k, v in tim.items()},
        "launches_per_sweep": {k: v[1] / t_steps for k, v in tim.items()},
    }
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cores = min(len(os.sched_getaffinity(0)), 16)
        v, secs = cpu_baseline(cores, args.cpu_sweeps)
        out["cpu_baseline"] = {"value": v, "unit": "walker-sweeps/s", "cores": cores, "kind": "port",
                               "sample": "%d oracle chains (one per core) x %d sweeps of the same 16x16 beta=8 "
                                         "workload, %.1f s; restatement of MonteCarlo.jl's algorithm, not the "
                                         "Julia package" % (cores, args.cpu_sweeps, secs)}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    mc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
