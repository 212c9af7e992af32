#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native Davidson/GD+k path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete solve of the workload by hip_dprimme (user matvec = device
CSR SpMV, CGS orthogonalisation, projection update, fused Ritz/residual update,
Rayleigh-Ritz) from the same deterministic start vector to the target residual norm.
`value` = eigenpairs/s = K * numEvals / (time of the K timed solves), max over ranks.
The operator, start vector and all panels are resident in HBM before the timed region.

Workloads (BASELINE.json configs / north star):
  lap3d_2m   (default) configs[1]: 3-D 7-pt Laplacian 125x126x127 (n = 2 000 250), double,
             blockSize 1, 10 smallest, PRIMME_GD_plusK, eps = 1e-8*|A|, |A| = 12, CSR int32
  lap2d_10m  north-star headline: 2-D 5-pt Laplacian 3162x3163 (n = 10 001 406), same settings, |A| = 8
N > 1: the SAME problem, rows partitioned over the ranks (strong scaling), RCCL
all-reduce only for the <= 4 KB inner-product panels, neighbour halo exchange in the matvec.

Extra objects on the JSON line (rank 0):
  roofline      dominant kernel class by device time, measured live with HIP events on the
                solver's stream (hipk_prof_*) over one more solve of the same workload right
                after the timed steps: algorithmic HBM bytes per launch / average launch
                duration, against the 8 TB/s HBM3E peak.
  north_star    one more solve of lap2d_10m (the north-star headline) with its own roofline: eigenpairs/s,
                per-class GB/s, the combined "CSR SpMV + ortho" fraction of the 8 TB/s peak.
  cpu_baseline  (N == 1) the real reference (oracle/_ref, PRIMME 3.2 + MKL) on the host
                cores for a bounded number of outer iterations of the same solve,
                extrapolated with the iteration count the full solve needs.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # checkers.py: test infrastructure

WORKLOADS = {
    "lap3d_2m": dict(dims=(125, 126, 127), aNorm=12.0, desc="3-D 7-pt Laplacian 125x126x127 CSR"),
    "lap2d_10m": dict(dims=(3162, 3163), aNorm=8.0, desc="2-D 5-pt Laplacian 3162x3163 CSR"),
    "lap3d_small": dict(dims=(60, 61, 62), aNorm=12.0, desc="3-D 7-pt Laplacian 60x61x62 CSR (dev)"),
}
KERNEL_CLASSES = ["dots_kernel (TN inner products: CGS overlaps + V'W)", "project_kernel (CGS update + norm)",
                  "ritz_kernel class = ritz_cgs_kernel + ritz_ov_kernel + ritz_kernel (fused R=Wh-theta Vh with its overlaps; restart X=Vh, Y=Wh)", "csr_stream_kernel (CSR SpMV)"]
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="lap3d_2m", choices=sorted(WORKLOADS))
    ap.add_argument("--num-evals", type=int, default=10)
    ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--operator", default="csr", choices=["csr", "stencil"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true", help="skip the extra 10 M-row solve")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    from primme_amd import _ffi as F
    from primme_amd import Operator, problems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # PRIMME_AMD_BENCH_DIST=1: run the multi-rank plumbing (process group, RCCL communicator,
    # in-stream reductions) even with one rank — the only way to exercise it on a one-GPU box
    dist_path = world > 1 or bool(os.environ.get("PRIMME_AMD_BENCH_DIST"))
    if dist_path and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
        os.environ["PRIMME_AMD_FORCE_COMM"] = "1"
    torch.cuda.set_device(local_rank)
    lib = F.load_product()

    comm = None
    if dist_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            assert lib.primme_amd_comm_unique_id(buf) == 0
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        uid = uid.cuda()
        dist.broadcast(uid, 0)
        raw = bytes(uid.cpu().numpy().tobytes())
        comm = C.c_void_p()
        assert lib.primme_amd_comm_create(C.byref(comm), raw, rank, world) == 0

    def barrier():
        torch.cuda.synchronize()
        if dist_path:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    from primme_amd.api import Session

    def run_workload(name, steps, warmup, time_profiled_solve=False):
        """K timed solves of one workload (operator, start vector and panels resident in HBM before the
        timed region), then ONE more solve with HIP events around every launch of the hot kernel
        classes for the roofline.  time_profiled_solve: report the (single) profiled solve as the timed
        one — used for the long north-star solve so that it runs once, not twice."""
        wl = WORKLOADS[name]
        dims = wl["dims"]
        n = int(np.prod(dims))
        # row partition: contiguous slabs, remainder spread over the first ranks
        base, rem = divmod(n, world)
        nloc = base + (1 if rank < rem else 0)
        row0 = rank * base + min(rank, rem)
        if args.operator == "csr":
            rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
            op = Operator(n, csr=(rp, ci, va), row0=row0, nrows=nloc)
        else:
            op = Operator(n, stencil=tuple(list(dims) + [1] * (3 - len(dims))), row0=row0, nrows=nloc)
        v0 = problems.start_vector(n, row0=row0, nrows=nloc)
        kw = dict(numEvals=args.num_evals, method="GD_plusK", eps=args.eps, aNorm=wl["aNorm"], v0=v0,
                  return_evecs=False, numProcs=world, procID=rank)
        # the matrix stays resident through a persistent session
        sess = Session(op, comm=comm, dtype=np.float64)
        last = None
        for _ in range(warmup):
            last = sess.solve(**kw)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = sess.solve(**kw)
        barrier()
        elapsed = time.perf_counter() - t0
        # roofline leg (kept out of the timed steps: two event records per launch cost a few %)
        lib.hipk_prof_reset()
        lib.hipk_prof_enable(1)
        barrier()
        t0 = time.perf_counter()
        lastp = sess.solve(**kw)
        barrier()
        tprof = time.perf_counter() - t0
        lib.hipk_prof_enable(0)
        if time_profiled_solve:
            elapsed, steps, last = tprof, 1, lastp
        if dist_path:
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        sess.close()

        ok = last.ret == 0 and last.initSize == args.num_evals and bool(
            np.all(last.resNorms <= args.eps * wl["aNorm"] * (1 + 1e-12)))
        exact = problems.laplacian_eigenvalues(dims, args.num_evals)
        eval_err = float(np.max(np.abs(np.sort(last.evals) - exact)))

        # ---- roofline of the dominant kernel class (rank 0's launches) ----
        prof = []
        for cls in range(4):
            ms, launches, nbytes = C.c_double(), C.c_long(), C.c_double()
            lib.hipk_prof_get(cls, C.byref(ms), C.byref(launches), C.byref(nbytes))
            prof.append((ms.value, launches.value, nbytes.value))
        dom = max(range(4), key=lambda c: prof[c][0])
        ms, launches, nbytes = prof[dom]
        achieved = (nbytes / max(ms, 1e-12)) / 1e6 if launches else 0.0   # bytes/ms -> GB/s
        # HBM bytes per launch from the PMC counters cannot be collected from inside this process:
        # they come from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over the
        # same workload, calibrated and summarised in profiles/ (null when no such pass is on file)
        traffic, tsrc = None, None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pj = pj.get(name, pj if pj.get("workload") == name else {})
            if pj.get("workload") == name and dom == 2 and world == 1 and pj.get("ritz_class_hbm_bytes_per_launch"):
                # a figure from a SEPARATE profiled process (PMC counters cannot be read from inside this one): labelled so
                traffic = round(pj["ritz_class_hbm_bytes_per_launch"])
                tsrc = "NOT measured in this run: " + pj.get("source", "profiles/pmc_traffic.json (separate --pmc passes, bytes per launch)")
        except Exception:
            pass
        # the inner loop the north star names: CSR SpMV + orthogonalisation (+ the fused residual /
        # projection passes that are part of the same iteration): all four classes together
        tot_ms = sum(p_[0] for p_ in prof)
        tot_bytes = sum(p_[2] for p_ in prof)
        so_ms = prof[3][0] + prof[1][0] + prof[0][0]
        so_bytes = prof[3][2] + prof[1][2] + prof[0][2]
        roofline = {
            "kernel": KERNEL_CLASSES[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
            "launches": launches, "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
            "alg_bytes_per_launch": round(nbytes / max(launches, 1)),
            "all_kernels": {KERNEL_CLASSES[c].split(" ")[0]: {
                "ms": round(prof[c][0], 2), "launches": prof[c][1],
                "GBps": round((prof[c][2] / max(prof[c][0], 1e-12)) / 1e6, 1)} for c in range(4)},
            "inner_loop_all_classes": {"GBps": round(tot_bytes / max(tot_ms, 1e-12) / 1e6, 1),
                                       "frac": round(tot_bytes / max(tot_ms, 1e-12) / 1e6 / HBM_PEAK_GBS, 4),
                                       "kernel_ms_per_solve": round(tot_ms, 1)},
            "spmv_plus_ortho": {"GBps": round(so_bytes / max(so_ms, 1e-12) / 1e6, 1),
                                "frac": round(so_bytes / max(so_ms, 1e-12) / 1e6 / HBM_PEAK_GBS, 4),
                                "classes": "csr spmv + CGS update (project) + TN inner products"},
        }
        res = {
            "value": round(steps * args.num_evals / elapsed, 4), "ms_per_step": round(1e3 * elapsed / steps, 3), "steps": steps,
            "config": {"workload": f"{name}: {wl['desc']}, n={n}, {args.num_evals} smallest, GD+k, "
                                   f"blockSize 1, eps={args.eps}*|A|, |A|={wl['aNorm']}, operator={args.operator}",
                       "partition": f"rows/{world}", "converged": ok, "max_eval_error_vs_analytic": eval_err,
                       "outer_iterations": last.stats["numOuterIterations"], "matvecs": last.stats["numMatvecs"],
                       "restarts": last.stats["numRestarts"],
                       "us_per_outer_iteration": round(1e6 * elapsed / steps / max(1, last.stats["numOuterIterations"]), 2)},
            "roofline": roofline}
        return res, last, dims, wl, n

    main_res, last, dims, wl, n = run_workload(args.workload, args.steps, args.warmup)
    out = {
        "metric": "eigenpairs/sec to target resNorm", "value": main_res["value"],
        "unit": "eigenpairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": main_res["config"], "roofline": main_res["roofline"],
    }
    if not args.no_north_star and args.workload != "lap2d_10m":
        # BASELINE.json's north-star headline: the 10 M-row 5-point Laplacian, one solve, profiled
        ns, _, _, _, _ = run_workload("lap2d_10m", 1, 0, time_profiled_solve=True)
        out["north_star"] = {"metric": "eigenpairs/sec to target resNorm", "value": ns["value"], "unit": "eigenpairs/s",
                             "seconds_per_solve": round(ns["ms_per_step"] / 1e3, 3), "n_gpus": world,
                             "note": "one solve, timed with the per-launch HIP events of the roofline leg enabled",
                             "config": ns["config"], "roofline": ns["roofline"]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--dims", *[str(d) for d in dims],
                   "--num-evals", str(args.num_evals), "--eps", str(args.eps), "--anorm", str(wl["aNorm"]),
                   "--max-matvecs", "60" if n > 5_000_000 else "150",
                   "--total-iterations", str(last.stats["numOuterIterations"])]
            if args.cpu_threads:
                cmd += ["--threads", str(args.cpu_threads)]
            r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            cb = json.loads(r.stdout.strip().splitlines()[-1])
            out["cpu_baseline"] = {
                "value": round(cb["value"], 5), "unit": "eigenpairs/s", "cores": cb["cores"], "kind": "reference",
                "sample": f"first {cb['sample_outer_iterations']} outer iterations of the same solve by the real "
                          f"reference dprimme (PRIMME 3.2 + MKL, OpenMP CSR matvec) = {cb['sample_seconds']:.1f} s; "
                          f"extrapolated to the {last.stats['numOuterIterations']} iterations the solve needs",
                "seconds_per_outer_iteration": cb["seconds_per_outer_iteration"],
                "sample_phase_seconds": {k: cb[k] for k in ("timeMatvec", "timeOrtho", "timeDense")}}
        except Exception as e:  # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "eigenpairs/s", "cores": 0, "kind": "reference",
                                   "sample": f"failed: {e!r}"}
    if comm is not None:
        lib.primme_amd_comm_destroy(comm)
    if rank == 0:
        print(json.dumps(out))
    if dist_path:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
