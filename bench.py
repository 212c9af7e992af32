#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native Davidson/GD+k path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete solve of the workload by hip_dprimme (user matvec = device CSR SpMV, CGS
orthogonalisation, projection update, fused Ritz/residual update, Rayleigh-Ritz) from the same deterministic
start vector to the target residual norm.  `value` = eigenpairs/s = steps * numEvals / (time of the timed
solves), max over ranks.  The operator, start vector and all panels are resident in HBM before the timed region.

Workloads (BASELINE.json north star / configs):
  lap2d_10m  (default) the configuration BASELINE.json's north_star quotes its target on: 2-D 5-pt Laplacian
             3162x3163 (n = 10 001 406), double, blockSize 1, 10 smallest, PRIMME_GD_plusK, eps = 1e-8*|A|, |A| = 8,
             CSR int32.  One solve takes ~27 s on one MI355X (30 846 outer iterations), so the timed solves are
             capped to about --budget-s seconds (default 180): steps = min(K, max(1, floor(budget / t_solve))) and
             warm-up = min(W, 1); `steps` / `warmup` on the JSON line are the numbers actually run, the requested
             ones are in config.  The timed solves run WITHOUT the per-launch events of the roofline leg.
  lap3d_2m   configs[1]: 3-D 7-pt Laplacian 125x126x127 (n = 2 000 250), same settings, |A| = 12; rides along as
             the `configs1` object (3 timed solves) when it is not the main workload.
N > 1: the SAME problem, rows partitioned over the ranks (strong scaling), the <= 4 KB inner-product panels
reduced over the library's communicator (peer-to-peer mailboxes / RCCL, see include/primme_amd_comm.h),
neighbour halo exchange in the matvec.  Launched by torch.distributed.run (RANK/WORLD_SIZE in the environment)
or, when WORLD_SIZE is unset and --gpus N > 1, by this script itself: it re-executes N ranks of itself.  It
refuses to run N ranks on fewer than N GPUs (PRIMME_AMD_BENCH_SHARE_GPU=1: ranks share devices over the
peer-to-peer transport — a functional experiment, never a scaling number; the JSON line says so).

Extra objects on the JSON line (rank 0):
  roofline      dominant kernel class by device time, measured live with HIP events on the solver's stream
                (hipk_prof_*) over one more solve of the same workload right after the timed steps: algorithmic
                HBM bytes per launch / average launch duration, against the 8 TB/s HBM3E peak and against the
                read / copy bandwidth measured in this process (frac_of_measured_stream).
  configs1      BASELINE configs[1] with its own roofline.
  cpu_baseline  (N == 1) the real reference (oracle/_ref, PRIMME 3.2 + MKL) on the host cores for a bounded
                number of outer iterations of the same solve, extrapolated with the iteration count the full
                solve needs.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "lap3d_2m": dict(dims=(125, 126, 127), aNorm=12.0, desc="3-D 7-pt Laplacian 125x126x127 CSR"),
    "lap2d_10m": dict(dims=(3162, 3163), aNorm=8.0, desc="2-D 5-pt Laplacian 3162x3163 CSR"),
    "lap3d_small": dict(dims=(60, 61, 62), aNorm=12.0, desc="3-D 7-pt Laplacian 60x61x62 CSR (dev)"),
}
KERNEL_CLASSES = ["dots_kernel (TN inner products: CGS overlaps + V'W)", "project_kernel (CGS update + norm)",
                  "ritz_kernel class = ritz_cgs_kernel + ritz_ov_kernel + ritz_kernel (fused R=Wh-theta Vh with its overlaps; restart X=Vh, Y=Wh)", "csr_stream_kernel (CSR SpMV)"]
HBM_PEAK_GBS = 8000.0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def spawn_ranks(n):
    """WORLD_SIZE unset and --gpus N > 1: this process becomes the launcher of N ranks of itself."""
    import torch
    have = torch.cuda.device_count()
    share = bool(os.environ.get("PRIMME_AMD_BENCH_SHARE_GPU"))
    if have < n and not share:
        raise SystemExit(f"bench.py: --gpus {n} needs {n} GPUs, this box has {have}; refusing to run {n} ranks on fewer devices "
                         "(a one-rank number must not be recorded as an N-GPU one)")
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r else None))
    rc = 0
    for r, p in enumerate(procs):
        p.communicate()
        if p.returncode != 0:
            rc = rc or p.returncode or 1
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise SystemExit(f"bench.py: a rank failed (exit code {rc})")
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="lap2d_10m", choices=sorted(WORKLOADS))
    ap.add_argument("--budget-s", type=float, default=180.0, help="cap on the seconds of timed solves of the main workload")
    ap.add_argument("--num-evals", type=int, default=10)
    ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--operator", default="csr", choices=["csr", "stencil"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true", help="skip the extra configs[1] solves")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)

    import numpy as np
    import torch
    from primme_amd import _ffi as F
    from primme_amd import Operator, problems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    share = bool(os.environ.get("PRIMME_AMD_BENCH_SHARE_GPU"))
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible (primme_amd has no CPU path)")
    if ndev < world and not share:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s) visible; refusing to put several ranks on one device")
    shared_devices = ndev < world
    # PRIMME_AMD_BENCH_DIST=1: run the multi-rank plumbing (process group, communicator, in-stream reductions)
    # even with one rank — exercises it on a one-GPU box
    dist_path = world > 1 or bool(os.environ.get("PRIMME_AMD_BENCH_DIST"))
    if dist_path and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
        os.environ["PRIMME_AMD_FORCE_COMM"] = "1"
    torch.cuda.set_device(local_rank % ndev)
    lib = F.load_product()

    comm = None
    transport = None
    if dist_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_devices:
            os.environ["PRIMME_AMD_COMM"] = "ipc"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        def make_id():
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_char * 128)()
                assert lib.primme_amd_comm_unique_id(buf) == 0
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if not shared_devices:
                uid = uid.cuda()
            dist.broadcast(uid, 0)
            return bytes(uid.cpu().numpy().tobytes())

        raw = make_id()
        if raw[:6] == b"PAIPC1":
            # the id of the peer-to-peer transport names a shared-memory segment of THIS node: if some rank cannot see it
            # (ranks in separate containers / IPC namespaces), every rank switches to RCCL before anybody waits for a rendez-vous
            seg = "/dev/shm" + raw[8:72].split(b"\0")[0].decode()
            seen = torch.tensor([1 if os.path.exists(seg) else 0], dtype=torch.int32, device="cpu" if shared_devices else "cuda")
            dist.all_reduce(seen, op=dist.ReduceOp.MIN)
            if int(seen.item()) == 0:
                if shared_devices:
                    raise SystemExit("bench.py: ranks sharing a device need the peer-to-peer transport, but its rendez-vous segment is not visible to every rank")
                if rank == 0:
                    print("bench.py: the rendez-vous segment of the peer-to-peer transport is not visible to every rank; using PRIMME_AMD_COMM=rccl", file=sys.stderr)
                    try:
                        os.unlink(seg)
                    except OSError:
                        pass
                os.environ["PRIMME_AMD_COMM"] = "rccl"
                raw = make_id()
        comm = C.c_void_p()
        assert lib.primme_amd_comm_create(C.byref(comm), raw, rank, world) == 0
        lib.primme_amd_comm_transport.restype = C.c_char_p
        lib.primme_amd_comm_transport.argtypes = [C.c_void_p]
        transport = lib.primme_amd_comm_transport(comm).decode()

    def barrier():
        torch.cuda.synchronize()
        if dist_path:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_path:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared_devices else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from primme_amd.api import Session

    # attainable bandwidth of THIS box, measured in this process: read-only with the panel kernels' access
    # pattern, and copy (read + write); SURVEY 8(d) asks for the fraction of it next to the fraction of 8 TB/s
    stream = {}
    if rank == 0:
        ctx = C.c_void_p()
        assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
        g = C.c_double()
        lib.hipk_read_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        lib.hipk_bandwidth_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        if lib.hipk_read_probe(ctx, 2 << 30, 10, C.byref(g)) == 0:
            stream["read_GBps"] = round(g.value, 1)
        if lib.hipk_bandwidth_probe(ctx, 1 << 30, 10, C.byref(g)) == 0:
            stream["copy_GBps"] = round(g.value, 1)
        lib.hipk_ctx_destroy(ctx)
    barrier()

    def run_workload(name, steps_req, warmup_req, budget_s):
        """Timed solves of one workload (operator, start vector and panels resident in HBM before the timed
        region), then ONE more solve with HIP events around every launch of the hot kernel classes for the
        roofline."""
        wl = WORKLOADS[name]
        dims = wl["dims"]
        n = int(np.prod(dims))
        # row partition: contiguous slabs of a multiple of 16 rows (the last rank takes the remainder): a slab with an odd
        # number of rows puts every second column of the caller's evecs array (leading dimension nLocal) off the 16-byte
        # boundary and the panel kernels then fall back to 8-byte loads (profiles/r04_two_ranks_one_device_kernel_stats.md)
        base = (n // world) // 16 * 16
        row0 = rank * base
        nloc = base if rank < world - 1 else n - base * (world - 1)
        if args.operator == "csr":
            rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
            op = Operator(n, csr=(rp, ci, va), row0=row0, nrows=nloc)
            nnz = int(rp[-1])
        else:
            op = Operator(n, stencil=tuple(list(dims) + [1] * (3 - len(dims))), row0=row0, nrows=nloc)
            nnz = 0
        v0 = problems.start_vector(n, row0=row0, nrows=nloc)
        kw = dict(numEvals=args.num_evals, method="GD_plusK", eps=args.eps, aNorm=wl["aNorm"], v0=v0,
                  return_evecs=False, numProcs=world, procID=rank)
        # the matrix stays resident through a persistent session
        sess = Session(op, comm=comm, dtype=np.float64)
        lib.hipk_csr_index_bytes.argtypes = [C.c_void_p]
        idx_bytes = lib.hipk_csr_index_bytes(dict(sess.handles)["csr"]) if args.operator == "csr" else 0
        last = None
        # first warm-up solve, timed on its own: it sizes the cap
        barrier()
        t0 = time.perf_counter()
        last = sess.solve(**kw)       # always at least one untimed solve (`warmup` on the JSON line = solves actually run)
        barrier()
        t_first = max_over_ranks(time.perf_counter() - t0)
        steps, warmup = steps_req, max(warmup_req, 1)
        cap_note = None
        if budget_s and t_first * steps_req > budget_s:
            steps = max(1, min(steps_req, int(budget_s // t_first)))
            warmup = 1
            cap_note = (f"one solve takes {t_first:.1f} s: timed solves capped to ~{budget_s:.0f} s "
                        f"({steps} of the {steps_req} requested steps, {warmup} of the {warmup_req} requested warm-up solves)")
        for _ in range(max(0, warmup - 1)):
            last = sess.solve(**kw)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = sess.solve(**kw)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        # roofline leg (kept out of the timed steps: two event records per launch cost a few %)
        lib.hipk_prof_reset()
        lib.hipk_prof_enable(1)
        barrier()
        lastp = sess.solve(**kw)
        barrier()
        lib.hipk_prof_enable(0)
        sess.close()

        ok = last.ret == 0 and last.initSize == args.num_evals and bool(
            np.all(last.resNorms <= args.eps * wl["aNorm"] * (1 + 1e-12)))
        exact = problems.laplacian_eigenvalues(dims, args.num_evals)
        eval_err = float(np.max(np.abs(np.sort(last.evals) - exact))) if last.ret == 0 else float("inf")
        # a solve that did not converge to the analytic spectrum is not a measurement: fail loudly (all ranks see the same result)
        if not ok or not (eval_err <= 1e-6 * wl["aNorm"]):
            raise SystemExit(f"bench.py: workload {name} on {world} rank(s) did not converge to the target (ret {last.ret}, "
                             f"{last.initSize} of {args.num_evals} pairs, largest eigenvalue error {eval_err:.3e}, "
                             f"largest residual norm {float(np.max(last.resNorms)) if len(last.resNorms) else float('nan'):.3e}); no number is reported")

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
        # the SpMV class on the bytes it really streams: the 16-bit index stream takes 2 of plain CSR's 4 index bytes per nonzero
        spmv_streamed = prof[3][2] - prof[3][1] * nnz * (4 - idx_bytes) if idx_bytes else prof[3][2]
        all_kernels = {KERNEL_CLASSES[c].split(" ")[0]: {
            "ms": round(prof[c][0], 2), "launches": prof[c][1],
            "GBps": round((prof[c][2] / max(prof[c][0], 1e-12)) / 1e6, 1)} for c in range(4)}
        all_kernels["csr_stream_kernel"]["accounting"] = "plain CSR: nnz*(8+4) + (m+1)*4 + 3*m*8 bytes per fused launch"
        all_kernels["csr_stream_kernel"]["GBps_streamed"] = round(spmv_streamed / max(prof[3][0], 1e-12) / 1e6, 1)
        all_kernels["csr_stream_kernel"]["streamed_accounting"] = f"{8 + idx_bytes} bytes per nonzero ({idx_bytes}-byte index stream) instead of 12"
        so_streamed = so_bytes - (prof[3][2] - spmv_streamed)
        roofline = {
            "kernel": KERNEL_CLASSES[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
            "measured_stream": stream or None,
            "frac_of_measured_stream": (round(achieved / stream["read_GBps"], 4) if stream.get("read_GBps") else None),
            "frac_of_measured_stream_note": "achieved / read_GBps: the dominant class reads (2k+L+1) columns per column it writes; copy_GBps counts read + write bytes",
            "launches": launches, "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
            "alg_bytes_per_launch": round(nbytes / max(launches, 1)),
            "all_kernels": all_kernels,
            "inner_loop_all_classes": {"GBps": round(tot_bytes / max(tot_ms, 1e-12) / 1e6, 1),
                                       "frac": round(tot_bytes / max(tot_ms, 1e-12) / 1e6 / HBM_PEAK_GBS, 4),
                                       "kernel_ms_per_solve": round(tot_ms, 1)},
            "spmv_plus_ortho": {"GBps": round(so_bytes / max(so_ms, 1e-12) / 1e6, 1),
                                "frac": round(so_bytes / max(so_ms, 1e-12) / 1e6 / HBM_PEAK_GBS, 4),
                                "frac_streamed_bytes": round(so_streamed / max(so_ms, 1e-12) / 1e6 / HBM_PEAK_GBS, 4),
                                "classes": "csr spmv + CGS update (project) + TN inner products; frac on plain-CSR bytes, "
                                           "frac_streamed_bytes with the SpMV on the bytes it streams"},
        }
        res = {
            "value": round(steps * args.num_evals / elapsed, 4), "ms_per_step": round(1e3 * elapsed / steps, 3), "steps": steps,
            "warmup": warmup,
            "config": {"workload": f"{name}: {wl['desc']}, n={n}, {args.num_evals} smallest, GD+k, "
                                   f"blockSize 1, eps={args.eps}*|A|, |A|={wl['aNorm']}, operator={args.operator}",
                       "partition": f"rows/{world}", "converged": ok, "max_eval_error_vs_analytic": eval_err,
                       "outer_iterations": last.stats["numOuterIterations"], "matvecs": last.stats["numMatvecs"],
                       "restarts": last.stats["numRestarts"],
                       "us_per_outer_iteration": round(1e6 * elapsed / steps / max(1, last.stats["numOuterIterations"]), 2),
                       "steps_requested": steps_req, "warmup_requested": warmup_req, "steps_cap": cap_note,
                       "profiled_solve_outer_iterations": lastp.stats["numOuterIterations"]},
            "roofline": roofline}
        if dist_path:
            res["config"]["transport"] = transport
            if shared_devices:
                res["config"]["devices"] = f"{world} ranks on {ndev} device(s): functional run, NOT a scaling number"
        return res, last, dims, wl, n

    main_res, last, dims, wl, n = run_workload(args.workload, args.steps, args.warmup, args.budget_s)
    out = {
        "metric": "eigenpairs/sec to target resNorm", "value": main_res["value"],
        "unit": "eigenpairs/s", "n_gpus": world, "steps": main_res["steps"], "warmup": main_res["warmup"],
        "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": main_res["config"], "roofline": main_res["roofline"],
    }
    if not args.no_configs1 and args.workload != "lap3d_2m":
        c1, _, _, _, _ = run_workload("lap3d_2m", 3, 1, 0.0)
        out["configs1"] = {"metric": "eigenpairs/sec to target resNorm", "value": c1["value"], "unit": "eigenpairs/s",
                           "ms_per_step": c1["ms_per_step"], "steps": c1["steps"], "n_gpus": world,
                           "config": c1["config"], "roofline": c1["roofline"]}

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
        barrier()
        lib.primme_amd_comm_destroy(comm)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_path:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
