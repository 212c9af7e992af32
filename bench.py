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
             CSR int32.  One solve takes ~25 s on one MI355X (30 846 outer iterations).  --steps K / --warmup W are
             honoured whenever the whole run fits --budget-s seconds of wall clock (default 1500: the driver gives
             a bench run 1800 s; `--steps 20 --warmup 5` needs ~900 s); otherwise warm-up drops to 1 solve and the
             timed solves to what fits.  `steps` / `warmup` on the JSON line are the numbers actually run, the
             requested ones are in config.  The timed solves run WITHOUT the per-launch events of the roofline leg.
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
  configs2      (N == 1) BASELINE configs[2]: Matrix-Market input — tests/golden/reference_driver/LUNDA.mtx read by the
                library's C reader (primme_amd_mm_read) and tiled block-diagonally 34 014 times by its C tiler
                (n = 5 000 058, 83.3 M nonzeros), JDQMR, block size 8, 20 eigenvalues closest to 4.4764e8, Jacobi.
  configs3      (N == 1) BASELINE configs[3] on one GPU: complex Hermitian band n = 4 M, 6 largest, block 4, GD+k.
  configs4      (N == 1) BASELINE configs[4] on one GPU: 8 M x 2 M CSR, 10 largest singular triplets, normal equations.
                Each with value / ms_per_step over its own timed solves and a roofline object (dominant kernel class
                by device time of one more, profiled solve).  --no-extra-configs skips the three.
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
                  "ritz_kernel class = ritz_cgs_kernel + ritz_ov_kernel + ritz_kernel (fused R=Wh-theta Vh with its overlaps; restart X=Vh, Y=Wh)", "csr_stream_kernel (CSR SpMV)",
                  "vector passes (axpy / copy / norms / QMR recurrences / Jacobi)"]
# the same five classes of the live profiler (csrc/hipk_internal.h: HIPK_PROF_*), named for any method / scalar type
GENERIC_CLASSES = ["TN panel products [Q V]'X (dots_kernel / dots_mfma_kernel / zdots_kernel)",
                   "NN panel updates X -= [Q V]c (project_kernel / project_mul_kernel / zproject_kernel)",
                   "fused Ritz / residual / restart update (ritz_*_kernel / zritz_kernel)",
                   "sparse operator (csr_*_block_kernel / zcsr_kernel / pb_matvec_kernel / pat_kernel)",
                   "vector passes (QMR recurrences, axpy / xpay / copy / gather, norms, Jacobi)"]
NCLS = 5
SPMV_FORMATS = {0: "CSR row tiles", 1: "panel-blocked", 2: "row patterns", 3: "stencil"}
T_START = time.perf_counter()
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
        # ranks > 0 print nothing that is needed: their stdout goes nowhere (a pipe nobody drains would block a chatty rank,
        # e.g. verbose RCCL logging, and with it the whole job); stderr stays the terminal's
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.DEVNULL if r else None))
    # all ranks are watched together: the first one that fails takes the others down at once (not after rank 0's
    # rendez-vous timeout)
    rc = 0
    live = list(procs)
    while live and not rc:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = code or 1
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                pass
        raise SystemExit(f"bench.py: a rank failed (exit code {rc}); the other ranks were stopped")
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="lap2d_10m", choices=sorted(WORKLOADS))
    ap.add_argument("--budget-s", type=float, default=1500.0,
                    help="wall-clock budget of the whole run: --steps / --warmup are honoured when the run fits, reduced otherwise")
    ap.add_argument("--num-evals", type=int, default=10)
    ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--operator", default="csr", choices=["csr", "stencil"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true", help="skip the extra configs[1] solves")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip configs[2] (Matrix-Market), configs[3] (complex), configs[4] (svds)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--comm", default=None, choices=["auto", "ipc", "rccl"],
                    help="N > 1: transport of the library's communicator (default: PRIMME_AMD_COMM, else auto = mailboxes + RCCL)")
    args = ap.parse_args()

    if args.comm:
        os.environ["PRIMME_AMD_COMM"] = args.comm
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    # The contract is ONE JSON line on stdout.  Libraries underneath write banners to file descriptor 1 (RCCL's version block,
    # gloo's "connected to N peer ranks"): from here on fd 1 is stderr, and the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
        lib.primme_amd_comm_transport.restype = C.c_char_p
        lib.primme_amd_comm_transport.argtypes = [C.c_void_p]
        lib.primme_amd_comm_selftest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]

        def make_id():
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_char * 128)()
                # one node (the launch contract of this benchmark): the library hands out a mailbox id when the mailboxes
                # can serve `world` ranks and an ncclUniqueId otherwise
                assert lib.primme_amd_comm_unique_id_for(buf, world, 0) == 0
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if not shared_devices:
                uid = uid.cuda()
            dist.broadcast(uid, 0)
            return bytes(uid.cpu().numpy().tobytes())

        def all_ranks(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if shared_devices else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        def make_comm(mode, required):
            """One communicator of the library on transport `mode` (auto | ipc | rccl): create, self-test (every collective against
            known data + the latency of a small all-reduce) — collectively; returns (handle or None, record).  A transport that
            does not come up or does not deliver is RECORDED (and ends the run only when `required`)."""
            os.environ["PRIMME_AMD_COMM"] = mode
            rec = {"requested": mode}
            raw = make_id()
            if raw[:6] == b"PAIPC1":
                # the id of the peer-to-peer transport names a shared-memory segment of THIS node: if some rank cannot see it
                # (ranks in separate containers / IPC namespaces), nobody waits for a rendez-vous that cannot happen
                seg = "/dev/shm" + raw[8:72].split(b"\0")[0].decode()
                if not all_ranks(os.path.exists(seg)):
                    if rank == 0:
                        try:
                            os.unlink(seg)
                        except OSError:
                            pass
                    rec.update(came_up=False, why="the rendez-vous segment of the peer-to-peer transport is not visible to every rank")
                    if shared_devices or (required and mode == "ipc"):
                        raise SystemExit("bench.py: " + rec["why"])
                    if mode != "rccl":
                        return make_comm("rccl", required)[0], dict(rec, fell_back_to="rccl")
                    return None, rec
            c = C.c_void_p()
            rc = lib.primme_amd_comm_create(C.byref(c), raw, rank, world)
            if not all_ranks(rc == 0):
                if rc == 0:
                    lib.primme_amd_comm_destroy(c)
                rec.update(came_up=False, why=f"primme_amd_comm_create returned {rc} on rank {rank}" if rc else "another rank could not create it")
                if required:
                    raise SystemExit(f"bench.py: rank {rank}: the communicator did not come up on transport {mode}: {rec['why']}")
                return None, rec
            us = C.c_double(-1.0)
            st_rc = lib.primme_amd_comm_selftest(c, None, 1000, C.byref(us))
            rec.update(came_up=True, transport=lib.primme_amd_comm_transport(c).decode(), selftest="passed" if st_rc == 0 else f"FAILED (code {st_rc})",
                       allreduce_us=round(us.value, 2),
                       checked="all-reduce of 1..4096 doubles, neighbour halo, all-gather / reduce-scatter of column blocks, integer exchange",
                       allreduce_us_is="wall clock per 8-double all-reduce over 1000 back-to-back reductions on this rank")
            if not all_ranks(st_rc == 0):
                lib.primme_amd_comm_destroy(c)
                if required:
                    raise SystemExit(f"bench.py: rank {rank}: the communicator self-test failed on transport {rec['transport']} (code {st_rc}); no number is reported")
                return None, rec
            return c, rec

        # Both transports, each with its self-test and all-reduce latency on the line (VERDICT r05 Next #7): the one the run is timed
        # on — --comm / PRIMME_AMD_COMM, default auto = mailboxes for the small reductions and halos + RCCL for bulk — and, when the
        # ranks sit on distinct devices and nothing was forced, RCCL alone as the second one (a few solves after the headline).
        primary_mode = args.comm or os.environ.get("PRIMME_AMD_COMM") or "auto"
        if shared_devices:
            primary_mode = "ipc"
        comm, comm_selftest = make_comm(primary_mode, True)
        transport = comm_selftest["transport"]
        comm_records = [comm_selftest]
        second = None
        if world > 1 and not shared_devices and not args.comm and transport != "rccl" and not os.environ.get("PRIMME_AMD_BENCH_ONE_TRANSPORT"):
            try:
                second, rec2 = make_comm("rccl", False)
            except SystemExit:
                raise
            except Exception as e:      # (deterministic on every rank: a Python-level error, not a transport failure)
                second, rec2 = None, {"requested": "rccl", "came_up": False, "why": repr(e)}
            comm_records.append(rec2)
        os.environ["PRIMME_AMD_COMM"] = primary_mode
        selftest_of = {comm.value: comm_selftest}
        if second is not None:
            selftest_of[second.value] = comm_records[-1]
        if rank == 0:
            for rec in comm_records:
                print("bench.py: transport " + json.dumps(rec), file=sys.stderr, flush=True)

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

    lib.hipk_prof_get.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]
    lib.hipk_prof_streamed.argtypes = [C.c_int]; lib.hipk_prof_streamed.restype = C.c_double

    def read_prof():
        out = []
        for cls in range(NCLS):
            ms, launches, nbytes = C.c_double(), C.c_long(), C.c_double()
            lib.hipk_prof_get(cls, C.byref(ms), C.byref(launches), C.byref(nbytes))
            # [3]: the bytes the launches really moved in the form they ran in (== [2] except for a sparse operator in a
            # compressed form: 2-byte index stream, row patterns, panel-blocked entries) — every `frac` on the line is taken on these
            out.append((ms.value, launches.value, nbytes.value, float(lib.hipk_prof_streamed(cls))))
        return out

    def mfma_evidence(key):
        """MFMA utilisation of the matrix-core panel kernels of a config: from a SEPARATE rocprofv3 --pmc pass (counters cannot be
        read from inside this process), summarised in profiles/pmc_mfma.json by scripts/pmc_mfma.py; labelled as such"""
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_mfma.json"))).get(key)
            if pj:
                return dict(pj, source="NOT measured in this run: " + pj.get("source", "profiles/pmc_mfma.json"))
        except Exception:
            pass
        return None

    def generic_roofline(prof, note=None):
        """roofline object of a profiled solve of any method: dominant kernel class by device time (HIP events on the
        solver's stream around every launch of the class), its algorithmic bytes per launch / its average launch duration"""
        dom = max(range(NCLS), key=lambda c: prof[c][0])
        ms, launches, nbytes, nstream = prof[dom]
        ach = (nstream / max(ms, 1e-12)) / 1e6 if launches else 0.0
        def cls_obj(c):
            o = {"ms": round(prof[c][0], 2), "launches": prof[c][1], "GBps": round((prof[c][3] / max(prof[c][0], 1e-12)) / 1e6, 1)}
            o["frac"] = round(o["GBps"] / HBM_PEAK_GBS, 4)
            if abs(prof[c][2] - prof[c][3]) > 1e-9 * max(prof[c][2], 1.0):
                o["plain_csr_accounting_GBps"] = round((prof[c][2] / max(prof[c][0], 1e-12)) / 1e6, 1)
            return o
        r = {"kernel": GENERIC_CLASSES[dom], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": launches,
             "avg_launch_us": round(1e3 * ms / max(launches, 1), 2), "alg_bytes_per_launch": round(nbytes / max(launches, 1)),
             "streamed_bytes_per_launch": round(nstream / max(launches, 1)),
             "bytes_note": "achieved / frac / GBps: the bytes the launches move in the form they run in (streamed: 2-byte index stream, row patterns, "
                           "panel-blocked entries for the sparse operator; equal to the algorithmic count for every panel class); "
                           "plain_csr_accounting_GBps: the same time against nnz*(s+4) + (m+1)*4 + 2*m*s per column — an accounting, not a bandwidth",
             "all_kernels": {GENERIC_CLASSES[c].split(" (")[0]: cls_obj(c) for c in range(NCLS) if prof[c][1]},
             "kernel_ms_per_solve": round(sum(p_[0] for p_ in prof), 1)}
        if note:
            r["note"] = note
        return r

    def run_workload(name, steps_req, warmup_req, budget_s, comm=None, profile=True):
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
        transport = lib.primme_amd_comm_transport(comm).decode() if comm is not None else None
        lib.hipk_csr_index_bytes.argtypes = [C.c_void_p]
        lib.hipk_csr_format.argtypes = [C.c_void_p]
        lib.hipk_csr_product_bytes.argtypes = [C.c_void_p, C.c_int]; lib.hipk_csr_product_bytes.restype = C.c_double
        hcsr = dict(sess.handles)["csr"]
        idx_bytes = lib.hipk_csr_index_bytes(hcsr) if args.operator == "csr" else 0
        spmv_format = lib.hipk_csr_format(hcsr)
        # HBM bytes one fused product (scale + A t + t'At, two outputs) moves in the form that serves it
        spmv_bytes_fused = lib.hipk_csr_product_bytes(hcsr, 1)
        last = None
        # first warm-up solve, timed on its own: it sizes the cap
        barrier()
        t0 = time.perf_counter()
        last = sess.solve(**kw)       # always at least one untimed solve (`warmup` on the JSON line = solves actually run)
        barrier()
        t_first = max_over_ranks(time.perf_counter() - t0)
        steps, warmup = steps_req, max(warmup_req, 1)
        cap_note = None
        if budget_s:
            # what is left of the run's wall-clock budget after this point: the profiled solve, the other configs,
            # the CPU baseline sample and the set-up / tear-down around them
            reserve = t_first + (0.0 if world > 1 else 160.0) + 40.0
            avail = budget_s - (time.perf_counter() - T_START) - reserve
            if (warmup - 1 + steps) * t_first > avail:
                warmup = 1
                steps = max(1, min(steps_req, int(avail // t_first)))
                cap_note = (f"one solve takes {t_first:.1f} s and the run has a wall-clock budget of {budget_s:.0f} s: "
                            f"{steps} of the {steps_req} requested steps, {warmup} of the {warmup_req} requested warm-up solves")
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
        prof = read_prof()
        dom = max(range(NCLS), key=lambda c: prof[c][0])
        ms, launches, nbytes, nstream = prof[dom]
        achieved = (nstream / max(ms, 1e-12)) / 1e6 if launches else 0.0   # bytes/ms -> GB/s, on the bytes the class moves
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
        tot_ms = sum(p_[0] for p_ in prof[:4])
        tot_alg, tot_str = sum(p_[2] for p_ in prof[:4]), sum(p_[3] for p_ in prof[:4])
        so_ms = prof[3][0] + prof[1][0] + prof[0][0]
        so_alg, so_str = prof[3][2] + prof[1][2] + prof[0][2], prof[3][3] + prof[1][3] + prof[0][3]
        # EVERY `GBps` / `frac` below is on the bytes the launches really move (the sparse operator in the form that serves the
        # product: CSR row tiles stream 8 + 2 or 8 + 4 bytes per nonzero, the row-pattern form one byte per row); the
        # plain-CSR accounting of the same time is kept under *_plain_csr_accounting_GBps — an accounting, not a bandwidth
        def cls_obj(c):
            o = {"ms": round(prof[c][0], 2), "launches": prof[c][1], "GBps": round((prof[c][3] / max(prof[c][0], 1e-12)) / 1e6, 1)}
            o["frac"] = round(o["GBps"] / HBM_PEAK_GBS, 4)
            return o
        all_kernels = {KERNEL_CLASSES[c].split(" ")[0]: cls_obj(c) for c in range(NCLS)}
        sp = all_kernels["csr_stream_kernel"]
        sp["format"] = SPMV_FORMATS.get(spmv_format, str(spmv_format))
        sp["streamed_bytes_per_launch"] = round(prof[3][3] / max(prof[3][1], 1))
        sp["streamed_accounting"] = (
            "row-pattern form: m*(1 + 3*8) bytes per fused launch (one pattern byte per row, x once, y and the normalised vector written)"
            if spmv_format == 2 else f"{8 + idx_bytes} bytes per nonzero ({idx_bytes}-byte index stream) instead of 12")
        sp["plain_csr_accounting_GBps"] = round((prof[3][2] / max(prof[3][0], 1e-12)) / 1e6, 1)
        sp["plain_csr_accounting_note"] = ("nnz*(8+4) + (m+1)*4 + 3*m*8 bytes per fused launch — the ALGORITHMIC bytes of a CSR product; a compressed "
                                           "form moves fewer, so this figure may exceed the HBM peak and is no roofline fraction")
        roofline = {
            "kernel": KERNEL_CLASSES[dom], "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
            "measured_stream": stream or None,
            "frac_of_measured_stream": (round(achieved / stream["read_GBps"], 4) if stream.get("read_GBps") else None),
            "frac_of_measured_stream_note": "achieved / read_GBps: the dominant class reads (2k+L+1) columns per column it writes; copy_GBps counts read + write bytes",
            "launches": launches, "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
            "alg_bytes_per_launch": round(nbytes / max(launches, 1)),
            "all_kernels": all_kernels,
            "inner_loop_all_classes": {"GBps": round(tot_str / max(tot_ms, 1e-12) / 1e6, 1),
                                       "frac": round(tot_str / max(tot_ms, 1e-12) / 1e6 / HBM_PEAK_GBS, 4),
                                       "plain_csr_accounting_GBps": round(tot_alg / max(tot_ms, 1e-12) / 1e6, 1),
                                       "note": "GBps / frac on the bytes the launches move (the SpMV in the form it runs in)",
                                       "kernel_ms_per_solve": round(tot_ms, 1)},
            "spmv_plus_ortho": {"GBps": round(so_str / max(so_ms, 1e-12) / 1e6, 1),
                                "frac": round(so_str / max(so_ms, 1e-12) / 1e6 / HBM_PEAK_GBS, 4),
                                "plain_csr_accounting_GBps": round(so_alg / max(so_ms, 1e-12) / 1e6, 1),
                                "classes": "csr spmv + CGS update (project) + TN inner products; GBps / frac with the SpMV on the bytes it "
                                           "streams — north_star's target for this group is frac >= 0.60"},
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
        try:   # iterations enqueued before the host had seen the previous one / adopted, of the last solve
            pre = (C.c_long * 2)()
            lib.primme_amd_prelaunch_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
            lib.primme_amd_prelaunch_stats(C.cast(pre, C.POINTER(C.c_long)), C.cast(C.byref(pre, C.sizeof(C.c_long)), C.POINTER(C.c_long)))
            res["config"]["iterations_enqueued_ahead"] = {"launched": int(pre[0]), "adopted": int(pre[1])}
        except AttributeError:
            pass
        if dist_path:
            res["config"]["transport"] = transport
            res["config"]["comm_selftest"] = selftest_of.get(comm.value) if comm is not None else None
            res["config"]["transports"] = comm_records
            if shared_devices:
                res["config"]["devices"] = f"{world} ranks on {ndev} device(s): functional run, NOT a scaling number"
        return res, last, dims, wl, n

    main_res, last, dims, wl, n = run_workload(args.workload, args.steps, args.warmup, args.budget_s, comm=comm)
    out = {
        "metric": "eigenpairs/sec to target resNorm", "value": main_res["value"],
        "unit": "eigenpairs/s", "n_gpus": world, "steps": main_res["steps"], "warmup": main_res["warmup"],
        "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": main_res["config"], "roofline": main_res["roofline"],
    }
    if not args.no_configs1 and args.workload != "lap3d_2m":
        c1, _, _, _, _ = run_workload("lap3d_2m", 3, 1, 0.0, comm=comm)
        out["configs1"] = {"metric": "eigenpairs/sec to target resNorm", "value": c1["value"], "unit": "eigenpairs/s",
                           "ms_per_step": c1["ms_per_step"], "steps": c1["steps"], "n_gpus": world,
                           "config": c1["config"], "roofline": c1["roofline"]}

    # ---- N > 1: the same workloads once more on the OTHER transport (RCCL alone), so that one scaling run records the pair
    if dist_path and second is not None:
        try:
            r2, _, _, _, _ = run_workload(args.workload, min(args.steps, 3), 1, 0.0, comm=second)
            out["other_transport"] = {"transport": r2["config"].get("transport"), "value": r2["value"], "unit": "eigenpairs/s", "ms_per_step": r2["ms_per_step"],
                                      "steps": r2["steps"], "us_per_outer_iteration": r2["config"]["us_per_outer_iteration"],
                                      "outer_iterations": r2["config"]["outer_iterations"], "comm_selftest": r2["config"].get("comm_selftest")}
            if "configs1" in out:
                c1b, _, _, _, _ = run_workload("lap3d_2m", 3, 1, 0.0, comm=second)
                out["other_transport"]["configs1"] = {"value": c1b["value"], "ms_per_step": c1b["ms_per_step"],
                                                      "us_per_outer_iteration": c1b["config"]["us_per_outer_iteration"]}
        except SystemExit:
            raise
        except Exception as e:      # the second transport never takes the headline line down; the failure is on the line
            out["other_transport"] = {"value": None, "error": repr(e)}

    # ---- the other BASELINE configs on the driver-timed line (one GPU): Matrix-Market input, complex Hermitian, singular values
    def timed_solves(solve, nsolves):
        """1 warm-up solve, `nsolves` timed ones (barrier + device synchronisation on both sides), 1 profiled solve"""
        r = solve()
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsolves):
            r = solve()
        barrier()
        el = time.perf_counter() - t0
        lib.hipk_prof_reset(); lib.hipk_prof_enable(1)
        barrier()
        rp_ = solve()
        barrier()
        lib.hipk_prof_enable(0)
        return r, el, read_prof(), rp_

    def config2_lunda():
        """BASELINE configs[2]: LUNDA.mtx through the library's C Matrix-Market reader and block-diagonal tiler"""
        from primme_amd import ingest as ingest_c     # ctypes plumbing over primme_amd_mm_read / primme_amd_csr_tile_block_diagonal
        mtx = os.path.join(ROOT, "tests", "golden", "reference_driver", "LUNDA.mtx")
        rp0, ci0, va0, n0, _ = ingest_c.mm_read(lib, mtx)
        T, shift = 34014, 4.4764e8           # shift: DESIGN.md section 5 (SURVEY's 1.0e6 converges nowhere, reference included)
        rp, ci, va = ingest_c.tile_block_diagonal(lib, rp0, ci0, va0, T, 1.0, 1.0 / T)
        n = n0 * T
        A0 = np.zeros((n0, n0)); A0[np.repeat(np.arange(n0), np.diff(rp0)), ci0] = va0
        w = (np.linalg.eigvalsh(A0)[None, :] * (1.0 + np.arange(T) / T)[:, None]).ravel()
        aN = float(np.abs(w).max())
        want = np.sort(w[np.argsort(np.abs(w - shift))][:20])
        sess = Session(Operator(n, csr=(rp, ci, va)), dtype=np.float64)
        kw = dict(numEvals=20, target="closest_abs", targetShifts=[shift], method="JDQMR", maxBlockSize=8, eps=1e-8, aNorm=aN,
                  precond=("jacobi", shift), return_evecs=False)
        try:
            r, el, prof, _ = timed_solves(lambda: sess.solve(**kw), 3)
        finally:
            sess.close()
        err = float(np.max(np.abs(np.sort(r.evals) - want))) if r.ret == 0 else float("inf")
        ok = bool(r.ret == 0 and r.initSize == 20 and err <= 1e-10 * aN and np.all(r.resNorms <= 1e-8 * aN * (1 + 1e-6)))
        return {"metric": "eigenpairs/sec to target resNorm", "value": round(3 * 20 / el, 4), "unit": "eigenpairs/s", "ms_per_step": round(1e3 * el / 3, 3),
                "steps": 3, "n_gpus": 1, "dtype": "f64", "data": "Matrix-Market file (reference tests/LUNDA.mtx, committed fixture) tiled",
                "config": {"workload": f"configs[2]: LUNDA.mtx (147 x 147, 2449 nnz) read by primme_amd_mm_read, tiled block-diagonally {T}x by "
                                       f"primme_amd_csr_tile_block_diagonal (tile t scaled by 1 + t/T): n={n}, nnz={len(va)}; 20 eigenvalues closest to {shift:.4e}, "
                                       f"JDQMR, blockSize 8, Jacobi K = diag(A) - shift, eps=1e-8*|A|, |A|={aN:.4e}",
                           "converged": ok, "max_eval_error_vs_dense_truth": err, "outer_iterations": r.stats["numOuterIterations"],
                           "matvecs": r.stats["numMatvecs"], "restarts": r.stats["numRestarts"],
                           "shift_note": "SURVEY 8(d) suggests the shift 1.0e6; at that shift the live reference does not converge either "
                                         "(profiles/r04_config3_reference_shift_1e6.log), BASELINE.json names none: 4.4764e8 is this build's choice"},
                "roofline": dict(generic_roofline(prof), mfma=mfma_evidence("configs2"))}

    def config3_hermitian():
        """BASELINE configs[3] on ONE GPU (its 8-GPU row partition: tests/test_multigpu_rccl.py)"""
        n = 4_000_000
        rp, ci, va = problems.hermitian_banded_csr(n)
        sess = Session(Operator(n, csr=(rp, ci, va)), dtype=np.complex128)
        kw = dict(numEvals=6, target="largest", eps=1e-8, maxBlockSize=4, maxBasisSize=20, minRestartSize=8, method="GD_plusK",
                  iseed=(2, 3, 5, 7), return_evecs=False)
        try:
            r, el, prof, _ = timed_solves(lambda: sess.solve(**kw), 3)
        finally:
            sess.close()
        aN = r.params["aNorm"]
        bound = 3.0 + 2 * (1 / 2 + 1 / 3 + 1 / 4)        # Gershgorin: the largest eigenvalues sit just below it
        ok = bool(r.ret == 0 and r.initSize == 6 and np.all(r.resNorms <= 1e-8 * aN * (1 + 1e-6)) and np.all(r.evals <= bound) and r.evals[0] >= bound - 1.5)
        return {"metric": "eigenpairs/sec to target resNorm", "value": round(3 * 6 / el, 4), "unit": "eigenpairs/s", "ms_per_step": round(1e3 * el / 3, 3),
                "steps": 3, "n_gpus": 1, "dtype": "c128 (f64 complex)", "data": "synthetic",
                "config": {"workload": f"configs[3] on one GPU: complex Hermitian band, half-bandwidth 3, n={n}, nnz={len(va)}; 6 largest, GD+k, blockSize 4, "
                                       f"basis 20 / restart 8, eps=1e-8*|A|, |A|={aN:.4f} (estimated by the solver)",
                           "converged": ok, "outer_iterations": r.stats["numOuterIterations"], "matvecs": r.stats["numMatvecs"],
                           "restarts": r.stats["numRestarts"], "largest_eigenvalue": float(r.evals[0]), "gershgorin_bound": bound},
                "roofline": dict(generic_roofline(prof), mfma=mfma_evidence("configs3"))}

    def config4_svds():
        """BASELINE configs[4] on ONE GPU"""
        from primme_amd.svds_api import SvdsSession
        m_, n_, k = 8_000_000, 2_000_000, 10
        rp, ci, va = problems.svds_synthetic_csr(m_, n_)
        sess = SvdsSession(m_, n_, (rp, ci, va))
        try:
            r, el, prof, _ = timed_solves(lambda: sess.solve(numSvals=k, eps=1e-8, methodStage1="GD_plusK"), 3)
        finally:
            sess.close()
        tol = 1e-8 * r.params["aNorm"]
        # power iteration on the host: a lower bound of the largest singular value (same check as tests/test_full_size_configs_gpu.py)
        x = np.ones(n_)
        rows = np.repeat(np.arange(m_, dtype=np.int64), np.diff(rp))
        for _ in range(3):
            u = np.bincount(rows, weights=va * x[ci], minlength=m_)
            x = np.bincount(ci, weights=va * u[rows], minlength=n_)
            x /= np.linalg.norm(x)
        s1 = float(np.linalg.norm(np.bincount(rows, weights=va * x[ci], minlength=m_)))
        ok = bool(r.ret == 0 and r.initSize == k and np.all(r.resNorms <= 2 * tol) and np.all(np.diff(r.svals) <= 1e-12 * r.svals[0]) and r.svals[0] >= s1 * (1 - 1e-10))
        return {"metric": "singular triplets/sec to target resNorm", "value": round(3 * k / el, 4), "unit": "triplets/s", "ms_per_step": round(1e3 * el / 3, 3),
                "steps": 3, "n_gpus": 1, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"configs[4] on one GPU: A {m_} x {n_} CSR, 5 nonzeros per row at (i*p_q + q) mod n; {k} largest singular triplets, "
                                       f"normal equations (GD+k on A'A), eps=1e-8*|A|, |A|={r.params['aNorm']:.4f}",
                           "converged": ok, "outer_iterations": r.stats["numOuterIterations"], "matvecs": r.stats["numMatvecs"],
                           "largest_singular_value": float(r.svals[0]), "power_iteration_lower_bound": s1},
                "roofline": generic_roofline(prof, note="sparse-operator class: the panel-blocked form streams 12 bytes per entry + the per-(tile, panel) counts + y once (x out of the L2)")}

    if rank == 0 and world == 1 and not args.no_extra_configs:
        for key, fn in (("configs2", config2_lunda), ("configs3", config3_hermitian), ("configs4", config4_svds)):
            t0 = time.perf_counter()
            try:
                out[key] = fn()
                out[key]["wall_s_incl_setup"] = round(time.perf_counter() - t0, 1)
                if not out[key]["config"]["converged"]:
                    raise SystemExit(f"bench.py: {key} did not converge to its target: {json.dumps(out[key]['config'])}")
            except SystemExit:
                raise
            except Exception as e:      # an extra object never takes the headline line down; the failure is on the line
                out[key] = {"value": None, "error": repr(e)}

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
        if dist_path and second is not None:
            lib.primme_amd_comm_destroy(second)
        lib.primme_amd_comm_destroy(comm)
    if rank == 0:
        line = (json.dumps(out, default=lambda o: o.item() if hasattr(o, "item") else str(o)) + "\n").encode()
        while line:
            line = line[os.write(real_stdout, line):]
    if dist_path:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
