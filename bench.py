#!/usr/bin/env python3
"""Headline benchmark: training iterations / second (forward + backward + Adam + MSE) of 2D Gaussian
splatting on a 4096x4096 synthetic image with 1,000,000 Gaussians (BASELINE.json metric / configs[3]).

  python bench.py --gpus N --steps K --warmup W

One process per GPU.  With N > 1 and no launcher environment (WORLD_SIZE unset) this process only SPAWNS the N
ranks -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`
as a child, before anything here touches the GPU -- relays rank 0's JSON line and returns the child's exit code;
started under torch.distributed.run it is one of the ranks.

The image is split into N row slabs (whole tile rows).  --exchange halo (default for N > 1): slab OWNERSHIP -- a
rank holds, lists and updates only the splats that can reach its rows and exchanges the gradient rows of splats
held by more than one rank (one all_to_all over RCCL/xGMI per iteration).  --exchange dense: splats and Adam state
replicated, the N x 9 fp32 gradient array all-reduced every iteration (north_star's scheme).  The total work is
fixed as N grows ("strong" scaling).  Prints ONE JSON line on rank 0, which names the scheme that ran.

PyTorch is plumbing here (device memory for the all-reduce buffer, streams, torch.distributed); all
compute is the hand-written HIP library behind include/splat2d.h.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def host_cores():
    """CPU cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.999)))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(W, H, n, threads):
    """The oracle (a CPU restatement of the reference loop: kind "port"), timed on this box's host cores on
    a bounded sample: whole iterations of the same workload (forward/backward split over `threads` row slabs) until
    about 10 s have passed, at least one and at most eight."""
    import numpy as np
    import oracle_lib as O
    tgt = O.synthetic_target(W, H)
    o = O.OracleTrainer(tgt, n)
    t0 = time.perf_counter()
    done = 0
    while done < 8 and (done == 0 or time.perf_counter() - t0 < 10.0):
        o.step(threads=threads)
        done += 1
    total = time.perf_counter() - t0
    dt = total / done
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    out = {"value": 1.0 / dt, "unit": "iterations/s", "cores": threads, "kind": "port", "cpu_model": model,
           "sample": "%d iteration(s) of the same %dx%d / %d-Gaussian workload (oracle/s2d_oracle.c, gcc -O2 "
                     "-ffp-contract=off, forward+backward over %d row-slab threads, Adam+MSE single thread); "
                     "%.2f s = %.0f core-seconds" % (done, W, H, n, threads, total, total * threads)}
    # the reference itself is single-threaded: time one thread on a bounded sample (the top quarter of the rows,
    # forward + backward) and scale by the row count
    rows = max(16, (H // 4 // 16) * 16)   # a quarter of the rows
    o.init()
    t0 = time.perf_counter()
    o.forward(0, rows)
    o.backward(0, rows)
    d1 = time.perf_counter() - t0
    out["single_thread"] = {"value": 1.0 / (d1 * H / rows), "unit": "iterations/s", "cores": 1,
                            "sample": "forward+backward of rows [0,%d) of the same workload on one thread, %.2f s, "
                                      "scaled by %d/%d rows" % (rows, d1, H, rows)}
    return out


def kernel_source_digest():
    """Ties a recorded profile (profiles/traffic.json) to the kernels it measured: comments and whitespace do not count."""
    return importlib.import_module("2dgaussiansplatting_amd._build").kernel_source_digest()


STAGES = ["spawn", "start", "init", "selftest", "warmup", "timed", "psnr", "side-block", "report", "done"]


_STAGE = {"name": "spawn"}


def stage(rank, name):
    """Stage marker of a rank, on stderr: the parent's watchdog keeps the latest one so that a run that stops says where."""
    _STAGE["name"] = name
    sys.stderr.write("[bench stage] rank %d: %s\n" % (rank, name))   # ONE write: the ranks share the pipe
    sys.stderr.flush()


def start_rank_watchdog(args, rank, world):
    """A rank started by somebody else's launcher (the driver runs `python -m torch.distributed.run ... bench.py --gpus N`
    itself: no parent of ours watches it) must not end as a silent kill either.  A timer thread with the same budget: when it
    fires, rank 0 prints the ONE JSON line -- an error record with the stage it had reached -- and every rank leaves with
    exit code 4, which makes the launcher stop the job.  (A rank stuck inside a collective still runs this thread: the
    collectives release the interpreter lock.)  Under our own parent the timer is set a little later than the parent's, which
    reports more (every rank's stage, the stderr tail).  Returns the timer: cancel it once the result is out."""
    import threading
    budget = watchdog_budget(args) + (15.0 if os.environ.get("S2D_BENCH_PARENT") == "1" else 0.0)
    if rank != 0:
        budget += 5.0   # rank 0 writes the record before another rank's exit makes the launcher stop everybody

    def fire():
        why = ("rank %d of %d produced no result within %.0f s; it was at stage '%s' (a collective or a peer that never answered, or "
               "a device that stopped)" % (rank, world, budget, _STAGE["name"]))
        sys.stderr.write("bench.py watchdog: %s\n" % why)
        sys.stderr.flush()
        if rank == 0:
            sys.stdout.write(json.dumps({"metric": "train iters/sec (fwd+bwd+Adam) + PSNR vs ref; 4K img, 1M splats", "value": None,
                                         "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                         "error": why, "stage": _STAGE["name"], "watchdog_budget_s": budget}) + "\n")
            sys.stdout.flush()
        os._exit(4)

    t = threading.Timer(budget, fire)
    t.daemon = True
    t.start()
    return t


def watchdog_budget(args):
    """Wall-clock budget of one attempt of the child job, from the amount of work asked for: import + rendezvous + RCCL
    set-up (N first imports of torch on a fresh box take minutes) + the iterations (warm-up, timed block, side block, the run
    up to iteration 200 for the PSNR) at a generous 10 ms each.  239 s for the default --steps 100."""
    if args.watchdog_seconds > 0:
        return float(args.watchdog_seconds)
    return 235.0 + 0.01 * (args.warmup + 2 * args.steps + 200)


def run_ranks_once(n, argv, budget_s):
    """Start the N ranks as a CHILD torch.distributed.run in a process group of its own, relay its output, and stop it
    when the budget runs out.  This (parent) process never touches the GPU.  Returns a dict: rc (None = killed by the
    watchdog), json_seen, stage reached (the furthest any rank reported), the last stderr lines, seconds."""
    import collections
    import re
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["S2D_BENCH_PARENT"] = "1"   # the ranks' own watchdogs then fire after this parent's
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    state = {"json": False, "stage": "spawn", "per_rank": {}}
    tail = collections.deque(maxlen=25)

    def pump_out():
        for line in proc.stdout:   # rank 0's JSON line goes to stdout; anything else the ranks print there, to stderr
            is_json = line.lstrip().startswith("{")
            state["json"] = state["json"] or is_json
            out = sys.stdout if is_json else sys.stderr
            out.write(line)
            out.flush()

    def pump_err():
        for line in proc.stderr:
            marks = re.findall(r"\[bench stage\] rank (\d+): ([a-z-]+)", line)
            for r, name in marks:
                state["per_rank"][int(r)] = name
                if name in STAGES and STAGES.index(name) > STAGES.index(state["stage"]):
                    state["stage"] = name
            if not marks:
                tail.append(line.rstrip()[-300:])
            sys.stderr.write(line)
            sys.stderr.flush()

    pumps = [threading.Thread(target=pump_out, daemon=True), threading.Thread(target=pump_err, daemon=True)]
    for th in pumps:
        th.start()
    rc = None
    try:
        rc = proc.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        # the whole group: torch.distributed.run and every rank under it.  Never restarted in place: a second attempt
        # (run_ranks) is a FRESH child whose processes have not initialised the GPU yet
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
    for th in pumps:
        th.join(timeout=5.0)
    return {"rc": rc, "json_seen": state["json"], "stage": state["stage"], "stage_per_rank": dict(sorted(state["per_rank"].items())),
            "stderr_tail": list(tail), "seconds": time.perf_counter() - t0}


def spawn_ranks(n, argv, args):
    """N > 1 without a launcher.  Whatever happens the caller gets ONE JSON line on stdout: rank 0's result, or -- when
    the ranks die, or stop answering (a first-contact RCCL / peer-to-peer stall would otherwise end as a kill at the
    driver's limit with nothing written) -- an {"error": ...} record that names the stage reached and carries the last
    stderr lines; exit code non-zero then.  If the exchange scheme was left to the default and the run stopped AFTER the
    process group had come up, one fresh child is tried with --exchange dense (north_star's all-reduce) before giving up."""
    budget = watchdog_budget(args)
    attempts = []
    t_all = time.perf_counter()
    res = run_ranks_once(n, argv, budget)
    attempts.append(res)
    if res["rc"] == 0 and res["json_seen"]:
        return 0
    stalled = res["rc"] is None
    got_going = STAGES.index(res["stage"]) >= STAGES.index("selftest")
    left = 570.0 - (time.perf_counter() - t_all)
    if stalled and got_going and args.exchange is None and not args.launch_selftest and left >= 60.0:
        print("bench.py: the ranks stopped answering at stage '%s' with the default exchange (halo); one fresh attempt with "
              "--exchange dense" % res["stage"], file=sys.stderr)
        res = run_ranks_once(n, argv + ["--exchange", "dense"], min(budget, left))
        attempts.append(res)
        if res["rc"] == 0 and res["json_seen"]:
            return 0
    if not res["json_seen"]:
        why = ("the ranks stopped answering: no result %.0f s after the launch (watchdog budget %.0f s); the process group was killed"
               % (res["seconds"], budget)) if res["rc"] is None else "the ranks exited with code %s before rank 0 printed a result" % res["rc"]
        print(json.dumps({"metric": "train iters/sec (fwd+bwd+Adam) + PSNR vs ref; 4K img, 1M splats", "value": None,
                          "unit": "iterations/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                          "error": why, "stage": res["stage"], "stage_per_rank": res["stage_per_rank"],
                          "attempts": [{"exchange": "default" if k == 0 else "dense", "rc": a["rc"], "stage": a["stage"],
                                        "seconds": round(a["seconds"], 1)} for k, a in enumerate(attempts)],
                          "stderr_tail": res["stderr_tail"], "watchdog_budget_s": budget}), flush=True)
    return res["rc"] if res["rc"] not in (None, 0) else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--rebin-interval", type=int, default=0)
    ap.add_argument("--fp16-images", action="store_true",
                    help="BASELINE configs[4] 'fp16 colour / fp32 grads': framebuffer and target held as 4 x fp16 per pixel")
    ap.add_argument("--lean-block", dest="lean_block", action="store_true", default=None,
                    help="after the timed block (whose backward pass accumulates all nine gradients, like main.cpp:595-710), time the "
                         "same steps again with dL/d(opacity) skipped (what s2d_step does while optimizeOpacity is off, "
                         "main.cpp:317,735) and report iterations_per_s_lean beside value.  Default: on for one GPU")
    ap.add_argument("--no-lean-block", dest="lean_block", action="store_false")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--exchange", default=None, choices=["halo", "dense"],
                    help="N > 1: 'halo' (default) = slab ownership, ranks exchange gradient rows of boundary splats only "
                         "(distributed.HaloStep); 'dense' = replicated state, all-reduce of all N x 9 gradients.  Left to the "
                         "default, a failing all_to_all self-test switches to 'dense' (and the JSON says so); given by name, "
                         "it ends the run with exit code 3 instead")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                    "to rehearse the multi-process path with several ranks sharing one GPU)")
    ap.add_argument("--watchdog-seconds", type=float, default=0.0,
                    help="N > 1 started from the plain command: wall-clock budget of the child job before the parent stops it and "
                         "prints an error record (0 = derived from --steps: 235 s + 10 ms per iteration)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="only check the multi-process launch path: the ranks rendezvous, all-reduce their rank numbers "
                         "and rank 0 prints one JSON line; needs no GPU (tests/test_distributed_cpu.py)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args))

    if args.launch_selftest:
        import torch
        import torch.distributed as dist_mod
        import datetime
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        stage(rank, "start")
        dog = start_rank_watchdog(args, rank, world) if world > 1 else None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist_mod.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
        stage(rank, "init")
        if os.environ.get("S2D_BENCH_SELFTEST_HANG_RANK") == str(rank):
            while True:            # a rank that stops answering after the rendezvous (tests/test_distributed_cpu.py)
                time.sleep(3600)
        v = torch.tensor([rank + 1], dtype=torch.int64)
        if world > 1:
            stage(rank, "selftest")
            dist_mod.all_reduce(v)
            dist_mod.barrier()
        if os.environ.get("S2D_BENCH_SELFTEST_FAIL_RANK") == str(rank):
            sys.exit(7)  # lets the test see that a failing rank fails the whole command
        if dog is not None:
            dog.cancel()
        if rank == 0:
            print(json.dumps({"selftest": "launch", "world": world, "sum": int(v.item()), "gpus_arg": args.gpus}))
        if world > 1:
            dist_mod.destroy_process_group()
        return

    import numpy as np
    import torch
    S2D = importlib.import_module("2dgaussiansplatting_amd")
    D = importlib.import_module("2dgaussiansplatting_amd.distributed")

    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stage(rank, "start")
    dog = start_rank_watchdog(args, rank, world) if world > 1 else None
    if world != args.gpus:
        args.gpus = world  # the launcher's world size wins
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the trainer has no CPU fallback")
    device = local_rank % torch.cuda.device_count()  # == local_rank on a full node; ranks share a GPU only in gloo rehearsals
    if args.backend == "nccl" and world > torch.cuda.device_count():
        sys.exit("RCCL needs one GPU per rank: %d ranks, %d GPUs" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    dist = None
    # S2D_BENCH_FORCE_DIST=1 (under a launcher with one rank): take the N > 1 code -- process group on RCCL, the
    # collectives on the context's stream -- with a world of one; how a one-GPU box rehearses that path
    use_dist = world > 1 or ("WORLD_SIZE" in os.environ and os.environ.get("S2D_BENCH_FORCE_DIST") == "1")
    if use_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # 120 s, not torch's 10 minutes (longer than the driver's own limit): a rendezvous or a collective that a peer
        # never joins ends this rank with an error the parent's record can quote
        tmo = datetime.timedelta(seconds=120)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device), timeout=tmo)
        else:
            dist.init_process_group(backend=args.backend, timeout=tmo)
    stage(rank, "init")
    # which physical device this rank runs on: the record must PROVE that N distinct GPUs took part
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "device": device, "pid": os.getpid(),
          "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
          "uuid": str(getattr(props, "uuid", "")), "name": props.name, "arch": getattr(props, "gcnArchName", "")}
    ranks_info = [me]
    if dist is not None:
        ranks_info = [None] * dist.get_world_size()
        dist.all_gather_object(ranks_info, me)

    W, H, n = args.width, args.height, args.splats
    r0, r1 = D.slab_rows(H, rank, world)
    # a real (non-default) stream: its handle goes into the C ABI, so the library's kernels, the RCCL
    # all-reduce and the timing events below are all ordered on the same HIP stream
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    grads = torch.zeros(n * 9, dtype=torch.float32, device="cuda")
    t = S2D.Trainer(W, H, n, device=device, row_begin=r0, row_end=r1,
                    rebin_interval=args.rebin_interval, fp16_images=args.fp16_images, stream=stream.cuda_stream)
    t.bind_grads(grads.data_ptr())
    # `value` is the reference's whole iteration: all NINE gradients are accumulated, dL/d(opacity) included (main.cpp:703-704
    # accumulates it every iteration whether or not "Optimize opacity" is on)
    t.lean_backward = False
    t.set_target_synthetic()
    t.init()

    exchange_requested = args.exchange or "halo"
    exchange = exchange_requested if use_dist else "none"
    stage(rank, "selftest")
    if exchange == "halo" and not D.all_to_all_selftest(dist, "cuda"):
        # the collective pattern slab ownership needs misbehaved on this stack (every rank agreed on that: all-reduce)
        if args.exchange == "halo":
            # asked for by name: a scaling curve must not change scheme between two values of N without anybody noticing
            if rank == 0:
                print("bench.py: --exchange halo was requested and its all_to_all self-test failed on this stack", file=sys.stderr)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
        # default scheme: use the replicated-state one instead -- slower, same results; the JSON line names both
        if rank == 0:
            print("bench.py: all_to_all self-test failed, falling back to --exchange dense", file=sys.stderr)
        exchange = "dense"
    if exchange == "halo":
        # forward, backward, exchange of the gradient rows of splats held by more than one rank, Adam on held splats
        step = D.HaloStep(t, D.HipHaloOps(t, n, "cuda"), dist, rank, world, H)
    else:
        # forward + backward, all-reduce(grads), Adam; every 64 iterations the replicas' parameter checksums are compared
        ops = D.HipHaloOps(t, n, "cuda") if use_dist else None
        all_ids = torch.arange(n, dtype=torch.int32, device="cuda") if use_dist else None
        step = D.SlabStep(t, grads, dist, params=(lambda: ops.rows_gather(D.ROWS_SPLATS, all_ids)) if use_dist else None,
                          check_interval=64)

    def one_step(ev=None):
        if ev is None:
            step()
        else:  # HIP events around the raster kernel (fused forward + backward), on the stream it is launched on
            step(before_raster=lambda: ev[0].record(stream), after_raster=lambda: ev[1].record(stream))

    def timed_block(steps):
        """EXACTLY `steps` iterations between barrier + synchronize on both sides; wall clock = max over ranks.  Also, from
        HIP events on the stream: every step's duration and the raster kernel's launch duration (over the steps that
        rebuilt no tile list: there the events bracket exactly that kernel + the ~6 us squared-error reduction; on a
        rebuild step they would also span the void optimistic launch, the rebuild and the relaunch)."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # step boundaries on the stream
        reb = []
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            marks[k].record(stream)
            before = t.rebuild_count()
            one_step(ev[k])
            reb.append(t.rebuild_count() != before)  # host-side counter, no synchronisation
        marks[steps].record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        t.synchronize()  # raises if a parameter went non-finite
        reb = np.array(reb, dtype=bool)
        step_ms = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(steps)], dtype=np.float64)
        k_ms = np.array([a.elapsed_time(b) for a, b in ev], dtype=np.float64)
        k_use = k_ms[~reb] if (~reb).any() else k_ms
        red = torch.tensor([dt, float(k_use.mean()) if steps else 0.0], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
        steady = step_ms[~reb] if (~reb).any() else step_ms
        return {"seconds": float(red[0].item()), "kernel_ms": float(red[1].item()), "step_ms": step_ms, "rebuilds": reb,
                "steady_ms": float(steady.mean()) if steps else float("nan")}

    stage(rank, "warmup")
    for _ in range(args.warmup):
        one_step()
    stage(rank, "timed")
    blk = timed_block(args.steps)
    dt, step_ms, rebuilds = blk["seconds"], blk["step_ms"], blk["rebuilds"]

    done = t.stats()["iterations"]
    first = done - args.steps
    # PSNR after a fixed 200 iterations: keep training, untimed, up to iteration 200, and
    # read iteration 199's squared error from the device-side trace ring (it holds the last 65536 iterations)
    stage(rank, "psnr")
    extra = max(0, 200 - done)
    for _ in range(extra):
        one_step()
    torch.cuda.synchronize()
    psnr_iter = 199 if done + extra - 200 < 60000 else done + extra - 1
    sq200 = torch.from_numpy(t.sqerr_trace(psnr_iter, 1)).cuda()
    # Side block, outside `value`: the same steps with dL/d(opacity) left out (S2D_BWD_SKIP_OPACITY_GRAD: what s2d_step
    # does by itself while "Optimize opacity" is off, because Adam reads dSplats.opacity only with it on, main.cpp:735) --
    # same trajectory, the NEED_OP = false instantiation of the raster kernel.
    lean = None
    late = None
    want_lean = args.lean_block if args.lean_block is not None else world == 1
    if want_lean and args.steps > 0:
        # ... and, first, the SAME nine-gradient steps timed again here, after iteration 200: an iteration gets cheaper as
        # training goes on (front splats sharpen, pixels saturate earlier), so `value` -- timed over the first iterations after
        # init(), the dearest of a run -- reads lower than any later window of the same run.  (Round 3 timed its nine-gradient
        # block at this point of the run.)
        stage(rank, "side-block")
        for _ in range(min(args.warmup, 5)):
            one_step()
        fb = timed_block(args.steps)
        late = {"iterations_per_s": args.steps / fb["seconds"], "ms_per_step": 1e3 * fb["seconds"] / args.steps,
                "iterations_per_s_median": 1e3 / float(np.median(fb["step_ms"])), "iterations_per_s_steady_state": 1e3 / fb["steady_ms"],
                "steps_with_list_rebuild": int(fb["rebuilds"].sum()), "kernel_ms": fb["kernel_ms"], "steps": args.steps,
                "first_iteration": int(t.stats()["iterations"]) - args.steps,
                "backward": "all nine gradients, like `value`; timed after iteration 200 instead of right after init()"}
        t.lean_backward = True
        for _ in range(min(args.warmup, 5)):
            one_step()
        lb = timed_block(args.steps)
        t.lean_backward = False
        lean = {"iterations_per_s": args.steps / lb["seconds"], "ms_per_step": 1e3 * lb["seconds"] / args.steps,
                "iterations_per_s_median": 1e3 / float(np.median(lb["step_ms"])), "iterations_per_s_steady_state": 1e3 / lb["steady_ms"],
                "steps_with_list_rebuild": int(lb["rebuilds"].sum()), "kernel_ms": lb["kernel_ms"], "steps": args.steps,
                "backward": "8 of the 9 gradients: dL/dopacity skipped (nothing reads it while optimizeOpacity is off, main.cpp:317,735)"}
    stage(rank, "report")
    if dist is not None and sq200 is not None:
        D.reduce_sqerr(sq200, dist)
    sq = torch.from_numpy(t.sqerr_trace(first, args.steps)).cuda()
    bwd_ms = blk["kernel_ms"]
    if dist is not None:
        D.reduce_sqerr(sq, dist)
    mse_last = float(sq[-1].item()) / (H * W * 3) if args.steps else float("nan")
    stats = t.stats()

    if rank == 0:
        its = args.steps / dt
        # dominant kernel: raster_fused_kernel = forward + backward walk of every tile.  Algorithmic bytes per launch
        # (SURVEY.md section 8d's per-unit figures): per pixel of the slab the forward part writes
        # the framebuffer (16 B), the backward part reads it (16 B) and the target (16 B); per splat the parameters are
        # read once per pass (2 x 36 B) and the 9 gradient floats written once (36 B).  (The events also span the
        # ~6 us squared-error reduction queued behind it.)
        bwd_bytes = (24.0 if args.fp16_images else 48.0) * W * (r1 - r0) + 108.0 * n  # fp16 images: 8 B per pixel access
        bwd_s = bwd_ms * 1e-3
        achieved = bwd_bytes / bwd_s / 1e9 if bwd_s > 0 else 0.0
        # HBM traffic of that kernel from the PMC counters: a RECORDED figure (rocprofv3 cannot run inside this
        # process), valid only for the kernel build and the launch it was measured on -- dropped otherwise
        traffic, traffic_note, pmc = None, "no PMC pass recorded for this build / launch", None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and world == 1 and (W, H, n) == (4096, 4096, 1000000) and not args.fp16_images:
            try:
                tj = json.load(open(tp))
                if tj.get("kernel_source_digest") == kernel_source_digest():
                    traffic = tj.get("dominant_kernel_bytes_per_launch")
                    traffic_note = "recorded: %s" % tj.get("source")
                    pmc = tj
                else:
                    traffic_note = "profiles/traffic.json was measured on another kernel build (digest %s): dropped" % tj.get("kernel_source_digest")
            except Exception:
                pass
        # "PSNR vs ref": the CPU oracle's trajectory on this very workload (tools/make_bench_reference_trace.py)
        psnr_ref = None
        rp = os.path.join(ROOT, "tests", "golden", "bench_reference_trace.json")
        if os.path.exists(rp):
            rj = json.load(open(rp))
            if (rj["width"], rj["height"], rj["n_splats"]) == (W, H, n) and 0 <= psnr_iter < len(rj["mse"]) and not args.fp16_images:
                psnr_ref = 10.0 * float(np.log10(255.0 ** 2 / rj["mse"][psnr_iter]))
        psnr_gpu = (10.0 * float(np.log10(255.0 ** 2 / (float(sq200[0].item()) / (H * W * 3)))) if sq200 is not None and float(sq200[0].item()) > 0 else None)
        steady = step_ms[~rebuilds] if (~rebuilds).any() else step_ms
        out = {
            "metric": "train iters/sec (fwd+bwd+Adam) + PSNR vs ref; 4K img, 1M splats",
            "value": its,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / max(args.steps, 1),
            # per-step durations from HIP events on the stream (rank 0): median, and the steps that rebuilt no tile list
            "ms_per_step_median": float(np.median(step_ms)) if args.steps else None,
            "iterations_per_s_median": (1e3 / float(np.median(step_ms))) if args.steps else None,
            "iterations_per_s_steady_state": (1e3 / float(steady.mean())) if args.steps else None,
            "steps_with_list_rebuild": int(rebuilds.sum()),
            "backward_in_value": "all nine gradients of main.cpp:595-710: dL/dpos, dL/dsx, dL/dsy, dL/drot, dL/dcolour, dL/dopacity",
            # side block, NOT in `value`: the same steps with dL/d(opacity) skipped (nothing reads it while "Optimize opacity" is
            # off, main.cpp:317,735; what s2d_step does by itself), timed the same way
            "iterations_per_s_lean": lean["iterations_per_s"] if lean else None,
            "lean_backward": lean,
            # the nine-gradient iteration again, later in the same run (NOT `value`): iterations get cheaper as training goes on
            "iterations_per_s_after_200_iterations": late["iterations_per_s"] if late else None,
            "nine_gradients_after_200_iterations": late,
            "value_window": "iterations %d..%d after init() (the dearest of a run)" % (first, first + args.steps - 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic target ref(x,y)=(x/W,1-x/W,y/H); splats from the reference's init() seeds",
            # which physical devices took part, gathered from the ranks themselves, and the size of the RCCL / gloo group
            "ranks": ranks_info,
            "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
            "collective_backend": (dist.get_backend() if dist is not None else "none"),
            "distinct_devices": len({(r or {}).get("uuid") or (r or {}).get("pci_bus_id") for r in ranks_info}),
            "config": {"workload": "%dx%d synthetic RGB, %d Gaussians, %s%s" % (
                           W, H, n, "fp16 colour / fp32 gradients" if args.fp16_images else "fp32",
                           " (BASELINE.json configs[3])" if (W, H, n, args.fp16_images) == (4096, 4096, 1000000, False) else
                           " (BASELINE.json configs[4])" if (W, H, n, args.fp16_images) == (8192, 8192, 4000000, True) else ""),
                       "width": W, "height": H, "n_splats": n,
                       "parallelism": "rowslab%d%s" % (world, ("+%s-%s" % ("rccl" if args.backend == "nccl" else args.backend,
                                                                              "halo-exchange" if exchange == "halo" else "allreduce-grads")) if use_dist else ""),
                       "rebin_interval": args.rebin_interval,
                       "exchange_requested": (exchange_requested + ("" if args.exchange else " (default)")) if use_dist else "none",
                       "exchange": exchange},
            "mse_last": mse_last,
            "psnr_db_last": (10.0 * float(np.log10(255.0 ** 2 / mse_last)) if mse_last > 0 else None),
            "psnr_db_at_iteration": psnr_iter,
            "psnr_db": psnr_gpu,
            "psnr_ref_db": psnr_ref,  # the CPU oracle (reference loop) at the same iteration of the same workload
            "psnr_delta_db": (psnr_gpu - psnr_ref) if (psnr_gpu is not None and psnr_ref is not None) else None,
            "iterations_total": stats["iterations"],
            "exchange_rank0": ({"scheme": "halo", "held_fraction": float(((step.mask >> rank) & 1).float().mean().item()),
                                "rows_exchanged_per_iteration": int(sum(step.splits)), "state_handovers": int(step.handed_over)}
                               if isinstance(step, D.HaloStep) else
                               {"scheme": "dense" if use_dist else "none", "replica_checksum_checks": getattr(step, "checks", 0)}),
            "pairs_binned_rank0": stats["pairs_binned"],
            "rebins_rank0": stats["rebins"],
            # SURVEY.md section 8d prices this path in HBM bytes, so `roofline` is the HBM one; what actually bounds the kernel is
            # named beside it (`bound_measured`): it issues vector instructions, the framebuffer traffic is ~6 % of the pipe
            "roofline": {"bound": "hbm", "kernel": "raster_fused_kernel (all nine gradients: its NEED_OP instantiation)",
                         "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel_ms": bwd_ms, "algorithmic_bytes_per_launch": bwd_bytes,
                         "kernel_ms_note": "mean over the launches that did work; rocprofv3's AverageNs for this kernel also counts "
                                           "one void ~20 us launch per list rebuild (profiles/r04/README.md)",
                         "bound_measured": "valu-issue",
                         "bound_measured_note": "vector-instruction issue, with the LDS pipe and per-wave latency within ~1.3x of it "
                                                "(perturbation experiments, profiles/r03/r03_bound_experiments.txt; priced instruction "
                                                "account ~93 % vector issue, profiles/r03/r03_tile_clock.txt); HBM is the idle resource"},
        }
        if pmc is not None and pmc.get("SQ_INSTS_VALU_per_launch"):
            # counter-derived, from the same digest-tied PMC passes as `traffic` (per working launch of the dominant kernel)
            valu = float(pmc["SQ_INSTS_VALU_per_launch"])
            out["roofline"]["valu"] = {"wave_instructions_per_launch": valu, "gpu_cycles_per_launch": pmc.get("gpu_cycles_per_launch"),
                                       "simd_cycles_per_valu_instruction": pmc.get("simd_cycles_per_valu_instruction"),
                                       "lds_wave_instructions_per_launch": pmc.get("SQ_INSTS_LDS_per_launch"),
                                       "salu_wave_instructions_per_launch": pmc.get("SQ_INSTS_SALU_per_launch"),
                                       "source": pmc.get("sq_source")}
        if world == 1:
            # The path has no dense contraction (no MFMA) and the raster kernels are nowhere near bandwidth-bound, so beside the
            # HBM roofline report the useful fp32 VALU rate: active (pixel, splat) pairs per pass, counted by the
            # kernels in one extra untimed iteration, x 30 flop (forward, main.cpp:523-533) + 110 flop (backward,
            # main.cpp:607-709) per pair, against the 157.3 TFLOP/s fp32 vector peak.
            sp = t.get_splats()
            with S2D.Trainer(W, H, n, device=device, count_pairs=True, fp16_images=args.fp16_images) as tc:
                tc.set_target_synthetic()
                tc.set_splats(sp)
                tc.forward()
                tc.backward()   # a counting context runs the two separate kernels (same walks as the fused one)
                tc.synchronize()
                cs = tc.stats()
            flops = 30.0 * cs["fwd_active"] + 110.0 * cs["bwd_active"]
            if "valu" in out["roofline"]:
                # useful flops / issued lane-operations (64 lanes per wave instruction; no FMA contraction: one flop per lane-op)
                out["roofline"]["valu"]["useful_lane_fraction"] = flops / (out["roofline"]["valu"]["wave_instructions_per_launch"] * 64.0)
            out["compute"] = {"unit": "TFLOP/s", "achieved": flops * its / 1e12, "peak": 157.3,
                              "frac": flops * its / 1e12 / 157.3, "active_pairs_per_pass": cs["bwd_active"],
                              "visited_pairs_per_pass": cs["bwd_visited"], "staged_list_entries_per_pass": cs["bwd_staged"],
                              "lanes_per_executed_wave_entry": cs["bwd_active"] / max(cs["bwd_wave_execs"], 1),
                              "executed_wave_entries_per_pass": cs["bwd_wave_execs"],
                              "note": "useful fp32 VALU flops only (no FMA contraction is allowed: one flop per instruction); the raster kernel sits "
                                      "between vector issue, the LDS pipe and per-wave latency (DESIGN.md section 4, "
                                      "profiles/r03/r03_bound_experiments.txt)"}
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or host_cores()
            out["cpu_baseline"] = cpu_baseline(W, H, n, threads)
        else:
            out["cpu_baseline"] = None
        if dog is not None:
            dog.cancel()
        print(json.dumps(out))
    elif dog is not None:
        dog.cancel()
    t.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
