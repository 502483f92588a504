#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X N-body Coulomb-oscillator engine.

Metric (BASELINE.json): particle-steps/sec (+ Gpair-interactions/sec), N = 1M FMM-3D p = 6, at 1/2/4/8 GPUs.

A "step" is one leapfrog step (integrator.cuh:68-96) of the whole particle set with the kd-tree FMM
evaluator + elastic term (coulombOscillatorFMMKD3, main3.cu:59-63), i.e. K(dt/2) D(dt) F K(dt/2),
with the tree rebuilt on every evaluation (the reference CPU driver's behaviour) unless --tree-steps
says otherwise.  Inputs are the reference's synthetic Gaussian ball (main3.cu:113-137 over the stream of
:662-664, produced by the library's host-only nbco_init_gaussian) resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the parent starts N fresh rank processes itself (before it
touches torch or the GPU) and relays rank 0's JSON line; under torchrun (WORLD_SIZE set) every process is one
rank.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (vector)"
FLOP_PER_PAIR = 20                   # one directed pair interaction = 20 flop


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", dest="n", type=int, default=1048576, help="particles per GPU (not --n: torchrun would read that as an abbreviation of its own flags)")
    ap.add_argument("--order", type=int, default=6)
    ap.add_argument("--tree-steps", type=int, default=1)
    ap.add_argument("--rebalance", type=int, default=16,
                    help="multi-GPU: force evaluations between two re-partitions of the kd-domains (top log2(G) splits)")
    ap.add_argument("--gather-partition", action="store_true",
                    help="multi-GPU: re-partition by all-gathering the state (nbco_dist_partition) instead of the distributed selection")
    ap.add_argument("--sharded-one", action="store_true",
                    help="diagnostics: run the sharded (multi-GPU) pipeline in a world of ONE rank over RCCL -- every stage and collective of "
                         "DomainRun on one card -- to price the pipeline itself against the plain single-GPU path")
    ap.add_argument("--no-let", action="store_true",
                    help="multi-GPU: all-gather whole node and position blocks instead of the locally-essential-tree exchange")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (one rank per GPU); gloo = rehearsal with several ranks sharing one card")
    ap.add_argument("--workload", default="fmm_kd", choices=["fmm_kd", "fmm_oct", "direct"],
                    help="fmm_kd: kd-tree FMM (the nbco3 path, default); fmm_oct: uniform octree with traceless multipoles; direct: O(N^2)")
    ap.add_argument("--far-fp64", action="store_true", help="fmm_oct only: multipole / local expansions and all far-field operators in double")
    ap.add_argument("--dens-inhom", type=float, default=1.0, help="the reference's -i option (deeper trees for clustered inputs)")
    ap.add_argument("--dt", type=float, default=5e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-calls", action="store_true", help="one nbco_integrate call per step instead of one nbco_integrate_steps call for the timed region")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the tree-reuse, strong-scaling, octree and nbco3 CLI legs after the timed region")
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (warm-up + K steps between two barriers) is run this many times from the same initial state; "
                         "value / ms_per_step are the median, every repeat is listed")
    ap.add_argument("--dump-state", metavar="PREFIX", default=None,
                    help="tests: every rank writes its [pos | vel | acc] rows after the last timed step to PREFIX.rank<r>.npy")
    ap.add_argument("--strict", action="store_true",
                    help="exit with status 3 (after printing the line) when the run fell back from the distributed re-partition or the LET "
                         "exchange to their all-gather forms (`fallbacks` in the output says so either way)")
    ap.add_argument("--late-from", type=int, default=1000,
                    help="extra leg `late_phase`: the same K steps timed again after this many steps of the simulation (0 = skip)")
    ap.add_argument("--profile-all", action="store_true", help="record HIP events around every phase (perturbs value)")
    ap.add_argument("--init-state", metavar="FILE.npy", default=None,
                    help="diagnostics: start from a state written by --dump-state (one GPU) instead of the Gaussian ball, e.g. a late phase")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="KEY=VALUE",
                    help="diagnostics: an nbco_opts field for the engine, e.g. p2p_mutual=0 (one-directional near-field kernel) or m2l_first=1")
    return ap.parse_args(argv)


# ---- N > 1 without a launcher: start the ranks ourselves --------------------------------------------------------------
def spawn_ranks(args):
    """One fresh python process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment), started before this
    process has imported torch or made any GPU call.  Rank 0's stdout is captured and its JSON line relayed; a failing rank
    fails the run (the others are terminated by PID, nothing is restarted)."""
    world = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one host thread per rank: the ranks' host sides poll and stage small tensors; an OpenMP pool per rank (one thread per
        # core of the host, spinning between parallel regions) eats the box's CPU share and stalls every collective
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    failed = None
    pending = set(range(world))
    while pending and failed is None:      # poll: a crashed rank must not leave the others hanging in a collective
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        if pending and failed is None:
            time.sleep(0.2)
    if failed is not None:
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        sys.stderr.write("bench.py: rank %d exited with status %d\n" % failed)
        return 1
    out0.seek(0)
    line = None
    for l in out0.read().splitlines():
        if l.startswith("{"):
            line = l
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    print(line)
    return 0


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this
    same command, tools/rocprof_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 wide reads)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        k = json.load(f)["kernels"]
    for name, v in k.items():
        if name.startswith(kernel):
            return v["hbm_bytes_fetch_doubled"], os.path.relpath(files[-1], ROOT)
    return None, None


def measured_ceiling():
    """fraction of the nominal fp32 peak that the bare pair bodies reach on register operands (tools/pair_ceiling.hip, summary
    committed under profiles/): {"pair": the 13-instruction one-directional body, "mutual": the 16-instruction Newton-III step},
    each as (sustained, burst) -- burst = launches of 0.2 ms, the length of the near-field kernel inside a step"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pair_ceiling.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        res = json.load(f)["results"]
    out = {}
    # (the packed body is what p2p_kernel runs since round 3; summaries from before hold the scalar body only)
    packed = any(r["variant"] in ("pair_pk", "pair4_pk") for r in res)
    for key, variants in (("pair", ("pair_pk", "pair4_pk") if packed else ("pair", "pair4")), ("mutual", ("mutual_dpp",))):
        for mode in ("sustained", "burst"):
            v = [r["frac_of_157.3"] for r in res if r["variant"] in variants and r.get("mode", "sustained") == mode]
            out["%s_%s" % (key, mode)] = max(v) if v else None
    return out, os.path.relpath(files[-1], ROOT)


# ---- synthetic workload: the initial condition of main3.cu:629-692 (Gaussian ball, centred, RMS-normalised) --------------
SIGMA_X = (0.003, 0.001, 0.01)      # main3.cu:244
OMEGA0 = (1.095, 1.0, 1.0)          # main3.cu:241
XI = 2e-6                           # main3.cu:240
REF_SEED = 5351550349027530206      # main3.cu:662
REF_DISCARD = 1248                  # main3.cu:664


def gaussian_ball(n, rank=0):
    """[pos | vel | acc] of n particles from the library's host-only nbco_init_gaussian: the reference's initGA over
    mt19937_64(REF_SEED + rank) after discard(1248).  Rank 0 therefore starts from exactly the state `nbco3 -n <n>` starts
    from; further ranks draw further samples of the same ball from their own streams."""
    import ctypes as C
    from coulomb_oscillators_amd import lib_path
    lib = C.CDLL(lib_path())
    lib.nbco_init_gaussian.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int]
    lib.nbco_init_gaussian.restype = C.c_int
    sx = np.array(SIGMA_X, dtype=np.float32)
    su = np.array(OMEGA0, dtype=np.float32) * sx
    buf = np.zeros((3, n, 3), dtype=np.float32)
    rc = lib.nbco_init_gaussian(buf.ctypes.data, n, sx.ctypes.data, su.ctypes.data, REF_SEED + rank, REF_DISCARD, 0)
    if rc != 0:
        raise RuntimeError("nbco_init_gaussian failed with status %d" % rc)
    return buf


def coulomb_params(n_system):
    """par[] of main3.cu:685-692: {xi / N, 0, 0, omega0^2}"""
    om = np.array(OMEGA0, dtype=np.float32)
    return np.array([np.float32(XI) / np.float32(n_system), 0, 0, om[0] * om[0], om[1] * om[1], om[2] * om[2]], dtype=np.float32)


def host_cpu():
    """(physical cores this process may run on, CPU model string) from /proc/cpuinfo and the affinity mask"""
    model, cores, cur = "unknown", set(), {}
    try:
        allowed = os.sched_getaffinity(0)
    except Exception:
        allowed = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if ":" in line:
                    k, v = [t.strip() for t in line.split(":", 1)]
                    cur[k] = v
                elif cur:
                    if "model name" in cur:
                        model = cur["model name"]
                    cpu = int(cur.get("processor", -1))
                    if allowed is None or cpu in allowed:
                        cores.add((cur.get("physical id", "0"), cur.get("core id", str(cpu))))
                    cur = {}
    except OSError:
        pass
    n = len(cores) or (len(allowed) if allowed else (os.cpu_count() or 1))
    return n, model


def cpu_quota():
    """CPUs' worth of time the cgroup of this process may use (None: no limit found)"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2
            q, p = f.read().split()[:2]
            return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # v1
            q, p = float(f.read()), float(g.read())
            return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(args, n):
    """The oracle (a port of the reference's multithreaded CPU path) on a bounded sample of the same workload: one
    std::thread per physical host core this process can actually keep busy (the box's CPU quota caps it: more threads than
    that only get throttled)."""
    from oracle import pyoracle as po
    o = po.Oracle(np.float32)
    phys, model = host_cpu()
    quota = cpu_quota()
    cores = phys if quota is None else max(1, min(phys, int(quota + 0.999)))
    model = "%s (%d physical cores visible, cgroup CPU quota %s)" % (model, phys, "none" if quota is None else "%.4g" % quota)
    buf = gaussian_ball(n)                  # the same state the GPU run starts from (rank 0)
    par = coulomb_params(n)
    if args.workload == "fmm_kd":
        kind, kw = po.KIND_FMM_KD, dict(p=args.order, unsort=False, threads=cores)
        steps = args.cpu_steps
        sample = "%d leapfrog steps of N=%d kd-tree FMM p=%d (+1 warm-up), %d std::threads" % (steps, n, args.order, cores)
        o.compute_force(kind, buf, par, **kw)
        t0 = time.perf_counter()
        for _ in range(steps):
            o.integrate(po.SCHEME_LEAPFROG, kind, buf, par, args.dt, **kw)
        dt = time.perf_counter() - t0
        return {"value": n * steps / dt, "unit": "particle-steps/s", "cores": cores, "cpu": model, "kind": "port", "sample": sample}
    # direct: rows are independent, time a slice of the targets against all sources
    ns = min(n, 32768)
    sub = buf[:, :ns].copy()
    t0 = time.perf_counter()
    o.direct3(sub[0], par, threads=cores)
    dt = time.perf_counter() - t0
    rate = ns * ns / dt                     # pair interactions / s
    return {"value": rate / n, "unit": "particle-steps/s", "cores": cores, "cpu": model, "kind": "port",
            "sample": "direct3 on N=%d (pair rate %.3g/s scaled to N=%d), %d std::threads" % (ns, rate, n, cores)}


def cli_leg(n, order, iters):
    """Throughput of the drop-in binary itself: `nbco3 -n N -p P -iters K` prints the wall time of its integration loop behind the
    first snapshot and, separately, from iteration 9 on (`Steady loop time`: behind the cold builds of a fresh process) -- the
    figure to hold against `tree_reuse`, which times the library in its stride with the same options (tree_steps 8, m2l_first).
    (The cost of a step grows as the ball evolves -- 0.5 ms per step over the first 200 steps, 2.8 ms around step 2000, where the
    cloud has focused -- so runs of different length must not be differenced.)"""
    import re
    import tempfile
    exe = os.path.join(ROOT, "coulomb_oscillators_amd", "host", "nbco3")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([exe, "-n", str(n), "-p", str(order), "-iters", str(iters), "-steps", "100000", "-o", tmp],
                           capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return {"error": r.stderr[-300:]}
    m = re.search(r"Loop time: ([0-9.eE+-]+) s, (\d+) iterations", r.stdout)
    if not m:
        return {"error": "no loop time in the output"}
    per_iter = float(m.group(1)) / int(m.group(2))
    out = {"command": "nbco3 -n %d -p %d -iters %d -steps 100000 (tree_steps 8, m2l_first: the reference GPU driver's defaults)" % (n, order, iters),
           "loop_s": float(m.group(1)), "ms_per_step": 1e3 * per_iter, "cli_particle_steps_per_s": n / per_iter}
    m = re.search(r"Steady loop time: ([0-9.eE+-]+) s, (\d+) iterations", r.stdout)
    if m:
        out["steady_ms_per_step"] = 1e3 * float(m.group(1)) / int(m.group(2))
        out["steady_iterations"] = int(m.group(2))
        # the library over the SAME stretch: a fresh context, the binary's options, the same initial state, iterations 9 .. K
        try:
            import torch
            from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG
            eng = Engine(fmm_order=order, unsort=0, tree_steps=8, m2l_first=1, sync=0)
            d = torch.from_numpy(gaussian_ball(n)).cuda()
            prm = torch.from_numpy(coulomb_params(n)).cuda()
            eng.compute_force(EVAL_FMM_KDTREE, d, n, prm)
            for k in (1, 8):
                eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, int(m.group(2)))
            torch.cuda.synchronize()
            lib = 1e3 * (time.perf_counter() - t0) / int(m.group(2))
            eng.close()
            out["library_same_stretch_ms_per_step"] = lib
            out["cli_over_library"] = out["steady_ms_per_step"] / lib
        except Exception as e:   # noqa: BLE001
            out["library_same_stretch_error"] = str(e)[:200]
    return out


def cli_dist_leg(gpus, n_system, order, iters, tree_steps, rebalance):
    """The C++ multi-GPU host (one process per GPU over RCCL directly, host/nbco3_dist.cpp) on the same system: its own loop time
    next to the Python-orchestrated number.  Runs while this benchmark's ranks sit idle."""
    import re
    import tempfile
    exe = os.path.join(ROOT, "coulomb_oscillators_amd", "host", "nbco3_dist")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [exe, "-gpus", str(gpus), "-n", str(n_system), "-p", str(order), "-iters", str(iters), "-steps", "100000", "-tree-steps", str(tree_steps),
               "-rebalance", str(rebalance), "-o", tmp]
        # its own process group and a short leash: should the ranks hang in a collective that the supervisor cannot see (it only
        # sees exits), the whole group is killed and the leg says so -- an extra leg must not take the line down
        import signal
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.communicate()
            return {"error": "nbco3_dist did not finish within 240 s (killed)", "command": " ".join(["nbco3_dist"] + cmd[1:-2])}

        class _R:
            returncode, stdout, stderr = p.returncode, so, se
        r = _R
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-300:], "status": r.returncode}
    m = re.search(r"Loop time: ([0-9.eE+-]+) s, (\d+) iterations, (\d+) ranks, partition fallbacks (\d+), capped evaluations (\d+), repeated (\d+)", r.stdout)
    if not m:
        return {"error": "no loop time in the output"}
    per_iter = float(m.group(1)) / int(m.group(2))
    return {"command": " ".join(["nbco3_dist"] + cmd[1:-2]), "loop_s": float(m.group(1)), "iterations": int(m.group(2)), "ranks": int(m.group(3)),
            "partition_fallbacks": int(m.group(4)), "let_capped_evals": int(m.group(5)), "let_redos": int(m.group(6)), "ms_per_step": 1e3 * per_iter, "value": n_system / per_iter, "unit": "particle-steps/s"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))        # nothing below has run: no torch import, no GPU call in this process

    import torch
    import torch.distributed as dist
    from coulomb_oscillators_amd import (Engine, EVAL_DIRECT, EVAL_FMM_KDTREE, EVAL_FMM_TRACELESS, INTEG_LEAPFROG, DomainRun,
                                         SlabRun, TorchComm)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.set_num_threads(1)   # (see spawn_ranks; also when the ranks come from torchrun)
    # Libraries below may write to stdout on their own (RCCL prints a version banner when its first communicator comes up): the
    # process's stdout carries the ONE JSON line and nothing else, everything before it goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        devidx = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(devidx)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", devidx))
        else:
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(0)
        if args.sharded_one:
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                port = s_.getsockname()[1]
            dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device("cuda", 0))
    n = args.n
    sharded = (world > 1 or args.sharded_one) and args.workload == "fmm_kd"
    slabbed = world > 1 and args.workload == "fmm_oct"      # ONE system of --particles, slabs of the cell order (strong scaling)
    kind = {"fmm_kd": EVAL_FMM_KDTREE, "fmm_oct": EVAL_FMM_TRACELESS, "direct": EVAL_DIRECT}[args.workload]
    if args.far_fp64 and args.workload != "fmm_oct":
        raise SystemExit("--far-fp64 needs --workload fmm_oct")
    dom = "direct" if args.workload == "direct" else "p2p"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce(x, op):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())

    fallbacks = []     # every switch from a default form to its all-gather form, with the reason (empty = the defaults ran)

    def timed_run(n_local, steps, warmup, profile, repeats=1):
        """`repeats` times from the same initial state: accelerations, warm-up, then `steps` leapfrog steps between two barriers.
        Returns (elapsed of every repeat, max over ranks; engine, run, run_steps, profile, particles of the system)"""
        n_sys = world * n_local if sharded else n_local      # particles of ONE physical system
        # every rank draws n_local particles of the same Gaussian ball from its own stream; sharded run: their union is the
        # N = world * n_local system, the kd-domains are cut by the first partition
        buf = gaussian_ball(n_local, 0 if slabbed else rank)
        if args.init_state:
            st = np.load(args.init_state).astype(np.float32).reshape(3, -1, 3)
            assert st.shape[1] == n_local and world == 1, "--init-state: one GPU, the particle count of the dump"
            buf = np.ascontiguousarray(st)
        d0 = torch.from_numpy(buf).cuda()
        d = d0.clone()
        prm = torch.from_numpy(coulomb_params(n_sys)).cuda()
        extra_opts = {}
        for kv in args.engine_opt:
            k, v = kv.split("=", 1)
            extra_opts[k] = float(v) if k in ("tree_radius", "eps2", "dens_inhom") else int(v)
        eng = Engine(**{**dict(fmm_order=args.order, unsort=0, tree_steps=args.tree_steps, sync=0, far_fp64=int(args.far_fp64),
                               dens_inhom=args.dens_inhom), **extra_opts})
        run = None
        if sharded:
            run = DomainRun(eng, n_sys, TorchComm(always_collective=args.sharded_one), rebalance=args.rebalance, let=not args.no_let,
                            gather_partition=args.gather_partition or None)
        elif slabbed:
            run = SlabRun(eng, n_local, TorchComm())

        def note(what):
            if what not in fallbacks:
                fallbacks.append(what)

        def start():
            """state := the initial state; accelerations (main3.cu:836-839)"""
            d.copy_(d0)
            if args.tree_steps > 1:
                # the context still holds the tree of the END of the repeat before: a new state starts a new rebuild schedule
                eng.set(tree_steps=1)
                eng.set(tree_steps=args.tree_steps)
            if sharded:
                # The re-partition without gathering the state and the LET exchange are the two stages no single-card rehearsal can
                # run over RCCL between several cards: should one fail on every rank alike, the run agrees on that, carries on with
                # its all-gather form and SAYS SO (`fallbacks`; --strict turns it into a failing exit status).  Pivot ties beyond the
                # distributed select are handled inside DomainRun.partition and counted there.
                ok, why = 1, ""
                before = run.partition_fallbacks
                try:
                    run.partition(d[0].reshape(-1), d[1].reshape(-1))
                except Exception as e:   # noqa: BLE001
                    if not run.dpart:
                        raise
                    print("rank %d: distributed re-partition failed (%s)" % (rank, e), file=sys.stderr)
                    ok, why = 0, str(e)[:160]
                if run.dpart and reduce(float(ok), dist.ReduceOp.MIN) < 1:
                    note("partition: distributed selection failed, all-gather of the state instead (%s)" % (why or "on another rank"))
                    run.use_gather_partition()
                    run.partition(d[0].reshape(-1), d[1].reshape(-1))
                if run.partition_fallbacks > before:
                    note("partition: more pivot ties than the distributed selection resolves, all-gather of the state instead")
                ok, why = 1, ""
                try:
                    run.force(prm)
                    if run.let:
                        eng.dist_let_check()
                except Exception as e:   # noqa: BLE001
                    if not run.let:
                        raise
                    print("rank %d: LET exchange failed (%s)" % (rank, e), file=sys.stderr)
                    ok, why = 0, str(e)[:160]
                if run.let and reduce(float(ok), dist.ReduceOp.MIN) < 1:
                    note("exchange: LET exchange failed, all-gather of whole node and position blocks instead (%s)" % (why or "on another rank"))
                    run.let = False
                    run.last_exchange_bytes = None
                    run.partition(d[0].reshape(-1), d[1].reshape(-1))
                    run.force(prm)
            elif slabbed:
                run.set_state(d[0], d[1])
                run.force(prm)
            else:
                eng.compute_force(kind, d, n_local, prm)

        if sharded:
            step = lambda: run.leapfrog(prm, args.dt)
            # K steps with ONE pass between two force evaluations (nbco_dist_turnaround), as the single-GPU path does
            run_steps = (lambda k: run.leapfrog_steps(prm, args.dt, k)) if not args.step_calls else (lambda k: [step() for _ in range(k)])
        elif slabbed:
            step = lambda: run.leapfrog(prm, args.dt)

            def run_steps(k):
                for _ in range(k):
                    step()
        else:
            step = lambda: eng.integrate(INTEG_LEAPFROG, kind, d, n_local, prm, args.dt)
            # K steps = one nbco_integrate_steps call (the reference's loop between two snapshots, main3.cu:840-870): same final
            # state as K calls of nbco_integrate, bit for bit (tests/test_gpu_integrate_steps.py)
            run_steps = (lambda k: eng.integrate_steps(INTEG_LEAPFROG, kind, d, n_local, prm, args.dt, k)) if not args.step_calls else \
                (lambda k: [step() for _ in range(k)])
        times = []
        for rep in range(max(1, repeats)):
            start()
            run_steps(warmup)
            if profile:      # HIP events around the near-field launches of the timed steps only (all repeats accumulate)
                eng.profile(True if args.profile_all else [dom])
                if rep == 0:
                    eng.profile_reset()
            barrier()
            t0 = time.perf_counter()
            if rep == 0 and os.environ.get("NBCO_BENCH_PYPROFILE") and rank == int(os.environ["NBCO_BENCH_PYPROFILE"]) - 1:   # 1 = rank 0, 2 = rank 1, ..
                # diagnostics: where the host spends the timed steps (python side of a sharded run), top of the cumulative list to stderr
                import cProfile, pstats
                pr = cProfile.Profile()
                pr.enable()
                run_steps(steps)
                pr.disable()
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(30)
            else:
                run_steps(steps)
            barrier()
            times.append(reduce(time.perf_counter() - t0, dist.ReduceOp.MAX))
            if profile:
                eng.profile(False)
        prof = eng.profile_get() if profile else None
        state = run.buf if (sharded or slabbed) else d
        assert torch.isfinite(state).all(), "non-finite state after the timed steps"
        if args.dump_state and profile:
            np.save("%s.rank%d.npy" % (args.dump_state, rank), state.detach().cpu().numpy().reshape(3, -1, 3))
        return times, eng, run, run_steps, prof, n_sys

    times, eng, run, run_steps, prof, n_sys = timed_run(n, args.steps, args.warmup, True, args.repeats)
    elapsed = sorted(times)[len(times) // 2]       # the median repeat is the value

    if args.workload == "fmm_kd":
        info = eng.kd_info()
        pairs_per_eval = int(info.directed_p2p)   # this rank's directed pair interactions per evaluation
        extra = {"L": info.L, "p2p_pairs": int(info.p2p_pairs), "m2l_pairs": int(info.m2l_pairs), "build_mode": int(info.build_mode),
                 "warm_builds": int(info.warm_builds), "warm_misses": int(info.warm_misses),
                 "near_field": ("mutual (Newton III), %d half(s) of 32 per leaf" % info.p2p_halves) if info.p2p_halves else "one-directional"}
        if args.engine_opt:
            extra["engine_opt"] = args.engine_opt
        if sharded:
            extra.update({"n_system": n_sys, "rebalance_every": args.rebalance,
                          "exchange": "LET (all-gather of traversal records + variable all-to-all)" if run.let else "all-gather",
                          # evaluations of this run (last repeat) in the capped form -- segments sized from the evaluation before, no host
                          # round trip in the middle -- and capped attempts that were void and repeated in the exact form
                          "let_capped_evals": getattr(run, "let_capped_evals", 0), "let_redos": getattr(run, "let_redos", 0),
                          "partition": "distributed selection + all-to-all of the movers" if run.dpart else "all-gather of the state + redundant selection",
                          "partition_bytes_per_gpu": run.partition_bytes,
                          "exchange_bytes_per_eval_per_gpu": run.exchange_bytes(),
                          "allgather_bytes_per_eval_per_gpu": run.allgather_bytes(), "backend": args.backend})
    elif args.workload == "fmm_oct":
        info = eng.oct_info()
        pairs_per_eval = 0    # the octree path keeps no pair counter: no roofline entry for this workload
        extra = {"L": info.L, "m2l_entries": int(info.m2l_entries), "p2p_chunks": int(info.p2p_chunks),
                 "far_field": "fp64" if info.real_bytes == 8 else "fp32"}
        if slabbed:
            extra.update({"parallelism": "slabs of the sorted cell keys x%d (state replicated, targets partitioned), one padded all-gather of the acceleration slabs per evaluation" % world,
                          "exchange_bytes_per_eval_per_gpu": run.exchange_bytes(), "slab_particles": [int(y - x) for x, y in zip(run.bounds[:-1], run.bounds[1:])],
                          "backend": args.backend})
    else:
        pairs_per_eval = n * n
        extra = {}
    pairs_all = reduce(float(pairs_per_eval), dist.ReduceOp.SUM)
    legs = not args.no_extra_legs

    # the reference's GPU driver rebuilds the kd-tree every tree_steps = 8 evaluations (fmm_cart3_kdtree.cuh:1619); the
    # headline value above rebuilds every step (CPU-driver semantics), this is the amortised figure beside it
    reuse = None
    if legs and world == 1 and args.workload == "fmm_kd" and args.tree_steps == 1:
        eng.set(tree_steps=8, m2l_first=1)       # exactly what `nbco3` runs with (host/nbco3.cpp; the `cli` leg below)
        run_steps(8)
        barrier()
        t1 = time.perf_counter()
        k8 = max(8, (args.steps // 8) * 8)
        run_steps(k8)
        barrier()
        e8 = time.perf_counter() - t1
        i8 = eng.kd_info()
        reuse = {"tree_steps": 8, "m2l_first": 1, "steps": k8, "ms_per_step": 1e3 * e8 / k8, "value": n * k8 / e8, "unit": "particle-steps/s",
                 "warm_builds": int(i8.warm_builds), "warm_misses": int(i8.warm_misses)}
        eng.set(tree_steps=1, m2l_first=0)

    # the same steps with the mutual (Newton III) near-field kernel, opts.p2p_mutual: its pair kernel runs at a higher fraction of
    # the fp32 peak, the step as a whole is slower (reaction records: written, linked, summed) -- which is why it is not the default
    mutual = None
    if legs and world == 1 and args.workload == "fmm_kd" and not any(o.startswith("p2p_mutual") for o in args.engine_opt):
        eng.set(p2p_mutual=1)
        run_steps(max(2, args.warmup))
        if eng.kd_info().p2p_halves:
            eng.profile([dom])
            eng.profile_reset()
            barrier()
            tm = time.perf_counter()
            run_steps(args.steps)
            barrier()
            em = time.perf_counter() - tm
            pm = eng.profile_get()[dom]
            eng.profile(False)
            pairs_m = int(eng.kd_info().directed_p2p)
            avg_m = pm[0] * 1e-3 / max(pm[1], 1)
            mutual = {"opts": "p2p_mutual=1", "kernel": "p2p_mutual_kernel", "ms_per_step": 1e3 * em / args.steps, "value": n * args.steps / em,
                      "avg_launch_ms": avg_m * 1e3, "pairs_per_launch": pairs_m,
                      "frac": pairs_m * FLOP_PER_PAIR / avg_m / 1e12 / FP32_VECTOR_PEAK_TFLOPS}
        eng.set(p2p_mutual=0)

    # The headline times the first steps of a simulation whose lists grow as it evolves (the ball focuses, a few ejected particles
    # stretch the outer leaves): the same K steps, rebuilt every step, timed again `--late-from` steps in.  The stretch in between
    # runs with the reference GPU driver's tree_steps = 8 (cheaper, same physics to its accuracy).
    late = None
    if legs and world == 1 and args.workload == "fmm_kd" and args.late_from > 0:
        done = args.warmup + args.steps + (8 + max(8, (args.steps // 8) * 8) if reuse else 0) + ((max(2, args.warmup) + args.steps) if mutual else 0)
        todo = max(0, args.late_from - done)
        eng.set(tree_steps=8)
        run_steps(todo - todo % 8)
        eng.set(tree_steps=args.tree_steps)
        run_steps(todo % 8 + 2)
        barrier()
        tl = time.perf_counter()
        run_steps(args.steps)
        barrier()
        el = time.perf_counter() - tl
        il = eng.kd_info()
        late = {"steps_from": done + todo + 2, "steps": args.steps, "tree_steps": args.tree_steps, "ms_per_step": 1e3 * el / args.steps,
                "value": n * args.steps / el, "unit": "particle-steps/s", "directed_pairs_per_eval": int(il.directed_p2p), "p2p_pairs": int(il.p2p_pairs),
                "m2l_pairs": int(il.m2l_pairs), "warm_builds": int(il.warm_builds), "warm_misses": int(il.warm_misses)}
        # where those steps spend their time: a few more of them with HIP events around every phase (outside the timed stretch)
        eng.profile(True)
        eng.profile_reset()
        run_steps(4)
        torch.cuda.synchronize()
        eng.profile(False)
        late["phase_ms_per_step"] = {k: v[0] / 4 for k, v in eng.profile_get().items() if v[1]}
        dp = late["directed_pairs_per_eval"]
        if late["phase_ms_per_step"].get("p2p"):
            late["near_field_frac"] = dp * FLOP_PER_PAIR / (late["phase_ms_per_step"]["p2p"] * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS

    # the metric reads "N = 1M ... at 1/2/4/8 GPUs": ONE system of --particles cut into `world` kd-domains (strong scaling),
    # next to the headline value, which keeps --particles per GPU (weak scaling)
    strong = None
    if legs and sharded and n % world == 0 and n // world >= 4096:
        eng.close()
        t_s, eng_s, run_s, _, _, nsys_s = timed_run(n // world, args.steps, args.warmup, False)
        e_s = t_s[0]
        strong = {"scaling": "strong", "n_system": nsys_s, "n_per_gpu": n // world, "ms_per_step": 1e3 * e_s / args.steps,
                  "value": nsys_s * args.steps / e_s, "unit": "particle-steps/s",
                  "exchange_bytes_per_eval_per_gpu": run_s.exchange_bytes(), "allgather_bytes_per_eval_per_gpu": run_s.allgather_bytes()}
        eng_s.close()

    value = (n if slabbed else world * n) * args.steps / elapsed
    if args.workload == "fmm_kd":
        wl = ("FMM-3D kd-tree p=%d, N=%d per GPU (one system of %d), leapfrog, Gaussian ball, tree rebuilt every %d step(s)"
              % (args.order, n, n_sys, args.tree_steps))
        metric = "particle-steps/sec (+ Gpair-interactions/sec), N=%d FMM-3D kd-tree p=%d, %d GPU(s)" % (n_sys, args.order, world)
    elif args.workload == "fmm_oct":
        wl = "FMM-3D uniform octree, traceless multipoles p=%d, ONE system of N=%d, leapfrog, Gaussian ball" % (args.order, n)
        metric = "particle-steps/sec, N=%d FMM-3D cartesian traceless (octree) p=%d, %d GPU(s)" % (n, args.order, world)
    else:
        wl = "direct O(N^2) 3D, N=%d per GPU, leapfrog" % n
        metric = "particle-steps/sec (+ Gpair-interactions/sec), N=%d direct O(N^2), %d replica(s)" % (n, world)
    out = {
        "metric": metric,
        "value": value,
        "unit": "particle-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "repeats": {"n": len(times), "what": "warm-up + K steps between two barriers, each repeat from the same initial state; value = the median repeat",
                    "ms_per_step": [1e3 * t / args.steps for t in times],
                    "spread": (max(times) - min(times)) / elapsed if len(times) > 1 else None},
        "higher_is_better": True,
        "scaling": "strong" if slabbed else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "gpair_per_s": pairs_all * args.steps / elapsed / 1e9,
        "fallbacks": fallbacks,
        "rccl_ranks": (dist.get_world_size() if (dist.is_initialized() and args.backend == "nccl") else None),
        "tree_reuse": reuse,
        "near_field_mutual": mutual,
        "late_phase": late,
        "strong": strong,
        "config": {"workload": wl, "n_per_gpu": n, "order": args.order, "dt": args.dt,
                   "init": "reference stream mt19937_64(%d + rank), discard %d (main3.cu:662-664)" % (REF_SEED, REF_DISCARD),
                   "parallelism": ("kd-domain sharding x%d, %s per evaluation" % (world, "LET exchange (all-gather of traversal records, all-to-all of the listed multipoles + positions)" if run.let else "one all-gather of nodes + positions")) if sharded
                   else ("single GPU" if world == 1 else "independent replicas x%d" % world), **extra},
    }
    # the near-field launch of every rank: (average duration, directed pairs) -> min / max fraction of the peak over the ranks
    rank_fracs = None
    if world > 1 and pairs_per_eval and prof[dom][1]:
        mine = torch.tensor([prof[dom][0] * 1e-3 / prof[dom][1], float(pairs_per_eval)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_fracs = [float(t[1]) * FLOP_PER_PAIR / float(t[0]) / 1e12 / FP32_VECTOR_PEAK_TFLOPS for t in allr]
    if rank == 0:
        ms, launches = prof[dom]
        if launches and pairs_per_eval:
            avg_s = ms * 1e-3 / launches
            ach = pairs_per_eval * FLOP_PER_PAIR / avg_s / 1e12
            kname = ("p2p_mutual_kernel" if args.workload == "fmm_kd" and info.p2p_halves else "p2p_kernel") if dom == "p2p" else "direct_tiles"
            default_cfg = world == 1 and n == 1048576 and args.order == 6
            traffic, src = measured_traffic(kname) if default_cfg else (None, None)
            ceil, ceil_src = measured_ceiling()
            ceil_frac = (ceil or {}).get("mutual_burst" if kname == "p2p_mutual_kernel" else "pair_burst")
            out["roofline"] = {"bound": "valu_fp32", "kernel": kname,
                               "achieved": ach, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP32_VECTOR_PEAK_TFLOPS,
                               "frac_of_measured_ceiling": (ach / FP32_VECTOR_PEAK_TFLOPS / ceil_frac) if ceil_frac else None,
                               "measured_ceiling_frac": ceil_frac, "measured_ceilings": ceil, "measured_ceiling_source": ceil_src,
                               "traffic": traffic, "traffic_unit": "B/launch", "traffic_source": src,
                               "pairs_per_launch": pairs_per_eval, "avg_launch_ms": avg_s * 1e3, "flop_per_pair": FLOP_PER_PAIR,
                               "launches_averaged": int(launches), "rank": 0,
                               "frac_over_ranks": {"min": min(rank_fracs), "max": max(rank_fracs), "all": rank_fracs} if rank_fracs else None}
        if args.profile_all:
            out["phase_ms_per_step"] = {k: v[0] / args.steps for k, v in prof.items() if v[1]}
    # further single-GPU legs, outside the timed region (rank 0 of a one-GPU run only)
    if rank == 0 and world == 1 and legs and args.workload == "fmm_kd":
        eng.close()
        # BASELINE configs[2] by name: the uniform-octree evaluator with traceless multipoles on the same ball.  The cubic
        # grid over the anisotropic ball leaves 625 of 32 768 cells occupied: an almost all-pairs P2P run.
        try:
            eo = Engine(fmm_order=args.order, unsort=0, sync=0)
            bo = torch.from_numpy(gaussian_ball(n)).cuda()
            po = torch.from_numpy(coulomb_params(n)).cuda()
            eo.compute_force(EVAL_FMM_TRACELESS, bo, n, po)
            torch.cuda.synchronize()
            ko = 3
            t2 = time.perf_counter()
            for _ in range(ko):
                eo.integrate(INTEG_LEAPFROG, EVAL_FMM_TRACELESS, bo, n, po, args.dt)
            torch.cuda.synchronize()
            eo_s = (time.perf_counter() - t2) / ko
            oi = eo.oct_info()
            out["octree_traceless"] = {"workload": "FMM-3D cartesian traceless (uniform octree) p=%d, N=%d, leapfrog, same ball" % (args.order, n),
                                       "steps": ko, "ms_per_step": 1e3 * eo_s, "value": n / eo_s, "unit": "particle-steps/s", "L": oi.L}
            eo.close()
            del bo
        except Exception as e:   # an extra leg must not take the headline down with it
            out["octree_traceless"] = {"error": str(e)[:200]}
        try:
            out["cli"] = cli_leg(n, args.order, args.warmup + args.steps + 26)   # (about the stretch the legs above cover)
        except Exception as e:
            out["cli"] = {"error": str(e)[:200]}
    if world > 1 or args.sharded_one:
        barrier()
        dist.destroy_process_group()       # the other ranks are done; what follows runs on rank 0 with the GPUs idle
    if rank == 0:
        if legs and args.workload == "fmm_kd" and (sharded or world == 1):
            # the C++ multi-GPU host on the same system (weak-scaling shape: world x --particles), same rebuild cadence
            try:
                eng.close()
                out["cli_dist"] = cli_dist_leg(world, n_sys, args.order, args.warmup + args.steps, args.tree_steps, args.rebalance)
            except Exception as e:
                out["cli_dist"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and args.workload != "fmm_oct":
            # (also on the N > 1 lines: rank 0's host, after the timed region; the per-GPU workload of --particles)
            out["cpu_baseline"] = cpu_baseline(args, n)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        if args.strict and fallbacks:
            sys.exit(3)


if __name__ == "__main__":
    main()
