"""nbco3 host CLI (coulomb_oscillators_amd/host/nbco3.cpp): argument handling on CPU, simulation and
-test modes on the GPU.  Reference behaviour: main3.cu:247-623 (flags), :629-652 / :855-858 (state
files), :790-811 (-test)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "coulomb_oscillators_amd", "host")
EXE = os.path.join(HOST, "nbco3")


@pytest.fixture(scope="module")
def nbco3(engine_lib):
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return EXE


def run(exe, *args, cwd=None):
    return subprocess.run([exe, *args], capture_output=True, text=True, cwd=cwd, timeout=600)


def test_help_and_argument_errors(nbco3):
    r = run(nbco3, "-h")
    assert r.returncode == 0 and "Usage: nbco3 [options] [input]" in r.stdout
    for args, msg in ((["-n"], "missing argument to '-n'"), (["-n", "0"], "invalid argument to '-n': 0"),
                      (["-ds", "-1"], "invalid argument to '-ds'"), (["-integ", "rk4"], "invalid argument to '-integ'"),
                      (["-eps", "1e-30"], "too small argument to '-eps'"), (["-bogus"], "unrecognised option '-bogus'"),
                      (["-omega0", "1"], "missing argument(s) to '-omega0'"), (["-cpu", "-test"], "need the GPU"),
                      (["-cpu-threads", "0"], "invalid argument to '-cpu-threads'")):
        r = run(nbco3, *args)
        assert r.returncode != 0 and msg in r.stderr, (args, r.stderr)


@pytest.mark.gpu
def test_simulation_snapshots_match_engine(nbco3, engine, oracle32, tmp_path):
    """`nbco3 -n 4096 -p 4 -iters 4 -steps 2`: args.txt, snapshot names / sizes, final state equal to driving the
    C ABI directly from the same initial state; a snapshot resumes as [input]."""
    import torch
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG
    n, p = 4096, 4
    out = tmp_path / "out"
    out.mkdir()
    r = run(nbco3, "-n", str(n), "-p", str(p), "-iters", "4", "-steps", "2", "-o", str(out))
    assert r.returncode == 0, r.stderr
    assert (out / "args.txt").read_text().split()[1:] == ["-n", str(n), "-p", str(p), "-iters", "4", "-steps", "2", "-o", str(out)]
    names = sorted(f for f in os.listdir(out) if f.endswith(".bin"))
    assert names == ["out0_0.000500.bin", "out2_0.000500.bin", "out4_0.000500.bin"]      # -iters n runs n+1 iterations (main3.cu:357)
    # (not in the reference: the wall time of the loop behind the first snapshot, read by bench.py's `cli` leg)
    import re
    m = re.search(r"Loop time: ([0-9.eE+-]+) s, (\d+) iterations", r.stdout)
    assert m and int(m.group(2)) == 4 and float(m.group(1)) > 0
    snap = np.fromfile(out / "out4_0.000500.bin", dtype=np.float32)
    assert snap.size == 2 * n * 3
    snap = snap.reshape(2, n, 3)
    # same run through the ABI: reference init stream, precompute, 5 leapfrog steps, GPU-driver options
    buf = oracle32.init_reference(n)
    par = oracle32.params(n)
    engine.set(fmm_order=p, unsort=0, tree_steps=8, m2l_first=1)
    d = torch.from_numpy(buf.copy()).cuda()
    prm = torch.from_numpy(par).cuda()
    engine.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    for _ in range(5):
        engine.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, float(np.float32(5e-4)))
    got = d.cpu().numpy()
    np.testing.assert_allclose(snap[0], got[0], rtol=0, atol=1e-6 * np.abs(got[0]).max())
    np.testing.assert_allclose(snap[1], got[1], rtol=0, atol=1e-5 * np.abs(got[1]).max())
    # resume from a snapshot: N is inferred from the file size (main3.cu:636)
    out2 = tmp_path / "out2"
    out2.mkdir()
    r = run(nbco3, "-p", str(p), "-iters", "0", "-steps", "1", "-o", str(out2), str(out / "out0_0.000500.bin"))
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(out2 / "out0_0.000500.bin") == 2 * n * 12


@pytest.mark.gpu
def test_cli_test_mode_reproduces_reference_table(nbco3):
    """`nbco3 -n 4096 -test` prints the mean relative errors the reference's own run recorded."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_recorded.json")) as f:
        want = json.load(f)["test_mode_relerr"]["values"]
    r = run(nbco3, "-n", "4096", "-test")
    assert r.returncode == 0, r.stderr
    got = [float(l.split("Relative error:")[1]) for l in r.stdout.splitlines() if "Relative error:" in l]
    assert len(got) == 10
    # the CLI runs the GPU traversal order (m2l_first), which moves a few leaf pairs from P2P to M2L:
    # errors are at the reference level, not identical to the CPU-path table
    for g, w in zip(got, want):
        assert 0.5 * w < g < 1.6 * w, (got, want)
    assert "Average time:" in r.stdout


@pytest.mark.gpu
def test_cli_reuse_mode_errors_stay_at_truncation_level(nbco3):
    """`nbco3 -n 8192 -p 4 -test2` (main3.cu:812-831): tree_steps + 1 evaluations while the particles move in the trap and the
    tree is reused; every printed error stays at the order-4 truncation level."""
    r = run(nbco3, "-n", "8192", "-p", "4", "-test2")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("Relative error after")]
    assert [int(l.split()[3]) for l in lines] == list(range(9))          # tree_steps = 8 (constants.cuh:45) -> 9 evaluations
    errs = [float(l.split(":")[1]) for l in lines]
    assert all(0 < e < 0.1 for e in errs), errs                          # order-4 truncation level on the Gaussian ball: 0.05
    assert max(errs) < 1.5 * errs[0], errs                                   # reuse does not degrade the first evaluation's error much


@pytest.mark.gpu
def test_cli_accuracy_search_stays_inside_the_grid(nbco3, tmp_path):
    """`nbco3 -n 4096 -accuracy 1e-2 -iters 0` (main3.cu:737-788): prints the best (r, p) of the search grid :739-740 and an
    error below the bound, then runs the simulation with them."""
    out = tmp_path / "out"
    out.mkdir()
    r = run(nbco3, "-n", "4096", "-accuracy", "1e-2", "-iters", "0", "-steps", "1", "-o", str(out))
    assert r.returncode == 0, r.stderr
    assert r.stdout.count(".") >= 42                                        # one dot per grid point (7 radii x 6 orders)
    best = [l for l in r.stdout.splitlines() if l.startswith("Best parameters:")]
    assert len(best) == 1, r.stdout
    f = dict(kv.split(" = ") for kv in best[0][len("Best parameters: "):].split(", "))
    assert float(f["r"]) in [pytest.approx(v) for v in (1.11, 1.25, 1.43, 1.67, 2, 2.5, 3)]
    assert int(f["p"]) in range(1, 7)
    assert 0 < float(f["error"]) < 1e-2 and float(f["time"]) > 0
    assert os.path.getsize(out / "out0_0.000500.bin") == 2 * 4096 * 12
    # an impossible bound fails the way the reference does
    r = run(nbco3, "-n", "4096", "-accuracy", "1e-12", "-o", str(out))
    assert r.returncode != 0 and "Optimization failed!" in r.stdout


def test_snap2d_writes_the_viewer_format(nbco3, tmp_path):
    """nbco_snap2d: fp32 [pos n x 3 | vel n x 3] -> doubles [pos n x 2 | vel n x 2] (Graphics/main.cpp:155,181-184), single files
    and a renamed series out<steps k>_<dt>.bin -> out<20 k>_0.005000.bin"""
    tool = os.path.join(HOST, "nbco_snap2d")
    rng = np.random.default_rng(3)
    n = 257
    src, dst = tmp_path / "in", tmp_path / "view"
    src.mkdir(); dst.mkdir()
    states = [rng.standard_normal((2, n, 3)).astype(np.float32) for _ in range(3)]
    for k, s in enumerate(states):
        s.tofile(src / ("out%d_0.000500.bin" % (200 * k)))
    r = run(tool, "-axes", "xz", str(src / "out0_0.000500.bin"), str(dst / "one.bin"))
    assert r.returncode == 0, r.stderr
    one = np.fromfile(dst / "one.bin", dtype=np.float64).reshape(2, n, 2)
    np.testing.assert_array_equal(one, states[0][:, :, [0, 2]].astype(np.float64))
    r = run(tool, "-series", str(src), "200", "5e-4", str(dst))
    assert r.returncode == 0 and "3 frame(s) written" in r.stdout, r.stderr
    for k, s in enumerate(states):
        f = np.fromfile(dst / ("out%d_0.005000.bin" % (20 * k)), dtype=np.float64)
        assert f.size == 4 * n                                             # the viewer reads nBodies = bytes / 4 / sizeof(double)
        np.testing.assert_array_equal(f.reshape(2, n, 2), s[:, :, :2].astype(np.float64))
    (src / "bad.bin").write_bytes(b"12345")
    assert run(tool, str(src / "bad.bin"), str(dst / "x.bin")).returncode != 0
    assert run(tool, "-axes", "xx", str(src / "out0_0.000500.bin"), str(dst / "x.bin")).returncode != 0
    assert run(tool, str(src / "missing.bin"), str(dst / "x.bin")).returncode != 0


@pytest.mark.gpu
def test_snapshots_in_input_order(nbco3, oracle32, tmp_path):
    """`-snapshot-order input`: every particle keeps its row through all tree rebuilds (the engine composes the permutations,
    opts.track_order); the same run in tree order holds the same particles in another order"""
    n, p, iters = 8192, 4, 24          # tree_steps = 8: the tree is rebuilt -- and the state permuted -- four times
    a, b = tmp_path / "tree", tmp_path / "input"
    a.mkdir(); b.mkdir()
    common = ["-n", str(n), "-p", str(p), "-iters", str(iters), "-steps", str(iters)]
    assert run(nbco3, *common, "-o", str(a)).returncode == 0
    r = run(nbco3, *common, "-snapshot-order", "input", "-o", str(b))
    assert r.returncode == 0, r.stderr
    name = "out%d_0.000500.bin" % iters
    st_tree = np.fromfile(a / name, dtype=np.float32).reshape(2, n, 3)
    st_in = np.fromfile(b / name, dtype=np.float32).reshape(2, n, 3)
    key = lambda x: np.lexsort((x[:, 2], x[:, 1], x[:, 0]))
    np.testing.assert_array_equal(st_tree[0][key(st_tree[0])], st_in[0][key(st_in[0])])      # same particles ...
    assert not np.array_equal(st_tree[0], st_in[0])                                          # ... other order
    # identity: after 25 steps of dt = 5e-4 a particle has moved ~1 % of sigma from where the initial state put it
    init = oracle32.init_reference(n)
    sig = np.array([0.003, 0.001, 0.01], dtype=np.float32)
    assert (np.abs(st_in[0] - init[0]) / sig).max() < 0.25
    assert (np.abs(st_tree[0] - init[0]) / sig).max() > 1.0
    assert run(nbco3, "-snapshot-order", "sideways").returncode != 0


def test_cpu_path_runs_config_one_without_a_gpu(nbco3, oracle32, tmp_path):
    """`nbco3 -cpu` (BASELINE config 1: direct O(N^2), leapfrog, C++20 threads, no GPU involved): same files as the GPU run, and the
    trajectory equals the oracle's direct3 leapfrog bit for bit (both sum in the reference's order, direct.cuh:192-226)"""
    from oracle.pyoracle import KIND_DIRECT3, SCHEME_LEAPFROG
    n, iters = 600, 5
    out = tmp_path / "out"
    out.mkdir()
    r = run(nbco3, "-cpu", "-cpu-threads", "3", "-n", str(n), "-iters", str(iters), "-steps", str(iters), "-o", str(out))
    assert r.returncode == 0, r.stderr
    snap = np.fromfile(out / ("out%d_0.000500.bin" % iters), dtype=np.float32).reshape(2, n, 3)
    buf = oracle32.init_reference(n)
    par = oracle32.params(n)
    oracle32.compute_force(KIND_DIRECT3, buf, par, threads=2)
    for _ in range(iters + 1):                              # -iters n runs n + 1 iterations (main3.cu:357)
        oracle32.integrate(SCHEME_LEAPFROG, KIND_DIRECT3, buf, par, float(np.float32(5e-4)), threads=2)
    np.testing.assert_array_equal(snap, buf[:2])
    # the other integrators run too
    for integ in ("eu", "fr", "pefrl"):
        assert run(nbco3, "-cpu", "-n", "64", "-iters", "1", "-steps", "1", "-integ", integ, "-o", str(out)).returncode == 0


def test_dist_host_argument_errors(nbco3):
    tool = os.path.join(HOST, "nbco3_dist")
    assert "Usage: nbco3_dist -gpus G" in run(tool, "-h").stdout
    for args in (["-gpus", "3"], ["-gpus", "2", "-n", "4097"], ["-bogus"], ["-n"]):
        assert run(tool, *args).returncode != 0, args


def test_dist_host_supervisor_stops_the_other_ranks_when_one_fails(nbco3, tmp_path):
    """The launcher is a supervisor, not a rank: when a rank exits with a failure, the ranks that would otherwise wait in their
    next collective for ever are killed and the exit status is non-zero.  Here (no GPU needed): rank 1 never comes back
    (NBCO3_DIST_HANG), rank 0 fails on its own -- without a GPU at its first HIP / RCCL call, with ONE GPU because two were asked
    for -- and the command must return promptly with a failure instead of hanging."""
    import time
    tool = os.path.join(HOST, "nbco3_dist")
    t0 = time.time()
    r = subprocess.run([tool, "-gpus", "2", "-n", "8192", "-iters", "2", "-o", str(tmp_path)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, NBCO3_DIST_HANG="1"))
    assert r.returncode != 0
    assert "stopping the other 1 rank(s)" in r.stderr, r.stderr[-1000:]
    assert time.time() - t0 < 60


@pytest.mark.gpu
def test_dist_host_failing_rank_gives_a_failing_exit_status(nbco3, tmp_path):
    """a rank that dies in the middle of the loop (test hook: status 9 before iteration 2) -> the launcher reports it"""
    tool = os.path.join(HOST, "nbco3_dist")
    r = subprocess.run([tool, "-gpus", "1", "-n", "8192", "-p", "3", "-iters", "4", "-steps", "4", "-o", str(tmp_path)], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, NBCO3_DIST_FAIL="0:2"))
    assert r.returncode == 9, (r.returncode, r.stderr[-1000:])
    assert "rank 0 exited with status 9" in r.stderr


@pytest.mark.gpu
def test_dist_host_falls_back_to_the_gathered_partition_on_pivot_ties(nbco3, tmp_path):
    """Positions on a lattice: with several ranks far more than 64 particles tie with every pivot, which the distributed
    re-partition reports as NBCO_ERR_UNSUPPORTED (tests/test_gpu_dist.py checks that report, in lockstep).  One rank selects no
    pivot, so here the report is injected (NBCO3_DIST_FAKE_TIES): the host switches to nbco_dist_partition for the rest of the run
    instead of exiting, and the run equals one started with -partition gather bit for bit.  -tree-steps is exercised on the way."""
    tool = os.path.join(HOST, "nbco3_dist")
    env = dict(os.environ, NBCO3_DIST_QUANTISE="2e-4", NBCO3_DIST_FAKE_TIES="1")
    snaps = []
    for mode in (("-partition", "dist"), ("-partition", "gather")):
        out = tmp_path / mode[1]
        out.mkdir()
        r = subprocess.run([tool, "-gpus", "1", "-n", "16384", "-p", "3", "-iters", "5", "-steps", "5", "-rebalance", "2", "-tree-steps", "2", "-o", str(out), *mode],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("partition fallbacks 1" if mode[1] == "dist" else "partition fallbacks 0") in r.stdout, r.stdout[-300:]
        snaps.append(np.fromfile(out / "out5_0.000500.bin", dtype=np.float32))
    assert np.isfinite(snaps[0]).all()
    np.testing.assert_array_equal(snaps[0], snaps[1])


@pytest.mark.gpu
def test_dist_host_capped_exchange_equals_the_exact_one(nbco3, tmp_path):
    """`-exchange let` (default) runs every evaluation after the first with nbco_dist_let_pack_capped / _finish_capped / _settle --
    no host synchronisation in the middle of an evaluation -- and `-exchange let-exact` never does: same snapshots bit for bit.
    With NBCO3_DIST_VOID=2 the third capped attempt is declared void (as if a count had outgrown its segment) and repeated in the
    exact form: still the same snapshots, through tree reuse as well."""
    import re
    tool = os.path.join(HOST, "nbco3_dist")
    base = [tool, "-gpus", "1", "-n", "32768", "-p", "4", "-iters", "9", "-steps", "9", "-rebalance", "4", "-tree-steps", "2"]
    snaps = {}
    for name, extra, env in (("capped", (), {}), ("exact", ("-exchange", "let-exact"), {}), ("void", (), {"NBCO3_DIST_VOID": "2"})):
        out = tmp_path / name
        out.mkdir()
        r = subprocess.run(base + ["-o", str(out), *extra], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        m = re.search(r"capped evaluations (\d+), repeated (\d+)", r.stdout)
        assert m, r.stdout[-300:]
        capped, repeated = int(m.group(1)), int(m.group(2))
        if name == "capped":
            assert capped == 10 and repeated == 0     # 1 + 10 evaluations, the first one exact
        elif name == "exact":
            assert capped == 0 and repeated == 0
        else:
            assert capped == 9 and repeated == 1
        snaps[name] = np.fromfile(out / "out9_0.000500.bin", dtype=np.float32)
    assert np.isfinite(snaps["capped"]).all()
    np.testing.assert_array_equal(snaps["capped"], snaps["exact"])
    np.testing.assert_array_equal(snaps["void"], snaps["exact"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("-exchange", "gather", "-partition", "gather"), ("-exchange", "let", "-partition", "gather")])
def test_dist_host_one_rank_over_rccl(nbco3, engine, oracle32, tmp_path, mode):
    """`nbco3_dist -gpus 1`: the C++ multi-GPU host (one process per GPU, RCCL all-gathers on a stream of their own, the
    two-stage exchange of INTEGRATION.md section 4) with a world of one -- every collective and every nbco_dist_* stage runs.  Its
    snapshot equals the single-GPU engine driven through the plain ABI (same tree order, same forces: the sharded evaluation
    is the single-GPU one bit for bit with the one-directional near-field kernel)."""
    import torch
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG
    tool = os.path.join(HOST, "nbco3_dist")
    n, p, iters = 32768, 5, 3
    out = tmp_path / "out"
    out.mkdir()
    # -rebalance 1: the domains are cut again before every evaluation, so the (one) domain's root box is the current bounding box
    # as in the single-GPU build; between cuts a domain keeps the union of its inherited box and its particles' bounds, which is
    # a slightly different -- equally valid -- tree (tests/test_gpu_dist.py::test_sharded_stale_domains_stay_correct)
    # mode: default = LET exchange + re-partition without gathering the state (grouped ncclSend / ncclRecv, all-reduces and
    # all-gathers named by the library); the all-gather forms are kept behind -exchange / -partition
    r = run(tool, "-gpus", "1", "-n", str(n), "-p", str(p), "-iters", str(iters), "-steps", str(iters), "-rebalance", "1", "-o", str(out), *mode)
    assert r.returncode == 0, r.stderr[-2000:]
    snap = np.fromfile(out / ("out%d_0.000500.bin" % iters), dtype=np.float32).reshape(2, n, 3)
    buf = oracle32.init_reference(n)
    par = oracle32.params(n)
    engine.set(fmm_order=p, unsort=0, tree_steps=1)
    d = torch.from_numpy(buf.copy()).cuda()
    prm = torch.from_numpy(par).cuda()
    def force():
        # the sharded stages return the Coulomb part; the host adds the elastic term with nbco_add_elastic (the fused
        # nbco_force rounds that sum once less)
        engine.fmm_cart3_kdtree(d, d[2], n, prm)
        engine.add_elastic(d[0], d[2], n, prm[3:])
    force()
    for _ in range(iters + 1):
        # the host composes K D F K from nbco_step calls (integrator.cuh:68-96); same arithmetic as the unfused ABI sequence
        engine.step(d[1], d[2], 0.5 * float(np.float32(5e-4)), n)
        engine.step(d[0], d[1], float(np.float32(5e-4)), n)
        force()
        engine.step(d[1], d[2], 0.5 * float(np.float32(5e-4)), n)
    got = d.cpu().numpy()
    np.testing.assert_array_equal(snap[0], got[0])
    np.testing.assert_array_equal(snap[1], got[1])
