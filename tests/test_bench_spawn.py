"""bench.py --gpus N without a launcher: the parent starts the ranks itself (before any torch / GPU call), relays rank 0's
JSON line and fails when a rank fails.  The CPU test drives the spawn logic with stub ranks; the GPU test runs the real
two-rank path (DomainRun + the HIP engine + a gloo process group, both ranks on the one card)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_stub(tmp_path, body, gpus=2):
    """a copy of bench.py whose rank processes run `body` instead of main()"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    marker = "    import torch\n    import torch.distributed as dist\n"
    assert marker in src
    src = src.replace(marker, "    assert 'torch' not in sys.modules\n" + body + "\n    return\n" + marker, 1)
    # the spawning parent must not have imported torch either
    src = src.replace("        sys.exit(spawn_ranks(args))", "        assert 'torch' not in sys.modules\n        sys.exit(spawn_ranks(args))", 1)
    path = tmp_path / "bench.py"
    path.write_text(src)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, str(path), "--gpus", str(gpus)], capture_output=True, text=True, env=env, timeout=120)


def test_parent_spawns_ranks_and_relays_rank0(tmp_path):
    body = ("    r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
            "    assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
            "    print('noise from rank %d' % r)\n"
            "    if r == 0:\n"
            "        print(json.dumps({'n_gpus': w, 'rank': r}))")
    r = _run_stub(tmp_path, body, gpus=3)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 3, "rank": 0}


def test_failing_rank_fails_the_run(tmp_path):
    body = ("    r = int(os.environ['RANK'])\n"
            "    if r == 1:\n"
            "        sys.exit(7)\n"
            "    time.sleep(60)")      # the surviving rank would hang in a collective: the parent must end it
    r = _run_stub(tmp_path, body)
    assert r.returncode != 0
    assert "rank 1 exited with status 7" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_two_ranks_on_one_card_through_the_real_path(engine_lib):
    """`bench.py --gpus 2 --backend gloo --particles 65536 --steps 2`: two processes, two engine contexts on the one card,
    collectives over gloo -- DomainRun with the HIP engine and a real process group."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--particles", "65536",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["config"]["n_system"] == 131072
    assert out["value"] > 0 and out["ms_per_step"] > 0          # bench.py itself asserts a finite state on every rank
    assert out["config"]["allgather_bytes_per_eval_per_gpu"] > 0
    assert out["scaling"] == "weak" and out["strong"]["n_system"] == 65536 and out["strong"]["value"] > 0


@pytest.mark.gpu
def test_two_process_run_equals_the_single_gpu_evaluation_bit_for_bit(engine_lib, tmp_path):
    """The same two-process path (LET exchange, distributed re-partition, fused turnaround; gloo between the two processes on
    the one card) with the domains cut before every evaluation: the rank-concatenated [pos | vel | acc] after the timed steps
    equals a single-GPU run of the same 2 x 32768-particle system through the plain ABI -- every row, every bit.  Also: no
    fallback was taken, and the line says so."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from coulomb_oscillators_amd import Engine
    nl, G, p, warm, steps = 32768, 2, 6, 1, 2
    prefix = str(tmp_path / "state")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(G), "--backend", "gloo", "--particles", str(nl), "--steps", str(steps),
                        "--warmup", str(warm), "--rebalance", "1", "--repeats", "1", "--no-extra-legs", "--no-cpu-baseline", "--strict",
                        "--dump-state", prefix], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["fallbacks"] == [] and out["n_gpus"] == G and out["rccl_ranks"] is None      # gloo: no RCCL communicator
    assert "LET" in out["config"]["exchange"] and "distributed" in out["config"]["partition"]
    got = np.concatenate([np.load("%s.rank%d.npy" % (prefix, q)) for q in range(G)], axis=1)
    # the same system on one GPU: rank q drew its nl particles from the stream of seed + q
    n = G * nl
    balls = [bench.gaussian_ball(nl, q) for q in range(G)]
    buf = np.concatenate(balls, axis=1)
    d = torch.from_numpy(buf).cuda()
    prm = torch.from_numpy(bench.coulomb_params(n)).cuda()
    e = Engine(fmm_order=p, unsort=0, tree_steps=1, sync=0)
    dt = float(np.float32(5e-4))

    def force():      # the sharded stages return the Coulomb part, the elastic term is added separately (one rounding more than nbco_force)
        e.fmm_cart3_kdtree(d, d[2], n, prm)
        e.add_elastic(d[0], d[2], n, prm[3:])
    force()
    for _ in range(warm + steps):
        e.step(d[1], d[2], 0.5 * dt, n)
        e.step(d[0], d[1], dt, n)
        force()
        e.step(d[1], d[2], 0.5 * dt, n)
    torch.cuda.synchronize()
    want = d.cpu().numpy()
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[2], want[2])
