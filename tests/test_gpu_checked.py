"""The checked build (libnbco_hip_checked.so: every index a kernel reads from a list is range-checked on the device) under
NBCO_POISON=1 (every new scratch allocation filled with 0x7f bytes, the way a recycled allocation holds stale data).

Background (post-mortem of round 1's abort): a traversal that overflowed a frontier region left the region's tail
unwritten, the next launch read it up to its capacity and classified whatever the recycled buffer held -- node ids of a deeper
tree -- and faulted.  A fresh process gets zero-filled memory, which is why the test passed when it was run alone.  The poisoned
allocations make that situation deterministic, the device-side checks turn any such read into a count instead of a fault.

Runs once, in a subprocess (the checked library is selected with NBCO_LIB before the package loads)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKED = os.path.join(ROOT, "coulomb_oscillators_amd", "libnbco_hip_checked.so")

SCRIPT = textwrap.dedent('''
    import json, sys
    import numpy as np, torch
    sys.path.insert(0, %r)
    from coulomb_oscillators_amd import Engine, EngineError, LoopbackWorld, EVAL_FMM_KDTREE, INTEG_LEAPFROG, lib_path
    from oracle.pyoracle import Oracle
    assert lib_path().endswith("libnbco_hip_checked.so")
    o = Oracle(np.float32)
    out = {}
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()

    # 1. the overflow of round 1: one region of list_factor * nodes / 8 pairs, growth off
    n, p = 65536, 6
    buf, par = o.init_reference(n), dev(o.params(n))
    big = Engine(fmm_order=p, unsort=0)                      # leaves node ids of a deeper tree behind in the allocator's pool
    dbig = dev(o.init_reference(1 << 20))
    big.compute_force(EVAL_FMM_KDTREE, dbig, 1 << 20, dev(o.params(1 << 20)))
    torch.cuda.synchronize()
    big.close(); del dbig
    e = Engine(fmm_order=p, unsort=0, list_factor=1, list_grow=0)
    d = dev(buf[:2]); before = d.clone()
    a = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    try:
        e.fmm_cart3_kdtree(d, a, n, par)
        out["overflow_reported"] = False
    except EngineError as err:
        out["overflow_reported"] = "list capacity" in str(err)
    torch.cuda.synchronize()
    out["state_untouched"] = bool(torch.equal(d, before))
    out["violations_after_overflow"] = e.violations()
    e.set(list_factor=48, list_grow=1)
    e.fmm_cart3_kdtree(d, a, n, par)                          # the context keeps working
    torch.cuda.synchronize()
    _, a_ref = o.fmm_kd(buf[:2], o.params(n), p=p, threads=8, unsort=False)
    mag = np.linalg.norm(a_ref, axis=1)
    out["err_after_recovery"] = float((np.linalg.norm(a.cpu().numpy() - a_ref, axis=1) / (mag + mag.mean())).max())
    e.close()

    # 2. the production paths: both near-field kernels, tree reuse, caller's order, odd sizes, octree, two kd-domains
    for (nn, pp, kw) in ((65536, 6, dict(unsort=0, tree_steps=4, p2p_mutual=1)), (65536, 6, dict(unsort=0, p2p_mutual=0)), (5000, 4, dict(unsort=1)),
                         (30001, 5, dict(unsort=1, p2p_mutual=1)), (100000, 6, dict(unsort=0, p2p_mutual=1)), (4097, 3, dict(unsort=0, tree_steps=2))):
        b, pr = dev(o.init_reference(nn)), dev(o.params(nn))
        en = Engine(fmm_order=pp, **kw)
        en.compute_force(EVAL_FMM_KDTREE, b, nn, pr)
        for _ in range(3):
            en.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, b, nn, pr, 5e-4)
        torch.cuda.synchronize()
        assert torch.isfinite(b).all()
        en.close()
    nn = 20000
    b = dev(o.init_reference(nn, test_mode=True)[:2]); acc = torch.zeros((nn, 3), device="cuda")
    en = Engine(fmm_order=6)
    en.fmm_cart3_traceless(b, acc, nn, dev(o.params(nn)))
    torch.cuda.synchronize()
    assert torch.isfinite(acc).all()
    nn = 32768
    st = o.init_reference(nn)
    w = LoopbackWorld([Engine(fmm_order=6, unsort=0, p2p_mutual=1) for _ in range(2)], nn)
    h = nn // 2
    w.partition([dev(st[0][:h]), dev(st[0][h:])], [dev(st[1][:h]), dev(st[1][h:])])
    w.force(dev(o.params(nn)), elastic=False)
    torch.cuda.synchronize()
    assert all(torch.isfinite(r.acc).all() for r in w.runs)
    out["violations_total"] = en.violations()
    print(json.dumps(out))
''') % ROOT


def test_checked_build_with_poisoned_allocations():
    if not os.path.exists(CHECKED):
        pytest.fail("libnbco_hip_checked.so is not built (make -C coulomb_oscillators_amd/csrc)")
    env = dict(os.environ, NBCO_LIB=CHECKED, NBCO_POISON="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["overflow_reported"] and out["state_untouched"]
    assert out["violations_after_overflow"] == [0] * 8, out       # no launch read a slot that nothing had written
    assert out["err_after_recovery"] < 1e-5
    assert out["violations_total"] == [0] * 8, out


def test_production_build_has_no_checks(engine):
    from coulomb_oscillators_amd import EngineError
    with pytest.raises(EngineError, match="not the checked build"):
        engine.violations()
