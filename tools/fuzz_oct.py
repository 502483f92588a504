#!/usr/bin/env python3
"""Randomised differential run of the octree evaluator (fp32 and fp64 far field) against the oracle (GPU box):
python tools/fuzz_oct.py [seed].  Cell keys must be identical, forces within 1e-5 of the fp32 oracle."""
import sys, numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from coulomb_oscillators_amd import Engine
from oracle.pyoracle import Oracle
from nbutil import force_err
o = Oracle(np.float32); o64 = Oracle(np.float64)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(30):
    n = int(rng.choice([rng.integers(2, 300), rng.integers(300, 9000), rng.integers(9000, 120000)]))
    p = int(rng.integers(1, 9)); radius = float(rng.choice([1.0, 1.0, 2.0])); dens = float(rng.choice([1.0, 1.0, 0.5, 8.0]))
    kind = rng.choice(["gauss", "cube", "clumps"])
    f64 = int(rng.integers(0, 2))
    if kind == "gauss": buf = o.init_reference(n)
    elif kind == "cube": buf = o.init_reference(n, test_mode=True)
    else:
        buf = np.zeros((3, n, 3), dtype=np.float32); c = rng.standard_normal((8, 3)).astype(np.float32)
        buf[0] = c[rng.integers(0, 8, n)] + 0.05 * rng.standard_normal((n, 3)).astype(np.float32)
    par = o.params(n)
    pv, want = o.fmm_oct_traceless(buf[:2], par, p=p, threads=8, radius=radius, dens_inhom=dens)
    keys = o.oct_tree(n)["keys"]
    e = Engine(fmm_order=p, tree_radius=radius, dens_inhom=dens, far_fp64=f64)
    d = torch.from_numpy(buf[:2].copy()).cuda(); a = torch.zeros((n, 3), device="cuda")
    try:
        e.fmm_cart3_traceless(d, a, n, torch.from_numpy(par).cuda()); torch.cuda.synchronize()
        same = np.array_equal(e.oct_array("keys").astype(np.int64), keys)
        err = force_err(a.cpu().numpy(), want)
        ok = same and err < 1e-5
        note = ""
        if same and not ok:
            _, w64 = o64.fmm_oct_traceless(buf[:2].astype(np.float64), par.astype(np.float64), p=p, threads=8, radius=radius, dens_inhom=dens)
            if np.array_equal(o64.oct_tree(n)["keys"], keys):
                eg, ec = force_err(a.cpu().numpy(), w64), force_err(want, w64); note = f" [vs fp64: gpu {eg:.2e} cpu32 {ec:.2e}]"; ok = eg <= 2 * ec + 1e-6
        if not ok: bad += 1
        print("OK " if ok else "BAD", f"n={n} p={p} {kind} r={radius} i={dens} f64={f64} L={e.oct_info().L} err={err:.2e} keys={same}{note}", flush=True)
    except Exception as ex:
        bad += 1; print("EXC", n, p, kind, radius, dens, ex, flush=True)
    e.close()
print("bad:", bad)
