"""Phase timestamps inside the build and traversal kernels (one workgroup each), from the profiling build of the library:

    make -C coulomb_oscillators_amd/csrc prof
    NBCO_LIB=coulomb_oscillators_amd/libnbco_hip_prof.so python tools/subtree_prof.py [n]

  mark ..  kd_subtree_kernel: 16 s + k = phase k of in-LDS selection level s, 4xx = rank sort, output, leaf centres
  part ..  sel_partition_kernel of level 7: start, loads issued, description fetched, select resolved, classified (10: wave 0 done),
           cursors reserved, stores issued, drained, completion counter, tail
  trav ..  traverse_kernel, every launch: time spent in each phase (program order)

Marks are 100 MHz wall-clock reads of thread 0 after the barriers that end a phase.  NBCO_SUBTREE_TWICE=1 launches the subtree
kernel twice in a row (the marks then belong to the second, instruction-cache-warm launch).  Diagnostics only."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG  # noqa: E402
from coulomb_oscillators_amd import engine as E  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    eng = Engine(fmm_order=6, unsort=0, sync=0)
    import bench
    buf = torch.from_numpy(bench.gaussian_ball(n)).cuda()
    par = torch.from_numpy(bench.coulomb_params(n)).cuda()
    eng.compute_force(EVAL_FMM_KDTREE, buf, n, par)
    for _ in range(5):
        eng.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, buf, n, par, 5e-4)
    torch.cuda.synchronize()
    lib = C.CDLL(E.lib_path())
    out = (C.c_longlong * 512)()
    rc = lib.nbco_debug_subtree_prof(out)
    assert rc == 0, rc
    t = np.array(out[:], dtype=np.int64)
    t0 = t[500]
    rel = {int(k): float((t[k] - t0) / 100.0) for k in range(512) if t[k] != 0}   # microseconds since kernel entry
    keys = sorted(rel, key=lambda k: rel[k])
    prev = 0.0
    rows = []
    for k in keys:
        rows.append((k, round(rel[k], 2), round(rel[k] - prev, 2)))
        prev = rel[k]
    for r in rows:
        print("mark %3d  t=%8.2f us  +%6.2f" % r)
    print(json.dumps({"n": n, "marks": rows}))
    if hasattr(lib, "nbco_debug_trav_prof"):
        out = (C.c_longlong * 432)()
        assert lib.nbco_debug_trav_prof(out) == 0
        tv = np.array(out[:], dtype=np.int64).reshape(36, 12)
        names = {0: "start", 1: "sizes", 2: "pair", 3: "records", 4: "classified", 5: "scanned", 7: "barrier", 6: "slots", 8: "stored", 9: "drained",
                 10: "barrier2", 11: "end"}
        order = [1, 2, 3, 4, 5, 7, 6, 8, 9, 10, 11]   # program order of the marks (the slots are awaited after the barrier)
        print("trav it  " + " ".join("%10s" % names[k] for k in order))
        for it in range(36):
            if tv[it, 0] == 0:
                continue
            row = []
            prev = tv[it, 0]
            for k in order:
                if tv[it, k] == 0:
                    row.append("         -")
                    continue
                row.append("%10.2f" % ((tv[it, k] - prev) / 100.0))
                prev = tv[it, k]
            print("trav %2d  " % it + " ".join(row))
    if hasattr(lib, "nbco_debug_partition_prof"):
        out = (C.c_longlong * 64)()
        assert lib.nbco_debug_partition_prof(out) == 0
        t = np.array(out[:], dtype=np.int64)
        ks = sorted([k for k in range(64) if t[k] != 0], key=lambda k: t[k])
        prev = t[ks[0]] if ks else 0
        for k in ks:
            print("part %3d  t=%8.2f us  +%6.2f" % (k, (t[k] - t[ks[0]]) / 100.0, (t[k] - prev) / 100.0))
            prev = t[k]


if __name__ == "__main__":
    main()
