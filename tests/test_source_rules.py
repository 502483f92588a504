"""Source-level invariants that the GPU parity tests only catch once in millions of pairs.

The tree geometry and the admissibility test must round exactly like the oracle: no fused multiply-adds.  hipcc contracts a*b + c
into an fma unless `#pragma clang fp contract(off)` is in force where the function is DEFINED; the pragmas toggle through the
files.  (Round 2: `kd_admissible_rec` was added behind a `contract(fast)` line -- 1 M2L decision in 6.8 million differed from the
oracle's, found by tools/fuzz_kd.py.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "coulomb_oscillators_amd", "csrc")

# functions whose float arithmetic decides integers (tree structure, list membership)
MUST_NOT_CONTRACT = {
    "k_fmm_kd.hip": ["kd_admissible", "kd_admissible_rec", "longest_axis", "parent_centre"],
    "k_kdselect.hip": ["ties_and_boxes"],
    "k_fmm_oct.hip": ["oct_scalars_kernel", "oct_keys_kernel"],
    "k_dpart.hip": ["dp_root_kernel", "dp_boxes_kernel"],
}


def _state_at_definitions(path, names):
    state, out = "default", {}
    for line in open(path):
        m = re.search(r"#pragma clang fp contract\((\w+)\)", line)
        if m:
            state = m.group(1)
        for n in names:
            if n not in out and re.search(r"\b%s\s*\(" % re.escape(n), line) and ("__device__" in line or "__global__" in line):
                out[n] = state
    return out


def test_rounding_critical_functions_are_compiled_without_contraction():
    for fname, names in MUST_NOT_CONTRACT.items():
        got = _state_at_definitions(os.path.join(CSRC, fname), names)
        for n in names:
            assert n in got, "%s: definition of %s not found" % (fname, n)
            assert got[n] == "off", "%s: %s is defined under fp contract(%s)" % (fname, n, got[n])
