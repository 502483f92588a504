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
    "kd_build_kernels.hpp": ["kd_admissible", "longest_axis", "parent_centre"],      # (sections of k_fmm_kd.hip, included there)
    "kd_traverse_kernels.hpp": ["kd_admissible_rec"],
    "k_farfield.hip": ["parent_centre", "centre_of"],
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


# ---- no kernel may ask for the dispatch packet ---------------------------------------------------------------------------------
# An indexable private array (`out[na++]`) is promoted to LDS by the compiler, and the promoted form takes the workgroup size from the
# AQL dispatch packet, which lives in HOST memory: one scalar load across the link at the head of the kernel, 5-25 us per launch
# (kd_subtree_kernel, round 2, found with tools/subtree_prof.py).  The kernel descriptors of the built library say which kernels
# have the dispatch / queue pointer enabled (kernel_code_properties, bits 1 and 2).
def _device_elfs(blob):
    import struct
    at = 0
    while True:
        at = blob.find(b"\x7fELF", at)
        if at < 0:
            return
        if struct.unpack_from("<H", blob, at + 18)[0] == 224:   # EM_AMDGPU
            yield at
        at += 4


def _kernel_descriptors(blob, base):
    import struct
    shoff = struct.unpack_from("<Q", blob, base + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", blob, base + 0x3A)
    secs = []
    for i in range(shnum):
        o = base + shoff + i * shentsize
        name, typ, _flags, addr, off, size, link, _info, _align, entsize = struct.unpack_from("<IIQQQQIIQQ", blob, o)
        secs.append(dict(type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    for s in secs:
        if s["type"] != 2:   # SHT_SYMTAB
            continue
        strtab = secs[s["link"]]
        for k in range(s["size"] // s["entsize"]):
            st_name, _info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", blob, base + s["off"] + k * s["entsize"])
            end = blob.index(b"\0", base + strtab["off"] + st_name)
            name = blob[base + strtab["off"] + st_name:end].decode()
            if not name.endswith(".kd") or size != 64 or shndx == 0 or shndx >= len(secs):
                continue
            sec = secs[shndx]
            kd = base + sec["off"] + (value - sec["addr"])
            yield name[:-3], struct.unpack_from("<H", blob, kd + 56)[0], struct.unpack_from("<I", blob, kd + 4)[0]


def test_no_kernel_reads_the_dispatch_packet():
    lib = os.path.join(ROOT, "coulomb_oscillators_amd", "libnbco_hip.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("libnbco_hip.so is not built")
    blob = open(lib, "rb").read()
    seen, bad = 0, []
    for base in _device_elfs(blob):
        for name, props, _scratch in _kernel_descriptors(blob, base):
            seen += 1
            if props & 0x6:   # ENABLE_SGPR_DISPATCH_PTR | ENABLE_SGPR_QUEUE_PTR
                bad.append(name)
    assert seen > 50, "kernel descriptors not found (%d)" % seen
    assert not bad, "kernels that read the dispatch / queue packet from host memory: %s" % bad


# The kernels of a step's critical path keep everything in registers: a private (scratch) segment means spills or an indexed
# private array -- memory round trips inside latency-bound kernels, and a scratch set-up on every launch.  (Round 2: 36 bytes in
# list_segsort_kernel<true> after two more values per entry were kept across the ranking loop of its 8-entries-per-lane case.)
# The high orders of the generated far-field operators (p >= 8, fp64 above all) do spill; they are not on this list.
NO_SCRATCH = ["kd_subtree_kernel", "sel_partition_kernel", "sel_hist_warm_kernel", "sel_hist_kernel", "traverse_kernel", "traverse_init_kernel",
              "traverse_finish_kernel", "list_fill_kernel", "list_segsort_kernel", "p2p_kernel", "p2p_mutual_kernel", "kd_turnaround_kernel",
              "kd_prep_kernel", "kd_centres_kernel", "kd_centres_top_kernel", "direct_tiles"]


def test_critical_path_kernels_use_no_scratch():
    lib = os.path.join(ROOT, "coulomb_oscillators_amd", "libnbco_hip.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("libnbco_hip.so is not built")
    blob = open(lib, "rb").read()
    found, bad = set(), []
    for base in _device_elfs(blob):
        for name, _props, scratch in _kernel_descriptors(blob, base):
            for k in NO_SCRATCH:
                if k in name:
                    found.add(k)
                    if scratch:
                        bad.append((name, scratch))
    assert found == set(NO_SCRATCH), "kernels not found in the library: %s" % sorted(set(NO_SCRATCH) - found)
    assert not bad, "critical-path kernels with a scratch segment (bytes per lane): %s" % bad


# ---- particles are loaded whole ------------------------------------------------------------------------------------------------
# for `axis_of(P[i], axis)` -- a float4 load followed by a three-way select -- hipcc 7.2 narrowed the load to ONE dword at
# a SELECTED ADDRESS and left the address register unset on the axis == 2 path: a GPU memory fault that only inputs splitting
# along z reached.  The kernels that pick a coordinate by a run-time axis therefore load the particle as a whole
# (k_dpart.hip: load_particle) or take it from registers / LDS that hold whole vectors.  This rule reads the ISA of the built
# library: in those kernels (a) the particle array is read with global_load_dwordx4 / dwordx3 (all three coordinates) and (b) no narrower global / flat load uses an
# address register that a v_cndmask wrote just before it -- the shape of the miscompiled select.
PARTICLE_KERNELS = ["dp_hist_kernel", "dp_ties_kernel", "dp_count_kernel", "dp_scatter_kernel", "sel_hist_kernel", "sel_hist_warm_kernel",
                    "sel_partition_kernel", "kd_subtree_kernel"]


def _disassemble(blob, base):
    import struct
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        return None
    shoff = struct.unpack_from("<Q", blob, base + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", blob, base + 0x3A)
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(blob[base:base + shoff + shentsize * shnum])
        f.flush()
        return subprocess.run([objdump, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout


def test_run_time_axis_kernels_load_whole_particles():
    import pytest
    lib = os.path.join(ROOT, "coulomb_oscillators_amd", "libnbco_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libnbco_hip.so is not built")
    blob = open(lib, "rb").read()
    narrow = re.compile(r"(?:global|flat)_load_(?:dword|dwordx2|ubyte|sbyte|ushort|sshort)\s+v\d+(?:\[[^\]]+\])?, (?:v\[(\d+):(\d+)\]|v(\d+))")
    seen, wide, bad = set(), set(), []
    for base in _device_elfs(blob):
        text = _disassemble(blob, base)
        if text is None:
            pytest.skip("llvm-objdump not found")
        kern, hist = None, []
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                hits = [k for k in PARTICLE_KERNELS if k in m.group(1)]
                kern, hist = (hits[0] if hits else None), []
                if kern:
                    seen.add(kern)
                continue
            ins = line.split("//")[0].strip()
            if not kern or not ins:
                continue
            if ins.startswith("global_load_dwordx4") or ins.startswith("global_load_dwordx3"):   # x, y, z (the unused w may be dropped)
                wide.add(kern)
            m = narrow.match(ins)
            if m:
                regs = {int(m.group(1)), int(m.group(2))} if m.group(1) else {int(m.group(3))}
                for h in hist[-12:]:
                    c = re.match(r"v_cndmask_b32\S*\s+v(\d+),", h)
                    if c and int(c.group(1)) in regs:
                        bad.append((kern, ins, h))
            hist.append(ins)
    assert seen == set(PARTICLE_KERNELS), "kernels not found in the library: %s" % sorted(set(PARTICLE_KERNELS) - seen)
    assert wide == seen, "kernels that never read a whole particle (global_load_dwordx3 / x4): %s" % sorted(seen - wide)
    assert not bad, "narrow loads at a selected address (the shape of the DESIGN 9a miscompile): %s" % bad[:4]
