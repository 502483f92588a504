#!/usr/bin/env python3
"""gen_p2p.py -- writes p2p_gen.inc: the pair loop of the near-field kernel (k_p2p.hpp) over ONE staged source tile as a
hand-scheduled instruction stream on hand-numbered registers.

Why not leave it to hipcc: on gfx950 a vector instruction with three VGPR sources (v_fma_f32, and v_fmac_f32, whose
accumulator is the third) issues every 2 cycles only if its three source registers are NOT all even-numbered or all
odd-numbered; otherwise it takes 3.14 (tools/bank_probe.py, profiles/r03a_bank_probe.json: 64 of the 256 bank patterns of
v_fma, exactly the same-parity ones; two-source instructions never conflict; v_rsq_f32 is 4 cycles; an SGPR or a register
named twice does not help).  hipcc allocates registers without regard to this: the compiled pair body had 4 of its 12
instructions on the slow pattern (its `v_fma r2, dz, dz, s_eps2` always is) and ran at 40.6 cycles per 64 pairs against 26
for 11 x 2 + 4 (tools/p2p_lab.hip timeline).  Here the differences dx, dy, dz live in EVEN registers and r^2 / r^-1 / r^-3
and eps^2 in ODD ones, so that

    r2  = fma(dz, dz, eps2)      even even odd
    r2 += dy * dy ; r2 += dx * dx   odd  even even
    acc += d * w                 any  even odd

never meet the pattern, whatever registers the compiler picks for the accumulators and the target.

The block evaluates the `n_src` (a multiple of 4) sources of a tile stored as packed xyz triplets at LDS address %[addr] --
group-uniform ds_read_b128 broadcasts, three per four sources -- against the lane's target (px, py, pz) and adds
(target - source) r^-3 to (tx, ty, tz).  Reads for the next four sources are issued as soon as the subtractions have
consumed the current ones; v_rsq_f32 results are first used two instructions later (the trans-use wait state of gfx940+).
Same arithmetic, same order of the sums as the C++ loop it replaces (P2P_PAIR): bit-identical partial sums.
"""
import sys

R = 44            # first clobbered register (even)


class Layout:
    """v[R .. R+11]: four packed sources; then per source in flight k: dx = T+6k, dy = T+6k+2, dz = T+6k+4 (even), ri = T+6k+1,
    w = T+6k+3 (odd); eps^2 in the spare odd register T+5"""

    def __init__(self, inflight):
        self.inflight = inflight            # 4: all subtractions of a group first (36 registers); 2: pair by pair (24 registers)
        self.S = R
        self.T = R + 12
        self.EPS = self.T + 5
        self.nregs = 12 + 6 * inflight

    def regs(self, k):
        b = self.T + 6 * (k % self.inflight)
        return b, b + 2, b + 4, b + 1, b + 3   # dx dy dz ri w


def tile_block(n_src, lay, diag=None):
    """diag (tools/p2p_lab.hip only): 'noread' = the LDS reads of the first four sources only (what do the reads cost?)"""
    assert n_src % 4 == 0
    groups = n_src // 4
    S, EPS = lay.S, lay.EPS
    out = [f"v_mov_b32 v{EPS}, %[eps]"]

    def reads(q):
        return [f"ds_read_b128 v[{S + 4 * j}:{S + 4 * j + 3}], %[addr] offset:{48 * q + 16 * j}" for j in range(3)]

    def subs(k):   # d = target - source; source k of the group = S[3k .. 3k+2]
        dx, dy, dz, ri, w = lay.regs(k)
        return [f"v_sub_f32 v{dx}, %[px], v{S + 3 * k}", f"v_sub_f32 v{dy}, %[py], v{S + 3 * k + 1}", f"v_sub_f32 v{dz}, %[pz], v{S + 3 * k + 2}"]

    def pair(a, b):   # two sources interleaved: a v_rsq_f32 result is first used two instructions later
        xa, ya, za, ra, wa = lay.regs(a)
        xb, yb, zb, rb, wb = lay.regs(b)
        return [
            f"v_fma_f32 v{ra}, v{za}, v{za}, v{EPS}", f"v_fma_f32 v{rb}, v{zb}, v{zb}, v{EPS}",
            f"v_fmac_f32 v{ra}, v{ya}, v{ya}", f"v_fmac_f32 v{rb}, v{yb}, v{yb}",
            f"v_fmac_f32 v{ra}, v{xa}, v{xa}", f"v_fmac_f32 v{rb}, v{xb}, v{xb}",
            f"v_rsq_f32 v{ra}, v{ra}", f"v_rsq_f32 v{rb}, v{rb}",
            f"v_mul_f32 v{wa}, v{ra}, v{ra}", f"v_mul_f32 v{wb}, v{rb}, v{rb}",
            f"v_mul_f32 v{wa}, v{wa}, v{ra}", f"v_mul_f32 v{wb}, v{wb}, v{rb}",
            f"v_fmac_f32 %[tx], v{xa}, v{wa}", f"v_fmac_f32 %[ty], v{ya}, v{wa}", f"v_fmac_f32 %[tz], v{za}, v{wa}",
            f"v_fmac_f32 %[tx], v{xb}, v{wb}", f"v_fmac_f32 %[ty], v{yb}, v{wb}", f"v_fmac_f32 %[tz], v{zb}, v{wb}",
        ]

    out += reads(0)
    for q in range(groups):
        out.append("s_waitcnt lgkmcnt(0)")
        nxt = reads(q + 1) if q + 1 < groups and diag != "noread" else []
        if lay.inflight == 4:
            out += subs(0) + subs(1) + subs(2) + subs(3) + nxt + pair(0, 1) + pair(2, 3)
        else:
            out += subs(0) + subs(1) + pair(0, 1) + subs(2) + subs(3) + nxt + pair(2, 3)
    return out


def emit(f, name, n_src, lay, diag=None):
    lines = tile_block(n_src, lay, diag)
    clob = ", ".join(f'"v{r}"' for r in range(R, R + lay.nregs))
    f.write(f"// {n_src} sources, {lay.inflight} in flight: {sum(1 for l in lines if l.startswith('v_'))} vector instructions, "
            f"{sum(1 for l in lines if l.startswith('ds_'))} LDS reads; owns v{R}..v{R + lay.nregs - 1}\n")
    f.write(f"__device__ __forceinline__ void {name}(unsigned addr, float px, float py, float pz, float eps2, float &tx, float &ty, float &tz)\n{{\n")
    f.write("\tasm volatile(\n")
    for l in lines:
        f.write(f'\t\t"{l}\\n\\t"\n')
    f.write('\t\t: [tx] "+v"(tx), [ty] "+v"(ty), [tz] "+v"(tz)\n')
    f.write('\t\t: [addr] "v"(addr), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [eps] "v"(eps2)\n')
    f.write(f"\t\t: {clob}, \"memory\");\n}}\n\n")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "p2p_gen.inc"
    with open(out, "w") as f:
        f.write("// generated by gen_p2p.py -- do not edit (register layout and the reason for it: gen_p2p.py)\n\n")
        for n in (8, 16, 32, 64):
            emit(f, f"p2p_tile_gen{n}", n, Layout(2))
        emit(f, "p2p_tile_gen32_wide", 32, Layout(4))
        if len(sys.argv) > 2 and sys.argv[2] == "lab":
            emit(f, "p2p_tile_lab_noread", 32, Layout(2), "noread")


if __name__ == "__main__":
    main()
