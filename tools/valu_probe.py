#!/usr/bin/env python3
"""valu_probe.py -- writes build/valu_probe.hip: hand-allocated instruction streams that split the VALU "ceiling" of gfx950
into CLOCK and ISSUE.

Every kernel runs one straight-line block of vector instructions on explicitly numbered registers `iters` times.  Each wave
stamps s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its loop, so the host can print, per variant and
per occupancy:
    cycles per instruction  = d(s_memtime) / (iters * instructions * waves per SIMD)
    clock held              = d(s_memtime) / d(s_memrealtime) * 100 MHz
(MI355X_MICROARCH.md, "DVFS give-back" item 6).  tools/pair_ceiling.hip only prints wall-clock rates.

Register banks: VGPR n lives in bank n % 4.  The variants of one instruction differ only in which banks its operands come
from, so an issue cost above 2 cycles that moves with the bank pattern is an operand-fetch conflict.

    python3 tools/valu_probe.py build/valu_probe.hip && hipcc --offload-arch=gfx950 -O2 -o build/valu_probe build/valu_probe.hip
"""
import sys

KERNELS = []   # (name, n_instr_per_block, flop_per_block_per_lane, pairs_per_block_per_lane, asm lines)


def add(name, lines, flop=0, pairs=0, n=None):
    KERNELS.append((name, n if n is not None else len(lines), flop, pairs, lines))


NACC = 24   # accumulators v16.. (stride chosen per variant)

# ---- plain FMA, operand banks varied -------------------------------------------------------------------------------------
# VOP3 fma: dst = a*b + c.  acc registers v16+4k (bank 0), multiplier / addend in chosen banks.
def fma_block(acc_of, b, c, n=48):
    return [f"v_fma_f32 v{acc_of(k)}, v{acc_of(k)}, v{b}, v{c}" for k in range(n)]

# accumulators spread over all banks (v16..v63), b = v1 (bank 1), c = v2 (bank 2)
add("fma_mixed", fma_block(lambda k: 16 + k, 1, 2), flop=96)
# all three sources in different banks: acc bank 0, b bank 1, c bank 2
add("fma_b012", fma_block(lambda k: 16 + 4 * (k % 24), 1, 2), flop=96)
# acc and b in the same bank (0, 0, 2)
add("fma_b002", fma_block(lambda k: 16 + 4 * (k % 24), 4, 2), flop=96)
# all three in one bank (0, 0, 0)
add("fma_b000", fma_block(lambda k: 16 + 4 * (k % 24), 4, 8), flop=96)
# VOP2 fmac with the same patterns: dst += a*b
add("fmac_b012", [f"v_fmac_f32 v{16 + 4 * (k % 24)}, v1, v2" for k in range(48)], flop=96)
add("fmac_b011", [f"v_fmac_f32 v{16 + 4 * (k % 24)}, v1, v5" for k in range(48)], flop=96)
add("fmac_b000", [f"v_fmac_f32 v{16 + 4 * (k % 24)}, v4, v8" for k in range(48)], flop=96)
# two-operand instructions
add("mul_b12", [f"v_mul_f32 v{16 + k}, v1, v2" for k in range(48)], flop=48)
add("sub_b12", [f"v_sub_f32 v{16 + k}, v1, v2" for k in range(48)], flop=48)
add("mov", [f"v_mov_b32 v{16 + k}, v1" for k in range(48)])
# DPP forms
add("sub_dpp", [f"v_sub_f32_dpp v{16 + k}, v1, v2 row_ror:{1 + k % 15} row_mask:0xf bank_mask:0xf" for k in range(48)], flop=48)
add("mov_dpp", [f"v_mov_b32_dpp v{16 + k}, v1 row_ror:{1 + k % 15} row_mask:0xf bank_mask:0xf" for k in range(48)])
add("fmac_dpp", [f"v_fmac_f32_dpp v{16 + 4 * (k % 24)}, v1, v2 row_ror:{1 + k % 15} row_mask:0xf bank_mask:0xf" for k in range(48)], flop=96)
# packed fp32 (two fp32 operations per lane and instruction, operands in aligned register pairs): does a wave issue it as fast as
# the plain form -- twice the arithmetic per issue slot -- or at half the rate?
add("pk_fma", [f"v_pk_fma_f32 v[{16 + 2 * (k % 24)}:{17 + 2 * (k % 24)}], v[{16 + 2 * (k % 24)}:{17 + 2 * (k % 24)}], v[2:3], v[4:5]" for k in range(48)], flop=192)
add("pk_mul", [f"v_pk_mul_f32 v[{16 + 2 * (k % 24)}:{17 + 2 * (k % 24)}], v[2:3], v[4:5]" for k in range(48)], flop=96)
add("pk_add", [f"v_pk_add_f32 v[{16 + 2 * (k % 24)}:{17 + 2 * (k % 24)}], v[2:3], v[4:5]" for k in range(48)], flop=96)
# .. and next to the transcendental unit: 12 packed fma (= 24 fma) per rsq pair
_pkmix = []
for g in range(4):
    _pkmix += [f"v_pk_fma_f32 v[{16 + 2 * ((6 * g + k) % 24)}:{17 + 2 * ((6 * g + k) % 24)}], v[{16 + 2 * ((6 * g + k) % 24)}:{17 + 2 * ((6 * g + k) % 24)}], v[2:3], v[4:5]" for k in range(6)]
    _pkmix += [f"v_rsq_f32 v{120 + 2 * g}, v6", f"v_rsq_f32 v{121 + 2 * g}, v7"]
add("pk_fma6_rsq2", _pkmix, flop=96)
# the packed pair body of p2p_kernel on register operands: per step two pairs = 3 pk_add, 3 pk_fma (r^2), 2 v_rsq, 2 pk_mul, 3 pk_fma;
# `sets` independent steps interleaved instruction by instruction
def pk_pair_steps(sets):
    # registers per set: d (6: 3 pairs), r (2), c (2); accumulators shared v[8:13]; target pairs v[2:3] v[4:5] v[6:7]; sources v[14:19]; eps s/v[20:21]
    out = []
    base = [24 + 10 * k for k in range(sets)]
    def ins(f):
        for b in base:
            out.append(f(b))
    ins(lambda b: f"v_pk_add_f32 v[{b}:{b+1}], v[2:3], v[14:15] neg_lo:[0,1] neg_hi:[0,1]")
    ins(lambda b: f"v_pk_add_f32 v[{b+2}:{b+3}], v[4:5], v[16:17] neg_lo:[0,1] neg_hi:[0,1]")
    ins(lambda b: f"v_pk_add_f32 v[{b+4}:{b+5}], v[6:7], v[18:19] neg_lo:[0,1] neg_hi:[0,1]")
    ins(lambda b: f"v_pk_fma_f32 v[{b+6}:{b+7}], v[{b+4}:{b+5}], v[{b+4}:{b+5}], v[20:21]")
    ins(lambda b: f"v_pk_fma_f32 v[{b+6}:{b+7}], v[{b+2}:{b+3}], v[{b+2}:{b+3}], v[{b+6}:{b+7}]")
    ins(lambda b: f"v_pk_fma_f32 v[{b+6}:{b+7}], v[{b}:{b+1}], v[{b}:{b+1}], v[{b+6}:{b+7}]")
    ins(lambda b: f"v_rsq_f32 v{b+6}, v{b+6}")
    ins(lambda b: f"v_rsq_f32 v{b+7}, v{b+7}")
    ins(lambda b: f"v_pk_mul_f32 v[{b+8}:{b+9}], v[{b+6}:{b+7}], v[{b+6}:{b+7}]")
    ins(lambda b: f"v_pk_mul_f32 v[{b+8}:{b+9}], v[{b+8}:{b+9}], v[{b+6}:{b+7}]")
    ins(lambda b: f"v_pk_fma_f32 v[8:9], v[{b}:{b+1}], v[{b+8}:{b+9}], v[8:9]")
    ins(lambda b: f"v_pk_fma_f32 v[10:11], v[{b+2}:{b+3}], v[{b+8}:{b+9}], v[10:11]")
    ins(lambda b: f"v_pk_fma_f32 v[12:13], v[{b+4}:{b+5}], v[{b+8}:{b+9}], v[12:13]")
    return out
for _sets in (1, 2, 4):
    add(f"pk_pair_x{_sets}", pk_pair_steps(_sets), flop=40 * _sets, pairs=2 * _sets)
# transcendental
add("rsq", [f"v_rsq_f32 v{16 + k}, v{1 + k % 3}" for k in range(48)])
# 12 fma : 1 rsq
_mix = []
for g in range(4):
    _mix += [f"v_fma_f32 v{16 + 4 * (12 * g + k) % 96}, v{16 + 4 * (12 * g + k) % 96}, v1, v2" for k in range(12)]
    _mix.append(f"v_rsq_f32 v{120 + g}, v3")
add("fma12_rsq1", _mix, flop=96)

# ---- one-directional pair body (k_p2p.hpp P2P_PAIR): target in v4..v6, eps2 in v7, sums v8..v10, source k in v(16+4k..) ----
def pair_body(src, tmp, tgt=(4, 5, 6), acc=(8, 9, 10), eps=7):
    sx, sy, sz = src
    dx, dy, dz, r2, w = tmp
    return [
        f"v_sub_f32 v{dx}, v{tgt[0]}, v{sx}", f"v_sub_f32 v{dy}, v{tgt[1]}, v{sy}", f"v_sub_f32 v{dz}, v{tgt[2]}, v{sz}",
        f"v_fma_f32 v{r2}, v{dz}, v{dz}, v{eps}", f"v_fma_f32 v{r2}, v{dy}, v{dy}, v{r2}", f"v_fma_f32 v{r2}, v{dx}, v{dx}, v{r2}",
        f"v_rsq_f32 v{r2}, v{r2}",
        f"v_mul_f32 v{w}, v{r2}, v{r2}", f"v_mul_f32 v{w}, v{w}, v{r2}",
        f"v_fmac_f32 v{acc[0]}, v{dx}, v{w}", f"v_fmac_f32 v{acc[1]}, v{dy}, v{w}", f"v_fmac_f32 v{acc[2]}, v{dz}, v{w}",
    ]

# four sources back to back, each with its own temporaries; serial inside a source (what a naive schedule gives one wave)
_p = []
for k in range(4):
    _p += pair_body((16 + 4 * k, 17 + 4 * k, 18 + 4 * k), (40 + 8 * k, 41 + 8 * k, 42 + 8 * k, 43 + 8 * k, 44 + 8 * k))
add("pair_serial", _p, flop=80, pairs=4)
# the same four sources interleaved instruction by instruction (what hipcc emits for the unrolled tile loop)
_q = [pair_body((16 + 4 * k, 17 + 4 * k, 18 + 4 * k), (40 + 8 * k, 41 + 8 * k, 42 + 8 * k, 43 + 8 * k, 44 + 8 * k)) for k in range(4)]
add("pair_interleaved", [_q[k][i] for i in range(12) for k in range(4)], flop=80, pairs=4)

# ---- mutual step (p2p_mutual_kernel): target v4..6, source v12..14 (rotated by DPP), eps v7, a: v8..10, b: v20..22 ----------
def mutual_step(s, tmp, nop=True, react=True, vop2=False):
    dx, dy, dz, r2, w, wr = tmp
    ror = f"row_ror:{s} row_mask:0xf bank_mask:0xf"
    back = f"row_ror:{(16 - s) % 16} row_mask:0xf bank_mask:0xf"
    if vop2:
        # d = target - source (v_subrev), so the own sums are plain VOP2 v_fmac (4-byte encodings) and the reaction takes the
        # negation as a DPP source modifier
        out = [
            f"v_subrev_f32_dpp v{dx}, v12, v4 {ror}", f"v_subrev_f32_dpp v{dy}, v13, v5 {ror}", f"v_subrev_f32_dpp v{dz}, v14, v6 {ror}",
            f"v_fma_f32 v{r2}, v{dz}, v{dz}, v7", f"v_fmac_f32 v{r2}, v{dy}, v{dy}", f"v_fmac_f32 v{r2}, v{dx}, v{dx}",
            f"v_rsq_f32 v{r2}, v{r2}",
            f"v_mul_f32 v{w}, v{r2}, v{r2}", f"v_mul_f32 v{w}, v{w}, v{r2}",
            f"v_fmac_f32 v8, v{dx}, v{w}", f"v_fmac_f32 v9, v{dy}, v{w}", f"v_fmac_f32 v10, v{dz}, v{w}",
        ]
        if react:
            if nop:
                out.append("s_nop 1")
            out += [f"v_mov_b32_dpp v{wr}, v{w} {back}",
                    f"v_fmac_f32_dpp v20, -v{dx}, v{wr} {back}", f"v_fmac_f32_dpp v21, -v{dy}, v{wr} {back}", f"v_fmac_f32_dpp v22, -v{dz}, v{wr} {back}"]
        return out
    out = [
        f"v_sub_f32_dpp v{dx}, v12, v4 {ror}", f"v_sub_f32_dpp v{dy}, v13, v5 {ror}", f"v_sub_f32_dpp v{dz}, v14, v6 {ror}",
        f"v_fma_f32 v{r2}, v{dz}, v{dz}, v7", f"v_fma_f32 v{r2}, v{dy}, v{dy}, v{r2}", f"v_fma_f32 v{r2}, v{dx}, v{dx}, v{r2}",
        f"v_rsq_f32 v{r2}, v{r2}",
        f"v_mul_f32 v{w}, v{r2}, v{r2}", f"v_mul_f32 v{w}, v{w}, v{r2}",
        f"v_fma_f32 v8, -v{dx}, v{w}, v8", f"v_fma_f32 v9, -v{dy}, v{w}, v9", f"v_fma_f32 v10, -v{dz}, v{w}, v10",
    ]
    if react:
        if nop:
            out.append("s_nop 1")
        out += [f"v_mov_b32_dpp v{wr}, v{w} {back}",
                f"v_fmac_f32_dpp v20, v{dx}, v{wr} {back}", f"v_fmac_f32_dpp v21, v{dy}, v{wr} {back}", f"v_fmac_f32_dpp v22, v{dz}, v{wr} {back}"]
    return out

# as issued today: 15 steps, each led by s_nop 1 in front of its DPP tail (step 0 needs no rotation and is left out here)
_m = []
for s in range(1, 16):
    _m += mutual_step(s, (40, 41, 42, 43, 44, 45))
add("mutual_nop", _m, flop=15 * 40, pairs=30, n=15 * 16)

# two steps interleaved: the head of step s+1 (sub, r^2, rsq, r^-3, own sums) is issued between the producer of w(s) and the
# DPP tail of step s, so every DPP read is >= 2 instructions behind the write of its source and no s_nop is needed
def mutual_pipelined(steps, sets, vop2=False):
    heads, tails = [], []
    for i, s in enumerate(steps):
        t = sets[i % len(sets)]
        full = mutual_step(s, t, nop=False, vop2=vop2)
        heads.append(full[:12])
        tails.append(full[12:])
    out = list(heads[0])
    for i in range(len(steps)):
        if i + 1 < len(steps):
            # weave: tail(i) behind the first instructions of head(i+1)
            h = heads[i + 1]
            out += h[:3] + tails[i][:1] + h[3:6] + tails[i][1:2] + h[6:7] + tails[i][2:3] + h[7:9] + tails[i][3:4] + h[9:]
        else:
            out += ["s_nop 1"] + tails[i]
    return out

add("mutual_pipe2", mutual_pipelined(list(range(1, 16)), [(40, 41, 42, 43, 44, 45), (48, 49, 50, 51, 52, 53)]), flop=15 * 40, pairs=30, n=15 * 16)
add("mutual_pipe2_vop2", mutual_pipelined(list(range(1, 16)), [(40, 41, 42, 43, 44, 45), (48, 49, 50, 51, 52, 53)], vop2=True), flop=15 * 40, pairs=30, n=15 * 16)
# temporaries of the two interleaved steps in different banks from the fixed operands (v4..v7, v8..v10, v12..v14, v20..v22)
add("mutual_pipe2_banks", mutual_pipelined(list(range(1, 16)), [(41, 46, 51, 56, 61, 66), (71, 76, 81, 86, 91, 96)], vop2=True), flop=15 * 40, pairs=30, n=15 * 16)
# the same without the reaction half (13 -> 12 instructions: what a one-directional DPP step costs)
_o = []
for s in range(1, 16):
    _o += mutual_step(s, (40 + 8 * (s % 2), 41 + 8 * (s % 2), 42 + 8 * (s % 2), 43 + 8 * (s % 2), 44 + 8 * (s % 2), 45), react=False)
add("oneway_dpp", _o, flop=15 * 20, pairs=15)

# ---- the pair loop as csrc/gen_p2p.py emits it (differences in even registers, r^2 / r^-1 / r^-3 / eps^2 in odd ones: no
#      instruction reads three registers of one parity), without its LDS reads: what does the bare stream cost at 4, 6, 8 waves? ----
def good_regs(k):          # per source in flight: dx dy dz (even), ri w (odd)
    b = 24 + 6 * k
    return b, b + 2, b + 4, b + 1, b + 3

def good_pair(a, b, src=lambda k: (12 + 3 * (k % 4), 13 + 3 * (k % 4), 14 + 3 * (k % 4)), eps=29, acc=(8, 9, 10)):
    xa, ya, za, ra, wa = good_regs(a)
    xb, yb, zb, rb, wb = good_regs(b)
    sa, sb = src(a), src(b)
    return [
        f"v_sub_f32 v{xa}, v4, v{sa[0]}", f"v_sub_f32 v{ya}, v5, v{sa[1]}", f"v_sub_f32 v{za}, v6, v{sa[2]}",
        f"v_sub_f32 v{xb}, v4, v{sb[0]}", f"v_sub_f32 v{yb}, v5, v{sb[1]}", f"v_sub_f32 v{zb}, v6, v{sb[2]}",
        f"v_fma_f32 v{ra}, v{za}, v{za}, v{eps}", f"v_fma_f32 v{rb}, v{zb}, v{zb}, v{eps}",
        f"v_fmac_f32 v{ra}, v{ya}, v{ya}", f"v_fmac_f32 v{rb}, v{yb}, v{yb}",
        f"v_fmac_f32 v{ra}, v{xa}, v{xa}", f"v_fmac_f32 v{rb}, v{xb}, v{xb}",
        f"v_rsq_f32 v{ra}, v{ra}", f"v_rsq_f32 v{rb}, v{rb}",
        f"v_mul_f32 v{wa}, v{ra}, v{ra}", f"v_mul_f32 v{wb}, v{rb}, v{rb}",
        f"v_mul_f32 v{wa}, v{wa}, v{ra}", f"v_mul_f32 v{wb}, v{wb}, v{rb}",
        f"v_fmac_f32 v{acc[0]}, v{xa}, v{wa}", f"v_fmac_f32 v{acc[1]}, v{ya}, v{wa}", f"v_fmac_f32 v{acc[2]}, v{za}, v{wa}",
        f"v_fmac_f32 v{acc[0]}, v{xb}, v{wb}", f"v_fmac_f32 v{acc[1]}, v{yb}, v{wb}", f"v_fmac_f32 v{acc[2]}, v{zb}, v{wb}",
    ]

add("tile_2way", good_pair(0, 1) + good_pair(2, 3), flop=80, pairs=4)
# four sources interleaved instruction by instruction, two accumulator sets
_g = [good_pair(0, 1, acc=(8, 9, 10)), good_pair(2, 3, acc=(16, 17, 18))]
add("tile_4way", [_g[i % 2][i // 2] for i in range(48)], flop=80, pairs=4)
# the same stream with eps^2 in an even register: the first v_fma of every source reads three even registers (1 in 12 slow)
add("tile_2way_eps_even", good_pair(0, 1, eps=28) + good_pair(2, 3, eps=28), flop=80, pairs=4)
# ... and with the group-uniform LDS reads of the real block (three ds_read_b128 per four sources, address in v3)
def _with_reads(body, off):
    return ["s_waitcnt lgkmcnt(0)"] + body[:6] + body[24:30] + [f"ds_read_b128 v[{12 + 4 * j}:{15 + 4 * j}], v3 offset:{off + 16 * j}" for j in range(3)] + body[6:24] + body[30:]
_t = []
for q in range(4):
    _t += _with_reads(good_pair(0, 1) + good_pair(2, 3), 48 * q)
add("tile_2way_lds", _t, flop=320, pairs=16, n=192)

# ---- what does a v_rsq_f32 cost inside a stream of other vector instructions?  N independent v_fma per v_rsq, and the same
#      number of v_rsq issued back to back behind the v_fma (pairs = v_rsq count, so that the rate column counts them) ----------
for nf in (2, 4, 8, 12, 24, 48):
    groups = max(1, 48 // nf)
    lines = []
    for g in range(groups):
        lines += [f"v_fma_f32 v{16 + 2 * ((g * nf + k) % 20)}, v{16 + 2 * ((g * nf + k) % 20)}, v1, v3" for k in range(nf)]   # even acc, odd operands
        lines.append(f"v_rsq_f32 v{5 + 2 * (g % 4)}, v{7 + 2 * (g % 4)}")
    add(f"mix_fma{nf}_rsq1", lines, flop=2 * nf * groups, pairs=groups)
add("mix_fma48_rsq4_batched", [f"v_fma_f32 v{16 + 2 * (k % 20)}, v{16 + 2 * (k % 20)}, v1, v3" for k in range(48)] + [f"v_rsq_f32 v{5 + 2 * g}, v{7 + 2 * g}" for g in range(4)],
    flop=96, pairs=4)
add("mix_fma48_only", [f"v_fma_f32 v{16 + 2 * (k % 20)}, v{16 + 2 * (k % 20)}, v1, v3" for k in range(48)], flop=96)
add("mix_rsq16_only", [f"v_rsq_f32 v{16 + 2 * (k % 20)}, v{7 + 2 * (k % 4)}" for k in range(16)], pairs=16)
# the pair loop with its v_rsq replaced by a move: what the other eleven instructions cost where they stand
add("tile_2way_norsq", [l.replace("v_rsq_f32", "v_mov_b32") for l in good_pair(0, 1) + good_pair(2, 3)], flop=80, pairs=4)
# other transcendental / quarter-rate candidates in the same mix
add("mix_fma12_rcp1", [l.replace("v_rsq_f32", "v_rcp_f32") for l in (lambda: [x for g in range(4) for x in ([f"v_fma_f32 v{16 + 2 * ((g * 12 + k) % 20)}, v{16 + 2 * ((g * 12 + k) % 20)}, v1, v3" for k in range(12)] + [f"v_rsq_f32 v{5 + 2 * g}, v{7 + 2 * g}"])])()], flop=96, pairs=4)
add("mix_fma12_sqrt1", [l.replace("v_rsq_f32", "v_sqrt_f32") for l in (lambda: [x for g in range(4) for x in ([f"v_fma_f32 v{16 + 2 * ((g * 12 + k) % 20)}, v{16 + 2 * ((g * 12 + k) % 20)}, v1, v3" for k in range(12)] + [f"v_rsq_f32 v{5 + 2 * g}, v{7 + 2 * g}"])])()], flop=96, pairs=4)
add("mix_fma12_rsqf16", [l.replace("v_rsq_f32", "v_rsq_f16") for l in (lambda: [x for g in range(4) for x in ([f"v_fma_f32 v{16 + 2 * ((g * 12 + k) % 20)}, v{16 + 2 * ((g * 12 + k) % 20)}, v1, v3" for k in range(12)] + [f"v_rsq_f32 v{5 + 2 * g}, v{7 + 2 * g}"])])()], flop=96, pairs=4)

HEADER = r"""// generated by tools/valu_probe.py -- do not edit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Stamp { unsigned long long cyc, real; };
struct Variant { const char *name; void (*fn)(const float *, float *, Stamp *, int); int ninstr; int flop; int pairs; int max_waves; };
"""

import re


def kernel_text(name, lines):
    used = sorted({int(m) for l in lines for m in re.findall(r"\bv(\d+)\b", l)} | {8})
    clob = ", ".join(f'"v{r}"' for r in used)
    waves = 8 if max(used) < 56 else (6 if max(used) < 72 else 4)
    j = "\\n\\t\"\n\t\t\""
    body = j.join(lines)
    lds = any(l.startswith("ds_") for l in lines)
    init = j.join([f"v_mul_f32 v{r}, {1.0 + 0.03125 * (r % 29):.6f}, %0" for r in used if not (lds and r == 3)] + (["v_mov_b32 v3, %1"] if lds else []))
    fin = j.join(["v_mov_b32 %0, v8"] + [f"v_add_f32 %0, %0, v{r}" for r in used if r != 8])
    return f"""
__global__ __launch_bounds__(256, {waves}) void k_{name}(const float *in, float *out, Stamp *st, int iters)
{{
	float x = in[threadIdx.x & 63], r;
	__shared__ float lds_tile[4][2][96];
	unsigned la = 0;
	if ({1 if lds else 0})
	{{
		for (int q = threadIdx.x; q < 4 * 2 * 96; q += 256) (&lds_tile[0][0][0])[q] = in[q & 1023] * 1.5f;
		__syncthreads();
		la = (unsigned)(size_t)(__attribute__((address_space(3))) void *)&lds_tile[threadIdx.x >> 6][(threadIdx.x >> 5) & 1][0];
	}}
	asm volatile("{init}" :: "v"(x), "v"(la) : {clob});
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; ++it)
		asm volatile("{body}" ::: {clob});
	const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
	asm volatile("{fin}" : "=v"(r) :: {clob});
	out[blockIdx.x * 256 + threadIdx.x] = r;
	if ((threadIdx.x & 63) == 0) {{ st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0; st[blockIdx.x * 4 + (threadIdx.x >> 6)].real = q1 - q0; }}
}}
""", waves


MAIN = r"""
int main(int argc, char **argv)
{
	CHK(hipSetDevice(0));
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount, max_blocks = cus * 8;
	float *in, *out; Stamp *st;
	CHK(hipMalloc(&in, 4096)); CHK(hipMalloc(&out, sizeof(float) * 256 * (size_t)max_blocks)); CHK(hipMalloc(&st, sizeof(Stamp) * 4 * (size_t)max_blocks));
	std::vector<float> h(1024); srand(7);
	for (auto &v : h) v = 0.5f + (float)(rand() & 0xFFFF) / 262144.f;
	CHK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	std::vector<Stamp> hs(4 * (size_t)max_blocks);
	FILE *f = argc > 1 ? fopen(argv[1], "w") : nullptr;
	if (f) fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"note\": \"cycles = s_memtime ticks (shader clock) per instruction of ONE SIMD's stream: ticks of a wave / (iters * instructions * waves per SIMD); clock = d(s_memtime)/d(s_memrealtime) * 100 MHz, median over waves; ms = HIP events\",\n \"results\": [\n", prop.gcnArchName, cus);
	bool first = true;
	const double target_ms = argc > 2 ? atof(argv[2]) : 20.0;
	for (const Variant &v : variants)
		for (int w : {1, 2, 4, 5, 6, 8})
		{
			if (w > v.max_waves) continue;
			const int grid = cus * w;
			int iters = 64;
			hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, in, out, st, iters);
			CHK(hipDeviceSynchronize());
			// size the timed launch to ~target_ms from a short calibration launch
			CHK(hipEventRecord(e0)); hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, in, out, st, 2000); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float cal = 0; CHK(hipEventElapsedTime(&cal, e0, e1));
			iters = std::max(2000, (int)(2000 * target_ms / cal));
			CHK(hipEventRecord(e0)); hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, in, out, st, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
			CHK(hipMemcpy(hs.data(), st, sizeof(Stamp) * 4 * (size_t)grid, hipMemcpyDeviceToHost));
			std::vector<double> cyc, clk;
			for (size_t i = 0; i < 4 * (size_t)grid; ++i) { cyc.push_back((double)hs[i].cyc); clk.push_back((double)hs[i].cyc / (double)hs[i].real * 100.0); }
			std::nth_element(cyc.begin(), cyc.begin() + cyc.size() / 2, cyc.end());
			std::nth_element(clk.begin(), clk.begin() + clk.size() / 2, clk.end());
			const double c = cyc[cyc.size() / 2], mhz = clk[clk.size() / 2];
			const double cpi = c / ((double)iters * v.ninstr * w);
			const double tflops = (double)grid * 256 * v.flop * iters / (ms * 1e-3) / 1e12;
			const double pairs = (double)grid * 256 * v.pairs * iters / (ms * 1e-3);
			printf("%-18s w=%d  %7.3f ms  %6.3f cyc/instr  clock %6.0f MHz  %7.2f TFLOP/s (%.3f of 157.3)  %.3e pairs/s\n", v.name, w, ms, cpi, mhz, tflops, tflops / 157.3, pairs);
			if (f) { fprintf(f, "%s  {\"variant\": \"%s\", \"waves_per_simd\": %d, \"instr_per_block\": %d, \"ms\": %.3f, \"cycles_per_instr\": %.4f, \"clock_mhz\": %.0f, \"tflops\": %.2f, \"frac_of_157.3\": %.4f, \"pairs_per_s\": %.4e}", first ? "" : ",\n", v.name, w, v.ninstr, ms, cpi, mhz, tflops, tflops / 157.3, pairs); first = false; }
		}
	if (f) { fprintf(f, "\n]}\n"); fclose(f); }
	return 0;
}
"""


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "build/valu_probe.hip"
    with open(out, "w") as f:
        f.write(HEADER)
        only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
        table = []
        for name, n, flop, pairs, lines in KERNELS:
            if only and not any(name.startswith(o) for o in only):
                continue
            text, waves = kernel_text(name, lines)
            f.write(text)
            table.append((name, n, flop, pairs, waves))
        f.write("static const Variant variants[] = {\n")
        for name, n, flop, pairs, waves in table:
            f.write(f'\t{{"{name}", k_{name}, {n}, {flop}, {pairs}, {waves}}},\n')
        f.write("};\n")
        f.write(MAIN)


if __name__ == "__main__":
    main()
