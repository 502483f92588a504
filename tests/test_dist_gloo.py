"""CPU tests of the multi-GPU orchestration (coulomb_oscillators_amd/dist.py): world_size = 2 over gloo.

The compute stages of the C-ABI library need a GPU, so here a small numpy test double speaks the same
three-stage protocol (partition / local / finish) with an O(n^2) sum; what is under test is the host logic
that the GPU box cannot rehearse with one card: buffer layout of the exchanges, rank order of the gathered
blocks, the rebalance cadence and the leapfrog sequencing across real processes.  The sharded numerics
themselves are checked bit for bit on the GPU (tests/test_gpu_dist.py)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch

from coulomb_oscillators_amd.dist import DomainRun, SingleComm, SlabRun, TorchComm


class _Layout:
    pass


class NumpyDomainEngine:
    """dist_* protocol of include/nbco.h in numpy (float64 arithmetic, float32 storage)."""
    HDR = 8  # floats in the node block: [rank, n_local, sum x, sum y, sum z, 0, 0, 0]

    def __init__(self, eps2=1e-4):
        self.eps2 = eps2
        self.calls = []

    def dist_layout(self, n_global, world, rank):
        assert n_global % world == 0 and world & (world - 1) == 0
        lay = _Layout()
        lay.world, lay.rank, lay.n_global, lay.n_local = world, rank, n_global, n_global // world
        lay.d = world.bit_length() - 1
        lay.nodes_bytes, lay.pos_bytes = 4 * self.HDR, 16 * lay.n_local
        self.lay = lay
        return lay

    def dist_partition(self, state_all, n_global, world, rank, state_local):
        self.calls.append("partition")
        N, nl = n_global, n_global // world
        st = state_all.numpy()
        pos, vel = st[:3 * N].reshape(N, 3), st[3 * N:].reshape(N, 3)
        order = np.arange(N)
        seg = [(0, N)]
        for _ in range(self.lay.d):           # balanced median splits along the longest box axis, stable
            nxt = []
            for (a, b) in seg:
                p = pos[order[a:b]]
                ax = int(np.argmax(p.max(0) - p.min(0)))
                order[a:b] = order[a:b][np.argsort(p[:, ax], kind="stable")]
                m = a + (b - a + 1) // 2
                nxt += [(a, m), (m, b)]
            seg = nxt
        mine = order[rank * nl:(rank + 1) * nl]
        out = state_local.numpy()
        out[:3 * nl] = pos[mine].reshape(-1)
        out[3 * nl:6 * nl] = vel[mine].reshape(-1)

    def dist_local(self, buf, n_local, nodes_send, pos_send):
        self.calls.append("local")
        p = buf.numpy()[:3 * n_local].reshape(n_local, 3)
        hdr = nodes_send.numpy().view(np.float32)
        hdr[:] = 0
        hdr[0], hdr[1] = self.lay.rank, n_local
        hdr[2:5] = p.sum(0)
        p4 = pos_send.numpy().view(np.float32).reshape(n_local, 4)
        p4[:, :3] = p
        p4[:, 3] = 0

    def dist_finish(self, nodes_all, pos_all, buf, a_local, param=None):
        self.calls.append("finish")
        G, nl = self.lay.world, self.lay.n_local
        hdr = nodes_all.numpy().view(np.float32).reshape(G, self.HDR)
        p4 = pos_all.numpy().view(np.float32).reshape(G, nl, 4)
        for r in range(G):                     # gathered blocks arrive in rank order and describe their positions
            assert hdr[r, 0] == r and hdr[r, 1] == nl
            np.testing.assert_allclose(hdr[r, 2:5], p4[r, :, :3].sum(0), rtol=1e-5, atol=1e-5)
        own = buf.numpy()[:3 * nl].reshape(nl, 3)
        np.testing.assert_array_equal(own, p4[self.lay.rank, :, :3])
        src = p4.reshape(G * nl, 4)[:, :3].astype(np.float64)
        dx = own.astype(np.float64)[:, None, :] - src[None, :, :]
        r2 = (dx * dx).sum(-1) + self.eps2
        acc = (dx / r2[..., None] ** 1.5).sum(1)
        scale = float(param[0]) if param is not None else 1.0
        a_local.numpy()[:] = (acc * scale).astype(np.float32).reshape(-1)

    def minmax(self, p, n):
        q = p.numpy()[:3 * n].reshape(n, 3)
        return torch.from_numpy(np.stack([q.min(0), q.max(0)]))

    def energy_fmm(self, buf, n, param):
        b = buf.numpy().astype(np.float64)
        x, v = b[:3 * n].reshape(n, 3), b[3 * n:6 * n].reshape(n, 3)
        k = param.numpy()[3:6].astype(np.float64)
        return [0.5 * (v * v).sum(), 0.5 * (k * x * x).sum(), 0.0]

    def step(self, b, a, ds, n):
        b.numpy()[:3 * n] += np.float32(ds) * a.numpy()[:3 * n]

    def add_elastic(self, p, a, n, k):
        kk = k.numpy()[:3]
        a.numpy().reshape(n, 3)[:] -= p.numpy().reshape(n, 3) * kk


class NumpySplitEngine(NumpyDomainEngine):
    """the same evaluation through the two-stage exchange of include/nbco.h (nbco_dist_local_geom / _local_mpole /
    _finish_traverse / _finish_rest): the node block travels as [rank, n_local, 0, 0] first and [sum x, sum y, sum z, 0]
    behind the traversal half"""

    def dist_layout(self, n_global, world, rank):
        lay = super().dist_layout(n_global, world, rank)
        lay.csz_bytes, lay.mpole_bytes = 16, 16
        assert lay.nodes_bytes == lay.csz_bytes + lay.mpole_bytes
        return lay

    def dist_local_geom(self, buf, n_local, pos_send, csz_send):
        self.calls.append("geom")
        p = buf.numpy()[:3 * n_local].reshape(n_local, 3)
        csz_send.numpy().view(np.float32)[:] = (self.lay.rank, n_local, 0, 0)
        p4 = pos_send.numpy().view(np.float32).reshape(n_local, 4)
        p4[:, :3] = p
        p4[:, 3] = 0

    def dist_local_mpole(self, buf, n_local, mpole_send):
        self.calls.append("mpole")
        p = buf.numpy()[:3 * n_local].reshape(n_local, 3)
        m = mpole_send.numpy().view(np.float32)
        m[:3] = p.sum(0)
        m[3] = 0

    def dist_finish_traverse(self, csz_all, pos_all):
        self.calls.append("traverse")
        G, nl = self.lay.world, self.lay.n_local
        hdr = csz_all.numpy().view(np.float32).reshape(G, 4)
        for r in range(G):
            assert hdr[r, 0] == r and hdr[r, 1] == nl
        self._pos_all = pos_all

    def dist_finish_rest(self, mpole_all, buf, a_local, param=None):
        self.calls.append("rest")
        G, nl = self.lay.world, self.lay.n_local
        nodes = torch.zeros(G * 4 * self.HDR, dtype=torch.uint8)
        hdr = nodes.numpy().view(np.float32).reshape(G, self.HDR)
        hdr[:, 0] = np.arange(G)
        hdr[:, 1] = nl
        hdr[:, 2:6] = mpole_all.numpy().view(np.float32).reshape(G, 4)
        super().dist_finish(nodes, self._pos_all, buf, a_local, param)
        self.calls.pop()   # (the "finish" recorded by the base class)


class NumpyLetEngine(NumpySplitEngine):
    """the LET protocol of include/nbco.h (nbco_dist_let_*): counts through an all-gather, then two all-to-alls with splits that
    differ per receiver.  The double has no tree to prune with, so it sends every particle exactly once -- the first k_r of them
    to receiver r as position records {x, y, z, global index}, the others as "node" records {global index, x, y, z} -- and the
    receiver checks that both streams together cover every foreign particle exactly once.  Rank 1 reports list overflow in the
    first round of its first evaluation, which must trigger a second selection round on every rank."""

    def dist_layout(self, n_global, world, rank):
        lay = super().dist_layout(n_global, world, rank)
        lay.let_node_bytes, lay.let_counts = 16, 2 * world + 2
        self.rounds = 0
        return lay

    def _k(self, r):
        return (self.lay.n_local * (r + 1)) // (self.lay.world + 1)

    def dist_let_local_geom(self, buf, n_local, csz_send):
        self.calls.append("geom")
        csz_send.numpy().view(np.float32)[:] = (self.lay.rank, n_local, 0, 0)
        self._own = buf.numpy()[:3 * n_local].reshape(n_local, 3).copy()

    def dist_let_local_mpole(self, buf, n_local):
        self.calls.append("mpole")

    def dist_let_select(self, csz_all, counts_send):
        self.calls.append("select")
        G, nl, me = self.lay.world, self.lay.n_local, self.lay.rank
        hdr = csz_all.numpy().view(np.float32).reshape(G, 4)
        for r in range(G):
            assert hdr[r, 0] == r and hdr[r, 1] == nl
        c = counts_send.numpy()
        c[:] = 0
        for r in range(G):
            if r != me:
                c[2 * r], c[2 * r + 1] = nl - self._k(r), self._k(r)
        self.rounds += 1
        c[2 * G] = 1 if (me == 1 and self.rounds == 1) else 0

    def dist_let_pack(self, M, pos_send, mp_send):
        self.calls.append("pack")
        G, nl, me = self.lay.world, self.lay.n_local, self.lay.rank
        M = M.numpy().reshape(G, 2 * G + 2)
        assert not M[:, 2 * G].any()
        idx = (me * nl + np.arange(nl)).astype(np.int32).view(np.float32)
        ps, ms = pos_send.numpy(), mp_send.numpy()
        op = on = 0
        for r in range(G):
            if r == me:
                continue
            k = self._k(r)
            assert (M[me, 2 * r], M[me, 2 * r + 1]) == (nl - k, k)
            ps[op:op + k, :3], ps[op:op + k, 3] = self._own[:k], idx[:k]
            ms[on:on + nl - k, 0], ms[on:on + nl - k, 1:] = idx[k:], self._own[k:]
            op, on = op + k, on + nl - k
        assert (op, on) == (ps.shape[0], ms.shape[0])

    def dist_let_finish(self, M, pos_recv, mp_recv, buf, a_local, param=None):
        self.calls.append("finish")
        G, nl, me = self.lay.world, self.lay.n_local, self.lay.rank
        M = M.numpy().reshape(G, 2 * G + 2)
        pr, mr = pos_recv.numpy(), mp_recv.numpy()
        assert pr.shape[0] == M[:, 2 * me + 1].sum() and mr.shape[0] == M[:, 2 * me].sum()
        # records from rank s occupy the s-th segment
        seg = np.concatenate([np.full(M[s_, 2 * me + 1], s_) for s_ in range(G)]) if pr.shape[0] else np.zeros(0, dtype=int)
        np.testing.assert_array_equal(pr[:, 3].copy().view(np.int32) // nl, seg)
        self._finish_records(pr, mr, buf, a_local, param)

    def _finish_records(self, pr, mr, buf, a_local, param):
        G, nl, me = self.lay.world, self.lay.n_local, self.lay.rank
        src = np.full((G * nl, 3), np.nan, dtype=np.float32)
        src[me * nl:(me + 1) * nl] = self._own
        seen = np.zeros(G * nl, dtype=int)
        seen[me * nl:(me + 1) * nl] = 1
        for ids, xyz in ((pr[:, 3].copy().view(np.int32), pr[:, :3]), (mr[:, 0].copy().view(np.int32), mr[:, 1:])):
            np.add.at(seen, ids, 1)
            src[ids] = xyz
        assert (seen == 1).all()
        own = buf.numpy()[:3 * nl].reshape(nl, 3)
        dx = own.astype(np.float64)[:, None, :] - src.astype(np.float64)[None, :, :]
        r2 = (dx * dx).sum(-1) + self.eps2
        acc = (dx / r2[..., None] ** 1.5).sum(1)
        scale = float(param[0]) if param is not None else 1.0
        a_local.numpy()[:] = (acc * scale).astype(np.float32).reshape(-1)


class NumpyCappedLetEngine(NumpyLetEngine):
    """+ the capped form (nbco_dist_let_pack_capped / _finish_capped / _settle): whole segments travel, free records carry index -1,
    the counts are only looked at when the evaluation is queued.  Rank 0 raises the build flag in its 5th selection -- a capped
    attempt -- which every rank must declare void and repeat in the exact form."""
    VOID_ROUND = 5

    def dist_layout(self, n_global, world, rank):
        lay = super().dist_layout(n_global, world, rank)
        lay.ntot_local = lay.n_local
        return lay

    def dist_let_select(self, csz_all, counts_send):
        super().dist_let_select(csz_all, counts_send)
        if self.lay.rank == 0 and self.rounds == self.VOID_ROUND:
            counts_send.numpy()[2 * self.lay.world + 1] = 1

    def dist_let_pack_capped(self, caps_out, pos_send, mp_send):
        self.calls.append("packc")
        G, nl, me = self.lay.world, self.lay.n_local, self.lay.rank
        caps = caps_out.numpy().reshape(G, 2)
        assert caps[me].tolist() == [0, 0]
        ps, ms = pos_send.numpy(), mp_send.numpy()
        assert (ps.shape[0], ms.shape[0]) == (caps[:, 1].sum(), caps[:, 0].sum())
        free = np.array([-1], dtype=np.int32).view(np.float32)[0]
        ps[:], ms[:] = 0, 0
        ps[:, 3], ms[:, 0] = free, free
        idx = (me * nl + np.arange(nl)).astype(np.int32).view(np.float32)
        op = on = 0
        for r in range(G):
            k = self._k(r)
            if r != me:
                assert k <= caps[r, 1] and nl - k <= caps[r, 0]   # (the double's counts never change: a segment always holds them)
                ps[op:op + k, :3], ps[op:op + k, 3] = self._own[:k], idx[:k]
                ms[on:on + nl - k, 0], ms[on:on + nl - k, 1:] = idx[k:], self._own[k:]
            op, on = op + caps[r, 1], on + caps[r, 0]

    def dist_let_finish_capped(self, caps_in, pos_recv, mp_recv, buf, a_local, param=None):
        self.calls.append("finishc")
        G, me = self.lay.world, self.lay.rank
        caps = caps_in.numpy().reshape(G, 2)
        pr, mr = pos_recv.numpy(), mp_recv.numpy()
        assert (pr.shape[0], mr.shape[0]) == (caps[:, 1].sum(), caps[:, 0].sum()) and caps[me].tolist() == [0, 0]
        pr = pr[pr[:, 3].copy().view(np.int32) >= 0]
        mr = mr[mr[:, 0].copy().view(np.int32) >= 0]
        self._finish_records(pr, mr, buf, a_local, param)

    def dist_let_settle(self, ok):
        self.calls.append("settle%d" % int(ok))


class _Step:
    """nbco_dist_step as the ctypes Engine returns it"""
    def __init__(self, op=0, send_off=0, recv_off=0, count=0, row_bytes=0, rows_send=(), rows_recv=()):
        self.op, self.send_off, self.recv_off, self.count, self.row_bytes = op, send_off, recv_off, count, row_bytes
        self.rows_send, self.rows_recv = list(rows_send), list(rows_recv)


class NumpyRepartEngine(NumpySplitEngine):
    """the re-partition protocol of include/nbco.h (nbco_dist_repartition_begin / _next: the engine names a collective on the
    caller's workspace, the caller runs it).  The double walks through all four kinds -- MIN and SUM all-reduce, all-gather,
    all-to-all with uneven rows -- and cuts the same domains as NumpyDomainEngine.dist_partition: global bounds by MIN (the
    maxima inverted), the particle count by SUM, every rank's [pos | vel] by all-gather, then each rank keeps what it owns and
    receives the rest through the all-to-all."""

    def dist_repartition_workspace(self, n_global, world):
        nl = n_global // world
        return 64 + 24 * nl + 24 * n_global + 2 * 24 * nl

    def dist_repartition_begin(self, state_local, n_global, world, rank, work):
        self.calls.append("partition")
        self._rp = dict(state=state_local, N=n_global, G=world, me=rank, work=work, stage=0)
        nl = n_global // world
        p = state_local.numpy()[:3 * nl].reshape(nl, 3)
        w = work.numpy()
        key = (p.min(0) * 1e6).astype(np.int32), (-p.max(0) * 1e6).astype(np.int32)
        w[:24].view(np.int32)[:] = np.concatenate(key)
        return _Step(op=1, send_off=0, count=6)

    def dist_repartition_next(self):
        rp = self._rp
        N, G, me, w = rp["N"], rp["G"], rp["me"], rp["work"].numpy()
        nl = N // G
        st = rp["state"].numpy()
        rp["stage"] += 1
        if rp["stage"] == 1:
            b = w[:24].view(np.int32)
            p = st[:3 * nl].reshape(nl, 3)
            assert (b[:3] <= (p.min(0) * 1e6).astype(np.int32)).all() and (b[3:] <= (-p.max(0) * 1e6).astype(np.int32)).all()
            w[32:36].view(np.int32)[:] = nl
            return _Step(op=2, send_off=32, count=1)
        if rp["stage"] == 2:
            assert w[32:36].view(np.int32)[0] == N
            w[64:64 + 24 * nl].view(np.float32)[:] = st[:6 * nl]
            return _Step(op=3, send_off=64, recv_off=64 + 24 * nl, count=24 * nl)
        if rp["stage"] == 3:
            allst = w[64 + 24 * nl: 64 + 24 * nl + 24 * N].view(np.float32).reshape(G, 2, nl, 3)
            full = torch.from_numpy(np.concatenate([allst[:, 0].reshape(-1), allst[:, 1].reshape(-1)]))
            # destination of every particle: the domain the gathered selection puts it in
            dest = np.empty(N, dtype=int)
            pos_all, vel_all = full.numpy()[:3 * N].reshape(N, 3), full.numpy()[3 * N:].reshape(N, 3)
            for g in range(G):
                out = torch.zeros(6 * nl)
                NumpyDomainEngine.dist_partition(self, full, N, G, g, out)
                self.calls.pop()
                mine = out.numpy()[:3 * nl].reshape(nl, 3)
                idx = {tuple(r): i for i, r in enumerate(map(tuple, pos_all))}
                dest[[idx[tuple(r)] for r in map(tuple, mine)]] = g
            mydest = dest[me * nl:(me + 1) * nl]
            order = np.argsort(mydest, kind="stable")
            send = np.concatenate([pos_all[me * nl:(me + 1) * nl][order], vel_all[me * nl:(me + 1) * nl][order]], axis=1).astype(np.float32)
            off_s = 64 + 24 * nl + 24 * N
            w[off_s: off_s + 24 * nl].view(np.float32)[:] = send.reshape(-1)
            rows_send = [int((mydest == g).sum()) for g in range(G)]
            rows_recv = [int((dest[s_ * nl:(s_ + 1) * nl] == me).sum()) for s_ in range(G)]
            assert sum(rows_recv) == nl
            rp["off_r"] = off_s + 24 * nl
            return _Step(op=4, send_off=off_s, recv_off=rp["off_r"], count=nl, row_bytes=24, rows_send=rows_send, rows_recv=rows_recv)
        rec = w[rp["off_r"]: rp["off_r"] + 24 * nl].view(np.float32).reshape(nl, 6)
        st[:3 * nl] = rec[:, :3].reshape(-1)
        st[3 * nl:6 * nl] = rec[:, 3:].reshape(-1)
        return _Step(op=0)


def _system(n, seed=11):
    rng = np.random.default_rng(seed)
    pos = rng.standard_normal((n, 3)).astype(np.float32)
    vel = (0.1 * rng.standard_normal((n, 3))).astype(np.float32)
    par = np.array([1.0 / n, 0, 0, 1.0, 1.5, 2.0], dtype=np.float32)
    return pos, vel, par


def _drive(run, par, steps, dt):
    run.force(par)
    for _ in range(steps):
        run.leapfrog(par, dt)
    return torch.cat([run.pos.view(-1, 3), run.vel.view(-1, 3), run.acc.view(-1, 3)], dim=1).numpy()


def _worker(rank, world, port, n, steps, dt, rebalance, outdir, split=False, let=False, repart=False, capped=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        pos, vel, par = _system(n)
        nl = n // world
        eng = NumpyRepartEngine() if repart else (NumpyCappedLetEngine() if capped else NumpyLetEngine() if let else (NumpySplitEngine() if split else NumpyDomainEngine()))
        run = DomainRun(eng, n, TorchComm(), device=torch.device("cpu"), rebalance=rebalance)
        assert (run.world, run.rank, run.n_local) == (world, rank, nl) and run.split == split and run.let == let and run.dpart == repart and run.capped == capped
        if let:
            assert run.exchange_bytes() == run.allgather_bytes()   # (nothing evaluated yet)
        run.partition(torch.from_numpy(pos[rank * nl:(rank + 1) * nl]).reshape(-1), torch.from_numpy(vel[rank * nl:(rank + 1) * nl]).reshape(-1))
        res = _drive(run, torch.from_numpy(par), steps, dt)
        np.save(os.path.join(outdir, "rank%d.npy" % rank), res)
        mm = run.minmax().numpy()
        kin, ela, _ = run.energy(torch.from_numpy(par))
        np.save(os.path.join(outdir, "scal%d.npy" % rank), np.concatenate([mm.ravel(), [kin, ela]]))
        with open(os.path.join(outdir, "calls%d.txt" % rank), "w") as f:
            f.write(" ".join(eng.calls))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,split,let,repart,capped", [(2, False, False, False, False), (4, False, False, False, False), (2, True, False, False, False),
                                                           (4, True, False, False, False), (2, True, True, False, False), (4, True, True, False, False),
                                                           (2, True, False, True, False), (4, True, False, True, False), (2, True, True, False, True),
                                                           (4, True, True, False, True)])
def test_domain_run_over_gloo_matches_single_process(world, split, let, repart, capped):
    import torch.multiprocessing as mp
    n, steps, dt, rebalance = 512, 5, 1e-2, 2
    pos, vel, par = _system(n)
    one = DomainRun(NumpyDomainEngine(), n, SingleComm(), device=torch.device("cpu"), rebalance=rebalance)
    one.partition(torch.from_numpy(pos).reshape(-1), torch.from_numpy(vel).reshape(-1))
    ref = _drive(one, torch.from_numpy(par), steps, dt)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), n, steps, dt, rebalance, d, split, let, repart, capped), nprocs=world, join=True)
        got = np.concatenate([np.load(os.path.join(d, "rank%d.npy" % r)) for r in range(world)])
        calls = [open(os.path.join(d, "calls%d.txt" % r)).read().split() for r in range(world)]
        scal = [np.load(os.path.join(d, "scal%d.npy" % r)) for r in range(world)]
    # reductions over the domains: every rank holds the same global bounds and energies
    for sc in scal:
        np.testing.assert_array_equal(sc, scal[0])
    np.testing.assert_allclose(scal[0][:3], ref[:, 0:3].min(0), rtol=1e-5)
    np.testing.assert_allclose(scal[0][3:6], ref[:, 0:3].max(0), rtol=1e-5)
    np.testing.assert_allclose(scal[0][6], 0.5 * (ref[:, 3:6].astype(np.float64) ** 2).sum(), rtol=1e-4)
    # same particles, same trajectories (the summation order differs: float64 inside, so ~1 ulp of float32)
    ka, kb = np.argsort(got[:, 0]), np.argsort(ref[:, 0])
    np.testing.assert_allclose(got[ka], ref[kb], rtol=2e-5, atol=2e-6)
    # protocol: partition first, then local/finish pairs, a re-partition after every `rebalance` evaluations
    want = ["partition"]
    ev = 0
    rounds = 0
    for _ in range(steps + 1):
        if ev >= rebalance:
            want.append("partition")
            ev = 0
        if capped and len(want) > 1:
            # every evaluation after the first is attempted in the capped form; the attempt in which rank 0 raised its flag is void
            # on EVERY rank and repeated in the exact form
            rounds += 1
            void = rounds == NumpyCappedLetEngine.VOID_ROUND
            want += ["geom", "mpole", "select", "packc", "finishc", "settle%d" % (0 if void else 1)]
            if void:
                rounds += 1
                want += ["geom", "mpole", "select", "pack", "finish"]
        elif let:
            rounds += 2
            # (every rank repeats the selection in the round in which rank 1 reported overflow: its first evaluation)
            want += ["geom", "mpole", "select"] + (["select"] if len(want) == 1 else []) + ["pack", "finish"]
        else:
            want += ["geom", "mpole", "traverse", "rest"] if split else ["local", "finish"]
        ev += 1
    assert all(c == want for c in calls), calls


def test_exchange_bytes_and_views():
    n, world = 256, 1
    run = DomainRun(NumpyDomainEngine(), n, SingleComm(), device=torch.device("cpu"))
    assert run.exchange_bytes() == 0
    assert run.pos.numel() == run.vel.numel() == run.acc.numel() == 3 * n
    assert run.pos.data_ptr() + 12 * n == run.vel.data_ptr()


# ---- SlabRun (uniform-octree evaluators: replicated state, partitioned targets) over gloo ---------------------------------------
class NumpySlabEngine:
    """fmm_oct_shard of the engine in numpy: sorts the state by x-layer keys (identically on every rank), evaluates the direct sum
    for the rank's slab of that order, leaves the other accelerations untouched"""

    def __init__(self, layers=16, eps2=1e-4):
        self.layers, self.eps2 = layers, eps2

    def fmm_oct_shard(self, buf, a, n, param, world, rank, symmetric=False):
        st = buf.numpy()
        pos, vel = st[:3 * n].reshape(n, 3), st[3 * n:6 * n].reshape(n, 3)
        lo, hi = pos[:, 0].min(), pos[:, 0].max()
        key = np.minimum(((pos[:, 0] - lo) / max(hi - lo, 1e-30) * self.layers).astype(np.int64), self.layers - 1)
        order = np.argsort(key, kind="stable")
        pos[:], vel[:], key = pos[order], vel[order], key[order]
        first = np.searchsorted(key, np.arange(self.layers + 1))            # first particle of every layer
        cells = [int(np.searchsorted(first, (n * r) // world)) for r in range(world + 1)]
        b = [int(first[min(c, self.layers)]) for c in cells]
        b[-1] = n
        p0, p1 = b[rank], b[rank + 1]
        dx = pos[p0:p1].astype(np.float64)[:, None, :] - pos.astype(np.float64)[None, :, :]
        r2 = (dx * dx).sum(-1) + self.eps2
        scale = float(param[0]) if param is not None else 1.0
        a.numpy().reshape(n, 3)[p0:p1] = ((dx / r2[..., None] ** 1.5).sum(1) * scale).astype(np.float32)
        return b

    def step(self, b, a, ds, n):
        b.numpy()[:3 * n] += np.float32(ds) * a.numpy()[:3 * n]

    def add_elastic(self, p, a, n, k):
        a.numpy().reshape(n, 3)[:] -= p.numpy().reshape(n, 3) * k.numpy()[:3]


def _slab_drive(run, pos, vel, par, steps, dt):
    run.set_state(torch.from_numpy(pos), torch.from_numpy(vel))
    run.acc.fill_(float("nan"))
    run.force(par)
    for _ in range(steps):
        run.leapfrog(par, dt)
    return run.buf.numpy().copy()


def _slab_worker(rank, world, port, n, steps, dt, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        pos, vel, par = _system(n, seed=5)
        run = SlabRun(NumpySlabEngine(), n, TorchComm(), device=torch.device("cpu"))
        res = _slab_drive(run, pos, vel, torch.from_numpy(par), steps, dt)
        np.save(os.path.join(outdir, "slab%d.npy" % rank), res)
        np.save(os.path.join(outdir, "bytes%d.npy" % rank), np.array([run.exchange_bytes()] + list(run.bounds)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_slab_run_over_gloo_matches_single_process(world):
    import torch.multiprocessing as mp
    n, steps, dt = 600, 3, 1e-2
    pos, vel, par = _system(n, seed=5)
    one = SlabRun(NumpySlabEngine(), n, SingleComm(), device=torch.device("cpu"))
    ref = _slab_drive(one, pos, vel, torch.from_numpy(par), steps, dt)
    assert np.isfinite(ref).all()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_slab_worker, args=(world, _free_port(), n, steps, dt, d), nprocs=world, join=True)
        got = [np.load(os.path.join(d, "slab%d.npy" % r)) for r in range(world)]
        meta = [np.load(os.path.join(d, "bytes%d.npy" % r)) for r in range(world)]
    for r in range(world):
        # every rank ends with the complete state, identical to the single-process run (same sums in the same order)
        np.testing.assert_array_equal(got[r], ref)
        b = meta[r][1:]
        assert b[0] == 0 and b[-1] == n and meta[r][0] == (world - 1) * 12 * np.diff(b).max()
