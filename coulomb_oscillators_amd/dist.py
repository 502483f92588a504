"""Multi-GPU orchestration of the kd-tree FMM: kd-domain sharding with one all-gather per force evaluation.

SURVEY 8(e): the balanced kd-tree's level-log2(G) nodes hold exactly N/G particles each
(fmm_cart3_kdtree.cuh:109-137), so GPU g owns the subtree of node 2^d - 1 + g.  The C-ABI library does the
three compute stages (``nbco_dist_partition`` / ``nbco_dist_local`` / ``nbco_dist_finish``, include/nbco.h)
and never communicates; this module moves the two exchange buffers with ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

One process per GPU::

    run = DomainRun(Engine(fmm_order=6, ...), n_global, TorchComm())
    run.partition(pos_mine, vel_mine)          # rebalance: every `rebalance` steps
    run.leapfrog(param, dt)                    # or run.force(param)

The exchange per evaluation is an all-gather of the tree-ordered positions and of the node block (csz + multipoles
of the domain's subtree), the latter in two stages: the traversal records leave with the positions, the multipoles
follow under the traversal.  Forces need no reduction because cross-domain pairs are evaluated one-directionally on
the owner of the target.
"""
import torch


class _Done:
    def wait(self):
        return True


class TorchComm:
    """all-gather over a torch.distributed process group (one rank per GPU)."""

    def __init__(self, group=None, always_collective=False):
        """always_collective: run the collectives even in a world of one (tests: the RCCL code paths, dtypes and reduce ops on the
        one card the GPU box has)"""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.single = self.world == 1 and not always_collective

    def all_gather(self, out, inp):
        if self.single:
            out.copy_(inp)
        elif self.dist.get_backend(self.group) == "gloo" and inp.is_cuda:
            # rehearsal mode (several ranks on one card / CPU-only transport): stage through host memory
            host = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, inp.cpu(), group=self.group)
            out.copy_(host)
        else:
            self.dist.all_gather_into_tensor(out, inp, group=self.group)


    def all_gather_start(self, out, inp):
        """all_gather that returns at once; .wait() on the result orders the CURRENT stream behind the collective"""
        if self.single or (self.dist.get_backend(self.group) == "gloo" and inp.is_cuda):
            self.all_gather(out, inp)
            return _Done()
        return self.dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

    def all_reduce_i32(self, t, op):
        """in-place MIN / SUM of an int32 tensor"""
        if self.single:
            return
        rop = self.dist.ReduceOp.MIN if op == "min" else self.dist.ReduceOp.SUM
        if self.dist.get_backend(self.group) == "gloo" and t.is_cuda:
            host = t.cpu()
            self.dist.all_reduce(host, op=rop, group=self.group)
            t.copy_(host)
        else:
            self.dist.all_reduce(t, op=rop, group=self.group)

    def all_to_all(self, out, inp, out_rows, in_rows):
        """variable all-to-all along dim 0: in_rows[r] rows of `inp` go to rank r, out_rows[s] rows of `out` come from rank s"""
        if self.single:
            out.copy_(inp)
        elif self.dist.get_backend(self.group) == "gloo" and inp.is_cuda:
            host = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_to_all_single(host, inp.cpu(), list(out_rows), list(in_rows), group=self.group)
            out.copy_(host)
        else:
            self.dist.all_to_all_single(out, inp, list(out_rows), list(in_rows), group=self.group)

    def all_reduce(self, t, op="sum"):
        """in-place reduction of a small tensor of scalars (energies, bounds)"""
        if self.single:
            return t
        ops = {"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}
        if self.dist.get_backend(self.group) == "gloo" and t.is_cuda:
            host = t.cpu()
            self.dist.all_reduce(host, op=ops[op], group=self.group)
            t.copy_(host)
        else:
            self.dist.all_reduce(t, op=ops[op], group=self.group)
        return t


class SingleComm:
    """world of one: the exchange degenerates to a copy (used by bench.py --gpus 1 style checks)."""
    world, rank = 1, 0

    def all_gather(self, out, inp):
        out.copy_(inp)

    def all_gather_start(self, out, inp):
        out.copy_(inp)
        return _Done()

    def all_to_all(self, out, inp, out_rows, in_rows):
        out.copy_(inp)

    def all_reduce_i32(self, t, op):
        return

    def all_reduce(self, t, op="sum"):
        return t


_ERR_UNSUPPORTED = 4   # NBCO_ERR_UNSUPPORTED (include/nbco.h)


def run_dist_step(comm, work, st):
    """the collective a nbco_dist_step describes, on the uint8 workspace tensor `work` (include/nbco.h)"""
    G = comm.world
    if st.op in (1, 2):
        comm.all_reduce_i32(work[st.send_off: st.send_off + 4 * st.count].view(torch.int32), "min" if st.op == 1 else "sum")
    elif st.op == 3:
        comm.all_gather(work[st.recv_off: st.recv_off + G * st.count], work[st.send_off: st.send_off + st.count])
    elif st.op == 4:
        w = st.row_bytes // 4
        rs, rr = [int(st.rows_send[r]) for r in range(G)], [int(st.rows_recv[r]) for r in range(G)]
        inp = work[st.send_off: st.send_off + st.row_bytes * sum(rs)].view(torch.float32).view(-1, w)
        out = work[st.recv_off: st.recv_off + st.row_bytes * sum(rr)].view(torch.float32).view(-1, w)
        comm.all_to_all(out, inp, rr, rs)
    else:
        raise ValueError("unknown nbco_dist_step op %d" % st.op)


class DomainRun:
    """State and per-step protocol of ONE rank.

    `engine` needs dist_layout / dist_partition / dist_local / dist_finish / step / add_elastic (the
    ctypes Engine, or a test double with the same methods for the CPU tests).
    """

    def __init__(self, engine, n_global, comm, device=None, rebalance=8, let=None, gather_partition=None):
        self.eng = engine
        self.comm = comm
        self.world, self.rank = comm.world, comm.rank
        self.n_global = int(n_global)
        self.lay = engine.dist_layout(self.n_global, self.world, self.rank)
        self.n_local = int(self.lay.n_local)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.rebalance = int(rebalance)
        f32, u8 = torch.float32, torch.uint8
        nl, G = self.n_local, self.world
        self.buf = torch.zeros(9 * nl, dtype=f32, device=self.device)          # [pos | vel | acc], float3 AoS each
        # re-partition: distributed (no rank gathers the state) when engine and transport offer it; gather_partition=True keeps
        # the all-gather + redundant selection (nbco_dist_partition)
        can_dpart = all(hasattr(engine, m) for m in ("dist_repartition_begin", "dist_repartition_next")) and hasattr(comm, "all_reduce_i32") \
            and hasattr(comm, "all_to_all") and self.world <= 32
        self.dpart = can_dpart and not gather_partition
        self.state_all = None if self.dpart else torch.empty(6 * self.n_global, dtype=f32, device=self.device)
        self.work = torch.empty(engine.dist_repartition_workspace(self.n_global, self.world), dtype=u8, device=self.device) if self.dpart else None
        self.partition_bytes = None
        self.partition_fallbacks = 0   # cuts that fell back from the distributed re-partition to the gathered one (pivot ties)
        self.nodes_send = torch.empty(int(self.lay.nodes_bytes), dtype=u8, device=self.device)
        self.pos_send = torch.empty(int(self.lay.pos_bytes), dtype=u8, device=self.device)
        self.nodes_all = torch.empty(G * int(self.lay.nodes_bytes), dtype=u8, device=self.device)
        self.pos_all = torch.empty(G * int(self.lay.pos_bytes), dtype=u8, device=self.device)
        # the node block in two parts (views of the same buffers): traversal records / multipoles
        self.csz_bytes = int(getattr(self.lay, "csz_bytes", 0))
        self.split = self.csz_bytes > 0 and all(hasattr(engine, m) for m in ("dist_local_geom", "dist_local_mpole", "dist_finish_traverse",
                                                                             "dist_finish_rest")) and hasattr(comm, "all_gather_start")
        # locally-essential-tree exchange (nbco_dist_let_*): on by default when engine and transport offer it
        can_let = self.split and all(hasattr(engine, m) for m in ("dist_let_select", "dist_let_pack", "dist_let_finish")) and hasattr(comm, "all_to_all") \
            and int(getattr(self.lay, "let_counts", 0)) > 0
        self.let = can_let if let is None else (bool(let) and can_let)
        if self.let:
            S = int(self.lay.let_counts)
            self.counts_send = torch.zeros(S, dtype=torch.int64, device=self.device)
            self.counts_all = torch.zeros(G * S, dtype=torch.int64, device=self.device)
            self.rec = int(self.lay.let_node_bytes) // 4
            self._scratch = {}
        # capped form of the LET exchange (nbco_dist_let_pack_capped): no host round trip in the middle of the evaluation; the
        # segments are sized from the count matrix of the evaluation before, so the first evaluation (and one whose counts
        # outgrew them) runs the exact form
        self.capped = self.let and all(hasattr(engine, m) for m in ("dist_let_pack_capped", "dist_let_finish_capped", "dist_let_settle"))
        self._prevM = None
        self._counts_host = None
        self.let_capped_evals = 0     # evaluations that stood in the capped form
        self.let_redos = 0            # capped attempts declared void and repeated in the exact form
        self.last_exchange_bytes = None
        self.evals = 0

    def use_gather_partition(self):
        """switch to nbco_dist_partition (all-gather of the state + redundant selection)"""
        self.dpart = False
        if self.state_all is None:
            self.state_all = torch.empty(6 * self.n_global, dtype=torch.float32, device=self.device)

    def _rows(self, name, rows, width):
        """(rows, width) float32 view of a grow-only scratch buffer"""
        need = max(int(rows), 1) * width
        t = self._scratch.get(name)
        if t is None or t.numel() < need:
            t = torch.empty(int(need * 1.25) + 1024, dtype=torch.float32, device=self.device)
            self._scratch[name] = t
        return t[: int(rows) * width].view(int(rows), width)

    # views of the local state
    @property
    def pos(self):
        return self.buf[: 3 * self.n_local]

    @property
    def vel(self):
        return self.buf[3 * self.n_local: 6 * self.n_local]

    @property
    def acc(self):
        return self.buf[6 * self.n_local:]

    def exchange_bytes(self):
        """bytes this rank receives per force evaluation: of the last one with the LET exchange, else the all-gather's"""
        if self.let and self.last_exchange_bytes is not None:
            return self.last_exchange_bytes
        return self.allgather_bytes()

    def allgather_bytes(self):
        """bytes received per evaluation when whole node and position blocks are all-gathered"""
        return (self.world - 1) * (int(self.lay.nodes_bytes) + int(self.lay.pos_bytes))

    # ---- rebalance: gather the full state, redo the top log2(G) median splits, keep the own domain ----
    def partition(self, pos_mine=None, vel_mine=None):
        nl, N = self.n_local, self.n_global
        pos_mine = self.pos if pos_mine is None else pos_mine
        vel_mine = self.vel if vel_mine is None else vel_mine
        if self.dpart:
            # no rank ever holds more than its own particles: histograms, ties and bounds through small collectives, then one
            # all-to-all of [pos | vel] by destination (nbco_dist_repartition_*)
            if pos_mine.data_ptr() != self.pos.data_ptr():
                self.pos.copy_(pos_mine.reshape(-1))
            if vel_mine.data_ptr() != self.vel.data_ptr():
                self.vel.copy_(vel_mine.reshape(-1))
            moved = 0
            try:
                st = self.eng.dist_repartition_begin(self.buf, N, self.world, self.rank, self.work)
                while st.op != 0:
                    run_dist_step(self.comm, self.work, st)
                    if st.op == 4:
                        moved += st.row_bytes * (sum(int(st.rows_recv[r]) for r in range(self.world)) - int(st.rows_recv[self.rank]))
                    elif st.op == 3:
                        moved += (self.world - 1) * st.count
                    else:
                        moved += 2 * 4 * st.count   # (an all-reduce moves about twice its payload per rank)
                    st = self.eng.dist_repartition_next()
                self.partition_bytes = moved
                self.evals = 0
                return
            except Exception as e:
                # More pivot ties on one rank than the distributed select resolves (lattice / planar / duplicated coordinates):
                # NBCO_ERR_UNSUPPORTED.  Every rank sees the same gathered tie counts, so the failure is collective, and it is
                # reported before the local state has been touched: all ranks switch to the gathered form, which takes any input.
                if getattr(e, "status", None) != _ERR_UNSUPPORTED:
                    raise
                self.use_gather_partition()
                self.partition_fallbacks += 1
                pos_mine, vel_mine = self.pos, self.vel
        self.partition_bytes = (self.world - 1) * 24 * nl
        self.comm.all_gather(self.state_all[: 3 * N], pos_mine.contiguous().view(-1))
        self.comm.all_gather(self.state_all[3 * N:], vel_mine.contiguous().view(-1))
        self.eng.dist_partition(self.state_all, N, self.world, self.rank, self.buf)
        self.evals = 0

    # ---- one force evaluation -------------------------------------------------------------------
    def local(self):
        self.eng.dist_local(self.buf, self.n_local, self.nodes_send, self.pos_send)

    def exchange(self):
        self.comm.all_gather(self.nodes_all, self.nodes_send)
        self.comm.all_gather(self.pos_all, self.pos_send)

    def finish(self, param=None, elastic=True):
        self.eng.dist_finish(self.nodes_all, self.pos_all, self.buf, self.acc, param)
        if elastic and param is not None:
            self.eng.add_elastic(self.pos, self.acc, self.n_local, param[3:])
        self.evals += 1

    def force(self, param=None, elastic=True):
        if self.rebalance > 0 and self.evals >= self.rebalance:
            self.partition()
        if self.let:
            self._force_let(param)
            if elastic and param is not None:
                self.eng.add_elastic(self.pos, self.acc, self.n_local, param[3:])
            self.evals += 1
            return
        if self.split:
            # Three all-gathers, each started as soon as its data exists: positions + traversal records after the subtree
            # build, multipoles after the upward pass.  The traversal (which needs no multipoles) is enqueued behind the
            # first two; the multipoles -- 224 of the 240 bytes per node at order 6 -- travel under it.
            cb, G = self.csz_bytes, self.world
            csz_send, mp_send = self.nodes_send[:cb], self.nodes_send[cb:]
            csz_all, mp_all = self.nodes_all[: G * cb], self.nodes_all[G * cb:]
            self.eng.dist_local_geom(self.buf, self.n_local, self.pos_send, csz_send)
            h_pos = self.comm.all_gather_start(self.pos_all, self.pos_send)
            h_csz = self.comm.all_gather_start(csz_all, csz_send)
            self.eng.dist_local_mpole(self.buf, self.n_local, mp_send)
            h_mp = self.comm.all_gather_start(mp_all, mp_send)
            h_pos.wait()
            h_csz.wait()
            self.eng.dist_finish_traverse(csz_all, self.pos_all)
            self._wait_far_field(h_mp)
            self.eng.dist_finish_rest(mp_all, self.buf, self.acc, param)
            if elastic and param is not None:
                self.eng.add_elastic(self.pos, self.acc, self.n_local, param[3:])
            self.evals += 1
            return
        if hasattr(self.eng, "dist_local_build") and hasattr(self.comm, "all_gather_start"):
            # the positions travel while the multipoles are still being computed
            self.eng.dist_local_build(self.buf, self.n_local, self.pos_send)
            h_pos = self.comm.all_gather_start(self.pos_all, self.pos_send)
            self.eng.dist_local_upward(self.buf, self.n_local, self.nodes_send)
            h_nodes = self.comm.all_gather_start(self.nodes_all, self.nodes_send)
            h_pos.wait()
            h_nodes.wait()
        else:
            self.local()
            self.exchange()
        self.finish(param, elastic)

    # ---- LET exchange: one small all-gather (traversal records), the traversal, then two all-to-alls of exactly the
    # multipoles and positions the other ranks' lists name (include/nbco.h, nbco_dist_let_*) --------------------------
    def _let_counts(self, csz_all, gather):
        """selection + gathered count matrix on the host, [sender][let_counts] (int64); repeats while a rank reports list overflow"""
        G, S = self.world, int(self.lay.let_counts)
        for _ in range(8):
            self.eng.dist_let_select(csz_all, self.counts_send)
            M = gather()
            if not bool(M.view(G, S)[:, 2 * G].any()):
                return M
        raise RuntimeError("LET exchange: the traversal lists kept overflowing (raise list_factor)")

    def _let_splits(self, M):
        G, S, me = self.world, int(self.lay.let_counts), self.rank
        M2 = M.view(G, S)
        return (M2[me, 0:2 * G:2].tolist(), M2[me, 1:2 * G:2].tolist(), M2[:, 2 * me].tolist(), M2[:, 2 * me + 1].tolist())

    @staticmethod
    def cap_table(prevM, G, S, lay=None):
        """segment sizes of the capped exchange from the count matrix of the evaluation before: (nodes, particles), each [sender][receiver]
        (int64, host).  A quarter of head room plus a constant (never more than a domain holds); every rank holds the same
        matrix, hence the same table."""
        M2 = prevM.view(G, S)[:, : 2 * G]
        nodes, parts = M2[:, 0::2].clone(), M2[:, 1::2].clone()
        nodes += nodes // 4 + 64
        parts += parts // 4 + 512
        if lay is not None:
            nodes.clamp_(max=int(lay.ntot_local))
            parts.clamp_(max=int(lay.n_local))
        idx = torch.arange(G)
        nodes[idx, idx] = 0
        parts[idx, idx] = 0
        return nodes, parts

    @staticmethod
    def caps_hold(M, capn, capp, G, S):
        """the verdict on a capped attempt: no flag in the gathered counts and every count within its segment"""
        M2 = M.view(G, S)
        if bool(M2[:, 2 * G:].any()):
            return False
        return bool((M2[:, 0:2 * G:2] <= capn).all()) and bool((M2[:, 1:2 * G:2] <= capp).all())

    def _counts_to_host(self):
        """start the copy of the gathered counts to the host; returns wait() -> the matrix (host int64)"""
        if self.device.type != "cuda":
            M = self.counts_all.clone()
            return lambda: M
        if self._counts_host is None:
            self._counts_host = torch.empty(self.counts_all.numel(), dtype=torch.int64).pin_memory()
            self._counts_ev = torch.cuda.Event()
        self._counts_host.copy_(self.counts_all, non_blocking=True)
        self._counts_ev.record()

        def wait():
            self._counts_ev.synchronize()
            return self._counts_host.clone()
        return wait

    def _force_let_capped(self, param):
        """one attempt in the capped form; False = void (the caller repeats the evaluation in the exact form)"""
        cb, G, S, me = self.csz_bytes, self.world, int(self.lay.let_counts), self.rank
        csz_send, csz_all = self.nodes_send[:cb], self.nodes_all[: G * cb]
        capn, capp = self._caps
        send_n, send_p, recv_n, recv_p = capn[me].tolist(), capp[me].tolist(), capn[:, me].tolist(), capp[:, me].tolist()
        caps_out = torch.stack([capn[me], capp[me]], 1).reshape(-1).contiguous()
        caps_in = torch.stack([capn[:, me], capp[:, me]], 1).reshape(-1).contiguous()
        self.eng.dist_let_local_geom(self.buf, self.n_local, csz_send)
        h_csz = self.comm.all_gather_start(csz_all, csz_send)
        self.eng.dist_let_local_mpole(self.buf, self.n_local)
        h_csz.wait()
        self.eng.dist_let_select(csz_all, self.counts_send)
        self.comm.all_gather(self.counts_all, self.counts_send)
        counts = self._counts_to_host()
        pos_send, mp_send = self._rows("ps", sum(send_p), 4), self._rows("ms", sum(send_n), self.rec)
        pos_recv, mp_recv = self._rows("pr", sum(recv_p), 4), self._rows("mr", sum(recv_n), self.rec)
        self.eng.dist_let_pack_capped(caps_out, pos_send, mp_send)
        self.comm.all_to_all(pos_recv, pos_send, recv_p, send_p)
        self.comm.all_to_all(mp_recv, mp_send, recv_n, send_n)
        self.eng.dist_let_finish_capped(caps_in, pos_recv, mp_recv, self.buf, self.acc, param)
        # everything is queued: only now look at the counts (they left the GPU long ago)
        M = counts()
        ok = self.caps_hold(M, capn, capp, G, S)
        self.eng.dist_let_settle(ok)
        if not ok:
            self.let_redos += 1
            return False
        self._set_prev(M)
        self.let_capped_evals += 1
        self.last_exchange_bytes = (G - 1) * (cb + 8 * S) + 16 * sum(recv_p) + 4 * self.rec * sum(recv_n)
        return True

    def _set_prev(self, M):
        self._prevM = M
        if self.capped:
            self._caps = self.cap_table(M, self.world, int(self.lay.let_counts), self.lay)

    def _force_let(self, param):
        cb, G = self.csz_bytes, self.world
        csz_send, csz_all = self.nodes_send[:cb], self.nodes_all[: G * cb]
        if self.capped and self._prevM is not None and self._force_let_capped(param):
            return

        def gather():
            self.comm.all_gather(self.counts_all, self.counts_send)
            return self.counts_all.cpu()   # the one host synchronisation of the evaluation
        for _ in range(6):
            self.eng.dist_let_local_geom(self.buf, self.n_local, csz_send)
            h_csz = self.comm.all_gather_start(csz_all, csz_send)
            self.eng.dist_let_local_mpole(self.buf, self.n_local)
            h_csz.wait()
            M = self._let_counts(csz_all, gather)
            if not bool(M.view(G, int(self.lay.let_counts))[:, 2 * G + 1].any()):
                break   # (else: some rank's build was flagged -- its flag came with the counts -- and everybody starts over)
        else:
            raise RuntimeError("LET exchange: a tree build kept being flagged")
        send_n, send_p, recv_n, recv_p = self._let_splits(M)
        pos_send, mp_send = self._rows("ps", sum(send_p), 4), self._rows("ms", sum(send_n), self.rec)
        pos_recv, mp_recv = self._rows("pr", sum(recv_p), 4), self._rows("mr", sum(recv_n), self.rec)
        self.eng.dist_let_pack(M, pos_send, mp_send)
        self.comm.all_to_all(pos_recv, pos_send, recv_p, send_p)
        self.comm.all_to_all(mp_recv, mp_send, recv_n, send_n)
        self.eng.dist_let_finish(M, pos_recv, mp_recv, self.buf, self.acc, param)
        self._set_prev(M)
        self.last_exchange_bytes = (G - 1) * (cb + 8 * int(self.lay.let_counts)) + 16 * sum(recv_p) + 4 * self.rec * sum(recv_n)

    def _wait_far_field(self, handle):
        """Order the consumers of the gathered multipoles behind `handle`.  Only the engine's second stream reads them
        (nbco_aux_stream), so that stream waits and the compute stream goes on with the near-field lists; engines without
        one (CPU test doubles) or a world of one wait on the current stream."""
        ext = getattr(self, "_aux_ext", None)
        if ext is None:
            ext = False
            if hasattr(self.eng, "aux_stream") and self.device.type == "cuda" and not isinstance(handle, _Done):
                ptr = self.eng.aux_stream()
                if ptr:
                    ext = torch.cuda.ExternalStream(ptr, device=self.device)
            self._aux_ext = ext
        if ext:
            with torch.cuda.stream(ext):
                handle.wait()
        else:
            handle.wait()

    # ---- reductions over all domains: a handful of scalars through an all-reduce -----------------
    def minmax(self):
        """component-wise bounds of all positions, (2, 3) tensor [min; max] (reductions.cuh:67-80)"""
        mm = self.eng.minmax(self.pos, self.n_local).clone()
        self.comm.all_reduce(mm[0], "min")
        self.comm.all_reduce(mm[1], "max")
        return mm

    def energy(self, param):
        """(kinetic, elastic, coulomb) energy of the whole system at the positions of the last force evaluation: every rank sums
        its own particles (the Coulomb part from the interaction lists of that evaluation, nbco_energy_fmm), one all-reduce of
        three scalars"""
        kin, ela, cou = self.eng.energy_fmm(self.buf, self.n_local, param)
        t = torch.tensor([kin, ela, cou], dtype=torch.float64, device=self.device)
        self.comm.all_reduce(t, "sum")
        return float(t[0]), float(t[1]), float(t[2])

    # ---- kick-drift-kick leapfrog on the local state (integrator.cuh:68-80) --------------------------
    def leapfrog(self, param, dt, elastic=True, first=False):
        nl = self.n_local
        if first:
            self.force(param, elastic)
        self.eng.step(self.vel, self.acc, 0.5 * dt, nl)
        self.eng.step(self.pos, self.vel, dt, nl)
        self.force(param, elastic)
        self.eng.step(self.vel, self.acc, 0.5 * dt, nl)


    def leapfrog_steps(self, param, dt, steps, elastic=True):
        """`steps` kick-drift-kick steps; between two force evaluations ONE pass over the domain's state (nbco_dist_turnaround)
        instead of add_elastic + three step kernels + the next build's prologue.  Same final state as `steps` calls of leapfrog()."""
        nl = self.n_local
        if steps <= 0:
            return
        if not hasattr(self.eng, "dist_turnaround") or param is None:
            for _ in range(steps):
                self.leapfrog(param, dt, elastic)
            return
        self.eng.step(self.vel, self.acc, 0.5 * dt, nl)
        self.eng.step(self.pos, self.vel, dt, nl)
        for s in range(steps):
            self.force(param, elastic=False)
            if s + 1 < steps:
                self.eng.dist_turnaround(self.buf, nl, param, dt, 1.0, elastic)
        if elastic:
            self.eng.add_elastic(self.pos, self.acc, nl, param[3:])
        self.eng.step(self.vel, self.acc, 0.5 * dt, nl)


class LoopbackWorld:
    """G domains driven in lockstep inside ONE process / on ONE GPU (one Engine context per domain).

    The collectives become concatenations, everything else is the production code path; this is how the
    1-GPU box checks the sharded evaluation against the single-GPU one.
    """

    class _Comm:
        def __init__(self, world, rank):
            self.world, self.rank = world, rank

        def all_gather(self, out, inp):   # filled in by LoopbackWorld
            raise RuntimeError("loopback domains exchange through LoopbackWorld")

        def all_gather_start(self, out, inp):
            raise RuntimeError("loopback domains exchange through LoopbackWorld")

        all_to_all = all_reduce_i32 = all_gather

    def __init__(self, engines, n_global, device=None, rebalance=0, gather_partition=None):
        G = len(engines)
        self.runs = [DomainRun(e, n_global, LoopbackWorld._Comm(G, r), device=device, rebalance=rebalance, gather_partition=gather_partition)
                     for r, e in enumerate(engines)]
        self.G = G
        self._prevM = None
        self.let_capped_evals = self.let_redos = 0

    def partition(self, pos_parts, vel_parts):
        if all(r.dpart for r in self.runs):
            try:
                return self._repartition(pos_parts, vel_parts)
            except Exception as e:   # pivot ties beyond the distributed select: the gathered form, as DomainRun.partition does
                if getattr(e, "status", None) != _ERR_UNSUPPORTED:
                    raise
                for r in self.runs:
                    r.use_gather_partition()
                    r.partition_fallbacks += 1
                pos_parts, vel_parts = [r.pos for r in self.runs], [r.vel for r in self.runs]
        state = torch.cat([torch.cat([p.reshape(-1) for p in pos_parts]), torch.cat([v.reshape(-1) for v in vel_parts])])
        for r in self.runs:
            r.state_all.copy_(state)
            r.partition_bytes = (r.world - 1) * 24 * r.n_local
            r.eng.dist_partition(r.state_all, r.n_global, r.world, r.rank, r.buf)
            r.evals = 0

    def _repartition(self, pos_parts, vel_parts):
        """nbco_dist_repartition_* in lockstep: every collective becomes arithmetic over the ranks' workspaces"""
        runs, G = self.runs, self.G
        for r, p, v in zip(runs, pos_parts, vel_parts):
            if p.data_ptr() != r.pos.data_ptr():
                r.pos.copy_(p.reshape(-1))
            if v.data_ptr() != r.vel.data_ptr():
                r.vel.copy_(v.reshape(-1))
        sts = [r.eng.dist_repartition_begin(r.buf, r.n_global, G, r.rank, r.work) for r in runs]
        moved = [0] * G
        while sts[0].op != 0:
            op = sts[0].op
            assert all(s_.op == op and s_.count == sts[0].count for s_ in sts)
            st = sts[0]
            if op in (1, 2):
                views = [r.work[st.send_off: st.send_off + 4 * st.count].view(torch.int32) for r in runs]
                stack = torch.stack(views)
                red = stack.min(0).values if op == 1 else stack.sum(0, dtype=torch.int32)
                for v in views:
                    v.copy_(red)
            elif op == 3:
                allb = torch.cat([r.work[st.send_off: st.send_off + st.count] for r in runs])
                for r in runs:
                    r.work[st.recv_off: st.recv_off + G * st.count].copy_(allb)
            elif op == 4:
                w = st.row_bytes
                for r, sr in zip(runs, sts):
                    parts = []
                    for s_, ss in zip(runs, sts):
                        off = ss.send_off + w * sum(int(ss.rows_send[q]) for q in range(r.rank))
                        assert int(ss.rows_send[r.rank]) == int(sr.rows_recv[s_.rank])
                        parts.append(s_.work[off: off + w * int(ss.rows_send[r.rank])])
                    got = torch.cat(parts)
                    r.work[sr.recv_off: sr.recv_off + got.numel()].copy_(got)
            for r, sr in zip(runs, sts):   # bytes received, as DomainRun.partition counts them
                if op == 4:
                    moved[r.rank] += sr.row_bytes * (sum(int(sr.rows_recv[q]) for q in range(G)) - int(sr.rows_recv[r.rank]))
                elif op == 3:
                    moved[r.rank] += (G - 1) * sr.count
                else:
                    moved[r.rank] += 2 * 4 * sr.count
            sts = [r.eng.dist_repartition_next() for r in runs]
        for r in runs:
            r.partition_bytes = moved[r.rank]
            r.evals = 0

    def _force_let_capped(self, param, elastic, squeeze):
        """one attempt in the capped form, in lockstep; squeeze(capn, capp) may shrink the table (tests of the void path)"""
        runs, G = self.runs, self.G
        cb, S = runs[0].csz_bytes, int(runs[0].lay.let_counts)
        capn, capp = DomainRun.cap_table(self._prevM, G, S, runs[0].lay)
        if squeeze is not None:
            squeeze(capn, capp)
        for r in runs:
            r.eng.dist_let_local_geom(r.buf, r.n_local, r.nodes_send[:cb])
        csz = torch.cat([r.nodes_send[:cb] for r in runs])
        for r in runs:
            r.eng.dist_let_local_mpole(r.buf, r.n_local)
            r.nodes_all[: G * cb].copy_(csz)
            r.eng.dist_let_select(r.nodes_all[: G * cb], r.counts_send)
        counts = torch.cat([r.counts_send for r in runs])   # (stays on the device until everything is queued)
        sends = []
        for r in runs:
            me = r.rank
            ps, ms = r._rows("ps", int(capp[me].sum()), 4), r._rows("ms", int(capn[me].sum()), r.rec)
            r.eng.dist_let_pack_capped(torch.stack([capn[me], capp[me]], 1).reshape(-1).contiguous(), ps, ms)
            sends.append((ps, ms))
        for r in runs:
            me = r.rank
            pp, mm = [], []
            for s_, (ps, ms) in enumerate(sends):
                op, on = int(capp[s_, :me].sum()), int(capn[s_, :me].sum())
                pp.append(ps[op: op + int(capp[s_, me])])
                mm.append(ms[on: on + int(capn[s_, me])])
            pos_recv, mp_recv = torch.cat(pp).contiguous(), torch.cat(mm).contiguous()
            r.eng.dist_let_finish_capped(torch.stack([capn[:, me], capp[:, me]], 1).reshape(-1).contiguous(), pos_recv, mp_recv, r.buf, r.acc, param)
        M = counts.cpu()
        ok = DomainRun.caps_hold(M, capn, capp, G, S)
        for r in runs:
            r.eng.dist_let_settle(ok)
        if not ok:
            self.let_redos += 1
            return False
        self._prevM = M
        self.let_capped_evals += 1
        for r in runs:
            r.last_exchange_bytes = (G - 1) * (cb + 8 * S) + 16 * int(capp[:, r.rank].sum()) + 4 * r.rec * int(capn[:, r.rank].sum())
            if elastic and param is not None:
                r.eng.add_elastic(r.pos, r.acc, r.n_local, param[3:])
            r.evals += 1
        return True

    def force_let(self, param=None, elastic=True, tamper=None, capped=False, squeeze=None):
        """the LET exchange in lockstep; tamper(rank, pos_recv, mp_recv) may damage what a rank received (guard tests);
        capped: the form without a host round trip in the middle (after a first evaluation in the exact form)"""
        runs, G = self.runs, self.G
        cb, S = runs[0].csz_bytes, int(runs[0].lay.let_counts)
        if capped and self._prevM is not None and self._force_let_capped(param, elastic, squeeze):
            return
        for _ in range(6):
            for r in runs:
                r.eng.dist_let_local_geom(r.buf, r.n_local, r.nodes_send[:cb])
            csz = torch.cat([r.nodes_send[:cb] for r in runs])
            for r in runs:
                r.eng.dist_let_local_mpole(r.buf, r.n_local)
                r.nodes_all[: G * cb].copy_(csz)
            for _ in range(8):
                for r in runs:
                    r.eng.dist_let_select(r.nodes_all[: G * cb], r.counts_send)
                M = torch.cat([r.counts_send for r in runs]).cpu()
                if not bool(M.view(G, S)[:, 2 * G].any()):
                    break
            else:
                raise RuntimeError("LET exchange: the traversal lists kept overflowing")
            if not bool(M.view(G, S)[:, 2 * G + 1].any()):
                break   # (else: some rank's build was flagged and everybody starts over)
        else:
            raise RuntimeError("LET exchange: a tree build kept being flagged")
        M2 = M.view(G, S)
        self._prevM = M
        sends = []
        for r in runs:
            send_n, send_p, _, _ = r._let_splits(M)
            ps, ms = r._rows("ps", sum(send_p), 4), r._rows("ms", sum(send_n), r.rec)
            r.eng.dist_let_pack(M, ps, ms)
            sends.append((ps, ms, send_p, send_n))
        for r in runs:
            me = r.rank
            pp, mm = [], []
            for s_, (ps, ms, send_p, send_n) in enumerate(sends):
                op, on = sum(send_p[:me]), sum(send_n[:me])
                pp.append(ps[op: op + send_p[me]])
                mm.append(ms[on: on + send_n[me]])
            pos_recv, mp_recv = torch.cat(pp).contiguous(), torch.cat(mm).contiguous()
            if tamper is not None:
                tamper(me, pos_recv, mp_recv)
            r.eng.dist_let_finish(M, pos_recv, mp_recv, r.buf, r.acc, param)
            r.last_exchange_bytes = (G - 1) * (cb + 8 * S) + 16 * int(M2[:, 2 * me + 1].sum()) + 4 * r.rec * int(M2[:, 2 * me].sum())
            if elastic and param is not None:
                r.eng.add_elastic(r.pos, r.acc, r.n_local, param[3:])
            r.evals += 1

    def force(self, param=None, elastic=True, split=None, let=False, capped=False, squeeze=None):
        """split=None: the two-stage exchange (records, then multipoles) when the engines offer it; False: one node block;
        let=True: the LET exchange (capped=True: its form without a host round trip, see force_let)"""
        if let:
            return self.force_let(param, elastic, capped=capped, squeeze=squeeze)
        runs = self.runs
        if split is None:
            split = all(hasattr(r.eng, "dist_finish_traverse") and r.csz_bytes > 0 for r in runs)
        if not split:
            for r in runs:
                r.local()
            nodes = torch.cat([r.nodes_send for r in runs])
            pos = torch.cat([r.pos_send for r in runs])
            for r in runs:
                r.nodes_all.copy_(nodes)
                r.pos_all.copy_(pos)
                r.finish(param, elastic)
            return
        cb, G = runs[0].csz_bytes, self.G
        for r in runs:
            r.eng.dist_local_geom(r.buf, r.n_local, r.pos_send, r.nodes_send[:cb])
        pos = torch.cat([r.pos_send for r in runs])
        csz = torch.cat([r.nodes_send[:cb] for r in runs])
        for r in runs:
            r.eng.dist_local_mpole(r.buf, r.n_local, r.nodes_send[cb:])
        mp = torch.cat([r.nodes_send[cb:] for r in runs])
        for r in runs:
            r.pos_all.copy_(pos)
            r.nodes_all[: G * cb].copy_(csz)
            r.eng.dist_finish_traverse(r.nodes_all[: G * cb], r.pos_all)
        for r in runs:
            r.nodes_all[G * cb:].copy_(mp)
            r.eng.dist_finish_rest(r.nodes_all[G * cb:], r.buf, r.acc, param)
            if elastic and param is not None:
                r.eng.add_elastic(r.pos, r.acc, r.n_local, param[3:])
            r.evals += 1


class SlabRun:
    """The uniform-octree evaluators (nbco_fmm_traceless / nbco_fmm_symmetric) on G GPUs: slabs of the sorted cell keys.

    Every rank holds the whole state [pos | vel | acc] and builds the whole tree; rank r evaluates the accelerations of its slab
    of the cell order (``nbco_fmm_oct_shard``), one all-gather of the slabs (padded to the largest) completes the array on every
    rank, and every rank integrates all particles.  The result equals the single-GPU evaluation bit for bit.
    """

    def __init__(self, engine, n, comm, device=None, symmetric=False):
        self.eng, self.comm = engine, comm
        self.world, self.rank = comm.world, comm.rank
        self.n = int(n)
        self.symmetric = bool(symmetric)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.buf = torch.zeros(9 * self.n, dtype=torch.float32, device=self.device)
        self._pad = None
        self.bounds = None

    @property
    def pos(self):
        return self.buf[: 3 * self.n]

    @property
    def vel(self):
        return self.buf[3 * self.n: 6 * self.n]

    @property
    def acc(self):
        return self.buf[6 * self.n:]

    def set_state(self, pos, vel):
        self.pos.copy_(pos.reshape(-1))
        self.vel.copy_(vel.reshape(-1))

    def exchange_bytes(self):
        """bytes this rank receives per evaluation (the padded all-gather of the acceleration slabs)"""
        if self.bounds is None:
            return 0
        mx = max(b - a for a, b in zip(self.bounds[:-1], self.bounds[1:]))
        return (self.world - 1) * 12 * mx

    def slab(self, param):
        """this rank's part of the evaluation; returns (send block, largest slab, boundaries)"""
        b = self.eng.fmm_oct_shard(self.buf, self.acc, self.n, param, self.world, self.rank, symmetric=self.symmetric)
        self.bounds = b
        mx = max(max(y - x for x, y in zip(b[:-1], b[1:])), 1)
        if self._pad is None or self._pad[0].numel() < 3 * mx:
            cap = int(3 * mx * 1.25) + 64
            self._pad = (torch.zeros(cap, dtype=torch.float32, device=self.device), torch.zeros(self.world * cap, dtype=torch.float32, device=self.device))
        send = self._pad[0][: 3 * mx]
        lo, hi = b[self.rank], b[self.rank + 1]
        send[: 3 * (hi - lo)].copy_(self.acc[3 * lo: 3 * hi])
        return send, mx, b

    def assemble(self, recv, mx, b):
        for r in range(self.world):
            if r != self.rank and b[r + 1] > b[r]:
                self.acc[3 * b[r]: 3 * b[r + 1]].copy_(recv[3 * mx * r: 3 * mx * r + 3 * (b[r + 1] - b[r])])

    def force(self, param=None, elastic=True):
        send, mx, b = self.slab(param)
        if self.world > 1:
            recv = self._pad[1][: self.world * 3 * mx]
            self.comm.all_gather(recv, send)
            self.assemble(recv, mx, b)
        if elastic and param is not None:
            self.eng.add_elastic(self.pos, self.acc, self.n, param[3:])

    def leapfrog(self, param, dt, elastic=True):
        n = self.n
        self.eng.step(self.vel, self.acc, 0.5 * dt, n)
        self.eng.step(self.pos, self.vel, dt, n)
        self.force(param, elastic)
        self.eng.step(self.vel, self.acc, 0.5 * dt, n)


class LoopbackSlabs:
    """G SlabRuns in lockstep on one card (tests): the all-gather becomes a concatenation"""

    class _Comm:
        def __init__(self, world, rank):
            self.world, self.rank = world, rank

    def __init__(self, engines, n, device=None, symmetric=False):
        G = len(engines)
        self.runs = [SlabRun(e, n, LoopbackSlabs._Comm(G, r), device=device, symmetric=symmetric) for r, e in enumerate(engines)]

    def set_state(self, pos, vel):
        for r in self.runs:
            r.set_state(pos, vel)

    def force(self, param=None, elastic=True):
        parts = [r.slab(param) for r in self.runs]
        mx, b = parts[0][1], parts[0][2]
        assert all(p[1] == mx and p[2] == b for p in parts), "the ranks disagree on the slab boundaries"
        recv = torch.cat([p[0] for p in parts])
        for r in self.runs:
            r.assemble(recv, mx, b)
            if elastic and param is not None:
                r.eng.add_elastic(r.pos, r.acc, r.n, param[3:])
