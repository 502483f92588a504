"""GPU tests of the multi-GPU kd-domain sharding on ONE card: G domains, one context each, driven in
lockstep by LoopbackWorld (the all-gathers become concatenations; everything else is the production path,
through the C ABI).  Bar: the sharded evaluation equals the single-GPU evaluation -- tree order of the particles and
velocities carried along BIT FOR BIT; accelerations bit for bit with the one-directional near-field kernel
(opts.p2p_mutual = 0) and to summation-order rounding (2e-6) with the mutual one, where a cross-domain leaf pair is
evaluated by the target's owner instead of arriving as the other leaf's reaction -- and so inherits its parity with the
oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MUTUAL = [0, 1]   # opts.p2p_mutual


def same_acc(got, ref, mutual, n):
    """accelerations of a sharded evaluation against the single-GPU ones"""
    import torch
    from nbutil import force_err
    if not mutual:
        return torch.equal(got, ref)
    return force_err(got.cpu().numpy().reshape(n, 3), ref.cpu().numpy().reshape(n, 3)) < 2e-6


def make_state(oracle, n, kind="reference"):
    if kind == "reference":
        buf = oracle.init_reference(n)
        return np.ascontiguousarray(buf[0]), np.ascontiguousarray(buf[1])
    rng = np.random.default_rng(1234 + n)
    if kind == "uniform":
        return rng.random((n, 3), dtype=np.float32), rng.standard_normal((n, 3)).astype(np.float32)
    # two clumps of different density: exercises uneven interaction lists across the domain boundaries
    a = rng.standard_normal((n // 2, 3)).astype(np.float32) * 0.1
    b = rng.standard_normal((n - n // 2, 3)).astype(np.float32) + np.float32(2.0)
    return np.concatenate([a, b]), rng.standard_normal((n, 3)).astype(np.float32)


def single_gpu(n, pos, vel, par, **opts):
    import torch
    from coulomb_oscillators_amd import Engine
    e = Engine(**opts)
    buf = torch.cat([torch.from_numpy(pos).reshape(-1), torch.from_numpy(vel).reshape(-1), torch.zeros(3 * n)]).cuda()
    e.fmm_cart3_kdtree(buf, buf[6 * n:], n, par)
    torch.cuda.synchronize()
    return e, buf


@pytest.mark.parametrize("n,G,p", [(32768, 2, 6), (40000, 8, 5)])
def test_single_block_exchange_gives_the_same(oracle32, n, G, p):
    """nbco_dist_local + nbco_dist_finish (one node block per rank) and the two-stage exchange are the same evaluation"""
    import torch
    pos, vel = make_state(oracle32, n, "clumps")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1)
    out = []
    for split in (False, None):
        world = loopback(n, G, pos, vel, **opts)
        for _ in range(2):
            world.force(par, elastic=False, split=split)
        torch.cuda.synchronize()
        out.append(torch.cat([torch.cat([r.pos, r.vel, r.acc]) for r in world.runs]))
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("mutual", MUTUAL)
def test_sharded_lists_grow_on_demand(oracle32, mutual):
    """a domain whose traversal overflows its lists doubles them and repeats its half of the evaluation on its own (no
    collective involved); the result is still the single-GPU one"""
    import torch
    n, G, p = 65536, 2, 6
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    e1, ref = single_gpu(n, pos, vel, par, fmm_order=p, unsort=0, tree_steps=1, p2p_mutual=mutual)
    for split in (None, False):
        w = loopback(n, G, pos, vel, fmm_order=p, unsort=0, tree_steps=1, list_factor=1, list_grow=1, p2p_mutual=mutual)
        w.force(par, elastic=False, split=split)
        torch.cuda.synchronize()
        got = torch.cat([torch.cat([r.pos for r in w.runs]), torch.cat([r.vel for r in w.runs])])
        assert torch.equal(got, ref[:6 * n])
        assert same_acc(torch.cat([r.acc for r in w.runs]), ref[6 * n:], mutual, n)


def test_far_field_waits_on_the_second_stream(oracle32):
    """the multipoles may arrive after nbco_dist_finish_traverse has been enqueued, and only the context's second stream
    (nbco_aux_stream) has to wait for them: emulate a late all-gather with a slow copy on a side stream, make the
    second stream wait for it the way DomainRun._wait_far_field does, and compare with the plain evaluation"""
    import torch
    n, G, p = 32768, 2, 6
    pos, vel = make_state(oracle32, n, "clumps")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1)
    ref = loopback(n, G, pos, vel, **opts)
    ref.force(par, elastic=False)
    torch.cuda.synchronize()
    want = torch.cat([torch.cat([r.pos, r.vel, r.acc]) for r in ref.runs])

    world = loopback(n, G, pos, vel, **opts)
    runs = world.runs
    cb = runs[0].csz_bytes
    for r in runs:
        r.eng.dist_local_geom(r.buf, r.n_local, r.pos_send, r.nodes_send[:cb])
    pos_all = torch.cat([r.pos_send for r in runs])
    csz_all = torch.cat([r.nodes_send[:cb] for r in runs])
    for r in runs:
        r.eng.dist_local_mpole(r.buf, r.n_local, r.nodes_send[cb:])
    mp_all = torch.cat([r.nodes_send[cb:] for r in runs])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for r in runs:
        r.pos_all.copy_(pos_all)
        r.nodes_all[: G * cb].copy_(csz_all)
        r.nodes_all[G * cb:].zero_()                       # stale contents until the "gather" lands
        r.eng.dist_finish_traverse(r.nodes_all[: G * cb], r.pos_all)
        with torch.cuda.stream(side):
            torch.cuda._sleep(40_000_000)                  # ~20 ms: far longer than the traversal
            r.nodes_all[G * cb:].copy_(mp_all)
            landed = torch.cuda.Event()
            landed.record(side)
        aux = torch.cuda.ExternalStream(r.eng.aux_stream())
        with torch.cuda.stream(aux):
            landed.wait()                                  # what Work.wait() of an asynchronous all-gather does
        r.eng.dist_finish_rest(r.nodes_all[G * cb:], r.buf, r.acc, par)
    torch.cuda.synchronize()
    got = torch.cat([torch.cat([r.pos, r.vel, r.acc]) for r in runs])
    assert torch.equal(got, want)


def loopback(n, G, pos, vel, gather_partition=None, **opts):
    """(the domains are cut by the distributed re-partition, nbco_dist_repartition_*, unless gather_partition is set)"""
    import torch
    from coulomb_oscillators_amd import Engine, LoopbackWorld
    engines = [Engine(**opts) for _ in range(G)]
    world = LoopbackWorld(engines, n, gather_partition=gather_partition)
    nl = n // G
    # the initial ownership is arbitrary: contiguous slices of the caller's order
    world.partition([torch.from_numpy(pos[r * nl:(r + 1) * nl]).cuda() for r in range(G)],
                    [torch.from_numpy(vel[r * nl:(r + 1) * nl]).cuda() for r in range(G)])
    return world


@pytest.mark.parametrize("n,G,p,kind", [(32768, 2, 6, "reference"), (32768, 4, 4, "reference"), (32768, 8, 6, "clumps"),
                                        (40000, 8, 5, "uniform"), (24576, 2, 3, "clumps"), (1 << 20, 4, 6, "reference"), (32768, 4, 10, "uniform")])
@pytest.mark.parametrize("mutual", MUTUAL)
def test_sharded_equals_single_gpu(oracle32, n, G, p, kind, mutual):
    """(the exchange runs in its two-stage form: traversal records + positions first, multipoles behind the traversal)"""
    import torch
    pos, vel = make_state(oracle32, n, kind)
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1, p2p_mutual=mutual)
    e1, ref = single_gpu(n, pos, vel, par, **opts)
    world = loopback(n, G, pos, vel, **opts)
    world.force(par, elastic=False)
    torch.cuda.synchronize()
    nl = n // G
    got_pos = torch.cat([r.pos for r in world.runs])
    got_vel = torch.cat([r.vel for r in world.runs])
    got_acc = torch.cat([r.acc for r in world.runs])
    assert torch.equal(got_pos, ref[:3 * n]), "tree order of the positions differs"
    assert torch.equal(got_vel, ref[3 * n:6 * n]), "velocities were not carried along"
    assert torch.isfinite(got_acc).all()
    assert same_acc(got_acc, ref[6 * n:], mutual, n), "accelerations differ from the single-GPU evaluation"
    # every domain only pays for its own share of the interactions (+ the boundary)
    i1 = e1.kd_info()
    tot = sum(r.eng.kd_info().p2p_pairs for r in world.runs)
    assert i1.p2p_pairs <= tot <= 2 * i1.p2p_pairs
    assert all(r.eng.kd_info().L == i1.L for r in world.runs)


@pytest.mark.parametrize("n,G,p,kind", [(32768, 2, 6, "reference"), (32768, 4, 4, "reference"), (32768, 8, 6, "clumps"), (40000, 8, 5, "uniform"),
                                        (24576, 2, 3, "clumps"), (1 << 20, 4, 6, "reference"), (1 << 20, 8, 6, "reference"), (32768, 4, 10, "uniform")])
@pytest.mark.parametrize("mutual", MUTUAL)
def test_let_exchange_equals_single_gpu(oracle32, n, G, p, kind, mutual):
    """the locally-essential-tree exchange (nbco_dist_let_*): every rank receives only the sources its lists name, the result is
    the single-GPU one bit for bit, the guard is silent, and fewer bytes travel than with the all-gather"""
    import torch
    pos, vel = make_state(oracle32, n, kind)
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1, p2p_mutual=mutual)
    e1, ref = single_gpu(n, pos, vel, par, **opts)
    world = loopback(n, G, pos, vel, **opts)
    for _ in range(2):   # the second evaluation runs over buffers the first one has used
        world.force(par, elastic=False, let=True)
    for r in world.runs:
        r.eng.dist_let_check()
    got_acc = torch.cat([r.acc for r in world.runs])
    assert torch.equal(torch.cat([r.pos for r in world.runs]), ref[:3 * n])
    assert torch.equal(torch.cat([r.vel for r in world.runs]), ref[3 * n:6 * n])
    assert same_acc(got_acc, ref[6 * n:], mutual, n), "accelerations differ from the single-GPU evaluation"
    for r in world.runs:
        assert r.exchange_bytes() < r.allgather_bytes()
    if n == 1 << 20 and G == 8:
        ratio = sum(r.allgather_bytes() for r in world.runs) / sum(r.exchange_bytes() for r in world.runs)
        print(f"LET exchange at 8 x 128k: {ratio:.1f}x fewer bytes than the all-gather")
        assert ratio > 2
    # the energy pass reads the same sources through the same lists
    e_ref = e1.energy_fmm(ref, n, par)
    e_let = np.sum([r.eng.energy_fmm(r.buf, r.n_local, par) for r in world.runs], axis=0)
    np.testing.assert_allclose(e_let, e_ref, rtol=1e-9)


def test_let_guard_reports_a_source_that_did_not_arrive(oracle32):
    """a node record or a leaf's positions lost on the way (here: overwritten by a duplicate of another record) is in some
    interaction list of the receiver: the guard names it, nbco_dist_let_check fails"""
    import torch
    from coulomb_oscillators_amd import EngineError
    n, G, p = 32768, 4, 5
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()

    def lose_node(rank, pos_recv, mp_recv):
        if rank == 1:
            mp_recv[3].copy_(mp_recv[4])

    def lose_leaf(rank, pos_recv, mp_recv):
        if rank == 2:
            L = world.runs[0].lay.L
            leaf = (pos_recv.view(torch.int32)[:, 3].long() << L) // n   # leaf of a global particle index (evalBox's ranges)
            gone = leaf == leaf[0]
            pos_recv[gone] = pos_recv[~gone][0]

    for tamper, what in ((lose_node, "multipole of node"), (lose_leaf, "leaf")):
        world = loopback(n, G, pos, vel, fmm_order=p, unsort=0, tree_steps=1)
        world.force_let(par, elastic=False, tamper=tamper)
        bad = 0
        for r in world.runs:
            try:
                r.eng.dist_let_check()
            except EngineError as e:
                assert what in str(e)
                bad += 1
        assert bad == 1


@pytest.mark.parametrize("mutual", MUTUAL)
def test_let_lists_grow_on_demand(oracle32, mutual):
    """list overflow in one rank's traversal is reported through the count exchange and repaired by a second selection round"""
    import torch
    n, G, p = 65536, 2, 6
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    e1, ref = single_gpu(n, pos, vel, par, fmm_order=p, unsort=0, tree_steps=1, p2p_mutual=mutual)
    w = loopback(n, G, pos, vel, fmm_order=p, unsort=0, tree_steps=1, list_factor=1, list_grow=1, p2p_mutual=mutual)
    w.force(par, elastic=False, let=True)
    for r in w.runs:
        r.eng.dist_let_check()
    assert torch.equal(torch.cat([torch.cat([r.pos for r in w.runs]), torch.cat([r.vel for r in w.runs])]), ref[:6 * n])
    assert same_acc(torch.cat([r.acc for r in w.runs]), ref[6 * n:], mutual, n)


def _rows(t, n):
    """rows [x y z vx vy vz] of a domain's state, sorted: the particle SET"""
    a = np.concatenate([t[:3 * n].cpu().numpy().reshape(n, 3), t[3 * n:6 * n].cpu().numpy().reshape(n, 3)], axis=1)
    return a[np.lexsort(a.T[::-1])]


def _in_input_order(dom_pos, pos):
    """every domain holds its particles in the order of the input (the gathered state)"""
    key = {row.tobytes(): i for i, row in enumerate(np.ascontiguousarray(pos))}
    if len(key) != len(pos):
        return   # (coincident particles: no unique input index)
    for d in dom_pos:
        idx = np.fromiter((key[row.tobytes()] for row in np.ascontiguousarray(d)), dtype=np.int64, count=len(d))
        assert np.all(np.diff(idx) > 0), "a domain's particles are not in the order of the gathered state"


@pytest.mark.parametrize("gather", [True, None])
def test_pivot_ties_under_a_one_axis_chain_follow_the_input_order(oracle32, gather):
    """A rod along z: the top splits and the first local ones all cut z, so the stable-sort chain of those nodes holds no other
    axis, and two particles with the pivot's z are told apart by the LAST key alone -- the index in the input, which for a
    domain's local build is the index in its local state.  Both partitions therefore hand every domain its particles in the
    order of the gathered state (the selection levels and the scatter passes leave them in launch order): the sharded tree
    equals the single-GPU tree bit for bit, on every run."""
    import torch
    n, G, p = 32768, 4, 4
    rng = np.random.default_rng(77)
    pos = (rng.random((n, 3), dtype=np.float32) * np.array([0.01, 0.01, 1.0], dtype=np.float32)).astype(np.float32)
    vel = rng.standard_normal((n, 3)).astype(np.float32)
    order = np.argsort(pos[:, 2], kind="stable")
    # the medians of every node of global levels 0 .. 6 (cuts along z all of them: the box is 100 : 1): the last particle of the left
    # half gets the z of the first one of the right half, their x and y differ
    for l in range(0, 7):
        for j in range(1 << l):
            s, e = (n * j) >> l, (n * (j + 1)) >> l
            m = (s + e) // 2
            pos[order[m - 1], 2] = pos[order[m], 2]
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1, p2p_mutual=0)
    e1, ref = single_gpu(n, pos, vel, par, **opts)
    sd = e1.kd_array("splitdim")
    assert (sd[:127] == 2).all(), "the test's premise: levels 0 .. 6 split along z"
    for rep in range(3):
        world = loopback(n, G, pos, vel, gather_partition=gather, **opts)
        assert all(bool(r.dpart) == (gather is None) for r in world.runs)
        _in_input_order(torch.cat([r.pos for r in world.runs]).cpu().numpy().reshape(G, n // G, 3), pos)
        world.force(par, elastic=False)
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([r.pos for r in world.runs]), ref[:3 * n]), "tree order of the positions differs"
        assert torch.equal(torch.cat([r.vel for r in world.runs]), ref[3 * n:6 * n])
        assert torch.equal(torch.cat([r.acc for r in world.runs]), ref[6 * n:])
        for r in world.runs:
            r.eng.close()


@pytest.mark.parametrize("n,G,kind", [(32768, 2, "reference"), (65536, 8, "clumps"), (40000, 4, "uniform"), (1 << 20, 8, "reference"), (32768, 1, "uniform"),
                                      (131072, 16, "reference")])
def test_distributed_repartition_equals_gathered_partition(oracle32, n, G, kind):
    """nbco_dist_repartition_* (no rank gathers the state) and nbco_dist_partition (all-gather + redundant selection) cut the same
    domains: same particle sets, and after one evaluation the same tree-ordered state bit for bit"""
    import torch
    pos, vel = make_state(oracle32, n, kind)
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=4, unsort=0, tree_steps=1)
    a = loopback(n, G, pos, vel, gather_partition=True, **opts)
    b = loopback(n, G, pos, vel, **opts)
    assert all(r.dpart for r in b.runs) and not any(r.dpart for r in a.runs)
    nl = n // G
    for ra, rb in zip(a.runs, b.runs):
        np.testing.assert_array_equal(_rows(ra.buf, nl), _rows(rb.buf, nl))
        assert rb.partition_bytes is not None
        # ... in the same order, the gathered state's (the local index is the last key of the local build's sort chain)
        assert torch.equal(ra.buf[:6 * nl], rb.buf[:6 * nl])
    _in_input_order(torch.cat([r.pos for r in a.runs]).cpu().numpy().reshape(G, nl, 3), pos)
    if G > 1:
        assert sum(r.partition_bytes for r in b.runs) < sum(r.partition_bytes for r in a.runs)
    a.force(par, elastic=False)
    b.force(par, elastic=False)
    torch.cuda.synchronize()
    for ra, rb in zip(a.runs, b.runs):
        assert torch.equal(ra.buf, rb.buf)
    # a second cut from the evolved, domain-ordered state: few particles move
    for w in (a, b):
        for r in w.runs:
            r.eng.step(r.pos, r.vel, 0.05, nl)
        w.partition([r.pos for r in w.runs], [r.vel for r in w.runs])
    for ra, rb in zip(a.runs, b.runs):
        np.testing.assert_array_equal(_rows(ra.buf, nl), _rows(rb.buf, nl))
        assert torch.equal(ra.buf[:6 * nl], rb.buf[:6 * nl])
    if G > 1:
        assert sum(r.partition_bytes for r in b.runs) < 0.5 * sum(r.partition_bytes for r in a.runs)


def test_distributed_repartition_orders_pivot_ties_like_the_sort_chain(oracle32):
    """particles that tie with a pivot on the split axis, spread over several ranks: the first `need` of them in the order
    (next ancestor axes, original index) go left -- same domains as the gathered selection"""
    import torch
    n, G = 32768, 4
    pos, vel = make_state(oracle32, n, "uniform")
    rng = np.random.default_rng(3)
    # the root splits along its longest axis; give 40 particles scattered over the input exactly the median coordinate of that axis,
    # and another 30 the median of the left half along ITS axis
    ax = int(np.argmax(pos.max(0) - pos.min(0)))
    med = np.sort(pos[:, ax])[n // 2 - 1]
    pos[rng.choice(n, 40, replace=False), ax] = med
    left = np.flatnonzero(pos[:, ax] < med)
    ext = pos[left].max(0) - pos[left].min(0)
    ax2 = int(np.argmax(ext))
    med2 = np.sort(pos[left, ax2])[len(left) // 2]
    pos[rng.choice(left, 30, replace=False), ax2] = med2
    opts = dict(fmm_order=3, unsort=0, tree_steps=1)
    a = loopback(n, G, pos, vel, gather_partition=True, **opts)
    b = loopback(n, G, pos, vel, **opts)
    nl = n // G
    for ra, rb in zip(a.runs, b.runs):
        np.testing.assert_array_equal(_rows(ra.buf, nl), _rows(rb.buf, nl))


def test_distributed_repartition_falls_back_to_the_gathered_form_on_heavy_ties(oracle32):
    """Coordinates on a lattice: far more than 64 particles of every rank tie with each pivot, which nbco_dist_repartition_*
    reports as NBCO_ERR_UNSUPPORTED -- on every rank alike, before the local state is touched.  The world switches to
    nbco_dist_partition (which takes any input) instead of failing, cuts the same domains as a world that used the gathered
    form from the start, and stays on it for later cuts (DomainRun.partition, which force() calls for the periodic rebalance,
    takes the same path)."""
    import torch
    n, G = 32768, 4
    pos, vel = make_state(oracle32, n, "uniform")
    q = (pos.max(0) - pos.min(0)).max() / 24.0
    pos = (np.round(pos / q) * q).astype(np.float32)       # ~25 distinct values per axis: thousands of ties per pivot
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=3, unsort=0, tree_steps=1)
    a = loopback(n, G, pos, vel, gather_partition=True, **opts)
    b = loopback(n, G, pos, vel, **opts)
    assert all(r.partition_fallbacks == 1 and not r.dpart for r in b.runs)
    assert all(r.partition_fallbacks == 0 for r in a.runs)
    nl = n // G
    for ra, rb in zip(a.runs, b.runs):
        np.testing.assert_array_equal(_rows(ra.buf, nl), _rows(rb.buf, nl))
    for w in (a, b):
        w.force(par, elastic=False)
        for r in w.runs:
            r.eng.step(r.pos, r.vel, 0.05, nl)
        w.partition([r.pos for r in w.runs], [r.vel for r in w.runs])      # a later cut: the gathered form, no second fallback
        w.force(par, elastic=False)
    torch.cuda.synchronize()
    assert all(r.partition_fallbacks == 1 and not r.dpart for r in b.runs)
    for ra, rb in zip(a.runs, b.runs):
        assert torch.equal(ra.buf, rb.buf) and bool(torch.isfinite(rb.acc).all())


def test_sharded_matches_oracle(oracle32):
    """end to end against the CPU oracle (1e-5 relative, the bar of the single-GPU path)"""
    import torch
    from nbutil import force_err
    n, G, p = 16384, 4, 6
    o = oracle32
    buf = o.init_reference(n)
    par = o.params(n)
    want = o.fmm_kd(buf[:2].copy(), par, p=p, threads=4, unsort=True)[1]
    world = loopback(n, G, np.ascontiguousarray(buf[0]), np.ascontiguousarray(buf[1]), fmm_order=p, unsort=0, tree_steps=1)
    world.force(torch.from_numpy(par).cuda(), elastic=False)
    torch.cuda.synchronize()
    # undo the tree order: the carried positions identify the particles (all distinct)
    got_pos = torch.cat([r.pos for r in world.runs]).cpu().numpy().reshape(n, 3)
    got_acc = torch.cat([r.acc for r in world.runs]).cpu().numpy().reshape(n, 3)
    key = lambda x: np.lexsort((x[:, 2], x[:, 1], x[:, 0]))
    ka, kb = key(got_pos), key(buf[0])
    np.testing.assert_array_equal(got_pos[ka], buf[0][kb])
    acc = np.empty_like(got_acc)
    acc[kb] = got_acc[ka]
    assert force_err(acc, want) < 1e-5


@pytest.mark.parametrize("let", [False, True])
def test_sharded_leapfrog_with_tree_reuse(oracle32, let):
    """several leapfrog steps with opts.tree_steps = rebalance = 3: trajectories stay identical to the single GPU"""
    import torch
    from coulomb_oscillators_amd import Engine
    n, G, p, steps, dt = 32768, 4, 5, 7, 1e-3
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=3, p2p_mutual=0)   # bit-identical trajectories need the one-directional kernel
    # single GPU, same kernel sequence
    e = Engine(**opts)
    buf = torch.cat([torch.from_numpy(pos).reshape(-1), torch.from_numpy(vel).reshape(-1), torch.zeros(3 * n)]).cuda()
    P, V, A = buf[:3 * n], buf[3 * n:6 * n], buf[6 * n:]

    def f1():
        e.fmm_cart3_kdtree(buf, A, n, par)
        e.add_elastic(P, A, n, par[3:])
    f1()
    for _ in range(steps):
        e.step(V, A, 0.5 * dt, n); e.step(P, V, dt, n); f1(); e.step(V, A, 0.5 * dt, n)
    # sharded
    world = loopback(n, G, pos, vel, **opts)
    for r in world.runs:
        r.rebalance = 0
    nl = n // G

    def fG(k):
        if k > 0 and k % 3 == 0:
            world.partition([r.pos for r in world.runs], [r.vel for r in world.runs])
        world.force(par, elastic=True, let=let)
    fG(0)
    for s in range(steps):
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
        fG(s + 1)
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([r.pos for r in world.runs]), P)
    assert torch.equal(torch.cat([r.vel for r in world.runs]), V)
    assert torch.equal(torch.cat([r.acc for r in world.runs]), A)


def test_layout_and_argument_errors(engine):
    from coulomb_oscillators_amd import EngineError
    engine.set(fmm_order=6)
    lay = engine.dist_layout(1 << 20, 8, 3)
    assert (lay.d, lay.n_local, lay.L_local) == (3, (1 << 20) // 8, lay.L - 3)
    assert lay.nodes_bytes == lay.ntot_local * (16 + 4 * 56) and lay.pos_bytes == 16 * lay.n_local
    for bad in [dict(n_global=1 << 20, world=3, rank=0), dict(n_global=(1 << 20) + 1, world=2, rank=0),
                dict(n_global=1 << 20, world=2, rank=2), dict(n_global=4096, world=2, rank=0)]:
        with pytest.raises(EngineError):
            engine.dist_layout(**bad)
    engine.set(fmm_order=10)
    assert engine.dist_layout(1 << 20, 2, 0).order == 10


def test_sharded_stale_domains_stay_correct(oracle32):
    """tree rebuilt every step but the domains re-cut only at the start: particles leave their domain's box
    (the local root box must grow with them) and the result must stay an accurate force field"""
    import torch
    from coulomb_oscillators_amd import Engine
    n, G, p, steps, dt = 65536, 4, 6, 6, 0.05
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    world = loopback(n, G, pos, vel, fmm_order=p, unsort=0, tree_steps=1)
    nl = n // G
    world.force(par, elastic=True)
    for _ in range(steps):
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
        world.force(par, elastic=False)
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl)
    torch.cuda.synchronize()
    P = torch.cat([r.pos for r in world.runs]).contiguous()
    A = torch.cat([r.acc for r in world.runs])
    assert torch.isfinite(A).all()
    # yardstick: the single-GPU FMM on the same final positions, both against the direct sum (the reference's
    # default opening radius gives a mean relative error of ~1e-2 at p = 6)
    e = Engine(fmm_order=p, unsort=1)
    ref, one = torch.zeros_like(A), torch.zeros_like(A)
    e.direct3(P, ref, n, par)
    e.fmm_cart3_kdtree(P.clone(), one, n, par)
    torch.cuda.synchronize()

    def mean_rel(x):
        x, r = x.cpu().numpy().reshape(n, 3).astype(np.float64), ref.cpu().numpy().reshape(n, 3).astype(np.float64)
        return float((np.linalg.norm(x - r, axis=1) / np.linalg.norm(r, axis=1)).mean())
    err_sharded, err_single = mean_rel(A), mean_rel(one)
    assert err_single < 2e-2
    assert err_sharded < 1.25 * err_single, (err_sharded, err_single)


def test_sharded_local_builds_use_the_warm_select(oracle32):
    """between two cuts of the domains every rank rebuilds its subtree from slowly moving particles: the local builds run the
    one-pass (warm) median select, and the run equals the one with cold builds bit for bit"""
    import os
    import torch
    from coulomb_oscillators_amd import Engine, LoopbackWorld
    n, G, p, steps, dt = 1 << 17, 2, 4, 6, 5e-4
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    nl = n // G
    out, warm_builds = [], []
    for warm in ("1", "0"):
        os.environ["NBCO_SEL_WARM"] = warm
        try:
            engines = [Engine(fmm_order=p, unsort=0, tree_steps=1) for _ in range(G)]
        finally:
            del os.environ["NBCO_SEL_WARM"]
        world = LoopbackWorld(engines, n)
        world.partition([torch.from_numpy(pos[r * nl:(r + 1) * nl]).cuda() for r in range(G)], [torch.from_numpy(vel[r * nl:(r + 1) * nl]).cuda() for r in range(G)])
        world.force(par, elastic=True, let=True)
        for _ in range(steps):
            for r in world.runs:
                r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
            world.force(par, elastic=True, let=True)
            for r in world.runs:
                r.eng.step(r.vel, r.acc, 0.5 * dt, nl)
        torch.cuda.synchronize()
        out.append(torch.cat([r.buf for r in world.runs]))
        warm_builds.append(sum(int(r.eng.kd_info().warm_builds) for r in world.runs))
    assert torch.equal(out[0], out[1])
    assert warm_builds[0] >= G * (steps - 1) and warm_builds[1] == 0


def test_collectives_over_rccl_in_a_world_of_one(oracle32):
    """the one card cannot host two RCCL ranks, but a world of one still runs every collective DomainRun issues through RCCL
    (torch.distributed backend "nccl"): int32 MIN / SUM all-reduces, byte and int64 all-gathers, the asynchronous all-gather,
    the all-to-all with row splits -- dtypes, views into the workspace, reduce ops.  Result: the single-GPU one."""
    import socket
    import torch
    import torch.distributed as dist
    from coulomb_oscillators_amd import Engine, DomainRun, TorchComm
    n, p, steps, dt = 65536, 4, 3, 5e-4
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    e1, ref = single_gpu(n, pos, vel, par, fmm_order=p, unsort=0, tree_steps=1)
    e1.add_elastic(ref[:3 * n], ref[6 * n:], n, par[3:])
    for _ in range(steps):
        e1.step(ref[3 * n:6 * n], ref[6 * n:], 0.5 * dt, n); e1.step(ref[:3 * n], ref[3 * n:6 * n], dt, n)
        e1.fmm_cart3_kdtree(ref, ref[6 * n:], n, par)
        e1.add_elastic(ref[:3 * n], ref[6 * n:], n, par[3:])
        e1.step(ref[3 * n:6 * n], ref[6 * n:], 0.5 * dt, n)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        for let in (True, False):
            run = DomainRun(Engine(fmm_order=p, unsort=0, tree_steps=1), n, TorchComm(always_collective=True), rebalance=1, let=let)
            assert run.dpart and run.let == let
            run.partition(torch.from_numpy(pos).cuda().reshape(-1), torch.from_numpy(vel).cuda().reshape(-1))
            run.force(par)
            for _ in range(steps):
                run.leapfrog(par, dt)       # (rebalance = 1: the domain is cut again before every evaluation)
            if let:
                run.eng.dist_let_check()
            kin = run.energy(par)[0]
            torch.cuda.synchronize()
            assert torch.equal(run.buf, ref), "let=%s" % let
            assert kin > 0
            # and K more steps with the fused pass in between (DomainRun.leapfrog_steps) against K more calls of leapfrog
            twin = DomainRun(Engine(fmm_order=p, unsort=0, tree_steps=1), n, TorchComm(always_collective=True), rebalance=1, let=let)
            twin.partition(torch.from_numpy(pos).cuda().reshape(-1), torch.from_numpy(vel).cuda().reshape(-1))
            twin.force(par)
            for _ in range(steps + 4):
                twin.leapfrog(par, dt)
            run.leapfrog_steps(par, dt, 4)
            torch.cuda.synchronize()
            assert torch.equal(run.buf, twin.buf), "leapfrog_steps, let=%s" % let
    finally:
        dist.destroy_process_group()


def test_flagged_build_restarts_the_let_evaluation(oracle32):
    """LET form: no host round trip behind the local build -- a flagged build (here: the warm select misses after the positions
    were stretched by 30 %) is reported with the counts and every rank starts the evaluation over; same result as with cold
    builds, which are never flagged here"""
    import os
    import torch
    from coulomb_oscillators_amd import Engine, LoopbackWorld
    n, G, p = 1 << 17, 2, 4
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    nl = n // G
    out, misses = [], []
    for warm in ("1", "0"):
        os.environ["NBCO_SEL_WARM"] = warm
        try:
            engines = [Engine(fmm_order=p, unsort=0, tree_steps=1) for _ in range(G)]
        finally:
            del os.environ["NBCO_SEL_WARM"]
        world = LoopbackWorld(engines, n)
        world.partition([torch.from_numpy(pos[r * nl:(r + 1) * nl]).cuda() for r in range(G)], [torch.from_numpy(vel[r * nl:(r + 1) * nl]).cuda() for r in range(G)])
        world.force(par, elastic=False, let=True)
        world.force(par, elastic=False, let=True)
        for r in world.runs:
            r.pos.mul_(1.3)
        world.force(par, elastic=False, let=True)
        world.force(par, elastic=False, let=True)
        for r in world.runs:
            r.eng.dist_let_check()
        torch.cuda.synchronize()
        out.append(torch.cat([r.buf for r in world.runs]))
        misses.append(sum(int(r.eng.kd_info().warm_misses) for r in world.runs))
    assert torch.equal(out[0], out[1])
    assert misses[0] >= 1 and misses[1] == 0


def _leapfrog_let(world, par, dt, steps, nl, **kw):
    """kick-drift-kick with the LET exchange in lockstep (first force included)"""
    world.force(par, elastic=True, let=True, **kw)
    for _ in range(steps):
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
        world.force(par, elastic=True, let=True, **kw)
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl)


@pytest.mark.parametrize("n,G,p,kind,opts", [(32768, 2, 6, "reference", {}), (65536, 4, 4, "clumps", {}), (1 << 17, 8, 5, "reference", {}),
                                             (32768, 4, 6, "uniform", {"far_fp64": 1}), (1 << 17, 4, 6, "reference", {"tree_steps": 3}),
                                             (65536, 4, 6, "reference", {"p2p_mutual": 1})])
def test_capped_let_exchange_equals_the_exact_one(oracle32, n, G, p, kind, opts):
    """nbco_dist_let_pack_capped / _finish_capped / _settle: segments sized from the evaluation before, free records marked, no host
    round trip in the middle -- same state as the exact exchange bit for bit over a run of steps, every evaluation after the
    first stood in the capped form, the guard is silent"""
    import torch
    pos, vel = make_state(oracle32, n, kind)
    par = torch.from_numpy(oracle32.params(n)).cuda()
    steps, dt, nl = 5, 5e-4, n // G
    o = dict(fmm_order=p, unsort=0, tree_steps=1)
    o.update(opts)
    out = []
    for capped in (False, True):
        w = loopback(n, G, pos, vel, **o)
        _leapfrog_let(w, par, dt, steps, nl, capped=capped)
        for r in w.runs:
            r.eng.dist_let_check()
        torch.cuda.synchronize()
        out.append(torch.cat([r.buf for r in w.runs]))
        if capped:
            assert w.let_capped_evals == steps and w.let_redos == 0
            for r in w.runs:
                assert r.exchange_bytes() < r.allgather_bytes()
    assert torch.isfinite(out[0]).all()
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("what", ["nodes", "particles", "both"])
@pytest.mark.parametrize("tree_steps", [1, 3])
def test_capped_let_attempt_that_overflows_is_void_and_repeated(oracle32, what, tree_steps):
    """segments too small for what was selected: records are dropped (nothing is written out of a segment), the gathered counts say
    so on every rank, nbco_dist_let_settle(0) declares the attempt void, the evaluation is repeated in the exact form -- same
    state as a run that never overflowed; the guard words of the void attempt (it DID miss sources) are dropped, not reported"""
    import torch
    n, G, p, steps, dt = 65536, 4, 5, 6, 5e-4
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    nl = n // G
    calls = [0]

    def squeeze(capn, capp):
        calls[0] += 1
        if calls[0] in (2, 4):   # (attempt 3 follows a repeated evaluation, attempt 5 another)
            if what in ("nodes", "both"):
                capn[1, 2] = 3
                capn[3, 0] //= 2
            if what in ("particles", "both"):
                capp[2, 1] = 17
                capp[0, 3] //= 3

    out = []
    for capped in (False, True):
        w = loopback(n, G, pos, vel, fmm_order=p, unsort=0, tree_steps=tree_steps)
        _leapfrog_let(w, par, dt, steps, nl, **(dict(capped=True, squeeze=squeeze) if capped else {}))
        for r in w.runs:
            r.eng.dist_let_check()
        torch.cuda.synchronize()
        out.append(torch.cat([r.buf for r in w.runs]))
        if capped:
            assert w.let_redos == 2 and w.let_capped_evals == steps - 2
    assert torch.equal(out[0], out[1])


def test_capped_let_attempt_with_a_flagged_build_is_repeated(oracle32):
    """capped form: a flagged build (the warm select misses after the positions were stretched by 30 %) is only seen once the whole
    evaluation is queued -- it ran on a tree that is not the reference's.  The flag travels with the counts, every rank declares
    the attempt void and repeats it (the flagged rank with a cold build): same result as with cold builds, which are never flagged"""
    import os
    import torch
    from coulomb_oscillators_amd import Engine, LoopbackWorld
    n, G, p = 1 << 17, 2, 4
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    nl = n // G
    out, misses, redos = [], [], []
    for warm in ("1", "0"):
        os.environ["NBCO_SEL_WARM"] = warm
        try:
            engines = [Engine(fmm_order=p, unsort=0, tree_steps=1) for _ in range(G)]
        finally:
            del os.environ["NBCO_SEL_WARM"]
        world = LoopbackWorld(engines, n)
        world.partition([torch.from_numpy(pos[r * nl:(r + 1) * nl]).cuda() for r in range(G)], [torch.from_numpy(vel[r * nl:(r + 1) * nl]).cuda() for r in range(G)])
        world.force(par, elastic=False, let=True, capped=True)
        world.force(par, elastic=False, let=True, capped=True)
        for r in world.runs:
            r.pos.mul_(1.3)
        world.force(par, elastic=False, let=True, capped=True)
        world.force(par, elastic=False, let=True, capped=True)
        for r in world.runs:
            r.eng.dist_let_check()
        torch.cuda.synchronize()
        out.append(torch.cat([r.buf for r in world.runs]))
        misses.append(sum(int(r.eng.kd_info().warm_misses) for r in world.runs))
        redos.append(world.let_redos)
    assert torch.equal(out[0], out[1])
    assert misses[0] >= 1 and misses[1] == 0
    assert redos[0] >= 1 and redos[1] == 0


@pytest.mark.parametrize("tree_steps,recut", [(1, 0), (3, 0), (1, 4)])
def test_sharded_turnaround_equals_step_kernels(oracle32, tree_steps, recut):
    """nbco_dist_turnaround (one pass between two force evaluations of a sharded leapfrog run: elastic term, both half kicks, drift,
    next build's prologue) against nbco_add_elastic + three nbco_step calls + the build's own prologue: same state bit for bit,
    through tree reuse and through a re-cut of the domains in the middle (which discards the prologue)"""
    import torch
    n, G, p, steps, dt = 1 << 17, 4, 4, 7, 5e-4
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    nl = n // G
    out = []
    for fused in (False, True):
        w = loopback(n, G, pos, vel, fmm_order=p, unsort=0, tree_steps=tree_steps)
        w.force(par, elastic=True, let=True)
        for r in w.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
        for k in range(steps):
            if recut and k == recut:
                w.partition([r.pos for r in w.runs], [r.vel for r in w.runs])
            w.force(par, elastic=False, let=True)
            last = k + 1 == steps
            for r in w.runs:
                if fused and not last:
                    r.eng.dist_turnaround(r.buf, nl, par, dt)
                else:
                    r.eng.add_elastic(r.pos, r.acc, nl, par[3:])
                    r.eng.step(r.vel, r.acc, 0.5 * dt, nl)
                    if not last:
                        r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
        torch.cuda.synchronize()
        # (the fused pass does not write the accelerations of the steps in between: compare positions, velocities and the last ones)
        out.append(torch.cat([r.buf for r in w.runs]))
    assert torch.equal(out[0], out[1])


def test_energy_after_a_turnaround_fails_loudly(oracle32):
    """nbco_dist_turnaround (like the fused pass of nbco_integrate_steps) drifts the particles and may overwrite the tree-ordered
    position copy with the next build's input: the interaction lists of the last evaluation no longer belong to any state the
    caller can hand in, so nbco_energy_fmm must refuse instead of combining old lists with new positions.  The same holds once a
    re-partition has begun (its first stage packs into the same buffer)."""
    import torch
    from coulomb_oscillators_amd import EngineError
    n, G, dt = 32768, 2, 5e-4
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    w = loopback(n, G, pos, vel, fmm_order=4, unsort=0, tree_steps=1)
    w.force(par, elastic=False, let=True)
    r = w.runs[0]
    assert np.isfinite(r.eng.energy_fmm(r.buf, r.n_local, par)).all()          # right behind an evaluation: fine
    r.eng.dist_turnaround(r.buf, r.n_local, par, dt)
    with pytest.raises(EngineError, match="no kd-tree evaluation"):
        r.eng.energy_fmm(r.buf, r.n_local, par)
    r1 = w.runs[1]
    assert np.isfinite(r1.eng.energy_fmm(r1.buf, r1.n_local, par)).all()
    r1.eng.dist_repartition_begin(r1.buf, n, G, 1, r1.work)
    with pytest.raises(EngineError, match="no kd-tree evaluation"):
        r1.eng.energy_fmm(r1.buf, r1.n_local, par)


def test_distributed_repartition_at_8m_particles(oracle32):
    """n_global = 2^23 (BASELINE config 4 is made of such sizes): the windows of the distributed median selection need their
    THIRD histogram pass (11 + 11 + 10 bits) at a size where the first two leave thousands of keys -- the same particle sets as
    the gathered selection (whose level-0 node, above 2^22 particles, takes the three-pass select of k_kdselect.hip)"""
    import torch
    n, G = 1 << 23, 2
    pos, vel = make_state(oracle32, n, "uniform")
    opts = dict(fmm_order=3, unsort=0, tree_steps=1)
    a = loopback(n, G, pos, vel, gather_partition=True, **opts)
    b = loopback(n, G, pos, vel, **opts)
    assert all(r.dpart and r.partition_fallbacks == 0 for r in b.runs)
    nl = n // G
    for ra, rb in zip(a.runs, b.runs):
        # (a 4M-row lexsort per domain would take most of the test's time: compare the sets through order-independent sums of the
        # rows' bit patterns and, exactly, through the sorted split coordinate)
        A, B = ra.buf[:6 * nl].view(torch.int32).to(torch.int64), rb.buf[:6 * nl].view(torch.int32).to(torch.int64)
        assert int(A.sum()) == int(B.sum()) and int((A * A).sum()) == int((B * B).sum())
        for ax in range(3):
            assert torch.equal(torch.sort(ra.buf[ax:3 * nl:3]).values, torch.sort(rb.buf[ax:3 * nl:3]).values)
    # the cut itself: everything of rank 0 lies on one side of everything of rank 1 along the root's split axis
    lo, hi = b.runs[0].pos.view(nl, 3), b.runs[1].pos.view(nl, 3)
    ax = int(torch.argmax(torch.from_numpy(pos.max(0) - pos.min(0))))
    assert float(lo[:, ax].max()) <= float(hi[:, ax].min())
    assert sum(r.partition_bytes for r in b.runs) < sum(r.partition_bytes for r in a.runs)


def test_config_four_at_its_own_size_in_lockstep(oracle32):
    """BASELINE config 4 ("FMM-3D kd-tree p=6, N=16M, domain-decomposed 8 x MI355X") at its own size on the one card: N = 2^24
    particles of the reference's ball, eight kd-domains of 2^21 driven in lockstep -- distributed re-partition (three histogram
    passes at this size), local builds with trees of 2^18 leaves, LET exchange in the exact and then in the capped form.  Tree
    order, velocities and accelerations equal the single-GPU evaluation of the same 16M system bit for bit, and every domain
    receives a small fraction of what the all-gather would move."""
    import torch
    n, G, p = 1 << 24, 8, 6
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1)
    e1, ref = single_gpu(n, pos, vel, par, **opts)
    info = e1.kd_info()
    assert info.L == 19 and info.n == n
    ref = ref.clone()
    e1.close()
    assert bool(torch.isfinite(ref).all())
    world = loopback(n, G, pos, vel, **opts)
    assert all(r.dpart and r.partition_fallbacks == 0 for r in world.runs)
    for capped in (False, True):
        world.force(par, elastic=False, let=True, capped=capped)
        for r in world.runs:
            r.eng.dist_let_check()
        torch.cuda.synchronize()
        got = torch.cat([r.buf.view(3, -1, 3) for r in world.runs], dim=1).reshape(-1)
        assert torch.equal(got[:6 * n], ref[:6 * n]), "tree order of positions / velocities differs (capped=%s)" % capped
        assert torch.equal(got[6 * n:], ref[6 * n:]), "accelerations differ from the single-GPU evaluation (capped=%s)" % capped
    assert world.let_capped_evals == 1 and world.let_redos == 0
    ratio = sum(r.allgather_bytes() for r in world.runs) / sum(r.exchange_bytes() for r in world.runs)
    print("config 4 in lockstep: LET exchange moves %.1fx fewer bytes than the all-gather (%.1f MB per domain and evaluation)"
          % (ratio, np.mean([r.exchange_bytes() for r in world.runs]) / 1e6))
    assert ratio > 10
    for r in world.runs:
        r.eng.close()


@pytest.mark.parametrize("let", [False, True])
def test_sharded_far_fp64_equals_single_gpu(oracle32, let):
    """opts.far_fp64 in the sharded evaluation (G = 4): the exchanged multipole blocks / LET node records are doubles, and the
    accelerations, the tree order and the double tuples of the own subtree equal the single-GPU far_fp64 evaluation bit for bit"""
    import torch
    n, G, p = 32768, 4, 10
    pos, vel = make_state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1, far_fp64=1)
    e1, ref = single_gpu(n, pos, vel, par, **opts)
    assert e1.kd_info().real_bytes == 8
    world = loopback(n, G, pos, vel, **opts)
    assert int(world.runs[0].lay.mpole_bytes) == int(world.runs[0].lay.ntot_local) * 8 * (p * (p + 1) * (p + 2) // 6)
    world.force(par, elastic=False, let=let)
    torch.cuda.synchronize()
    got = torch.cat([r.buf.view(3, -1, 3) for r in world.runs], dim=1).reshape(-1)
    assert torch.equal(got, ref)
    assert bool(torch.isfinite(got).all())
