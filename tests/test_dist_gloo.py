"""World-size-2 rehearsal of the view-sharded step on CPU (gloo): fmhr_amd.dist's partition + the ONE packed all-reduce of
[photometric share without its divisor | the rest | loss slots] + the combine must reproduce the single-process full-batch result.
The compute is the CPU oracle (the product has no CPU path); what is under test is the host-side protocol the GPU ranks run over
RCCL (HamStage2.step(group=...) does the same in place with the combine kernel)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_step(oracle, sc, s0, s1, n_total, reg_grads):
    """what a view-sharded rank hands to the exchange, on the oracle: the photometric term with the L1 divisor left out (P), the
    silhouette term + this rank's regularisers (R), the loss slots and the LOCAL valid-pixel count — no exchange in between."""
    sl = slice(s0, s1)
    d = sc["delta"].clone().requires_grad_(True)
    a = sc["albedo"][None].clone().requires_grad_(True)
    H, W = sc["H"], sc["W"]
    with torch.no_grad():
        _, pos = oracle.clip_positions(sc["vertices"] + sc["delta"], sc["w2c"][sl], sc["proj"][sl])
        rast, _ = oracle.rasterize(None, pos.contiguous(), sc["faces"], (H, W))
        n_valid = ((sc["mask"][sl] > 0) & (rast[..., 3] > 0)).sum().to(torch.float32)
    out = oracle.ham_stage2_losses(sc["vertices"], d, a, sc["sh0"][sl], sc["w2c"][sl], sc["proj"][sl], sc["img"][sl],
                                   sc["mask"][sl], sc["valid_mask"][sl], sc["faces"], (H, W), dict(sc["weights"]), sc["elm"],
                                   sfs_denominator=1.0, mse_denominator=float(n_total * H * W))
    pd, pa = torch.autograd.grad(out["sfs"], (d, a), retain_graph=True)
    reg_terms = out["lap"] + out["albedo"] + out["edge"] + out["delta"]
    rd, ra = torch.autograd.grad(out["mask"] + (reg_terms if reg_grads else 0.0 * reg_terms), (d, a))
    r = 1.0 if reg_grads else 0.0
    share = {k: out[k].detach() * (r if k in ("lap", "albedo", "edge", "delta") else 1.0) for k in ("sfs", "lap", "albedo", "mask", "edge", "delta")}
    losses = torch.stack([share[k] for k in ("sfs", "lap", "albedo", "mask", "edge", "delta")] + [torch.zeros(()), n_valid]).to(torch.float32)
    return losses, pd, pa[0], rd, ra[0]


def _reference_step(oracle, sc):
    """the single-process full-batch step with the reference's divisors (mesh_sfs_optim.py:289,295)"""
    d = sc["delta"].clone().requires_grad_(True)
    a = sc["albedo"][None].clone().requires_grad_(True)
    out = oracle.ham_stage2_losses(sc["vertices"], d, a, sc["sh0"], sc["w2c"], sc["proj"], sc["img"], sc["mask"], sc["valid_mask"],
                                   sc["faces"], (sc["H"], sc["W"]), dict(sc["weights"]), sc["elm"])
    out["loss"].backward()
    losses = torch.stack([out[k].detach() for k in ("sfs", "lap", "albedo", "mask", "edge", "delta", "loss")]).to(torch.float32)
    return losses, d.grad, a.grad[0], float(out["n_valid"]) if "n_valid" in out else None


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    import ham_oracle as oracle
    from fmhr_amd import dist as fdist
    from checker_scenes import prepare
    r, w, _ = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # the direct RCCL communicator (StreamAllReduce) only exists on the nccl backend: everywhere else the helper hands back the torch group
    assert fdist.make_stream_all_reduce(dist.group.WORLD) is dist.group.WORLD
    try:
        fdist.StreamAllReduce(dist.group.WORLD)
        raise AssertionError("StreamAllReduce must refuse a gloo group")
    except RuntimeError:
        pass
    sc = prepare(oracle, (5, 64, 48, 10, 14), 7)           # 5 views over 2 ranks: ragged 3 + 2 split
    s0, s1 = fdist.shard_views(sc["views"], rank, world)
    sh = fdist.ShardedStep(sc["V"], torch.device("cpu"))
    ls, gd, ga = sh.reduce(*_oracle_step(oracle, sc, s0, s1, sc["views"], rank == 0))
    if rank == 0:
        q.put((ls.numpy().copy(), gd.numpy().copy(), ga.numpy().copy(), (s0, s1)))       # by value (see test_gpu_sharded.py)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_match_single_process(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fmhr_amd import dist as fdist
    from checker_scenes import prepare
    from util import rel_err
    assert [fdist.shard_views(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [fdist.shard_views(48, r, 8) for r in range(8)] == [(6 * r, 6 * r + 6) for r in range(8)]
    assert fdist.shard_views(3, 3, 4) == (3, 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ls, gd, ga, shard0 = q.get(timeout=240)
    ls, gd, ga = torch.from_numpy(ls), torch.from_numpy(gd), torch.from_numpy(ga)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert shard0 == (0, 3)
    sc = prepare(oracle, (5, 64, 48, 10, 14), 7)
    ref_l, ref_gd, ref_ga, _ = _reference_step(oracle, sc)
    assert rel_err(ls[:7], ref_l[:7]) < 1e-5, (ls, ref_l)
    assert float(ls[7]) > 100                      # the summed LOCAL counts = the batch-global valid-pixel count
    one = fdist.ShardedStep(sc["V"], torch.device("cpu")).reduce(*_oracle_step(oracle, sc, 0, 5, 5, True))
    assert float(one[0][7]) == float(ls[7])
    assert rel_err(gd, ref_gd) < 1e-5 and rel_err(ga, ref_ga) < 1e-5


def _frames_worker(rank, world, port, q, n_frames):
    """BASELINE.json configs[4] in small: `n_frames` frames x 4 views of a two-hand mesh on `world` ranks, partitioned frames first,
    then views (fmhr_amd.dist.shard_frames_views); a frame is an independent problem, only the ranks that share a frame exchange."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    import ham_oracle as oracle
    from fmhr_amd import dist as fdist
    from checker_scenes import prepare
    fdist.init_from_env(backend="gloo")
    part = fdist.shard_frames_views(n_frames, 4, rank, world)
    groups = fdist.frame_groups(n_frames, world)                  # collective: every rank creates every group
    out = {}
    for f in range(*part["frames"]):
        sc = prepare(oracle, (4, 48, 40, 8, 10, 0, 2), 50 + f)   # frame f: its own two-hand geometry / targets
        v0, v1 = part["views"]
        sh = fdist.ShardedStep(sc["V"], torch.device("cpu"), group=groups[f], exchange=part["peers"] is not None)
        ls, gd, ga = sh.reduce(*_oracle_step(oracle, sc, v0, v1, 4, part["view_rank"] == 0))
        out[f] = (ls.numpy().copy(), gd.numpy().copy(), ga.numpy().copy())
    q.put((rank, part["frames"], part["views"], part["peers"], out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n_frames", [2, 1], ids=["frames2_no_exchange", "frames1_view_split"])
def test_frames_then_views_partition(oracle, n_frames):
    """world 2: two frames -> one frame per rank, all views, NO collective; one frame -> the two ranks split its views and all-reduce
    inside the frame's subgroup.  Either way every frame's result equals the single-process step of that frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fmhr_amd import dist as fdist
    from checker_scenes import prepare
    from util import rel_err
    # the partition itself, at the sizes of BASELINE.json configs[4] (8 frames x 32 views)
    assert [fdist.shard_frames_views(8, 32, r, 8)["frames"] for r in range(8)] == [(r, r + 1) for r in range(8)]
    assert all(fdist.shard_frames_views(8, 32, r, 8)["peers"] is None for r in range(8))
    p16 = [fdist.shard_frames_views(8, 32, r, 16) for r in range(16)]
    assert [p["frames"] for p in p16] == [(r // 2, r // 2 + 1) for r in range(16)] and p16[5]["views"] == (16, 32) and p16[5]["peers"] == [4, 5]
    assert fdist.shard_frames_views(8, 32, 1, 2)["frames"] == (4, 8)
    with pytest.raises(ValueError):
        fdist.shard_frames_views(8, 32, 0, 12)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frames_worker, args=(r, 2, port, q, n_frames)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = {}
    for rank, frames, views, peers, out in res:
        if n_frames == 2:
            assert frames == (rank, rank + 1) and views == (0, 4) and peers is None
        else:
            assert frames == (0, 1) and views == (2 * rank, 2 * rank + 2) and peers == [0, 1]
        seen.update({(f, rank): v for f, v in out.items()})
    for f in range(n_frames):
        sc = prepare(oracle, (4, 48, 40, 8, 10, 0, 2), 50 + f)
        ref_l, ref_gd, ref_ga, _ = _reference_step(oracle, sc)
        for (ff, rank), (ls, gd, ga) in seen.items():
            if ff != f:
                continue
            assert rel_err(torch.from_numpy(ls)[:7], ref_l[:7]) < 1e-5
            assert rel_err(torch.from_numpy(gd), ref_gd) < 1e-5 and rel_err(torch.from_numpy(ga), ref_ga) < 1e-5


# ---- the negotiation of the direct RCCL communicator (fmhr_amd.dist.negotiate_stream_all_reduce) with fakes over gloo ------------
class _FakeComm:
    """stands in for StreamAllReduce: `mode` says what goes wrong on `bad_rank` (prepare raises / connect never completes / the test
    vector comes out wrong); the good ranks' connect() would wait for the peer like ncclCommInitRank does — the protocol must never
    put them there when a peer is not coming."""
    entered_connect = False

    def __init__(self, group, mode, bad_rank):
        self.group, self.mode, self.bad = group, mode, dist.get_rank(group) == bad_rank
        self.aborted = False

    def prepare(self):
        if self.mode == "prepare" and self.bad:
            raise OSError("librccl.so: cannot open shared object file")

    def id_bytes(self):
        return bytes(range(128)) if dist.get_rank(self.group) == 0 else bytes(128)

    def set_id(self, raw):
        assert raw == bytes(range(128)), "the id must reach every rank"

    def connect(self, timeout):
        import time
        _FakeComm.entered_connect = True
        if self.mode == "connect" and self.bad:
            time.sleep(timeout)                                     # a communicator that never comes up: polled until the deadline
            raise TimeoutError("ncclCommInitRankConfig still in progress at its deadline")
        # a good rank: the non-blocking initialisation completes (or, with a missing peer, runs into the same deadline)
        if self.mode == "connect":
            time.sleep(timeout)
            raise TimeoutError("peer never arrived")

    def all_reduce_(self, t, stream=None):
        dist.all_reduce(t, group=self.group)
        if self.mode == "verify" and self.bad:
            t.add_(1.0)
        return t

    def abort(self):
        self.aborted = True

    def close(self):
        pass


def _negotiation_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from fmhr_amd import dist as fdist
    fdist.init_from_env(backend="gloo")
    cpu = torch.device("cpu")
    res = {}
    for mode in ("prepare", "connect", "verify", "ok"):
        _FakeComm.entered_connect = False
        made = []

        def factory(g, mode=mode):
            made.append(_FakeComm(g, mode, bad_rank=1))
            return made[-1]
        t0 = time.monotonic()
        out, why = fdist.negotiate_stream_all_reduce(dist.group.WORLD, cpu, factory, timeout=1.0, check_numel=257)
        res[mode] = dict(fell_back=out is dist.group.WORLD, why=why, entered_connect=_FakeComm.entered_connect,
                         seconds=time.monotonic() - t0, aborted=made[0].aborted, is_fake=isinstance(out, _FakeComm))
        # whatever happened, the torch group still works on every rank (nobody is stuck, nobody left a collective half-entered)
        x = torch.ones(3)
        dist.all_reduce(x)
        assert float(x[0]) == world
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_stream_all_reduce_negotiation_cannot_hang():
    """ncclCommInitRank is a blocking collective — a rank that failed before it left the others inside it for
    ever.  Two gloo ranks, rank 1 failing in each phase in turn: (a) prepare fails on one rank -> NO rank enters connect(); (b) connect
    never completes -> every rank is out after the deadline and falls back; (c) the test vector disagrees on one rank -> all fall back,
    communicators aborted; (d) all fine -> the direct object on both.  Every case ends with a working torch group on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_negotiation_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank in (0, 1):
        r = got[rank]
        assert r["prepare"]["fell_back"] and r["prepare"]["why"] == "prepare" and not r["prepare"]["entered_connect"], (rank, r["prepare"])
        assert r["prepare"]["seconds"] < 5.0
        assert r["connect"]["fell_back"] and r["connect"]["why"] == "connect" and r["connect"]["seconds"] < 10.0, (rank, r["connect"])
        assert r["verify"]["fell_back"] and r["verify"]["why"] == "verify" and r["verify"]["aborted"], (rank, r["verify"])
        assert r["ok"]["is_fake"] and r["ok"]["why"] is None and not r["ok"]["aborted"], (rank, r["ok"])


# ---- the negotiation of the one-shot peer-to-peer all-reduce (fmhr_amd.dist.negotiate_peer_all_reduce) with fakes over gloo ---------
class _FakePeer:
    """stands in for PeerAllReduce: `mode` says what goes wrong on `bad_rank` — create raises (no uncached allocation / no IPC export),
    import raises (a handle that cannot be mapped), the reduction gives up inside the kernel (a wait ran into its bound), the result is
    wrong on one rank (a peer's block served out of a cache)"""

    def __init__(self, group, numel, mode, bad_rank):
        self.group, self.numel, self.mode, self.bad = group, numel, mode, dist.get_rank(group) == bad_rank
        self.closed, self.imported, self.calls = False, None, 0

    def create(self):
        if self.mode == "peer-create" and self.bad:
            raise RuntimeError("fmhr_peer_create failed (-2): hipExtMallocWithFlags(uncached): out of memory")

    def handle(self):
        return bytes([dist.get_rank(self.group) + 1] * 64)

    def import_peers(self, handles):
        assert [h[0] for h in handles] == [r + 1 for r in range(dist.get_world_size(self.group))], "every rank's handle must reach every rank"
        self.imported = len(handles)
        if self.mode == "peer-import" and self.bad:
            raise RuntimeError("fmhr_peer_import failed (-2): hipIpcOpenMemHandle: invalid argument")

    def all_reduce_(self, t, stream=None):
        dist.all_reduce(t, group=self.group)
        self.calls += 1
        if self.mode == "peer-verify" and self.bad and self.calls == 3:      # the THIRD exchange: a stale block would first show after the first
            t.add_(1.0)
        return t

    def gave_up(self):
        return self.mode == "gave-up" and self.bad

    def close(self):
        self.closed = True

    abort = close


def _peer_negotiation_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from fmhr_amd import dist as fdist
    fdist.init_from_env(backend="gloo")
    cpu = torch.device("cpu")
    res = {}
    for mode in ("peer-create", "peer-import", "peer-verify", "gave-up", "ok"):
        made = []

        def factory(g, numel, mode=mode):
            made.append(_FakePeer(g, numel, mode, bad_rank=1))
            return made[-1]
        out, why = fdist.negotiate_peer_all_reduce(dist.group.WORLD, cpu, 260, factory=factory, timeout=2.0)
        res[mode] = dict(none=out is None, why=why, closed=made[0].closed, imported=made[0].imported, calls=made[0].calls, is_fake=isinstance(out, _FakePeer))
        x = torch.ones(3)
        dist.all_reduce(x)                                          # the torch group still works on every rank
        assert float(x[0]) == world
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_peer_all_reduce_negotiation_falls_back_together():
    """The one-shot peer-to-peer all-reduce is set up phase by phase over the torch group (create + export, all_gather of the handles,
    import, four verified exchanges): whatever fails on ONE rank — the allocation, the mapping of a peer's block, a wait inside the kernel
    that ran into its bound, a wrong result in a later exchange — EVERY rank ends without the object (the caller goes on to RCCL), its
    block freed, the torch group intact; all fine: the object on both."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_peer_negotiation_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank in (0, 1):
        r = got[rank]
        assert r["peer-create"]["none"] and r["peer-create"]["why"] == "peer-create" and r["peer-create"]["imported"] is None, (rank, r["peer-create"])
        assert r["peer-import"]["none"] and r["peer-import"]["why"] == "peer-import" and r["peer-import"]["closed"] and r["peer-import"]["calls"] == 0
        assert r["peer-verify"]["none"] and r["peer-verify"]["why"] == "peer-verify" and r["peer-verify"]["closed"] and r["peer-verify"]["calls"] == 4
        assert r["gave-up"]["none"] and r["gave-up"]["why"] == "peer-verify" and r["gave-up"]["closed"]
        assert r["ok"]["is_fake"] and r["ok"]["why"] is None and not r["ok"]["closed"] and r["ok"]["imported"] == 2, (rank, r["ok"])
    # rank 0 (the good one) closed its block when rank 1's create failed
    assert got[0]["peer-create"]["closed"]


# ---- the loop's minibatches over the ranks (fmhr_amd.dist.shard_batch) and the exchange monitor ---------------------------------------
def _sub_scene(sc, idx):
    """the views `idx` of a scene, as a scene"""
    sub = dict(sc)
    for k in ("sh0", "w2c", "proj", "img", "mask", "valid_mask"):
        sub[k] = sc[k][idx].contiguous()
    sub["views"] = len(idx)
    return sub


def _loop_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    torch.set_num_threads(2)
    import ham_oracle as oracle
    from checker_scenes import prepare
    from fmhr_amd import dist as fdist
    fdist.init_from_env(backend="gloo")
    sc = prepare(oracle, (4, 64, 48, 10, 14), 11)
    perm, batch = [2, 0, 3, 1], 3                               # an epoch of the loop: a batch of 3 views, then the ragged last one
    out = []
    for k in range(0, 4, batch):
        views = perm[k:k + batch]
        part = fdist.shard_batch(len(views), sc["V"], rank, world)
        s0, s1 = part["views"]
        sh = fdist.ShardedStep(sc["V"], torch.device("cpu"))
        if part["n_local"] == 0:                                # no view of this batch: a ZERO share — but the exchange is entered all the same
            z = torch.zeros(sc["V"], 3)
            share = (torch.zeros(8), z, z, z, z)
        else:
            # (the oracle evaluates the regularisers whole: the first rank with views takes them; the kernels split them by part["reg_range"])
            share = _oracle_step(oracle, _sub_scene(sc, views[s0:s1]), 0, s1 - s0, len(views), rank == 0)
        ls, gd, ga = sh.reduce(*share)
        out.append((part["n_local"], part["reg_range"], ls.numpy().copy(), gd.numpy().copy(), ga.numpy().copy()))
    # the monitor: identical tensors pass; a rank that holds other bits, or whose exchange reports a failure, raises ON EVERY RANK
    mon = fdist.ExchangeMonitor(dist.group.WORLD, None, torch.device("cpu"))
    same = torch.arange(12, dtype=torch.float32).view(4, 3)
    verdicts = [mon.verify([same, same * 2])["ok"]]
    for case in ("bits", "gave_up", "nan", "apart", "dead"):
        t = same.clone()
        if case == "bits" and rank == 1:
            t[2, 1] = torch.nextafter(t[2, 1], torch.tensor(100.0))          # ONE bit of one element on one rank
        if case == "apart" and rank == 1:
            t = t * 1.5 + 3.0                                                 # ranks that stopped exchanging: other values altogether
        if case == "nan" and rank == 1:
            t[0, 0] = float("nan")

        class Ex:
            numel = 64

            def gave_up(self, case=case):
                return case == "gave_up" and rank == 0

            def all_reduce_(self, t, stream=None, case=case):                # the probe goes through this: "dead" hands the local data back
                if case != "dead":
                    dist.all_reduce(t)
                return t
        m = fdist.ExchangeMonitor(dist.group.WORLD, Ex(), torch.device("cpu"))
        try:
            m.verify([t], resync=[t])
            verdicts.append(("ok", m.last["same_bits"], m.last["spread_max"], m.last["resyncs"], float(t[2, 1])))
        except fdist.ExchangeError as e:
            verdicts.append(("raised", m.last["same_bits"], m.last["exchange_failed"], m.last["non_finite"], m.last["probe_ok"]))
    q.put((rank, out, verdicts))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_loop_batches_over_two_ranks_with_a_zero_view_rank(oracle):
    """the product's loop on N ranks.  An epoch of 4 views in batches of 3 on 2 ranks: the batch of 3 splits 2 + 1, the
    ragged last batch of ONE view leaves rank 1 without a view — it enters the exchange with a zero share; the MSE divisor is the BATCH's
    n * H * W on every rank.  Both batches must equal the single-process step on the batch's views.  And the exchange monitor: ONE differing
    bit on one rank is reported (same_bits False, a spread of an ulp) and repaired (rank 0's values broadcast) without an error — a
    reduction need not be bit-reproducible across ranks —; a failed exchange on one rank, a NaN on one rank, altogether different values,
    and an exchange object that hands every rank its own data back (the PROBE through the exchange path) each raise on BOTH."""
    from checker_scenes import prepare
    from fmhr_amd import dist as fdist
    from util import rel_err
    assert fdist.shard_batch(1, 99, 0, 2) == dict(views=(0, 1), n_local=1, active=1, reg_range=(0, 99))
    assert fdist.shard_batch(1, 99, 1, 2) == dict(views=(1, 1), n_local=0, active=1, reg_range=(0, 0))
    assert [fdist.shard_batch(8, 16, r, 8)["views"] for r in range(8)] == [(r, r + 1) for r in range(8)]        # the demo conf on 8 GPUs: one view each
    assert [fdist.shard_batch(2, 16, r, 8)["n_local"] for r in range(8)] == [1, 1, 0, 0, 0, 0, 0, 0]
    assert [fdist.shard_batch(2, 16, r, 8)["reg_range"] for r in range(3)] == [(0, 8), (8, 16), (0, 0)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (o, v)) for r, o, v in (q.get(timeout=240) for _ in range(2)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sc = prepare(oracle, (4, 64, 48, 10, 14), 11)
    assert [o[0] for o in got[0][0]] == [2, 1] and [o[0] for o in got[1][0]] == [1, 0]
    for b, views in enumerate(([2, 0, 3], [1])):
        ref_l, ref_gd, ref_ga, _ = _reference_step(oracle, _sub_scene(sc, views))
        for rank in (0, 1):
            _, _, ls, gd, ga = got[rank][0][b]
            assert rel_err(torch.from_numpy(ls)[:7], ref_l[:7]) < 1e-5, (b, rank)
            assert rel_err(torch.from_numpy(gd), ref_gd) < 1e-5 and rel_err(torch.from_numpy(ga), ref_ga) < 1e-5, (b, rank)
    for rank in (0, 1):
        v = got[rank][1]
        assert v[0] is True
        assert v[1][0] == "ok" and v[1][1] is False and 0 < v[1][2] < 1e-6 and v[1][3] == 1 and v[1][4] == 7.0, v      # (7.0: rank 0's value on both)
        assert v[2] == ("raised", True, True, False, True), v
        assert v[3][0] == "raised" and v[3][3] is True, v
        assert v[4] == ("raised", False, False, False, True), v              # apart: the exchange works, the parameters do not agree
        assert v[5] == ("raised", True, False, False, False), v              # dead: identical parameters (so far), but the exchange does not reduce
